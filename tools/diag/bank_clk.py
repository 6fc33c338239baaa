"""Clock counts of k_mix_dec_mfma's waves on configs[2]'s geometry (PEBBLEGPU_BANK_CLK=1 makes the library print them per launch)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PEBBLEGPU_BANK_CLK"] = "0"
import pebblesdr_amd as P  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fs, C = (int(sys.argv[2]) if len(sys.argv) > 2 else 2048000), 256
rx = P.ReceiverBank(fs, C, True, False, 0, max_superframes=k)
for c in range(C):
    rx.set_mode(c, P.DM_USB); rx.set_mixer(c, (-0.45 + 0.9 * c / C) * fs); rx.set_bandpass(c, 300, 3000)
n = k * rx.superframe
rng = np.random.default_rng(1)
x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.05).astype(np.complex64)
buf = P.DeviceBuffer.from_array(x.view(np.float32))
for _ in range(200):
    rx.process_device(buf.ptr, n)
rx.synchronize()
os.environ["PEBBLEGPU_BANK_CLK"] = "1"
for _ in range(3):
    rx.process_device(buf.ptr, n)
rx.synchronize()
