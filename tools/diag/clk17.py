import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["PEBBLEGPU_BANK_CLK"] = "0"
import pebblesdr_amd as P
fs, C = 2400000, 32
rx = P.ReceiverBank(fs, C, True, False, 0, max_superframes=1)
for c in range(C):
    rx.set_mode(c, P.DM_USB); rx.set_mixer(c, (-0.4 + 0.8 * (c + 0.5) / C) * fs); rx.set_bandpass(c, 300, 3000)
n = rx.superframe
print("chain", rx.chain(), "superframe", n)
rng = np.random.default_rng(1)
x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.05).astype(np.complex64)
buf = P.DeviceBuffer.from_array(x.view(np.float32))
rx.process_device(buf.ptr, n); rx.synchronize()
os.environ["PEBBLEGPU_BANK_CLK"] = "1"
for _ in range(2):
    rx.process_device(buf.ptr, n)
rx.synchronize()
a = rx.audio()
print("nonzero fraction per 256 outputs:", [float((np.abs(a[0][i:i+256]) > 0).mean()) for i in range(0, 2048, 256)])
