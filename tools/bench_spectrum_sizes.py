"""Time the display transform alone for 2048 / 4096 / 8192 bins on the bench batch (16384 frames of 2048 samples)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pebblesdr_amd as P  # noqa: E402

fs = 20_000_000
for bins in ([int(a) for a in sys.argv[1:]] or (2048, 4096, 8192, 16384, 32768)):
    rx = P.ReceiverBank(fs, 1, True, True, bins, max_superframes=256)
    n = 256 * rx.superframe
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 0.1
    buf = P.DeviceBuffer.from_array(x.view(np.float32))
    rx.set_profiling(True)  # one stream per call: the transform is timed alone
    for _ in range(3):
        rx.process_device(buf.ptr, n)
    rx.synchronize()
    ms = []
    for _ in range(10):
        rx.process_device(buf.ptr, n)
        ms.append(rx.last_ms(1))
    t = float(np.median(ms))
    alg = (n // 2048) * (8 * 2048 + 4 * bins)
    print(json.dumps({"bins": bins, "frames": n // 2048, "kernel": rx.kernel_name(1), "spectrum_ms": t, "algorithmic_GBps": alg / t / 1e6}))
    del rx, buf
