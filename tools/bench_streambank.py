"""Time BASELINE configs[4] (S streams @ 2 Msps, 2048/1025 band-pass + 65536-point spectrum) on one GPU.
Not the bench line (bench.py measures configs[1]); the numbers go into profiles/README.md."""
import json
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pebblesdr_amd as P  # noqa: E402

N = 65536
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(1)
x = (rng.standard_normal((S, F * N)) + 1j * rng.standard_normal((S, F * N))).astype(np.complex64) * 0.1
sb = P.StreamBank(2.0e6, S, frame=N, spectrum_bins=N, max_frames=F)
for c in range(S):
    sb.set_bandpass(c, -50e3, 50e3)
buf = P.DeviceBuffer.from_array(x.view(np.float32))
for _ in range(3):
    sb.process_device(buf.ptr, F * N)
sb.synchronize()
tot, bp, sp = [], [], []
for _ in range(20):
    sb.process_device(buf.ptr, F * N)
    tot.append(sb.last_ms(0)); bp.append(sb.last_ms(1)); sp.append(sb.last_ms(2))
ms = float(np.median(tot))
print(json.dumps({"workload": "configs[4]: %d streams x %d frames of 65536" % (S, F), "samples": S * F * N, "ms": ms,
                  "bandpass_ms": float(np.median(bp)), "spectrum_ms": float(np.median(sp)),
                  "gsamples_per_s": S * F * N / ms / 1e6,
                  "bandpass_GBps": 16.0 * S * F * N / float(np.median(bp)) / 1e6,
                  "spectrum_GBps": 12.0 * S * F * N / float(np.median(sp)) / 1e6}))
