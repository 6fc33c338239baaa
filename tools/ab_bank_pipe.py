"""DEVELOPER-ONLY: back-to-back calls of the configs[2] bank and the configs[3] shard without per-kernel events (ms per call), for A/B runs of
PEBBLEGPU_BANK_PIPELINE=0|1 on one box.  Usage: python tools/ab_bank_pipe.py [2|3] [calls]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pebblesdr_amd as P  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "2"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
fs, C, modes, k = (2048000, 256, [P.DM_USB], 8) if which == "2" else (100000000, 512, [P.DM_AM, P.DM_USB], 1)
k = int(sys.argv[3]) if len(sys.argv) > 3 else k  # super-frames per call
rx = P.ReceiverBank(fs, C, True, False, 0, max_superframes=k)
for c in range(C):
    rx.set_mode(c, modes[c % len(modes)])
    rx.set_mixer(c, (c - C / 2) * (0.8 * fs / C))
    rx.set_bandpass(c, 300, 3000) if modes[c % len(modes)] == P.DM_USB else rx.set_bandpass(c, -4000, 4000)
n = k * rx.superframe
rng = np.random.default_rng(1)
x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 0.05
if os.environ.get("AB_INPUT") == "bench":  # the bench's input: tones in a sample of the channels + LCG noise
    import bench
    x = bench.make_bank_input(fs, n, [(c - C / 2) * (0.8 * fs / C) for c in range(C)], 3)
buf = P.DeviceBuffer.from_array(x.view(np.float32))
for _ in range(300):
    rx.process_device(buf.ptr, n)
rx.synchronize()
best, host = 1e9, 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(calls):
        rx.process_device(buf.ptr, n)
    t1 = time.perf_counter()
    rx.synchronize()
    best = min(best, (time.perf_counter() - t0) / calls * 1e3)
    host = min(host, (t1 - t0) / calls * 1e3)
print("configs[%s] k=%d PEBBLEGPU_BANK_PIPELINE=%s: %.4f ms per call, the host queues one in %.4f ms (%s)" % (which, k, os.environ.get("PEBBLEGPU_BANK_PIPELINE", "unset"), best, host, rx.kernel_name(2)))
