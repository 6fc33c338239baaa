"""Workload for counter passes on the BASELINE configs[4] shard (run as `rocprofv3 --pmc ... -- python3 tools/pmc_streambank.py`):
128 streams @ 2 Msps x 4 frames of 65536 samples per call, FastFIR 2048/1025 + 65536-point spectrum, four calls -- the geometry
bench.py's configs[4] leg times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pebblesdr_amd as P  # noqa: E402

if os.environ.get("PMC_CALIBRATE"):  # the traffic passes: two copies of known size first (tools/parse_traffic.py scales FETCH_SIZE by them)
    P.binding.probe_copy_gbps(16, 1 << 30, 2)
    P.binding.probe_copy_gbps(8, 1 << 30, 2)
S, N, F = 128, 65536, 4
rng = np.random.default_rng(4)
x = np.empty((S, F * N), dtype=np.complex64)
for s in range(S):
    x[s] = (rng.standard_normal(F * N, dtype=np.float32) + 1j * rng.standard_normal(F * N, dtype=np.float32)) * np.float32(0.1)
sb = P.StreamBank(2.0e6, S, frame=N, spectrum_bins=N, max_frames=F)
for c in range(S):
    sb.set_bandpass(c, -50e3, 50e3)
buf = P.DeviceBuffer.from_array(x.view(np.float32))
for _ in range(4):
    sb.process_device(buf.ptr, F * N)
sb.synchronize()
print("done", S * F * N)
