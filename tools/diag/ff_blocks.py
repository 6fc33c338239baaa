"""DEVELOPER-ONLY: the stream bank's band-pass with 2048- and 4096-point overlap-save blocks (1025 taps either way: the same linear
convolution, 3072 instead of 1024 new samples per block).  Measured: 0.2179 against 0.2133 ms for 33 M samples -- k_fastfir<4096> (workgroup
transform) spends per point what it saves in points."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pebblesdr_amd as P
S, N, F = 128, 12288, 21
rng = np.random.default_rng(1)
x = (rng.standard_normal((S, F * N)) + 1j * rng.standard_normal((S, F * N))).astype(np.complex64) * 0.1
buf = P.DeviceBuffer.from_array(x.view(np.float32))
for fft in (2048, 4096):
    sb = P.StreamBank(2.0e6, S, frame=N, spectrum_bins=16384, max_frames=F, fastfir_fft=fft, fastfir_taps=1025)
    for c in range(S):
        sb.set_bandpass(c, -50e3, 50e3)
    for _ in range(3):
        sb.process_device(buf.ptr, F * N, what=1)
    sb.synchronize()
    bp = []
    for _ in range(20):
        sb.process_device(buf.ptr, F * N, what=1)
        bp.append(sb.last_ms(1))
    print(fft, "bandpass ms", float(np.median(bp)), "ns per sample", float(np.median(bp)) * 1e6 / (S * F * N))
    sb.close()
