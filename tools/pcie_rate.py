"""DEVELOPER-ONLY: what the bench workload runs at when the host hands over pageable host buffers (PCIe both ways) -- the
figure DESIGN.md section 5 quotes next to the HBM-resident rate.  Never the bench value."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
import pebblesdr_amd as P
rx = P.ReceiverBank(bench.FS, n_channels=1, shared_input=True, wfm=True, spectrum_bins=bench.BINS, max_superframes=256)
rx.set_mixer(0, bench.MIX_HZ)
n = 256 * rx.superframe
x = P.binding.to_f32_iq(bench.make_input(n, 1000))
raw = np.clip(np.round(np.asarray(x).view(np.float32).ravel() * 127), -128, 127).astype(np.int8)  # HackRF-shape pairs
dbuf = P.DeviceBuffer(x.nbytes); rbuf = P.DeviceBuffer(raw.nbytes)
for _ in range(2):
    dbuf.upload(x); rx.process_device(dbuf.ptr, n); rx.synchronize()
def t(f, k=5):
    best = 1e9
    for _ in range(k):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best
def f_float():
    dbuf.upload(x); rx.process_device(dbuf.ptr, n); a = rx.audio()
def f_raw():
    rbuf.upload(raw); rx.process_raw_device(rbuf.ptr, n, 0, 0, 1.0); a = rx.audio()
def f_raw_spec():
    rbuf.upload(raw); rx.process_raw_device(rbuf.ptr, n, 0, 0, 1.0); a = rx.audio(); s = rx.spectrum()
for name, f in (("float2 in (268 MB H2D) + audio out", f_float), ("int8 in (67 MB H2D) + audio out", f_raw), ("int8 in + audio + full spectrum out (537 MB D2H)", f_raw_spec)):
    s = t(f)
    print("%-55s %.2f ms  %.1f Gsamples/s" % (name, s * 1e3, n / s / 1e9))
