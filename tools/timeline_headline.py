"""DEVELOPER-ONLY workload for a kernel-trace of the settled headline loop: 150 back-to-back steps (run under rocprofv3
--kernel-trace, then tools/trace_timeline.py <dir> 14 for the last two steps)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pebblesdr_amd as P  # noqa: E402

rx = P.ReceiverBank(bench.FS, n_channels=1, shared_input=True, wfm=True, spectrum_bins=bench.BINS, max_superframes=256)
rx.set_mixer(0, bench.MIX_HZ)
n = 256 * rx.superframe
buf = P.DeviceBuffer.from_array(P.binding.to_f32_iq(bench.make_input(n, 1000)))
for _ in range(150):
    rx.process_device(buf.ptr, n)
rx.synchronize()
print("done", n)
