"""Workload for the SQ-counter passes on the display transform (run as `rocprofv3 --pmc ... -- python3 tools/pmc_spectrum.py`,
the interpreter directly after `--`): three single-stream calls of the bench workload (per-kernel profiling keeps every
kernel of a call on one stream, so each dispatch is counted alone).  tools/pmc_summary.py prints the per-kernel averages."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pebblesdr_amd as P  # noqa: E402

bins = int(sys.argv[1]) if len(sys.argv) > 1 else bench.BINS
rx = P.ReceiverBank(bench.FS, n_channels=1, shared_input=True, wfm=True, spectrum_bins=bins, max_superframes=256)
rx.set_mixer(0, bench.MIX_HZ)
rx.set_profiling(True)
n = 256 * rx.superframe
buf = P.DeviceBuffer.from_array(P.binding.to_f32_iq(bench.make_input(n, 1000)))
for _ in range(3):
    rx.process_device(buf.ptr, n)
rx.synchronize()
print("done", n)
