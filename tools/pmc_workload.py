"""Workload for the HBM-traffic PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`):
two calibration copies of known size (16-byte and 8-byte lanes, 1 GiB each way) followed by three steps of the
bench workload (bench.py's configs[1] batch).  tools/parse_traffic.py turns the two CSVs into profiles/*.json."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pebblesdr_amd as P  # noqa: E402

P.binding.probe_copy_gbps(16, 1 << 30, 2)
P.binding.probe_copy_gbps(8, 1 << 30, 2)
rx = P.ReceiverBank(bench.FS, n_channels=1, shared_input=True, wfm=True, spectrum_bins=bench.BINS, max_superframes=256)
rx.set_mixer(0, bench.MIX_HZ)
n = 256 * rx.superframe
buf = P.DeviceBuffer.from_array(P.binding.to_f32_iq(bench.make_input(n, 1000)))
for _ in range(3):
    rx.process_device(buf.ptr, n)
rx.synchronize()
print("done", n)
