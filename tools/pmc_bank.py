"""Workload for counter passes on the shared-stream banks (run as `rocprofv3 --pmc ... -- python3 tools/pmc_bank.py [2|3]`):
BASELINE configs[2] (256 USB channels @ 2.048 Msps, 8 super-frames) or the configs[3] shard (512 AM/USB @ 100 Msps), three calls."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pebblesdr_amd as P  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "2"
if os.environ.get("PMC_CALIBRATE"):  # the traffic passes: two copies of known size first (tools/parse_traffic.py scales FETCH_SIZE by them)
    P.binding.probe_copy_gbps(16, 1 << 30, 2)
    P.binding.probe_copy_gbps(8, 1 << 30, 2)
fs, C, modes, k = (2048000, 256, [P.DM_USB], 8) if which == "2" else (100000000, 512, [P.DM_AM, P.DM_USB], 1)
rx = P.ReceiverBank(fs, C, True, False, 0, max_superframes=k)
for c in range(C):
    rx.set_mode(c, modes[c % len(modes)])
    rx.set_mixer(c, (c - C / 2) * (0.8 * fs / C))
    rx.set_bandpass(c, 300, 3000) if modes[c % len(modes)] == P.DM_USB else rx.set_bandpass(c, -4000, 4000)
n = k * rx.superframe
rng = np.random.default_rng(1)
x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 0.05
buf = P.DeviceBuffer.from_array(x.view(np.float32))
for _ in range(4):
    rx.process_device(buf.ptr, n)
rx.synchronize()
print("done", n)
