"""DEVELOPER-ONLY: socket power while the library's streaming-copy probe runs (what a byte across HBM costs)."""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pebblesdr_amd as P  # noqa: E402


def power():
    out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        if "(W):" in line:
            return float(line.split("(W):")[1])
    return -1.0


print("idle %.0f W" % power())
for lanes in (16, 8):
    res = {}

    def work():
        res["gbps"] = P.binding.probe_copy_gbps(lanes, 1 << 30, 14000)  # 14000 copies of 1 GiB each way

    t = threading.Thread(target=work)
    t.start()
    time.sleep(2.0)
    w = [power() for _ in range(3)]
    t.join()
    print("copy with %d-byte lanes: %s GB/s (read + write), %s W" % (lanes, res["gbps"], w))
