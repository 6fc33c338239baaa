"""DEVELOPER-ONLY: bench.py's configs[2] / configs[3] legs alone in a fresh process (the same code path: bench.run_bank), to compare with
the numbers the full bench run reports for them after its headline legs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pebblesdr_amd as P  # noqa: E402


class A:
    steps, warmup = 20, 3


which = sys.argv[1] if len(sys.argv) > 1 else "2"
if which == "2":
    out = bench.run_bank(P, "configs[2]", 2_048_000, 256, [P.DM_USB], 8, 0, 1, 0, lambda: None, A, None)
else:
    out = bench.run_bank(P, "configs[3] shard", 100_000_000, bench.C3_PER_GPU, [P.DM_AM, P.DM_USB], 1, 0, 1, 0, lambda: None, A, None)
print(json.dumps({k: out[k] for k in ("workload", "ms_per_step", "steps", "settle_steps")}))
