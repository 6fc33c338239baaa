"""CPU tier: the N>1 launch path of bench.py with world_size 2 over gloo -- the exact rendezvous the driver uses
(torch.distributed.run, 127.0.0.1), the barriers around the timed region, the max-over-ranks of the elapsed time
and the whole-job aggregation.  Channels are independent shards: there is no data-path collective
to test, only that every stream is owned by exactly one rank and that the slowest rank sets the time."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_control_plane_over_gloo():
    steps = 5
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "0",
           "--control-plane-only", "--superframes", "4"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["scaling"] == "weak"
    assert d["streams_by_rank"] == [[0], [1]]
    # the banks shard the same way: rank r owns configs[3] channels [512 r, 512 r + 511] and configs[4] streams [128 r, 128 r + 127]
    assert d["configs3_channel_range_by_rank"] == [[0, 511], [512, 1023]]
    assert d["configs4_stream_range_by_rank"] == [[0, 127], [128, 255]]
    # rank 1 sleeps 4 ms per step, rank 0 2 ms: the reported time is the slower rank's
    assert d["ms_per_step"] >= 4.0
    # whole-job aggregate: both ranks' samples over the max time
    want = 2 * steps * 4 * 131072 / (d["ms_per_step"] * 1e-3 * steps) / 1e6
    assert abs(d["value"] - want) / want < 0.02


def test_single_rank_needs_no_process_group():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--control-plane-only",
                        "--superframes", "1"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["streams_by_rank"] == [[0]]


def test_more_ranks_than_devices_is_refused():
    """bench.py must not put two ranks on one GPU (or run at all without one): non-zero exit, no JSON line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--headline-only", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
