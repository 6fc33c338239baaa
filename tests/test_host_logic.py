"""CPU tier: host logic that needs no device -- input generators, sharding arithmetic, bench JSON contract."""
import json
import os
import subprocess
import sys

import numpy as np

from tests.signals import lcg_noise, lcg_uniform, tones

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lcg_matches_scalar_recurrence():
    s, want = 7, []
    for _ in range(10000):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        want.append(s / 2 ** 32)
    assert np.array_equal(lcg_uniform(10000, 7), np.array(want))
    z = lcg_noise(5, 3, 2.0)
    u = lcg_uniform(10, 3)
    assert np.allclose(z, 2.0 * ((u[0::2] - 0.5) + 1j * (u[1::2] - 0.5)))


def test_tones_phase_continuity():
    a = tones(1e6, 100, [(1.0, 12345.0)])
    b = tones(1e6, 50, [(1.0, 12345.0)], n0=50)
    assert np.allclose(a[50:], b)


def test_bench_shard_arithmetic():
    sys.path.insert(0, ROOT)
    import bench
    # weak scaling: every rank owns its own stream(s); nothing is exchanged
    for world in (1, 2, 4, 8):
        shards = [bench.shard_streams(world, r, per_rank=1) for r in range(world)]
        flat = [s for sh in shards for s in sh]
        assert flat == list(range(world))
    assert bench.aggregate_msps(samples_per_rank=10_000_000, world=4, seconds=0.5) == 80.0
