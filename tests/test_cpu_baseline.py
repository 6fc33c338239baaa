"""CPU tier: the multithreaded CPU-baseline driver (oracle/cpu_baseline.cpp, what bench.py's cpu_baseline leg times) computes
the same chain as the oracle it is built from -- its -O3 -march=native build against the -O2 -ffp-contract=off oracle on the
same regenerated input -- and reports the fields bench.py forwards."""
import json
import os
import subprocess

import numpy as np

import oracle as O
from tests.signals import lcg_noise

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_build", "cpu_baseline")


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_build/cpu_baseline"])


def test_native_build_matches_the_oracle_on_the_wfm_chain():
    build()
    out = subprocess.check_output([EXE, "1", "0", "1", "check"]).decode().split()
    got = np.array(out, dtype=np.float64).reshape(-1, 2)
    got = got[:, 0] + 1j * got[:, 1]
    fs, n = 20_000_000, 64 * 2048
    t = np.arange(n) / fs
    x = 0.5 * np.exp(1j * (2 * np.pi * 1.0e6 * t + 75.0 * np.sin(2 * np.pi * 1000 * t))) + lcg_noise(n, 99, 1e-2)
    x = (np.round(x.real * 128) + 1j * np.round(x.imag * 128)) / 128.0
    ref = O.Receiver(fs, 2048, 8192)
    ref.set_mode(O.FMM)
    ref.set_mixer(1.0e6)
    audio = None
    for f in range(64):
        a, _ = ref.process(x[f * 2048:(f + 1) * 2048])
        if len(a):
            audio = a
            break
    assert audio is not None and len(got) == 16
    assert np.abs(got - audio[:16]).max() <= 1e-9 * max(1.0, np.abs(audio).max())


def test_reports_one_thread_and_all_threads():
    build()
    for wl in (1, 2, 3, 4):
        d = json.loads(subprocess.check_output([EXE, str(wl), "0.2", "2"]).decode())
        assert d["workload"] == wl and d["threads"] == 2 and d["dtype"] == "f64"
        assert d["msamples_per_s_1thread"] > 0 and d["msamples_per_s_all_threads"] > 0
        assert d["cpu_model"]
