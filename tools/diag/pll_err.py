"""DEVELOPER-ONLY diagnostic: per-frame rel-RMS of the NFM and SAM demodulators against the oracle at the four demodulator rates,
and of the oracle against ITSELF when its band-pass output is rounded to fp32 (how much of the gap is the float loop state)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
import pebblesdr_amd as P  # noqa: E402
from tests.signals import lcg_noise  # noqa: E402


def rr(a, b):
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2)) / max(np.sqrt(np.mean(np.abs(b) ** 2)), 1e-30))


for fs in (2048000, 2400000, 5000000, 25000000):
    for mode in ("NFM", "SAM"):
        rx = P.ReceiverBank(fs, 1, True, False, 0, max_superframes=1)
        rate = int(rx.info.demod_rate_int)
        stages = sum(int(np.log2(st)) for _, st in rx.chain())
        fc = 0.17 * fs
        lo, hi = (-4000, 4000) if mode == "NFM" else (-5000, 5000)
        rx.set_mode(0, P.DM_FMN if mode == "NFM" else P.DM_SAM); rx.set_mixer(0, fc); rx.set_bandpass(0, lo, hi)
        sf = rx.superframe
        K = 6
        t = np.arange(K * sf) / fs
        if mode == "NFM":
            x = 0.1 * np.exp(1j * (2 * np.pi * fc * t + 2.0 * np.sin(2 * np.pi * 800 * t)))
        else:
            x = 0.1 * (1 + 0.4 * np.cos(2 * np.pi * 900 * t)) * np.exp(2j * np.pi * (fc + 25.0) * t)
        x = x + lcg_noise(K * sf, 5, 1e-4)
        g = np.concatenate([rx.process(x[k * sf:(k + 1) * sf])[0] for k in range(K)], axis=1)[0]
        mix = O.Mixer(fs); mix.set_frequency(fc)
        dec = O.Decimator(fs, 30000)
        z = np.concatenate([dec.process(mix.process(x[k * sf:(k + 1) * sf])) for k in range(K)]) * 10 ** (2 * stages / 20.0)
        ff = O.FastFIR(); ff.setup(lo, hi, 0, rate)
        y = np.concatenate([ff.process(z[k:k + 2048]) for k in range(0, len(z), 2048)])
        mk = (lambda: O.DemodNFM(rate)) if mode == "NFM" else (lambda: O.DemodSAM(rate))
        d1, d2 = mk(), mk()
        r = np.concatenate([d1.process(y[k:k + 2048]) for k in range(0, len(y), 2048)])
        r32 = np.concatenate([d2.process(y[k:k + 2048].astype(np.complex64).astype(np.complex128)) for k in range(0, len(y), 2048)])
        if mode == "SAM":
            f = lambda v: (v.real + v.imag) / 2
        else:
            f = lambda v: v
        print("%-3s fs %8d rate %6d  device vs oracle: %s | oracle(fp32-rounded input) vs oracle: %s" % (
            mode, fs, rate, " ".join("%.1e" % rr(f(g[k * 2048:(k + 1) * 2048]), f(r[k * 2048:(k + 1) * 2048])) for k in range(K)),
            " ".join("%.1e" % rr(f(r32[k * 2048:(k + 1) * 2048]), f(r[k * 2048:(k + 1) * 2048])) for k in range(K))))
