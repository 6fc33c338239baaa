"""DEVELOPER-ONLY diagnostic: where does the AM-behind-AGC path lose accuracy?  Per super-frame rel-RMS of the device against the
oracle for AM / USB with and without the fast AGC on the same input."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
import pebblesdr_amd as P  # noqa: E402
from tests.signals import lcg_noise  # noqa: E402

fs, n = 2048000, 2048
fc = 100e3


def run(mode, agc, depth3):
    rx = P.ReceiverBank(fs, 1, True, False, 0, max_superframes=1)
    ref = O.Receiver(fs, n, 0)
    gm, om = (P.DM_AM, O.AM) if mode == "AM" else (P.DM_USB, O.USB)
    lo, hi = (-5000, 5000) if mode == "AM" else (300, 3000)
    rx.set_mode(0, gm); rx.set_mixer(0, fc); rx.set_bandpass(0, lo, hi)
    ref.set_mode(om); ref.set_mixer(fc); ref.set_filter(lo, hi)
    if agc:
        rx.set_agc(0, 1, 20); ref.set_agc(1, 20)
    sf = rx.superframe
    N = 6 * sf
    t = np.arange(N) / fs
    x = 0.1 * (1 + 0.5 * np.cos(2 * np.pi * 600 * t)) * (1 + depth3 * np.sin(2 * np.pi * 3.0 * t)) * np.exp(2j * np.pi * (fc + (1000 if mode == "USB" else 0)) * t) + lcg_noise(N, 9, 1e-4)
    errs = []
    for k in range(6):
        g = rx.process(x[k * sf:(k + 1) * sf])[0][0]
        r = np.concatenate([ref.process(x[k * sf + f * n:k * sf + (f + 1) * n], want_spectrum=False)[0] for f in range(sf // n)])
        errs.append(float(np.sqrt(np.mean(np.abs(g - r) ** 2)) / np.sqrt(np.mean(np.abs(r) ** 2))))
    print("%-4s agc=%d depth3=%.1f  " % (mode, agc, depth3) + " ".join("%.1e" % e for e in errs))


for mode in ("AM", "USB"):
    for agc in (0, 1):
        for d in (0.0, 0.8):
            run(mode, agc, d)

# where and how does the AGC path drift?  gain ratio device / oracle along the stream (USB, fast AGC, deep slow fading)
rx = P.ReceiverBank(fs, 1, True, False, 0, max_superframes=1)
ref = O.Receiver(fs, n, 0)
rx.set_mode(0, P.DM_USB); rx.set_mixer(0, fc); rx.set_bandpass(0, 300, 3000); rx.set_agc(0, 1, 20)
ref.set_mode(O.USB); ref.set_mixer(fc); ref.set_filter(300, 3000); ref.set_agc(1, 20)
sf = rx.superframe
N = 8 * sf
t = np.arange(N) / fs
x = 0.1 * (1 + 0.8 * np.sin(2 * np.pi * 3.0 * t)) * np.exp(2j * np.pi * (fc + 1000) * t) + lcg_noise(N, 9, 1e-4)
G, R = [], []
for k in range(8):
    G.append(rx.process(x[k * sf:(k + 1) * sf])[0][0])
    R.append(np.concatenate([ref.process(x[k * sf + f * n:k * sf + (f + 1) * n], want_spectrum=False)[0] for f in range(sf // n)]))
G, R = np.concatenate(G), np.concatenate(R)
ratio = (G * np.conj(R)).real / (np.abs(R) ** 2 + 1e-30)
for i in range(0, len(G), 1024):
    sl = slice(i, i + 1024)
    print("samples %6d..: |R| %.4f  gain ratio - 1: mean %+.2e  min %+.2e max %+.2e" % (i, np.sqrt(np.mean(np.abs(R[sl]) ** 2)), np.mean(ratio[sl] - 1), np.min(ratio[sl] - 1), np.max(ratio[sl] - 1)))
