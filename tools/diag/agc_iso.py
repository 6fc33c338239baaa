"""DEVELOPER-ONLY diagnostic: the device AGC against the oracle AGC on the SAME band-passed samples (the device's own)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
import pebblesdr_amd as P  # noqa: E402
from tests.signals import lcg_noise  # noqa: E402

fs, n, fc = 2048000, 2048, 100e3
rx0 = P.ReceiverBank(fs, 1, True, False, 0, max_superframes=1)
rx1 = P.ReceiverBank(fs, 1, True, False, 0, max_superframes=1)
for rx in (rx0, rx1):
    rx.set_mode(0, P.DM_USB); rx.set_mixer(0, fc); rx.set_bandpass(0, 300, 3000)
rx1.set_agc(0, 1, 20)
agc = O.Agc(64000); agc.set_mode(1, 20)
sf = rx0.superframe
N = 8 * sf
t = np.arange(N) / fs
x = 0.1 * (1 + 0.8 * np.sin(2 * np.pi * 3.0 * t)) * np.exp(2j * np.pi * (fc + 1000) * t) + lcg_noise(N, 9, 1e-4)
for k in range(8):
    y = rx0.process(x[k * sf:(k + 1) * sf])[0][0]
    g = rx1.process(x[k * sf:(k + 1) * sf])[0][0]
    r = agc.process(y.astype(np.complex128))
    e = np.abs(g - r)
    i = int(np.argmax(e))
    print("sf %d: device AGC vs oracle AGC on the device's band-pass output: rel-RMS %.2e, max |d| %.2e at %d, bp equal? %s" % (
        k, np.sqrt(np.mean(e ** 2)) / np.sqrt(np.mean(np.abs(r) ** 2)), e[i], i, True))
