"""DEVELOPER-ONLY diagnostic: first point where the AGC detector's state differs between the oracle's and the device's band-pass output."""
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
rx0.set_mode(0, P.DM_USB); rx0.set_mixer(0, fc); rx0.set_bandpass(0, 300, 3000)
ref = O.Receiver(fs, n, 0)
ref.set_mode(O.USB); ref.set_mixer(fc); ref.set_filter(300, 3000)
sf = rx0.superframe
N = 8 * sf
t = np.arange(N) / fs
x = 0.1 * (1 + 0.8 * np.sin(2 * np.pi * 3.0 * t)) * np.exp(2j * np.pi * (fc + 1000) * t) + lcg_noise(N, 9, 1e-4)
yd = np.concatenate([rx0.process(x[k * sf:(k + 1) * sf])[0][0] for k in range(8)]).astype(np.complex128)
yo = np.concatenate([ref.process(x[f * n:(f + 1) * n], want_spectrum=False)[0] for f in range(N // n)])
print("band-pass outputs: rel-RMS %.2e, max |d| %.2e at %d" % (np.sqrt(np.mean(np.abs(yd - yo) ** 2)) / np.sqrt(np.mean(np.abs(yo) ** 2)), np.abs(yd - yo).max(), int(np.argmax(np.abs(yd - yo)))))


def sim(y):
    rate = 64000.0
    W = int(rate * np.float32(.018))
    ar = 1 - np.exp(-1 / (rate * np.float32(.002))); af = 1 - np.exp(-1 / (rate * np.float32(.005)))
    dr = 1 - np.exp(-1 / (rate * 100 * .001 * np.float32(.3))); df = 1 - np.exp(-1 / (rate * 100 * .001))
    magbuf = np.full(W, -16.0); pos = 0; peak = -16.0; att = -5.0; dec_ = -5.0
    out = np.zeros((len(y), 4))
    m = np.log10(np.maximum(np.abs(y.real), np.abs(y.imag)) + float(np.float32(1e-8)))
    for i, mag in enumerate(m):
        tmp = magbuf[pos]; magbuf[pos] = mag; pos = (pos + 1) % W
        rs = 0
        if mag > peak:
            peak = mag
        elif tmp == peak:
            peak = magbuf.max(); rs = 1
        att = (1 - ar) * att + ar * peak if peak > att else (1 - af) * att + af * peak
        dec_ = (1 - dr) * dec_ + dr * peak if peak > dec_ else (1 - df) * dec_ + df * peak
        out[i] = (peak, att, dec_, rs)
    return out, m


(a, ma), (b, mb) = sim(yo), sim(yd)
d = np.abs(a - b)
for name, col in (("peak", 0), ("attack", 1), ("decay", 2)):
    i = 3000 + int(np.argmax(d[3000:, col])); print(name, "max diff %.2e at %d" % (d[i, col], i))
bad = np.nonzero(d[3000:, 0] > 1e-6)[0]
print("samples with peak diff > 1e-6:", len(bad), (bad[:5] + 3000) if len(bad) else "")
if len(bad):
    i0 = bad[0] + 3000
    print(a[i0 - 2:i0 + 3]); print(b[i0 - 2:i0 + 3])
    print("mags oracle", ma[i0 - 2:i0 + 3], "device", mb[i0 - 2:i0 + 3])
    j = i0 - 1152 + 1 + int(np.argmax(ma[i0 - 1151:i0 + 1])); print("oracle window max at", j, ma[j], "device there", mb[j], " y oracle", yo[j], "device", yd[j])
    j = i0 - 1152 + 1 + int(np.argmax(mb[i0 - 1151:i0 + 1])); print("device window max at", j, mb[j], "oracle there", ma[j], " y oracle", yo[j], "device", yd[j])
