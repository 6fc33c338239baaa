"""Where the one-kernel bank decimator differs from the two-kernel route: error per 16 outputs of a few channels (diagnosis)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pebblesdr_amd as P  # noqa: E402
from tests.signals import tones, lcg_noise  # noqa: E402


def run(env, C, ksf, calls, fs=2048000):
    for k in ("PEBBLEGPU_BANK_DEC", "PEBBLEGPU_NO_FUSED_DEC", "PEBBLEGPU_BANK_WAVES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rx = P.ReceiverBank(fs, C, True, False, 0, max_superframes=ksf)
    for i in range(C):
        rx.set_mode(i, P.DM_USB); rx.set_mixer(i, (-0.4 + 0.8 * (i + 0.5) / C) * fs); rx.set_bandpass(i, 300, 3000)
    sf = rx.superframe
    x = tones(fs, (1 + calls * ksf) * sf, [(0.003, (-0.4 + 0.8 * (c + 0.5) / C) * fs + 1000.0 + 3.1 * c, 0.3 * c) for c in range(C)]) + lcg_noise((1 + calls * ksf) * sf, 3, 1e-3)
    out = [rx.process(x[:sf])[0]]
    names = [rx.kernel_name(2)]
    for k in range(calls):
        out.append(rx.process(x[sf + k * ksf * sf:sf + (k + 1) * ksf * sf])[0])
        names.append(rx.kernel_name(2))
    rx.close()
    return np.concatenate(out, axis=1), names


if __name__ == "__main__":
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    ksf = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    fs = int(sys.argv[3]) if len(sys.argv) > 3 else 2048000
    a, na = run({"PEBBLEGPU_NO_FUSED_DEC": "1"}, C, ksf, 2, fs)
    b, nb = run({"PEBBLEGPU_BANK_DEC": "1"}, C, ksf, 2, fs)
    print(na, nb)
    ref = np.sqrt(np.mean(np.abs(a) ** 2))
    for c in (0, C - 1):
        e = np.abs(a[c] - b[c]).reshape(-1, 16).max(axis=1) / ref
        bad = np.nonzero(e > 1e-5)[0]
        print("channel", c, "blocks of 16 outputs:", len(e), "bad:", len(bad), "first/last bad:", bad[:3].tolist(), bad[-3:].tolist(), "max err", float(e.max()), "at", int(np.argmax(e)))
        print("   log10 err per block from 120:", " ".join("%.0f" % np.log10(max(v, 1e-9)) for v in e[120:200]))
        z = (np.abs(b[c]).reshape(-1, 16).max(axis=1) == 0)
        print("   all-zero blocks of 16 in b:", np.nonzero(z)[0][:60].tolist())
        d = np.abs(a[c] - b[c])[2048 + 1024:].reshape(-1, 16) / ref   # by position inside a block of 16 outputs, past the first call and the filter's memory
        print("   mean err by output position mod 16:", " ".join("%.1e" % v for v in d.mean(axis=0)))
