"""Time the other BASELINE configs on one GPU (not the bench line; numbers go to profiles/README.md).
  configs[2]: 2.048 Msps shared stream -> 256 tuned SSB channels (mixer + decimate + FastFIR + pass-through demod)
  configs[3] shard: 100 Msps shared stream -> 512 AM/SSB channels"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pebblesdr_amd as P  # noqa: E402


def run(name, fs, C, modes, k):
    rx = P.ReceiverBank(fs, C, True, False, 0, max_superframes=k)
    for c in range(C):
        rx.set_mode(c, modes[c % len(modes)])
        rx.set_mixer(c, (c - C / 2) * (0.8 * fs / C))
        rx.set_bandpass(c, 300, 3000) if modes[c % len(modes)] == P.DM_USB else rx.set_bandpass(c, -4000, 4000)
    n = k * rx.superframe
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 0.05
    buf = P.DeviceBuffer.from_array(x.view(np.float32))
    for _ in range(2):
        rx.process_device(buf.ptr, n)
    rx.synchronize()
    rx.set_profiling(True)
    ms = []
    parts = {w: [] for w in (2, 3, 4, 5)}
    for _ in range(6):
        rx.process_device(buf.ptr, n)
        ms.append(rx.last_ms(0))
        for w in parts:
            parts[w].append(rx.last_ms(w))
    t = float(np.median(ms))
    print(json.dumps({"workload": name, "fs": fs, "channels": C, "input_samples": n, "chain": rx.chain(), "D": rx.D, "ms": t,
                      "input_Msps": n / t / 1e3, "channel_Msps": n * C / t / 1e3,
                      "mix_dec1_ms": float(np.median(parts[2])), "cascade_ms": float(np.median(parts[3])),
                      "fastfir_ms": float(np.median(parts[4])), "demod_ms": float(np.median(parts[5]))}))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "2"
    if which == "2":
        run("configs[2]: 2.048 Msps -> 256 SSB channels", 2048000, 256, [P.DM_USB], 8)
    else:
        run("configs[3] shard: 100 Msps -> 512 AM/SSB channels", 100000000, 512, [P.DM_AM, P.DM_USB], 1)
