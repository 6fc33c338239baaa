"""In-process A/B of the shared-stream bank decimator routes on configs[2]'s geometry (2.048 Msps -> 256 USB channels).
Variants are receivers created under different environment switches (read when a receiver is created), timed in alternating
rounds on one device: the whole call and the decimator kernel alone (per-kernel events), for k super-frames per call.
  python tools/ab_bank_dec.py [k ...]      e.g. 8 32 128"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pebblesdr_amd as P  # noqa: E402

VARIANTS = [
    ("k_mix_dec_fused (four-wave pipeline)", {"PEBBLEGPU_BANK_DEC": "0"}),
    ("k_mix_dec_mfma, 1 wave/SIMD", {"PEBBLEGPU_BANK_DEC": "1", "PEBBLEGPU_BANK_WAVES": "1"}),
    ("k_mix_dec_mfma, 2 waves/SIMD", {"PEBBLEGPU_BANK_DEC": "1", "PEBBLEGPU_BANK_WAVES": "2"}),
    ("k_mix_dec_mfma, 3 waves/SIMD", {"PEBBLEGPU_BANK_DEC": "1", "PEBBLEGPU_BANK_WAVES": "3"}),
]
if os.environ.get("AB_DBG"):  # timing experiments: parts of the kernel switched off (results wrong)
    VARIANTS = [("mfma W1", {"PEBBLEGPU_BANK_DEC": "1", "PEBBLEGPU_BANK_WAVES": "1"})]
DBG = os.environ.get("AB_DBG")
KEYS = ("PEBBLEGPU_BANK_DEC", "PEBBLEGPU_BANK_WAVES", "PEBBLEGPU_FUSED_L")


def make(env, fs, C, k):
    for key in KEYS:
        os.environ.pop(key, None)
    os.environ.update(env)
    rx = P.ReceiverBank(fs, C, True, False, 0, max_superframes=k)
    for c in range(C):
        rx.set_mode(c, P.DM_USB)
        rx.set_mixer(c, -960e3 + 7.5e3 * c)
        rx.set_bandpass(c, 300, 3000)
    return rx


def main():
    ks = [int(a) for a in sys.argv[1:]] or [8, 32]
    if DBG:
        os.environ["PEBBLEGPU_BANK_DBG"] = DBG
    fs, C = 2048000, 256
    for k in ks:
        rxs = [(name, make(env, fs, C, k)) for name, env in VARIANTS]
        n = k * rxs[0][1].superframe
        rng = np.random.default_rng(1)
        x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.05).astype(np.complex64)
        buf = P.DeviceBuffer.from_array(x.view(np.float32))
        for _, rx in rxs:
            for _ in range(30):
                rx.process_device(buf.ptr, n)
            rx.synchronize()
        res = {name: {"call": [], "dec": [], "fir": []} for name, _ in rxs}
        for rnd in range(5):
            for name, rx in rxs:
                rx.set_profiling(False)
                for _ in range(40):
                    rx.process_device(buf.ptr, n)
                rx.synchronize()
                res[name]["call"].append(rx.mean_ms(0, 30))
                rx.set_profiling(True)
                for _ in range(6):
                    rx.process_device(buf.ptr, n)
                rx.synchronize()
                res[name]["dec"].append(rx.mean_ms(2, 4))
                res[name]["fir"].append(rx.mean_ms(4, 4))
                res[name]["kernel"] = rx.kernel_name(2)
        for name, _ in rxs:
            r = res[name]
            call = float(np.median(r["call"]))
            print(json.dumps({"k_superframes": k, "variant": name, "kernel": r["kernel"], "call_ms": round(call, 4), "call_ms_min": round(min(r["call"]), 4),
                              "decimator_ms": round(float(np.median(r["dec"])), 4), "fastfir_ms": round(float(np.median(r["fir"])), 4),
                              "channel_Msamples_per_s": round(C * n / call / 1e3, 1)}), flush=True)
        for _, rx in rxs:
            rx.close()
        buf.free()


if __name__ == "__main__":
    main()
