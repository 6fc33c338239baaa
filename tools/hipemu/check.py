#!/usr/bin/env python3
"""DEVELOPER-ONLY: run libpebblegpu's kernels under the fiber emulator and compare with the oracle.

Usage: tools/hipemu/build_emu.sh && python tools/hipemu/check.py [case ...]
Catches indexing / halo / scan mistakes without a GPU.  Says nothing about speed and is not a test
tier: the graded parity tests are tests/test_parity_*.py (-m gpu) against the real HIP build.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from pebblesdr_amd import binding as B  # noqa: E402
from pebblesdr_amd import steps as St  # noqa: E402
from tests.signals import lcg_noise, tones  # noqa: E402

L = B.load_library(os.path.join(ROOT, "tools", "hipemu", "_build", "libpebblegpu_emu.so"))


def rel_rms(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2)) / max(np.sqrt(np.mean(np.abs(b) ** 2)), 1e-12))


def report(name, err, tol):
    print("%-44s %.3e  (tol %.1e) %s" % (name, err, tol, "ok" if err <= tol else "FAIL"))
    return err <= tol


def case_spectrum():
    ok = True
    fs = 2.048e6
    for bins in (2048, 4096, 8192):
        x = tones(fs, 3 * 2048, [(10 ** (-10 / 20), 123456.7), (10 ** (-40 / 20), -700001.3)]) + lcg_noise(3 * 2048, 1, 1e-4)
        ref = O.Spectrum(bins, 2048)
        sp = St.Spectrum(bins, fs, 2048, lib=L)
        for f in range(3):
            r = ref.process(x[f * 2048:(f + 1) * 2048])
            g, _ = sp.fftSpectrum(x[f * 2048:(f + 1) * 2048])
            m = r > -110
            ok &= report("spectrum step bins=%d frame %d max|dB|" % (bins, f), float(np.abs(g - r)[m].max()), 0.05)
    return ok


def case_spectrum_batch():
    """several frames per call (frame chains, parking/prefetch, previous-frame average carried in registers and across calls)"""
    ok = True
    fs, n = 20_000_000, 2048
    for bins in (2048, 4096, 8192):
        rx = B.ReceiverBank(fs, 1, True, True, bins, max_superframes=1, lib=L)
        ref = O.Spectrum(bins, 2048)
        sf = rx.superframe
        x = tones(fs, 2 * sf, [(10 ** (-10 / 20), 1234567.0), (10 ** (-40 / 20), -7000013.0)]) + lcg_noise(2 * sf, 1, 1e-4)
        for call in range(2):
            _, sp = rx.process(x[call * sf:(call + 1) * sf])
            worst = 0.0
            for f in range(sf // n):
                r = ref.process(x[call * sf + f * n:call * sf + (f + 1) * n])
                if call or f:
                    m = r > -110
                    worst = max(worst, float(np.abs(sp[0, f] - r)[m].max()))
            ok &= report("spectrum batch bins=%d call %d (%d frames) max|dB|" % (bins, call, sf // n), worst, 0.05)
    return ok


def case_mixer():
    fs = 2.048e6
    x = tones(fs, 4 * 2048, [(0.5, 100e3), (0.2, -300e3)]) + lcg_noise(4 * 2048, 2, 1e-3)
    ref = O.Mixer(fs); mx = St.Mixer(fs, 2048, lib=L)
    ok = True
    for f in range(4):
        if f == 0:
            ref.set_frequency(100e3); mx.setFrequency(100e3)
        if f == 2:
            ref.set_frequency(-250e3); mx.setFrequency(-250e3)
        r = ref.process(x[f * 2048:(f + 1) * 2048]); g = mx.processBlock(x[f * 2048:(f + 1) * 2048])
        ok &= report("mixer step frame %d" % f, rel_rms(g, r), 2e-6)
    return ok


def case_decimator():
    ok = True
    for fs, bw, n in [(2048000, 30000, 2048), (2048000, 200000, 2048), (20000000, 200000, 2048), (20000000, 30000, 16384)]:
        x = tones(fs, 3 * n, [(0.5, 1000.0), (0.3, fs / 5)]) + lcg_noise(3 * n, 3, 1e-2)
        ref = O.Decimator(fs, bw)
        d = St.Decimator(fs, n, lib=L)
        rate = d.buildDecimationChain(fs, bw)
        assert rate == ref.rate, (rate, ref.rate)
        for f in range(3):
            r = ref.process(x[f * n:(f + 1) * n]); g = d.process(x[f * n:(f + 1) * n])
            ok &= report("decimator step fs=%d bw=%d frame %d" % (fs, bw, f), rel_rms(g, r), 2e-6)
    return ok


def case_fastfir():
    ok = True
    for (fft, taps), (lo, hi) in [((2048, 1025), (300, 3000)), ((2048, 1025), (-5000, 5000)), ((8192, 4097), (-5000, 5000))]:
        n = 2048
        x = tones(64000, 6 * n, [(0.4873, 1000.0), (0.3, -12000.0)]) + lcg_noise(6 * n, 4, 1e-3)
        ref = O.FastFIR(fft, taps); ref.setup(lo, hi, 0, 64000)
        f = St.FastFIR(fft, taps, lib=L); f.SetupParameters(lo, hi, 0, 64000)
        for k in range(6):
            r = ref.process(x[k * n:(k + 1) * n]); g = f.ProcessData(x[k * n:(k + 1) * n])
            assert len(r) == len(g), (len(r), len(g))
            if len(r):
                ok &= report("fastfir %d/%d (%d,%d) call %d" % (fft, taps, lo, hi, k), rel_rms(g, r), 5e-6)
    return ok


def case_demod():
    ok = True
    n = 2048
    t = np.arange(4 * n) / 64000.0
    am = (0.3 * (1 + 0.5 * np.cos(2 * np.pi * 1000 * t))) * np.exp(2j * np.pi * 0.1) + lcg_noise(4 * n, 5, 1e-4)
    ref = O.DemodAM(64000, 10000); d = St.Demod(64000, 256000, n, lib=L)
    d.setDemodMode(B.DM_AM); d.setBandwidth(10000)
    for k in range(4):
        r = ref.process(am[k * n:(k + 1) * n]); g = d.processBlock(am[k * n:(k + 1) * n])
        ok &= report("AM demod call %d" % k, rel_rms(g, r), 1e-5)
    fsw = 256000
    tw = np.arange(5 * n) / fsw
    fm = 0.5 * np.exp(1j * (75000 / 1000.0) * np.sin(2 * np.pi * 1000 * tw)) + lcg_noise(5 * n, 6, 1e-4)
    tw = np.arange(15 * n) / fsw
    fm = 0.5 * np.exp(1j * (75000 / 1000.0) * np.sin(2 * np.pi * 1000 * tw)) + lcg_noise(15 * n, 6, 1e-4)
    ref = O.DemodWFM(fsw); d = St.Demod(64000, fsw, 12 * n, lib=L); d.setDemodMode(B.DM_FMM)
    for k, ln in enumerate((n, n, 3 * n, 10 * n - 100)):
        off = [0, n, 2 * n, 5 * n][k]
        r = ref.process(fm[off:off + ln]); g = d.processBlock(fm[off:off + ln])
        ok &= report("WFM demod call %d (n=%d)" % (k, ln), rel_rms(g, r), 1e-5)
    # PLL demods
    t = np.arange(6 * n) / 64000.0
    fm = 0.3 * np.exp(1j * 3.0 * np.sin(2 * np.pi * 1000 * t)) + lcg_noise(6 * n, 7, 1e-3)
    ref = O.DemodNFM(64000); d = St.Demod(64000, 256000, 2 * n, lib=L); d.setDemodMode(B.DM_FMN)
    for k, (off, ln) in enumerate(((0, n), (n, n), (2 * n, 2 * n), (4 * n, 1000))):
        r = ref.process(fm[off:off + ln]); g = d.processBlock(fm[off:off + ln])
        ok &= report("NFM demod call %d" % k, rel_rms(g, r), 1e-5)
    am = (0.3 * (1 + 0.5 * np.cos(2 * np.pi * 800 * t))) * np.exp(2j * np.pi * 30 * t) + lcg_noise(6 * n, 8, 1e-3)
    ref = O.DemodSAM(64000); d = St.Demod(64000, 256000, 2 * n, lib=L); d.setDemodMode(B.DM_SAM)
    for k, (off, ln) in enumerate(((0, n), (n, n), (2 * n, 2 * n), (4 * n, 1000))):
        r = ref.process(am[off:off + ln]); g = d.processBlock(am[off:off + ln])
        ok &= report("SAM demod call %d in-phase path" % k, rel_rms((g.real + g.imag) / 2, (r.real + r.imag) / 2), 1e-5)
        ok &= report("SAM demod call %d quadrature path (chaotic in the reference)" % k, rel_rms((g.real - g.imag) / 2, (r.real - r.imag) / 2), 5e-2)
    return ok


def case_receiver():
    ok = True
    fs, n = 2048000, 2048
    # narrow AM, single channel, host frame path
    ref = O.Receiver(fs, n, 4096); ref.set_mode(O.AM); ref.set_mixer(100e3); ref.set_filter(-5000, 5000)
    rx = B.ReceiverBank(fs, 1, True, False, 4096, lib=L)
    rx.set_mode(0, B.DM_AM); rx.set_mixer(0, 100e3); rx.set_bandpass(0, -5000, 5000)
    nfr = 2 * 32
    t = np.arange(nfr * n) / fs
    x = 10 ** (-10 / 20) * (1 + 0.5 * np.cos(2 * np.pi * 1000 * t)) * np.exp(2j * np.pi * 100e3 * t) + lcg_noise(nfr * n, 1, 3e-4)
    for f in range(nfr):
        ra, rs = ref.process(x[f * n:(f + 1) * n]); ga, gs = rx.process_iq(x[f * n:(f + 1) * n], want_spectrum=(f < 3))
        assert len(ra) == len(ga), (f, len(ra), len(ga))
        if f in (1, 2):
            m = rs > -110
            ok &= report("rx spectrum frame %d" % f, float(np.abs(gs - rs)[m].max()), 0.05)
        if len(ra):
            ok &= report("rx AM audio after frame %d" % f, rel_rms(ga, ra), 1e-5)
    # 8 USB channels, shared input, device path, two super-frames
    C = 8
    ref = [O.Receiver(fs, n, 0) for _ in range(C)]
    rx = B.ReceiverBank(fs, C, True, False, 0, max_superframes=2, lib=L)
    fcs = [-960e3 + 7.5e3 * 16 * c for c in range(C)]
    for c in range(C):
        ref[c].set_mode(O.USB); ref[c].set_mixer(fcs[c]); ref[c].set_filter(300, 3000)
        rx.set_mode(c, B.DM_USB); rx.set_mixer(c, fcs[c]); rx.set_bandpass(c, 300, 3000)
    sf = rx.superframe
    x = tones(fs, 3 * sf, [(0.05, fc + 1000.0) for fc in fcs]) + lcg_noise(3 * sf, 3, 1e-3)
    outs = []
    for lo, hi in ((0, sf), (sf, 3 * sf)):
        a, _ = rx.process(x[lo:hi])
        outs.append(a)
    g = np.concatenate(outs, axis=1)
    for c in range(C):
        r = np.concatenate([ref[c].process(x[f * n:(f + 1) * n])[0] for f in range(3 * sf // n)])
        ok &= report("bank USB channel %d" % c, rel_rms(g[c], r), 1e-5)
    # WFM bank at 2.048M
    ref = O.Receiver(fs, n, 0); ref.set_mode(O.FMM); ref.set_mixer(200e3)
    rx = B.ReceiverBank(fs, 1, True, True, 0, max_superframes=3, lib=L)
    rx.set_mixer(0, 200e3)
    sf = rx.superframe
    t = np.arange(3 * sf) / fs
    x = 0.5 * np.exp(1j * (2 * np.pi * 200e3 * t + 75.0 * np.sin(2 * np.pi * 1000 * t))) + lcg_noise(3 * sf, 2, 1e-3)
    a, _ = rx.process(x)
    r = np.concatenate([ref.process(x[f * n:(f + 1) * n])[0] for f in range(3 * sf // n)])
    ok &= report("bank WFM mono", rel_rms(a[0], r), 1e-5)
    return ok


def case_audio_tail():
    """AGC on the narrow branch and the fractional resampler after the demod"""
    ok = True
    fs, n = 2048000, 2048
    # USB bank with AGC MED on channel 0, manual gain on channel 1, resampled to 11025 Hz
    C = 2
    ref = [O.Receiver(fs, n, 0) for _ in range(C)]
    rx = B.ReceiverBank(fs, C, True, False, 0, max_superframes=2, lib=L, audio_rate=11025)
    fcs = [-400e3, 300e3]
    for c in range(C):
        ref[c].set_mode(O.USB); ref[c].set_mixer(fcs[c]); ref[c].set_filter(300, 3000); ref[c].set_audio_rate(11025)
        rx.set_mode(c, B.DM_USB); rx.set_mixer(c, fcs[c]); rx.set_bandpass(c, 300, 3000)
    ref[0].set_agc(2, 30); rx.set_agc(0, 2, 30)
    ref[1].set_agc(0, 30); rx.set_agc(1, 0, 30)
    sf = rx.superframe
    t = np.arange(5 * sf) / fs
    env = 0.02 + 0.3 * (np.sin(2 * np.pi * 2.5 * t) > 0)
    x = env * (np.exp(2j * np.pi * (fcs[0] + 1000.0) * t) + np.exp(2j * np.pi * (fcs[1] + 1700.0) * t)) + lcg_noise(5 * sf, 3, 1e-4)
    outs = []
    for lo, hi in ((0, sf), (sf, 3 * sf), (3 * sf, 5 * sf)):
        a, _ = rx.process(x[lo:hi])
        outs.append(a)
    g = np.concatenate(outs, axis=1)
    for c in range(C):
        r = np.concatenate([ref[c].process(x[f * n:(f + 1) * n])[0] for f in range(5 * sf // n)])
        ok &= report("USB + AGC(%s) + resampler channel %d: count %d vs %d" % ("MED" if c == 0 else "OFF/30", c, g.shape[1], len(r)), float(abs(g.shape[1] - len(r))), 0.0)
        m = min(g.shape[1], len(r))
        ok &= report("USB + AGC + resampler channel %d" % c, rel_rms(g[c][:m], r[:m]), 1e-5)
    # WFM resampled to 48 kHz
    ref = O.Receiver(fs, n, 0); ref.set_mode(O.FMM); ref.set_mixer(200e3); ref.set_audio_rate(48000)
    rx = B.ReceiverBank(fs, 1, True, True, 0, max_superframes=3, lib=L, audio_rate=48000)
    rx.set_mixer(0, 200e3)
    sf = rx.superframe
    t = np.arange(4 * sf) / fs
    x = 0.5 * np.exp(1j * (2 * np.pi * 200e3 * t + 75.0 * np.sin(2 * np.pi * 1000 * t))) + lcg_noise(4 * sf, 2, 1e-3)
    a1, _ = rx.process(x[:3 * sf]); a2, _ = rx.process(x[3 * sf:])
    g = np.concatenate([a1, a2], axis=1)
    r = np.concatenate([ref.process(x[f * n:(f + 1) * n])[0] for f in range(4 * sf // n)])
    ok &= report("WFM + resampler count %d vs %d" % (g.shape[1], len(r)), float(abs(g.shape[1] - len(r))), 0.0)
    m = min(g.shape[1], len(r))
    ok &= report("WFM + resampler 48 kHz", rel_rms(g[0][:m], r[:m]), 1e-5)
    return ok


def case_streambank():
    ok = True
    fs, S, N = 2.0e6, 2, 65536
    x = np.stack([tones(fs, 2 * N, [(0.4, 123456.7), (0.01, -700001.3), (0.2, 20000.0)]) + lcg_noise(2 * N, 7, 1e-4),
                  tones(fs, 2 * N, [(0.3, -40000.0), (0.05, 500000.0)]) + lcg_noise(2 * N, 8, 1e-3)])
    bands = [(-50e3, 50e3), (-100e3, -10e3)]
    sb = B.StreamBank(fs, S, frame=N, spectrum_bins=N, max_frames=1, lib=L)
    refs = []
    for c in range(S):
        sb.set_bandpass(c, *bands[c])
        f = O.FastFIR(2048, 1025); f.setup(bands[c][0], bands[c][1], 0.0, fs)
        refs.append((f, O.Spectrum(N, N, lift_clamp=True)))
    for call in range(2):
        blk = x[:, call * N:(call + 1) * N]
        y, sp = sb.process(blk)
        for c in range(S):
            ry = refs[c][0].process(blk[c]); rs = refs[c][1].process(blk[c])
            ok &= report("streambank call %d stream %d band-pass" % (call, c), rel_rms(y[c], ry), 5e-6)
            m = rs > -100
            ok &= report("streambank call %d stream %d 65536 bins max|dB|" % (call, c), float(np.abs(sp[c, 0] - rs)[m].max()), 0.05)
    return ok


CASES = {"spectrum_batch": case_spectrum_batch, "audio_tail": case_audio_tail, "streambank": case_streambank, "spectrum": case_spectrum, "mixer": case_mixer, "decimator": case_decimator, "fastfir": case_fastfir,
         "demod": case_demod, "receiver": case_receiver}

if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    allok = True
    for nm in names:
        t0 = time.time()
        ok = CASES[nm]()
        print("== %s: %s (%.1fs)" % (nm, "ok" if ok else "FAIL", time.time() - t0))
        allok &= ok
    sys.exit(0 if allok else 1)
