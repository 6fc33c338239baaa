"""Synthetic IQ inputs shared by the tests, the bench and the dev tools (SURVEY.md 8d recipe).

x[n] = sum_k A_k exp(j 2 pi f_k n / Fs) + sigma * (LCG-uniform - 0.5), LCG s = s*1664525 + 1013904223 (32-bit).
Inputs are regenerated from (formula, seed), never stored.
"""
import numpy as np


def lcg_uniform(n, seed):
    """n uniform numbers in [0, 1) from the 32-bit LCG (vectorised by jumping: s_k = a^k s_0 + c (a^k-1)/(a-1))."""
    a, c, m = 1664525, 1013904223, 1 << 32
    out = np.empty(n, dtype=np.uint64)
    s = np.uint64(seed & 0xFFFFFFFF)
    # generate in blocks with plain python ints for exactness of the recurrence, vectorised per block via cumulative affine maps
    block = 4096
    # affine maps A_i(s) = mul[i]*s + add[i] for i = 1..block
    mul = np.empty(block, dtype=np.uint64)
    add = np.empty(block, dtype=np.uint64)
    mm, aa = 1, 0
    for i in range(block):
        mm = (mm * a) % m
        aa = (aa * a + c) % m
        mul[i] = mm
        add[i] = aa
    pos = 0
    sv = int(s)
    while pos < n:
        k = min(block, n - pos)
        vals = (mul[:k] * np.uint64(sv) + add[:k]) & np.uint64(m - 1)
        out[pos:pos + k] = vals
        sv = int(vals[k - 1])
        pos += k
    return out.astype(np.float64) / float(m)


def lcg_noise(n, seed, sigma):
    """complex noise: sigma*(u-0.5) on I and on Q, consecutive LCG draws (I first)"""
    u = lcg_uniform(2 * n, seed)
    return sigma * ((u[0::2] - 0.5) + 1j * (u[1::2] - 0.5))


def tones(fs, n, specs, n0=0):
    """sum of complex tones; specs = [(amplitude, freq_hz[, phase_rad])]"""
    t = (np.arange(n, dtype=np.float64) + n0) / float(fs)
    x = np.zeros(n, dtype=np.complex128)
    for sp in specs:
        a, f = sp[0], sp[1]
        ph = sp[2] if len(sp) > 2 else 0.0
        x += a * np.exp(1j * (2 * np.pi * f * t + ph))
    return x
