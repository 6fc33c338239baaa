"""Test input: an FM broadcast multiplex with an RDS subcarrier, built from the published RDS baseband coding (IEC 62106:
26-bit blocks = 16 information bits + a 10-bit checkword with an offset word per block, differential coding, biphase
symbols at 1187.5 bit/s on a suppressed 57 kHz carrier).  Nothing of this is taken from the reference: it is what a
broadcaster sends, and the reference's decoder (demod_wfm.cpp:296-357, 576-757) has to find the groups in it."""
import numpy as np

# generator polynomial g(x) = x^10 + x^8 + x^7 + x^5 + x^4 + x^3 + 1 and the offset words of blocks A, B, C, C', D
_POLY = 0x5B9
OFFSET = {"A": 0x0FC, "B": 0x198, "C": 0x168, "Cp": 0x350, "D": 0x1B4}


def checkword(info16, offset):
    """remainder of info(x) * x^10 divided by g(x), plus the block's offset word"""
    reg = info16 << 10
    for bit in range(25, 9, -1):
        if reg & (1 << bit):
            reg ^= _POLY << (bit - 10)
    return (reg & 0x3FF) ^ offset


def group_bits(a, b, c, d):
    """104 bits of one group, most significant bit first; version B groups (bit 11 of block B) carry offset C' on block C"""
    bits = []
    for word, key in ((a, "A"), (b, "B"), (c, "Cp" if (b & 0x0800) else "C"), (d, "D")):
        blk = (word << 10) | checkword(word, OFFSET[key])
        bits += [(blk >> k) & 1 for k in range(25, -1, -1)]
    return bits


def make_groups(n, seed=1, pi=0x54A8):
    """n groups: 0A (programme service name) and 2A (radio text) in turn, random payloads, the odd 2B among them"""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        if k % 5 == 4:
            b = (2 << 12) | 0x0800 | (int(rng.integers(0, 16)))      # 2B: block C repeats the PI code
            out.append((pi, b, pi, int(rng.integers(0, 65536))))
        elif k % 2:
            b = (2 << 12) | int(rng.integers(0, 16))                 # 2A
            out.append((pi, b, int(rng.integers(0, 65536)), int(rng.integers(0, 65536))))
        else:
            b = (0 << 12) | int(rng.integers(0, 4))                  # 0A
            out.append((pi, b, int(rng.integers(0, 65536)), int(rng.integers(0, 65536))))
    return out


def rds_baseband(groups, fs, n, lead_bits=40, seed=2):
    """biphase baseband of the differentially coded bit stream at rate fs, n samples, unit amplitude: a data bit is one cycle of a
    sine at the bit rate (the shaped biphase symbol is close to that), its sign the differentially coded bit"""
    rng = np.random.default_rng(seed)
    bits = list(rng.integers(0, 2, lead_bits))
    for g in groups:
        bits += group_bits(*g)
    bits = np.array(bits, dtype=np.int64)
    diff = np.bitwise_xor.accumulate(bits)                           # e[n] = d[n] xor e[n-1]
    t = np.arange(n) / fs * (57000.0 / 48.0)                         # time in bit periods
    k = np.floor(t).astype(np.int64)
    sym = np.where(k < len(diff), 2.0 * diff[np.minimum(k, len(diff) - 1)] - 1.0, 0.0)
    return sym * np.sin(2.0 * np.pi * (t - k))


def fm_multiplex(groups, fs, n, rds_level=0.08, pilot_level=0.09, audio_level=0.35, subcarrier_offset_hz=0.0, seed=3, deviation=75000.0):
    """complex FM signal at rate fs carrying left/right tones, the 19 kHz pilot, the 38 kHz difference signal and the RDS subcarrier
    (3 x pilot + subcarrier_offset_hz)"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    left = np.sin(2 * np.pi * 1000.0 * t) + 0.5 * np.sin(2 * np.pi * 3300.0 * t + 0.3)
    right = np.sin(2 * np.pi * 1700.0 * t + 1.0) + 0.3 * rng.standard_normal(n).cumsum() / np.sqrt(n)
    pilot_phase = 2 * np.pi * 19000.0 * t
    m = audio_level * 0.5 * (left + right) + pilot_level * np.sin(pilot_phase) \
        + audio_level * 0.5 * (left - right) * np.sin(2 * pilot_phase) \
        + rds_level * rds_baseband(groups, fs, n) * np.cos(3 * pilot_phase + 2 * np.pi * subcarrier_offset_hz * t)
    phase = 2 * np.pi * deviation * np.cumsum(m) / fs
    return np.exp(1j * phase)
