"""Python faces of the stand-alone process steps (pebblegpu_mixer_* ... pebblegpu_spectrum_*).

Same names, argument meaning and return conventions as the reference classes they mirror:
Mixer (pebblelib/mixer.h), Decimator (pebblelib/decimator.h:229-252), CFastFIR (pebblelib/fastfir.h),
Demod (application/demod.h) and FFT::fftSpectrum (pebblelib/fft.h:38).  Host complex128 in and out.
"""
import ctypes as C

import numpy as np

from .binding import check, load_library

_dp = C.POINTER(C.c_double)


def _c128(x):
    return np.ascontiguousarray(x, dtype=np.complex128)


class _Step:
    _destroy = None

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            getattr(self.L, self._destroy)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mixer(_Step):
    _destroy = "pebblegpu_mixer_destroy"

    def __init__(self, sample_rate, buffer_size, device=0, lib=None):
        self.L = lib or load_library()
        self.n = buffer_size
        self.h = C.c_void_p()
        check(self.L, self.L.pebblegpu_mixer_create(device, int(sample_rate), int(buffer_size), C.byref(self.h)))

    def setFrequency(self, f):
        check(self.L, self.L.pebblegpu_mixer_set_frequency(self.h, float(f)))

    def processBlock(self, x):
        x = _c128(x)
        assert len(x) == self.n
        out = _dp()
        check(self.L, self.L.pebblegpu_mixer_process(self.h, x.ctypes.data_as(_dp), C.byref(out)))
        if C.addressof(out.contents) == x.ctypes.data:
            return x  # f == 0: the reference returns its input pointer (mixer.cpp:51-53)
        return np.ctypeslib.as_array(out, shape=(2 * self.n,)).view(np.complex128).copy()


class Decimator(_Step):
    _destroy = "pebblegpu_decimator_destroy"

    def __init__(self, sample_rate, buffer_size, device=0, lib=None):
        self.L = lib or load_library()
        self.h = C.c_void_p()
        check(self.L, self.L.pebblegpu_decimator_create(device, int(sample_rate), int(buffer_size), C.byref(self.h)))

    def buildDecimationChain(self, sample_rate_in, protect_bw, sample_rate_out=0):
        r = C.c_float()
        check(self.L, self.L.pebblegpu_decimator_build_chain(self.h, int(sample_rate_in), int(protect_bw), int(sample_rate_out), C.byref(r)))
        return r.value

    def decBy2Stages(self):
        s = C.c_uint32()
        check(self.L, self.L.pebblegpu_decimator_dec_by2_stages(self.h, C.byref(s)))
        return s.value

    def process(self, x):
        x = _c128(x)
        out = np.empty(len(x), dtype=np.complex128)
        n = C.c_uint32()
        check(self.L, self.L.pebblegpu_decimator_process(self.h, x.ctypes.data_as(_dp), out.ctypes.data_as(_dp), len(x), C.byref(n)))
        return out[: n.value].copy()


class DownConvert(_Step):
    """CDownConvert (pebblelib/downconvert.h:25-50)"""
    _destroy = "pebblegpu_downconvert_destroy"

    def __init__(self, max_in_length=65536, device=0, lib=None):
        self.L = lib or load_library()
        self.h = C.c_void_p()
        check(self.L, self.L.pebblegpu_downconvert_create(device, int(max_in_length), C.byref(self.h)))

    def SetDataRate(self, in_rate, max_bw, simple=False):
        r = C.c_double()
        check(self.L, self.L.pebblegpu_downconvert_set_data_rate(self.h, float(in_rate), float(max_bw), 1 if simple else 0, C.byref(r)))
        return r.value

    def SetDataRateSimple(self, in_rate, max_bw):
        return self.SetDataRate(in_rate, max_bw, True)

    def SetFrequency(self, f):
        check(self.L, self.L.pebblegpu_downconvert_set_frequency(self.h, float(f)))

    def SetCwOffset(self, off):
        check(self.L, self.L.pebblegpu_downconvert_set_cw_offset(self.h, float(off)))

    def stages(self):
        n = C.c_uint32()
        taps = (C.c_uint32 * 16)()
        check(self.L, self.L.pebblegpu_downconvert_stages(self.h, C.byref(n), taps, 16))
        return [int(taps[i]) for i in range(n.value)]

    def ProcessData(self, x):
        x = _c128(x)
        out = np.empty(len(x), dtype=np.complex128)
        n = C.c_uint32()
        check(self.L, self.L.pebblegpu_downconvert_process(self.h, len(x), x.ctypes.data_as(_dp), out.ctypes.data_as(_dp), C.byref(n)))
        return out[: n.value].copy()


class FastFIR(_Step):
    _destroy = "pebblegpu_fastfir_destroy"

    def __init__(self, fft_size=0, fir_size=0, device=0, lib=None):
        self.L = lib or load_library()
        self.fft_size = fft_size or 2048
        self.h = C.c_void_p()
        check(self.L, self.L.pebblegpu_fastfir_create(device, fft_size, fir_size, C.byref(self.h)))

    def SetupParameters(self, lo, hi, offset, sample_rate):
        check(self.L, self.L.pebblegpu_fastfir_setup(self.h, float(lo), float(hi), float(offset), float(sample_rate)))

    def ProcessData(self, x):
        x = _c128(x)
        out = np.empty(len(x) + self.fft_size, dtype=np.complex128)
        n = C.c_int()
        check(self.L, self.L.pebblegpu_fastfir_process(self.h, len(x), x.ctypes.data_as(_dp), out.ctypes.data_as(_dp), C.byref(n)))
        return out[: n.value].copy()


class Demod(_Step):
    _destroy = "pebblegpu_demod_destroy"

    def __init__(self, sample_rate, wfm_sample_rate, buffer_size, device=0, lib=None):
        self.L = lib or load_library()
        self.h = C.c_void_p()
        check(self.L, self.L.pebblegpu_demod_create(device, int(sample_rate), int(wfm_sample_rate), int(buffer_size), C.byref(self.h)))

    def setDemodMode(self, mode):
        check(self.L, self.L.pebblegpu_demod_set_mode(self.h, int(mode)))

    def setBandwidth(self, bw):
        check(self.L, self.L.pebblegpu_demod_set_bandwidth(self.h, float(bw)))

    def processBlock(self, x):
        x = _c128(x)
        out = _dp()
        check(self.L, self.L.pebblegpu_demod_process(self.h, x.ctypes.data_as(_dp), len(x), C.byref(out)))
        if C.addressof(out.contents) == x.ctypes.data:
            return x  # pass-through modes return `in` (demod.cpp:127-138)
        return np.ctypeslib.as_array(out, shape=(2 * len(x),)).view(np.complex128).copy()

    def getNextRdsGroupData(self, cap=4096):
        """dmFMS: the groups Demod::fmStereo's one getNextRdsGroupData per processBlock call took from the queue since the last call of
        this method -> ((n, 4) uint16 blocks A..D, (n,) bool: the function's return value)"""
        g = np.zeros((cap, 4), dtype=np.uint16)
        chg = np.zeros(cap, dtype=np.uint8)
        n = C.c_uint32(0)
        check(self.L, self.L.pebblegpu_demod_rds_groups(self.h, g.ctypes.data_as(C.c_void_p), chg.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return g[:n.value].copy(), chg[:n.value].astype(bool)

    def getStereoLock(self):
        """Demod_WFM::getStereoLock -> (m_PilotLocked after the last processBlock call, changed since the last call of this method)"""
        lk, chg = C.c_int32(0), C.c_int32(0)
        check(self.L, self.L.pebblegpu_demod_stereo_lock(self.h, C.byref(lk), C.byref(chg)))
        return bool(lk.value), bool(chg.value)

    def rdsData(self, cap=1 << 16):
        """m_RdsData of the last processBlock call (the matched filter's output in front of the bit slicer)"""
        d = np.zeros(cap, dtype=np.float64)
        n = C.c_uint32(0)
        check(self.L, self.L.pebblegpu_demod_rds_signal(self.h, d.ctypes.data_as(_dp), cap, C.byref(n)))
        return d[:min(n.value, cap)].copy()


class Spectrum(_Step):
    _destroy = "pebblegpu_spectrum_destroy"

    def __init__(self, fft_size, sample_rate, samples_per_buffer, device=0, lib=None):
        self.L = lib or load_library()
        self.h = C.c_void_p()
        check(self.L, self.L.pebblegpu_spectrum_create(device, int(fft_size), float(sample_rate), int(samples_per_buffer), C.byref(self.h)))
        b = C.c_uint32()
        check(self.L, self.L.pebblegpu_spectrum_bins(self.h, C.byref(b)))
        self.bins = b.value

    def fftSpectrum(self, x):
        """-> (dB array [bins], overload flag)"""
        x = _c128(x)
        out = np.empty(self.bins, dtype=np.float64)
        ov = C.c_int()
        check(self.L, self.L.pebblegpu_spectrum_process(self.h, x.ctypes.data_as(_dp), len(x), out.ctypes.data_as(_dp), C.byref(ov)))
        return out, bool(ov.value)
