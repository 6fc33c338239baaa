"""ctypes face of the CPU oracle (oracle/pebble_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by pebblesdr_amd.  See oracle/pebble_oracle.h for what it restates and how it is pinned.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpebble_oracle.so")

AM, SAM, FMN, FMM, FMS, DSB, LSB, USB, CWL, CWU, DIGL, DIGU, NONE = range(13)


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms).  Safe to call repeatedly."""
    src = [os.path.join(_HERE, f) for f in ("pebble_oracle.c", "pebble_oracle.h", "hb_taps.h")]
    if (not force) and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpebble_oracle.so"])
    return _SO


_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.po_downconvert_new.restype = C.c_void_p
        L.po_downconvert_free.argtypes = [C.c_void_p]
        L.po_downconvert_set_frequency.argtypes = [C.c_void_p, C.c_double]
        L.po_downconvert_set_cw_offset.argtypes = [C.c_void_p, C.c_double]
        L.po_downconvert_set_data_rate.restype = C.c_double
        L.po_downconvert_set_data_rate.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        L.po_downconvert_chain_len.argtypes = [C.c_void_p]
        L.po_downconvert_stage_taps.argtypes = [C.c_void_p, C.c_int]
        L.po_downconvert_process.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.po_decimator_new.restype = C.c_void_p
        L.po_decimator_free.argtypes = [C.c_void_p]
        L.po_decimator_build.restype = C.c_double
        L.po_decimator_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.po_decimator_chain_len.argtypes = [C.c_void_p]
        L.po_decimator_dec_by2_stages.restype = C.c_uint32
        L.po_decimator_dec_by2_stages.argtypes = [C.c_void_p]
        L.po_decimator_stage.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.po_decimator_process.restype = C.c_uint32
        L.po_decimator_process.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.po_fft.argtypes = [_dp, C.c_uint32, C.c_int]
        L.po_fastfir_new.restype = C.c_void_p
        L.po_fastfir_new.argtypes = [C.c_uint32, C.c_uint32]
        L.po_fastfir_free.argtypes = [C.c_void_p]
        L.po_fastfir_setup.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.po_fastfir_process.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.po_fastfir_coef.restype = _dp
        L.po_fastfir_coef.argtypes = [C.c_void_p]
        L.po_spectrum_new.restype = C.c_void_p
        L.po_spectrum_new.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int]
        L.po_spectrum_free.argtypes = [C.c_void_p]
        L.po_spectrum_bins.restype = C.c_uint32
        L.po_spectrum_bins.argtypes = [C.c_void_p]
        L.po_spectrum_coherent_gain.restype = C.c_double
        L.po_spectrum_coherent_gain.argtypes = [C.c_void_p]
        L.po_spectrum_window.restype = _dp
        L.po_spectrum_window.argtypes = [C.c_void_p]
        L.po_spectrum_process.argtypes = [C.c_void_p, _dp, C.c_uint32, _dp]
        L.po_receiver_new.restype = C.c_void_p
        L.po_receiver_new.argtypes = [C.c_uint32] * 5
        L.po_receiver_free.argtypes = [C.c_void_p]
        L.po_receiver_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.po_receiver_set_mixer.argtypes = [C.c_void_p, C.c_double]
        L.po_receiver_set_filter.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.po_receiver_demod_rate.restype = C.c_double
        L.po_receiver_demod_rate.argtypes = [C.c_void_p, C.c_int]
        L.po_receiver_dec_stages.restype = C.c_uint32
        L.po_receiver_dec_stages.argtypes = [C.c_void_p, C.c_int]
        L.po_receiver_process.restype = C.c_uint32
        L.po_receiver_process.argtypes = [C.c_void_p, _dp, C.c_uint32, _dp, _dp]
        L.po_receiver_set_agc.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.po_receiver_set_conditioners.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.po_receiver_set_anf.argtypes = [C.c_void_p, C.c_int]
        L.po_receiver_set_squelch.argtypes = [C.c_void_p, C.c_double]
        L.po_iq_balance.argtypes = [C.c_double, C.c_double, _dp, _dp, C.c_int]
        L.po_anf_init.argtypes = [C.c_void_p]
        L.po_anf_process.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.po_nb_init.argtypes = [C.c_void_p]
        L.po_nb_enable.argtypes = [C.c_void_p, C.c_int]
        L.po_nb1_process.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.po_nb2_process.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.po_receiver_set_audio_rate.argtypes = [C.c_void_p, C.c_uint32]
        L.po_demod_wfm_rds_rate.restype = C.c_double
        L.po_demod_wfm_rds_rate.argtypes = [C.c_void_p]
        L.po_demod_wfm_rds_last.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.po_demod_wfm_rds_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.po_demod_wfm_rds_pushed.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.po_demod_wfm_next_rds_group.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.po_demod_wfm_free.argtypes = [C.c_void_p]
        L.po_receiver_rds_polled.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.po_fd_estimate.restype = C.c_double
        L.po_fd_estimate.argtypes = [_dp, C.c_int, C.c_uint32, C.c_float, C.c_float, C.c_double, _dp]
        L.po_agc_new.restype = C.c_void_p
        L.po_agc_new.argtypes = [C.c_double]
        L.po_agc_free.argtypes = [C.c_void_p]
        L.po_agc_set_mode.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.po_agc_process.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.po_resampler_new.restype = C.c_void_p
        L.po_resampler_new.argtypes = [C.c_int]
        L.po_resampler_free.argtypes = [C.c_void_p]
        L.po_resampler_process.restype = C.c_int
        L.po_resampler_process.argtypes = [C.c_void_p, C.c_int, C.c_double, _dp, _dp]
        L.po_resampler_time.restype = C.c_double
        L.po_resampler_time.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _c128(x):
    return np.ascontiguousarray(x, dtype=np.complex128)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def normalize_iq(raw, fmt, order=0, gain=1.0):
    """DeviceInterfaceBase::normalizeIQ batch variants (pebblelib/deviceinterfacebase.cpp:648-838) and the WAV PCM16 scaling
    (wavfile.cpp:299-300), restated in numpy (fp64).  fmt: 0 int8 /128, 1 uint8 (v-128)/128, 2 int16 /32768, 3 float32,
    4 WAV int16 /32767.  order: 0 IQ, 1 QI, 2 I only, 3 Q only."""
    v = np.asarray(raw).astype(np.float64).reshape(-1, 2)
    if fmt == 1:
        v = v - 128.0
    scale = {0: 1 / 128.0, 1: 1 / 128.0, 2: 1 / 32768.0, 3: 1.0, 4: 1 / 32767.0}[fmt] * gain
    i, q = v[:, 0] * scale, v[:, 1] * scale
    if order == 0:
        return i + 1j * q
    if order == 1:
        return q + 1j * i
    if order == 2:
        return i + 1j * i
    return q + 1j * q


class _MixerS(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fs", "freq", "inc", "osc_cos", "osc_sin", "last_re", "last_im")]


class Mixer:
    """pebblelib/mixer.cpp"""

    def __init__(self, fs):
        self.s = _MixerS()
        lib().po_mixer_init(C.byref(self.s), C.c_double(fs))

    def set_frequency(self, f):
        lib().po_mixer_set_frequency(C.byref(self.s), C.c_double(f))

    def process(self, x):
        x = _c128(x)
        out = np.empty_like(x)
        r = lib().po_mixer_process(C.byref(self.s), _ptr(x), _ptr(out), C.c_uint32(len(x)))
        return out if r else x.copy()


class Decimator:
    """pebblelib/decimator.cpp (vDSP path)"""

    def __init__(self, fs_in, protect_bw, fs_out_min=0):
        self.h = lib().po_decimator_new()
        self.rate = lib().po_decimator_build(self.h, int(fs_in), int(protect_bw), int(fs_out_min))

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_decimator_free(self.h)
            self.h = None

    @property
    def dec_by2_stages(self):
        return lib().po_decimator_dec_by2_stages(self.h)

    def chain(self):
        """[(ntaps, stride)] -- ntaps 0 means CIC3"""
        out = []
        for i in range(lib().po_decimator_chain_len(self.h)):
            nt, st, de = C.c_int(), C.c_uint32(), C.c_int()
            lib().po_decimator_stage(self.h, i, C.byref(nt), C.byref(st), C.byref(de))
            out.append((nt.value, st.value))
        return out

    @property
    def total_decimation(self):
        d = 1
        for _, s in self.chain():
            d *= s
        return d

    def process(self, x):
        x = _c128(x)
        out = np.empty(len(x) + 8, dtype=np.complex128)
        n = lib().po_decimator_process(self.h, _ptr(x), _ptr(out), C.c_uint32(len(x)))
        return out[:n].copy()


class DownConvert:
    """pebblelib/downconvert.cpp (CDownConvert: quadrature-oscillator mixer + cascade of decimate-by-2 stages)"""

    def __init__(self):
        self.h = lib().po_downconvert_new()
        self.rate = None

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_downconvert_free(self.h)
            self.h = None

    def set_data_rate(self, in_rate, max_bw, simple=False):
        self.rate = lib().po_downconvert_set_data_rate(self.h, float(in_rate), float(max_bw), 1 if simple else 0)
        return self.rate

    def set_frequency(self, f):
        lib().po_downconvert_set_frequency(self.h, float(f))

    def set_cw_offset(self, off):
        lib().po_downconvert_set_cw_offset(self.h, float(off))

    def chain(self):
        """tap counts of the stages, 0 = CIC3"""
        return [lib().po_downconvert_stage_taps(self.h, i) for i in range(lib().po_downconvert_chain_len(self.h))]

    def process(self, x):
        x = _c128(x)
        out = np.empty(len(x) + 8, dtype=np.complex128)
        n = lib().po_downconvert_process(self.h, C.c_int(len(x)), _ptr(x), _ptr(out))
        return out[:n].copy()


def fft(x, inverse=False):
    x = _c128(x).copy()
    lib().po_fft(_ptr(x), C.c_uint32(len(x)), C.c_int(-1 if inverse else 1))
    return x


class FastFIR:
    """pebblelib/fastfir.cpp"""

    def __init__(self, fft_size=2048, fir_size=1025):
        self.fft_size, self.fir_size = fft_size, fir_size
        self.h = lib().po_fastfir_new(fft_size, fir_size)

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_fastfir_free(self.h)
            self.h = None

    def setup(self, lo, hi, offset, fs):
        return lib().po_fastfir_setup(self.h, lo, hi, offset, fs)

    def coef(self):
        p = lib().po_fastfir_coef(self.h)
        return np.ctypeslib.as_array(p, shape=(2 * self.fft_size,)).view(np.complex128).copy()

    def process(self, x):
        x = _c128(x)
        out = np.empty(len(x) + self.fft_size, dtype=np.complex128)
        n = lib().po_fastfir_process(self.h, len(x), _ptr(x), _ptr(out))
        return out[:n].copy()


class Spectrum:
    """FFT::fftSpectrum (pebblelib/fft.cpp) as SignalSpectrum configures it"""

    def __init__(self, fft_size, samples_per_buffer, window=True, lift_clamp=False):
        self.h = lib().po_spectrum_new(fft_size, samples_per_buffer, 0 if window else 1, 1 if lift_clamp else 0)
        self.bins = lib().po_spectrum_bins(self.h)
        self.spb = samples_per_buffer

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_spectrum_free(self.h)
            self.h = None

    @property
    def coherent_gain(self):
        return lib().po_spectrum_coherent_gain(self.h)

    def window(self):
        return np.ctypeslib.as_array(lib().po_spectrum_window(self.h), shape=(self.spb,)).copy()

    def process(self, x):
        x = _c128(x)
        out = np.empty(self.bins, dtype=np.float64)
        lib().po_spectrum_process(self.h, _ptr(x), len(x), _ptr(out))
        return out


class _FirS(C.Structure):
    _fields_ = [("ntaps", C.c_int), ("state", C.c_int), ("fs", C.c_double), ("coef", C.c_double * 150),
                ("icoef", C.c_double * 150), ("qcoef", C.c_double * 150),
                ("zre", C.c_double * 75), ("zim", C.c_double * 75)]


class Fir:
    """pebblelib/fir.cpp CFir (low-pass design + complex process)"""

    def __init__(self):
        self.s = _FirS()

    def init_lp(self, ntaps, scale, astop, fpass, fstop, fs):
        f = lib().po_fir_init_lp
        f.argtypes = [C.c_void_p, C.c_int] + [C.c_double] * 5
        return f(C.byref(self.s), ntaps, scale, astop, fpass, fstop, fs)

    def taps(self):
        return np.array(self.s.coef[: self.s.ntaps])

    def generate_hb(self, freq_offset):
        lib().po_fir_generate_hb(C.byref(self.s), C.c_double(freq_offset))

    def iq_taps(self):
        return np.array(self.s.icoef[: self.s.ntaps]), np.array(self.s.qcoef[: self.s.ntaps])

    def process(self, x):
        x = _c128(x)
        out = np.empty_like(x)
        lib().po_fir_process_cpx(C.byref(self.s), C.c_int(len(x)), _ptr(x), _ptr(out))
        return out


class _IirS(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("a1", "a2", "b0", "b1", "b2", "w1a", "w2a", "w1b", "w2b")]


class Iir:
    """pebblelib/iir.cpp CIir"""

    def __init__(self, kind, f0, q, fs):
        self.s = _IirS()
        f = getattr(lib(), "po_iir_init_" + kind)
        f.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        f(C.byref(self.s), f0, q, fs)

    def coeffs(self):
        return (self.s.b0, self.s.b1, self.s.b2, self.s.a1, self.s.a2)

    def process(self, x):
        x = _c128(x)
        out = np.empty_like(x)
        lib().po_iir_process_cpx(C.byref(self.s), C.c_int(len(x)), _ptr(x), _ptr(out))
        return out


class _AmS(C.Structure):
    _fields_ = [("fs", C.c_double), ("dc", C.c_double), ("dc_last", C.c_double), ("lp", _FirS)]


class DemodAM:
    """application/demod/demod_am.cpp processBlockFiltered"""

    def __init__(self, fs, bandwidth=None):
        self.s = _AmS()
        lib().po_demod_am_init(C.byref(self.s), C.c_double(fs))
        if bandwidth is not None:
            self.set_bandwidth(bandwidth)

    def set_bandwidth(self, bw):
        lib().po_demod_am_set_bandwidth(C.byref(self.s), C.c_double(bw))

    @property
    def ntaps(self):
        return self.s.lp.ntaps

    def process(self, x):
        x = _c128(x)
        out = np.empty_like(x)
        lib().po_demod_am_process(C.byref(self.s), _ptr(x), _ptr(out), C.c_int(len(x)))
        return out


class _NfmS(C.Structure):
    _fields_ = [("fs", C.c_double)] + [(n, C.c_float) for n in ("err_dc", "nco_freq", "nco_lo", "nco_hi", "phase", "alpha", "beta",
                                                                "dc_alpha", "out_gain")] + [("lp", _FirS)]


class DemodNFM:
    """application/demod/demod_nfm.cpp processBlockNCO"""

    def __init__(self, fs):
        self.s = _NfmS()
        lib().po_demod_nfm_init(C.byref(self.s), C.c_double(fs))

    @property
    def ntaps(self):
        return self.s.lp.ntaps

    def process(self, x):
        x = _c128(x)
        out = np.empty_like(x)
        lib().po_demod_nfm_process(C.byref(self.s), _ptr(x), _ptr(out), C.c_int(len(x)))
        return out


class _SamS(C.Structure):
    _fields_ = [("fs", C.c_double)] + [(n, C.c_float) for n in ("lo", "hi", "freq", "phase", "alpha", "beta")] + \
               [(n, C.c_double) for n in ("dc_re", "dc_re_last", "dc_im", "dc_im_last")] + [("bp", _FirS)]


class DemodSAM:
    """application/demod/demod_sam.cpp processBlock"""

    def __init__(self, fs):
        self.s = _SamS()
        lib().po_demod_sam_init(C.byref(self.s), C.c_double(fs))

    @property
    def ntaps(self):
        return self.s.bp.ntaps

    def process(self, x):
        x = _c128(x)
        out = np.empty_like(x)
        lib().po_demod_sam_process(C.byref(self.s), _ptr(x), _ptr(out), C.c_int(len(x)))
        return out


class _WfmS(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fs", "d1_re", "d1_im", "deemph_alpha", "deemph_re", "deemph_im")] + \
               [("mono_lp", _IirS), ("notch", _IirS), ("lp", _FirS), ("hilbert", _FirS), ("pilot_bp", _IirS)] + \
               [(n, C.c_double) for n in ("nco_phase", "nco_freq", "nco_lo", "nco_hi", "pll_alpha", "pll_beta", "err_ave", "err_alpha",
                                          "phase_adjust")] + [("pilot_locked", C.c_int), ("rds", C.c_void_p)]


class _RdsGroup(C.Structure):
    _fields_ = [(n, C.c_uint16) for n in ("a", "b", "c", "d")]


class DemodWFM:
    """application/demod/demod_wfm.cpp processDataMono"""

    def __init__(self, fs):
        self.s = _WfmS()
        lib().po_demod_wfm_init(C.byref(self.s), C.c_double(fs))

    @property
    def ntaps(self):
        return self.s.lp.ntaps

    def process(self, x):
        x = _c128(x)
        out = np.empty_like(x)
        lib().po_demod_wfm_process_mono(C.byref(self.s), _ptr(x), _ptr(out), C.c_int(len(x)))
        return out

    def process_stereo(self, x):
        """processDataStereo, audio part: -> (complex (left, right), pilot-lock flag of the block)"""
        x = _c128(x)
        out = np.empty_like(x)
        locked = lib().po_demod_wfm_process_stereo(C.byref(self.s), _ptr(x), _ptr(out), C.c_int(len(x)))
        return out, bool(locked)

    def __del__(self):
        try:
            lib().po_demod_wfm_free(C.byref(self.s))
        except Exception:
            pass

    # the RDS branch of processDataStereo (demod_wfm.cpp:296-357, 488-786)
    @property
    def rds_rate(self):
        return lib().po_demod_wfm_rds_rate(C.byref(self.s))

    def rds_last(self):
        """(m_RdsData, bit-sync resonator output) of the last process_stereo call"""
        n = lib().po_demod_wfm_rds_last(C.byref(self.s), None, None, 0)
        data = np.empty(n, dtype=np.float64)
        sync = np.empty(n, dtype=np.float64)
        lib().po_demod_wfm_rds_last(C.byref(self.s), _ptr(data), _ptr(sync), n)
        return data, sync

    def rds_bits(self):
        """the bits handed to processNewRdsBit since the last call of this method"""
        buf = np.empty(1 << 20, dtype=np.uint8)
        n = lib().po_demod_wfm_rds_bits(C.byref(self.s), buf.ctypes.data_as(C.c_void_p), len(buf))
        return buf[:n].copy()

    def rds_pushed(self):
        """the groups put into m_RdsGroupQueue since the last call of this method, (n, 4) uint16; a cleared queue shows as a zero group"""
        arr = (_RdsGroup * 4096)()
        n = lib().po_demod_wfm_rds_pushed(C.byref(self.s), arr, 4096)
        return np.array([[g.a, g.b, g.c, g.d] for g in arr[:n]], dtype=np.uint16).reshape(-1, 4)

    def next_rds_group(self):
        """getNextRdsGroupData: None when the queue is empty, else ((a, b, c, d), changed)"""
        g = _RdsGroup()
        ch = C.c_int(0)
        if not lib().po_demod_wfm_next_rds_group(C.byref(self.s), C.byref(g), C.byref(ch)):
            return None
        return (g.a, g.b, g.c, g.d), bool(ch.value)


def fd_estimate(spectrum_db, spectrum_rate, bp_lo, bp_hi, mixer_freq):
    """SignalStrength::fdEstimate -> (peakDb, avgDb, snrDb, floorDb)"""
    sp = np.ascontiguousarray(spectrum_db, dtype=np.float64)
    out = np.empty(4, dtype=np.float64)
    lib().po_fd_estimate(_ptr(sp), len(sp), int(spectrum_rate), float(bp_lo), float(bp_hi), float(mixer_freq), _ptr(out))
    return out


class _AnfS(C.Structure):
    _fields_ = [("coeff", C.c_double * 90), ("delay", C.c_double * 1024), ("head", C.c_int), ("last", C.c_int)]


class Anf:
    """NoiseFilter::ProcessBlock (ANF), application/noisefilter.cpp"""

    def __init__(self):
        self.s = _AnfS()
        lib().po_anf_init(C.byref(self.s))

    def process(self, x):
        x = _c128(x)
        out = np.empty_like(x)
        lib().po_anf_process(C.byref(self.s), _ptr(x), _ptr(out), len(x))
        return out


class _NbS(C.Structure):
    _fields_ = [("nb_avg_mag", C.c_float), ("nb2_avg_mag", C.c_float), ("spike_count", C.c_int), ("nb2_avg", C.c_double * 2),
                ("delay", C.c_double * 16), ("head", C.c_int), ("last", C.c_int)]


class NoiseBlanker:
    """NoiseBlanker::ProcessBlock / ProcessBlock2, application/noiseblanker.cpp"""

    def __init__(self):
        self.s = _NbS()
        lib().po_nb_init(C.byref(self.s))

    def enable(self, which):
        lib().po_nb_enable(C.byref(self.s), int(which))

    def process(self, x, which=1):
        x = _c128(x)
        out = np.empty_like(x)
        (lib().po_nb1_process if which == 1 else lib().po_nb2_process)(C.byref(self.s), _ptr(x), _ptr(out), len(x))
        return out


def iq_balance(x, gain, phase):
    """IQBalance::ProcessBlock on one block"""
    x = _c128(x)
    out = np.empty_like(x)
    lib().po_iq_balance(float(gain), float(phase), _ptr(x), _ptr(out), len(x))
    return out


class Agc:
    """application/agc.cpp"""

    def __init__(self, sample_rate):
        self.h = lib().po_agc_new(float(sample_rate))

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_agc_free(self.h)
            self.h = None

    def set_mode(self, mode, threshold):
        lib().po_agc_set_mode(self.h, int(mode), int(threshold))

    def process(self, x):
        x = _c128(x)
        out = np.empty(len(x), dtype=np.complex128)
        lib().po_agc_process(self.h, _ptr(x), _ptr(out), len(x))
        return out


class Resampler:
    """CFractResampler (complex), pebblelib/fractresampler.cpp"""

    def __init__(self, max_input):
        self.h = lib().po_resampler_new(int(max_input))
        self.cap = int(max_input)

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_resampler_free(self.h)
            self.h = None

    @property
    def float_time(self):
        return lib().po_resampler_time(self.h)

    def process(self, x, rate):
        x = _c128(x)
        out = np.empty(int(len(x) / rate) + 4, dtype=np.complex128)
        n = lib().po_resampler_process(self.h, len(x), float(rate), _ptr(x), _ptr(out))
        return out[:n].copy()


class Receiver:
    """Receiver::processIQData DSP skeleton (application/receiver.cpp:758-1009)"""

    def __init__(self, fs, frames_per_buffer=2048, spectrum_bins=4096, fastfir_fft=2048, fastfir_taps=1025):
        self.n = frames_per_buffer
        self.bins = max(2048, spectrum_bins) if spectrum_bins else 0
        self.cap = max(frames_per_buffer, fastfir_fft)
        self.h = lib().po_receiver_new(int(fs), frames_per_buffer, spectrum_bins, fastfir_fft, fastfir_taps)

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_receiver_free(self.h)
            self.h = None

    def set_mode(self, mode):
        lib().po_receiver_set_mode(self.h, mode)

    def set_mixer(self, f):
        lib().po_receiver_set_mixer(self.h, f)

    def set_filter(self, lo, hi):
        return lib().po_receiver_set_filter(self.h, lo, hi)

    def demod_rate(self, wfm=False):
        return lib().po_receiver_demod_rate(self.h, 1 if wfm else 0)

    def dec_stages(self, wfm=False):
        return lib().po_receiver_dec_stages(self.h, 1 if wfm else 0)

    def rds_polled(self):
        """dmFMS: the groups Demod::fmStereo popped since the last call of this method -> ((n, 4) uint16, (n,) bool changed)"""
        arr = (_RdsGroup * 4096)()
        ch = np.zeros(4096, dtype=np.uint8)
        n = lib().po_receiver_rds_polled(self.h, arr, ch.ctypes.data_as(C.c_void_p), 4096)
        return np.array([[g.a, g.b, g.c, g.d] for g in arr[:n]], dtype=np.uint16).reshape(-1, 4), ch[:n].astype(bool)

    def set_agc(self, mode, threshold):
        lib().po_receiver_set_agc(self.h, int(mode), int(threshold))

    def set_conditioners(self, flags, iq_gain=1.0, iq_phase=0.0):
        lib().po_receiver_set_conditioners(self.h, int(flags), float(iq_gain), float(iq_phase))

    def set_anf(self, on=True):
        lib().po_receiver_set_anf(self.h, 1 if on else 0)

    def set_squelch(self, squelch_db):
        """Receiver::squelchChanged (receiver.cpp:704-707); -120 = never gate"""
        lib().po_receiver_set_squelch(self.h, float(squelch_db))

    def set_audio_rate(self, rate):
        lib().po_receiver_set_audio_rate(self.h, int(rate))

    def process(self, frame, want_spectrum=True):
        """one frame -> (audio ndarray (possibly empty), spectrum ndarray or None)"""
        x = _c128(frame)
        audio = np.empty(self.cap, dtype=np.complex128)
        spec = np.empty(self.bins, dtype=np.float64) if (want_spectrum and self.bins) else None
        n = lib().po_receiver_process(self.h, _ptr(x), len(x), _ptr(audio), _ptr(spec) if spec is not None else None)
        return audio[:n].copy(), spec
