"""ctypes bindings for libpebblegpu (include/pebblegpu.h) and the receiver-bank convenience class."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = "libpebblegpu.so"

(DM_AM, DM_SAM, DM_FMN, DM_FMM, DM_FMS, DM_DSB, DM_LSB, DM_USB, DM_CWL, DM_CWU, DM_DIGL, DM_DIGU, DM_NONE) = range(13)

# every symbol include/pebblegpu.h declares (tests/test_abi_symbols.py checks the header against this list too)
SYMBOLS = [
    "pebblegpu_last_error", "pebblegpu_abi_version", "pebblegpu_device_count",
    "pebblegpu_malloc", "pebblegpu_free", "pebblegpu_memcpy_h2d", "pebblegpu_memcpy_d2h", "pebblegpu_memset",
    "pebblegpu_device_synchronize", "pebblegpu_probe_copy_gbps", "pebblegpu_normalize_iq",
    "pebblegpu_receiver_create", "pebblegpu_receiver_destroy", "pebblegpu_receiver_info",
    "pebblegpu_set_mixer_freq", "pebblegpu_set_bandpass", "pebblegpu_set_demod_mode", "pebblegpu_receiver_rds_groups", "pebblegpu_receiver_stereo_lock", "pebblegpu_set_agc", "pebblegpu_set_conditioners", "pebblegpu_set_noise_filter", "pebblegpu_set_squelch", "pebblegpu_receiver_process_raw",
    "pebblegpu_receiver_process", "pebblegpu_receiver_audio", "pebblegpu_receiver_spectrum", "pebblegpu_receiver_zoom_spectrum",
    "pebblegpu_receiver_ingest_acquire", "pebblegpu_receiver_ingest_submit", "pebblegpu_receiver_process_ingested", "pebblegpu_receiver_last_ms", "pebblegpu_receiver_kernel_name", "pebblegpu_receiver_mean_ms", "pebblegpu_receiver_set_profiling", "pebblegpu_receiver_enable_signal_strength", "pebblegpu_receiver_signal_strength", "pebblegpu_receiver_synchronize", "pebblegpu_process_iq",
    "pebblegpu_streambank_create", "pebblegpu_streambank_destroy", "pebblegpu_streambank_set_bandpass",
    "pebblegpu_streambank_process", "pebblegpu_streambank_filtered", "pebblegpu_streambank_spectrum",
    "pebblegpu_streambank_last_ms", "pebblegpu_streambank_synchronize",
    "pebblegpu_mixer_create", "pebblegpu_mixer_destroy", "pebblegpu_mixer_set_frequency", "pebblegpu_mixer_process",
    "pebblegpu_decimator_create", "pebblegpu_decimator_destroy", "pebblegpu_decimator_build_chain",
    "pebblegpu_decimator_dec_by2_stages", "pebblegpu_decimator_process",
    "pebblegpu_downconvert_create", "pebblegpu_downconvert_destroy", "pebblegpu_downconvert_set_data_rate", "pebblegpu_downconvert_set_frequency",
    "pebblegpu_downconvert_set_cw_offset", "pebblegpu_downconvert_stages", "pebblegpu_downconvert_process", "pebblegpu_downconvert_process_device",
    "pebblegpu_downconvert_synchronize",
    "pebblegpu_fastfir_create", "pebblegpu_fastfir_destroy", "pebblegpu_fastfir_setup", "pebblegpu_fastfir_process",
    "pebblegpu_demod_create", "pebblegpu_demod_destroy", "pebblegpu_demod_set_mode", "pebblegpu_demod_set_bandwidth",
    "pebblegpu_demod_process", "pebblegpu_demod_rds_groups", "pebblegpu_demod_rds_signal", "pebblegpu_demod_stereo_lock",
    "pebblegpu_spectrum_create", "pebblegpu_spectrum_destroy", "pebblegpu_spectrum_bins", "pebblegpu_spectrum_process",
]


class PebbleGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libpebblegpu error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("sample_rate", C.c_double),
        ("frames_per_buffer", C.c_uint32), ("n_channels", C.c_uint32), ("shared_input", C.c_uint32),
        ("wfm", C.c_uint32), ("spectrum_bins", C.c_uint32), ("fastfir_fft", C.c_uint32),
        ("fastfir_taps", C.c_uint32), ("max_superframes", C.c_uint32), ("audio_rate", C.c_uint32), ("hires_bins", C.c_uint32),
        ("reserved", C.c_uint32 * 3),
    ]


class StreamBankConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("sample_rate", C.c_double), ("n_streams", C.c_uint32),
        ("frame", C.c_uint32), ("spectrum_bins", C.c_uint32), ("fastfir_fft", C.c_uint32), ("fastfir_taps", C.c_uint32),
        ("max_frames", C.c_uint32), ("reserved", C.c_uint32 * 5),
    ]


class Info(C.Structure):
    _fields_ = [
        ("demod_rate", C.c_double), ("demod_rate_int", C.c_uint32), ("dec_by2_stages", C.c_uint32),
        ("total_decimation", C.c_uint32), ("chain_len", C.c_uint32), ("stage_taps", C.c_uint32 * 16),
        ("stage_stride", C.c_uint32 * 16), ("superframe", C.c_uint64), ("n_streams", C.c_uint32),
        ("spectrum_bins", C.c_uint32),
    ]


def library_path():
    return os.path.join(_HERE, _LIBNAME)


_lib = None


def _declare(L):
    vp, u32, u64, i32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
    dp = C.POINTER(C.c_double)
    L.pebblegpu_last_error.restype = C.c_char_p
    L.pebblegpu_malloc.argtypes = [i32, C.c_size_t, C.POINTER(vp)]
    L.pebblegpu_free.argtypes = [i32, vp]
    L.pebblegpu_memcpy_h2d.argtypes = [i32, vp, vp, C.c_size_t]
    L.pebblegpu_memcpy_d2h.argtypes = [i32, vp, vp, C.c_size_t]
    L.pebblegpu_memset.argtypes = [i32, vp, i32, C.c_size_t]
    L.pebblegpu_device_synchronize.argtypes = [i32]
    L.pebblegpu_normalize_iq.argtypes = [i32, i32, i32, dbl, vp, u64, vp]
    L.pebblegpu_probe_copy_gbps.argtypes = [i32, i32, C.c_size_t, i32, C.POINTER(C.c_float)]
    L.pebblegpu_receiver_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.pebblegpu_receiver_destroy.argtypes = [vp]
    L.pebblegpu_receiver_info.argtypes = [vp, C.POINTER(Info)]
    L.pebblegpu_set_mixer_freq.argtypes = [vp, u32, dbl]
    L.pebblegpu_set_bandpass.argtypes = [vp, u32, dbl, dbl]
    L.pebblegpu_set_demod_mode.argtypes = [vp, u32, i32]
    L.pebblegpu_set_agc.argtypes = [vp, u32, i32, i32]
    L.pebblegpu_set_conditioners.argtypes = [vp, u32, i32, dbl, dbl]
    L.pebblegpu_set_noise_filter.argtypes = [vp, u32, i32]
    L.pebblegpu_set_squelch.argtypes = [vp, u32, C.c_double]
    L.pebblegpu_receiver_process.argtypes = [vp, vp, u64]
    L.pebblegpu_receiver_process_raw.argtypes = [vp, i32, i32, C.c_double, vp, u64]
    L.pebblegpu_receiver_audio.restype = vp
    L.pebblegpu_receiver_audio.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.pebblegpu_receiver_spectrum.restype = vp
    L.pebblegpu_receiver_spectrum.argtypes = [vp, C.POINTER(u64)]
    L.pebblegpu_receiver_zoom_spectrum.restype = vp
    L.pebblegpu_receiver_zoom_spectrum.argtypes = [vp, C.POINTER(u64), C.POINTER(u32)]
    L.pebblegpu_receiver_last_ms.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.pebblegpu_receiver_mean_ms.argtypes = [vp, i32, u32, C.POINTER(C.c_float)]
    L.pebblegpu_receiver_set_profiling.argtypes = [vp, i32]
    L.pebblegpu_receiver_ingest_acquire.argtypes = [vp, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p)]
    L.pebblegpu_receiver_ingest_submit.argtypes = [vp, C.c_uint32, C.c_uint64]
    L.pebblegpu_receiver_process_ingested.argtypes = [vp, C.c_uint32, i32, i32, C.c_double, C.c_uint64]
    L.pebblegpu_receiver_kernel_name.restype = C.c_char_p
    L.pebblegpu_receiver_kernel_name.argtypes = [vp, i32]
    L.pebblegpu_receiver_enable_signal_strength.argtypes = [vp, i32]
    L.pebblegpu_receiver_signal_strength.restype = vp
    L.pebblegpu_receiver_signal_strength.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.pebblegpu_receiver_synchronize.argtypes = [vp]
    L.pebblegpu_process_iq.argtypes = [vp, dp, C.c_uint16, dp, C.POINTER(u32), dp]
    L.pebblegpu_streambank_create.argtypes = [C.POINTER(StreamBankConfig), C.POINTER(vp)]
    L.pebblegpu_streambank_destroy.argtypes = [vp]
    L.pebblegpu_streambank_set_bandpass.argtypes = [vp, u32, dbl, dbl]
    L.pebblegpu_streambank_process.argtypes = [vp, vp, u64, u32]
    L.pebblegpu_streambank_filtered.restype = vp
    L.pebblegpu_streambank_filtered.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.pebblegpu_streambank_spectrum.restype = vp
    L.pebblegpu_streambank_spectrum.argtypes = [vp, C.POINTER(u64), C.POINTER(u32)]
    L.pebblegpu_streambank_last_ms.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.pebblegpu_streambank_synchronize.argtypes = [vp]
    # stand-alone steps
    L.pebblegpu_mixer_create.argtypes = [i32, u32, u32, C.POINTER(vp)]
    L.pebblegpu_mixer_destroy.argtypes = [vp]
    L.pebblegpu_mixer_set_frequency.argtypes = [vp, dbl]
    L.pebblegpu_mixer_process.argtypes = [vp, dp, C.POINTER(dp)]
    L.pebblegpu_decimator_create.argtypes = [i32, u32, u32, C.POINTER(vp)]
    L.pebblegpu_decimator_destroy.argtypes = [vp]
    L.pebblegpu_decimator_build_chain.argtypes = [vp, u32, u32, u32, C.POINTER(C.c_float)]
    L.pebblegpu_decimator_dec_by2_stages.argtypes = [vp, C.POINTER(u32)]
    L.pebblegpu_decimator_process.argtypes = [vp, dp, dp, u32, C.POINTER(u32)]
    L.pebblegpu_downconvert_create.argtypes = [i32, u32, C.POINTER(vp)]
    L.pebblegpu_downconvert_destroy.argtypes = [vp]
    L.pebblegpu_downconvert_set_data_rate.argtypes = [vp, C.c_double, C.c_double, i32, C.POINTER(C.c_double)]
    L.pebblegpu_downconvert_set_frequency.argtypes = [vp, C.c_double]
    L.pebblegpu_downconvert_set_cw_offset.argtypes = [vp, C.c_double]
    L.pebblegpu_downconvert_stages.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), u32]
    L.pebblegpu_downconvert_process.argtypes = [vp, u32, dp, dp, C.POINTER(u32)]
    L.pebblegpu_downconvert_process_device.argtypes = [vp, vp, u32, C.POINTER(vp), C.POINTER(u32)]
    L.pebblegpu_downconvert_synchronize.argtypes = [vp]
    L.pebblegpu_fastfir_create.argtypes = [i32, u32, u32, C.POINTER(vp)]
    L.pebblegpu_fastfir_destroy.argtypes = [vp]
    L.pebblegpu_fastfir_setup.argtypes = [vp, dbl, dbl, dbl, dbl]
    L.pebblegpu_fastfir_process.argtypes = [vp, i32, dp, dp, C.POINTER(i32)]
    L.pebblegpu_demod_create.argtypes = [i32, u32, u32, u32, C.POINTER(vp)]
    L.pebblegpu_demod_destroy.argtypes = [vp]
    L.pebblegpu_demod_set_mode.argtypes = [vp, i32]
    L.pebblegpu_demod_set_bandwidth.argtypes = [vp, dbl]
    L.pebblegpu_demod_process.argtypes = [vp, dp, i32, C.POINTER(dp)]
    L.pebblegpu_demod_rds_groups.argtypes = [vp, vp, vp, u32, C.POINTER(u32)]
    L.pebblegpu_demod_rds_signal.argtypes = [vp, dp, u32, C.POINTER(u32)]
    L.pebblegpu_demod_stereo_lock.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.pebblegpu_receiver_stereo_lock.argtypes = [vp, u32, C.POINTER(i32), C.POINTER(i32)]
    L.pebblegpu_receiver_rds_groups.argtypes = [vp, u32, vp, vp, u32, C.POINTER(u32)]
    L.pebblegpu_spectrum_create.argtypes = [i32, u32, dbl, u32, C.POINTER(vp)]
    L.pebblegpu_spectrum_destroy.argtypes = [vp]
    L.pebblegpu_spectrum_bins.argtypes = [vp, C.POINTER(u32)]
    L.pebblegpu_spectrum_process.argtypes = [vp, dp, i32, dp, C.POINTER(i32)]
    return L


def load_library(path=None):
    """Load libpebblegpu.so (built in-tree by __graft_entry__.build()).  No fallback of any kind."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    p = path or library_path()
    if not os.path.exists(p):
        raise PebbleGpuError(-2, "%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                 "(hipcc --offload-arch=gfx950); there is no CPU implementation" % p)
    L = _declare(C.CDLL(p))
    if path is None:
        _lib = L
    return L


def check(L, rc):
    if rc != 0:
        raise PebbleGpuError(rc, (L.pebblegpu_last_error() or b"").decode("utf-8", "replace"))


def probe_copy_gbps(lane_bytes=16, nbytes=1 << 30, iters=10, device=0, lib=None):
    L = lib or load_library()
    g = C.c_float()
    check(L, L.pebblegpu_probe_copy_gbps(device, lane_bytes, nbytes, iters, C.byref(g)))
    return g.value


IQ_S8, IQ_U8, IQ_S16, IQ_F32, IQ_WAV16 = range(5)
IQO_IQ, IQO_QI, IQO_IONLY, IQO_QONLY = range(4)


def normalize_iq(raw, fmt, order=IQO_IQ, gain=1.0, device=0, lib=None):
    """raw: numpy array of interleaved I,Q in the device's native type -> complex64 array converted on the GPU"""
    L = lib or load_library()
    raw = np.ascontiguousarray(raw)
    n = raw.size // 2
    src = DeviceBuffer.from_array(raw, device, L)
    dst = DeviceBuffer(8 * n, device, L)
    try:
        check(L, L.pebblegpu_normalize_iq(device, fmt, order, float(gain), C.c_void_p(src.ptr), n, C.c_void_p(dst.ptr)))
        return dst.download(np.complex64, n)
    finally:
        src.free()
        dst.free()


class DeviceBuffer:
    """A device allocation owned through the C ABI (pebblegpu_malloc / pebblegpu_free)."""

    def __init__(self, nbytes, device=0, lib=None):
        self.L = lib or load_library()
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(self.L, self.L.pebblegpu_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_array(cls, a, device=0, lib=None):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes, device, lib)
        b.upload(a)
        return b

    def upload(self, a, offset=0):
        a = np.ascontiguousarray(a)
        check(self.L, self.L.pebblegpu_memcpy_h2d(self.device, C.c_void_p(self.ptr + offset), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def download(self, dtype, count, offset=0):
        out = np.empty(count, dtype=dtype)
        check(self.L, self.L.pebblegpu_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset), out.nbytes))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            self.L.pebblegpu_free(self.device, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def to_f32_iq(x):
    """complex array -> interleaved float32 (n, 2) device layout"""
    x = np.asarray(x)
    if x.dtype == np.complex64:
        return np.ascontiguousarray(x).view(np.float32)
    return np.ascontiguousarray(x.astype(np.complex64)).view(np.float32)


class ReceiverBank:
    """C tuned channels over one shared stream or C independent streams (pebblegpu_receiver_*)."""

    def __init__(self, sample_rate, n_channels=1, shared_input=True, wfm=False, spectrum_bins=0,
                 frames_per_buffer=2048, fastfir_fft=0, fastfir_taps=0, max_superframes=1, device=0, lib=None, audio_rate=0, hires_bins=0):
        self.L = lib or load_library()
        cfg = Config()
        cfg.struct_size = C.sizeof(Config)
        cfg.device = device
        cfg.sample_rate = float(sample_rate)
        cfg.frames_per_buffer = frames_per_buffer
        cfg.n_channels = n_channels
        cfg.shared_input = 1 if shared_input else 0
        cfg.wfm = 1 if wfm else 0
        cfg.spectrum_bins = spectrum_bins
        cfg.fastfir_fft = fastfir_fft
        cfg.fastfir_taps = fastfir_taps
        cfg.max_superframes = max_superframes
        cfg.audio_rate = audio_rate
        cfg.hires_bins = hires_bins
        self.h = C.c_void_p()
        check(self.L, self.L.pebblegpu_receiver_create(C.byref(cfg), C.byref(self.h)))
        self.device = device
        self.n_channels = n_channels
        self.nf = frames_per_buffer
        info = Info()
        check(self.L, self.L.pebblegpu_receiver_info(self.h, C.byref(info)))
        self.info = info
        self.superframe = int(info.superframe)
        self.n_streams = int(info.n_streams)
        self.bins = int(info.spectrum_bins)
        self.D = int(info.total_decimation)

    def chain(self):
        return [(int(self.info.stage_taps[i]), int(self.info.stage_stride[i])) for i in range(self.info.chain_len)]

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.pebblegpu_receiver_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mixer(self, ch, f):
        check(self.L, self.L.pebblegpu_set_mixer_freq(self.h, ch, float(f)))

    def set_bandpass(self, ch, lo, hi):
        check(self.L, self.L.pebblegpu_set_bandpass(self.h, ch, float(lo), float(hi)))

    def set_mode(self, ch, mode):
        check(self.L, self.L.pebblegpu_set_demod_mode(self.h, ch, int(mode)))

    def stereo_lock(self, ch):
        """Demod_WFM::getStereoLock of a dmFMS channel -> (pilot lock of the last frame, changed since the last call)"""
        lk, chg = C.c_int32(0), C.c_int32(0)
        check(self.L, self.L.pebblegpu_receiver_stereo_lock(self.h, ch, C.byref(lk), C.byref(chg)))
        return bool(lk.value), bool(chg.value)

    def rds_groups(self, ch, cap=4096):
        """dmFMS channel of a WFM bank: what Demod::fmStereo popped from the RDS group queue since the last call ->
        ((n, 4) uint16 blocks A..D, (n,) bool: getNextRdsGroupData's return value)"""
        g = np.zeros((cap, 4), dtype=np.uint16)
        chg = np.zeros(cap, dtype=np.uint8)
        n = C.c_uint32(0)
        check(self.L, self.L.pebblegpu_receiver_rds_groups(self.h, ch, g.ctypes.data_as(C.c_void_p), chg.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return g[:n.value].copy(), chg[:n.value].astype(bool)

    def set_conditioners(self, stream, flags, iq_gain=1.0, iq_phase=0.0):
        check(self.L, self.L.pebblegpu_set_conditioners(self.h, stream, int(flags), float(iq_gain), float(iq_phase)))

    def set_noise_filter(self, ch, on=True):
        check(self.L, self.L.pebblegpu_set_noise_filter(self.h, ch, 1 if on else 0))

    def set_squelch(self, ch, squelch_db):
        """Receiver::squelchChanged: below squelch_db (avgDb of the latest spectrum) a call ends after the band-pass with no audio"""
        check(self.L, self.L.pebblegpu_set_squelch(self.h, ch, float(squelch_db)))

    def set_agc(self, ch, agc_mode, threshold):
        check(self.L, self.L.pebblegpu_set_agc(self.h, ch, int(agc_mode), int(threshold)))

    def process_device(self, dptr, n_samples):
        check(self.L, self.L.pebblegpu_receiver_process(self.h, C.c_void_p(dptr), int(n_samples)))

    def process_raw_device(self, dptr, n_samples, fmt, iq_order=0, gain=1.0):
        """raw device-format IQ pairs (pebblegpu_iq_format) already on the device -> normalizeIQ + the full call"""
        check(self.L, self.L.pebblegpu_receiver_process_raw(self.h, int(fmt), int(iq_order), float(gain), C.c_void_p(dptr), int(n_samples)))

    def ingest_buffer(self, slot, nbytes, dtype=np.int8):
        """the slot's pinned host buffer as a numpy array (valid until the slot is acquired again with a larger size)"""
        p = C.c_void_p()
        check(self.L, self.L.pebblegpu_receiver_ingest_acquire(self.h, int(slot), int(nbytes), C.byref(p)))
        n = int(nbytes) // np.dtype(dtype).itemsize
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,))

    def ingest_submit(self, slot, nbytes):
        check(self.L, self.L.pebblegpu_receiver_ingest_submit(self.h, int(slot), int(nbytes)))

    def process_ingested(self, slot, n_samples, fmt, iq_order=0, gain=1.0):
        check(self.L, self.L.pebblegpu_receiver_process_ingested(self.h, int(slot), int(fmt), int(iq_order), float(gain), int(n_samples)))

    def synchronize(self):
        check(self.L, self.L.pebblegpu_receiver_synchronize(self.h))

    def enable_signal_strength(self, on=True):
        check(self.L, self.L.pebblegpu_receiver_enable_signal_strength(self.h, 1 if on else 0))

    def signal_strength(self):
        """-> float32 [C, frames, 4] = (peakDb, avgDb, snrDb, floorDb) of the last call"""
        f, pitch = C.c_uint64(), C.c_uint64()
        p = self.L.pebblegpu_receiver_signal_strength(self.h, C.byref(f), C.byref(pitch))
        self.synchronize()
        out = np.empty((self.n_channels, int(f.value), 4), dtype=np.float32)
        for c in range(self.n_channels):
            check(self.L, self.L.pebblegpu_memcpy_d2h(self.device, out[c].ctypes.data_as(C.c_void_p), C.c_void_p(p + c * int(pitch.value) * 16), out[c].nbytes))
        return out

    def set_profiling(self, per_kernel=True):
        check(self.L, self.L.pebblegpu_receiver_set_profiling(self.h, 1 if per_kernel else 0))

    def last_ms(self, which=0):
        ms = C.c_float()
        check(self.L, self.L.pebblegpu_receiver_last_ms(self.h, which, C.byref(ms)))
        return ms.value

    def kernel_name(self, which):
        return (self.L.pebblegpu_receiver_kernel_name(self.h, which) or b"").decode()

    def mean_ms(self, which=0, last_k=1):
        ms = C.c_float()
        check(self.L, self.L.pebblegpu_receiver_mean_ms(self.h, which, last_k, C.byref(ms)))
        return ms.value

    def audio(self):
        """-> complex64 [C, n] of the last call"""
        n, pitch = C.c_uint64(), C.c_uint64()
        p = self.L.pebblegpu_receiver_audio(self.h, C.byref(n), C.byref(pitch))
        n, pitch = int(n.value), int(pitch.value)
        self.synchronize()
        out = np.empty((self.n_channels, n), dtype=np.complex64)
        for c in range(self.n_channels):  # rows are pitched on the device
            check(self.L, self.L.pebblegpu_memcpy_d2h(self.device, out[c].ctypes.data_as(C.c_void_p), C.c_void_p(p + c * pitch * 8), n * 8))
        return out

    def spectrum(self):
        """-> float32 [streams, frames, bins] of the last call"""
        n = C.c_uint64()
        p = self.L.pebblegpu_receiver_spectrum(self.h, C.byref(n))
        frames = int(n.value)
        self.synchronize()
        out = np.empty((self.n_streams, frames, self.bins), dtype=np.float32)
        check(self.L, self.L.pebblegpu_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(p), out.nbytes))
        return out

    def zoom_spectrum(self):
        """-> float32 [C, frames, hires_bins]: SignalSpectrum::zoomed of every decimated frame of the last call"""
        f, b = C.c_uint64(), C.c_uint32()
        p = self.L.pebblegpu_receiver_zoom_spectrum(self.h, C.byref(f), C.byref(b))
        self.synchronize()
        out = np.empty((self.n_channels, int(f.value), int(b.value)), dtype=np.float32)
        if out.size:
            check(self.L, self.L.pebblegpu_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(p), out.nbytes))
        return out

    def process(self, iq):
        """iq: complex [streams, n] (or [n] for one stream).  Returns (audio [C, n/D], spectrum or None)."""
        iq = np.atleast_2d(np.asarray(iq))
        assert iq.shape[0] == self.n_streams, "expected %d streams" % self.n_streams
        buf = DeviceBuffer.from_array(to_f32_iq(iq), self.device, self.L)
        try:
            self.process_device(buf.ptr, iq.shape[1])
            a = self.audio()
            s = self.spectrum() if self.bins else None
        finally:
            buf.free()
        return a, s

    def process_iq(self, frame, want_spectrum=False):
        """Host single-frame path (CB_ProcessIQData shape).  -> (audio complex128 [n_audio], spectrum or None)"""
        x = np.ascontiguousarray(frame, dtype=np.complex128)
        dp = C.POINTER(C.c_double)
        cap = max(self.nf, self.superframe // self.D)
        audio = np.empty(cap, dtype=np.complex128)
        spec = np.empty(self.bins, dtype=np.float64) if (want_spectrum and self.bins) else None
        n_audio = C.c_uint32()
        check(self.L, self.L.pebblegpu_process_iq(self.h, x.ctypes.data_as(dp), len(x), audio.ctypes.data_as(dp), C.byref(n_audio),
                                                 spec.ctypes.data_as(dp) if spec is not None else None))
        return audio[: n_audio.value].copy(), spec


class StreamBank:
    """S full-rate streams through the overlap-save band-pass and the display transform (pebblegpu_streambank_*)."""

    BANDPASS, SPECTRUM = 1, 2

    def __init__(self, sample_rate, n_streams, frame=65536, spectrum_bins=65536, fastfir_fft=0, fastfir_taps=0,
                 max_frames=1, device=0, lib=None):
        self.L = lib or load_library()
        cfg = StreamBankConfig()
        cfg.struct_size = C.sizeof(StreamBankConfig)
        cfg.device = device
        cfg.sample_rate = float(sample_rate)
        cfg.n_streams = n_streams
        cfg.frame = frame
        cfg.spectrum_bins = spectrum_bins
        cfg.fastfir_fft = fastfir_fft
        cfg.fastfir_taps = fastfir_taps
        cfg.max_frames = max_frames
        self.h = C.c_void_p()
        check(self.L, self.L.pebblegpu_streambank_create(C.byref(cfg), C.byref(self.h)))
        self.device, self.n_streams, self.frame = device, n_streams, frame

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.pebblegpu_streambank_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bandpass(self, stream, lo, hi):
        check(self.L, self.L.pebblegpu_streambank_set_bandpass(self.h, stream, float(lo), float(hi)))

    def process_device(self, dptr, n_samples, what=3):
        check(self.L, self.L.pebblegpu_streambank_process(self.h, C.c_void_p(dptr), int(n_samples), int(what)))

    def synchronize(self):
        check(self.L, self.L.pebblegpu_streambank_synchronize(self.h))

    def last_ms(self, which=0):
        ms = C.c_float()
        check(self.L, self.L.pebblegpu_streambank_last_ms(self.h, which, C.byref(ms)))
        return ms.value

    def spectrum_kernels(self):
        """label of the kernels behind last_ms(2)"""
        return ("k_big_cols + k_big_rows" if os.environ.get("PEBBLEGPU_BIG_SPLIT32") == "1" else "k_big256_cols + k_big256_rows") if self.frame == 65536 else "k_spectrum"

    def filtered(self):
        n, pitch = C.c_uint64(), C.c_uint64()
        p = self.L.pebblegpu_streambank_filtered(self.h, C.byref(n), C.byref(pitch))
        self.synchronize()
        out = np.empty((self.n_streams, int(n.value)), dtype=np.complex64)
        if out.size:
            check(self.L, self.L.pebblegpu_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(p), out.nbytes))
        return out

    def spectrum(self):
        f, b = C.c_uint64(), C.c_uint32()
        p = self.L.pebblegpu_streambank_spectrum(self.h, C.byref(f), C.byref(b))
        self.synchronize()
        out = np.empty((self.n_streams, int(f.value), int(b.value)), dtype=np.float32)
        if out.size:
            check(self.L, self.L.pebblegpu_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(p), out.nbytes))
        return out

    def process(self, iq, what=3):
        """iq: complex [streams, n] -> (filtered [S, n] or None, spectrum [S, frames, bins] or None)"""
        iq = np.atleast_2d(np.asarray(iq))
        assert iq.shape[0] == self.n_streams
        buf = DeviceBuffer.from_array(to_f32_iq(iq), self.device, self.L)
        try:
            self.process_device(buf.ptr, iq.shape[1], what)
            y = self.filtered() if what & 1 else None
            s = self.spectrum() if what & 2 else None
        finally:
            buf.free()
        return y, s
