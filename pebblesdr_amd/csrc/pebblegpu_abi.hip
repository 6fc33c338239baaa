// pebblegpu_abi.hip -- extern "C" surface declared in include/pebblegpu.h (receiver bank + memory plumbing).
#include <cmath>
#include <new>
#include "receiver.h"

struct pebblegpu_receiver {
    pg::Receiver rx;
};

using pg::fail;
namespace pg {
int probe_copy(int lane_bytes, size_t bytes, int iters, float *gbps);
}

extern "C" {

const char *pebblegpu_last_error(void) { return pg::last_error().c_str(); }
int pebblegpu_abi_version(void) { return PEBBLEGPU_ABI_VERSION; }

int pebblegpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int need_device(int device)
{
    int n = pebblegpu_device_count();
    if (n <= 0) return fail(PEBBLEGPU_E_NO_DEVICE, "no HIP device visible: libpebblegpu has no CPU path");
    if (device < 0 || device >= n) return fail(PEBBLEGPU_E_INVALID, "device %d out of range (0..%d)", device, n - 1);
    PG_HIP(hipSetDevice(device));
    return 0;
}

int pebblegpu_malloc(int device, size_t bytes, void **dptr)
{
    if (!dptr) return fail(PEBBLEGPU_E_INVALID, "null dptr");
    if (int rc = need_device(device)) return rc;
    PG_HIP(hipMalloc(dptr, bytes));
    return 0;
}
int pebblegpu_free(int device, void *dptr)
{
    if (int rc = need_device(device)) return rc;
    PG_HIP(hipFree(dptr));
    return 0;
}
// The receiver and stream-bank objects queue their work on private non-blocking streams, which a null-stream hipMemcpy does
// not order against: both copies first wait for everything queued on the device, so a host that refills an input buffer or
// reads an output through them can never race a process call (they block anyway; the wait is what makes them safe).
int pebblegpu_memcpy_h2d(int device, void *dst, const void *src, size_t bytes)
{
    if (int rc = need_device(device)) return rc;
    PG_HIP(hipDeviceSynchronize());
    PG_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
int pebblegpu_memcpy_d2h(int device, void *dst, const void *src, size_t bytes)
{
    if (int rc = need_device(device)) return rc;
    PG_HIP(hipDeviceSynchronize());
    PG_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}
int pebblegpu_memset(int device, void *dst, int value, size_t bytes)
{
    if (int rc = need_device(device)) return rc;
    PG_HIP(hipMemset(dst, value, bytes));
    return 0;
}
int pebblegpu_device_synchronize(int device)
{
    if (int rc = need_device(device)) return rc;
    PG_HIP(hipDeviceSynchronize());
    return 0;
}

int pebblegpu_probe_copy_gbps(int device, int lane_bytes, size_t bytes, int iters, float *gbps)
{
    if (!gbps || iters <= 0 || bytes < 4096 || (lane_bytes != 8 && lane_bytes != 16)) return fail(PEBBLEGPU_E_INVALID, "bad argument");
    if (int rc = need_device(device)) return rc;
    return pg::probe_copy(lane_bytes, bytes & ~(size_t)4095, iters, gbps);
}

int pebblegpu_normalize_iq(int device, int format, int iq_order, double gain, const void *d_src, uint64_t n_samples, void *d_dst)
{
    if (!d_src || !d_dst || format < 0 || format > PEBBLEGPU_IQ_WAV16 || iq_order < 0 || iq_order > 3 || n_samples == 0)
        return fail(PEBBLEGPU_E_INVALID, "bad argument");
    if (int rc = need_device(device)) return rc;
    return pg::run_normalize_iq(format, iq_order, gain, d_src, (long long)n_samples, (float2 *)d_dst, nullptr, true);
}

int pebblegpu_receiver_create(const pebblegpu_config *cfg, pebblegpu_receiver **out)
{
    if (!cfg || !out) return fail(PEBBLEGPU_E_INVALID, "null argument");
    if (cfg->struct_size != sizeof(pebblegpu_config)) return fail(PEBBLEGPU_E_INVALID, "pebblegpu_config size mismatch (ABI %d)", PEBBLEGPU_ABI_VERSION);
    if (int rc = need_device(cfg->device)) return rc;
    pebblegpu_receiver *h = new (std::nothrow) pebblegpu_receiver();
    if (!h) return fail(PEBBLEGPU_E_INVALID, "out of host memory");
    int rc = h->rx.create(cfg);
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return 0;
}

int pebblegpu_receiver_destroy(pebblegpu_receiver *rx)
{
    delete rx;
    return 0;
}

int pebblegpu_receiver_info(const pebblegpu_receiver *h, pebblegpu_info *info)
{
    if (!h || !info) return fail(PEBBLEGPU_E_INVALID, "null argument");
    const pg::Receiver &r = h->rx;
    memset(info, 0, sizeof(*info));
    info->demod_rate = (double)r.chain.rate;
    info->demod_rate_int = r.demod_rate_int;
    info->dec_by2_stages = r.chain.dec_by2;
    info->total_decimation = r.chain.total;
    info->chain_len = (uint32_t)r.chain.stages.size();
    for (size_t i = 0; i < r.chain.stages.size() && i < 16; i++) {
        info->stage_taps[i] = (uint32_t)r.chain.stages[i].ntaps;
        info->stage_stride[i] = r.chain.stages[i].stride;
    }
    info->superframe = r.superframe;
    info->n_streams = r.S;
    info->spectrum_bins = r.bins;
    return 0;
}

int pebblegpu_set_mixer_freq(pebblegpu_receiver *h, uint32_t channel, double freq_hz)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.set_mixer(channel, freq_hz);
}
int pebblegpu_set_bandpass(pebblegpu_receiver *h, uint32_t channel, double lo_hz, double hi_hz)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.set_bandpass(channel, lo_hz, hi_hz);
}
int pebblegpu_set_agc(pebblegpu_receiver *h, uint32_t channel, int agc_mode, int threshold)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.set_agc(channel, agc_mode, threshold);
}
int pebblegpu_set_conditioners(pebblegpu_receiver *h, uint32_t stream, int flags, double iq_gain, double iq_phase)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.set_conditioners(stream, flags, iq_gain, iq_phase);
}
int pebblegpu_set_noise_filter(pebblegpu_receiver *h, uint32_t channel, int on)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.set_noise_filter(channel, on != 0);
}
int pebblegpu_set_squelch(pebblegpu_receiver *h, uint32_t channel, double squelch_db)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.set_squelch(channel, squelch_db);
}
int pebblegpu_set_demod_mode(pebblegpu_receiver *h, uint32_t channel, int mode)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.set_mode(channel, mode);
}

int pebblegpu_receiver_rds_groups(pebblegpu_receiver *h, uint32_t channel, pebblegpu_rds_group *groups, uint8_t *changed, uint32_t cap, uint32_t *n)
{
    if (!h || !n) return fail(PEBBLEGPU_E_INVALID, "null argument");
    static_assert(sizeof(pebblegpu_rds_group) == sizeof(pg::RdsGroup), "group layout");
    return h->rx.rds_groups(channel, reinterpret_cast<pg::RdsGroup *>(groups), changed, cap, n);
}

int pebblegpu_receiver_stereo_lock(pebblegpu_receiver *h, uint32_t channel, int *pilot_lock, int *changed)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.stereo_lock(channel, pilot_lock, changed);
}

int pebblegpu_receiver_process(pebblegpu_receiver *h, const void *d_iq, uint64_t n_samples)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.process((const float2 *)d_iq, n_samples, h->rx.bins != 0, true);
}
int pebblegpu_receiver_process_raw(pebblegpu_receiver *h, int format, int iq_order, double gain, const void *d_raw, uint64_t n_samples)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.process_raw(format, iq_order, gain, d_raw, n_samples);
}
int pebblegpu_receiver_ingest_acquire(pebblegpu_receiver *h, uint32_t slot, uint64_t bytes, void **host_ptr)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.ingest_acquire(slot, bytes, host_ptr);
}
int pebblegpu_receiver_ingest_submit(pebblegpu_receiver *h, uint32_t slot, uint64_t bytes)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.ingest_submit(slot, bytes);
}
int pebblegpu_receiver_process_ingested(pebblegpu_receiver *h, uint32_t slot, int format, int iq_order, double gain, uint64_t n_samples)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.process_ingested(slot, format, iq_order, gain, n_samples);
}
const void *pebblegpu_receiver_audio(const pebblegpu_receiver *h, uint64_t *samples_per_channel, uint64_t *pitch_samples)
{
    if (!h) return nullptr;
    if (samples_per_channel) *samples_per_channel = h->rx.last_audio_n;
    if (pitch_samples) *pitch_samples = (uint64_t)h->rx.audio_pitch();
    return h->rx.audio_ptr();
}
const void *pebblegpu_receiver_spectrum(const pebblegpu_receiver *h, uint64_t *frames_per_stream)
{
    if (!h) return nullptr;
    if (frames_per_stream) *frames_per_stream = h->rx.last_spec_frames;
    return h->rx.d_spec;
}
const void *pebblegpu_receiver_zoom_spectrum(const pebblegpu_receiver *h, uint64_t *frames_per_channel, uint32_t *bins)
{
    if (!h || !h->rx.zoom_bins) return nullptr;
    if (frames_per_channel) *frames_per_channel = h->rx.last_zoom_frames;
    if (bins) *bins = h->rx.zoom_bins;
    return h->rx.d_zoom;
}
int pebblegpu_receiver_enable_signal_strength(pebblegpu_receiver *h, int on)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.enable_smeter(on != 0);
}
const void *pebblegpu_receiver_signal_strength(const pebblegpu_receiver *h, uint64_t *frames, uint64_t *pitch_frames)
{
    if (!h || !h->rx.smeter_on) return nullptr;
    if (frames) *frames = h->rx.last_spec_frames;
    if (pitch_frames) *pitch_frames = (uint64_t)h->rx.smeter_pitch;
    return h->rx.d_smeter;
}
int pebblegpu_receiver_synchronize(pebblegpu_receiver *h)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.sync();
}
// 0 whole call; 1 spectrum; 2 mixer+first stage; 3 later stages; 4 FastFIR; 5 demod
static int kernel_ms(const pebblegpu_receiver *h, int which, uint32_t last_k, float *ms)
{
    if (!h || !ms || which < 0 || which > 5) return fail(PEBBLEGPU_E_INVALID, "bad argument");
    if (int rc = const_cast<pebblegpu_receiver *>(h)->rx.close_timing()) return rc;
    const pg::Timers &t = h->rx.tm;
    if (t.calls == 0) return fail(PEBBLEGPU_E_INVALID, "no call has been made yet");
    if (last_k == 0) last_k = 1;
    if (last_k > (uint32_t)pg::Timers::kRing - 1) last_k = pg::Timers::kRing - 1;  // (the oldest slot's start may be an event the newest call has re-recorded)
    if ((uint64_t)last_k > t.calls) last_k = (uint32_t)t.calls;
    static const int a[6] = {0, 0, 1, 2, 3, 4}, b[6] = {6, 1, 2, 3, 4, 5};
    double sum = 0;
    for (uint32_t k = 0; k < last_k; k++) {
        const hipEvent_t *ev = t.ev[(t.calls - 1 - k) % pg::Timers::kRing];
        if (which >= 2 && !t.detailed[(t.calls - 1 - k) % pg::Timers::kRing])
            return fail(PEBBLEGPU_E_INVALID, "per-kernel times need pebblegpu_receiver_set_profiling(rx, 1) before the calls");
        float one = 0;
        const bool has_mid = t.has_mid[(t.calls - 1 - k) % pg::Timers::kRing];
        const hipEvent_t endev = t.end_ev[(t.calls - 1 - k) % pg::Timers::kRing];
        PG_HIP(hipEventSynchronize(endev));
        if (has_mid) PG_HIP(hipEventSynchronize(ev[1]));  // pipelined calls: the transform's stream ends on its own
        if (which == 1 && !has_mid) continue;  // no display transform in that call: 0 ms
        const hipEvent_t start = t.start_ev[(t.calls - 1 - k) % pg::Timers::kRing];
        PG_HIP(hipEventElapsedTime(&one, a[which] == 0 ? start : ev[a[which]], b[which] == 6 ? endev : ev[b[which]]));
        if (which == 0 && has_mid) {  // the call lasted until the later of its two pipelines
            float other = 0;
            PG_HIP(hipEventElapsedTime(&other, start, ev[1]));
            if (other > one) one = other;
        }
        sum += one;
    }
    *ms = (float)(sum / last_k);
    return 0;
}
const char *pebblegpu_receiver_kernel_name(const pebblegpu_receiver *h, int which)
{
    if (!h) return "";
    return h->rx.kernel_name(which);
}
int pebblegpu_receiver_set_profiling(pebblegpu_receiver *h, int per_kernel)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    h->rx.profile_detail = per_kernel != 0;
    return 0;
}
int pebblegpu_receiver_last_ms(const pebblegpu_receiver *h, int which, float *ms) { return kernel_ms(h, which, 1, ms); }
int pebblegpu_receiver_mean_ms(const pebblegpu_receiver *h, int which, uint32_t last_k, float *ms) { return kernel_ms(h, which, last_k, ms); }
int pebblegpu_process_iq(pebblegpu_receiver *h, const double *iq, uint16_t n, double *audio, uint32_t *n_audio,
                         double *spectrum_db)
{
    if (!h) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return h->rx.process_iq(iq, n, audio, n_audio, spectrum_db);
}

}  // extern "C"
