// streambank.hip -- S full-rate streams through CFastFIR + fftSpectrum (include/pebblegpu.h, "Stream bank").
#include <new>
#include "receiver.h"

using pg::fail;

struct pebblegpu_streambank {
    pebblegpu_streambank_config cfg{};
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // the band-pass of a call that also asks for the spectrum runs here, beside the transform
    bool side_ok = true, side = false;
    pg::FastFirCore ff;
    pg::SpectrumCore sp;
    float2 *d_tail = nullptr, *d_tail_alt = nullptr, *d_filt = nullptr;  // (the two overlap buffers swap after every band-pass call)
    float *d_spec = nullptr;
    uint64_t cap = 0, last_n = 0, last_frames = 0;
    hipEvent_t ev[4] = {};
    bool timed = false;
};

extern "C" {

int pebblegpu_streambank_destroy(pebblegpu_streambank *sb)
{
    if (!sb) return 0;
    (void)hipSetDevice(sb->cfg.device);
    if (sb->stream2) {
        (void)hipStreamSynchronize(sb->stream2);
        (void)hipStreamDestroy(sb->stream2);
    }
    if (sb->stream) {
        (void)hipStreamSynchronize(sb->stream);
        (void)hipStreamDestroy(sb->stream);
    }
    sb->ff.release();
    sb->sp.release();
    void *p[] = {sb->d_tail, sb->d_tail_alt, sb->d_filt, sb->d_spec};
    for (void *q : p) if (q) (void)hipFree(q);
    for (hipEvent_t e : sb->ev) if (e) (void)hipEventDestroy(e);
    delete sb;
    return 0;
}

int pebblegpu_streambank_create(const pebblegpu_streambank_config *cfg, pebblegpu_streambank **out)
{
    if (!cfg || !out) return fail(PEBBLEGPU_E_INVALID, "null argument");
    if (cfg->struct_size != sizeof(pebblegpu_streambank_config)) return fail(PEBBLEGPU_E_INVALID, "pebblegpu_streambank_config size mismatch");
    if (!cfg->n_streams || !cfg->frame || !cfg->max_frames || !(cfg->sample_rate > 0)) return fail(PEBBLEGPU_E_INVALID, "bad stream bank configuration");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PEBBLEGPU_E_NO_DEVICE, "no HIP device");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(PEBBLEGPU_E_INVALID, "device %d out of range", cfg->device);
    pebblegpu_streambank *sb = new (std::nothrow) pebblegpu_streambank();
    if (!sb) return fail(PEBBLEGPU_E_INVALID, "out of host memory");
    sb->cfg = *cfg;
    if (!sb->cfg.fastfir_fft) sb->cfg.fastfir_fft = 2048;
    if (!sb->cfg.fastfir_taps) sb->cfg.fastfir_taps = 1025;
    const uint32_t S = cfg->n_streams;
    sb->cap = (uint64_t)cfg->frame * cfg->max_frames;
    int rc = 0;
    auto body = [&]() -> int {
        PG_HIP(hipSetDevice(cfg->device));
        PG_HIP(hipStreamCreateWithFlags(&sb->stream, hipStreamNonBlocking));
        PG_HIP(hipStreamCreateWithFlags(&sb->stream2, hipStreamNonBlocking));
        { const char *e = getenv("PEBBLEGPU_SB_SIDE"); sb->side_ok = e && e[0] == '1'; }  // opt-in: measured equal (below)
        if (int r = sb->ff.init(S, sb->cfg.fastfir_fft, sb->cfg.fastfir_taps)) return r;
        if (int r = sb->sp.init(S, cfg->frame, cfg->spectrum_bins)) return r;
        if (cfg->frame % (uint64_t)sb->ff.block_len()) return fail(PEBBLEGPU_E_SIZE, "frame %u is not a multiple of the band-pass block %lld", cfg->frame, sb->ff.block_len());
        const size_t ov = sb->cfg.fastfir_taps - 1;
        PG_HIP(hipMalloc((void **)&sb->d_tail, sizeof(float2) * ov * S));
        PG_HIP(hipMemset(sb->d_tail, 0, sizeof(float2) * ov * S));  // m_pFFTOverlapBuf starts at zero, fastfir.cpp:104-105
        PG_HIP(hipMalloc((void **)&sb->d_tail_alt, sizeof(float2) * ov * S));
        PG_HIP(hipMemset(sb->d_tail_alt, 0, sizeof(float2) * ov * S));
        PG_HIP(hipMalloc((void **)&sb->d_filt, sizeof(float2) * sb->cap * S));
        PG_HIP(hipMalloc((void **)&sb->d_spec, sizeof(float) * (size_t)sb->sp.bins * cfg->max_frames * S));
        for (hipEvent_t &e : sb->ev) PG_HIP(hipEventCreate(&e));
        // CFastFIR's constructor state: lo -1, hi 1, offset 1, rate 1 -> an all-zero filter until the first setup
        return 0;
    };
    rc = body();
    if (rc) { pebblegpu_streambank_destroy(sb); return rc; }
    *out = sb;
    return 0;
}

int pebblegpu_streambank_set_bandpass(pebblegpu_streambank *sb, uint32_t stream, double lo, double hi)
{
    if (!sb) return fail(PEBBLEGPU_E_INVALID, "null argument");
    if (stream >= sb->cfg.n_streams) return fail(PEBBLEGPU_E_INVALID, "stream %u out of range", stream);
    PG_HIP(hipSetDevice(sb->cfg.device));
    bool ok = false;
    if (int rc = sb->ff.design(sb->stream, stream, lo, hi, 0.0, sb->cfg.sample_rate, &ok)) return rc;
    if (!ok) return fail(PEBBLEGPU_E_FILTER_PARAM, "Filter Parameter error (lo %g hi %g rate %g)", lo, hi, sb->cfg.sample_rate);
    return 0;
}

int pebblegpu_streambank_process(pebblegpu_streambank *sb, const void *d_iq, uint64_t n, uint32_t what)
{
    if (!sb || (!d_iq && n)) return fail(PEBBLEGPU_E_INVALID, "null argument");
    if (n % sb->cfg.frame) return fail(PEBBLEGPU_E_SIZE, "n_samples %llu is not a multiple of the frame %u", (unsigned long long)n, sb->cfg.frame);
    if (n > sb->cap) return fail(PEBBLEGPU_E_SIZE, "n_samples %llu above the capacity %llu", (unsigned long long)n, (unsigned long long)sb->cap);
    PG_HIP(hipSetDevice(sb->cfg.device));
    const float2 *in = static_cast<const float2 *>(d_iq);
    sb->last_n = 0;
    sb->last_frames = 0;
    if (n == 0) return 0;
    PG_HIP(hipEventRecord(sb->ev[0], sb->stream));
    // Both asked for: the band-pass (bound by its two transforms per block: vector units + LDS) and the display transform (the 65536-point
    // one is bound by what it moves through HBM) read the same input and share nothing else.  Side by side on two streams (fork at the
    // call's start event, join at its end; PEBBLEGPU_SB_SIDE=1 when the bank is created) they do NOT overlap: 0.4345 / 0.4368 ms per
    // configs[4] call against 0.4346 / 0.4422 one after the other -- the band-pass's 32768 small workgroups fill every CU's LDS first and the
    // transform's workgroups wait for them; with the band-pass's occupancy cut (8 / 16 / 30 kB of extra LDS per workgroup) both get slower
    // (0.46 / 0.48 / 0.52).  Opt-in, not the default.
    const bool side = sb->side_ok && (what & 3u) == 3u;
    sb->side = side;
    hipStream_t fs = side ? sb->stream2 : sb->stream;
    if (side) PG_HIP(hipStreamWaitEvent(fs, sb->ev[0], 0));
    if (what & 1u) {
        float2 *next = sb->ff.fft_n == 2048 ? sb->d_tail_alt : nullptr;
        if (int rc = sb->ff.run_ext(fs, in, (long long)n, sb->d_tail, (long long)n, sb->d_filt, (long long)n, next)) return rc;
        if (next) std::swap(sb->d_tail, sb->d_tail_alt);
        sb->last_n = n;
    }
    PG_HIP(hipEventRecord(sb->ev[1], fs));
    if (what & 2u) {
        const long long F = (long long)(n / sb->cfg.frame);
        if (int rc = sb->sp.run(sb->stream, in, (long long)n, F, sb->d_spec, nullptr, nullptr, !side)) return rc;
        sb->last_frames = (uint64_t)F;
    }
    if (side) {
        PG_HIP(hipEventRecord(sb->ev[3], sb->stream));       // where the transform ended
        PG_HIP(hipStreamWaitEvent(sb->stream, sb->ev[1], 0));  // join
    }
    PG_HIP(hipEventRecord(sb->ev[2], sb->stream));
    sb->timed = true;
    return 0;
}

const void *pebblegpu_streambank_filtered(const pebblegpu_streambank *sb, uint64_t *n, uint64_t *pitch)
{
    if (!sb) return nullptr;
    if (n) *n = sb->last_n;
    if (pitch) *pitch = sb->last_n;
    return sb->d_filt;
}
const void *pebblegpu_streambank_spectrum(const pebblegpu_streambank *sb, uint64_t *frames, uint32_t *bins)
{
    if (!sb) return nullptr;
    if (frames) *frames = sb->last_frames;
    if (bins) *bins = sb->sp.bins;
    return sb->d_spec;
}
int pebblegpu_streambank_last_ms(const pebblegpu_streambank *sb, int which, float *ms)
{
    if (!sb || !ms) return fail(PEBBLEGPU_E_INVALID, "null argument");
    if (!sb->timed) return fail(PEBBLEGPU_E_INVALID, "no process call yet");
    if (which < 0 || which > 2) return fail(PEBBLEGPU_E_INVALID, "which must be 0..2");
    PG_HIP(hipSetDevice(sb->cfg.device));
    PG_HIP(hipEventSynchronize(sb->ev[2]));
    // side by side both groups start at the call's start event; the transform's own end is ev[3]
    const int a = (which == 2 && !sb->side) ? 1 : 0, b = which == 1 ? 1 : (which == 2 && sb->side ? 3 : 2);
    PG_HIP(hipEventElapsedTime(ms, sb->ev[a], sb->ev[b]));
    return 0;
}
int pebblegpu_streambank_synchronize(pebblegpu_streambank *sb)
{
    if (!sb) return fail(PEBBLEGPU_E_INVALID, "null argument");
    PG_HIP(hipSetDevice(sb->cfg.device));
    PG_HIP(hipStreamSynchronize(sb->stream));
    PG_HIP(hipStreamSynchronize(sb->stream2));
    return 0;
}

}  // extern "C"
