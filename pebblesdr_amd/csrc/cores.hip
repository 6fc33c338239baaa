// cores.hip -- the device cores (see receiver.h): state allocation, parameter design upload, kernel launches.
// This is the only translation unit that instantiates kernels.
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <hip/hip_ext.h>
#include "kernels_demod.h"
#include "kernels_fastfir.h"
#include "kernels_frontend.h"
#include <algorithm>
#include "kernels_fused_dec.h"
#include "kernels_bank_dec.h"
#include "kernels_spectrum.h"
#include "receiver.h"

namespace pg {

std::string &last_error()
{
    static thread_local std::string e;
    return e;
}
int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

int HistBuf::alloc(int channels, int hist_len, long long capacity)
{
    hist = (hist_len + 1) & ~1;  // keep data 16-byte aligned
    cap = capacity;
    chans = channels;
    pitch = (hist + capacity + 1) & ~1LL;
    const size_t bytes = (size_t)pitch * channels * sizeof(float2);
    PG_HIP(hipMalloc((void **)&base, bytes));
    PG_HIP(hipMemset(base, 0, bytes));
    return 0;
}
void HistBuf::release()
{
    if (base) (void)hipFree(base);
    base = nullptr;
}

static design::M2 section_matrix(const ScanSection &s)
{
    if (s.type == kBiquadDf2) return design::M2{-s.c[3], -s.c[4], 1.0, 0.0};
    return design::M2{s.type == kOnePoleDiff ? s.c[0] : 1.0 - s.c[0], 0, 0, 0};
}

void fill_scan_section(ScanSection &s, int type, const double *c)
{
    memset(&s, 0, sizeof(s));
    s.type = type;
    for (int i = 0; i < (type == kBiquadDf2 ? 5 : 1); i++) s.c[i] = c[i];
    const design::M2 m = section_matrix(s);
    for (int k = 0; k < 6; k++) {
        const design::M2 p = design::m2_pow(m, (uint64_t)kSeg << k);
        s.P[k][0] = p.a; s.P[k][1] = p.b; s.P[k][2] = p.c; s.P[k][3] = p.d;
    }
}

// how many kSub-sample sub-chunks of zero-state warm-up make radius^samples < tol; -1 if more than 8
int scan_warm_subchunks(const ScanSection *secs, int nsec, double tol)
{
    double r = 0;
    for (int i = 0; i < nsec; i++) r = std::fmax(r, design::spectral_radius(section_matrix(secs[i])));
    if (r <= 0) return 1;
    if (r >= 1) return -1;
    const double need = std::log(tol) / std::log(r) * 1.5 + 64;  // margin for the non-normal transient
    const int subs = (int)std::ceil(need / kSub);
    return subs <= 8 ? (subs < 1 ? 1 : subs) : -1;
}

int make_twiddles(int n, float2 **d_tw)
{
    // per-pass tables in the layout fft_lds.h reads (fft_tw_off)
    if (!(n == 2048 || n == 4096 || n == 8192)) return fail(PEBBLEGPU_E_UNSUPPORTED, "no FFT plan for %d points", n);
    std::vector<float2> h((size_t)fft_tw_off(n, n));
    int P = 1;
    for (int pass = 0; pass < fft_passes(n); pass++) {
        const int R = fft_radix(n, pass);
        if (P > 1) {
            float2 *t = h.data() + fft_tw_off(n, P);
            for (int k = 0; k < P; k++)
                for (int e = 0; e < 3; e++) {
                    const long long idx = ((long long)k * (n / (P * R)) << e) % n;
                    const double a = -design::kTwoPi * (double)idx / (double)n;
                    t[e * P + k] = make_float2((float)std::cos(a), (float)std::sin(a));
                }
        }
        P *= R;
    }
    PG_HIP(hipMalloc((void **)d_tw, sizeof(float2) * h.size()));
    PG_HIP(hipMemcpy(*d_tw, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
    return 0;
}

int make_twiddles_t128(float2 **d_tw)
{
    // fft_t128.h: pass B W^{16k}, W^{32k}, W^{64k} (k < 16); pass C W^{k}, W^{2k}, W^{4k}, W^{8k} (k < 128); W = exp(-2 pi i / 2048)
    std::vector<float2> t2((size_t)kTw128Count);
    auto W = [](long long idx) {
        const double a = -design::kTwoPi * (double)(idx % 2048) / 2048.0;
        return make_float2((float)std::cos(a), (float)std::sin(a));
    };
    for (int k = 0; k < 16; k++)
        for (int e = 0; e < 3; e++) t2[kTw128B + e * 16 + k] = W((long long)(16 * k) << e);
    for (int k = 0; k < 128; k++)
        for (int e = 0; e < 4; e++) t2[kTw128C + e * 128 + k] = W((long long)k << e);
    PG_HIP(hipMalloc((void **)d_tw, sizeof(float2) * t2.size()));
    PG_HIP(hipMemcpy(*d_tw, t2.data(), sizeof(float2) * t2.size(), hipMemcpyHostToDevice));
    return 0;
}

// k_spectrum_t128 (8192 bins = four transforms x[n] W_8192^{n q} of a 2048-sample frame): one such table per q with the per-work-item
// part of that factor, W_8192^{t q}, folded in -- pass B's entries times W_8192^{16 q << e}, pass C's times W_8192^{q << e}
int make_twiddles_t128q(float2 **d_tw)
{
    std::vector<float2> t2((size_t)4 * kTw128Count);
    auto W = [](long long idx) {
        const double a = -design::kTwoPi * (double)(idx % 8192) / 8192.0;
        return make_float2((float)std::cos(a), (float)std::sin(a));
    };
    for (int q = 0; q < 4; q++) {
        float2 *t = t2.data() + (size_t)q * kTw128Count;
        for (int k = 0; k < 16; k++)
            for (int e = 0; e < 3; e++) t[kTw128B + e * 16 + k] = W((long long)(64 * k + 16 * q) << e);
        for (int k = 0; k < 128; k++)
            for (int e = 0; e < 4; e++) t[kTw128C + e * 128 + k] = W((long long)(4 * k + q) << e);
    }
    PG_HIP(hipMalloc((void **)d_tw, sizeof(float2) * t2.size()));
    PG_HIP(hipMemcpy(*d_tw, t2.data(), sizeof(float2) * t2.size(), hipMemcpyHostToDevice));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// OscBank
// ------------------------------------------------------------------------------------------------
int OscBank::init(uint32_t channels, double sample_rate)
{
    static_assert(sizeof(OscBank::Dyn) == kChanOscDynBytes && offsetof(ChanOsc, inc) == kChanOscDynBytes, "ChanOsc layout");
    C = channels;
    fs = sample_rate;
    ctl.assign(C, Ctl());
    h_osc.assign(C, ChanOsc());
    for (int i = 0; i < 2; i++) {
        PG_HIP(hipHostMalloc((void **)&h_dyn[i], sizeof(Dyn) * C, 0));
        PG_HIP(hipEventCreate(&h_done[i]));
    }
    PG_HIP(hipMalloc((void **)&d_osc, sizeof(ChanOsc) * C));
    PG_HIP(hipMemset(d_osc, 0, sizeof(ChanOsc) * C));
    std::vector<float> amp(kAmpTab);
    design::mixer_amplitudes(amp.data(), kAmpTab, &a_inf);
    PG_HIP(hipMalloc((void **)&d_amp, sizeof(float) * kAmpTab));
    PG_HIP(hipMemcpy(d_amp, amp.data(), sizeof(float) * kAmpTab, hipMemcpyHostToDevice));
    return 0;
}
void OscBank::release()
{
    for (int i = 0; i < 2; i++) {
        if (h_dyn[i]) (void)hipHostFree(h_dyn[i]);
        if (h_done[i]) (void)hipEventDestroy(h_done[i]);
        h_dyn[i] = nullptr;
        h_done[i] = nullptr;
    }
    if (d_osc) (void)hipFree(d_osc);
    if (d_amp) (void)hipFree(d_amp);
    if (d_adv) (void)hipFree(d_adv);
    d_adv = nullptr;
    d_osc = nullptr;
    d_amp = nullptr;
}
void OscBank::retune(uint32_t ch, double f)
{
    Ctl &c = ctl[ch];
    c.freq = f;
    c.inc = (-f) / fs;  // m_frequency = -f; m_oscInc = TWOPI*m_frequency/m_sampleRate  (mixer.cpp:31-34)
    c.phase0 = 0;       // m_lastOsc = (1, 0)  (mixer.cpp:37-38)
    c.n0 = 0;
    c.dirty = true;
    adv_stale = true;
}
// CDownConvert::SetFrequency (pebblelib/downconvert.cpp:100-112): a new increment for the oscillator as it stands -- phase and amplitude
// recurrence go on (m_Osc1 is not touched), where Mixer::setFrequency starts over
void OscBank::retune_keep(uint32_t ch, double f)
{
    Ctl &c = ctl[ch];
    c.freq = f;
    c.inc = (-f) / fs;
    c.dirty = true;
    adv_stale = true;
}
int OscBank::upload(hipStream_t s)
{
    // retuned channels: rebuild the constant part (phasor step tables) and upload the whole block, synchronously (rare)
    for (uint32_t ch = 0; ch < C; ch++) {
        Ctl &c = ctl[ch];
        if (!c.dirty) continue;
        ChanOsc &o = h_osc[ch];
        for (int d = 0; d < kMaxTaps; d++) {
            double ph = (double)d * c.inc;
            ph -= std::floor(ph);
            o.step[d] = make_float2((float)std::cos(design::kTwoPi * ph), (float)std::sin(design::kTwoPi * ph));
        }
        double ph = 512.0 * c.inc;
        ph -= std::floor(ph);
        o.step512 = make_float2((float)std::cos(design::kTwoPi * ph), (float)std::sin(design::kTwoPi * ph));
        o.inc = c.inc;
        o.mix_on = (force_mix || (-c.freq) != 0) ? 1u : 0u;  // if (m_frequency == 0) return in;  (mixer.cpp:51-53; CDownConvert has no such exit)
        o.phase0 = c.phase0;
        o.n0 = 0;
        PG_HIP(hipMemcpyAsync(d_osc + ch, &o, sizeof(ChanOsc), hipMemcpyHostToDevice, s));
        PG_HIP(hipStreamSynchronize(s));
        c.dirty = false;
        dyn_epoch++;
    }
    if (device_advance && dev_dyn_valid && C > (uint32_t)kOscInline) return 0;  // the previous call's tail launch advanced them on the device
    dyn_epoch++;
    // every call: phase and amplitude-transient position, 16 bytes per channel, from pinned ping-pong staging
    const int cur = h_idx;
    h_idx ^= 1;
    PG_HIP(hipEventSynchronize(h_done[cur]));  // copy issued two uploads ago: finished long since, never waits in steady state
    Dyn *hd = h_dyn[cur];
    for (uint32_t ch = 0; ch < C; ch++) {
        hd[ch].phase0 = ctl[ch].phase0;
        hd[ch].n0 = (uint32_t)(ctl[ch].n0 > (uint64_t)kAmpTab ? (uint64_t)kAmpTab : ctl[ch].n0);
        hd[ch].mix_on = h_osc[ch].mix_on;
    }
    inline_dyn.use = 0;
    if (allow_inline && C <= (uint32_t)kOscInline) {  // small bank: the consumer takes these 16 bytes per channel as kernel arguments
        for (uint32_t ch = 0; ch < C; ch++) {
            inline_dyn.d[ch].phase0 = hd[ch].phase0;
            inline_dyn.d[ch].n0 = hd[ch].n0;
            inline_dyn.d[ch].mix_on = hd[ch].mix_on;
        }
        inline_dyn.use = 1;
        return 0;
    }
    PG_HIP(hipMemcpy2DAsync(d_osc, sizeof(ChanOsc), hd, sizeof(Dyn), sizeof(Dyn), C, hipMemcpyHostToDevice, s));
    PG_HIP(hipEventRecord(h_done[cur], s));
    return 0;
}
int OscBank::advance_job(hipStream_t s, uint64_t n, OscAdvance *oa)
{
    memset(oa, 0, sizeof(*oa));
    dev_dyn_valid = false;
    if (!device_advance || C <= (uint32_t)kOscInline) return 0;
    if (!d_adv) PG_HIP(hipMalloc((void **)&d_adv, sizeof(double) * C));
    if (adv_stale || adv_n != n) {  // (rare: a retune or another call length)
        std::vector<double> a(C);
        for (uint32_t ch = 0; ch < C; ch++) {
            long double p = (long double)n * (long double)ctl[ch].inc;
            p -= floorl(p);
            a[ch] = (double)p;
        }
        PG_HIP(hipMemcpyAsync(d_adv, a.data(), sizeof(double) * C, hipMemcpyHostToDevice, s));
        PG_HIP(hipStreamSynchronize(s));
        adv_n = n;
        adv_stale = false;
    }
    oa->osc = d_osc;
    oa->adv = d_adv;
    oa->adv_n = (uint32_t)(n > 0xffffffffull ? 0xffffffffull : n);
    oa->osc_count = C;
    dev_dyn_valid = true;
    return 0;
}
bool OscBank::any_transient() const
{
    for (const auto &c : ctl)
        if (c.n0 < (uint64_t)kAmpTab) return true;
    return false;
}
void OscBank::advance(uint64_t n)
{
    for (auto &c : ctl) {
        long double p = (long double)c.phase0 + (long double)n * (long double)c.inc;
        p -= floorl(p);
        c.phase0 = (double)p;
        if (c.phase0 >= 1.0) c.phase0 = 0.0;
        c.n0 += n;
    }
}

int run_normalize_iq(int fmt, int order, double gain, const void *d_src, long long n, float2 *d_dst, hipStream_t s, bool wait, const float *final_scale)
{
    double scale = gain;
    if (final_scale != nullptr) scale = (double)*final_scale;  // the caller has already folded the format's constant in (a gain of 0 included)
    else if (fmt == 0 || fmt == 1) scale *= 1 / 128.0;   // deviceinterfacebase.cpp:651,689
    else if (fmt == 2) scale *= 1 / 32768.0;             // :729
    else if (fmt == 4) scale *= 1 / 32767.0;             // wavfile.cpp:299-300
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    launch(k_normalize_iq, dim3((unsigned)blocks), dim3(256), s, d_src, d_dst, n, fmt, order, (float)scale);
    PG_HIP(hipGetLastError());
    if (wait) PG_HIP(hipStreamSynchronize(s));
    return 0;
}

// streaming-copy probe: the chip's practical HBM ceiling next to which the kernels are priced
int probe_copy(int lane_bytes, size_t bytes, int iters, float *gbps)
{
    hipStream_t s;
    hipEvent_t e0, e1;
    void *a = nullptr, *b = nullptr;
    PG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    PG_HIP(hipEventCreate(&e0));
    PG_HIP(hipEventCreate(&e1));
    PG_HIP(hipMalloc(&a, bytes));
    PG_HIP(hipMalloc(&b, bytes));
    PG_HIP(hipMemsetAsync(a, 1, bytes, s));
    const dim3 grid(256 * 8), block(256);
    for (int it = -1; it < iters; it++) {
        if (it == 0) PG_HIP(hipEventRecord(e0, s));
        if (lane_bytes == 16) launch(k_probe_copy16, grid, block, s, (const float4 *)a, (float4 *)b, (long long)(bytes / 16));
        else launch(k_probe_copy8, grid, block, s, (const float2 *)a, (float2 *)b, (long long)(bytes / 8));
    }
    PG_HIP(hipEventRecord(e1, s));
    PG_HIP(hipEventSynchronize(e1));
    float ms = 0;
    PG_HIP(hipEventElapsedTime(&ms, e0, e1));
    *gbps = (float)(2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9);  // read + write
    (void)hipFree(a); (void)hipFree(b);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(s);
    return 0;
}

// stand-alone mixer launch (Mixer::processBlock), used by the Mixer step
int run_mixer(hipStream_t s, const float2 *d_in, float2 *d_out, long long n, const OscBank &osc)
{
    launch(k_mixer, dim3(cdiv(n, 1024), osc.C), dim3(256), s, d_in, n, 0, d_out, n, n, (const ChanOsc *)osc.d_osc,
           (const float *)osc.d_amp, osc.a_inf);
    PG_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// DecimCore: k_mix_dec1 (mixer + first merged stage) -> buf0 -> k_cascade (every later stage, fused) -> final
// ------------------------------------------------------------------------------------------------
static size_t mixdec_lds_bytes(const FirTaps &t)
{
    const int H = t.cic3 ? t.stride : t.ntaps - 1;
    const int span = 256 * t.stride + H;
    return (size_t)(span + span / t.stride + 2) * sizeof(float2);
}

// ---- k_mix_dec_mfma's instantiations: fronts (4 pairs: hb11 x S; 12: a merged CIC3 in front of it) x the halfband triples the reference's ladder
// puts behind them between 1 and 200 Msps (decimator.cpp:64-149) ----
struct BankVariant {
    int np, t1, t2, t3, hy, nstate, minw;
    void *kern;
};
template <int NP, int T1, int T2, int T3, int MINW>
static BankVariant bank_variant_of()
{
    using G = FusedDecGeom<T1, T2, T3>;
    return BankVariant{NP, T1, T2, T3, G::HY, G::N1 + G::N2 + G::N3, MINW, (void *)k_mix_dec_mfma<NP, T1, T2, T3, 0, MINW>};
}
static const std::vector<BankVariant> &bank_variants()
{
    static const std::vector<BankVariant> v = {
        bank_variant_of<4, 15, 19, 31, 2>(),  bank_variant_of<4, 15, 23, 43, 2>(),  bank_variant_of<4, 15, 23, 47, 2>(),  bank_variant_of<4, 15, 19, 35, 2>(),
        bank_variant_of<4, 15, 27, 59, 2>(),  bank_variant_of<4, 19, 27, 59, 2>(),  // (227 / 231 registers since the main loop stopped copying its fetched samples: half a SIMD's file)
        bank_variant_of<12, 15, 23, 47, 1>(), bank_variant_of<12, 15, 19, 35, 1>(), bank_variant_of<12, 15, 27, 59, 1>(),
    };
    return v;
}
bool bank_variant(int np, int t1, int t2, int t3, int *hy, int *nstate, int *minw)
{
    for (const BankVariant &b : bank_variants())
        if (b.np == np && b.t1 == t1 && b.t2 == t2 && b.t3 == t3) {
            *hy = b.hy; *nstate = b.nstate; *minw = b.minw;
            return true;
        }
    return false;
}

template <int NP>
static int launch_bank(void *kern, unsigned n_wg, hipStream_t s, const float2 *d_in, float2 *out, const ChanOsc *osc, const OscDynInline &dyn, const float2 *xh,
                       float2 *xh_out, const float2 *y0h, float2 *y0s, float2 *mixed, const BankDecParams<NP> &bp, hipEvent_t stop = nullptr)
{
    using K = void (*)(const float2 *, float2 *, const ChanOsc *, OscDynInline, const float2 *, float2 *, const float2 *, float2 *, float2 *, BankDecParams<NP>);
    // stop: an event that completes with THIS dispatch (the packet's own completion signal) -- recorded afterwards it would be a packet
    // of its own in the queue, ~6 us of idle GPU in front of the next call's decimator
    if (stop) hipExtLaunchKernelGGL(reinterpret_cast<K>(kern), dim3(n_wg), dim3(256), 0, s, nullptr, stop, 0, d_in, out, osc, dyn, xh, xh_out, y0h, y0s, mixed, bp);
    else launch(reinterpret_cast<K>(kern), dim3(n_wg), dim3(256), s, d_in, out, osc, dyn, xh, xh_out, y0h, y0s, mixed, bp);
    PG_HIP(hipGetLastError());
    return 0;
}

// the call's decimator as one launch of k_mix_dec_mfma (the caller has checked the sizes)
int DecimCore::run_bank_mfma(hipStream_t s, const float2 *d_in, long long n, const OscBank &osc, bool had_state, const OscAdvance *oa)
{
    const bool cic = fused_front;
    const HistBuf &y0b = cic ? buf1 : buf0;
    const int np = cic ? 12 : 4;
    const BankVariant *bv = nullptr;
    for (const BankVariant &b : bank_variants())
        if (b.np == np && b.t1 == casc.ntaps[0] && b.t2 == casc.ntaps[1] && b.t3 == casc.ntaps[2]) bv = &b;
    if (!bv) return fail(PEBBLEGPU_E_UNSUPPORTED, "no k_mix_dec_mfma instance for this chain");
    const long long g32 = cdiv(C, 32);
    const int warm = (bv->hy - 8) / 8;
    // The first-stage outputs in front of the call's start (block 0 needs seven of them even with the running sums handed over; the whole
    // look-back without): from the stage-0 head-room when the previous call took another route, else straight from what the previous
    // launch staged (two staging buffers, written alternately: this launch's history waves write the other one)
    const float2 *y0_hist = y0_pending ? (const float2 *)(d_y0stage2[y0_cur] + fused_hy) : (const float2 *)y0b.data();
    const long long y0_hist_pitch = y0_pending ? (long long)fused_hy : y0b.pitch;
    // the oscillators' advance rides on the launch when the caller handed it over (banks whose oscillators live on the device)
    static const int adv_mode = [] { const char *e = getenv("PEBBLEGPU_BANK_OSC_ADV"); return e ? atoi(e) : 1; }();  // 0: leave the oscillators to the tail launch (A/B)
    const bool adv = adv_mode != 0 && oa != nullptr && oa->osc != nullptr && oa->osc_count == C;
    if (adv && !d_dyn[0]) {
        for (int i = 0; i < 2; i++) PG_HIP(hipMalloc((void **)&d_dyn[i], sizeof(OscDyn) * C));
    }
    if (osc.dyn_epoch != dyn_epoch_seen) dyn_valid = false;
    dyn_epoch_seen = osc.dyn_epoch;
    // One wave per SIMD pays the fewest warm-up blocks; two overlap what a lone wave leaves idle (measured on hb11 x 4, 15/19/31: 1200
    // clocks per block alone, 2075 for each of two) -- worth it once a chunk is long against its warm-up: from 128 outputs per chunk on
    int waves = bank_waves;
    // (a receiver that runs two-stage calls keeps one wave per SIMD at every batch size: the previous call's band-pass needs the other
    // half of the register file beside it -- 0.2385 ms per configs[2] call of 32 super-frames against 0.2546, 0.875 against 0.905 at 128)
    if (waves == 0) waves = (!fin2.base && cdiv(len_out, 2 * std::max(1LL, 1024LL / g32)) >= 128) ? 2 : 1;
    if (waves > bv->minw) waves = bv->minw;  // (the instances with the longest halfbands need more than half a SIMD's registers)
    long long pairs_target = 1024LL * waves / g32;
    if (cic) pairs_target /= 2;  // (twelve pairs of lines per output instead of one window: the blocks are bound by what they fetch, and every chunk
                                 // fetches its 30 warm-up blocks again -- measured on configs[3]: 0.081 ms at 1024 waves, 0.075 at 512, 0.12 at 256)
    if (pairs_target < 1) pairs_target = 1;
    // a power of two (it divides the call's 2048 k outputs: the last chunk is a whole one), the nearest to the target above
    long long L = 16;
    if (fused_L > 0) {
        while (L * 2 <= fused_L) L *= 2;
    } else {
        const long long want = cdiv(len_out, 2 * pairs_target);
        while (L < want && L < 2048) L *= 2;
    }
    while (len_out % L != 0 && L > 16) L /= 2;
    while (2 * L <= warm) L *= 2;  // (only the first two chunks may reach in front of the call's start)
    const int S0 = first.stride, Sfs = cic ? S0 * wide_stride : S0;  // input samples per first-stage (hb11) output
    const long long pairs = cdiv(cdiv(len_out, L), 2);
    static const int hist_split = [] { const char *e = getenv("PEBBLEGPU_BANK_HSPLIT"); return e ? atoi(e) : 4; }();
    const unsigned n_wg = (unsigned)(8 * cdiv(pairs, 8) * cdiv(g32, 4) + cdiv(g32, 4) * hist_split);  // main workgroups, then the history waves'
    static unsigned long long *d_clk = nullptr;  // diagnosis only: PEBBLEGPU_BANK_CLK=1 prints the waves' clock counts of every such launch
    static size_t clk_cap = 0;
    const char *eclk = getenv("PEBBLEGPU_BANK_CLK");
    const bool want_clk = eclk && eclk[0] == '1';
    if (want_clk && clk_cap < (size_t)n_wg * 16) {
        if (d_clk) (void)hipFree(d_clk);
        PG_HIP(hipMalloc((void **)&d_clk, sizeof(unsigned long long) * n_wg * 16));
        clk_cap = (size_t)n_wg * 16;
    }
    if (want_clk) PG_HIP(hipMemsetAsync(d_clk, 0, sizeof(unsigned long long) * n_wg * 16, s));
    auto common = [&](auto &bp) {
        bp.n_out = len_out;
        bp.out_pitch = fin.pitch;
        bp.y0_pitch = y0_hist_pitch;
        bp.n_in = n;
        bp.S = Sfs;
        bp.L = (int)L;
        bp.n_chan = (int)C;
        bp.n_chunks = (int)cdiv(len_out, L);
        bp.n_groups = (int)g32;
        bp.hist_pitch = kMaxTaps;
        bp.xh = xh_depth;
        bp.a_inf = osc.a_inf;
        bp.gain = casc.gain;  // (the first stage's own gain is 1 in a chain of several stages)
        bp.hist_split = hist_split;
        bp.cic_s0 = cic ? S0 : 0;
        bp.state_in = had_state ? d_bank_state[bank_state_parity] : nullptr;
        bp.state_out = d_bank_state[bank_state_parity ^ 1];
        bp.clk = want_clk ? d_clk : nullptr;
        bp.dyn_in = adv && dyn_valid ? d_dyn[dyn_parity] : nullptr;
        bp.dyn_out = adv ? d_dyn[dyn_parity ^ 1] : nullptr;
        bp.osc_rw = adv ? oa->osc : nullptr;
        bp.adv = adv ? oa->adv : nullptr;
        bp.adv_n = adv ? oa->adv_n : 0;
    };
    const float2 *xh = d_xhist[hist_parity];
    float2 *xh_out = d_xhist[hist_parity ^ 1], *mixed = d_hist_mixed[hist_parity ^ 1];
    if (!cic) {
        BankDecParams<4> bp;
        memset(&bp, 0, sizeof(bp));
        common(bp);
        bp.min_off = -10;
        bp.max_off = 0;
        bp.ctr1 = -4.0;  // the hb11's centre tap sits on sample S j - 5; the oscillator of sample i is e^{j 2 pi (phase0 + (i + 1) inc)}
        static const int kE[4] = {0, 1, 3, 5};
        for (int p = 0; p < 4; p++) {
            bp.oa[p] = -5 + kE[p];
            bp.ob[p] = -5 - kE[p];
            bp.e[p] = (float)kE[p];
            bp.g[p] = p == 0 ? 0.5f * bank_taps.h[5] : bank_taps.h[5 + kE[p]];
        }
        void *kern = bv->kern;
        const char *edbg = getenv("PEBBLEGPU_BANK_DBG");  // timing experiments on the configs[2] instance (wrong results)
        const int dbg = edbg ? atoi(edbg) : 0;
        if (dbg && bv->t1 == 15 && bv->t2 == 19 && bv->t3 == 31)
            kern = dbg == 1 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 1> : dbg == 2 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 2> : dbg == 3 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 3>
                 : dbg == 4 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 4> : dbg == 8 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 8> : dbg == 16 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 16>
                 : dbg == 128 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 128> : dbg == 256 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 256> : dbg == 512 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 512> : dbg == 32 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 32> : dbg == 64 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 64> : dbg == 96 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 96> : dbg == 31 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 31> : dbg == 100 ? (void *)k_mix_dec_mfma<4, 15, 19, 31, 0, 1> : kern;
        if (dbg && bv->t1 == 19 && bv->t2 == 27 && bv->t3 == 59)
            kern = dbg == 1 ? (void *)k_mix_dec_mfma<4, 19, 27, 59, 1> : dbg == 2 ? (void *)k_mix_dec_mfma<4, 19, 27, 59, 2> : dbg == 4 ? (void *)k_mix_dec_mfma<4, 19, 27, 59, 4>
                 : dbg == 8 ? (void *)k_mix_dec_mfma<4, 19, 27, 59, 8> : dbg == 16 ? (void *)k_mix_dec_mfma<4, 19, 27, 59, 16> : dbg == 31 ? (void *)k_mix_dec_mfma<4, 19, 27, 59, 31> : kern;
        if (int rc = launch_bank<4>(kern, n_wg, s, d_in, fin.data(), (const ChanOsc *)osc.d_osc, osc.inline_dyn, xh, xh_out, y0_hist, d_y0stage2[y0_cur ^ 1], mixed, bp, done_event)) return rc;
    } else {
        // CIC3 at stride S0 in the reference's merged form (decimator.cpp:719-737: output k = .125 (od_k + ev_{k-1} + 3 (od_{k-1} + ev_k)) of the
        // sample pairs (ev, od)_P = x[S0 P], x[S0 P + 1]) under the hb11 at stride 16: relative to sample S j, pair q = -11 .. 0 carries
        // We[q] = (3 h[q+10] + h[q+11]) / 8 on its even and Wo[q] = (h[q+10] + 3 h[q+11]) / 8 on its odd sample; the response is symmetric about
        // (-11 S0 + 1) / 2 (We[q] = Wo[-11-q]), which makes twelve (later, earlier) sample pairs with one weight each
        BankDecParams<12> bp;
        memset(&bp, 0, sizeof(bp));
        common(bp);
        bp.min_off = -11 * S0;
        bp.max_off = 1;
        bp.ctr1 = 0.5 * (-11.0 * S0 + 1.0) + 1.0;
        auto h = [&](int d) { return d >= 0 && d <= 10 ? wide_fir.h[d] : 0.f; };
        for (int i = 0; i < 6; i++) {
            const int q = -11 + i;
            bp.oa[2 * i] = S0 * (-11 - q) + 1;  bp.ob[2 * i] = S0 * q;          bp.g[2 * i] = (3.f * h(q + 10) + h(q + 11)) * 0.125f;
            bp.oa[2 * i + 1] = S0 * (-11 - q);  bp.ob[2 * i + 1] = S0 * q + 1;  bp.g[2 * i + 1] = (h(q + 10) + 3.f * h(q + 11)) * 0.125f;
        }
        for (int p = 0; p < 12; p++) bp.e[p] = 0.5f * (float)(bp.oa[p] - bp.ob[p]);
        if (int rc = launch_bank<12>(bv->kern, n_wg, s, d_in, fin.data(), (const ChanOsc *)osc.d_osc, osc.inline_dyn, xh, xh_out, y0_hist, d_y0stage2[y0_cur ^ 1], mixed, bp, done_event)) return rc;
    }
    done_recorded = done_event != nullptr;
    if (want_clk) {
        std::vector<unsigned long long> h((size_t)n_wg * 16);
        PG_HIP(hipStreamSynchronize(s));
        PG_HIP(hipMemcpy(h.data(), d_clk, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        std::vector<double> per, ghz;
        std::vector<std::pair<double, size_t>> slow;
        double mx = 0;
        for (size_t w = 0; w < (size_t)n_wg * 4; w++)
            if (h[4 * w + 2]) {
                per.push_back((double)h[4 * w] / (double)h[4 * w + 2]);
                ghz.push_back((double)h[4 * w] / ((double)h[4 * w + 1] * 10.0));
                slow.push_back({per.back(), w});
                if ((double)h[4 * w + 1] > mx) mx = (double)h[4 * w + 1];
            }
        std::sort(per.begin(), per.end());
        std::sort(ghz.begin(), ghz.end());
        std::sort(slow.begin(), slow.end());
        if (!per.empty()) {
            fprintf(stderr, "k_mix_dec_mfma: %zu waves, L %d, clocks per block min %.0f median %.0f max %.0f; shader clock median %.2f GHz; longest wave %.1f us\n", per.size(),
                    (int)L, per.front(), per[per.size() / 2], per.back(), ghz[ghz.size() / 2], mx / 100.0);
            fprintf(stderr, "   slowest (clocks per block : workgroup.wave pair):");
            for (size_t i = slow.size() > 12 ? slow.size() - 12 : 0; i < slow.size(); i++)
                fprintf(stderr, " %.0f:%zu.%zu p%llu", slow[i].first, slow[i].second / 4, slow[i].second % 4, h[4 * slow[i].second + 3]);
            size_t over = 0;
            for (double v : per) over += v > 1.25 * per[per.size() / 2];
            fprintf(stderr, "\n   waves more than 25 %% over the median: %zu\n", over);
            fprintf(stderr, "   workgroups with a wave more than 8 %% over the median:");
            size_t last = (size_t)-1;
            for (size_t w = 0; w < (size_t)n_wg * 4; w++)
                if (h[4 * w + 2] && (double)h[4 * w] / (double)h[4 * w + 2] > 1.08 * per[per.size() / 2] && w / 4 != last) {
                    fprintf(stderr, " %zu", w / 4);
                    last = w / 4;
                }
            fprintf(stderr, "\n");
        }
    }
    hist_parity ^= 1;
    bank_state_parity ^= 1;
    bank_state_valid = true;
    front_name = "k_mix_dec_mfma";
    rest_name = "";
    last_fused = true;
    last_mfma = true;
    y0_cur ^= 1;
    y0_pending = true;
    // (with dyn_in the launch advanced the device blocks too; without, the caller's tail launch does and the next call finds dyn_out current)
    osc_advanced = adv && dyn_valid;
    if (adv) dyn_parity ^= 1;
    dyn_valid = adv;
    return 0;
}

// The first-stage history a k_mix_dec_mfma launch staged (d_y0stage) into the stage-0 head-room, where the other routes and a launch
// without the halfbands' running sums look for it.  Not needed between two such launches: queued only when a call wants it.
int DecimCore::flush_y0(hipStream_t s)
{
    if (!y0_pending) return 0;
    const HistBuf &y0b = fused_front ? buf1 : buf0;
    std::vector<TailJob> jobs;
    jobs.push_back(TailJob{d_y0stage2[y0_cur], (long long)fused_hy, (long long)fused_hy, fused_hy, 0, y0b.base + (y0b.hist - fused_hy), y0b.pitch});
    y0_pending = false;
    return run_save_tails(s, jobs, C, nullptr);
}

int DecimCore::init(uint32_t channels, const design::Chain &c, long long max_in, int last_hist, float last_gain)
{
    release();
    chain = c;
    C = channels;
    const size_t ns = chain.stages.size();
    if (ns - 1 > (size_t)kMaxCascade) return fail(PEBBLEGPU_E_UNSUPPORTED, "decimation chain has %zu stages; at most %d are built", ns, kMaxCascade + 1);
    const design::Stage &s0 = chain.stages[0];
    memset(&first, 0, sizeof(first));
    first.ntaps = s0.ntaps;
    first.stride = (int)s0.stride;
    first.cic3 = s0.ntaps == 0;
    first.gain = ns == 1 ? last_gain : 1.f;
    for (int p = 0; p < s0.ntaps; p++) first.h[p] = (float)design::halfband_taps(s0.design)[p];
    if (mixdec_lds_bytes(first) > 150 * 1024) return fail(PEBBLEGPU_E_UNSUPPORTED, "first-stage stride %u too wide for the LDS tile", s0.stride);
    // banks of >= 16 channels behind an hb11 first stage mix in registers when their input is one shared stream
    bank_front = !first.cic3 && first.ntaps == kFrontT1 && (C >= 16 || C == 1);
    memset(&bank_taps, 0, sizeof(bank_taps));
    bank_taps.stride = first.stride;
    for (int p = 0; p < kFrontT1 && p < first.ntaps; p++) bank_taps.h[p] = first.h[p];
    // a wide second stage is peeled off the fused cascade (see receiver.h)
    wide = ns >= 3 && chain.stages[1].stride >= 8 && chain.stages[1].ntaps > 0;
    size_t kfirst = 1;
    if (wide) {
        const design::Stage &w = chain.stages[1];
        wide_taps = w.ntaps;
        wide_stride = (int)w.stride;
        std::vector<float> wt(kMaxTaps, 0.f);
        for (int p = 0; p < w.ntaps; p++) wt[p] = (float)design::halfband_taps(w.design)[p];
        PG_HIP(hipMalloc((void **)&d_wide_taps, sizeof(float) * kMaxTaps));
        PG_HIP(hipMemcpy(d_wide_taps, wt.data(), sizeof(float) * kMaxTaps, hipMemcpyHostToDevice));
        kfirst = 2;
        fused_front = first.cic3 != 0 && w.ntaps == kFrontT1;  // k_mix_cic_hb is built for the hb11 the ladder always picks here
        memset(&wide_fir, 0, sizeof(wide_fir));
        wide_fir.stride = (int)w.stride;
        for (int p = 0; p < w.ntaps && p < kFrontT1; p++) wide_fir.h[p] = wt[p];
    }
    // the fused later stages
    memset(&casc, 0, sizeof(casc));
    casc.nst = (int)(ns - kfirst);
    casc.gain = last_gain;
    long long halo0 = 0, prod = 1, later = 1;
    for (size_t k = kfirst; k < ns; k++) {
        const design::Stage &st = chain.stages[k];
        casc.ntaps[k - kfirst] = st.ntaps;
        casc.stride[k - kfirst] = (int)st.stride;
        for (int p = 0; p < st.ntaps; p++) casc.h[k - kfirst][p] = (float)design::halfband_taps(st.design)[p];
        halo0 += (long long)(st.ntaps - 1) * prod;  // look-back of the whole cascade, in stage-0 output samples
        prod *= st.stride;
        later *= st.stride;
    }
    if (casc.nst > 0) {
        // largest power-of-two tile of final outputs whose two ping-pong buffers fit comfortably (<= 48 KiB)
        const long long lds_cap = 48LL * 1024;
        int outb = 256;
        for (;; outb /= 2) {
            long long c0 = outb, c1 = 0;
            for (int k = casc.nst; k >= 1; k--) {
                c1 = c0;
                c0 = c0 * casc.stride[k - 1] + casc.ntaps[k - 1] - 1;
            }
            // c0 = stage-0 samples a tile reads, c1 = outputs of the first fused stage (the largest intermediate)
            if ((c0 + c1) * (long long)sizeof(float2) <= lds_cap || outb == 1) {
                casc.outb = outb;
                casc.lds_half = (int)c0;
                casc_lds_bytes = (size_t)(c0 + c1) * sizeof(float2);
                break;
            }
        }
        if (casc_lds_bytes > 150 * 1024) return fail(PEBBLEGPU_E_UNSUPPORTED, "decimation cascade does not fit LDS");
    }
    // the whole decimator in one kernel for banks off one shared stream: k_mix_dec_mfma for every chain of the form [cic3 x S0,] hb11 x S,
    // three halfbands at stride 2 whose tap counts are in its table (bank_variant); k_mix_dec_fused<15, 19, 31>, its predecessor, for
    // hb11 x S, hb15, hb19, hb31 (PEBBLEGPU_BANK_DEC=0)
    {
        using FG = FusedDecGeom<15, 19, 31>;
        const char *env = getenv("PEBBLEGPU_NO_FUSED_DEC");
        const bool off = env && env[0] == '1';
        const bool three2 = casc.nst == 3 && casc.stride[0] == 2 && casc.stride[1] == 2 && casc.stride[2] == 2;
        const bool hb_front = bank_front && C >= 16 && !wide && first.stride <= 16;
        const bool cic_front = fused_front && C >= 16 && wide_stride == 16;
        fused_all = hb_front && three2 && casc.ntaps[0] == 15 && casc.ntaps[1] == 19 && casc.ntaps[2] == 31 && !off;
        const char *eb = getenv("PEBBLEGPU_BANK_DEC");
        bank_mfma = false;
        if (three2 && (hb_front || cic_front) && !off && !(eb && eb[0] == '0'))
            bank_mfma = bank_variant(cic_front ? 12 : 4, casc.ntaps[0], casc.ntaps[1], casc.ntaps[2], &fused_hy, &bank_nstate, &bank_minw);
        if (fused_all && !bank_mfma) fused_hy = FG::HY;
        if (fused_all || bank_mfma) {
            if (halo0 < fused_hy) halo0 = fused_hy;  // the first-stage buffer's head-room doubles as these kernels' first-stage history
            xh_depth = cic_front ? 512 : 16;         // raw samples in front of a call that its first windows reach (11 S0 + 1 with a CIC3 in front)
            if (fused_all) {
                fused_p = new FusedDecParams();
                memset(fused_p, 0, sizeof(*fused_p));
                fused_p->S = first.stride;
                fused_p->n_chan = (int)C;
                fused_p->hist_pitch = kMaxTaps;
                fused_p->gain0 = first.gain;
                fused_p->gain = casc.gain;
            }
            for (int i = 0; i < 2; i++) {
                PG_HIP(hipMalloc((void **)&d_xhist[i], sizeof(float2) * xh_depth));
                PG_HIP(hipMemset(d_xhist[i], 0, sizeof(float2) * xh_depth));
            }
            for (int i = 0; i < 2; i++) {  // (64 entries of slack: the kernel requests a block ahead of what it uses)
                PG_HIP(hipMalloc((void **)&d_y0stage2[i], sizeof(float2) * ((size_t)fused_hy * C + 64)));
                PG_HIP(hipMemset(d_y0stage2[i], 0, sizeof(float2) * ((size_t)fused_hy * C + 64)));
            }
            d_y0stage = d_y0stage2[0];
            const char *ew = getenv("PEBBLEGPU_BANK_WAVES");
            bank_waves = ew ? atoi(ew) : 0;  // 0: chosen per call
            if (bank_waves < 0 || bank_waves > 4) bank_waves = 0;
            if (bank_mfma) {
                for (int i = 0; i < 2; i++) {
                    PG_HIP(hipMalloc((void **)&d_bank_state[i], sizeof(float2) * (size_t)bank_nstate * C));
                    PG_HIP(hipMemset(d_bank_state[i], 0, sizeof(float2) * (size_t)bank_nstate * C));
                }
            }
            bank_state_valid = false;
        }
    }
    const long long len0 = max_in / s0.stride;
    if (ns == 1) {
        if (int rc = buf0.alloc((int)C, last_hist, len0)) return rc;
    } else {
        if (halo0 > 256 * 32) return fail(PEBBLEGPU_E_UNSUPPORTED, "cascade look-back %lld too deep", halo0);
        if (wide) {
            if (!fused_front) { if (int rc = buf0.alloc((int)C, wide_taps - 1, len0)) return rc; }
            if (int rc = buf1.alloc((int)C, (int)halo0, len0 / wide_stride)) return rc;
            if (int rc = fin.alloc((int)C, last_hist, len0 / wide_stride / later)) return rc;
        } else {
            if (int rc = buf0.alloc((int)C, (int)halo0, len0)) return rc;
            if (int rc = fin.alloc((int)C, last_hist, len0 / later)) return rc;
        }
    }
    for (int i = 0; i < 2; i++) {
        PG_HIP(hipMalloc((void **)&d_hist_mixed[i], sizeof(float2) * kMaxTaps * C));
        PG_HIP(hipMemset(d_hist_mixed[i], 0, sizeof(float2) * kMaxTaps * C));
        if (C == 1) {
            PG_HIP(hipMalloc((void **)&d_xtail_w[i], sizeof(float2) * 2048));
            PG_HIP(hipMemset(d_xtail_w[i], 0, sizeof(float2) * 2048));
        }

    }
    return 0;
}
void DecimCore::release()
{
    delete fused_p;
    fused_p = nullptr;
    for (int i = 0; i < 2; i++) {
        if (d_xhist[i]) (void)hipFree(d_xhist[i]);
        d_xhist[i] = nullptr;
    }
    for (int i = 0; i < 2; i++) {
        if (d_y0stage2[i]) (void)hipFree(d_y0stage2[i]);
        d_y0stage2[i] = nullptr;
    }
    d_y0stage = nullptr;
    for (int i = 0; i < 2; i++) {
        if (d_bank_state[i]) (void)hipFree(d_bank_state[i]);
        d_bank_state[i] = nullptr;
    }
    for (int i = 0; i < 2; i++) {
        if (d_dyn[i]) (void)hipFree(d_dyn[i]);
        d_dyn[i] = nullptr;
    }
    buf0.release();
    buf1.release();
    fin.release();
    fin2.release();
    fin3.release();
    if (d_wide_taps) (void)hipFree(d_wide_taps);
    d_wide_taps = nullptr;
    for (int i = 0; i < 2; i++) {
        if (d_hist_mixed[i]) (void)hipFree(d_hist_mixed[i]);
        d_hist_mixed[i] = nullptr;
        if (d_xtail_w[i]) (void)hipFree(d_xtail_w[i]);
        d_xtail_w[i] = nullptr;
    }
    if (d_c0tab) (void)hipFree(d_c0tab);
    d_c0tab = nullptr;
    if (d_ph_scratch) (void)hipFree(d_ph_scratch);
    d_ph_scratch = nullptr;
    ph_cap = 0;
}
int DecimCore::set_fuse_window(const float *d_window, const std::vector<float> &w)
{
    fuse_window = nullptr;
    if (C != 1 || w.size() != 2048 || !bank_front) return 0;
    static const int kD[7] = {0, 2, 4, 5, 6, 8, 10};
    std::vector<float> t(7 * 256);
    for (int i = 0; i < 7; i++)
        for (int jf = 0; jf < 256; jf++) t[256 * i + jf] = bank_taps.h[kD[i]] / w[(8 * jf - 10 + kD[i]) & 2047];
    h_r0 = t;
    h_c0.assign(t.size(), make_float2(0.f, 0.f));
    c0_valid = false;
    if (!d_c0tab) PG_HIP(hipMalloc((void **)&d_c0tab, sizeof(float2) * t.size()));
    fuse_window = d_window;
    return 0;
}
bool DecimCore::shape_for_spectrum() const
{
    return C == 1 && bank_front && !fused_all && !fused_front && !wide && first.stride == 8 && casc.nst == 3 && casc.stride[0] == 2 && casc.stride[1] == 2 &&
           casc.stride[2] == 2 && casc.ntaps[0] == 15 && casc.ntaps[1] == 23 && casc.ntaps[2] == 47 && buf0.hist <= 256 && d_xtail_w[0] != nullptr && d_c0tab != nullptr;
}
int DecimCore::fill_dec_fuse(hipStream_t s, DecFuse *df, const OscBank &osc, long long n)
{
    if (n <= 0 || n % 2048 != 0) return fail(PEBBLEGPU_E_SIZE, "%lld samples is not a whole number of 2048-sample frames", n);
    len0 = n / first.stride;
    len_out = n / (long long)chain.total;
    if (len0 > buf0.cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
    memset(df, 0, sizeof(*df));
    df->y = fin.data();
    df->y0_tail = buf0.data() + (len0 - 256);
    df->xtail = d_xtail_w[xtail_parity];
    df->xtail_next = d_xtail_w[xtail_parity ^ 1];
    const ChanOsc &o = osc.h_osc[0];
    df->phase0 = osc.inline_dyn.use ? osc.inline_dyn.d[0].phase0 : osc.ctl[0].phase0;
    df->inc = o.inc;
    df->a_inf = osc.a_inf;
    df->gain0 = first.gain;
    df->gain_last = casc.gain;
    df->mix_on = (int)o.mix_on;
    { static const int dbg = [] { const char *e = getenv("PEBBLEGPU_FUSE_DBG"); return e ? atoi(e) : 0; }(); df->dbg = dbg; }
    if (!c0_valid || c0_inc != o.inc || c0_mix != (int)o.mix_on) {
        // (rare: the first such call and the first after a retune.)  The copy is queued on the call's main stream, which follows
        // everything earlier calls queued on either stream, and waited for: the host table is free to change afterwards
        static const int kD[7] = {0, 2, 4, 5, 6, 8, 10};
        for (int i = 0; i < 7; i++)
            for (int jf = 0; jf < 256; jf++) {
                const float r = h_r0[256 * i + jf];
                const float2 st = o.mix_on ? o.step[kD[i]] : make_float2(1.f, 0.f);
                h_c0[256 * i + jf] = make_float2(r * st.x, r * st.y);
            }
        PG_HIP(hipMemcpyAsync(d_c0tab, h_c0.data(), sizeof(float2) * h_c0.size(), hipMemcpyHostToDevice, s));
        PG_HIP(hipStreamSynchronize(s));
        c0_valid = true;
        c0_inc = o.inc;
        c0_mix = (int)o.mix_on;
    }
    {
        double ph = 2048.0 * o.inc;
        ph -= std::floor(ph);
        df->wfr = make_float2((float)std::cos(design::kTwoPi * ph), (float)std::sin(design::kTwoPi * ph));
    }
    df->c0tab = d_c0tab;
    {
        // one 2 KiB row per frame chain of the transform's launch (chains of at most 32 frames, at least 512 of them)
        const size_t need = (size_t)(n / 2048 + 1024) * 256;
        if (need > ph_cap) {
            if (d_ph_scratch) (void)hipFree(d_ph_scratch);
            d_ph_scratch = nullptr;
            ph_cap = 0;
            PG_HIP(hipMalloc((void **)&d_ph_scratch, sizeof(float2) * need));
            ph_cap = need;
        }
        df->ph_scratch = d_ph_scratch;
    }
    return 0;
}
// the call's last 2048 samples, windowed: the look-back of a later call whose decimator runs inside the display transform
static __global__ __launch_bounds__(256) void k_window_tail(const float2 *__restrict__ in, long long n, const float *__restrict__ window, float2 *__restrict__ out,
                                                            RawSrc raw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const long long src = n - 2048 + i;
    const float2 v = raw.base ? raw_load(raw, src) : in[src];
    out[i] = cscale(v, window[i]);
}
int DecimCore::run_beside_spectrum(hipStream_t s, const float2 *d_in, long long in_pitch, bool shared_input, long long n, const OscBank &osc, const RawSrc *raw)
{
    // (fill_dec_fuse has set the lengths.)  The general kernels keep the mixed samples of the call's end as their history: left by the
    // same two-workgroup launch that serves k_mix_hb11_lean, so the next call may take either route
    const int R = 8;
    const long long j_first = (10 + first.stride - 1) / first.stride;
    const RawSrc rs = raw ? *raw : RawSrc{nullptr, 0, 0, 0.f, 0};
    launch(raw ? k_mix_hb11_bank<false, false, true> : k_mix_hb11_bank<false, false, false>, dim3(len0 / (4LL * R * 64) != 0 ? 2 : 1, 1), dim3(256), s, d_in, in_pitch,
           (int)shared_input, buf0.data(), buf0.pitch, len0, (const ChanOsc *)osc.d_osc, (const float2 *)d_hist_mixed[hist_parity], d_hist_mixed[hist_parity ^ 1],
           (int)kMaxTaps, (const float *)osc.d_amp, osc.a_inf, bank_taps, first.gain, osc.inline_dyn, 0, (int)C, R, j_first, rs);
    PG_HIP(hipGetLastError());
    hist_parity ^= 1;
    xtail_parity ^= 1;  // the transform's last workgroup writes the next call's look-back
    last_fused = false;
    front_name = "k_spectrum_t128 (decimator inside)";
    rest_name = "";
    return 0;
}
int DecimCore::run(hipStream_t s, const float2 *d_in, long long in_pitch, bool shared_input, long long n, const OscBank &osc,
                   hipEvent_t after_first, const RawSrc *raw, const OscAdvance *oa)
{
    if (raw && !raw_ready(osc)) return fail(PEBBLEGPU_E_INVALID, "raw-format input reached a decimator path that has no converting loads");
    if (n <= 0 || n % (long long)chain.total != 0)
        return fail(PEBBLEGPU_E_SIZE, "%lld samples is not a multiple of the decimation %u", n, chain.total);
    len0 = n / first.stride;
    // successive calls write successive output buffers (tail_job_out carries the consumer's look-back into the next one's head-room)
    if (fin3.base && rotate3) { std::swap(fin, fin2); std::swap(fin2, fin3); }  // (fin, fin2, fin3) <- (fin2, fin3, fin)
    else if (fin2.base) std::swap(fin, fin2);  // (either way the call writes what was fin2, whose head-room the last call's consumer filled)
    const HistBuf *src = &buf0;
    last_fused = false;
    last_mfma = false;
    osc_advanced = false;
    done_recorded = false;
    const bool had_bank_state = bank_state_valid;
    bank_state_valid = false;  // (set again below when this call takes the matrix-pipe route)
    const bool had_dyn = dyn_valid;
    dyn_valid = false;
    if ((fused_all || bank_mfma) && shared_input && !osc.any_transient()) {
        const HistBuf &y0b = fused_front ? buf1 : buf0;  // the first-stage (hb11) outputs' buffer of the other route: its head-room is the history
        if (n / ((long long)first.stride * (fused_front ? wide_stride : 1)) > y0b.cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
        len_out = n / (long long)chain.total;
        const long long groups = cdiv(C, 64);
        if (!fused_L) {
            const char *e = getenv("PEBBLEGPU_FUSED_L");
            fused_L = e ? atoi(e) : 0;
            if (fused_L < 16) fused_L = -1;  // choose per call
            fused_L &= ~15;
        }
        if (bank_mfma && (unsigned long long)n * 8 < 0xFFF00000ull && (unsigned long long)C * (unsigned long long)fin.pitch * 8 < 0xFFF00000ull &&
            len_out >= 64 && (reinterpret_cast<uintptr_t>(d_in) & 7) == 0) {  // (its sample fetches and result stores carry 32-bit byte offsets)
            dyn_valid = had_dyn;
            if (int rc = run_bank_mfma(s, d_in, n, osc, had_bank_state, oa)) return rc;
            if (after_first) PG_HIP(hipEventRecord(after_first, s));
            return 0;
        }
      if (int rc = flush_y0(s)) return rc;
      if (fused_all) {
        long long L = fused_L > 0 ? fused_L : ((len_out * groups / 704 + 15) & ~15LL);  // ~700 four-wave workgroups (measured best on 256 CUs: 96 for configs[2]); every chunk pays a 21-block warm-up
        if (L < 32) L = 32;
        fused_p->n_out = len_out;
        fused_p->out_pitch = fin.pitch;
        fused_p->y0_pitch = buf0.pitch;
        fused_p->L = (int)L;
        fused_p->a_inf = osc.a_inf;
        launch(k_mix_dec_fused<15, 19, 31>, dim3(cdiv(len_out, L), (unsigned)groups), dim3(256), s, d_in, fin.data(), (const ChanOsc *)osc.d_osc, osc.inline_dyn,
               (const float2 *)d_xhist[hist_parity], d_xhist[hist_parity ^ 1], (const float2 *)buf0.data(), d_y0stage, d_hist_mixed[hist_parity ^ 1], *fused_p);
        PG_HIP(hipGetLastError());
        hist_parity ^= 1;
        front_name = "k_mix_dec_fused";
        rest_name = "";
        last_fused = true;
        if (after_first) PG_HIP(hipEventRecord(after_first, s));
        return 0;
      }
    }
    if (int rc = flush_y0(s)) return rc;
    if (fused_front) {
        len1 = len0 / wide_stride;
        if (len1 > buf1.cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
        int cl_log2 = 0;
        while ((1u << cl_log2) < C && cl_log2 < 6) cl_log2++;  // channels across the lanes of a wave; the other lanes take further outputs
        const int R = cl_log2 == 6 ? 16 : 8;                      // outputs per lane: a channel's R*OL outputs leave the wave as one row segment
        const dim3 grid(cdiv(len1 + 1, 4LL * R * (64 >> cl_log2)), cdiv(C, 1u << cl_log2));
        const bool uni = cl_log2 == 6 && shared_input;  // the lanes of a wave share one window: scalar loads
        auto kern = osc.any_transient() ? (uni ? k_mix_cic_hb<true, true> : k_mix_cic_hb<true, false>) : (uni ? k_mix_cic_hb<false, true> : k_mix_cic_hb<false, false>);
        launch_lds(kern, grid, dim3(256), cl_log2 ? 4 * (size_t)front_tile_slots(cl_log2, R) * sizeof(float2) : 0, s, d_in, in_pitch, (int)shared_input, buf1.data(), buf1.pitch,
                   len1, (const ChanOsc *)osc.d_osc, (const float2 *)d_hist_mixed[hist_parity], d_hist_mixed[hist_parity ^ 1], (int)kMaxTaps,
                   (const float *)osc.d_amp, osc.a_inf, wide_fir, first.stride, 1.0f, osc.inline_dyn, cl_log2, (int)C, R);
        hist_parity ^= 1;
        front_name = "k_mix_cic_hb";
        if (after_first) PG_HIP(hipEventRecord(after_first, s));
        len_out = len1;
        src = &buf1;
    } else {
        if (len0 > buf0.cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
        if (bank_front && C == 1 && want_lds_free && !osc.any_transient()) {
            // beside the spectrum kernel: the lean one-channel kernel for every output inside the call, and the general one
            // (a handful of lanes) for the first outputs, whose windows reach back into the mixed history, and the new history
            const int R = 8;
            const long long j_first = (10 + first.stride - 1) / first.stride;
            const RawSrc rs = raw ? *raw : RawSrc{nullptr, 0, 0, 0.f, 0};
            auto lean = !raw ? k_mix_hb11_lean<-1> : rs.fmt == 0 ? k_mix_hb11_lean<0> : rs.fmt == 1 ? k_mix_hb11_lean<1> : rs.fmt == 2 ? k_mix_hb11_lean<2>
                             : rs.fmt == 3 ? k_mix_hb11_lean<3> : k_mix_hb11_lean<4>;
            static const bool edge_launch = [] { const char *e = getenv("PEBBLEGPU_LEAN_EDGE_LAUNCH"); return e && e[0] == '1'; }();  // A/B: the edges as a launch of their own
            launch(lean, dim3(cdiv(len0, 4LL * R * 64) + (edge_launch ? 0 : 1)), dim3(256), s, d_in, buf0.data(), len0, (const ChanOsc *)osc.d_osc,
                   osc.a_inf, bank_taps, first.gain, osc.inline_dyn, R, edge_launch ? -j_first : j_first, rs, (const float2 *)d_hist_mixed[hist_parity],
                   d_hist_mixed[hist_parity ^ 1]);
            front_name = "k_mix_hb11_lean";
            if (edge_launch)
                launch(raw ? k_mix_hb11_bank<false, false, true> : k_mix_hb11_bank<false, false, false>, dim3(len0 / (4LL * R * 64) != 0 ? 2 : 1, 1), dim3(256), s, d_in, in_pitch,
                       (int)shared_input, buf0.data(), buf0.pitch, len0, (const ChanOsc *)osc.d_osc, (const float2 *)d_hist_mixed[hist_parity], d_hist_mixed[hist_parity ^ 1],
                       (int)kMaxTaps, (const float *)osc.d_amp, osc.a_inf, bank_taps, first.gain, osc.inline_dyn, 0, (int)C, R, j_first, rs);
        } else if (bank_front && (C >= 16 ? shared_input : want_lds_free)) {
 // a bank off one shared stream: lanes = channels, windows in registers (k_mix_hb11_bank)
            int cl_log2 = 0;
            while ((1u << cl_log2) < C && cl_log2 < 6) cl_log2++;
            const int R = cl_log2 == 6 ? 16 : 8;
            const dim3 grid(cdiv(len0 + 1, 4LL * R * (64 >> cl_log2)), cdiv(C, 1u << cl_log2));
            const bool uni = cl_log2 == 6;
            auto kern = osc.any_transient() ? (uni ? k_mix_hb11_bank<true, true> : k_mix_hb11_bank<true, false>)
                                            : (uni ? k_mix_hb11_bank<false, true> : k_mix_hb11_bank<false, false>);
            launch_lds(kern, grid, dim3(256), cl_log2 ? 4 * (size_t)front_tile_slots(cl_log2, R) * sizeof(float2) : 0, s, d_in, in_pitch, (int)shared_input, buf0.data(),
                       buf0.pitch, len0, (const ChanOsc *)osc.d_osc, (const float2 *)d_hist_mixed[hist_parity], d_hist_mixed[hist_parity ^ 1], (int)kMaxTaps,
                       (const float *)osc.d_amp, osc.a_inf, bank_taps, first.gain, osc.inline_dyn, cl_log2, (int)C, R, -1LL, RawSrc{nullptr, 0, 0, 0.f, 0});
            front_name = "k_mix_hb11_bank";
        } else {
        // merged CIC3 over a shared stream: one workgroup mixes a group of channels from one fetch of the sample pairs
        const int cg = (first.cic3 && first.stride > 2 && shared_input) ? 8 : 1;
        size_t lds = mixdec_lds_bytes(first);
        if (cg > 1 && lds < (size_t)cg * 258 * sizeof(float4)) lds = (size_t)cg * 258 * sizeof(float4);
        launch_lds(k_mix_dec1, dim3(cdiv(len0, 256), cdiv(C, cg)), dim3(256), lds, s, d_in, in_pitch, (int)shared_input, buf0.data(),
                   buf0.pitch, len0, (const ChanOsc *)osc.d_osc, (const float2 *)d_hist_mixed[hist_parity], (int)kMaxTaps, (const float *)osc.d_amp,
                   osc.a_inf, first, d_hist_mixed[hist_parity ^ 1], osc.inline_dyn, cg, (int)C);  // its last block leaves the next call's mixed history
        front_name = "k_mix_dec1";
        }
        hist_parity ^= 1;
        if (after_first) PG_HIP(hipEventRecord(after_first, s));
        len_out = len0;
        if (wide) {  // the peeled stride->=8 stage: every output reads its own taps-long window straight from buf0
            len1 = len0 / wide_stride;
            launch(k_fir_dec, dim3(cdiv(len1, 256), C), dim3(256), s, (const float2 *)buf0.data(), buf0.pitch, buf1.data(), buf1.pitch, len1, wide_stride,
                   (const float *)d_wide_taps, (const float *)nullptr, 0, (const int *)nullptr, wide_taps, 1.0f, 0, (const int *)nullptr, Gate{nullptr, 0, 0});
            src = &buf1;
        }
    }
    rest_name = wide && !fused_front ? "k_fir_dec" : "";
    if (casc.nst > 0) {
        rest_name = wide && !fused_front ? "k_fir_dec + k_cascade" : "k_cascade";
        len_out = n / (long long)chain.total;
        const bool three2 = casc.nst == 3 && casc.stride[0] == 2 && casc.stride[1] == 2 && casc.stride[2] == 2;
        auto kern = k_cascade<0, 0, 0>;
        if (three2 && casc.ntaps[0] == 15 && casc.ntaps[1] == 23 && casc.ntaps[2] == 47) kern = k_cascade<15, 23, 47>;  // WFM from 20 Msps, narrow from 100 Msps
        else if (three2 && casc.ntaps[0] == 15 && casc.ntaps[1] == 19 && casc.ntaps[2] == 31) kern = k_cascade<15, 19, 31>;  // narrow at 2.048 Msps
        launch_lds(kern, dim3(cdiv(len_out, casc.outb), C), dim3(256), casc_lds_bytes, s, (const float2 *)src->data(), src->pitch, fin.data(),
                   fin.pitch, len_out, casc);
    }
    PG_HIP(hipGetLastError());
    if ((fused_all || bank_mfma) && shared_input && n >= xh_depth)  // the next call may take a one-kernel route: it wants this call's raw tail (hist_parity already flipped)
        PG_HIP(hipMemcpyAsync(d_xhist[hist_parity], d_in + (n - xh_depth), sizeof(float2) * xh_depth, hipMemcpyDeviceToDevice, s));
    if (fuse_window && n >= 2048 && shape_for_spectrum()) {  // the next call's decimator may run inside the display transform
        launch(k_window_tail, dim3(8), dim3(256), s, d_in, n, fuse_window, d_xtail_w[xtail_parity ^ 1], raw ? *raw : RawSrc{nullptr, 0, 0, 0.f, 0});
        xtail_parity ^= 1;
    }
    return 0;
}
void DecimCore::tail_jobs_dec(std::vector<TailJob> &jobs) const
{
    if (last_fused) {
        // the call's last first-stage outputs (staged by the kernel) become the stage-0 head-room, as if the buffer had been written
        // (behind k_mix_dec_mfma only when a later call asks for them there: flush_y0)
        const HistBuf &y0b = fused_front ? buf1 : buf0;
        if (!last_mfma) jobs.push_back(TailJob{d_y0stage, (long long)fused_hy, (long long)fused_hy, fused_hy, 0, y0b.base + (y0b.hist - fused_hy), y0b.pitch});
        return;
    }
    if (!fused_front && buf0.hist > 0 && casc.nst > 0) jobs.push_back(TailJob{buf0.data(), buf0.pitch, len0, buf0.hist, 0, nullptr, 0});
    if (wide && buf1.hist > 0) jobs.push_back(TailJob{buf1.data(), buf1.pitch, len1, buf1.hist, 0, nullptr, 0});
}
void DecimCore::tail_job_out(std::vector<TailJob> &jobs) const
{
    if (casc.nst == 0) {  // a single stage: its buffer is the output
        if (!fused_front && buf0.hist > 0) jobs.push_back(TailJob{buf0.data(), buf0.pitch, len0, buf0.hist, 0, nullptr, 0});
        return;
    }
    if (fin.hist <= 0) return;
    // (two output buffers: the next call writes the other one, whose head-room row starts at its base)
    if (fin2.base) jobs.push_back(TailJob{fin.data(), fin.pitch, len_out, fin.hist, 0, fin2.base, fin2.pitch});
    else jobs.push_back(TailJob{fin.data(), fin.pitch, len_out, fin.hist, 0, nullptr, 0});
}
void DecimCore::tail_jobs(std::vector<TailJob> &jobs) const
{
    tail_jobs_dec(jobs);
    tail_job_out(jobs);
}
int DecimCore::enable_double_out()
{
    if (casc.nst == 0 || !fin.base) return fail(PEBBLEGPU_E_UNSUPPORTED, "a single-stage chain has no separate output buffer to double");
    if (fin2.base) return 0;
    if (int rc = fin2.alloc(fin.chans, fin.hist, fin.cap)) return rc;
    const char *e = getenv("PEBBLEGPU_BANK_PIPE_BUFS");  // 2: the decimator waits for the consumer of the call before the last
    if (e && e[0] == '2') return 0;
    return fin3.alloc(fin.chans, fin.hist, fin.cap);
}

int fill_tail_jobs(TailJobs &tj, const std::vector<TailJob> &jobs, const OscAdvance *oa)
{
    if (jobs.size() > (size_t)kMaxTailJobs) return fail(PEBBLEGPU_E_INVALID, "too many history buffers");
    memset(&tj, 0, sizeof(tj));
    tj.count = (int)jobs.size();
    if (oa != nullptr && oa->osc != nullptr) tj.oa = *oa;
    int maxh = 1;
    for (size_t i = 0; i < jobs.size(); i++) {
        tj.job[i] = jobs[i];
        if (jobs[i].hist > maxh) maxh = jobs[i].hist;
    }
    if (maxh > 256 * 32) return fail(PEBBLEGPU_E_UNSUPPORTED, "history of %d samples too deep for the tail refresh", maxh);
    return 0;
}
// one strided real-tap FIR over `channels` rows (k_fir_dec with one tap set for all rows): y[o] = sum_p in[o stride + p - (ntaps - 1)] taps[p]
int run_fir_dec(hipStream_t s, const float2 *in, long long in_pitch, float2 *out, long long out_pitch, long long n_out, int stride,
                const float *d_taps, int ntaps, uint32_t channels)
{
    if (n_out <= 0) return 0;
    if (ntaps < 1 || ntaps > kMaxTaps) return fail(PEBBLEGPU_E_UNSUPPORTED, "%d taps: k_fir_dec holds at most %d", ntaps, kMaxTaps);
    launch(k_fir_dec, dim3(cdiv(n_out, 256), channels), dim3(256), s, in, in_pitch, out, out_pitch, n_out, stride, d_taps, (const float *)nullptr, 0,
           (const int *)nullptr, ntaps, 1.0f, 0, (const int *)nullptr, Gate{nullptr, 0, 0});
    PG_HIP(hipGetLastError());
    return 0;
}
int run_nap(hipStream_t s, unsigned ticks_100mhz)
{
    if (ticks_100mhz == 0) return 0;
    launch(k_nap, dim3(1), dim3(64), s, ticks_100mhz);
    PG_HIP(hipGetLastError());
    return 0;
}
int run_save_tails(hipStream_t s, const std::vector<TailJob> &jobs, uint32_t channels, const OscAdvance *oa)
{
    const bool adv = oa != nullptr && oa->osc != nullptr;
    if (jobs.empty() && !adv) return 0;
    TailJobs tj;
    if (int rc = fill_tail_jobs(tj, jobs, oa)) return rc;
    launch(k_save_tails, dim3(1, channels, (unsigned)(jobs.empty() ? 1 : jobs.size())), dim3(256), s, tj);
    PG_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// FastFirCore
// ------------------------------------------------------------------------------------------------
int FastFirCore::init(uint32_t channels, uint32_t fft_size, uint32_t fir_size)
{
    C = channels;
    fft_n = fft_size;
    taps = fir_size;
    if (!(fft_n == 2048 || fft_n == 4096 || fft_n == 8192)) return fail(PEBBLEGPU_E_UNSUPPORTED, "FastFIR FFT size %u not built", fft_n);
    if (taps < 2 || taps > fft_n) return fail(PEBBLEGPU_E_INVALID, "FastFIR taps must be in [2, fft size]");
    PG_HIP(hipMalloc((void **)&d_H, sizeof(float2) * (size_t)fft_n * C));
    PG_HIP(hipMemset(d_H, 0, sizeof(float2) * (size_t)fft_n * C));
    if (fft_n == 2048) { if (int rc = make_twiddles_t128(&d_tw128)) return rc; }
    return make_twiddles((int)fft_n, &d_tw);
}
void FastFirCore::release()
{
    if (d_H) (void)hipFree(d_H);
    if (d_tw) (void)hipFree(d_tw);
    if (d_tw128) (void)hipFree(d_tw128);
    d_H = d_tw = d_tw128 = nullptr;
}
int FastFirCore::design(hipStream_t s, uint32_t ch, double lo, double hi, double offset, double rate, bool *ok)
{
    std::vector<std::complex<double>> H;
    *ok = design::fastfir_design(fft_n, taps, lo, hi, offset, rate, H);
    if (!*ok) return 0;
    std::vector<float2> hf(fft_n);
    for (uint32_t i = 0; i < fft_n; i++) hf[i] = make_float2((float)H[i].real(), (float)H[i].imag());
    PG_HIP(hipMemcpyAsync(d_H + (size_t)ch * fft_n, hf.data(), sizeof(float2) * fft_n, hipMemcpyHostToDevice, s));
    PG_HIP(hipStreamSynchronize(s));
    return 0;
}
// k_fastfir_t128's grid: one-dimensional with the XCD-aware order (kernels_fastfir.h) unless PEBBLEGPU_FF_XCD=0 asks for (block, channel)
struct FfGrid { dim3 grid; int nb, nchan; };
static FfGrid ff_grid(long long nb, uint32_t channels)
{
    static const bool xcd = [] { const char *e = getenv("PEBBLEGPU_FF_XCD"); return !(e && e[0] == '0'); }();
    const long long total = nb * (long long)channels;
    if (!xcd || total >= (1LL << 31) - 8) return FfGrid{dim3((unsigned)nb, channels), (int)nb, 0};
    return FfGrid{dim3((unsigned)(8 * ((total + 7) / 8))), (int)nb, (int)channels};
}
int FastFirCore::run(hipStream_t s, const HistBuf &in, long long n, float2 *out, long long out_pitch)
{
    const int overlap = (int)taps - 1;
    const long long L = block_len();
    if (n % L != 0) return fail(PEBBLEGPU_E_SIZE, "FastFIR input %lld is not a multiple of its block %lld", n, L);
    const dim3 grid((unsigned)(n / L), C), block(256);
    const float2 *no_tail = nullptr;
    // (twiddles from the workgroup's LDS copy here: beside a bank's decimator -- two-stage calls -- the variant that reads them through the
    // vector cache measured 0.0833 ms per configs[2] call against 0.0820; alone, in the stream bank, it is the faster one: run_ext)
    static const bool tw_lds = [] { const char *e = getenv("PEBBLEGPU_FF_TWLDS"); return !(e && e[0] == '0'); }();
    const FfGrid fg = ff_grid(n / L, C);
    if (fft_n == 2048 && !tw_lds) launch(k_fastfir_t128<false>, fg.grid, dim3(128), s, (const float2 *)in.data(), in.pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw128, overlap, no_tail, (float2 *)nullptr, fg.nb, fg.nchan);
    else if (fft_n == 2048) launch(k_fastfir_t128<true>, fg.grid, dim3(128), s, (const float2 *)in.data(), in.pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw128, overlap, no_tail, (float2 *)nullptr, fg.nb, fg.nchan);
    else if (fft_n == 4096) launch(k_fastfir<4096>, grid, block, s, (const float2 *)in.data(), in.pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw, overlap, no_tail);
    else launch(k_fastfir<8192>, grid, block, s, (const float2 *)in.data(), in.pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw, overlap, no_tail);
    PG_HIP(hipGetLastError());
    return 0;
}
int FastFirCore::run_ext(hipStream_t s, const float2 *in, long long in_pitch, float2 *d_tail, long long n, float2 *out, long long out_pitch, float2 *d_tail_next)
{
    const int overlap = (int)taps - 1;
    const long long L = block_len();
    if (n % L != 0) return fail(PEBBLEGPU_E_SIZE, "FastFIR input %lld is not a multiple of its block %lld", n, L);
    if (n < overlap) return fail(PEBBLEGPU_E_SIZE, "FastFIR input %lld is shorter than its overlap %d", n, overlap);
    if (n == 0) return 0;
    const dim3 grid((unsigned)(n / L), C), block(256);
    const float2 *tail = d_tail;
    if (fft_n == 2048) {
        static const size_t pad = [] { const char *e = getenv("PEBBLEGPU_FF_PADLDS"); return e ? (size_t)atol(e) : (size_t)0; }();  // A/B: extra LDS per workgroup (lowers its occupancy)
        // twiddles through the vector cache: without the 4.5 KB copy per workgroup eight workgroups fit a CU's LDS instead of six -- configs[4]'s
        // band-pass 0.222 / 0.227 -> 0.215 / 0.211 ms in alternating runs (PEBBLEGPU_FF_TWLDS=1 brings the copy back)
        static const bool twg = [] { const char *e = getenv("PEBBLEGPU_FF_TWLDS"); return !(e && e[0] == '1'); }();
        const FfGrid fg = ff_grid(n / L, C);
        if (twg) launch_lds(k_fastfir_t128<false>, fg.grid, dim3(128), pad, s, in, in_pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw128, overlap, tail, d_tail_next, fg.nb, fg.nchan);
        else launch_lds(k_fastfir_t128<true>, fg.grid, dim3(128), pad, s, in, in_pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw128, overlap, tail, d_tail_next, fg.nb, fg.nchan);
        if (d_tail_next) {  // the kernel's last block has written the next call's overlap into the caller's other buffer
            PG_HIP(hipGetLastError());
            return 0;
        }
    }
    else if (fft_n == 4096) launch(k_fastfir<4096>, grid, block, s, in, in_pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw, overlap, tail);
    else launch(k_fastfir<8192>, grid, block, s, in, in_pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw, overlap, tail);
    PG_HIP(hipGetLastError());
    // m_pFFTOverlapBuf <- last taps-1 input samples of every row (fastfir.cpp:312-316); ordered after the kernel on s
    PG_HIP(hipMemcpy2DAsync(d_tail, sizeof(float2) * (size_t)overlap, in + (n - overlap), sizeof(float2) * (size_t)in_pitch,
                            sizeof(float2) * (size_t)overlap, C, hipMemcpyDeviceToDevice, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// AmCore
// ------------------------------------------------------------------------------------------------
int AmCore::init(uint32_t channels, double demod_rate, long long max_n)
{
    C = channels;
    rate = demod_rate;
    if (int rc = tmp.alloc((int)C, kMaxTaps, max_n)) return rc;
    PG_HIP(hipMalloc((void **)&d_taps, sizeof(float) * kMaxTaps * C));
    PG_HIP(hipMemset(d_taps, 0, sizeof(float) * kMaxTaps * C));
    PG_HIP(hipMalloc((void **)&d_ntaps, sizeof(int) * C));
    PG_HIP(hipMemset(d_ntaps, 0, sizeof(int) * C));
    PG_HIP(hipMalloc((void **)&d_list, sizeof(int) * C));
    PG_HIP(hipMalloc((void **)&d_state, sizeof(double) * 4 * C));
    PG_HIP(hipMemset(d_state, 0, sizeof(double) * 4 * C));
    const double alpha = (double)0.9999f;  // DC_ALPHA is a float literal, demod_am.cpp:36
    fill_scan_section(scan.sec[0], kOnePoleDiff, &alpha);
    return 0;
}
void AmCore::release()
{
    tmp.release();
    void *p[] = {d_taps, d_ntaps, d_list, d_state};
    for (void *q : p) if (q) (void)hipFree(q);
    d_taps = nullptr; d_ntaps = nullptr; d_list = nullptr; d_state = nullptr;
}
int AmCore::set_bandwidth(hipStream_t s, uint32_t ch, double bw)
{
    // InitLPFilter(0, 1.0, 50.0, bw, bw*1.8, rate); it also clears the FIR delay line (fir.cpp:297-303)
    const std::vector<double> h = design::fir_lowpass(0, 1.0, 50.0, bw, bw * 1.8, rate);
    std::vector<float> hf(kMaxTaps, 0.f);
    for (size_t i = 0; i < h.size(); i++) hf[i] = (float)h[i];
    const int nt = (int)h.size();
    PG_HIP(hipMemcpyAsync(d_taps + (size_t)ch * kMaxTaps, hf.data(), sizeof(float) * kMaxTaps, hipMemcpyHostToDevice, s));
    PG_HIP(hipMemcpyAsync(d_ntaps + ch, &nt, sizeof(int), hipMemcpyHostToDevice, s));
    PG_HIP(hipMemsetAsync(tmp.base + (size_t)ch * tmp.pitch, 0, sizeof(float2) * tmp.hist, s));
    PG_HIP(hipStreamSynchronize(s));
    return 0;
}
int AmCore::set_list(hipStream_t s, const std::vector<int> &am_channels)
{
    list = am_channels;
    if (!list.empty()) {
        PG_HIP(hipMemcpyAsync(d_list, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice, s));
        PG_HIP(hipStreamSynchronize(s));
    }
    return 0;
}
int AmCore::run(hipStream_t s, const float2 *in, long long in_pitch, float2 *out, long long out_pitch, long long n, Gate gate)
{
    last_n = 0;
    if (list.empty()) return 0;
    if (n > tmp.cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
    if (n < tmp.hist) return fail(PEBBLEGPU_E_SIZE, "AM demod needs at least %d samples per call", tmp.hist);
    const unsigned na = (unsigned)list.size();
    const long long nsub = (n + kSub - 1) / kSub;
    // One workgroup per channel walks the call sequentially (the 0.9999 pole forbids a warm-up), so it may read
    // and write the same state slot; a channel that leaves AM keeps its stale state like the idle Demod_AM object.
    launch(k_iir_scan<2, 1>, dim3(1, na), dim3(64), s, in, in_pitch, tmp.data(), tmp.pitch, n, scan, (const double *)d_state, d_state,
           (int)nsub, -1, (const int *)d_list, gate);
    launch(k_fir_dec, dim3(cdiv(n, 256), na), dim3(256), s, (const float2 *)tmp.data(), tmp.pitch, out, out_pitch, n, 1,
           (const float *)d_taps, (const float *)nullptr, (int)kMaxTaps, (const int *)d_ntaps, 0, 1.0f, 0, (const int *)d_list, gate);
    last_n = n;
    if (!defer_tail) launch(k_save_tail, dim3(cdiv(tmp.hist, 256), na), dim3(256), s, tmp.data(), tmp.pitch, n, tmp.hist, (const int *)d_list, gate);
    PG_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// PllCore: Demod_NFM (mode 0) or Demod_SAM (mode 1) for the listed channels
// ------------------------------------------------------------------------------------------------
int PllCore::init(uint32_t channels, double demod_rate, long long max_n, int which)
{
    C = channels;
    rate = demod_rate;
    mode = which;
    memset(&pp, 0, sizeof(pp));
    pp.mode = mode;
    std::vector<double> h;
    std::vector<float> hi(kMaxTaps, 0.f), hq(kMaxTaps, 0.f);
    if (mode == 0) {  // Demod_NFM::init, demod_nfm.cpp:44-66 (float members, demod_nfm.h:27-40)
        const double norm = design::kTwoPi / rate;
        pp.lo = (float)(-15000.0 * norm);  // FMPLL_RANGE
        pp.hi = (float)(15000.0 * norm);
        pp.alpha = (float)(2.0 * .707 * 3000.0 * norm);  // FMPLL_ZETA, FMPLL_BW
        pp.beta = (float)((double)(pp.alpha * pp.alpha) / (4.0 * .707 * .707));
        pp.out_gain = 1.0f;
        pp.dc_alpha = (float)(1.0 - std::exp(-1.0 / (rate * 0.001)));  // FMDC_ALPHA
        h = design::fir_lowpass(0, 1.0, 50.0, 3000.0, 1.6 * 3000.0, rate);
        for (size_t i = 0; i < h.size(); i++) hi[i] = hq[i] = (float)h[i];
    } else {          // Demod_SAM ctor, demod_sam.cpp:5-32
        const float zeta = 0.707f;
        pp.alpha = (float)(2.0 * zeta * 100 * design::kTwoPi / rate);
        pp.beta = (float)((double)(pp.alpha * pp.alpha) / (4.0 * zeta * zeta));
        pp.lo = (float)(-1000 * design::kTwoPi / rate);
        pp.hi = (float)(1000 * design::kTwoPi / rate);
        h = design::fir_lowpass(0, 1.0, 40.0, 4500, 5500, rate);
        const int nt = (int)h.size();
        for (int n = 0; n < nt; n++) {  // GenerateHBFilter(5000.0), fir.cpp:454-468
            const double a = (design::kTwoPi * 5000.0 / rate) * ((double)n - ((double)(nt - 1) / 2.0));
            hi[n] = (float)(2.0 * h[n] * std::cos(a));
            hq[n] = (float)(2.0 * h[n] * std::sin(a));
        }
    }
    ntaps = (int)h.size();
    // k_fir_dec computes sum_p x[i-(T-1)+p] h[p]; CFir computes sum_k coef[k] x[i-k]: store the taps reversed
    std::vector<float> ri(kMaxTaps, 0.f), rq(kMaxTaps, 0.f);
    for (int p = 0; p < ntaps; p++) { ri[p] = hi[ntaps - 1 - p]; rq[p] = hq[ntaps - 1 - p]; }
    PG_HIP(hipMalloc((void **)&d_taps_i, sizeof(float) * kMaxTaps));
    PG_HIP(hipMalloc((void **)&d_taps_q, sizeof(float) * kMaxTaps));
    PG_HIP(hipMemcpy(d_taps_i, ri.data(), sizeof(float) * kMaxTaps, hipMemcpyHostToDevice));
    PG_HIP(hipMemcpy(d_taps_q, rq.data(), sizeof(float) * kMaxTaps, hipMemcpyHostToDevice));
    if (int rc = tmp.alloc((int)C, kMaxTaps, max_n)) return rc;
    PG_HIP(hipMalloc((void **)&d_state, sizeof(PllState) * C));
    PG_HIP(hipMemset(d_state, 0, sizeof(PllState) * C));
    PG_HIP(hipMalloc((void **)&d_list, sizeof(int) * C));
    return 0;
}
void PllCore::release()
{
    tmp.release();
    void *p[] = {d_taps_i, d_taps_q, d_state, d_list};
    for (void *q : p) if (q) (void)hipFree(q);
    d_taps_i = d_taps_q = nullptr;
    d_state = nullptr;
    d_list = nullptr;
}
int PllCore::set_list(hipStream_t s, const std::vector<int> &channels)
{
    list = channels;
    if (!list.empty()) {
        PG_HIP(hipMemcpyAsync(d_list, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice, s));
        PG_HIP(hipStreamSynchronize(s));
    }
    return 0;
}
int PllCore::run(hipStream_t s, const float2 *in, long long in_pitch, float2 *out, long long out_pitch, long long n, Gate gate)
{
    if (list.empty()) return 0;
    if (n > tmp.cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
    const unsigned nl = (unsigned)list.size();
    launch(k_pll_demod, dim3(cdiv(nl, 64)), dim3(64), s, in, in_pitch, tmp.data(), tmp.pitch, n, pp, d_state, (const int *)d_list,
           (int)nl, gate);
    launch(k_fir_dec, dim3(cdiv(n, 256), nl), dim3(256), s, (const float2 *)tmp.data(), tmp.pitch, out, out_pitch, n, 1, (const float *)d_taps_i,
           (const float *)d_taps_q, 0, (const int *)nullptr, ntaps, 1.0f, mode == 1 ? 1 : 0, (const int *)d_list, gate);
    // the listed channels' FIR history (a gated channel keeps its own)
    launch(k_save_tail, dim3(cdiv(tmp.hist, 256), nl), dim3(256), s, tmp.data(), tmp.pitch, n, tmp.hist, (const int *)d_list, gate);
    PG_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// WfmCore
// ------------------------------------------------------------------------------------------------
int WfmCore::init(uint32_t channels, double demod_rate, long long max_n)
{
    C = channels;
    rate = demod_rate;
    max_n_ = max_n;
    stereo.assign(C, 0);
    lp_on = rate >= 150000;  // demod_wfm.cpp:210
    const design::Biquad l = design::biquad_lowpass(75000, 1.0, rate);   // demod_wfm.cpp:164
    const double lpc[5] = {l.b0, l.b1, l.b2, l.a1, l.a2};
    fill_scan_section(lp.sec[0], kBiquadDf2, lpc);
    const double da = 1.0 - std::exp(-1.0 / (rate * 75E-6));            // demod_wfm.cpp:181-183,454
    fill_scan_section(dn.sec[0], kOnePoleAvg, &da);
    const design::Biquad br = design::biquad_notch(19000.0, 5, rate);    // demod_wfm.cpp:178
    const double brc[5] = {br.b0, br.b1, br.b2, br.a1, br.a2};
    fill_scan_section(dn.sec[1], kBiquadDf2, brc);
    // samples are stored fp32 (eps 6e-8): a start-up residue below 1e-10 of the state is invisible
    warm_lp = scan_warm_subchunks(lp.sec, 1, 1e-10);
    warm_dn = scan_warm_subchunks(dn.sec, 2, 1e-10);
    const std::vector<double> h = design::fir_lowpass(0, 1.0, 60.0, 15000.0, 1.4 * 15000.0, rate);  // demod_wfm.cpp:175
    ntaps = (int)h.size();
    std::vector<float> hf(kMaxTaps, 0.f);
    for (size_t i = 0; i < h.size(); i++) hf[i] = (float)h[i];
    PG_HIP(hipMalloc((void **)&d_taps, sizeof(float) * kMaxTaps));
    PG_HIP(hipMemcpy(d_taps, hf.data(), sizeof(float) * kMaxTaps, hipMemcpyHostToDevice));
    // One kernel with no carried state when the cascades decay fast enough to be folded into FIRs (they do at every rate
    // the WFM chain produces: pole radii <= ~0.97); otherwise the exact sequential multi-kernel path below.
    {
        std::vector<double> hl(1, 1.0), ha;
        bool ok = true;
        if (lp_on) ok = design::cascade_impulse(std::vector<double>(), nullptr, std::vector<design::Biquad>(1, l), 1e-11, kWfmLpMax, hl);
        if (ok) ok = design::cascade_impulse(h, &da, std::vector<design::Biquad>(1, br), 1e-11, kWfmIrMax, ha);
        fused = ok && wfm_fir_lds_bytes((int)((ha.size() + 15) & ~(size_t)15), (int)hl.size()) <= 64 * 1024;
        if (fused) {
            L4 = (int)((ha.size() + 15) & ~(size_t)15);
            Llp = (int)hl.size();
            ha.resize((size_t)L4 + 16, 0.0);  // the kernel fetches its taps one chunk of 16 ahead
            std::vector<float> hlf(hl.begin(), hl.end()), haf(ha.begin(), ha.end());
            PG_HIP(hipMalloc((void **)&d_h, sizeof(float) * (L4 + 16)));
            PG_HIP(hipMemcpy(d_h, haf.data(), sizeof(float) * (L4 + 16), hipMemcpyHostToDevice));
            PG_HIP(hipMalloc((void **)&d_hlp, sizeof(float) * Llp));
            PG_HIP(hipMemcpy(d_hlp, hlf.data(), sizeof(float) * Llp, hipMemcpyHostToDevice));
            for (int i = 0; i < 2; i++) {
                PG_HIP(hipMalloc((void **)&d_xtail[i], sizeof(float2) * (size_t)(L4 + Llp) * C));
                PG_HIP(hipMemset(d_xtail[i], 0, sizeof(float2) * (size_t)(L4 + Llp) * C));
            }
            return 0;
        }
    }
    if (int rc = a.alloc((int)C, 2, max_n)) return rc;
    if (int rc = b.alloc((int)C, kMaxTaps, max_n)) return rc;
    if (int rc = c.alloc((int)C, 0, max_n)) return rc;
    for (int i = 0; i < 2; i++) {
        PG_HIP(hipMalloc((void **)&d_lp_state[i], sizeof(double) * 4 * C));
        PG_HIP(hipMemset(d_lp_state[i], 0, sizeof(double) * 4 * C));
        PG_HIP(hipMalloc((void **)&d_dn_state[i], sizeof(double) * 8 * C));
        PG_HIP(hipMemset(d_dn_state[i], 0, sizeof(double) * 8 * C));
    }
    return 0;
}
void WfmCore::release()
{
    a.release(); b.release(); c.release();
    void *p[] = {d_taps, d_lp_state[0], d_lp_state[1], d_dn_state[0], d_dn_state[1], d_h, d_hlp, d_xtail[0], d_xtail[1], d_stereo, d_pilot, d_hilb, d_stereo_list};
    d_stereo = nullptr;
    d_pilot = nullptr;
    d_hilb = nullptr;
    d_stereo_list = nullptr;
    lm.release();
    rds.release();
    for (void *q : p) if (q) (void)hipFree(q);
    d_taps = nullptr;
    d_lp_state[0] = d_lp_state[1] = d_dn_state[0] = d_dn_state[1] = nullptr;
    d_h = nullptr; d_hlp = nullptr; d_xtail[0] = d_xtail[1] = nullptr;
}
int WfmCore::set_stereo(uint32_t ch, bool on)
{
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    if (on && !fused) return fail(PEBBLEGPU_E_UNSUPPORTED, "dmFMS needs the single-kernel WFM path (this demodulator rate runs the sequential one)");
    if (on && !d_pilot) {  // first dmFMS channel of this object: the pilot loop's constants, state and the (L - R) rows
        const design::WfmPilotDesign pd = design::wfm_pilot_design(rate);
        memset(&pilot, 0, sizeof(pilot));
        pilot.b0 = pd.bp.b0; pilot.b2 = pd.bp.b2; pilot.a1 = pd.bp.a1; pilot.a2 = pd.bp.a2;
        pilot.nco_lo = pd.nco_lo; pilot.nco_hi = pd.nco_hi; pilot.alpha = pd.alpha; pilot.beta = pd.beta;
        pilot.err_alpha = pd.err_alpha; pilot.phase_adjust = pd.phase_adjust;
        pilot.L4 = L4;
        PG_HIP(hipMalloc((void **)&d_hilb, sizeof(double) * 122));
        PG_HIP(hipMemcpy(d_hilb, pd.hilb, sizeof(double) * 122, hipMemcpyHostToDevice));
        PG_HIP(hipMalloc((void **)&d_pilot, sizeof(WfmPilotState) * C));
        PG_HIP(hipMalloc((void **)&d_stereo_list, sizeof(int) * C));
        if (int rc = lm.alloc((int)C, L4 + 16, max_n_)) return rc;
        // initPilotPll (demod_wfm.cpp:371-386), once per object as in the reference: a channel that leaves dmFMS and comes back finds its
        // loop where it left it (its lock average still high: it stays dropped)
        std::vector<WfmPilotState> z(C);
        memset(z.data(), 0, sizeof(WfmPilotState) * C);
        for (auto &q : z) { q.nco_freq = pd.nco_freq0; q.quiet = 1LL << 40; }
        PG_HIP(hipMemcpy(d_pilot, z.data(), sizeof(WfmPilotState) * C, hipMemcpyHostToDevice));
        // the RDS members, initialised with the rest of setSampleRate (demod_wfm.cpp:187-191)
        const char *e = getenv("PEBBLEGPU_RDS");
        rds_enabled = !(e && e[0] == '0');
        if (rds_enabled) { if (int rc = rds.init(C, rate, max_n_)) return rc; }
    }
    if (stereo[ch] != (unsigned char)on) stereo_dirty = true;
    stereo[ch] = on;
    return 0;
}
int WfmCore::stereo_lock(hipStream_t s, uint32_t ch, int *lock, int *changed)
{
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    if (last_lock.size() != C) { last_lock.assign(C, 1); stereo_ran.assign(C, 0); }
    int now = 0;
    if (d_pilot && stereo_ran[ch]) {  // a block of processDataStereo has run: locked until the first block that ended without lock
        WfmPilotState st;
        PG_HIP(hipStreamSynchronize(s));
        PG_HIP(hipMemcpy(&st, d_pilot + ch, sizeof(st), hipMemcpyDeviceToHost));
        now = st.dropped ? 0 : 1;
    }
    if (lock) *lock = now;
    if (changed) *changed = now != last_lock[ch] ? 1 : 0;
    last_lock[ch] = (char)now;
    return 0;
}
int WfmCore::run(hipStream_t s, const float2 *in, long long in_pitch, float2 *out, long long out_pitch, long long n,
                 const std::vector<TailJob> *more_tails, const OscAdvance *oa, bool *carried)
{
    if (carried) *carried = false;
    if (n < kMaxTaps) return fail(PEBBLEGPU_E_SIZE, "WFM demod needs at least %d samples per call", kMaxTaps);
    last_n = n;
    if (stereo_dirty) {  // (rare: a mode change)
        if (!d_stereo) PG_HIP(hipMalloc((void **)&d_stereo, C));
        PG_HIP(hipMemcpyAsync(d_stereo, stereo.data(), C, hipMemcpyHostToDevice, s));
        std::vector<int> list;
        for (uint32_t ch = 0; ch < C; ch++) if (stereo[ch]) list.push_back((int)ch);
        n_stereo = (int)list.size();
        if (n_stereo) PG_HIP(hipMemcpyAsync(d_stereo_list, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice, s));
        PG_HIP(hipStreamSynchronize(s));
        stereo_dirty = false;
    }
    if (n_stereo) { if (int rc = rds.check(n, stereo_block)) return rc; }  // (before anything of the call is queued)
    if (fused) {
        WfmFirParams wp;
        memset(&wp, 0, sizeof(wp));
        wp.L4 = L4;
        wp.Llp = Llp;
        wp.gain = 0.25f;  // FMDEMOD_GAIN, demod_wfm.cpp:51
        const long long Lx = (long long)L4 + Llp;
        // next call's history = the last Lx samples of (old history | this call's input), into the other buffer; when the
        // call alone covers it the copy is a tail-refresh job (tail_jobs): carried by this launch when the caller hands over
        // its own jobs, else left to the caller's tail-refresh launch
        float2 *nt = d_xtail[parity ^ 1];
        deferred_in = nullptr;
        if (n >= Lx) {
            deferred_in = in;
            deferred_pitch = in_pitch;
        }
        const int groups = (int)cdiv(n, kWfmOutB);
        TailJobs tj;
        memset(&tj, 0, sizeof(tj));
        int extra = 0;
        if (more_tails != nullptr && n >= Lx) {
            std::vector<TailJob> jobs(*more_tails);
            jobs.push_back(TailJob{const_cast<float2 *>(in), in_pitch, n, (int)Lx, 0, nt, Lx});
            if (fill_tail_jobs(tj, jobs, oa) == 0) {
                extra = (int)jobs.size();
                if (carried) *carried = true;
                deferred_in = nullptr;  // (done here)
            } else {
                memset(&tj, 0, sizeof(tj));
            }
        }
        launch_lds(k_wfm_fir, dim3(groups + extra, C), dim3(512), wfm_fir_lds_bytes(L4, Llp), s, in, in_pitch, (const float2 *)d_xtail[parity],
                   out, out_pitch, n, wp, (const float *)d_h, (const float *)d_hlp, (const unsigned char *)d_stereo, groups, tj);
        PG_HIP(hipGetLastError());
        if (n < Lx) {
            PG_HIP(hipMemcpy2DAsync(nt, sizeof(float2) * Lx, d_xtail[parity] + n, sizeof(float2) * Lx, sizeof(float2) * (Lx - n), C, hipMemcpyDeviceToDevice, s));
            PG_HIP(hipMemcpy2DAsync(nt + (Lx - n), sizeof(float2) * Lx, in, sizeof(float2) * in_pitch, sizeof(float2) * n, C, hipMemcpyDeviceToDevice, s));
        }
        parity ^= 1;
        if (n_stereo) {
            if (last_lock.size() != C) { last_lock.assign(C, 1); stereo_ran.assign(C, 0); }
            for (uint32_t ch = 0; ch < C; ch++) if (stereo[ch]) stereo_ran[ch] = 1;
            // dmFMS before the pilot PLL's drop-out: the (L - R) part of the blocks that end locked, through the same audio response
            if (n > lm.cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
            PG_HIP(hipMemset2DAsync(lm.data(), sizeof(float2) * (size_t)lm.pitch, 0, sizeof(float2) * (size_t)n, C, s));
            WfmPilotParams pp = pilot;
            pp.block = stereo_block > 0 ? stereo_block : (int)n;
            launch(k_wfm_pilot, dim3(cdiv(n_stereo, 64)), dim3(64), s, in, in_pitch, n, pp, (const double *)d_hilb, d_pilot, lm.data(), lm.pitch,
                   (const int *)d_stereo_list, n_stereo);
            launch(k_wfm_lmr_fir, dim3(cdiv(n, 256), n_stereo), dim3(256), s, (const float2 *)lm.data(), lm.pitch, (const float *)d_h, L4, out, out_pitch, n,
                   (const WfmPilotState *)d_pilot, (const int *)d_stereo_list);
            std::vector<TailJob> lj(1, TailJob{lm.data(), lm.pitch, n, lm.hist, 0, nullptr, 0});
            if (int rc = run_save_tails(s, lj, C)) return rc;
            PG_HIP(hipGetLastError());
            // the RDS branch (demod_wfm.cpp:296-357) runs whether the pilot is locked or not
            if (int rc = rds.run(s, in, in_pitch, n, (const double *)d_hilb, (const int *)d_stereo_list, n_stereo, stereo_block)) return rc;
        }
        return 0;
    }
    if (n > a.cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
    const long long nsub = (n + kSub - 1) / kSub;
    // slow poles: one workgroup per channel walks the call sequentially with the exact carried state
    if (lp_on)
        launch(k_iir_scan<0, 1>, dim3(1, C), dim3(64), s, in, in_pitch, a.data(), a.pitch, n, lp, (const double *)d_lp_state[0], d_lp_state[0],
               (int)nsub, -1, (const int *)nullptr, Gate{nullptr, 0, 0});
    else
        launch(k_copy, dim3(cdiv(n, 256), C), dim3(256), s, in, in_pitch, a.data(), a.pitch, n);
    launch(k_discrim, dim3(cdiv(n, 256), C), dim3(256), s, (const float2 *)a.data(), a.pitch, b.data(), b.pitch, n, 0.25f);  // FMDEMOD_GAIN
    launch(k_fir_dec, dim3(cdiv(n, 256), C), dim3(256), s, (const float2 *)b.data(), b.pitch, c.data(), c.pitch, n, 1, (const float *)d_taps,
           (const float *)nullptr, 0, (const int *)nullptr, ntaps, 1.0f, 0, (const int *)nullptr, Gate{nullptr, 0, 0});
    launch(k_iir_scan<1, 2>, dim3(1, C), dim3(64), s, (const float2 *)c.data(), c.pitch, out, out_pitch, n, dn, (const double *)d_dn_state[0],
           d_dn_state[0], (int)nsub, -1, (const int *)nullptr, Gate{nullptr, 0, 0});
    PG_HIP(hipGetLastError());
    return 0;
}
void WfmCore::tail_jobs(std::vector<TailJob> &jobs) const
{
    if (fused) {
        if (deferred_in) jobs.push_back(TailJob{const_cast<float2 *>(deferred_in), deferred_pitch, last_n, L4 + Llp, 0, d_xtail[parity], (long long)(L4 + Llp)});
        return;
    }
    jobs.push_back(TailJob{a.data(), a.pitch, last_n, a.hist, 0, nullptr, 0});
    jobs.push_back(TailJob{b.data(), b.pitch, last_n, b.hist, 0, nullptr, 0});
}

int run_gate_eval(hipStream_t s, const float4 *d_smeter, long long smeter_pitch, int frames_per_sf, int k, const float *d_squelch, unsigned char *d_gate,
                  int stride, uint32_t channels)
{
    launch(k_gate_eval, dim3((unsigned)((channels * (unsigned)k + 255) / 256)), dim3(256), s, d_smeter, smeter_pitch, frames_per_sf, k, d_squelch, d_gate, stride, (int)channels);
    PG_HIP(hipGetLastError());
    return 0;
}
int run_gate_zero(hipStream_t s, float2 *audio, long long pitch, long long spf, const unsigned char *d_gate, int stride, uint32_t channels, int k)
{
    launch(k_gate_zero, dim3((unsigned)((spf + 1023) / 1024), channels, (unsigned)k), dim3(256), s, audio, pitch, spf, d_gate, stride);
    PG_HIP(hipGetLastError());
    return 0;
}

int run_signal_strength(hipStream_t s, const float *d_spec, long long stream_pitch, int bins, long long n_frames, const SmBins *d_bins,
                        float4 *d_out, long long out_pitch, uint32_t channels)
{
    if (n_frames <= 0) return 0;
    launch(k_signal_strength, dim3((unsigned)n_frames, channels), dim3(64), s, d_spec, stream_pitch, bins, n_frames, d_bins, d_out, out_pitch);
    PG_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// AgcCore
// ------------------------------------------------------------------------------------------------
int AgcCore::init(uint32_t channels, double demod_rate)
{
    C = channels;
    rate = demod_rate;
    host.assign(C, Host());
    PG_HIP(hipMalloc((void **)&d_state, sizeof(AgcState) * C));
    PG_HIP(hipMemset(d_state, 0, sizeof(AgcState) * C));
    PG_HIP(hipMalloc((void **)&d_list, sizeof(int) * C));
    std::vector<AgcState> init(1);
    memset(&init[0], 0, sizeof(AgcState));
    init[0].manual_gain = 1.0;  // AGC::AGC -> setAgcMode(AGC_OFF, 1): dBToAmplitude(1 / 5) = 1 (agc.cpp:29,241-245)
    for (uint32_t c = 0; c < C; c++) PG_HIP(hipMemcpy(d_state + c, &init[0], offsetof(AgcState, sig), hipMemcpyHostToDevice));
    return 0;
}
void AgcCore::release()
{
    if (d_state) (void)hipFree(d_state);
    if (d_list) (void)hipFree(d_list);
    d_state = nullptr;
    d_list = nullptr;
}
int AgcCore::set_mode(uint32_t ch, int mode, int threshold)
{
    if (ch >= C || mode < 0 || mode > 4) return fail(PEBBLEGPU_E_INVALID, "bad AGC channel or mode");
    Host &h = host[ch];
    h.mode = mode;
    h.threshold = threshold;
    h.dirty = true;
    list_dirty = true;
    return 0;
}
int AgcCore::apply(hipStream_t s)
{
    if (!list_dirty) return 0;
    for (uint32_t c = 0; c < C; c++) {
        Host &h = host[c];
        if (!h.dirty) continue;
        h.dirty = false;
        // setAgcMode picks the decay for the mode, then setParameters (agc.cpp:53-82, 237-300)
        int decay = 200;
        if (h.mode == 1) decay = 100; else if (h.mode == 3) decay = 500; else if (h.mode == 2) decay = 250; else if (h.mode == 4) decay = 2000;
        AgcState st;
        PG_HIP(hipStreamSynchronize(s));
        PG_HIP(hipMemcpy(&st, d_state + c, offsetof(AgcState, sig), hipMemcpyDeviceToHost));  // running averages survive a parameter change
        st.mode = h.mode;
        if (h.mode == 0) {
            st.manual_gain = std::pow(10, (double)(h.threshold / 5) / 20.0);  // integer division as written
            h.manual = st.manual_gain;
            PG_HIP(hipMemcpy(d_state + c, &st, offsetof(AgcState, sig), hipMemcpyHostToDevice));
            continue;
        }
        const int thr = -h.threshold;
        st.manual_gain = 1;
        h.manual = 1;
        if (!(h.use_hang == 0 && thr == h.thr && h.slope == 0.0 && decay == h.decay && h.sample_rate == rate)) {
            h.use_hang = 0; h.thr = thr; h.slope = 0; h.decay = decay;
            if (h.sample_rate != rate) {  // first set-up: clear the delay line and the averagers (agc.cpp:262-277)
                h.sample_rate = rate;
                std::vector<AgcState> full(1);
                memset(&full[0], 0, sizeof(AgcState));
                for (int i = 0; i < kAgcMaxDelayBuf; i++) full[0].mag[i] = -16.0;
                PG_HIP(hipMemcpy(d_state + c, &full[0], sizeof(AgcState), hipMemcpyHostToDevice));
                st.sig_ptr = 0; st.hang_timer = 0; st.peak = -16.0; st.decay_avg = -5.0; st.attack_avg = -5.0; st.mag_pos = 0;
            }
            const float kDelayTc = .015f, kWindowTc = .018f, kAttackRiseTc = .002f, kAttackFallTc = .005f, kDecayRatio = .3f, kOutScale = 0.7f;
            st.use_hang = 0;
            st.knee = (double)thr / 20.0;
            st.gain_slope = h.slope / 100.0;
            st.fixed_gain = kOutScale * std::pow(10.0, st.knee * (st.gain_slope - 1.0));
            st.attack_rise = 1.0 - std::exp(-1.0 / (rate * kAttackRiseTc));
            st.attack_fall = 1.0 - std::exp(-1.0 / (rate * kAttackFallTc));
            st.decay_rise = 1.0 - std::exp(-1.0 / (rate * (double)decay * .001 * kDecayRatio));
            st.hang_time = (int)(rate * (double)decay * .001);
            st.decay_fall = 1.0 - std::exp(-1.0 / (rate * (double)decay * .001));
            st.delay_samples = (int)(rate * kDelayTc);
            st.window_samples = (int)(rate * kWindowTc);
            if (st.delay_samples >= kAgcMaxDelayBuf - 1) st.delay_samples = kAgcMaxDelayBuf - 1;
            if (st.window_samples > kAgcMaxDelayBuf) return fail(PEBBLEGPU_E_UNSUPPORTED, "AGC window of %d samples exceeds the reference's own buffer", st.window_samples);
        }
        PG_HIP(hipMemcpy(d_state + c, &st, offsetof(AgcState, sig), hipMemcpyHostToDevice));
    }
    list.clear();
    for (uint32_t c = 0; c < C; c++)
        if ((host[c].mode != 0 || host[c].manual != 1.0) && !(c < muted.size() && muted[c])) list.push_back((int)c);
    if (!list.empty()) PG_HIP(hipMemcpyAsync(d_list, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice, s));
    PG_HIP(hipStreamSynchronize(s));
    list_dirty = false;
    return 0;
}
int AgcCore::run(hipStream_t s, float2 *buf, long long pitch, long long n, Gate gate)
{
    if (list.empty()) return 0;  // AGC_OFF with unit manual gain: out = 1.0 * in
    launch(k_agc, dim3(cdiv((long long)list.size(), 64)), dim3(64), s, buf, pitch, n, d_state, (const int *)d_list, (int)list.size(), gate);
    PG_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// ConditionCore, AnfCore
// ------------------------------------------------------------------------------------------------
int ConditionCore::init(uint32_t streams, uint32_t frame, double sample_rate, long long max_n)
{
    S = streams;
    nf = frame;
    fs = sample_rate;
    cap = max_n;
    host.assign(S, Host());
    const design::Biquad h = design::biquad_highpass(10, 0.7071, fs);  // DCRemoval ctor, dcremoval.cpp:5-9
    const double c[5] = {h.b0, h.b1, h.b2, h.a1, h.a2};
    fill_scan_section(dc.sec[0], kBiquadDf2, c);
    return 0;
}
void ConditionCore::release()
{
    void *p[] = {d_buf, d_dc_state, d_dc_list, d_iq, d_nb};
    for (void *q : p) if (q) (void)hipFree(q);
    d_buf = nullptr; d_dc_state = nullptr; d_dc_list = nullptr; d_iq = nullptr; d_nb = nullptr;
}
int ConditionCore::set(uint32_t stream, int flags, double gain, double phase)
{
    if (stream >= S || flags < 0 || flags > 15) return fail(PEBBLEGPU_E_INVALID, "bad stream or conditioner flags");
    Host &h = host[stream];
    if (!d_buf) {  // first use: the conditioned copy and the per-stream state
        PG_HIP(hipMalloc((void **)&d_buf, sizeof(float2) * (size_t)cap * S));
        PG_HIP(hipMalloc((void **)&d_dc_state, sizeof(double) * 4 * S));
        PG_HIP(hipMemset(d_dc_state, 0, sizeof(double) * 4 * S));
        PG_HIP(hipMalloc((void **)&d_dc_list, sizeof(int) * S));
        PG_HIP(hipMalloc((void **)&d_iq, sizeof(double2) * S));
        PG_HIP(hipMalloc((void **)&d_nb, sizeof(NbState) * S));
        std::vector<NbState> nb(S);
        memset(nb.data(), 0, sizeof(NbState) * S);
        for (auto &b : nb) { b.nb_avg_mag = 1; b.nb2_avg_mag = 1; }  // NoiseBlanker ctor, noiseblanker.cpp:9-15
        PG_HIP(hipMemcpy(d_nb, nb.data(), sizeof(NbState) * S, hipMemcpyHostToDevice));
    }
    // setNbEnabled(true) / setNb2Enabled(true) reset their averages (noiseblanker.cpp:20-37)
    const int turned_on = flags & ~h.flags;
    if (turned_on & 12) {
        NbState b;
        PG_HIP(hipDeviceSynchronize());
        PG_HIP(hipMemcpy(&b, d_nb + stream, sizeof(b), hipMemcpyDeviceToHost));
        if (turned_on & 4) { b.spike_count = 0; b.nb_avg_mag = 0; }
        if (turned_on & 8) { b.nb2_avg_mag = 0; b.nb2_avg[0] = b.nb2_avg[1] = 0; }
        PG_HIP(hipMemcpy(d_nb + stream, &b, sizeof(b), hipMemcpyHostToDevice));
    }
    h.flags = flags;
    h.gain = gain;
    h.phase = phase;
    dirty = true;
    return 0;
}
int ConditionCore::apply(hipStream_t s)
{
    if (!dirty) return 0;
    PG_HIP(hipStreamSynchronize(s));
    any = iq_any = nb_any = false;
    dc_list.clear();
    std::vector<double2> iq(S);
    for (uint32_t i = 0; i < S; i++) {
        const Host &h = host[i];
        if (h.flags) any = true;
        if (h.flags & 1) dc_list.push_back((int)i);
        iq[i] = (h.flags & 2) ? make_double2(h.gain, h.phase) : make_double2(-1.0, 0.0);
        if (h.flags & 2) iq_any = true;
        if (h.flags & 12) nb_any = true;
        const int nbf = (h.flags >> 2) & 3;
        PG_HIP(hipMemcpy(&d_nb[i].flags, &nbf, sizeof(int), hipMemcpyHostToDevice));
    }
    if (!dc_list.empty()) PG_HIP(hipMemcpy(d_dc_list, dc_list.data(), sizeof(int) * dc_list.size(), hipMemcpyHostToDevice));
    PG_HIP(hipMemcpy(d_iq, iq.data(), sizeof(double2) * S, hipMemcpyHostToDevice));
    dirty = false;
    return 0;
}
int ConditionCore::run(hipStream_t s, const float2 *d_in, long long in_pitch, long long n, const float2 **out, long long *out_pitch)
{
    *out = d_in;
    *out_pitch = in_pitch;
    if (!any) return 0;
    if (n > cap || n % nf != 0) return fail(PEBBLEGPU_E_SIZE, "conditioners take whole frames within the bank's capacity");
    PG_HIP(hipMemcpy2DAsync(d_buf, sizeof(float2) * (size_t)cap, d_in, sizeof(float2) * (size_t)in_pitch, sizeof(float2) * (size_t)n, S, hipMemcpyDeviceToDevice, s));
    if (!dc_list.empty()) {  // DCRemoval::process: CIir high-pass 10 Hz, exact carried state (its pole sits at 1 - 3e-6 .. 3e-5)
        const long long nsub = (n + kSub - 1) / kSub;
        launch(k_iir_scan<0, 1>, dim3(1, (unsigned)dc_list.size()), dim3(64), s, (const float2 *)d_buf, cap, d_buf, cap, n, dc, (const double *)d_dc_state,
               d_dc_state, (int)nsub, -1, (const int *)d_dc_list, Gate{nullptr, 0, 0});
    }
    if (iq_any) launch(k_iq_balance, dim3(cdiv(n / nf, 64), S), dim3(64), s, d_buf, cap, (int)nf, n / nf, (const double2 *)d_iq);
    if (nb_any) launch(k_noise_blank, dim3(cdiv(S, 64)), dim3(64), s, d_buf, cap, n, d_nb, (int)S);
    PG_HIP(hipGetLastError());
    *out = d_buf;
    *out_pitch = cap;
    return 0;
}

int AnfCore::init(uint32_t channels)
{
    C = channels;
    on.assign(C, 0);
    return 0;
}
void AnfCore::release()
{
    if (d_state) (void)hipFree(d_state);
    if (d_list) (void)hipFree(d_list);
    d_state = nullptr;
    d_list = nullptr;
}
int AnfCore::set(uint32_t ch, bool enable)
{
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    if (!d_state) {
        PG_HIP(hipMalloc((void **)&d_state, sizeof(AnfState) * C));
        PG_HIP(hipMemset(d_state, 0, sizeof(AnfState) * C));  // new CPX[] coefficients and the delay line start at zero
        PG_HIP(hipMalloc((void **)&d_list, sizeof(int) * C));
    }
    on[ch] = enable ? 1 : 0;
    dirty = true;
    return 0;
}
int AnfCore::apply(hipStream_t s)
{
    if (!dirty) return 0;
    list.clear();
    for (uint32_t c = 0; c < C; c++) if (on[c] && !(c < muted.size() && muted[c])) list.push_back((int)c);
    PG_HIP(hipStreamSynchronize(s));
    if (!list.empty()) PG_HIP(hipMemcpy(d_list, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice));
    dirty = false;
    return 0;
}
int AnfCore::run(hipStream_t s, float2 *buf, long long pitch, long long n, Gate gate)
{
    if (list.empty()) return 0;
    launch(k_anf, dim3((unsigned)list.size()), dim3(64), s, buf, pitch, n, d_state, (const int *)d_list, gate);
    PG_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// ResampCore
// ------------------------------------------------------------------------------------------------
int ResampCore::init(uint32_t channels, uint32_t frame, double rate, uint32_t frames_per_call)
{
    C = channels;
    nf = frame;
    dt = rate;
    max_frames = frames_per_call;
    if (!(dt > 0) || dt > 1e6) return fail(PEBBLEGPU_E_INVALID, "bad resampling ratio");
    std::vector<float> t((size_t)kSincLength);
    for (int i = 0; i < kSincLength; i++) {  // fractresampler.cpp:104-118
        const double window = (0.35875 - 0.48829 * std::cos((design::kTwoPi * i) / (kSincLength - 1)) +
                               0.14128 * std::cos((2.0 * design::kTwoPi * i) / (kSincLength - 1)) -
                               0.01168 * std::cos((3.0 * design::kTwoPi * i) / (kSincLength - 1)));
        const double fi = (design::kTwoPi / 2.0) * (double)(i - kSincLength / 2) / (double)kSincPeriodPts;
        t[i] = (float)(i != kSincLength / 2 ? window * std::sin(fi) / fi : 1.0);
    }
    PG_HIP(hipMalloc((void **)&d_sinc, sizeof(float) * kSincLength));
    PG_HIP(hipMemcpy(d_sinc, t.data(), sizeof(float) * kSincLength, hipMemcpyHostToDevice));
    for (int i = 0; i < 2; i++) {
        PG_HIP(hipMalloc((void **)&d_hist[i], sizeof(float2) * kSincPeriods * C));
        PG_HIP(hipMemset(d_hist[i], 0, sizeof(float2) * kSincPeriods * C));  // Init zeroes m_pInputBuf
        PG_HIP(hipHostMalloc((void **)&h_frames[i], sizeof(ResampFrame) * max_frames));
        PG_HIP(hipEventCreateWithFlags(&h_done[i], hipEventDisableTiming));
    }
    PG_HIP(hipMalloc((void **)&d_frames, sizeof(ResampFrame) * max_frames));
    return 0;
}
void ResampCore::release()
{
    if (d_sinc) (void)hipFree(d_sinc);
    if (d_frames) (void)hipFree(d_frames);
    for (int i = 0; i < 2; i++) {
        if (d_hist[i]) (void)hipFree(d_hist[i]);
        if (h_frames[i]) (void)hipHostFree(h_frames[i]);
        if (h_done[i]) (void)hipEventDestroy(h_done[i]);
        d_hist[i] = nullptr; h_frames[i] = nullptr; h_done[i] = nullptr;
    }
    d_sinc = nullptr;
    d_frames = nullptr;
}
int ResampCore::run(hipStream_t s, const float2 *in, long long in_pitch, long long n, float2 *out, long long out_pitch, long long *n_out)
{
    if (n % nf != 0 || n / nf > (long long)max_frames) return fail(PEBBLEGPU_E_SIZE, "resampler input %lld is not 1..%u frames of %u", n, max_frames, nf);
    if (n < kSincPeriods) return fail(PEBBLEGPU_E_SIZE, "resampler needs at least %d samples per call", kSincPeriods);
    const int F = (int)(n / nf);
    PG_HIP(hipEventSynchronize(h_done[pin]));  // the copy that last used this staging buffer has run
    ResampFrame *hf = h_frames[pin];
    // replay of the reference's time arithmetic, frame by frame (fractresampler.cpp:155-187)
    double t = float_time;
    int total = 0, max_per_frame = 0;
    for (int f = 0; f < F; f++) {
        hf[f].t_start = t;
        hf[f].out_offset = total;
        int it = (int)t, cnt = 0;
        while (it < (int)nf) {
            cnt++;
            t += dt;
            it = (int)t;
        }
        t -= (double)nf;
        hf[f].nout = cnt;
        total += cnt;
        if (cnt > max_per_frame) max_per_frame = cnt;
    }
    float_time = t;
    *n_out = total;
    if (total > out_pitch) return fail(PEBBLEGPU_E_SIZE, "resampler output %d exceeds its buffer", total);
    PG_HIP(hipMemcpyAsync(d_frames, hf, sizeof(ResampFrame) * F, hipMemcpyHostToDevice, s));
    PG_HIP(hipEventRecord(h_done[pin], s));
    pin ^= 1;
    if (total > 0) {
        const int sub = (int)cdiv(max_per_frame, 256);
        launch(k_resample, dim3((unsigned)(F * sub), C), dim3(256), s, in, in_pitch, (const float2 *)d_hist[parity], out, out_pitch, (int)nf, sub, dt,
               (const ResampFrame *)d_frames, (const float *)d_sinc);
        PG_HIP(hipGetLastError());
    }
    // m_pInputBuf[0..27] <- the last 28 inputs (fractresampler.cpp:189-193)
    PG_HIP(hipMemcpy2DAsync(d_hist[parity ^ 1], sizeof(float2) * kSincPeriods, in + (n - kSincPeriods), sizeof(float2) * in_pitch,
                            sizeof(float2) * kSincPeriods, C, hipMemcpyDeviceToDevice, s));
    parity ^= 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// SpectrumCore
// ------------------------------------------------------------------------------------------------
int SpectrumCore::init(uint32_t streams, uint32_t frame, uint32_t fft_size)
{
    S = streams;
    nf = frame;
    bins = fft_size;
    if (bins < 2048) bins = 2048;    // fft.cpp:74-75
    if (bins > 65535) bins = 65535;  // fft.cpp:76-77 (and then not a power of two)
    big = frame == (uint32_t)kBigN && fft_size == (uint32_t)kBigN;  // BASELINE config 5: past the fft.h:21 clamp on purpose
    if (big) bins = kBigN;
    const bool pow2_zp = nf == 2048 && (bins == 2048 || bins == 4096 || bins == 8192 || bins == 16384 || bins == 32768);
    // every other frame length (settings.cpp:57) goes through the general kernel: the frame rounded up to a power of two M <= 16384,
    // bins a power of two >= M (Accelerate's radix-2 transform takes no other sizes either)
    {
        int M = 256, lg = 8;
        while ((uint32_t)M < nf) { M *= 2; lg++; }
        int zl = 0;
        while (((uint32_t)M << zl) < bins) zl++;
        const bool ok = !big && M <= 16384 && ((uint32_t)M << zl) == bins && bins <= 32768;
        any = !big && !pow2_zp;
        if (any && !ok)
            return fail(PEBBLEGPU_E_UNSUPPORTED, "spectrum needs frames of at most 16384 samples and a power-of-two bin count of at least the frame length, at most 32768 (asked %u/%u)", nf, bins);
        if (ok) {
            any_M = M; any_logM = lg; any_zp_log2 = zl;
            std::vector<float2> tw((size_t)M / 2);
            for (int k = 0; k < M / 2; k++) tw[k] = make_float2((float)std::cos(-design::kTwoPi * k / M), (float)std::sin(-design::kTwoPi * k / M));
            PG_HIP(hipMalloc((void **)&d_twM, sizeof(float2) * tw.size()));
            PG_HIP(hipMemcpy(d_twM, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice));
        }
    }
    // One transform per 128-item workgroup (k_spectrum_q128) serves every power-of-two zero-padding; measured against the
    // shared-frame kernels on the bench batch it wins at 2048 bins (0.098 vs 0.105 ms) and loses at 4096 (0.198 vs 0.185) and
    // 8192 (0.44 vs 0.28: its ZP workgroups each re-read the frame and store 4-byte bins ZP*4 bytes apart), so it runs where
    // it wins and where nothing else exists (16384, 32768).  PEBBLEGPU_SPECTRUM_PERQ=1 forces it everywhere (A/B runs).
    { const char *e = getenv("PEBBLEGPU_SPECTRUM_W64"); use_w64 = e && e[0] == '1'; }  // the one-wave 8192-bin kernel (measured equal: opt-in)
    const char *env = getenv("PEBBLEGPU_SPECTRUM_PERQ");
    per_q = !big && !any && (bins == 2048 || bins > 8192 || (env && env[0] == '1'));
    std::vector<double> w;
    const double cg = design::blackman_harris(nf, w);
    std::vector<float> wf(nf);
    for (uint32_t i = 0; i < nf; i++) wf[i] = (float)w[i];
    PG_HIP(hipMalloc((void **)&d_window, sizeof(float) * nf));
    PG_HIP(hipMemcpy(d_window, wf.data(), sizeof(float) * nf, hipMemcpyHostToDevice));
    h_window = wf;
    // btab[q][m] = exp(-2*pi*i*(64*m*q)/bins): the wave-uniform factor of the pruned-FFT pre-twiddle W_bins^{n q},
    // n = lane + 64 m (the per-lane factor W_bins^{lane q} is computed in the kernel)
    const uint32_t zp = bins / nf;
    std::vector<float2> bt((size_t)zp * 32);
    for (uint32_t q = 0; q < zp; q++)
        for (uint32_t m = 0; m < 32; m++) {
            const uint64_t k = ((uint64_t)64 * m * q) % bins;
            const double a = -design::kTwoPi * (double)k / (double)bins;
            bt[(size_t)q * 32 + m] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    PG_HIP(hipMalloc((void **)&d_btab, sizeof(float2) * bt.size()));
    PG_HIP(hipMemcpy(d_btab, bt.data(), sizeof(float2) * bt.size(), hipMemcpyHostToDevice));
    if (per_q) {
        std::vector<float2> ft((size_t)zp * nf);
        for (uint32_t q = 0; q < zp; q++)
            for (uint32_t n = 0; n < nf; n++) {
                const double a = -design::kTwoPi * (double)(((uint64_t)n * q) % bins) / (double)bins;
                ft[(size_t)q * nf + n] = make_float2((float)(w[n] * std::cos(a)), (float)(w[n] * std::sin(a)));
            }
        PG_HIP(hipMalloc((void **)&d_ftab, sizeof(float2) * ft.size()));
        PG_HIP(hipMemcpy(d_ftab, ft.data(), sizeof(float2) * ft.size(), hipMemcpyHostToDevice));
        if (int rc = make_twiddles_t128(&d_tw128)) return rc;
    }
    if (bins == 8192 && !big && !per_q) {
        const char *e1 = getenv("PEBBLEGPU_T128_STAGGER"), *e2 = getenv("PEBBLEGPU_T128_PADLDS");
        stagger = e1 ? atoi(e1) : 3;  // measured on the bench batch: 0.300 ms as two 512-item workgroups per CU, 0.278 with the halves three intervals apart
        pad_lds = e2 ? atoi(e2) : 0;
        std::vector<float2> b2(4 * 16);
        for (int q = 0; q < 4; q++)
            for (int m = 0; m < 16; m++) {
                const double a = -design::kTwoPi * (double)((128 * m * q) % 8192) / 8192.0;
                b2[q * 16 + m] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        PG_HIP(hipMalloc((void **)&d_btab128, sizeof(float2) * b2.size()));
        PG_HIP(hipMemcpy(d_btab128, b2.data(), sizeof(float2) * b2.size(), hipMemcpyHostToDevice));
        if (int rc = make_twiddles_t128q(&d_tw128)) return rc;  // (k_spectrum_t128's own: one table per q)
    }
    scale = (float)(1.0 / (cg * (double)nf));  // /coherentGain then /maxBinPower, fft.cpp:347,355
    if (int rc = make_twiddles(2048, &d_tw_nf)) return rc;
    for (int i = 0; i < 2; i++) {
        PG_HIP(hipMalloc((void **)&d_prev[i], sizeof(float) * (size_t)bins * S));
        PG_HIP(hipMemset(d_prev[i], 0, sizeof(float) * (size_t)bins * S));  // the reference leaves these uninitialised (fft.cpp:107-115)
    }
    return 0;
}
void SpectrumCore::release()
{
    void *p[] = {d_window, d_btab, d_prev[0], d_prev[1], d_tw_nf, d_Y, d_btab128, d_tw128, d_ftab, d_twM};
    d_twM = nullptr;
    for (void *q : p) if (q) (void)hipFree(q);
    d_ftab = nullptr;
    d_window = nullptr; d_btab = nullptr; d_prev[0] = d_prev[1] = nullptr; d_tw_nf = nullptr; d_Y = nullptr; d_btab128 = d_tw128 = nullptr;
    y_cap = 0;
}
// FFT::fftSpectrum for any frame length (and for fewer samples than samplesPerBuffer: copied, zero-padded, not windowed, fft.cpp:129-157)
int SpectrumCore::run_any(hipStream_t s, const float2 *d_in, long long in_pitch, long long F, float *d_out, int n_in, bool windowed)
{
    if (!d_twM) return fail(PEBBLEGPU_E_UNSUPPORTED, "no general display transform for %u-sample frames and %u bins", nf, bins);
    if (n_in <= 0 || n_in > any_M || (uint32_t)n_in > nf) return fail(PEBBLEGPU_E_SIZE, "%d samples do not fit a frame of %u", n_in, nf);
    if (F == 0) return 0;
    SpectrumParams sp;
    sp.in_pitch = in_pitch;
    sp.n_frames = F;
    const int zp = 1 << any_zp_log2;
    long long G = (F * (long long)S * zp) / 1024;  // every chain recomputes one frame: chains as long as leaves about a thousand workgroups
    G = G < 1 ? 1 : (G > 32 ? 32 : G);
    sp.frames_per_group = (int)G;
    sp.scale = scale;
    sp.out_pitch = F * (long long)bins;
    AnySpecParams ap;
    ap.n_in = n_in;
    ap.frame = (int)nf;
    ap.M = any_M;
    ap.logM = any_logM;
    ap.zp_log2 = any_zp_log2;
    ap.windowed = windowed ? 1 : 0;
    launch_lds(k_spectrum_any, dim3((unsigned)(cdiv(F, G) * zp), S), dim3(256), sizeof(float2) * (size_t)any_M, s, d_in, d_out, (const float *)d_window, (const float2 *)d_twM,
               (const float *)d_prev[parity], d_prev[parity ^ 1], sp, ap);
    parity ^= 1;
    PG_HIP(hipGetLastError());
    return 0;
}
int SpectrumCore::run(hipStream_t s, const float2 *d_in, long long in_pitch, long long F, float *d_out, const RawSrc *raw, const DecFuse *df, bool nothing_beside)
{
    if (df && !dec_ready()) return fail(PEBBLEGPU_E_INVALID, "the decimator was handed to a display transform that cannot run it");
    last_fullc = nothing_beside && !df && dec_ready();
    if (raw && !raw_ready()) return fail(PEBBLEGPU_E_INVALID, "raw-format input reached a spectrum kernel that has no converting loads");
    if (any) {
        if (raw || df) return fail(PEBBLEGPU_E_INVALID, "the general display transform takes float2 input and runs no decimator");
        return run_any(s, d_in, in_pitch, F, d_out, (int)nf, true);
    }
    SpectrumParams sp;
    sp.in_pitch = in_pitch;
    sp.n_frames = F;
    if (big) {
        if (F == 0) return 0;
        // The four-step intermediate Y (8 bytes per point, written by pass A, read by pass B) is kept on the die: streams are
        // processed in batches whose Y fits well inside the 256 MiB Infinity Cache together with the batch's own input and output
        // (one 64 MiB buffer reused by every batch), instead of a call-sized Y that went out to HBM and back
        // (28 B moved per 12 B of algorithmic traffic).
        const long long per_stream = (long long)F * kBigN;                       // Y points per stream
        static const long long batch_mb = [] { const char *e = getenv("PEBBLEGPU_BIG_BATCH_MB"); return e ? atoll(e) : 0LL; }();  // 0: the whole call in one pair of launches (measured: 16 / 32 / 64 / 128 MiB batches 0.61 / 0.48 / 0.39 / 0.34 ms against 0.34 whole: the kernels are not bound by that traffic)
        long long bs = batch_mb > 0 ? (batch_mb << 20) / (long long)sizeof(float2) / per_stream : (long long)S;    // streams per batch
        bs = bs < 1 ? 1 : (bs > (long long)S ? (long long)S : bs);
        const size_t need = (size_t)bs * (size_t)per_stream;
        if (need > y_cap) {
            PG_HIP(hipStreamSynchronize(s));
            if (d_Y) (void)hipFree(d_Y);
            d_Y = nullptr;
            y_cap = 0;
            PG_HIP(hipMalloc((void **)&d_Y, sizeof(float2) * need));
            y_cap = need;
        }
        // a chain that does not start at frame 0 recomputes one frame: prefer chains as long as keeps >= 512 workgroups
        long long G = (F * bs * 8) / 512;
        G = G < 1 ? 1 : (G > 16 ? 16 : G);
        sp.frames_per_group = (int)G;
        sp.scale = scale;
        sp.out_pitch = F * (long long)bins;
        for (long long s0 = 0; s0 < (long long)S; s0 += bs) {
            const unsigned nb = (unsigned)((long long)S - s0 < bs ? (long long)S - s0 : bs);
            static const bool split32 = [] { const char *e = getenv("PEBBLEGPU_BIG_SPLIT32"); return e && e[0] == '1'; }();  // A/B: the 32 x 2048 split
            if (split32) {
                launch(k_big_cols, dim3((unsigned)(F * 8), nb), dim3(256), s, d_in + s0 * in_pitch, (long long)in_pitch, d_Y, (const float *)d_window, (long long)F);
                launch(k_big_rows, dim3((unsigned)(cdiv(F, G) * 8), nb), dim3(256), s, (const float2 *)d_Y, d_out + s0 * sp.out_pitch, (const float2 *)d_tw_nf,
                       (const float *)d_prev[parity] + s0 * kBigN, d_prev[parity ^ 1] + s0 * kBigN, sp);
            } else {
                launch(k_big256_cols, dim3((unsigned)(F * 8), nb), dim3(256), s, d_in + s0 * in_pitch, (long long)in_pitch, d_Y, (const float *)d_window, (long long)F);
                launch(k_big256_rows, dim3((unsigned)(cdiv(F, G) * 8), nb), dim3(256), s, (const float2 *)d_Y, d_out + s0 * sp.out_pitch,
                       (const float *)d_prev[parity] + s0 * kBigN, d_prev[parity ^ 1] + s0 * kBigN, sp);
            }
        }
        parity ^= 1;
        PG_HIP(hipGetLastError());
        return 0;
    }
    if (per_q) {
        // one (chain, q) per 128-item workgroup, eight workgroups per CU: aim at 2048 resident workgroups; every chain
        // recomputes one frame, so chains are as long as that allows
        const int zp = (int)(bins / nf);
        int zl = 0;
        while ((1 << zl) < zp) zl++;
        long long Gq = (F * (long long)S * zp) / 2048;
        Gq = Gq < 1 ? 1 : (Gq > 32 ? 32 : Gq);
        sp.frames_per_group = (int)Gq;
        sp.scale = scale;
        sp.out_pitch = F * (long long)bins;
        const long long chains = cdiv(F, Gq);
        static const bool f_regs = [] { const char *e = getenv("PEBBLEGPU_SPECTRUM_FREGS"); return e && e[0] == '1'; }();
        launch(f_regs ? k_spectrum_q128<true> : k_spectrum_q128<false>, dim3((unsigned)(8 * zp * cdiv(chains, 8)), S), dim3(128), s, d_in, d_out, (const float2 *)d_ftab, (const float2 *)d_tw128,
               (const float *)d_prev[parity], d_prev[parity ^ 1], sp, zl);
        parity ^= 1;
        PG_HIP(hipGetLastError());
        return 0;
    }
    if (bins == 8192 && use_w64) {
        // one-wave transforms (fft_w64.h), one frame chain per 256-item workgroup, two workgroups per CU: chains as long as
        // keeps every workgroup resident at once (every chain recomputes one frame)
        long long Gw = cdiv(F * (long long)S, 512);
        Gw = Gw < 1 ? 1 : (Gw > 32 ? 32 : Gw);
        sp.frames_per_group = (int)Gw;
        sp.scale = scale;
        sp.out_pitch = F * (long long)bins;
        const RawSrc rs = raw ? *raw : RawSrc{nullptr, 0, 0, 0.f, 0};
        const dim3 grid((unsigned)cdiv(F, Gw), S), block(256);
        auto go = [&](auto kern) { launch(kern, grid, block, s, d_in, d_out, (const float *)d_window, (const float2 *)d_btab, (const float *)d_prev[parity], d_prev[parity ^ 1], sp, rs); };
        switch (raw ? raw->fmt : -1) {
        case -1: go(k_spectrum_w64<-1>); break;
        case 0: go(k_spectrum_w64<0>); break;
        case 1: go(k_spectrum_w64<1>); break;
        case 2: go(k_spectrum_w64<2>); break;
        case 3: go(k_spectrum_w64<3>); break;
        default: go(k_spectrum_w64<4>); break;
        }
        parity ^= 1;
        PG_HIP(hipGetLastError());
        return 0;
    }
    if (bins == 8192) {
        // two-wave transforms, one frame chain per 512-thread workgroup; all workgroups resident at once (2 per CU on 256
        // CUs) when the batch allows: every chain recomputes one frame, so longer chains also mean less repeated work
        long long G8 = (F * (long long)S) / 512;
        G8 = G8 < 1 ? 1 : (G8 > 32 ? 32 : G8);
        sp.frames_per_group = (int)G8;
        sp.scale = scale;
        sp.out_pitch = F * (long long)bins;
        const RawSrc rs = raw ? *raw : RawSrc{nullptr, 0, 0, 0.f, 0};
        const float *pin = d_prev[parity];
        float *pout = d_prev[parity ^ 1];
        DecFuse none;
        memset(&none, 0, sizeof(none));
        const DecFuse &dfv = df ? *df : none;
        if (df) {  // the one-channel decimator rides in the transform's workgroups (two chains per workgroup, every sample format)
            const dim3 grid(cdiv(cdiv(F, G8), 2), S), block(1024);
            const int st = stagger > 0 ? stagger : 3;
            auto go = [&](auto kern) { launch(kern, grid, block, s, d_in, d_out, (const float *)d_window, (const float2 *)d_btab128, (const float2 *)d_tw128, pin, pout, sp, st, rs, dfv); };
            switch (raw ? raw->fmt : -1) {
            case -1: go(k_spectrum_t128<2, -1, true>); break;
            case 0: go(k_spectrum_t128<2, 0, true>); break;
            case 1: go(k_spectrum_t128<2, 1, true>); break;
            case 2: go(k_spectrum_t128<2, 2, true>); break;
            case 3: go(k_spectrum_t128<2, 3, true>); break;
            default: go(k_spectrum_t128<2, 4, true>); break;
            }
        } else if (last_fullc) {
            // nothing is to run beside this launch: pass C's fifteen twiddles per work-item formed once and held (126 registers: four such
            // waves fill a SIMD's register file) -- 0.200 ms for the bench batch against 0.224
            const dim3 grid(cdiv(cdiv(F, G8), 2), S), block(1024);
            const int st = stagger > 0 ? stagger : 3;
            auto go = [&](auto kern) { launch(kern, grid, block, s, d_in, d_out, (const float *)d_window, (const float2 *)d_btab128, (const float2 *)d_tw128, pin, pout, sp, st, rs, dfv); };
            switch (raw ? raw->fmt : -1) {
            case -1: go(k_spectrum_t128<2, -1, false, true>); break;
            case 0: go(k_spectrum_t128<2, 0, false, true>); break;
            case 1: go(k_spectrum_t128<2, 1, false, true>); break;
            case 2: go(k_spectrum_t128<2, 2, false, true>); break;
            case 3: go(k_spectrum_t128<2, 3, false, true>); break;
            default: go(k_spectrum_t128<2, 4, false, true>); break;
            }
        } else if (raw) {  // raw-format frames take the two-chain kernel (one instantiation per sample format)
            const dim3 grid(cdiv(cdiv(F, G8), 2), S), block(1024);
            const int st = stagger > 0 ? stagger : 7;
            auto go = [&](auto kern) { launch(kern, grid, block, s, d_in, d_out, (const float *)d_window, (const float2 *)d_btab128, (const float2 *)d_tw128, pin, pout, sp, st, rs, dfv); };
            switch (raw->fmt) {
            case 0: go(k_spectrum_t128<2, 0>); break;
            case 1: go(k_spectrum_t128<2, 1>); break;
            case 2: go(k_spectrum_t128<2, 2>); break;
            case 3: go(k_spectrum_t128<2, 3>); break;
            default: go(k_spectrum_t128<2, 4>); break;
            }
        } else if (stagger > 0)  // two chains per 1024-item workgroup, the second `stagger` barrier intervals behind the first
            launch(k_spectrum_t128<2, -1>, dim3(cdiv(cdiv(F, G8), 2), S), dim3(1024), s, d_in, d_out, (const float *)d_window,
                   (const float2 *)d_btab128, (const float2 *)d_tw128, pin, pout, sp, stagger, rs, dfv);
        else
            launch_lds(k_spectrum_t128<1, -1>, dim3(cdiv(F, G8), S), dim3(512), (size_t)pad_lds, s, d_in, d_out, (const float *)d_window,
                       (const float2 *)d_btab128, (const float2 *)d_tw128, pin, pout, sp, 0, rs, dfv);
        parity ^= 1;
        PG_HIP(hipGetLastError());
        return 0;
    }
    const int groups = 4 / (int)(bins / nf);  // wave groups (frames in flight) per workgroup
    long long G = (F * (long long)S) / (1024 * groups);  // aim for ~1024 workgroups; each group recomputes one extra frame
    G = G < 1 ? 1 : (G > 16 ? 16 : G);
    sp.frames_per_group = (int)G;
    sp.scale = scale;
    sp.out_pitch = F * (long long)bins;
    const dim3 grid(cdiv(F, G * groups), S), block(256);
    const float *pin = d_prev[parity];
    float *pout = d_prev[parity ^ 1];
    if (bins == 2048) launch(k_spectrum_1to1, grid, block, s, d_in, d_out, (const float *)d_window, (const float2 *)d_tw_nf, pin, pout, sp);
    else if (bins == 4096) launch(k_spectrum<2>, grid, block, s, d_in, d_out, (const float *)d_window, (const float2 *)d_btab, (const float2 *)d_tw_nf, pin, pout, sp);
    else return fail(PEBBLEGPU_E_UNSUPPORTED, "no spectrum kernel for %u bins", bins);
    parity ^= 1;
    PG_HIP(hipGetLastError());
    return 0;
}

}  // namespace pg
