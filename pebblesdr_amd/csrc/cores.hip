// cores.hip -- the device cores (see receiver.h): state allocation, parameter design upload, kernel launches.
// This is the only translation unit that instantiates kernels.
#include <cmath>
#include <cstdarg>
#include "kernels_demod.h"
#include "kernels_fastfir.h"
#include "kernels_frontend.h"
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
    std::vector<float2> h(n);
    for (int m = 0; m < n; m++) {
        const double a = -design::kTwoPi * (double)m / (double)n;
        h[m] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    PG_HIP(hipMalloc((void **)d_tw, sizeof(float2) * n));
    PG_HIP(hipMemcpy(*d_tw, h.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// OscBank
// ------------------------------------------------------------------------------------------------
int OscBank::init(uint32_t channels, double sample_rate)
{
    C = channels;
    fs = sample_rate;
    ctl.assign(C, Ctl());
    h_osc.assign(C, ChanOsc());
    PG_HIP(hipMalloc((void **)&d_osc, sizeof(ChanOsc) * C));
    std::vector<float> amp(kAmpTab);
    design::mixer_amplitudes(amp.data(), kAmpTab, &a_inf);
    PG_HIP(hipMalloc((void **)&d_amp, sizeof(float) * kAmpTab));
    PG_HIP(hipMemcpy(d_amp, amp.data(), sizeof(float) * kAmpTab, hipMemcpyHostToDevice));
    return 0;
}
void OscBank::release()
{
    if (d_osc) (void)hipFree(d_osc);
    if (d_amp) (void)hipFree(d_amp);
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
}
int OscBank::upload(hipStream_t s)
{
    for (uint32_t ch = 0; ch < C; ch++) {
        Ctl &c = ctl[ch];
        ChanOsc &o = h_osc[ch];
        if (c.dirty) {
            for (int d = 0; d < kMaxTaps; d++) {
                double ph = (double)d * c.inc;
                ph -= std::floor(ph);
                o.step[d] = make_float2((float)std::cos(design::kTwoPi * ph), (float)std::sin(design::kTwoPi * ph));
            }
            o.inc = c.inc;
            o.mix_on = (-c.freq) != 0 ? 1u : 0u;  // if (m_frequency == 0) return in;  (mixer.cpp:51-53)
            c.dirty = false;
        }
        o.phase0 = c.phase0;
        o.n0 = (uint32_t)(c.n0 > (uint64_t)kAmpTab ? (uint64_t)kAmpTab : c.n0);
    }
    PG_HIP(hipMemcpyAsync(d_osc, h_osc.data(), sizeof(ChanOsc) * C, hipMemcpyHostToDevice, s));
    PG_HIP(hipStreamSynchronize(s));
    return 0;
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

// stand-alone mixer launch (Mixer::processBlock), used by the Mixer step
int run_mixer(hipStream_t s, const float2 *d_in, float2 *d_out, long long n, const OscBank &osc)
{
    launch(k_mixer, dim3(cdiv(n, 1024), osc.C), dim3(256), s, d_in, n, 0, d_out, n, n, (const ChanOsc *)osc.d_osc,
           (const float *)osc.d_amp, osc.a_inf);
    PG_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// DecimCore
// ------------------------------------------------------------------------------------------------
int DecimCore::init(uint32_t channels, const design::Chain &c, long long max_in, int last_hist, float last_gain)
{
    release();
    chain = c;
    C = channels;
    const size_t ns = chain.stages.size();
    stage.assign(ns, HistBuf());
    taps.resize(ns);
    lens.assign(ns, 0);
    std::vector<float> htaps(ns * kMaxTaps + 1, 0.f);
    long long len = max_in;
    for (size_t s = 0; s < ns; s++) {
        const design::Stage &st = chain.stages[s];
        len /= st.stride;
        FirTaps &t = taps[s];
        memset(&t, 0, sizeof(t));
        t.ntaps = st.ntaps;
        t.stride = (int)st.stride;
        t.cic3 = st.ntaps == 0;
        t.gain = (s + 1 == ns) ? last_gain : 1.f;
        for (int p = 0; p < st.ntaps; p++) t.h[p] = htaps[s * kMaxTaps + p] = (float)design::halfband_taps(st.design)[p];
        const int hist = (s + 1 < ns) ? chain.stages[s + 1].ntaps - 1 : last_hist;
        if (int rc = stage[s].alloc((int)C, hist, len)) return rc;
    }
    PG_HIP(hipMalloc((void **)&d_taps, sizeof(float) * htaps.size()));
    PG_HIP(hipMemcpy(d_taps, htaps.data(), sizeof(float) * htaps.size(), hipMemcpyHostToDevice));
    PG_HIP(hipMalloc((void **)&d_hist_mixed, sizeof(float2) * kMaxTaps * C));
    PG_HIP(hipMemset(d_hist_mixed, 0, sizeof(float2) * kMaxTaps * C));
    return 0;
}
void DecimCore::release()
{
    for (auto &b : stage) b.release();
    stage.clear();
    if (d_taps) (void)hipFree(d_taps);
    if (d_hist_mixed) (void)hipFree(d_hist_mixed);
    d_taps = nullptr;
    d_hist_mixed = nullptr;
}
int DecimCore::run(hipStream_t s, const float2 *d_in, long long in_pitch, bool shared_input, long long n, const OscBank &osc,
                   hipEvent_t after_first)
{
    const size_t ns = chain.stages.size();
    if (n <= 0 || n % (long long)chain.total != 0)
        return fail(PEBBLEGPU_E_SIZE, "%lld samples is not a multiple of the decimation %u", n, chain.total);
    long long len = n;
    for (size_t k = 0; k < ns; k++) {
        len /= taps[k].stride;
        if (len > stage[k].cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
        // a stage whose output is shorter than its consumer's look-back cannot refresh that history from one call:
        // this is where the reference degrades to unfiltered sample dropping (decimator.cpp:602-625)
        if (len < stage[k].hist)
            return fail(PEBBLEGPU_E_SIZE, "frame too short for this chain: stage %zu yields %lld samples, its consumer needs %d", k, len,
                        stage[k].hist);
        lens[k] = len;
    }
    launch(k_mix_dec1, dim3(cdiv(lens[0], 256), C), dim3(256), s, d_in, in_pitch, (int)shared_input, stage[0].data(), stage[0].pitch,
           lens[0], (const ChanOsc *)osc.d_osc, (const float2 *)d_hist_mixed, (int)kMaxTaps, (const float *)osc.d_amp, osc.a_inf, taps[0]);
    launch(k_mix_tail, dim3(C), dim3(64), s, d_in, in_pitch, (int)shared_input, n, (const ChanOsc *)osc.d_osc, d_hist_mixed, (int)kMaxTaps,
           (const float *)osc.d_amp, osc.a_inf, taps[0].ntaps, taps[0].stride, taps[0].cic3);
    if (after_first) PG_HIP(hipEventRecord(after_first, s));
    for (size_t k = 1; k < ns; k++)
        launch(k_fir_dec, dim3(cdiv(lens[k], 256), C), dim3(256), s, (const float2 *)stage[k - 1].data(), stage[k - 1].pitch, stage[k].data(),
               stage[k].pitch, lens[k], taps[k].stride, (const float *)(d_taps + k * kMaxTaps), 0, (const int *)nullptr, taps[k].ntaps,
               taps[k].gain, (const int *)nullptr);
    PG_HIP(hipGetLastError());
    return 0;
}
void DecimCore::tail_jobs(std::vector<TailJob> &jobs) const
{
    for (size_t k = 0; k < stage.size(); k++)
        if (stage[k].hist > 0) jobs.push_back(TailJob{stage[k].data(), stage[k].pitch, lens[k], stage[k].hist, 0});
}

int run_save_tails(hipStream_t s, const std::vector<TailJob> &jobs, uint32_t channels)
{
    if (jobs.empty()) return 0;
    if (jobs.size() > (size_t)kMaxTailJobs) return fail(PEBBLEGPU_E_INVALID, "too many history buffers");
    TailJobs tj;
    memset(&tj, 0, sizeof(tj));
    tj.count = (int)jobs.size();
    int maxh = 1;
    for (size_t i = 0; i < jobs.size(); i++) {
        tj.job[i] = jobs[i];
        if (jobs[i].hist > maxh) maxh = jobs[i].hist;
    }
    launch(k_save_tails, dim3(cdiv(maxh, 256), channels, (unsigned)jobs.size()), dim3(256), s, tj);
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
    return make_twiddles((int)fft_n, &d_tw);
}
void FastFirCore::release()
{
    if (d_H) (void)hipFree(d_H);
    if (d_tw) (void)hipFree(d_tw);
    d_H = d_tw = nullptr;
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
int FastFirCore::run(hipStream_t s, const HistBuf &in, long long n, float2 *out, long long out_pitch)
{
    const int overlap = (int)taps - 1;
    const long long L = block_len();
    if (n % L != 0) return fail(PEBBLEGPU_E_SIZE, "FastFIR input %lld is not a multiple of its block %lld", n, L);
    const dim3 grid((unsigned)(n / L), C), block(256);
    if (fft_n == 2048) launch(k_fastfir<2048>, grid, block, s, (const float2 *)in.data(), in.pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw, overlap);
    else if (fft_n == 4096) launch(k_fastfir<4096>, grid, block, s, (const float2 *)in.data(), in.pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw, overlap);
    else launch(k_fastfir<8192>, grid, block, s, (const float2 *)in.data(), in.pitch, out, out_pitch, (const float2 *)d_H, (const float2 *)d_tw, overlap);
    PG_HIP(hipGetLastError());
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
int AmCore::run(hipStream_t s, const float2 *in, long long in_pitch, float2 *out, long long out_pitch, long long n)
{
    if (list.empty()) return 0;
    if (n > tmp.cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
    if (n < tmp.hist) return fail(PEBBLEGPU_E_SIZE, "AM demod needs at least %d samples per call", tmp.hist);
    const unsigned na = (unsigned)list.size();
    const long long nsub = (n + kSub - 1) / kSub;
    // One workgroup per channel walks the call sequentially (the 0.9999 pole forbids a warm-up), so it may read
    // and write the same state slot; a channel that leaves AM keeps its stale state like the idle Demod_AM object.
    launch(k_iir_scan<2, 1>, dim3(1, na), dim3(64), s, in, in_pitch, tmp.data(), tmp.pitch, n, scan, (const double *)d_state, d_state,
           (int)nsub, -1, (const int *)d_list);
    launch(k_fir_dec, dim3(cdiv(n, 256), na), dim3(256), s, (const float2 *)tmp.data(), tmp.pitch, out, out_pitch, n, 1,
           (const float *)d_taps, (int)kMaxTaps, (const int *)d_ntaps, 0, 1.0f, (const int *)d_list);
    launch(k_save_tail, dim3(cdiv(tmp.hist, 256), na), dim3(256), s, tmp.data(), tmp.pitch, n, tmp.hist, (const int *)d_list);
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
    if (int rc = a.alloc((int)C, 2, max_n)) return rc;
    if (int rc = b.alloc((int)C, kMaxTaps, max_n)) return rc;
    if (int rc = c.alloc((int)C, 0, max_n)) return rc;
    lp_on = rate >= 150000;  // demod_wfm.cpp:210
    const design::Biquad l = design::biquad_lowpass(75000, 1.0, rate);   // demod_wfm.cpp:164
    const double lpc[5] = {l.b0, l.b1, l.b2, l.a1, l.a2};
    fill_scan_section(lp.sec[0], kBiquadDf2, lpc);
    const double da = 1.0 - std::exp(-1.0 / (rate * 75E-6));            // demod_wfm.cpp:181-183,454
    fill_scan_section(dn.sec[0], kOnePoleAvg, &da);
    const design::Biquad br = design::biquad_notch(19000.0, 5, rate);    // demod_wfm.cpp:178
    const double brc[5] = {br.b0, br.b1, br.b2, br.a1, br.a2};
    fill_scan_section(dn.sec[1], kBiquadDf2, brc);
    warm_lp = scan_warm_subchunks(lp.sec, 1, 1e-13);
    warm_dn = scan_warm_subchunks(dn.sec, 2, 1e-13);
    const std::vector<double> h = design::fir_lowpass(0, 1.0, 60.0, 15000.0, 1.4 * 15000.0, rate);  // demod_wfm.cpp:175
    ntaps = (int)h.size();
    std::vector<float> hf(kMaxTaps, 0.f);
    for (size_t i = 0; i < h.size(); i++) hf[i] = (float)h[i];
    PG_HIP(hipMalloc((void **)&d_taps, sizeof(float) * kMaxTaps));
    PG_HIP(hipMemcpy(d_taps, hf.data(), sizeof(float) * kMaxTaps, hipMemcpyHostToDevice));
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
    void *p[] = {d_taps, d_lp_state[0], d_lp_state[1], d_dn_state[0], d_dn_state[1]};
    for (void *q : p) if (q) (void)hipFree(q);
    d_taps = nullptr;
    d_lp_state[0] = d_lp_state[1] = d_dn_state[0] = d_dn_state[1] = nullptr;
}
int WfmCore::run(hipStream_t s, const float2 *in, long long in_pitch, float2 *out, long long out_pitch, long long n)
{
    if (n > a.cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
    if (n < b.hist) return fail(PEBBLEGPU_E_SIZE, "WFM demod needs at least %d samples per call", b.hist);
    const long long nsub = (n + kSub - 1) / kSub;
    const int spb = 2;  // output sub-chunks per workgroup in chunk-parallel mode (each also re-runs warm_* before them)
    if (lp_on) {
        const unsigned gx = warm_lp < 0 ? 1u : cdiv(nsub, spb);
        launch(k_iir_scan<0, 1>, dim3(gx, C), dim3(64), s, in, in_pitch, a.data(), a.pitch, n, lp, (const double *)d_lp_state[parity],
               d_lp_state[parity ^ 1], warm_lp < 0 ? (int)nsub : spb, warm_lp, (const int *)nullptr);
    } else {
        launch(k_copy, dim3(cdiv(n, 256), C), dim3(256), s, in, in_pitch, a.data(), a.pitch, n);
    }
    launch(k_discrim, dim3(cdiv(n, 256), C), dim3(256), s, (const float2 *)a.data(), a.pitch, b.data(), b.pitch, n, 0.25f);  // FMDEMOD_GAIN
    launch(k_fir_dec, dim3(cdiv(n, 256), C), dim3(256), s, (const float2 *)b.data(), b.pitch, c.data(), c.pitch, n, 1, (const float *)d_taps, 0,
           (const int *)nullptr, ntaps, 1.0f, (const int *)nullptr);
    {
        const unsigned gx = warm_dn < 0 ? 1u : cdiv(nsub, spb);
        launch(k_iir_scan<1, 2>, dim3(gx, C), dim3(64), s, (const float2 *)c.data(), c.pitch, out, out_pitch, n, dn,
               (const double *)d_dn_state[parity], d_dn_state[parity ^ 1], warm_dn < 0 ? (int)nsub : spb, warm_dn, (const int *)nullptr);
    }
    parity ^= 1;
    last_n = n;
    PG_HIP(hipGetLastError());
    return 0;
}
void WfmCore::tail_jobs(std::vector<TailJob> &jobs) const
{
    jobs.push_back(TailJob{a.data(), a.pitch, last_n, a.hist, 0});
    jobs.push_back(TailJob{b.data(), b.pitch, last_n, b.hist, 0});
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
    if (nf != 2048 || !(bins == 2048 || bins == 4096 || bins == 8192))
        return fail(PEBBLEGPU_E_UNSUPPORTED, "spectrum needs 2048-sample frames and 2048/4096/8192 bins in this build (asked %u/%u)", nf, bins);
    std::vector<double> w;
    const double cg = design::blackman_harris(nf, w);
    std::vector<float> wf(nf);
    for (uint32_t i = 0; i < nf; i++) wf[i] = (float)w[i];
    PG_HIP(hipMalloc((void **)&d_window, sizeof(float) * nf));
    PG_HIP(hipMemcpy(d_window, wf.data(), sizeof(float) * nf, hipMemcpyHostToDevice));
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
    void *p[] = {d_window, d_btab, d_prev[0], d_prev[1], d_tw_nf};
    for (void *q : p) if (q) (void)hipFree(q);
    d_window = nullptr; d_btab = nullptr; d_prev[0] = d_prev[1] = nullptr; d_tw_nf = nullptr;
}
int SpectrumCore::run(hipStream_t s, const float2 *d_in, long long in_pitch, long long F, float *d_out)
{
    SpectrumParams sp;
    sp.in_pitch = in_pitch;
    sp.n_frames = F;
    const int groups = 4 / (int)(bins / nf);  // wave groups (frames in flight) per workgroup
    long long G = (F * (long long)S) / (1024 * groups);  // aim for ~1024 workgroups; each group recomputes one extra frame
    G = G < 1 ? 1 : (G > 16 ? 16 : G);
    sp.frames_per_group = (int)G;
    sp.scale = scale;
    sp.out_pitch = F * (long long)bins;
    const dim3 grid(cdiv(F, G * groups), S), block(256);
    const float *pin = d_prev[parity];
    float *pout = d_prev[parity ^ 1];
    if (bins == 2048) launch(k_spectrum<1>, grid, block, s, d_in, d_out, (const float *)d_window, (const float2 *)d_btab, (const float2 *)d_tw_nf, pin, pout, sp);
    else if (bins == 4096) launch(k_spectrum<2>, grid, block, s, d_in, d_out, (const float *)d_window, (const float2 *)d_btab, (const float2 *)d_tw_nf, pin, pout, sp);
    else launch(k_spectrum<4>, grid, block, s, d_in, d_out, (const float *)d_window, (const float2 *)d_btab, (const float2 *)d_tw_nf, pin, pout, sp);
    parity ^= 1;
    PG_HIP(hipGetLastError());
    return 0;
}

}  // namespace pg
