// steps.hip -- stand-alone process steps with the reference's per-class call shapes (host doubles in/out).
// Each wraps one device core with C = 1; see include/pebblegpu.h for the interface each entry point replaces.
#include <cmath>
#include <new>
#include "receiver.h"

namespace pg {
int run_mixer(hipStream_t s, const float2 *d_in, float2 *d_out, long long n, const OscBank &osc);

struct StepBase {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<float> hf;   // float staging
    std::vector<double> hd;  // double result buffer (valid until the next call, like ProcessStep::out)
    int open(int dev)
    {
        device = dev;
        PG_HIP(hipSetDevice(device));
        PG_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return 0;
    }
    void close_stream()
    {
        if (stream) {
            (void)hipSetDevice(device);
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
            stream = nullptr;
        }
    }
    int up(float2 *dst, const double *src, size_t n)
    {
        hf.resize(n * 2);
        for (size_t i = 0; i < n * 2; i++) hf[i] = (float)src[i];
        PG_HIP(hipMemcpyAsync(dst, hf.data(), sizeof(float2) * n, hipMemcpyHostToDevice, stream));
        PG_HIP(hipStreamSynchronize(stream));
        return 0;
    }
    int down(double *dst, const float2 *src, size_t n)
    {
        hf.resize(n * 2);
        PG_HIP(hipStreamSynchronize(stream));
        PG_HIP(hipMemcpy(hf.data(), src, sizeof(float2) * n, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n * 2; i++) dst[i] = (double)hf[i];
        return 0;
    }
};
}  // namespace pg

using pg::fail;

struct pebblegpu_mixer : pg::StepBase {
    uint32_t n = 0;
    pg::OscBank osc;
    float2 *d_in = nullptr, *d_out = nullptr;
};
struct pebblegpu_decimator : pg::StepBase {
    uint32_t fs = 0, cap = 0;
    bool built = false;
    pg::OscBank osc;  // frequency 0: the oscillator is bypassed, the fused kernel only decimates
    pg::DecimCore dec;
    float2 *d_in = nullptr;
};
// CDownConvert (pebblelib/downconvert.cpp): quadrature-oscillator mixer + a cascade of decimate-by-2 stages
struct pebblegpu_downconvert : pg::StepBase {
    uint32_t cap = 0;
    double in_rate = 100000.0, max_bw = 10000.0, out_rate = 0, nco_freq = 0, cw_offset = 0;  // ctor values, downconvert.cpp:63-77
    bool rate_set = false;
    pg::OscBank osc;
    std::vector<int> stages;          // indices into the stage table
    std::vector<pg::HistBuf> bufs;    // bufs[j]: input of stage j (head-room = its look-back); bufs[nst]: the result
    std::vector<float *> d_taps;      // [stage] the response it applies, oldest sample first
    std::vector<int> ntaps;
    float2 *d_in = nullptr;
    void drop_chain()
    {
        for (auto &b : bufs) b.release();
        for (float *t : d_taps) if (t) (void)hipFree(t);
        bufs.clear(); d_taps.clear(); ntaps.clear(); stages.clear();
    }
};
struct pebblegpu_fastfir : pg::StepBase {
    pg::FastFirCore ff;
    pg::HistBuf in;
    float2 *d_out = nullptr;
    long long cap = 0;
    std::vector<double> pend;  // samples waiting for a whole block (m_InBufInPos - (FIR-1))
    double lo = -1.0, hi = 1.0, offset = 1.0, rate = 1.0;  // fastfir.cpp:141-144
};
struct pebblegpu_demod : pg::StepBase {
    int mode = PEBBLEGPU_DM_AM;
    uint32_t cap = 0;
    pg::AmCore am;
    pg::PllCore nfm, sam;
    pg::WfmCore wfm;
    float2 *d_in = nullptr, *d_out = nullptr;
};
struct pebblegpu_spectrum : pg::StepBase {
    pg::SpectrumCore sp;
    float2 *d_in = nullptr;
    float *d_out = nullptr;
    std::vector<float> hs;
    int path = 0;  // 0: no call yet; 1: the frame-length kernels; 2: the general kernel (their previous-frame amplitudes are laid out differently)
};

extern "C" {

static int step_device(int device)
{
    int n = pebblegpu_device_count();
    if (n <= 0) return fail(PEBBLEGPU_E_NO_DEVICE, "no HIP device visible: libpebblegpu has no CPU path");
    if (device < 0 || device >= n) return fail(PEBBLEGPU_E_INVALID, "device %d out of range", device);
    return 0;
}

// ---------------- Mixer ----------------
int pebblegpu_mixer_create(int device, uint32_t sample_rate, uint32_t buffer_size, pebblegpu_mixer **out)
{
    if (!out || !buffer_size || !sample_rate) return fail(PEBBLEGPU_E_INVALID, "bad argument");
    if (int rc = step_device(device)) return rc;
    pebblegpu_mixer *m = new (std::nothrow) pebblegpu_mixer();
    if (!m) return fail(PEBBLEGPU_E_INVALID, "out of host memory");
    m->n = buffer_size;
    int rc = m->open(device);
    if (!rc) rc = m->osc.init(1, (double)sample_rate);
    if (!rc && hipMalloc((void **)&m->d_in, sizeof(float2) * buffer_size) != hipSuccess) rc = fail(PEBBLEGPU_E_HIP, "hipMalloc failed");
    if (!rc && hipMalloc((void **)&m->d_out, sizeof(float2) * buffer_size) != hipSuccess) rc = fail(PEBBLEGPU_E_HIP, "hipMalloc failed");
    if (rc) { pebblegpu_mixer_destroy(m); return rc; }
    m->hd.resize((size_t)buffer_size * 2);
    m->osc.retune(0, 0.0);  // ctor: setFrequency(0), mixer.cpp:13
    *out = m;
    return 0;
}
int pebblegpu_mixer_destroy(pebblegpu_mixer *m)
{
    if (!m) return 0;
    m->close_stream();
    m->osc.release();
    if (m->d_in) (void)hipFree(m->d_in);
    if (m->d_out) (void)hipFree(m->d_out);
    delete m;
    return 0;
}
int pebblegpu_mixer_set_frequency(pebblegpu_mixer *m, double f)
{
    if (!m) return fail(PEBBLEGPU_E_INVALID, "null handle");
    m->osc.retune(0, f);
    return 0;
}
int pebblegpu_mixer_process(pebblegpu_mixer *m, const double *in, const double **out)
{
    if (!m || !in || !out) return fail(PEBBLEGPU_E_INVALID, "null argument");
    if ((-m->osc.ctl[0].freq) == 0) { *out = in; return 0; }  // mixer.cpp:51-53
    PG_HIP(hipSetDevice(m->device));
    if (int rc = m->osc.upload(m->stream)) return rc;
    if (int rc = m->up(m->d_in, in, m->n)) return rc;
    if (int rc = pg::run_mixer(m->stream, m->d_in, m->d_out, (long long)m->n, m->osc)) return rc;
    if (int rc = m->down(m->hd.data(), m->d_out, m->n)) return rc;
    m->osc.advance(m->n);
    *out = m->hd.data();
    return 0;
}

// ---------------- Decimator ----------------
int pebblegpu_decimator_create(int device, uint32_t sample_rate, uint32_t buffer_size, pebblegpu_decimator **out)
{
    if (!out || !buffer_size || !sample_rate) return fail(PEBBLEGPU_E_INVALID, "bad argument");
    if (int rc = step_device(device)) return rc;
    pebblegpu_decimator *d = new (std::nothrow) pebblegpu_decimator();
    if (!d) return fail(PEBBLEGPU_E_INVALID, "out of host memory");
    d->fs = sample_rate;
    d->cap = buffer_size;
    int rc = d->open(device);
    if (!rc) rc = d->osc.init(1, (double)sample_rate);
    if (!rc && hipMalloc((void **)&d->d_in, sizeof(float2) * buffer_size) != hipSuccess) rc = fail(PEBBLEGPU_E_HIP, "hipMalloc failed");
    if (rc) { pebblegpu_decimator_destroy(d); return rc; }
    d->osc.retune(0, 0.0);
    *out = d;
    return 0;
}
int pebblegpu_decimator_destroy(pebblegpu_decimator *d)
{
    if (!d) return 0;
    d->close_stream();
    d->osc.release();
    d->dec.release();
    if (d->d_in) (void)hipFree(d->d_in);
    delete d;
    return 0;
}
int pebblegpu_decimator_build_chain(pebblegpu_decimator *d, uint32_t sample_rate_in, uint32_t protect_bw, uint32_t sample_rate_out,
                                    float *achieved_rate)
{
    if (!d) return fail(PEBBLEGPU_E_INVALID, "null handle");
    PG_HIP(hipSetDevice(d->device));
    const pg::design::Chain c = pg::design::build_chain(sample_rate_in, protect_bw, sample_rate_out);
    if (achieved_rate) *achieved_rate = c.rate;
    d->built = false;
    if (c.stages.empty()) { d->dec.release(); d->dec.chain = c; return 0; }  // "No decimation, just return" (decimator.cpp:155-160)
    if (c.stages.size() > (size_t)pg::kMaxStages) return fail(PEBBLEGPU_E_UNSUPPORTED, "chain too long");
    if (int rc = d->dec.init(1, c, (long long)d->cap, 0, 1.0f)) return rc;
    d->built = true;
    return 0;
}
int pebblegpu_decimator_dec_by2_stages(const pebblegpu_decimator *d, uint32_t *stages)
{
    if (!d || !stages) return fail(PEBBLEGPU_E_INVALID, "null argument");
    *stages = d->dec.chain.dec_by2;
    return 0;
}
int pebblegpu_decimator_process(pebblegpu_decimator *d, const double *in, double *out, uint32_t n, uint32_t *n_out)
{
    if (!d || !in || !out || !n_out) return fail(PEBBLEGPU_E_INVALID, "null argument");
    if (!d->built) {  // empty chain: copy through
        memcpy(out, in, sizeof(double) * 2 * n);
        *n_out = n;
        return 0;
    }
    if (n > d->cap) return fail(PEBBLEGPU_E_SIZE, "%u samples exceed bufferSize %u", n, d->cap);
    PG_HIP(hipSetDevice(d->device));
    if (int rc = d->osc.upload(d->stream)) return rc;
    if (int rc = d->up(d->d_in, in, n)) return rc;
    if (int rc = d->dec.run(d->stream, d->d_in, (long long)n, false, (long long)n, d->osc)) return rc;
    const long long no = d->dec.out_len();
    if (int rc = d->down(out, d->dec.out().data(), (size_t)no)) return rc;
    {
        std::vector<pg::TailJob> jobs;
        d->dec.tail_jobs(jobs);
        if (int rc = pg::run_save_tails(d->stream, jobs, 1)) return rc;
    }
    PG_HIP(hipStreamSynchronize(d->stream));
    *n_out = (uint32_t)no;
    return 0;
}

// ---------------- CDownConvert ----------------
int pebblegpu_downconvert_create(int device, uint32_t max_in_length, pebblegpu_downconvert **out)
{
    if (!out || !max_in_length) return fail(PEBBLEGPU_E_INVALID, "bad argument");
    if (int rc = step_device(device)) return rc;
    pebblegpu_downconvert *d = new (std::nothrow) pebblegpu_downconvert();
    if (!d) return fail(PEBBLEGPU_E_INVALID, "out of host memory");
    d->cap = max_in_length;
    int rc = d->open(device);
    if (!rc) rc = d->osc.init(1, d->in_rate);
    if (!rc && hipMalloc((void **)&d->d_in, sizeof(float2) * max_in_length) != hipSuccess) rc = fail(PEBBLEGPU_E_HIP, "hipMalloc failed");
    if (rc) { pebblegpu_downconvert_destroy(d); return rc; }
    d->osc.force_mix = true;       // no "frequency 0" exit: ProcessData always multiplies (downconvert.cpp:283-308)
    d->osc.retune(0, 0.0);         // m_Osc1 = (1, 0), m_NcoFreq = 0 (ctor)
    *out = d;
    return 0;
}
int pebblegpu_downconvert_destroy(pebblegpu_downconvert *d)
{
    if (!d) return 0;
    d->close_stream();
    d->osc.release();
    d->drop_chain();
    if (d->d_in) (void)hipFree(d->d_in);
    delete d;
    return 0;
}
static int downconvert_tune(pebblegpu_downconvert *d, double f)  // SetFrequency, downconvert.cpp:100-112
{
    d->nco_freq = -f + d->cw_offset;
    d->osc.retune_keep(0, -d->nco_freq);  // (OscBank negates as Mixer does: inc = -f / Fs)
    return 0;
}
int pebblegpu_downconvert_set_frequency(pebblegpu_downconvert *d, double f)
{
    if (!d) return fail(PEBBLEGPU_E_INVALID, "null handle");
    return downconvert_tune(d, f);
}
int pebblegpu_downconvert_set_cw_offset(pebblegpu_downconvert *d, double offset)
{
    if (!d) return fail(PEBBLEGPU_E_INVALID, "null handle");
    d->cw_offset = offset;  // SetCwOffset (downconvert.h:34): takes effect at the next SetFrequency
    return 0;
}
int pebblegpu_downconvert_set_data_rate(pebblegpu_downconvert *d, double in_rate, double max_bw, int simple, double *out_rate)
{
    if (!d || !(in_rate > 0) || !(max_bw > 0)) return fail(PEBBLEGPU_E_INVALID, "bad argument");
    PG_HIP(hipSetDevice(d->device));
    if (!d->rate_set || d->in_rate != in_rate || d->max_bw != max_bw) {  // :143-144 / :217-218
        const pg::design::DcChain c = pg::design::downconvert_chain(in_rate, max_bw, simple != 0);
        // MAX_DECSTAGES 10, "one more than max" (downconvert.h:23): a tenth stage overwrites the list's terminating NULL in the reference
        if (c.stages.size() > 9) return fail(PEBBLEGPU_E_UNSUPPORTED, "%zu decimate-by-2 stages: CDownConvert holds nine", c.stages.size());
        PG_HIP(hipStreamSynchronize(d->stream));
        d->drop_chain();
        d->in_rate = in_rate;
        d->max_bw = max_bw;
        d->out_rate = c.out_rate;
        d->stages = c.stages;
        long long len = d->cap;
        for (size_t j = 0; j <= c.stages.size(); j++) {
            int look = 4;
            if (j < c.stages.size()) {
                const std::vector<double> h = pg::design::downconvert_stage_response(c.stages[j]);
                look = (int)h.size() + 1;
                std::vector<float> hf(h.begin(), h.end());
                float *t = nullptr;
                PG_HIP(hipMalloc((void **)&t, sizeof(float) * hf.size()));
                d->d_taps.push_back(t);
                d->ntaps.push_back((int)hf.size());
                PG_HIP(hipMemcpy(t, hf.data(), sizeof(float) * hf.size(), hipMemcpyHostToDevice));
            }
            pg::HistBuf b;
            if (int rc = b.alloc(1, look, len + 2)) return rc;
            d->bufs.push_back(b);
            len /= 2;
        }
        // the oscillator keeps its phasor; its increment follows the new input rate
        d->osc.fs = in_rate;
        d->rate_set = true;
        // SetFrequency(m_NcoFreq), :205 / :232, as written: the STORED value (negated, offset included) is negated once more
        downconvert_tune(d, d->nco_freq);
    }
    if (out_rate) *out_rate = d->out_rate;
    return 0;
}
int pebblegpu_downconvert_stages(const pebblegpu_downconvert *d, uint32_t *n_stages, uint32_t *taps, uint32_t taps_cap)
{
    if (!d || !n_stages) return fail(PEBBLEGPU_E_INVALID, "null argument");
    *n_stages = (uint32_t)d->stages.size();
    for (size_t j = 0; taps && j < d->stages.size() && j < taps_cap; j++) taps[j] = (uint32_t)pg::design::downconvert_stage_taps(d->stages[j]);
    return 0;
}
// queues mixer + stages on the step's stream; *d_out / *n_out: the result row (device, valid until the next call)
static int downconvert_run(pebblegpu_downconvert *d, const float2 *d_x, uint32_t n, const float2 **d_out, uint32_t *n_out)
{
    if (!d->rate_set) return fail(PEBBLEGPU_E_INVALID, "SetDataRate first");
    const size_t nst = d->stages.size();
    if (n > d->cap) return fail(PEBBLEGPU_E_SIZE, "%u samples exceed the %u this object was created for", n, d->cap);
    if (n == 0 || (n & ((1u << nst) - 1)) != 0) return fail(PEBBLEGPU_E_SIZE, "InLength must be a multiple of 2^%zu (downconvert.cpp:246-247)", nst);
    if (int rc = d->osc.upload(d->stream)) return rc;
    if (int rc = pg::run_mixer(d->stream, d_x, d->bufs[0].data(), (long long)n, d->osc)) return rc;
    std::vector<pg::TailJob> jobs;
    long long len = n;
    for (size_t j = 0; j < nst; j++) {
        // the CIC3's newest tap is the pair's ODD sample: the same strided FIR read one sample later
        const float2 *src = d->bufs[j].data() + (d->stages[j] == 0 ? 1 : 0);
        if (int rc = pg::run_fir_dec(d->stream, src, d->bufs[j].pitch, d->bufs[j + 1].data(), d->bufs[j + 1].pitch, len / 2, 2, d->d_taps[j], d->ntaps[j], 1)) return rc;
        jobs.push_back(pg::TailJob{d->bufs[j].data(), d->bufs[j].pitch, len, d->bufs[j].hist, 0, nullptr, 0});
        len /= 2;
    }
    if (int rc = pg::run_save_tails(d->stream, jobs, 1)) return rc;
    d->osc.advance(n);
    *d_out = d->bufs[nst].data();
    *n_out = (uint32_t)len;
    return 0;
}
int pebblegpu_downconvert_process(pebblegpu_downconvert *d, uint32_t in_length, const double *in, double *out, uint32_t *n_out)
{
    if (!d || !in || !out || !n_out) return fail(PEBBLEGPU_E_INVALID, "null argument");
    PG_HIP(hipSetDevice(d->device));
    if (in_length > d->cap) return fail(PEBBLEGPU_E_SIZE, "%u samples exceed the %u this object was created for", in_length, d->cap);
    if (int rc = d->up(d->d_in, in, in_length)) return rc;
    const float2 *res = nullptr;
    uint32_t no = 0;
    if (int rc = downconvert_run(d, d->d_in, in_length, &res, &no)) return rc;
    if (int rc = d->down(out, res, no)) return rc;
    PG_HIP(hipStreamSynchronize(d->stream));
    *n_out = no;
    return 0;
}
int pebblegpu_downconvert_process_device(pebblegpu_downconvert *d, const void *d_iq, uint32_t in_length, const void **d_out, uint32_t *n_out)
{
    if (!d || !d_iq || !d_out || !n_out) return fail(PEBBLEGPU_E_INVALID, "null argument");
    PG_HIP(hipSetDevice(d->device));
    const float2 *res = nullptr;
    if (int rc = downconvert_run(d, (const float2 *)d_iq, in_length, &res, n_out)) return rc;
    *d_out = res;
    return 0;
}
int pebblegpu_downconvert_synchronize(pebblegpu_downconvert *d)
{
    if (!d) return fail(PEBBLEGPU_E_INVALID, "null handle");
    PG_HIP(hipSetDevice(d->device));
    PG_HIP(hipStreamSynchronize(d->stream));
    return 0;
}

// ---------------- CFastFIR ----------------
int pebblegpu_fastfir_create(int device, uint32_t fft_size, uint32_t fir_size, pebblegpu_fastfir **out)
{
    if (!out) return fail(PEBBLEGPU_E_INVALID, "null argument");
    if (int rc = step_device(device)) return rc;
    pebblegpu_fastfir *f = new (std::nothrow) pebblegpu_fastfir();
    if (!f) return fail(PEBBLEGPU_E_INVALID, "out of host memory");
    int rc = f->open(device);
    if (!rc) rc = f->ff.init(1, fft_size ? fft_size : 2048, fir_size ? fir_size : 1025);
    if (!rc) {
        f->cap = 64 * f->ff.block_len();
        rc = f->in.alloc(1, (int)f->ff.taps - 1, f->cap);
    }
    if (!rc && hipMalloc((void **)&f->d_out, sizeof(float2) * (size_t)f->cap) != hipSuccess) rc = fail(PEBBLEGPU_E_HIP, "hipMalloc failed");
    if (rc) { pebblegpu_fastfir_destroy(f); return rc; }
    *out = f;
    return 0;
}
int pebblegpu_fastfir_destroy(pebblegpu_fastfir *f)
{
    if (!f) return 0;
    f->close_stream();
    f->ff.release();
    f->in.release();
    if (f->d_out) (void)hipFree(f->d_out);
    delete f;
    return 0;
}
int pebblegpu_fastfir_setup(pebblegpu_fastfir *f, double lo, double hi, double offset, double rate)
{
    if (!f) return fail(PEBBLEGPU_E_INVALID, "null handle");
    if (lo == f->lo && hi == f->hi && offset == f->offset && rate == f->rate) return 0;  // fastfir.cpp:195-199
    f->lo = lo; f->hi = hi; f->offset = offset; f->rate = rate;
    PG_HIP(hipSetDevice(f->device));
    bool ok = false;
    if (int rc = f->ff.design(f->stream, 0, lo, hi, offset, rate, &ok)) return rc;
    if (!ok) return fail(PEBBLEGPU_E_FILTER_PARAM, "Filter Parameter error");
    return 0;
}
int pebblegpu_fastfir_process(pebblegpu_fastfir *f, int n, const double *in, double *out, int *n_out)
{
    if (!f || !n_out || (n > 0 && (!in || !out))) return fail(PEBBLEGPU_E_INVALID, "null argument");
    *n_out = 0;
    if (n <= 0) return 0;
    PG_HIP(hipSetDevice(f->device));
    f->pend.insert(f->pend.end(), in, in + (size_t)n * 2);
    const long long L = f->ff.block_len();
    long long avail = (long long)(f->pend.size() / 2), done = 0;
    while (avail - done >= L) {
        long long take = ((avail - done) / L) * L;
        if (take > f->cap) take = f->cap;
        if (int rc = f->up(f->in.data(), f->pend.data() + done * 2, (size_t)take)) return rc;
        if (int rc = f->ff.run(f->stream, f->in, take, f->d_out, take)) return rc;
        if (int rc = f->down(out + (size_t)(*n_out) * 2, f->d_out, (size_t)take)) return rc;
        // overlap for the next block = last taps-1 inputs (m_pFFTOverlapBuf)
        const int hist = f->in.hist;
        if (take >= hist) {
            PG_HIP(hipMemcpyAsync(f->in.data() - hist, f->in.data() + take - hist, sizeof(float2) * hist, hipMemcpyDeviceToDevice, f->stream));
        } else {  // FIR longer than one block: shift what is kept, then append
            PG_HIP(hipMemcpyAsync(f->in.data() - hist, f->in.data() - hist + take, sizeof(float2) * (hist - take), hipMemcpyDeviceToDevice, f->stream));
            PG_HIP(hipMemcpyAsync(f->in.data() - take, f->in.data(), sizeof(float2) * take, hipMemcpyDeviceToDevice, f->stream));
        }
        PG_HIP(hipStreamSynchronize(f->stream));
        done += take;
        *n_out += (int)take;
    }
    f->pend.erase(f->pend.begin(), f->pend.begin() + (size_t)done * 2);
    return 0;
}

// ---------------- Demod ----------------
int pebblegpu_demod_create(int device, uint32_t sample_rate, uint32_t wfm_sample_rate, uint32_t buffer_size, pebblegpu_demod **out)
{
    if (!out || !buffer_size) return fail(PEBBLEGPU_E_INVALID, "bad argument");
    if (int rc = step_device(device)) return rc;
    pebblegpu_demod *d = new (std::nothrow) pebblegpu_demod();
    if (!d) return fail(PEBBLEGPU_E_INVALID, "out of host memory");
    d->cap = buffer_size;
    int rc = d->open(device);
    if (!rc && sample_rate) rc = d->am.init(1, (double)sample_rate, buffer_size);
    if (!rc && sample_rate) {
        rc = d->am.set_bandwidth(d->stream, 0, 16000);  // Demod_AM ctor, demod_am.cpp:9
        if (!rc) rc = d->am.set_list(d->stream, std::vector<int>(1, 0));
    }
    if (!rc && sample_rate) rc = d->sam.init(1, (double)sample_rate, buffer_size, 1);   // Demod_SAM, demod.cpp:63
    if (!rc && sample_rate) rc = d->nfm.init(1, (double)sample_rate, buffer_size, 0);   // Demod_NFM, demod.cpp:64
    if (!rc && sample_rate) {
        rc = d->sam.set_list(d->stream, std::vector<int>(1, 0));
        if (!rc) rc = d->nfm.set_list(d->stream, std::vector<int>(1, 0));
    }
    if (!rc && wfm_sample_rate) rc = d->wfm.init(1, (double)wfm_sample_rate, buffer_size);
    d->wfm.stereo_block = 0;  // a processBlock call is one block
    if (!rc && hipMalloc((void **)&d->d_in, sizeof(float2) * buffer_size) != hipSuccess) rc = fail(PEBBLEGPU_E_HIP, "hipMalloc failed");
    if (!rc && hipMalloc((void **)&d->d_out, sizeof(float2) * buffer_size) != hipSuccess) rc = fail(PEBBLEGPU_E_HIP, "hipMalloc failed");
    if (rc) { pebblegpu_demod_destroy(d); return rc; }
    d->hd.resize((size_t)buffer_size * 2);
    *out = d;
    return 0;
}
int pebblegpu_demod_destroy(pebblegpu_demod *d)
{
    if (!d) return 0;
    d->close_stream();
    d->am.release();
    d->nfm.release();
    d->sam.release();
    d->wfm.release();
    if (d->d_in) (void)hipFree(d->d_in);
    if (d->d_out) (void)hipFree(d->d_out);
    delete d;
    return 0;
}
int pebblegpu_demod_set_mode(pebblegpu_demod *d, int mode)
{
    if (!d) return fail(PEBBLEGPU_E_INVALID, "null handle");
    if (mode < 0 || mode > PEBBLEGPU_DM_NONE) return fail(PEBBLEGPU_E_INVALID, "bad mode %d", mode);
    if ((mode == PEBBLEGPU_DM_AM || mode == PEBBLEGPU_DM_SAM || mode == PEBBLEGPU_DM_FMN) && d->am.C == 0) return fail(PEBBLEGPU_E_INVALID, "created without a narrow sample rate");
    if ((mode == PEBBLEGPU_DM_FMM || mode == PEBBLEGPU_DM_FMS) && d->wfm.C == 0) return fail(PEBBLEGPU_E_INVALID, "created without a WFM sample rate");
    if (d->wfm.C) {  // dmFMS: what processDataStereo delivers once its pilot PLL has dropped out (see pebblegpu.h)
        if (int rc = d->wfm.set_stereo(0, mode == PEBBLEGPU_DM_FMS)) return rc;
    }
    d->mode = mode;
    return 0;
}
int pebblegpu_demod_set_bandwidth(pebblegpu_demod *d, double bw)
{
    if (!d) return fail(PEBBLEGPU_E_INVALID, "null handle");
    if (d->mode != PEBBLEGPU_DM_AM) return 0;  // demod.cpp:230-239
    PG_HIP(hipSetDevice(d->device));
    return d->am.set_bandwidth(d->stream, 0, bw);
}
int pebblegpu_demod_process(pebblegpu_demod *d, const double *in, int n, const double **out)
{
    if (!d || !in || !out || n <= 0) return fail(PEBBLEGPU_E_INVALID, "bad argument");
    if (d->mode != PEBBLEGPU_DM_AM && d->mode != PEBBLEGPU_DM_FMM && d->mode != PEBBLEGPU_DM_FMS && d->mode != PEBBLEGPU_DM_SAM && d->mode != PEBBLEGPU_DM_FMN) {
        *out = in;  // demod.cpp:127-138
        return 0;
    }
    if ((uint32_t)n > d->cap) return fail(PEBBLEGPU_E_SIZE, "%d samples exceed bufferSize %u", n, d->cap);
    PG_HIP(hipSetDevice(d->device));
    if (int rc = d->up(d->d_in, in, (size_t)n)) return rc;
    int rc;
    if (d->mode == PEBBLEGPU_DM_AM) rc = d->am.run(d->stream, d->d_in, n, d->d_out, n, n);
    else if (d->mode == PEBBLEGPU_DM_SAM) rc = d->sam.run(d->stream, d->d_in, n, d->d_out, n, n);
    else if (d->mode == PEBBLEGPU_DM_FMN) rc = d->nfm.run(d->stream, d->d_in, n, d->d_out, n, n);
    else {
        rc = d->wfm.run(d->stream, d->d_in, n, d->d_out, n, n);
        if (!rc) {
            std::vector<pg::TailJob> jobs;
            d->wfm.tail_jobs(jobs);
            rc = pg::run_save_tails(d->stream, jobs, 1);
        }
    }
    if (rc) return rc;
    if (int rc2 = d->down(d->hd.data(), d->d_out, (size_t)n)) return rc2;
    *out = d->hd.data();
    return 0;
}
int pebblegpu_demod_rds_groups(pebblegpu_demod *d, pebblegpu_rds_group *groups, uint8_t *changed, uint32_t cap, uint32_t *n)
{
    if (!d || !n) return fail(PEBBLEGPU_E_INVALID, "null argument");
    *n = 0;
    if (d->wfm.C == 0) return fail(PEBBLEGPU_E_INVALID, "created without a WFM sample rate");
    PG_HIP(hipSetDevice(d->device));
    return d->wfm.rds.groups(d->stream, 0, reinterpret_cast<pg::RdsGroup *>(groups), changed, cap, n);
}
int pebblegpu_demod_stereo_lock(pebblegpu_demod *d, int *pilot_lock, int *changed)
{
    if (!d) return fail(PEBBLEGPU_E_INVALID, "null handle");
    if (d->wfm.C == 0) return fail(PEBBLEGPU_E_INVALID, "created without a WFM sample rate");
    PG_HIP(hipSetDevice(d->device));
    return d->wfm.stereo_lock(d->stream, 0, pilot_lock, changed);
}
int pebblegpu_demod_rds_signal(pebblegpu_demod *d, double *data, uint32_t cap, uint32_t *n)
{
    if (!d || !n) return fail(PEBBLEGPU_E_INVALID, "null argument");
    *n = 0;
    if (d->wfm.C == 0) return fail(PEBBLEGPU_E_INVALID, "created without a WFM sample rate");
    PG_HIP(hipSetDevice(d->device));
    return d->wfm.rds.signal(d->stream, 0, data, cap, n);
}

// ---------------- Spectrum ----------------
int pebblegpu_spectrum_create(int device, uint32_t fft_size, double sample_rate, uint32_t samples_per_buffer, pebblegpu_spectrum **out)
{
    (void)sample_rate;  // only feeds the bin width in the reference (fft.cpp:81)
    if (!out || fft_size == 0) return fail(PEBBLEGPU_E_INVALID, "bad argument");  // "if (_fftSize == 0) return; //Error"
    if (int rc = step_device(device)) return rc;
    pebblegpu_spectrum *s = new (std::nothrow) pebblegpu_spectrum();
    if (!s) return fail(PEBBLEGPU_E_INVALID, "out of host memory");
    int rc = s->open(device);
    if (!rc) rc = s->sp.init(1, samples_per_buffer, fft_size);
    if (!rc && hipMalloc((void **)&s->d_in, sizeof(float2) * samples_per_buffer) != hipSuccess) rc = fail(PEBBLEGPU_E_HIP, "hipMalloc failed");
    if (!rc && hipMalloc((void **)&s->d_out, sizeof(float) * s->sp.bins) != hipSuccess) rc = fail(PEBBLEGPU_E_HIP, "hipMalloc failed");
    if (rc) { pebblegpu_spectrum_destroy(s); return rc; }
    *out = s;
    return 0;
}
int pebblegpu_spectrum_destroy(pebblegpu_spectrum *s)
{
    if (!s) return 0;
    s->close_stream();
    s->sp.release();
    if (s->d_in) (void)hipFree(s->d_in);
    if (s->d_out) (void)hipFree(s->d_out);
    delete s;
    return 0;
}
int pebblegpu_spectrum_bins(const pebblegpu_spectrum *s, uint32_t *bins)
{
    if (!s || !bins) return fail(PEBBLEGPU_E_INVALID, "null argument");
    *bins = s->sp.bins;
    return 0;
}
int pebblegpu_spectrum_process(pebblegpu_spectrum *s, const double *in, int n, double *out_db, int *overload)
{
    if (!s || !in || !out_db) return fail(PEBBLEGPU_E_INVALID, "null argument");
    if (n <= 0 || (uint32_t)n > s->sp.nf)
        return fail(PEBBLEGPU_E_SIZE, "numSamples %d is not in 1..samplesPerBuffer (%u)", n, s->sp.nf);
    PG_HIP(hipSetDevice(s->device));
    int ov = 0;  // m_isOverload: any |re| or |im| above m_overLimit = 0.9 (fft.cpp:137-140), flagged on the host copy
    for (int i = 0; i < 2 * n; i++) if (std::fabs(in[i]) > 0.9) { ov = 1; break; }
    if (int rc = s->up(s->d_in, in, (size_t)n)) return rc;
    const int path = ((uint32_t)n == s->sp.nf && !s->sp.any) ? 1 : 2;
    if (s->path && s->path != path)
        return fail(PEBBLEGPU_E_UNSUPPORTED, "one spectrum object takes either whole buffers of %u samples or shorter ones (its previous-frame average is kept per kernel family): create a second object", s->sp.nf);
    s->path = path;
    if ((uint32_t)n == s->sp.nf) {
        if (int rc = s->sp.run(s->stream, s->d_in, n, 1, s->d_out, nullptr, nullptr, true)) return rc;  // (a step on its own stream: nothing beside it)
    } else {
        // fewer samples than samplesPerBuffer: copied, zero-padded, not windowed (fft.cpp:129-157)
        if (int rc = s->sp.run_any(s->stream, s->d_in, n, 1, s->d_out, n, false)) return rc;
    }
    PG_HIP(hipStreamSynchronize(s->stream));
    s->hs.resize(s->sp.bins);
    PG_HIP(hipMemcpy(s->hs.data(), s->d_out, sizeof(float) * s->sp.bins, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < s->sp.bins; i++) out_db[i] = (double)s->hs[i];
    if (overload) *overload = ov;
    return 0;
}

}  // extern "C"
