/*
 * pebble_oracle.c -- CPU restatement (fp64, scalar C) of PebbleSDR's per-frame IQ receive chain.
 * TEST INFRASTRUCTURE ONLY -- see pebble_oracle.h for the rules and the parity-pinning status.
 * Written from scratch from a reading of the reference; each function cites what it restates.
 */
#include "pebble_oracle.h"
#include "hb_taps.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* pebblelib/cpx.h:16-19 */
#define PO_PI 3.14159265358979323846264338328
#define PO_TWOPI 6.28318530717958647692528676656

/* ------------------------------------------------------------------------------------------------
 * Mixer -- pebblelib/mixer.cpp
 * ---------------------------------------------------------------------------------------------- */
void po_mixer_init(po_mixer *m, double fs)
{
    memset(m, 0, sizeof(*m));
    m->fs = fs;
    po_mixer_set_frequency(m, 0.0); /* ctor: setFrequency(0), mixer.cpp:13 */
}

/* mixer.cpp:25-40: sign flip, step phasor, oscillator reset to (1,0) */
void po_mixer_set_frequency(po_mixer *m, double f)
{
    m->freq = -f;
    m->inc = PO_TWOPI * m->freq / m->fs;
    m->osc_cos = cos(m->inc);
    m->osc_sin = sin(m->inc);
    m->last_re = 1.0;
    m->last_im = 0.0;
}

/* mixer.cpp:48-81: rotate, amplitude stabiliser 1.95-|last|^2, out = osc*in (cpx.h:203-206) */
int po_mixer_process(po_mixer *m, const double *in, double *out, uint32_t n)
{
    if (m->freq == 0) return 0;
    for (uint32_t i = 0; i < n; i++) {
        double ore = m->last_re * m->osc_cos - m->last_im * m->osc_sin;
        double oim = m->last_re * m->osc_sin + m->last_im * m->osc_cos;
        double gn = 1.95 - (m->last_re * m->last_re + m->last_im * m->last_im);
        m->last_re = gn * ore;
        m->last_im = gn * oim;
        double xr = in[2 * i], xi = in[2 * i + 1];
        out[2 * i] = (ore * xr) - (oim * xi);
        out[2 * i + 1] = (ore * xi) + (oim * xr);
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * Decimator -- pebblelib/decimator.cpp (vDSP path, m_useVdsp = m_combineStages = true, :9-10)
 * ---------------------------------------------------------------------------------------------- */
#define PO_MAX_STAGES 32
typedef struct {
    int design;       /* index into pebble_hb_designs */
    int ntaps;        /* 0 => CIC3 */
    uint32_t stride;  /* m_decimate */
    /* halfband history: last ntaps-1 inputs (m_lastXVDsp head), split re/im */
    double *hist_re, *hist_im;
    size_t hist_cap;
    /* CIC3 state m_xOdd / m_xEven */
    double xodd_re, xodd_im, xeven_re, xeven_im;
} po_stage;

struct po_decimator {
    po_stage st[PO_MAX_STAGES];
    int nst;
    uint32_t dec_by2;
    float rate; /* m_decimatedSampleRate is float, decimator.h:251 */
    /* ping-pong split-complex work buffers (m_splitComplexIn/Out) */
    double *buf[2][2];
    size_t buf_cap;
};

po_decimator *po_decimator_new(void)
{
    return (po_decimator *)calloc(1, sizeof(po_decimator));
}

static void po_decimator_clear(po_decimator *d)
{
    for (int i = 0; i < d->nst; i++) {
        free(d->st[i].hist_re);
        free(d->st[i].hist_im);
    }
    memset(d->st, 0, sizeof(d->st));
    d->nst = 0;
}

void po_decimator_free(po_decimator *d)
{
    if (!d) return;
    po_decimator_clear(d);
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) free(d->buf[a][b]);
    free(d);
}

/* decimator.cpp:64-149.  The if/else ladder picks the FIRST design (cic3, hb11, hb15, ...) whose
 * rate >= protectBw / wPass holds; identical consecutive picks double the previous stage's stride
 * (m_decimate *= 2, :141) instead of adding a stage. */
double po_decimator_build(po_decimator *d, uint32_t fs_in, uint32_t protect_bw, uint32_t fs_out_min)
{
    po_decimator_clear(d);
    d->rate = (float)fs_in;
    double protect = (double)protect_bw;
    uint32_t min_out = fs_out_min > 0 ? fs_out_min : 15000u; /* minDecimatedSampleRate, decimator.h:245 */
    d->dec_by2 = 0;
    int prev = -1;
    while (d->rate > (float)min_out) {
        int pick = -1;
        for (int k = 0; k < PEBBLE_HB_NDESIGNS; k++) {
            if ((double)d->rate >= protect / pebble_hb_designs[k].wpass) { pick = k; break; }
        }
        if (pick < 0) break; /* "Ran out of filters before minimum sample rate", :125-127 */
        d->dec_by2++;
        if (prev < 0 || d->st[prev].ntaps != pebble_hb_designs[pick].ntaps) {
            if (d->nst >= PO_MAX_STAGES) break;
            po_stage *s = &d->st[d->nst];
            memset(s, 0, sizeof(*s));
            s->design = pick;
            s->ntaps = pebble_hb_designs[pick].ntaps;
            s->stride = 2;
            prev = d->nst++;
        } else {
            d->st[prev].stride *= 2;
        }
        d->rate /= 2;
    }
    return (double)d->rate;
}

int po_decimator_chain_len(const po_decimator *d) { return d->nst; }
uint32_t po_decimator_dec_by2_stages(const po_decimator *d) { return d->dec_by2; }
void po_decimator_stage(const po_decimator *d, int i, int *ntaps, uint32_t *stride, int *design)
{
    if (ntaps) *ntaps = d->st[i].ntaps;
    if (stride) *stride = d->st[i].stride;
    if (design) *design = d->st[i].design;
}

static void po_grow(double **p, size_t *cap, size_t need)
{
    if (*cap >= need) return;
    double *q = (double *)calloc(need, sizeof(double));
    if (*p) { memcpy(q, *p, *cap * sizeof(double)); free(*p); }
    *p = q;
    *cap = need;
}

/* HalfbandFilter::processCIC3, decimator.cpp:719-737 (split-complex twin).  NOTE the merged form:
 * the loop advances by m_decimate but still only reads in[i], in[i+1]. */
static uint32_t po_stage_cic3(po_stage *s, const double *xre, const double *xim, double *yre, double *yim, uint32_t n)
{
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < n; i += s->stride) {
        double er = xre[i], ei = xim[i], orr = xre[i + 1], oi = xim[i + 1];
        yre[cnt] = .125 * (orr + s->xeven_re + 3.0 * (s->xodd_re + er));
        yim[cnt] = .125 * (oi + s->xeven_im + 3.0 * (s->xodd_im + ei));
        s->xodd_re = orr; s->xodd_im = oi;
        s->xeven_re = er; s->xeven_im = ei;
        cnt++;
    }
    return cnt;
}

/* HalfbandFilter::convolveVDsp2, decimator.cpp:593-659.  vDSP_zrdesampD semantics are the
 * pseudo-code quoted at :637-647: C[n] = sum_p A[n*DF+p]*F[p], summed left to right, all taps
 * (zeros included).  x_cap = readable length of the x buffers (for the fallback's over-read). */
static uint32_t po_stage_hb(po_stage *s, const double *xre, const double *xim, size_t x_cap,
                            double *yre, double *yim, uint32_t n)
{
    const double *h = pebble_hb_designs[s->design].h;
    uint32_t hlen = (uint32_t)s->ntaps, dlen = hlen - 1, dec = s->stride;
    size_t need = (size_t)dlen + (n > hlen ? n : hlen) + 1;
    if (s->hist_cap < need) {
        size_t c1 = s->hist_cap, c2 = s->hist_cap;
        po_grow(&s->hist_re, &c1, need);
        po_grow(&s->hist_im, &c2, need);
        s->hist_cap = need;
    }
    if (n < hlen) {
        /* :603-625 fallback: plain sample dropping, and a history refill that indexes x[i] up to
         * hLen-1 (past the n valid samples: the reference reads whatever the ping-pong buffer holds;
         * here that is the stale, initially zero, content of our own ping-pong buffer). */
        uint32_t cnt = 0;
        for (uint32_t i = 0; i < n; i += dec) { yre[cnt] = xre[i]; yim[cnt] = xim[i]; cnt++; }
        uint32_t save = dlen - n;
        for (uint32_t i = 0; i < hlen; i++) {
            if (i < save) { s->hist_re[i] = 0; s->hist_im[i] = 0; }
            else {
                s->hist_re[i] = (i < x_cap) ? xre[i] : 0.0;
                s->hist_im[i] = (i < x_cap) ? xim[i] : 0.0;
            }
        }
        return cnt;
    }
    memcpy(s->hist_re + dlen, xre, n * sizeof(double));
    memcpy(s->hist_im + dlen, xim, n * sizeof(double));
    uint32_t ylen = n / dec;
    for (uint32_t k = 0; k < ylen; k++) {
        double sr = 0, si = 0;
        const double *ar = s->hist_re + (size_t)k * dec, *ai = s->hist_im + (size_t)k * dec;
        for (uint32_t p = 0; p < hlen; p++) { sr += ar[p] * h[p]; si += ai[p] * h[p]; }
        yre[k] = sr; yim[k] = si;
    }
    memmove(s->hist_re, xre + (n - dlen), dlen * sizeof(double));
    memmove(s->hist_im, xim + (n - dlen), dlen * sizeof(double));
    return ylen;
}

/* Decimator::process, decimator.cpp:152-226 (vDSP branch): split, run the chain ping-pong, join.
 * The reference's scratch is capped at maxResultLen=32768 (decimator.h:193); here buffers grow. */
uint32_t po_decimator_process(po_decimator *d, const double *in, double *out, uint32_t n)
{
    if (d->nst == 0) { memcpy(out, in, (size_t)n * 2 * sizeof(double)); return n; }
    size_t need = (size_t)n * 2 + 64;
    if (d->buf_cap < need) {
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) { size_t c = d->buf_cap; po_grow(&d->buf[a][b], &c, need); }
        d->buf_cap = need;
    }
    for (uint32_t i = 0; i < n; i++) { d->buf[0][0][i] = in[2 * i]; d->buf[0][1][i] = in[2 * i + 1]; }
    int cur = 0;
    uint32_t rem = n;
    for (int i = 0; i < d->nst; i++) {
        po_stage *s = &d->st[i];
        if (s->ntaps == 0) rem = po_stage_cic3(s, d->buf[cur][0], d->buf[cur][1], d->buf[cur ^ 1][0], d->buf[cur ^ 1][1], rem);
        else rem = po_stage_hb(s, d->buf[cur][0], d->buf[cur][1], d->buf_cap, d->buf[cur ^ 1][0], d->buf[cur ^ 1][1], rem);
        cur ^= 1;
    }
    for (uint32_t i = 0; i < rem; i++) { out[2 * i] = d->buf[cur][0][i]; out[2 * i + 1] = d->buf[cur][1][i]; }
    return rem;
}

/* ------------------------------------------------------------------------------------------------
 * FFT -- the reference's default back end is Apple vDSP_fft_ziptD (closed source, absent here):
 * fftaccelerate.cpp:42-105.  Its published contract is the standard unnormalised DFT (forward
 * e^{-j2pi nk/N}, inverse e^{+j}); restated as an iterative radix-2 with fp64 twiddles from sin/cos.
 * ---------------------------------------------------------------------------------------------- */
void po_fft(double *x, uint32_t n, int dir)
{
    /* twiddle table W_n^k = exp(-2*pi*i*k/n), k < n/2, computed once per size (vDSP_create_fftsetupD does the same);
     * one small cache per thread so that several sizes and several threads (oracle/cpu_baseline.cpp) can be in use at once */
    enum { kTabs = 8 };
    static _Thread_local uint32_t tab_sizes[kTabs];
    static _Thread_local double *tabs[kTabs];
    static _Thread_local int tab_next = 0;
    double *tab = NULL;
    for (int i = 0; i < kTabs; i++)
        if (tab_sizes[i] == n) tab = tabs[i];
    if (!tab) {
        const int slot = tab_next;
        tab_next = (tab_next + 1) % kTabs;
        free(tabs[slot]);
        tab = (double *)malloc((size_t)n * sizeof(double));
        for (uint32_t k = 0; k < n / 2; k++) {
            double ang = -PO_TWOPI * (double)k / (double)n;
            tab[2 * k] = cos(ang);
            tab[2 * k + 1] = sin(ang);
        }
        tabs[slot] = tab;
        tab_sizes[slot] = n;
    }
    /* bit reversal */
    for (uint32_t i = 1, j = 0; i < n; i++) {
        uint32_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double tr = x[2 * i], ti = x[2 * i + 1];
            x[2 * i] = x[2 * j]; x[2 * i + 1] = x[2 * j + 1];
            x[2 * j] = tr; x[2 * j + 1] = ti;
        }
    }
    double sgn = dir >= 0 ? 1.0 : -1.0;
    for (uint32_t len = 2; len <= n; len <<= 1) {
        uint32_t half = len >> 1, step = n / len;
        for (uint32_t s = 0; s < n; s += len) {
            for (uint32_t k = 0; k < half; k++) {
                double wr = tab[2 * k * step], wi = sgn * tab[2 * k * step + 1];
                uint32_t a = s + k, b = s + k + half;
                double br = x[2 * b] * wr - x[2 * b + 1] * wi;
                double bi = x[2 * b] * wi + x[2 * b + 1] * wr;
                x[2 * b] = x[2 * a] - br; x[2 * b + 1] = x[2 * a + 1] - bi;
                x[2 * a] += br; x[2 * a + 1] += bi;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * CFastFIR -- pebblelib/fastfir.cpp
 * ---------------------------------------------------------------------------------------------- */
struct po_fastfir {
    uint32_t fft_size, fir_size;
    double *window;   /* m_pWindowTbl */
    double *coef;     /* m_pFilterCoef (freq domain after setup) */
    double *fftbuf;   /* m_pFFTBuf */
    double *overlap;  /* m_pFFTOverlapBuf */
    int inpos;        /* m_InBufInPos */
    double lo, hi, offset, fs;
};

/* fastfir.cpp:77-145: Blackman-Nuttall window table, zeroed buffers, m_InBufInPos = FIR-1 */
po_fastfir *po_fastfir_new(uint32_t fft_size, uint32_t fir_size)
{
    po_fastfir *f = (po_fastfir *)calloc(1, sizeof(*f));
    f->fft_size = fft_size; f->fir_size = fir_size;
    f->window = (double *)calloc(fir_size, sizeof(double));
    f->coef = (double *)calloc((size_t)fft_size * 2, sizeof(double));
    f->fftbuf = (double *)calloc((size_t)fft_size * 2, sizeof(double));
    f->overlap = (double *)calloc((size_t)fir_size * 2, sizeof(double));
    f->inpos = (int)fir_size - 1;
    for (uint32_t i = 0; i < fir_size; i++) {
        f->window[i] = (0.3635819
                        - 0.4891775 * cos((PO_TWOPI * i) / (fir_size - 1))
                        + 0.1365995 * cos((2.0 * PO_TWOPI * i) / (fir_size - 1))
                        - 0.0106411 * cos((3.0 * PO_TWOPI * i) / (fir_size - 1)));
    }
    f->lo = -1.0; f->hi = 1.0; f->offset = 1.0; f->fs = 1.0; /* :141-144 */
    return f;
}

void po_fastfir_free(po_fastfir *f)
{
    if (!f) return;
    free(f->window); free(f->coef); free(f->fftbuf); free(f->overlap); free(f);
}

/* fastfir.cpp:191-272 */
int po_fastfir_setup(po_fastfir *f, double lo, double hi, double offset, double fs)
{
    if (lo == f->lo && hi == f->hi && offset == f->offset && fs == f->fs) return 0;
    f->lo = lo; f->hi = hi; f->offset = offset; f->fs = fs;
    lo += offset; hi += offset;
    if ((lo >= hi) || (lo >= fs / 2.0) || (lo <= -fs / 2.0) || (hi >= fs / 2.0) || (hi <= -fs / 2.0))
        return -1; /* "Filter Parameter error": previous taps stay active */
    double nFL = lo / fs, nFH = hi / fs;
    double nFc = (nFH - nFL) / 2.0;
    double nFs = PO_TWOPI * (nFH + nFL) / 2.0;
    double fCenter = 0.5 * (double)(f->fir_size - 1);
    memset(f->coef, 0, (size_t)f->fft_size * 2 * sizeof(double));
    for (uint32_t i = 0; i < f->fir_size; i++) {
        double x = (double)i - fCenter, z;
        if ((double)i == fCenter) z = 2.0 * nFc;
        else z = sin(PO_TWOPI * x * nFc) / (PO_PI * x) * f->window[i];
        f->coef[2 * i] = z * cos(nFs * x) / (double)f->fft_size;
        f->coef[2 * i + 1] = z * sin(nFs * x) / (double)f->fft_size;
    }
    po_fft(f->coef, f->fft_size, +1);
    return 0;
}

const double *po_fastfir_coef(const po_fastfir *f) { return f->coef; }

/* fastfir.cpp:281-334: sample-at-a-time overlap-save exactly as written */
int po_fastfir_process(po_fastfir *f, int n, const double *in, double *out)
{
    int i = 0, outpos = 0, len = n;
    int N = (int)f->fft_size, T = (int)f->fir_size;
    if (!n) return 0;
    while (len--) {
        int j = f->inpos - (N - T + 1);
        if (j >= 0) { f->overlap[2 * j] = in[2 * i]; f->overlap[2 * j + 1] = in[2 * i + 1]; }
        f->fftbuf[2 * f->inpos] = in[2 * i]; f->fftbuf[2 * f->inpos + 1] = in[2 * i + 1];
        f->inpos++; i++;
        if (f->inpos >= N) {
            po_fft(f->fftbuf, f->fft_size, +1);
            for (int k = 0; k < N; k++) { /* CpxMpy :325-334 */
                double sr = f->fftbuf[2 * k], si = f->fftbuf[2 * k + 1];
                double mr = f->coef[2 * k], mi = f->coef[2 * k + 1];
                f->fftbuf[2 * k] = mr * sr - mi * si;
                f->fftbuf[2 * k + 1] = mr * si + mi * sr;
            }
            po_fft(f->fftbuf, f->fft_size, -1);
            for (j = T - 1; j < N; j++) { out[2 * outpos] = f->fftbuf[2 * j]; out[2 * outpos + 1] = f->fftbuf[2 * j + 1]; outpos++; }
            for (j = 0; j < T - 1; j++) { f->fftbuf[2 * j] = f->overlap[2 * j]; f->fftbuf[2 * j + 1] = f->overlap[2 * j + 1]; }
            f->inpos = T - 1;
        }
    }
    return outpos;
}

/* ------------------------------------------------------------------------------------------------
 * Spectrum -- pebblelib/fft.cpp + windowfunction.cpp + db.h
 * ---------------------------------------------------------------------------------------------- */
struct po_spectrum {
    uint32_t fft_size, spb;
    int window_type;
    double *window;      /* WindowFunction::window */
    double coherent_gain;
    double max_bin_power;
    double *time, *freq; /* m_timeDomain / unfolded */
    double *prev_power, *prev_amp; /* m_fftPower / m_fftAmplitude (zero-initialised here; the
                                      reference leaves them uninitialised, fft.cpp:107-115) */
};

po_spectrum *po_spectrum_new(uint32_t fft_size, uint32_t samples_per_buffer, int window_type, int lift_clamp)
{
    po_spectrum *s = (po_spectrum *)calloc(1, sizeof(*s));
    /* fft.cpp:72-79 clamp to [2048, 65535] */
    if (fft_size < 2048) fft_size = 2048;
    else if (fft_size > 65535 && !lift_clamp) fft_size = 65535;
    s->fft_size = fft_size; s->spb = samples_per_buffer; s->window_type = window_type;
    s->max_bin_power = 1.0 * samples_per_buffer; /* m_ampMax * m_samplesPerBuffer, fft.cpp:84 */
    s->window = (double *)calloc(samples_per_buffer, sizeof(double));
    s->time = (double *)calloc((size_t)fft_size * 2, sizeof(double));
    s->freq = (double *)calloc((size_t)fft_size * 2, sizeof(double));
    s->prev_power = (double *)calloc(fft_size, sizeof(double));
    s->prev_amp = (double *)calloc(fft_size, sizeof(double));
    if (window_type == 0) {
        /* windowfunction.cpp:49-51,214-235: float two_pi and float a0..a3, (i+0.5)/N phase */
        float two_pi = (float)PO_TWOPI;
        float a0 = 0.35875F, a1 = 0.48829F, a2 = 0.14128F, a3 = 0.01168F;
        double sum = 0;
        int N = (int)samples_per_buffer;
        for (int i = 0; i < N; i++) {
            s->window[i] = a0 - a1 * cos(two_pi * (i + 0.5) / N)
                           + a2 * cos(2.0 * two_pi * (i + 0.5) / N)
                           - a3 * cos(3.0 * two_pi * (i + 0.5) / N);
            sum += s->window[i];
        }
        s->coherent_gain = sum / N;
    } else {
        s->coherent_gain = 1.0; /* NONE: regenerate() leaves it untouched; not used on our paths */
    }
    return s;
}

void po_spectrum_free(po_spectrum *s)
{
    if (!s) return;
    free(s->window); free(s->time); free(s->freq); free(s->prev_power); free(s->prev_amp); free(s);
}
uint32_t po_spectrum_bins(const po_spectrum *s) { return s->fft_size; }
double po_spectrum_coherent_gain(const po_spectrum *s) { return s->coherent_gain; }
const double *po_spectrum_window(const po_spectrum *s) { return s->window; }

static double po_clip_db(double db) { return db < -120.0 ? -120.0 : (db > 0.0 ? 0.0 : db); } /* db.h:24-26 */

/* FFTAccelerate::fftSpectrum, fftaccelerate.cpp:106-119 -> m_applyWindow (fft.cpp:129-157),
 * forward FFT, m_unfoldInOrder (fft.cpp:207-213), calcPowerAverages (fft.cpp:324-399). */
int po_spectrum_process(po_spectrum *s, const double *in, uint32_t n, double *out_db)
{
    int overload = 0;
    uint32_t N = s->fft_size;
    if (s->window_type == 0 && n == s->spb) {
        for (uint32_t i = 0; i < s->spb; i++) {
            if (fabs(in[2 * i]) > 0.9 || fabs(in[2 * i + 1]) > 0.9) overload = 1;
            /* in[i] * windowCpx[i] with windowCpx = (w, 0): full complex product */
            double w = s->window[i];
            s->time[2 * i] = in[2 * i] * w - in[2 * i + 1] * 0.0;
            s->time[2 * i + 1] = in[2 * i] * 0.0 + in[2 * i + 1] * w;
        }
        for (uint32_t i = s->spb; i < N; i++) { s->time[2 * i] = 0; s->time[2 * i + 1] = 0; }
    } else {
        memset(s->time, 0, (size_t)N * 2 * sizeof(double));
        memcpy(s->time, in, (size_t)(n < N ? n : N) * 2 * sizeof(double));
    }
    po_fft(s->time, N, +1);
    uint32_t mid = N / 2;
    memcpy(s->freq + 2 * (size_t)mid, s->time, (size_t)mid * 2 * sizeof(double));
    memcpy(s->freq, s->time + 2 * (size_t)mid, (size_t)(N - mid) * 2 * sizeof(double));
    for (uint32_t i = 0; i < N; i++) {
        double re = s->freq[2 * i], im = s->freq[2 * i + 1];
        double asd = sqrt(re * re + im * im) / s->coherent_gain; /* DB::amplitude, db.h:28-30 */
        double psd = asd * asd;                                  /* amplitudeToPower, db.h:90-92 */
        psd /= s->max_bin_power;
        asd /= s->max_bin_power;
        double bin_amp = (asd + s->prev_amp[i]) / 2;             /* m_isAveraged = true, fft.cpp:101 */
        s->prev_power[i] = psd;
        s->prev_amp[i] = asd;
        double db = (bin_amp == 0) ? -120.0 : 20 * log10(bin_amp); /* amplitudeTodB, db.h:44-48 */
        out_db[i] = po_clip_db(db);
    }
    return overload;
}

/* ------------------------------------------------------------------------------------------------
 * CFir -- pebblelib/fir.cpp
 * ---------------------------------------------------------------------------------------------- */
static double po_izero(double x) /* fir.cpp:494-512 */
{
    double x2 = x / 2.0, sum = 1.0, ds = 1.0, di = 1.0, tmp;
    do {
        tmp = x2 / di;
        tmp *= tmp;
        ds *= tmp;
        sum += ds;
        di += 1.0;
    } while (ds >= 1e-9 * sum);
    return sum;
}

/* CFir::InitLPFilter, fir.cpp:246-337 (MAX_NUMCOEF = 75, filtercoef.h) */
int po_fir_init_lp(po_fir *f, int ntaps, double scale, double astop, double fpass, double fstop, double fs)
{
    double nfp = fpass / fs, nfs = fstop / fs, nfc = (nfs + nfp) / 2.0, beta;
    if (astop < 20.96) beta = 0;
    else if (astop >= 50.0) beta = .1102 * (astop - 8.71);
    else beta = .5842 * pow((astop - 20.96), 0.4) + .07886 * (astop - 20.96);
    f->ntaps = (int)((astop - 8.0) / (2.285 * PO_TWOPI * (nfs - nfp)) + 1);
    if (f->ntaps > 75) f->ntaps = 75;
    if (f->ntaps < 3) f->ntaps = 3;
    if (ntaps) f->ntaps = ntaps;
    double fCenter = .5 * (double)(f->ntaps - 1);
    double izb = po_izero(beta);
    for (int n = 0; n < f->ntaps; n++) {
        double x = (double)n - fCenter, c;
        if ((double)n == fCenter) c = 2.0 * nfc;
        else c = sin(PO_TWOPI * x * nfc) / (PO_PI * x);
        x = ((double)n - ((double)f->ntaps - 1.0) / 2.0) / (((double)f->ntaps - 1.0) / 2.0);
        f->coef[n] = scale * c * po_izero(beta * sqrt(1 - (x * x))) / izb;
    }
    for (int n = 0; n < f->ntaps; n++) f->coef[n + f->ntaps] = f->coef[n];
    for (int n = 0; n < f->ntaps * 2; n++) { f->icoef[n] = f->coef[n]; f->qcoef[n] = f->coef[n]; } /* fir.cpp:289-294 */
    for (int i = 0; i < f->ntaps; i++) { f->zre[i] = 0; f->zim[i] = 0; }
    f->state = 0;
    f->fs = fs;
    return f->ntaps;
}

/* CFir::GenerateHBFilter, fir.cpp:454-468: heterodyne the low-pass prototype into an I/Q (Hilbert) band-pass pair */
void po_fir_generate_hb(po_fir *f, double freq_offset)
{
    for (int n = 0; n < f->ntaps; n++) {
        double a = (PO_TWOPI * freq_offset / f->fs) * ((double)n - ((double)(f->ntaps - 1) / 2.0));
        f->icoef[n] = 2.0 * f->coef[n] * cos(a);
        f->qcoef[n] = 2.0 * f->coef[n] * sin(a);
    }
    for (int n = 0; n < f->ntaps; n++) { f->icoef[n + f->ntaps] = f->icoef[n]; f->qcoef[n + f->ntaps] = f->qcoef[n]; }
}

/* CFir::ProcessFilter (complex), fir.cpp:106-132: circular delay line, summation in slot order */
void po_fir_process_cpx(po_fir *f, int n, const double *in, double *out)
{
    for (int i = 0; i < n; i++) {
        f->zre[f->state] = in[2 * i];
        f->zim[f->state] = in[2 * i + 1];
        const double *hi = f->icoef + f->ntaps - f->state, *hq = f->qcoef + f->ntaps - f->state;
        double ar = hi[0] * f->zre[0], ai = hq[0] * f->zim[0];
        for (int j = 1; j < f->ntaps; j++) { ar += hi[j] * f->zre[j]; ai += hq[j] * f->zim[j]; }
        if (--f->state < 0) f->state += f->ntaps;
        out[2 * i] = ar; out[2 * i + 1] = ai;
    }
}

/* ------------------------------------------------------------------------------------------------
 * CIir -- pebblelib/iir.cpp (RBJ biquads, direct form 2)
 * ---------------------------------------------------------------------------------------------- */
static void po_iir_common(po_iir *q, double f0, double Q, double fs, double *w0, double *alpha, double *A)
{
    *w0 = PO_TWOPI * f0 / fs;
    *alpha = sin(*w0) / (2.0 * Q);
    *A = 1.0 / (1.0 + *alpha);
    q->a1 = *A * (-2.0 * cos(*w0));
    q->a2 = *A * (1.0 - *alpha);
    q->w1a = q->w2a = q->w1b = q->w2b = 0.0;
}
void po_iir_init_lp(po_iir *q, double f0, double Q, double fs) /* iir.cpp:88-103 */
{
    double w0, al, A; po_iir_common(q, f0, Q, fs, &w0, &al, &A);
    q->b0 = A * ((1.0 - cos(w0)) / 2.0); q->b1 = A * (1.0 - cos(w0)); q->b2 = A * ((1.0 - cos(w0)) / 2.0);
}
void po_iir_init_hp(po_iir *q, double f0, double Q, double fs) /* iir.cpp:110-125 */
{
    double w0, al, A; po_iir_common(q, f0, Q, fs, &w0, &al, &A);
    q->b0 = A * ((1.0 + cos(w0)) / 2.0); q->b1 = -A * (1.0 + cos(w0)); q->b2 = A * ((1.0 + cos(w0)) / 2.0);
}
void po_iir_init_bp(po_iir *q, double f0, double Q, double fs) /* iir.cpp:131-146 */
{
    double w0, al, A; po_iir_common(q, f0, Q, fs, &w0, &al, &A);
    q->b0 = A * al; q->b1 = 0.0; q->b2 = A * -al;
}
void po_iir_init_br(po_iir *q, double f0, double Q, double fs) /* iir.cpp:152-167 */
{
    double w0, al, A; po_iir_common(q, f0, Q, fs, &w0, &al, &A);
    q->b0 = A * 1.0; q->b1 = A * (-2.0 * cos(w0)); q->b2 = A * 1.0;
}
void po_iir_process_cpx(po_iir *q, int n, const double *in, double *out) /* iir.cpp:191-207 */
{
    for (int i = 0; i < n; i++) {
        double w0a = in[2 * i] - q->a1 * q->w1a - q->a2 * q->w2a;
        out[2 * i] = q->b0 * w0a + q->b1 * q->w1a + q->b2 * q->w2a;
        q->w2a = q->w1a; q->w1a = w0a;
        double w0b = in[2 * i + 1] - q->a1 * q->w1b - q->a2 * q->w2b;
        out[2 * i + 1] = q->b0 * w0b + q->b1 * q->w1b + q->b2 * q->w2b;
        q->w2b = q->w1b; q->w1b = w0b;
    }
}

/* ------------------------------------------------------------------------------------------------
 * AM demod -- application/demod/demod_am.cpp
 * ---------------------------------------------------------------------------------------------- */
void po_demod_am_init(po_demod_am *d, double fs)
{
    memset(d, 0, sizeof(*d));
    d->fs = fs;
    po_demod_am_set_bandwidth(d, 16000); /* ctor, demod_am.cpp:9 */
}
void po_demod_am_set_bandwidth(po_demod_am *d, double bw) /* demod_am.cpp:17-21 */
{
    po_fir_init_lp(&d->lp, 0, 1.0, 50.0, bw, bw * 1.8, d->fs);
}
void po_demod_am_process(po_demod_am *d, const double *in, double *out, int n) /* demod_am.cpp:40-64 */
{
    for (int i = 0; i < n; i++) {
        double mag = sqrt(in[2 * i] * in[2 * i] + in[2 * i + 1] * in[2 * i + 1]);
        d->dc = (0.9999f * d->dc_last) + mag; /* DC_ALPHA is a float literal, :36 */
        double am = d->dc - d->dc_last;
        d->dc_last = d->dc;
        out[2 * i] = am; out[2 * i + 1] = am;
    }
    po_fir_process_cpx(&d->lp, n, out, out);
}

/* ------------------------------------------------------------------------------------------------
 * NFM demod (PLL) -- application/demod/demod_nfm.cpp.  parity unpinned: the reference offers no vector for it.
 * ---------------------------------------------------------------------------------------------- */
void po_demod_nfm_init(po_demod_nfm *d, double fs) /* ctor :24-37, init :44-66 */
{
    memset(d, 0, sizeof(*d));
    d->fs = fs;
    double norm = PO_TWOPI / fs;
    d->nco_lo = (float)(-15000.0 * norm);                 /* FMPLL_RANGE */
    d->nco_hi = (float)(15000.0 * norm);
    d->alpha = (float)(2.0 * .707 * 3000.0 * norm);       /* FMPLL_ZETA, FMPLL_BW = VOICE_BANDWIDTH */
    d->beta = (float)((double)(d->alpha * d->alpha) / (4.0 * .707 * .707)); /* float*float, then double divide */
    d->out_gain = 1.0f;
    d->dc_alpha = (float)(1.0 - exp(-1.0 / (fs * 0.001))); /* FMDC_ALPHA */
    po_fir_init_lp(&d->lp, 0, 1.0, 50.0, 3000.0, 1.6 * 3000.0, fs);
}
void po_demod_nfm_process(po_demod_nfm *d, const double *in, double *out, int n) /* processBlockNCO :225-257 */
{
    for (int i = 0; i < n; i++) {
        double nco_sin = (double)sinf(d->phase), nco_cos = (double)cosf(d->phase); /* sin(float) -> float overload */
        double tr = nco_cos * in[2 * i] - nco_sin * in[2 * i + 1];
        double ti = nco_cos * in[2 * i + 1] + nco_sin * in[2 * i];
        double phzerror = -atan2(ti, tr);
        d->nco_freq = (float)((double)d->nco_freq + ((double)d->beta * phzerror));
        if (d->nco_freq > d->nco_hi) d->nco_freq = d->nco_hi;
        else if (d->nco_freq < d->nco_lo) d->nco_freq = d->nco_lo;
        d->phase = (float)((double)d->phase + ((double)d->nco_freq + (double)d->alpha * phzerror));
        d->err_dc = (float)((1.0 - (double)d->dc_alpha) * (double)d->err_dc + (double)d->dc_alpha * (double)d->nco_freq);
        float o = (d->nco_freq - d->err_dc) * d->out_gain; /* float arithmetic; CPX = real assigns imag 0 */
        out[2 * i] = (double)o;
        out[2 * i + 1] = 0.0;
    }
    d->phase = (float)fmod((double)d->phase, PO_TWOPI);
    po_fir_process_cpx(&d->lp, n, out, out);
}

/* ------------------------------------------------------------------------------------------------
 * SAM demod (PLL) -- application/demod/demod_sam.cpp.  parity unpinned: the reference offers no vector for it.
 * ---------------------------------------------------------------------------------------------- */
void po_demod_sam_init(po_demod_sam *d, double fs) /* ctor :5-32; sampleRate is quint32 there */
{
    memset(d, 0, sizeof(*d));
    d->fs = fs;
    int bw = 100, lim = 1000;
    float zeta = 0.707f;
    d->alpha = (float)(2.0 * zeta * bw * PO_TWOPI / fs);
    d->beta = (float)((double)(d->alpha * d->alpha) / (4.0 * zeta * zeta));
    d->lo = (float)(-lim * PO_TWOPI / fs);
    d->hi = (float)(lim * PO_TWOPI / fs);
    po_fir_init_lp(&d->bp, 0, 1.0, 40.0, 4500, 5500, fs);
    po_fir_generate_hb(&d->bp, 5000.0);
}
static double po_phase_cpx(double re, double im) /* CpxUtil::phaseCpx, cpx.cpp:5-21 */
{
    double t = atan(im / ((re == 0) ? 1e-200 : re));
    if (re < 0 && im < 0) t -= 3.14159265358979323846;
    else if (re < 0 && im >= 0) t += 3.14159265358979323846;
    return t;
}
void po_demod_sam_process(po_demod_sam *d, const double *in, double *out, int n) /* :41-112 */
{
    for (int i = 0; i < n; i++) {
        double sr = in[2 * i], si = in[2 * i + 1];
        double zr = (double)cosf(d->phase), zi = (double)sinf(d->phase);
        double pr = zr * sr - zi * si, pi = zr * si + zi * sr; /* z * sig */
        float diff = (float)(sqrt(sr * sr + si * si) * po_phase_cpx(pr, pi));
        d->freq = d->freq + d->beta * diff; /* float arithmetic */
        if (d->freq < d->lo) d->freq = d->lo;
        if (d->freq > d->hi) d->freq = d->hi;
        d->phase = d->phase + (d->freq + d->alpha * diff);
        while ((double)d->phase >= PO_TWOPI) d->phase = (float)((double)d->phase - PO_TWOPI);
        while (d->phase < 0) d->phase = (float)((double)d->phase + PO_TWOPI);
        d->dc_re = (0.9999f * d->dc_re_last) + pr;
        d->dc_im = (0.9999f * d->dc_im_last) + pi;
        out[2 * i] = d->dc_re - d->dc_re_last;
        out[2 * i + 1] = d->dc_im - d->dc_im_last;
        d->dc_re_last = d->dc_re;
        d->dc_im_last = d->dc_im;
    }
    po_fir_process_cpx(&d->bp, n, out, out);
    for (int i = 0; i < n; i++) {
        double tr = out[2 * i], ti = out[2 * i + 1];
        out[2 * i + 1] = tr - ti; /* upper sideband -> right */
        out[2 * i] = tr + ti;     /* lower sideband -> left */
    }
}

/* ------------------------------------------------------------------------------------------------
 * CDownConvert -- pebblelib/downconvert.cpp, filtercoef.h.  TEST INFRASTRUCTURE like the rest of this file.
 * ---------------------------------------------------------------------------------------------- */
#include "dc_taps.h"
#define PO_DC_MAX_STAGES 9       /* MAX_DECSTAGES 10 "one more than max" (downconvert.h:23): a tenth stage would overwrite the list's NULL */
#define PO_DC_MIN_OUTPUT_RATE (7900.0 * 2.0) /* downconvert.cpp:58 */
typedef struct {
    int design;                  /* index into pebble_dc_designs; 0 = CCicN3DecimateBy2 */
    int fixed11;                 /* CHalfBand11TapDecimateBy2 (no doubled tap) */
    double hist[2 * 64];         /* the last ntaps - 1 inputs (m_pHBFirBuf[0 .. FirLength-2], d0..d9, m_Xeven / m_Xodd) */
} po_dc_stage;
struct po_downconvert {
    double in_rate, max_bw, out_rate, nco_freq, nco_inc, cw_offset, osc_cos, osc_sin, osc1_re, osc1_im;
    po_dc_stage st[PO_DC_MAX_STAGES];
    int nst;
    double *work;
    int work_cap;
};
po_downconvert *po_downconvert_new(void) /* ctor, :63-77 */
{
    po_downconvert *d = (po_downconvert *)calloc(1, sizeof(*d));
    d->in_rate = 100000.0;
    d->max_bw = 10000.0;
    d->osc1_re = 1.0;
    return d;
}
void po_downconvert_free(po_downconvert *d)
{
    if (!d) return;
    free(d->work);
    free(d);
}
void po_downconvert_set_frequency(po_downconvert *d, double f) /* :100-112 */
{
    f = -f;
    d->nco_freq = f + d->cw_offset;
    d->nco_inc = PO_TWOPI * d->nco_freq / d->in_rate;
    d->osc_cos = cos(d->nco_inc);
    d->osc_sin = sin(d->nco_inc);
}
void po_downconvert_set_cw_offset(po_downconvert *d, double offset) { d->cw_offset = offset; }
double po_downconvert_set_data_rate(po_downconvert *d, double in_rate, double max_bw, int simple)
{
    double f = in_rate;
    if (d->in_rate != in_rate || d->max_bw != max_bw) { /* :143-144 / :217-218 */
        d->in_rate = in_rate;
        d->max_bw = max_bw;
        memset(d->st, 0, sizeof(d->st));
        d->nst = 0;
        if (simple) { /* SetDataRateSimple, :224-229: HB51 until the rate is at or under 400 kHz */
            while (f > 400000.0 && d->nst < PO_DC_MAX_STAGES) {
                d->st[d->nst++].design = PEBBLE_DC_NDESIGNS - 1;
                f /= 2.0;
            }
        } else {
            const double last_max = pebble_dc_designs[PEBBLE_DC_NDESIGNS - 1].max_a - pebble_dc_designs[PEBBLE_DC_NDESIGNS - 1].max_b;
            while (f > (d->max_bw / last_max) && f > PO_DC_MIN_OUTPUT_RATE && d->nst < PO_DC_MAX_STAGES) { /* :152 */
                for (int k = 0; k < PEBBLE_DC_NDESIGNS; k++) { /* the ladder, :154-203: first design whose limit the rate clears */
                    const double mx = pebble_dc_designs[k].max_a - pebble_dc_designs[k].max_b;
                    if (f >= d->max_bw / mx) {
                        d->st[d->nst].design = k;
                        d->st[d->nst].fixed11 = (k == 1); /* :158-161: the unrolled 11-tap class, not the generic one */
                        d->nst++;
                        break;
                    }
                }
                f /= 2.0;
            }
        }
        d->out_rate = f;
        /* SetFrequency(m_NcoFreq), :205 / :232, as written: the STORED (already negated, offset included) value goes through the
         * negation and the offset once more -- a SetDataRate that changes anything flips the sign of the tuned frequency.
         * Harmless in the reference's own order of calls (rate first, at 0 Hz; receiver.cpp:198, :718) */
        po_downconvert_set_frequency(d, d->nco_freq);
    }
    return d->out_rate;
}
int po_downconvert_chain_len(const po_downconvert *d) { return d->nst; }
int po_downconvert_stage_taps(const po_downconvert *d, int i) { return pebble_dc_designs[d->st[i].design].ntaps; }

/* one DecBy2 over n complex samples in x (in place), returns n / 2 */
static int po_dc_stage_run(po_dc_stage *s, int n, double *x)
{
    const int T = pebble_dc_designs[s->design].ntaps;
    if (T == 0) { /* CCicN3DecimateBy2::DecBy2, :517-533; hist = {Xeven, Xodd} */
        int j = 0;
        for (int i = 0; i < n; i += 2, j++) {
            const double er = x[2 * i], ei = x[2 * i + 1], odr = x[2 * i + 2], odi = x[2 * i + 3];
            x[2 * j] = .125 * (odr + s->hist[0] + 3.0 * (s->hist[2] + er));
            x[2 * j + 1] = .125 * (odi + s->hist[1] + 3.0 * (s->hist[3] + ei));
            s->hist[2] = odr; s->hist[3] = odi;
            s->hist[0] = er; s->hist[1] = ei;
        }
        return j;
    }
    const double *h = pebble_dc_designs[s->design].h;
    if (n < T) return n / 2; /* "safety net", :361-362 (the 11-tap class has none: callers keep n >= 11) */
    /* both halfband classes: buf = [T - 1 previous inputs | the n new ones], output i/2 from buf[i .. i + T - 1] */
    double *buf = (double *)malloc((size_t)(n + T) * 2 * sizeof(double));
    memcpy(buf, s->hist, (size_t)(T - 1) * 2 * sizeof(double));
    memcpy(buf + 2 * (T - 1), x, (size_t)n * 2 * sizeof(double));
    int no = 0;
    for (int i = 0; i < n; i += 2) {
        double ar, ai;
        if (s->fixed11) { /* CHalfBand11TapDecimateBy2, :429-489: H0 H2 H4 H5 H6 H8 H10 in this order */
            static const int idx[7] = {0, 2, 4, 5, 6, 8, 10};
            ar = 0; ai = 0;
            for (int q = 0; q < 7; q++) {
                if (q == 0) { ar = h[0] * buf[2 * i]; ai = h[0] * buf[2 * i + 1]; }
                else { ar += h[idx[q]] * buf[2 * (i + idx[q])]; ai += h[idx[q]] * buf[2 * (i + idx[q]) + 1]; }
            }
        } else { /* CHalfBandDecimateBy2::DecBy2, :368-384: tap 0 enters twice (the accumulator's start value AND j = 0) */
            ar = buf[2 * i] * h[0];
            ai = buf[2 * i + 1] * h[0];
            for (int j = 0; j < T; j += 2) {
                ar += buf[2 * (i + j)] * h[j];
                ai += buf[2 * (i + j) + 1] * h[j];
            }
            ar += buf[2 * (i + (T - 1) / 2)] * h[(T - 1) / 2];
            ai += buf[2 * (i + (T - 1) / 2) + 1] * h[(T - 1) / 2];
        }
        x[2 * no] = ar; x[2 * no + 1] = ai;
        no++;
    }
    memcpy(s->hist, buf + 2 * n, (size_t)(T - 1) * 2 * sizeof(double)); /* the last T - 1 inputs, :386-390 / :491-495 */
    free(buf);
    return no;
}

int po_downconvert_process(po_downconvert *d, int n, const double *in, double *out) /* ProcessData, :250-335 */
{
    if (d->work_cap < n) {
        free(d->work);
        d->work = (double *)malloc((size_t)n * 2 * sizeof(double));
        d->work_cap = n;
    }
    double *w = d->work;
    for (int i = 0; i < n; i++) { /* NCO_OSC, :288-293 and the product :305-307 */
        const double or_ = d->osc1_re * d->osc_cos - d->osc1_im * d->osc_sin;
        const double oi = d->osc1_im * d->osc_cos + d->osc1_re * d->osc_sin;
        const double gn = 1.95 - (d->osc1_re * d->osc1_re + d->osc1_im * d->osc1_im);
        d->osc1_re = gn * or_;
        d->osc1_im = gn * oi;
        const double xr = in[2 * i], xi = in[2 * i + 1];
        w[2 * i] = xr * or_ - xi * oi;
        w[2 * i + 1] = xr * oi + xi * or_;
    }
    int m = n;
    for (int j = 0; j < d->nst; j++) m = po_dc_stage_run(&d->st[j], m, w); /* :319-326 */
    memcpy(out, w, (size_t)m * 2 * sizeof(double));
    return m;
}

/* ------------------------------------------------------------------------------------------------
 * RDS branch of Demod_WFM -- application/demod/demod_wfm.cpp:296-357 (in processDataStereo), 488-761; constants of
 * application/demod/rbdsconstants.h.  Up to the group queue; what the GUI makes of a group (rdsdecode.cpp) is not restated
 * ---------------------------------------------------------------------------------------------- */
static double po_wfm_arctan2(double y, double x);
#define PO_RDS_Q_SIZE 100                       /* RDS_Q_SIZE, demod_wfm.h:21 */
#define PO_RDS_BITRATE (57000.0 / 48.0)         /* RDS_BITRATE */
static const uint32_t po_rds_parckh[16] = {     /* PARCKH, rbdsconstants.h: the last 16 rows of the parity-check matrix */
    0x2DC, 0x16E, 0x0B7, 0x287, 0x39F, 0x313, 0x355, 0x376, 0x1BB, 0x201, 0x3DC, 0x1EE, 0x0F7, 0x2A7, 0x38F, 0x31B
};
static const uint32_t po_rds_blk_offset[8] = {  /* BLK_OFFSET_TBL: syndromes of offset words A B C D, A B C' D */
    0x3D8, 0x3D4, 0x25C, 0x258, 0x3D8, 0x3D4, 0x3CC, 0x258
};
struct po_rds {
    double rate;
    po_downconvert dc;                          /* m_RdsDownConvert: SetDataRate(fs, 8000), SetFrequency(-57000), :187-188 */
    po_fir lp;                                  /* m_RdsBPFilter, :496 */
    double mcoef[2 * 75], mz[75];               /* m_RdsMatchedFilter: InitConstFir keeps m_Coef only, the real ProcessFilter uses it */
    int mtaps, mstate;
    po_iir bitsync;                             /* m_RdsBitSyncFilter, :522 */
    double nco_phase, nco_freq, nco_lo, nco_hi, pll_alpha, pll_beta; /* :492-503 */
    double last_sync, last_sync_slope, last_data;
    int last_bit;
    uint32_t in_bits;                           /* m_InBitStream */
    int bit_pos, cur_block, state, bgroup, block_errors;
    uint16_t block[4];
    po_rds_group q[PO_RDS_Q_SIZE], last_group;
    int qhead, qtail;
    /* for the tests: the last call's signals and what happened since the last drain */
    double *data, *sync; int n_last, cap_last;
    uint8_t *bits; int n_bits, cap_bits;
    po_rds_group *pushed; int n_pushed, cap_pushed;
};
static void po_rds_init(struct po_rds *r, double fs) /* setSampleRate :187-191 and initRds :490-537 */
{
    memset(r, 0, sizeof(*r));
    r->dc.in_rate = 100000.0; r->dc.max_bw = 10000.0; r->dc.osc1_re = 1.0;  /* CDownConvert's constructor, downconvert.cpp:63-77 */
    r->rate = po_downconvert_set_data_rate(&r->dc, fs, 8000.0, 0);
    po_downconvert_set_frequency(&r->dc, -57000.0);
    po_fir_init_lp(&r->lp, 0, 1.0, 40.0, 2400.0, 1.3 * 2400.0, r->rate);
    const double norm = PO_TWOPI / r->rate;
    r->nco_lo = (0.0 - 12.0) * norm;                        /* RDSPLL_RANGE 12 */
    r->nco_hi = (0.0 + 12.0) * norm;
    r->pll_alpha = 2.0 * .707 * 1.0 * norm;                 /* RDSPLL_ZETA .707, RDSPLL_BW 1 */
    r->pll_beta = (r->pll_alpha * r->pll_alpha) / (4.0 * .707 * .707);
    int len = (int)(r->rate / PO_RDS_BITRATE);              /* m_MatchCoefLength = SampleRate / RDS_BITRATE, an int */
    double coef[2 * 75 + 2];
    memset(coef, 0, sizeof(coef));
    for (int i = 0; i <= len; i++) {                        /* :508-517, as written (i = 0 divides by zero: 1/inf = 0, and -0 is stored last) */
        const double t = (double)i / r->rate, x = t * PO_RDS_BITRATE, x64 = 64.0 * x;
        coef[i + len] = .75 * cos(2.0 * PO_TWOPI * x) * ((1.0 / (1.0 / x - x64)) - (1.0 / (9.0 / x - x64)));
        coef[len - i] = -.75 * cos(2.0 * PO_TWOPI * x) * ((1.0 / (1.0 / x - x64)) - (1.0 / (9.0 / x - x64)));
    }
    len *= 2;                                               /* :518: the filter takes 2 len of the 2 len + 1 values */
    r->mtaps = len > 75 ? 75 : len;                         /* InitConstFir, fir.cpp:176-197 (MAX_NUMCOEF) */
    for (int i = 0; i < r->mtaps; i++) { r->mcoef[i] = coef[i]; r->mcoef[r->mtaps + i] = coef[i]; }
    po_iir_init_bp(&r->bitsync, PO_RDS_BITRATE, 500, r->rate);
    r->last_data = 0.0;                                     /* m_RdsLastData is never initialised in the reference; it decides one bit at most */
}
static void po_rds_log_push(struct po_rds *r, po_rds_group g)
{
    if (r->n_pushed == r->cap_pushed) {
        r->cap_pushed = r->cap_pushed ? 2 * r->cap_pushed : 64;
        r->pushed = (po_rds_group *)realloc(r->pushed, (size_t)r->cap_pushed * sizeof(po_rds_group));
    }
    r->pushed[r->n_pushed++] = g;
}
/* Demod_WFM::checkBlock, :708-757 */
static uint32_t po_rds_check_block(struct po_rds *r, uint32_t syndrome_offset, int use_fec)
{
    uint32_t testblock = (0x3FFFFFF & r->in_bits);
    uint32_t syndrome = testblock >> 16;
    for (int i = 0; i < 16; i++) {
        if (testblock & 0x8000) syndrome ^= po_rds_parckh[i];
        testblock <<= 1;
    }
    syndrome ^= syndrome_offset;
    if (syndrome && use_fec) {
        uint32_t correctmask = (1u << (26 - 1));
        for (int i = 0; i < 16; i++) { /* Meggitt decoder: bursts of up to five bits */
            if (syndrome & 0x200) {
                if (0 == (syndrome & 0x1F)) {
                    r->in_bits ^= correctmask;
                    syndrome <<= 1;
                } else {
                    syndrome <<= 1;
                    syndrome ^= 0x5B9; /* CRC_POLY */
                }
            } else {
                syndrome <<= 1;
            }
            correctmask >>= 1;
        }
        syndrome &= 0x3FF;
    }
    return syndrome;
}
static void po_rds_queue_group(struct po_rds *r)
{
    po_rds_group g = {r->block[0], r->block[1], r->block[2], r->block[3]};
    r->q[r->qhead++] = g;
    if (r->qhead >= PO_RDS_Q_SIZE) r->qhead = 0;
    po_rds_log_push(r, g);
}
/* Demod_WFM::processNewRdsBit, :576-700 (states STATE_BITSYNC 0, BLOCKSYNC 1, GROUPDECODE 2, GROUPRESYNC 3) */
static void po_rds_new_bit(struct po_rds *r, int bit)
{
    if (r->n_bits == r->cap_bits) {
        r->cap_bits = r->cap_bits ? 2 * r->cap_bits : 1024;
        r->bits = (uint8_t *)realloc(r->bits, (size_t)r->cap_bits);
    }
    r->bits[r->n_bits++] = (uint8_t)bit;
    r->in_bits = (r->in_bits << 1) | (uint32_t)bit;
    switch (r->state) {
    case 0:
        if (0 == po_rds_check_block(r, 0x3D8, 0)) {
            r->bit_pos = 0;
            r->bgroup = 0;
            r->block[0] = (uint16_t)(r->in_bits >> 10);
            r->cur_block = 1;
            r->state = 1;
        }
        break;
    case 1:
        r->bit_pos++;
        if (r->bit_pos >= 26) {
            r->bit_pos = 0;
            if (po_rds_check_block(r, po_rds_blk_offset[r->cur_block + r->bgroup], 0)) {
                r->state = 0;
            } else {
                r->block[r->cur_block] = (uint16_t)(r->in_bits >> 10);
                if ((1 == r->cur_block) && (r->block[r->cur_block] & 0x0800)) r->bgroup = 4; /* GROUPB_BIT */
                else r->bgroup = 0;
                if (r->cur_block >= 3) {
                    po_rds_queue_group(r);
                    r->cur_block = 0;
                    r->block_errors = 0;
                    r->state = 2;
                } else {
                    r->cur_block++;
                }
            }
        }
        break;
    case 2:
        r->bit_pos++;
        if (r->bit_pos >= 26) {
            r->bit_pos = 0;
            if (po_rds_check_block(r, po_rds_blk_offset[r->cur_block + r->bgroup], 1)) { /* USE_FEC 1 */
                r->block_errors++;
                if (r->block_errors > 5) { /* BLOCK_ERROR_LIMIT */
                    const po_rds_group zero = {0, 0, 0, 0};
                    r->qhead = r->qtail = 0;
                    r->q[r->qhead++] = zero;
                    po_rds_log_push(r, zero);
                    r->state = 0;
                } else {
                    r->cur_block++;
                    if (r->cur_block > 3) r->cur_block = 0;
                    if (0 != r->cur_block) r->state = 3;
                }
            } else {
                r->block[r->cur_block] = (uint16_t)(r->in_bits >> 10);
                if ((1 == r->cur_block) && (r->block[r->cur_block] & 0x0800)) r->bgroup = 4;
                else r->bgroup = 0;
                r->cur_block++;
                if (r->cur_block > 3) {
                    po_rds_queue_group(r);
                    r->cur_block = 0;
                    r->block_errors = 0;
                }
            }
        }
        break;
    case 3:
        r->bit_pos++;
        if (r->bit_pos >= 26) {
            r->bit_pos = 0;
            r->cur_block++;
            if (r->cur_block > 3) {
                r->cur_block = 0;
                r->state = 2;
            }
        }
        break;
    }
}
/* the RDS lines of processDataStereo, :296-357, on m_CpxRawFm (n complex values; the down-converter mixes them in place) */
static void po_rds_process(struct po_rds *r, int n, const double *cpx)
{
    double *raw = (double *)malloc((size_t)n * 2 * sizeof(double));
    const int len = po_downconvert_process(&r->dc, n, cpx, raw);      /* :297 */
    po_fir_process_cpx(&r->lp, len, raw, raw);                        /* :301 */
    if (r->cap_last < len) {
        r->cap_last = len;
        r->data = (double *)realloc(r->data, (size_t)len * sizeof(double));
        r->sync = (double *)realloc(r->sync, (size_t)len * sizeof(double));
    }
    double *mag = r->sync, *data = r->data;
    for (int i = 0; i < len; i++) { /* processRdsPll, :542-569 (fsincos there; sin and cos here) */
        const double sn = sin(r->nco_phase), cs = cos(r->nco_phase);
        const double tr = cs * raw[2 * i] - sn * raw[2 * i + 1];
        const double ti = cs * raw[2 * i + 1] + sn * raw[2 * i];
        const double err = -po_wfm_arctan2(ti, tr);
        r->nco_freq += (r->pll_beta * err);
        if (r->nco_freq > r->nco_hi) r->nco_freq = r->nco_hi;
        else if (r->nco_freq < r->nco_lo) r->nco_freq = r->nco_lo;
        r->nco_phase += (r->nco_freq + r->pll_alpha * err);
        mag[i] = ti;
    }
    r->nco_phase = fmod(r->nco_phase, PO_TWOPI);
    for (int i = 0; i < len; i++) { /* m_RdsMatchedFilter.ProcessFilter (real), fir.cpp:77-95 */
        r->mz[r->mstate] = mag[i];
        const double *h = &r->mcoef[r->mtaps - r->mstate];
        double acc = h[0] * r->mz[0];
        for (int j = 1; j < r->mtaps; j++) acc += h[j] * r->mz[j];
        if (--r->mstate < 0) r->mstate += r->mtaps;
        data[i] = acc;
    }
    for (int i = 0; i < len; i++) { /* :312-317: square, then the resonator (real CIir::ProcessFilter, iir.cpp:173-182) */
        const double w0 = data[i] * data[i] - r->bitsync.a1 * r->bitsync.w1a - r->bitsync.a2 * r->bitsync.w2a;
        mag[i] = r->bitsync.b0 * w0 + r->bitsync.b1 * r->bitsync.w1a + r->bitsync.b2 * r->bitsync.w2a;
        r->bitsync.w2a = r->bitsync.w1a; r->bitsync.w1a = w0;
    }
    for (int i = 0; i < len; i++) { /* :320-353 */
        const double slope = mag[i] - r->last_sync;
        r->last_sync = mag[i];
        if ((slope < 0.0) && (r->last_sync_slope * slope) < 0.0) {
            const int bit = (r->last_data >= 0) ? 1 : 0;
            po_rds_new_bit(r, bit ^ r->last_bit);
            r->last_bit = bit;
        }
        r->last_data = data[i];
        r->last_sync_slope = slope;
    }
    r->n_last = len;
    free(raw);
}

/* ------------------------------------------------------------------------------------------------
 * WFM mono demod -- application/demod/demod_wfm.cpp
 * ---------------------------------------------------------------------------------------------- */
void po_demod_wfm_init(po_demod_wfm *d, double fs) /* init()+setSampleRate(), demod_wfm.cpp:100-196 */
{
    memset(d, 0, sizeof(*d));
    d->fs = fs;
    po_iir_init_lp(&d->mono_lp, 75000, 1.0, fs);                     /* :164 */
    po_fir_init_lp(&d->lp, 0, 1.0, 60.0, 15000.0, 1.4 * 15000.0, fs); /* :175 (m_OutRate == rate) */
    po_iir_init_br(&d->notch, 19000.0, 5, fs);                        /* :178 */
    d->deemph_alpha = (1.0 - exp(-1.0 / (fs * 75E-6)));               /* :181-183, 451-457 */
    /* stereo members, setSampleRate :161-171 and initPilotPll :371-386 */
    d->phase_adjust = -7.267e-6 * fs + 3.677;                         /* PHASE_ADJ_M, PHASE_ADJ_B (:60-61, :161) */
    po_fir_init_const(&d->hilbert, fs);                               /* :167 InitConstFir(HILB_LENGTH, HILBLP_H) */
    po_fir_generate_hb(&d->hilbert, 42000);                           /* :168 */
    po_iir_init_bp(&d->pilot_bp, 19000.0, 500, fs);                   /* :171 */
    const double norm = PO_TWOPI / fs;
    d->nco_phase = 0.0;
    d->nco_freq = -19000.0;                                           /* :374, as written: Hz, not yet normalised */
    d->nco_lo = (d->nco_freq - 20.0) * norm;                          /* PILOTPLL_RANGE 20 */
    d->nco_hi = (d->nco_freq + 20.0) * norm;
    d->pll_alpha = 2.0 * .707 * 10.0 * norm;                          /* PILOTPLL_ZETA .707, PILOTPLL_BW 10 */
    d->pll_beta = (d->pll_alpha * d->pll_alpha) / (4.0 * .707 * .707);
    d->err_ave = 0.0;
    d->err_alpha = (1.0 - exp(-1.0 / (fs * .5)));                     /* LOCK_TIMECONST .5 */
    d->rds = (struct po_rds *)malloc(sizeof(struct po_rds));          /* :187-191 */
    po_rds_init(d->rds, fs);
}
void po_demod_wfm_free(po_demod_wfm *d)
{
    if (!d || !d->rds) return;
    free(d->rds->dc.work); free(d->rds->data); free(d->rds->sync); free(d->rds->bits); free(d->rds->pushed);
    free(d->rds);
    d->rds = NULL;
}
double po_demod_wfm_rds_rate(const po_demod_wfm *d) { return d->rds->rate; }
int po_demod_wfm_rds_last(const po_demod_wfm *d, double *data, double *sync, int cap)
{
    const int n = d->rds->n_last < cap ? d->rds->n_last : cap;
    if (data) memcpy(data, d->rds->data, (size_t)n * sizeof(double));
    if (sync) memcpy(sync, d->rds->sync, (size_t)n * sizeof(double));
    return d->rds->n_last;
}
int po_demod_wfm_rds_bits(po_demod_wfm *d, uint8_t *bits, int cap)
{
    const int n = d->rds->n_bits < cap ? d->rds->n_bits : cap;
    if (bits) memcpy(bits, d->rds->bits, (size_t)n);
    d->rds->n_bits = 0;
    return n;
}
int po_demod_wfm_rds_pushed(po_demod_wfm *d, po_rds_group *g, int cap)
{
    const int n = d->rds->n_pushed < cap ? d->rds->n_pushed : cap;
    if (g) memcpy(g, d->rds->pushed, (size_t)n * sizeof(po_rds_group));
    d->rds->n_pushed = 0;
    return n;
}
int po_demod_wfm_next_rds_group(po_demod_wfm *d, po_rds_group *g, int *changed) /* getNextRdsGroupData, :763-786 */
{
    struct po_rds *r = d->rds;
    if (changed) *changed = 0;
    if ((r->qhead == r->qtail) || !g) return 0;
    *g = r->q[r->qtail++];
    if (r->qtail >= PO_RDS_Q_SIZE) r->qtail = 0;
    if ((r->last_group.a != g->a) || (r->last_group.b != g->b) || (r->last_group.c != g->c) || (r->last_group.d != g->d)) {
        r->last_group = *g;
        if (changed) *changed = 1;
    }
    return 1;
}

/* HILBLP_H, demod_wfm.cpp:79-98: 61-tap symmetric low-pass prototype (Kaiser-Bessel, 30 kHz at 250 kHz); first 31 taps */
static const double po_hilb_half[31] = {
    -0.000389631665953405, 0.000115430826670992, 0.000945331102222503, 0.001582460677684605,
    0.001370803713784687, -0.000000000000000002, -0.002077413537668161, -0.003656132107176520,
    -0.003372610825000167, -0.000649815020884706, 0.003583263233560064, 0.006997162933343487,
    0.006990985399916562, 0.002383133886438500, -0.005324501734543406, -0.012092135317628615,
    -0.013212201698221963, -0.006168904735839018, 0.007082277142635906, 0.020017841466263672,
    0.024271835962039127, 0.014255112728911837, -0.008597071392140753, -0.034478282954624850,
    -0.048147195828726633, -0.035409729589347565, 0.009623663461671806, 0.080084441681677138,
    0.157278883310078170, 0.217148915611638180, 0.239688166538436750
};
void po_fir_init_const(po_fir *f, double fs) /* CFir::InitConstFir(NumTaps, pCoef, rate), fir.cpp:176-197 */
{
    memset(f, 0, sizeof(*f));
    f->ntaps = 61;
    f->fs = fs;
    for (int i = 0; i < 61; i++) {
        const double h = po_hilb_half[i <= 30 ? i : 60 - i];
        f->coef[i] = h; f->coef[61 + i] = h;
    }
}

/* Demod_WFM::arctan2, :792-821 -- the reference's own approximation, restated with its constants (2 pi where pi/2 would
 * be the textbook value) because the pilot PLL's trajectory depends on it */
static double po_wfm_arctan2(double y, double x)
{
    double angle;
    if (x == 0.0) {
        if (y > 0.0) return PO_TWOPI;
        if (y == 0.0) return 0.0;
        return -PO_TWOPI;
    }
    const double z = y / x;
    if (fabs(z) < 1.0) {
        angle = z / (1.0 + 0.2854 * z * z);
        if (x < 0.0) {
            if (y < 0.0) return angle - PO_PI;
            return angle + PO_PI;
        }
    } else {
        angle = PO_TWOPI - z / (z * z + 0.2854);
        if (y < 0.0) return angle - PO_PI;
    }
    return angle;
}

int po_demod_wfm_process_stereo(po_demod_wfm *d, const double *in, double *out, int n) /* :255-297, 359-362 */
{
    double *raw = (double *)malloc((size_t)n * sizeof(double));
    double *raw2 = (double *)calloc((size_t)n * 2 + 2, sizeof(double));
    double *cpx = (double *)malloc((size_t)n * 2 * sizeof(double));
    double *pil = (double *)malloc((size_t)n * 2 * sizeof(double));
    double *phase = (double *)malloc((size_t)n * sizeof(double));
    for (int i = 0; i < n; i++) { /* :258-263: no mono low-pass in front of the discriminator here */
        double d0r = in[2 * i], d0i = in[2 * i + 1];
        raw[i] = 0.25 * atan2((d->d1_re * d0i - d0r * d->d1_im), (d->d1_re * d0r + d->d1_im * d0i));
        d->d1_re = d0r; d->d1_im = d0i;
        raw2[2 * i] = raw[i]; raw2[2 * i + 1] = raw[i]; /* CFir real-in/complex-out puts the sample in both delay lines, fir.cpp:153-154 */
    }
    po_fir_process_cpx(&d->hilbert, n, raw2, cpx);      /* :268 */
    po_iir_process_cpx(&d->pilot_bp, n, cpx, pil);      /* :271 */
    for (int i = 0; i < n; i++) { /* processPilotPll, :392-429 */
        const double sn = sin(d->nco_phase), cs = cos(d->nco_phase);
        const double tr = cs * pil[2 * i] - sn * pil[2 * i + 1];
        const double ti = cs * pil[2 * i + 1] + sn * pil[2 * i];
        const double err = -po_wfm_arctan2(ti, tr);
        d->nco_freq += (d->pll_beta * err);
        if (d->nco_freq > d->nco_hi) d->nco_freq = d->nco_hi;
        else if (d->nco_freq < d->nco_lo) d->nco_freq = d->nco_lo;
        d->nco_phase += (d->nco_freq + d->pll_alpha * err);
        phase[i] = d->nco_phase + d->phase_adjust;
        d->err_ave = (1.0 - d->err_alpha) * d->err_ave + d->err_alpha * err * err;
    }
    d->nco_phase = fmod(d->nco_phase, PO_TWOPI);
    d->pilot_locked = d->err_ave < 0.05; /* LOCK_MAG_THRESHOLD */
    for (int i = 0; i < n; i++) { /* :272-293 */
        if (d->pilot_locked) {
            const double lmr = 2.0 * raw[i] * sin(phase[i] * 2.0);
            out[2 * i] = raw[i] + lmr; out[2 * i + 1] = raw[i] - lmr;
        } else {
            out[2 * i] = raw[i]; out[2 * i + 1] = raw[i];
        }
    }
    po_rds_process(d->rds, n, cpx);                     /* :296-357 */
    free(raw); free(raw2); free(cpx); free(pil); free(phase);
    /* :359-361, shared with the mono path */
    po_fir_process_cpx(&d->lp, n, out, out);
    for (int i = 0; i < n; i++) {
        d->deemph_re = (1.0 - d->deemph_alpha) * d->deemph_re + d->deemph_alpha * out[2 * i];
        d->deemph_im = (1.0 - d->deemph_alpha) * d->deemph_im + d->deemph_alpha * out[2 * i + 1];
        out[2 * i] = d->deemph_re * 2.0; out[2 * i + 1] = d->deemph_im * 2.0;
    }
    po_iir_process_cpx(&d->notch, n, out, out);
    return d->pilot_locked;
}
void po_demod_wfm_process_mono(po_demod_wfm *d, const double *in, double *out, int n) /* :207-232 */
{
    double *tmp = (double *)malloc((size_t)n * 2 * sizeof(double));
    if (d->fs >= 150000) po_iir_process_cpx(&d->mono_lp, n, in, tmp);
    else memcpy(tmp, in, (size_t)n * 2 * sizeof(double));
    for (int i = 0; i < n; i++) {
        double d0r = tmp[2 * i], d0i = tmp[2 * i + 1];
        double v = 0.25 * atan2((d->d1_re * d0i - d0r * d->d1_im), (d->d1_re * d0r + d->d1_im * d0i));
        out[2 * i] = v; out[2 * i + 1] = v;
        d->d1_re = d0r; d->d1_im = d0i;
    }
    free(tmp);
    po_fir_process_cpx(&d->lp, n, out, out);
    for (int i = 0; i < n; i++) { /* processDeemphasisFilter (complex), :476-485 */
        d->deemph_re = (1.0 - d->deemph_alpha) * d->deemph_re + d->deemph_alpha * out[2 * i];
        d->deemph_im = (1.0 - d->deemph_alpha) * d->deemph_im + d->deemph_alpha * out[2 * i + 1];
        out[2 * i] = d->deemph_re * 2.0; out[2 * i + 1] = d->deemph_im * 2.0;
    }
    po_iir_process_cpx(&d->notch, n, out, out);
}

/* ------------------------------------------------------------------------------------------------
 * Receiver::processIQData skeleton -- application/receiver.cpp
 * ---------------------------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------------------------------
 * AGC -- application/agc.{h,cpp}
 * ---------------------------------------------------------------------------------------------- */
#define PO_AGC_MAX_DELAY_BUF 2048 /* agc.h MAX_DELAY_BUF */
struct po_agc {
    int mode;
    double step_rate;      /* ProcessStep::sampleRate */
    int use_hang, threshold, decay;
    double manual_gain, sample_rate, slope_factor;
    double decay_avg, attack_avg, attack_rise, attack_fall, decay_rise, decay_fall;
    double fixed_gain, knee, gain_slope, peak;
    int sig_ptr, mag_pos, delay_samples, window_samples, hang_time, hang_timer;
    double sig[2 * PO_AGC_MAX_DELAY_BUF];
    double mag[PO_AGC_MAX_DELAY_BUF];
};

/* agc.h: the time constants are float constexprs; they promote to double in the products below */
static const float kAgcDelayTc = .015f, kAgcWindowTc = .018f, kAgcAttackRiseTc = .002f, kAgcAttackFallTc = .005f,
                   kAgcDecayRatio = .3f, kAgcReleaseTc = .05f, kAgcOutScale = 0.7f, kAgcMaxAmp = 1.0f, kAgcMinConst = 1e-8f;

static void po_agc_set_parameters(po_agc *a, int use_hang, int threshold, int slope_factor, int decay)
{
    if (a->mode == 0) { /* agc.cpp:239-246: manual gain, slider dB / 5 in integer arithmetic */
        a->manual_gain = pow(10, (double)(threshold / 5) / 20.0);
        return;
    }
    threshold = -threshold;
    a->manual_gain = 1;
    if (use_hang == a->use_hang && threshold == a->threshold && (double)slope_factor == a->slope_factor && decay == a->decay) return;
    a->use_hang = use_hang; a->threshold = threshold; a->slope_factor = slope_factor; a->decay = decay;
    if (a->sample_rate != a->step_rate) { /* :262-277 */
        a->sample_rate = a->step_rate;
        for (int i = 0; i < PO_AGC_MAX_DELAY_BUF; i++) { a->sig[2 * i] = a->sig[2 * i + 1] = 0.0; a->mag[i] = -16.0; }
        a->sig_ptr = 0; a->hang_timer = 0; a->peak = -16.0; a->decay_avg = -5.0; a->attack_avg = -5.0; a->mag_pos = 0;
    }
    a->knee = (double)a->threshold / 20.0;
    a->gain_slope = a->slope_factor / 100.0;
    a->fixed_gain = kAgcOutScale * pow(10.0, a->knee * (a->gain_slope - 1.0));
    a->attack_rise = 1.0 - exp(-1.0 / (a->sample_rate * kAgcAttackRiseTc));
    a->attack_fall = 1.0 - exp(-1.0 / (a->sample_rate * kAgcAttackFallTc));
    a->decay_rise = 1.0 - exp(-1.0 / (a->sample_rate * (double)a->decay * .001 * kAgcDecayRatio));
    a->hang_time = (int)(a->sample_rate * (double)a->decay * .001);
    if (a->use_hang) a->decay_fall = 1.0 - exp(-1.0 / (a->sample_rate * kAgcReleaseTc));
    else a->decay_fall = 1.0 - exp(-1.0 / (a->sample_rate * (double)a->decay * .001));
    a->delay_samples = (int)(a->sample_rate * kAgcDelayTc);
    a->window_samples = (int)(a->sample_rate * kAgcWindowTc);
    if (a->delay_samples >= PO_AGC_MAX_DELAY_BUF - 1) a->delay_samples = PO_AGC_MAX_DELAY_BUF - 1;
}

void po_agc_set_mode(po_agc *a, int mode, int threshold)
{
    int decay = 200;
    a->mode = mode;
    if (mode == 1) decay = 100; else if (mode == 3) decay = 500; else if (mode == 2) decay = 250; else if (mode == 4) decay = 2000;
    po_agc_set_parameters(a, 0, threshold, 0, decay);
}

po_agc *po_agc_new(double sample_rate)
{
    po_agc *a = (po_agc *)calloc(1, sizeof(*a));
    a->step_rate = sample_rate;
    a->sample_rate = 100.0; /* "Trigger init on first call to setup" */
    po_agc_set_mode(a, 0, 1);
    return a;
}
void po_agc_free(po_agc *a) { free(a); }

void po_agc_process(po_agc *a, const double *in, double *out, int n)
{
    if (a->mode == 0) {
        for (int i = 0; i < 2 * n; i++) out[i] = a->manual_gain * in[i];
        return;
    }
    for (int i = 0; i < n; i++) {
        const double ir = in[2 * i], ii = in[2 * i + 1];
        const double dr = a->sig[2 * a->sig_ptr], di = a->sig[2 * a->sig_ptr + 1];
        a->sig[2 * a->sig_ptr] = ir; a->sig[2 * a->sig_ptr + 1] = ii;
        if (++a->sig_ptr >= a->delay_samples) a->sig_ptr = 0;
        double mag = fabs(ir);
        const double mim = fabs(ii);
        if (mim > mag) mag = mim;
        mag = log10(mag + kAgcMinConst) - log10(kAgcMaxAmp);
        double tmp = a->mag[a->mag_pos];
        a->mag[a->mag_pos++] = mag;
        if (a->mag_pos >= a->window_samples) a->mag_pos = 0;
        if (mag > a->peak) a->peak = mag;
        else if (tmp == a->peak) {
            a->peak = -8.0;
            for (int k = 0; k < a->window_samples; k++) if (a->mag[k] > a->peak) a->peak = a->mag[k];
        }
        if (a->peak > a->attack_avg) a->attack_avg = (1.0 - a->attack_rise) * a->attack_avg + a->attack_rise * a->peak;
        else a->attack_avg = (1.0 - a->attack_fall) * a->attack_avg + a->attack_fall * a->peak;
        if (a->use_hang) {
            if (a->peak > a->decay_avg) { a->decay_avg = (1.0 - a->decay_rise) * a->decay_avg + a->decay_rise * a->peak; a->hang_timer = 0; }
            else if (a->hang_timer < a->hang_time) a->hang_timer++;
            else a->decay_avg = (1.0 - a->decay_fall) * a->decay_avg + a->decay_fall * a->peak;
        } else {
            if (a->peak > a->decay_avg) a->decay_avg = (1.0 - a->decay_rise) * a->decay_avg + a->decay_rise * a->peak;
            else a->decay_avg = (1.0 - a->decay_fall) * a->decay_avg + a->decay_fall * a->peak;
        }
        mag = a->attack_avg > a->decay_avg ? a->attack_avg : a->decay_avg;
        double gain;
        if (mag <= a->knee) gain = a->fixed_gain;
        else gain = kAgcOutScale * pow(10.0, mag * (a->gain_slope - 1.0));
        out[2 * i] = dr * gain;
        out[2 * i + 1] = di * gain;
    }
}

/* ------------------------------------------------------------------------------------------------
 * CFractResampler -- pebblelib/fractresampler.cpp
 * ---------------------------------------------------------------------------------------------- */
#define PO_SINC_PERIOD_PTS 10000
#define PO_SINC_PERIODS 28
#define PO_SINC_LENGTH (PO_SINC_PERIODS * PO_SINC_PERIOD_PTS + 1)
struct po_resampler {
    double *sinc, *inbuf;
    int cap;
    double float_time;
};

void po_resampler_sinc_table(double *t) /* fractresampler.cpp:104-118 */
{
    for (int i = 0; i < PO_SINC_LENGTH; i++) {
        const double window = (0.35875 - 0.48829 * cos((PO_TWOPI * i) / (PO_SINC_LENGTH - 1)) + 0.14128 * cos((2.0 * PO_TWOPI * i) / (PO_SINC_LENGTH - 1)) -
                               0.01168 * cos((3.0 * PO_TWOPI * i) / (PO_SINC_LENGTH - 1)));
        const double fi = PO_PI * (double)(i - PO_SINC_LENGTH / 2) / (double)PO_SINC_PERIOD_PTS;
        t[i] = i != PO_SINC_LENGTH / 2 ? window * sin(fi) / fi : 1.0;
    }
}

po_resampler *po_resampler_new(int max_input)
{
    po_resampler *r = (po_resampler *)calloc(1, sizeof(*r));
    r->cap = max_input + PO_SINC_PERIODS;
    r->sinc = (double *)malloc(sizeof(double) * PO_SINC_LENGTH);
    r->inbuf = (double *)calloc((size_t)r->cap * 2, sizeof(double));
    po_resampler_sinc_table(r->sinc);
    r->float_time = 0.0;
    return r;
}
void po_resampler_free(po_resampler *r) { if (r) { free(r->sinc); free(r->inbuf); free(r); } }
double po_resampler_time(const po_resampler *r) { return r->float_time; }

int po_resampler_process(po_resampler *r, int n, double rate, const double *in, double *out)
{
    int integer_time = (int)r->float_time, outsamples = 0;
    const double dt = rate;
    memcpy(r->inbuf + 2 * PO_SINC_PERIODS, in, sizeof(double) * 2 * (size_t)n);
    while (integer_time < n) {
        double ar = 0.0, ai = 0.0;
        for (int i = 1; i <= PO_SINC_PERIODS; i++) {
            const int j = integer_time + i;
            const int sindx = (int)(((double)j - r->float_time) * (double)PO_SINC_PERIOD_PTS);
            ar = ar + (r->inbuf[2 * j] * r->sinc[sindx]);
            ai = ai + (r->inbuf[2 * j + 1] * r->sinc[sindx]);
        }
        out[2 * outsamples] = ar; out[2 * outsamples + 1] = ai;
        outsamples++;
        r->float_time += dt;
        integer_time = (int)r->float_time;
    }
    r->float_time -= (double)n;
    memmove(r->inbuf, r->inbuf + 2 * (size_t)n, sizeof(double) * 2 * PO_SINC_PERIODS);
    return outsamples;
}

/* ------------------------------------------------------------------------------------------------
 * IQBalance, NoiseBlanker, NoiseFilter -- application/{iqbalance,noiseblanker,noisefilter}.cpp
 * ---------------------------------------------------------------------------------------------- */
void po_iq_balance(double gain_factor, double phase_factor, const double *in, double *out, int n)
{
    double t1r = 0, t1i = 0, t2r = 0, t2i = 0;
    const float mu = 0.0025f;
    for (int i = 0; i < n; i++) {
        const double orr = in[2 * i] * gain_factor;
        const double oi = in[2 * i + 1] + (in[2 * i] * phase_factor);
        /* t1 = out + t2 * conj(out) */
        t1r = orr + (t2r * orr + t2i * oi);
        t1i = oi + (t2i * orr - t2r * oi);
        /* t2 = t2 * (1 - mu*1e-6) - (t1*t1) * mu */
        const double sc = 1.0 - mu * 0.000001;
        const double sqr = t1r * t1r - t1i * t1i, sqi = 2.0 * t1r * t1i;
        t2r = t2r * sc - sqr * mu;
        t2i = t2i * sc - sqi * mu;
        out[2 * i] = t1r;
        out[2 * i + 1] = t1i;
    }
}

void po_nb_init(po_nb *b)
{
    memset(b, 0, sizeof(*b));
    b->nb_avg_mag = 1; b->nb2_avg_mag = 1; b->spike_count = 0; /* noiseblanker.cpp:9-15 */
}
void po_nb_enable(po_nb *b, int which)
{
    if (which == 1) { b->spike_count = 0; b->nb_avg_mag = 0; }
    else { b->nb2_avg_mag = 0; b->nb2_avg[0] = b->nb2_avg[1] = 0; }
}
void po_nb1_process(po_nb *b, const double *in, double *out, int n)
{
    const int size = 8, delay = 2, spike = 7;
    const double threshold = 3.3;
    for (int i = 0; i < n; i++) {
        const float mag = (float)sqrt(in[2 * i] * in[2 * i] + in[2 * i + 1] * in[2 * i + 1]);
        b->delay[2 * b->head] = in[2 * i]; b->delay[2 * b->head + 1] = in[2 * i + 1]; /* DelayLine::NewSample */
        b->last = b->head;
        b->head = b->head == 0 ? size - 1 : b->head - 1;
        b->nb_avg_mag = (float)((0.999 * b->nb_avg_mag) + (0.001 * mag));
        if (b->spike_count == 0 && mag > (b->nb_avg_mag * threshold)) b->spike_count = spike;
        if (b->spike_count > 0) {
            out[2 * i] = 0.0; out[2 * i + 1] = 0.0;
            b->spike_count--;
        } else {
            const int nx = (b->last + delay + 0) % size; /* NextDelay(0) */
            out[2 * i] = b->delay[2 * nx]; out[2 * i + 1] = b->delay[2 * nx + 1];
        }
    }
}
void po_nb2_process(po_nb *b, const double *in, double *out, int n)
{
    const double threshold = 3.3;
    for (int i = 0; i < n; i++) {
        const float mag = (float)sqrt(in[2 * i] * in[2 * i] + in[2 * i + 1] * in[2 * i + 1]);
        b->nb2_avg[0] = b->nb2_avg[0] * 0.75 + in[2 * i] * 0.25;
        b->nb2_avg[1] = b->nb2_avg[1] * 0.75 + in[2 * i + 1] * 0.25;
        b->nb2_avg_mag = (float)(0.999 * b->nb2_avg_mag + 0.001 * mag);
        if (mag > (threshold * b->nb2_avg_mag)) { out[2 * i] = b->nb2_avg[0]; out[2 * i + 1] = b->nb2_avg[1]; }
        else { out[2 * i] = in[2 * i]; out[2 * i + 1] = in[2 * i + 1]; }
    }
}

void po_anf_init(po_anf *a) { memset(a, 0, sizeof(*a)); }
void po_anf_process(po_anf *a, const double *in, double *out, int n)
{
    const int size = 512, delay = 64, taps = 45;
    const double rate = 0.01, leakage = 0.00001;
    const double scl1 = 1.0 - rate * leakage;
    for (int i = 0; i < n; i++) {
        const double inr = in[2 * i], ini = in[2 * i + 1]; /* the reference's in and out are distinct buffers */
        a->delay[2 * a->head] = inr; a->delay[2 * a->head + 1] = ini;
        a->last = a->head;
        a->head = a->head == 0 ? size - 1 : a->head - 1;
        double sosr = 0, sosi = 0, accr = 0, acci = 0;
        for (int j = 0; j < taps; j++) {
            const int nx = (a->last + delay + j) % size;
            const double dr = a->delay[2 * nx], di = a->delay[2 * nx + 1];
            sosr = sosr + (dr * dr); sosi = sosi + (di * di);
            accr = accr + a->coeff[2 * j] * dr; acci = acci + a->coeff[2 * j + 1] * di;
        }
        out[2 * i] = accr * 1.25; out[2 * i + 1] = acci * 1.25;
        double er = inr - accr, ei = ini - acci;
        er = er * (rate / (sosr + 1e-10));
        ei = ei * (rate / (sosi + 1e-10));
        for (int j = 0; j < taps; j++) {
            const int nx = (a->last + delay + j) % size;
            a->coeff[2 * j] = a->coeff[2 * j] * scl1 + er * a->delay[2 * nx];
            a->coeff[2 * j + 1] = a->coeff[2 * j + 1] * scl1 + ei * a->delay[2 * nx + 1];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * SignalStrength::fdEstimate -- application/signalstrength.cpp:287-380, pebblelib/db.h
 * ---------------------------------------------------------------------------------------------- */
static int po_qbound(int lo, int v, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static double po_power_to_db(double p) { return p == 0 ? -120.0 : 10 * log10(p); } /* DB::minDb = -120 (db.cpp) */
static double po_db_clip(double db) { return db < -120.0 ? -120.0 : (db > 0.0 ? 0.0 : db); }

double po_fd_estimate(const double *spectrum, int bins, uint32_t rate, float bp_lo, float bp_hi, double mixer_freq, double *out)
{
    double peak_pwr = 0, total = 0, noise_total = 0;
    const double bin_width = rate / (uint32_t)bins; /* integer division, then widened */
    const int center = bins / 2;
    int mixer_bin = (int)(center + (mixer_freq / bin_width));
    mixer_bin = po_qbound(0, mixer_bin, bins);
    int lo_bin = (int)(mixer_bin + (bp_lo / bin_width));
    lo_bin = po_qbound(0, lo_bin, bins);
    int hi_bin = (int)(mixer_bin + (bp_hi / bin_width));
    hi_bin = po_qbound(0, hi_bin, bins);
    const int bp_bins = hi_bin - lo_bin;
    const int nlo = po_qbound(0, lo_bin - bp_bins, bins), nhi = po_qbound(0, hi_bin + bp_bins, bins);
    int noise_bins = 0;
    for (int i = 0; i < bins; i++) {
        if (i < nlo) continue;
        const double pwr = pow(10, spectrum[i] / 10.0);
        if (i >= lo_bin && i <= hi_bin) {
            total += pwr;
            if (pwr > peak_pwr) peak_pwr = pwr;
        } else if (i >= nlo && i <= nhi) {
            noise_total += pwr;
            noise_bins++;
        }
        if (i >= nhi) break;
    }
    const double avg_pwr = total / bp_bins, noise_avg = noise_total / noise_bins;
    const double peak_db = po_db_clip(po_power_to_db(peak_pwr)), avg_db = po_db_clip(po_power_to_db(avg_pwr));
    const double floor_db = po_db_clip(po_power_to_db(noise_avg));
    double snr = (noise_avg == 0 || peak_pwr == 0) ? -120.0 : 10.0 * log10(peak_pwr / noise_avg);
    snr = snr < 0.0 ? 0.0 : (snr > 120.0 ? 120.0 : snr);
    if (out) { out[0] = peak_db; out[1] = avg_db; out[2] = snr; out[3] = floor_db; }
    return avg_db;
}

struct po_receiver {
    uint32_t fs, n;
    int mode;
    po_mixer mixer;
    po_decimator *dec, *dec_wfm;
    int demod_rate, wfm_rate; /* int members, receiver.h:165-166: fractional rates truncate */
    po_spectrum *spec;
    po_fastfir *bp;
    po_demod_am am;
    po_demod_nfm nfm;
    po_demod_sam sam;
    po_demod_wfm wfm;
    po_rds_group *rds_polled; uint8_t *rds_changed; int n_rds, cap_rds; /* what Demod::fmStereo took from the queue, demod.cpp:207-219 */
    double *mixed, *working, *samplebuf, *bpout, *demodout;
    uint32_t samplebuf_len;
    int cond_flags; double iq_gain, iq_phase; po_iir dc; po_nb nb; po_anf anf; int anf_on;
    double squelch_db, mixer_freq, bp_lo, bp_hi; double *last_spec; uint32_t spec_bins; /* squelch, receiver.cpp:704-707 */
    double *cond;            /* conditioned input frame */
    po_agc *agc;             /* receiver.cpp:264 AGC(m_demodSampleRate, m_demodFrames) */
    po_resampler *resamp;    /* receiver.cpp:184 */
    uint32_t audio_rate;     /* 0: resampRate == 1 branch (copyCPX) */
};

po_receiver *po_receiver_new(uint32_t fs, uint32_t n, uint32_t spectrum_bins, uint32_t fastfir_fft, uint32_t fastfir_taps)
{
    po_receiver *r = (po_receiver *)calloc(1, sizeof(*r));
    r->fs = fs; r->n = n; r->mode = PO_AM;
    po_mixer_init(&r->mixer, (double)fs);                                /* receiver.cpp:156 */
    r->dec = po_decimator_new();
    r->demod_rate = (int)po_decimator_build(r->dec, fs, 30000, 0);       /* :194-195 */
    r->dec_wfm = po_decimator_new();
    r->wfm_rate = (int)po_decimator_build(r->dec_wfm, fs, 200000, 0);    /* :212-213 */
    if (spectrum_bins) r->spec = po_spectrum_new(spectrum_bins, n, 0, 0); /* :221, signalspectrum.cpp:58 */
    r->squelch_db = -120.0; /* DB::minDb, receiverwidget.cpp:82 */
    r->spec_bins = spectrum_bins;
    if (spectrum_bins) r->last_spec = (double *)calloc(spectrum_bins, sizeof(double));
    r->bp = po_fastfir_new(fastfir_fft ? fastfir_fft : 2048, fastfir_taps ? fastfir_taps : 1025); /* :261 */
    po_demod_am_init(&r->am, (double)r->demod_rate);                     /* :228, demod.cpp:62 */
    po_demod_sam_init(&r->sam, (double)r->demod_rate);                   /* demod.cpp:63 */
    po_demod_nfm_init(&r->nfm, (double)r->demod_rate);                   /* demod.cpp:64 */
    po_demod_wfm_init(&r->wfm, (double)r->wfm_rate);                     /* demod.cpp:65 */
    r->mixed = (double *)calloc((size_t)n * 2, sizeof(double));
    r->working = (double *)calloc((size_t)n * 2, sizeof(double));
    r->samplebuf = (double *)calloc((size_t)n * 2, sizeof(double));
    r->bpout = (double *)calloc((size_t)n * 2 + 2 * (size_t)(fastfir_fft ? fastfir_fft : 2048), sizeof(double));
    r->demodout = (double *)calloc((size_t)n * 2 + 2 * (size_t)(fastfir_fft ? fastfir_fft : 2048), sizeof(double));
    r->cond = (double *)calloc((size_t)n * 2, sizeof(double));
    po_iir_init_hp(&r->dc, 10, 0.7071, (double)fs); /* dcremoval.cpp:5-9 */
    po_nb_init(&r->nb);
    po_anf_init(&r->anf);
    r->iq_gain = 1; r->iq_phase = 0;
    r->agc = po_agc_new((double)r->demod_rate);
    r->resamp = po_resampler_new((int)n + (int)(fastfir_fft ? fastfir_fft : 2048));
    r->audio_rate = 0;
    return r;
}

void po_receiver_free(po_receiver *r)
{
    if (!r) return;
    po_decimator_free(r->dec); po_decimator_free(r->dec_wfm);
    po_spectrum_free(r->spec); po_fastfir_free(r->bp);
    free(r->mixed); free(r->working); free(r->samplebuf); free(r->bpout); free(r->demodout);
    po_agc_free(r->agc); po_resampler_free(r->resamp); free(r->cond); free(r->last_spec);
    po_demod_wfm_free(&r->wfm); free(r->rds_polled); free(r->rds_changed);
    free(r);
}

void po_receiver_set_mode(po_receiver *r, int mode) { r->mode = mode; r->samplebuf_len = 0; } /* :640-655 */
void po_receiver_set_mixer(po_receiver *r, double f) { r->mixer_freq = f; po_mixer_set_frequency(&r->mixer, f); } /* :709-716 */
int po_receiver_set_filter(po_receiver *r, double lo, double hi) /* :658-664, bandpassfilter.cpp:38-46 */
{
    int rc = po_fastfir_setup(r->bp, (float)lo, (float)hi, 0, (double)(uint32_t)r->demod_rate);
    r->bp_lo = lo; r->bp_hi = hi; /* BandPassFilter::lowFreq/highFreq */
    if (r->mode == PO_AM) po_demod_am_set_bandwidth(&r->am, hi - lo); /* demod.cpp:230-239 */
    return rc;
}
double po_receiver_demod_rate(const po_receiver *r, int wfm) { return wfm ? r->wfm_rate : r->demod_rate; }
/* Demod::fmStereo, demod.cpp:196-226: one getNextRdsGroupData per processDataStereo call */
static void po_receiver_poll_rds(po_receiver *r)
{
    po_rds_group g; int changed = 0;
    if (!po_demod_wfm_next_rds_group(&r->wfm, &g, &changed)) return;
    if (r->n_rds == r->cap_rds) {
        r->cap_rds = r->cap_rds ? 2 * r->cap_rds : 64;
        r->rds_polled = (po_rds_group *)realloc(r->rds_polled, (size_t)r->cap_rds * sizeof(po_rds_group));
        r->rds_changed = (uint8_t *)realloc(r->rds_changed, (size_t)r->cap_rds);
    }
    r->rds_polled[r->n_rds] = g; r->rds_changed[r->n_rds++] = (uint8_t)changed;
}
int po_receiver_rds_polled(po_receiver *r, po_rds_group *g, uint8_t *changed, int cap)
{
    const int n = r->n_rds < cap ? r->n_rds : cap;
    if (g) memcpy(g, r->rds_polled, (size_t)n * sizeof(po_rds_group));
    if (changed) memcpy(changed, r->rds_changed, (size_t)n);
    r->n_rds = 0;
    return n;
}
uint32_t po_receiver_dec_stages(const po_receiver *r, int wfm)
{
    return po_decimator_dec_by2_stages(wfm ? r->dec_wfm : r->dec);
}

uint32_t po_receiver_process(po_receiver *r, const double *in, uint32_t n, double *audio, double *spectrum_db)
{
    /* :814-823 DC removal, IQ balance, noise blankers -- each returns its own buffer when enabled */
    if (r->cond_flags) {
        memcpy(r->cond, in, sizeof(double) * 2 * (size_t)n);
        if (r->cond_flags & 1) po_iir_process_cpx(&r->dc, (int)n, r->cond, r->cond);
        if (r->cond_flags & 2) po_iq_balance(r->iq_gain, r->iq_phase, r->cond, r->cond, (int)n);
        if (r->cond_flags & 4) po_nb1_process(&r->nb, r->cond, r->cond, (int)n);
        if (r->cond_flags & 8) po_nb2_process(&r->nb, r->cond, r->cond, (int)n);
        in = r->cond;
    }
    /* :826 SignalSpectrum::unprocessed (timer gate forced open: every frame) */
    if (r->spec && (spectrum_db || r->squelch_db > -120.0)) {
        po_spectrum_process(r->spec, in, n, r->last_spec); /* SignalSpectrum::getUnprocessed: the latest frame's */
        if (spectrum_db) memcpy(spectrum_db, r->last_spec, sizeof(double) * r->spec_bins);
    }
    const double *next = in;
    if (po_mixer_process(&r->mixer, in, r->mixed, n)) next = r->mixed; /* :867 / :910 */
    int wfm = (r->mode == PO_FMM || r->mode == PO_FMS);
    uint32_t cnt = po_decimator_process(wfm ? r->dec_wfm : r->dec, next, r->working, n); /* :868 / :911 */
    for (uint32_t i = 0; i < cnt; i++) { /* :873-875 / :922-924 */
        if (r->samplebuf_len < r->n) {
            r->samplebuf[2 * r->samplebuf_len] = r->working[2 * i];
            r->samplebuf[2 * r->samplebuf_len + 1] = r->working[2 * i + 1];
        }
        r->samplebuf_len++;
    }
    if (r->samplebuf_len < r->n) return 0; /* :878-879 / :927-928 */
    uint32_t ns = r->n;
    r->samplebuf_len = 0;
    if (wfm && r->squelch_db > -120.0 && r->last_spec) { /* :891-897 */
        if (po_fd_estimate(r->last_spec, (int)r->spec_bins, r->fs, -100000, 100000, r->mixer_freq, NULL) < r->squelch_db) return 0;
    }
    if (wfm) {
        /* :896 Demod::processBlock */
        /* demod.cpp:113-122: dmFMM -> fmMono, dmFMS -> fmStereo */
        if (!r->audio_rate) {
            if (r->mode == PO_FMS) { po_demod_wfm_process_stereo(&r->wfm, r->samplebuf, audio, (int)ns); po_receiver_poll_rds(r); }
            else po_demod_wfm_process_mono(&r->wfm, r->samplebuf, audio, (int)ns);
            return ns;
        }
        if (r->mode == PO_FMS) { po_demod_wfm_process_stereo(&r->wfm, r->samplebuf, r->demodout, (int)ns); po_receiver_poll_rds(r); }
        else po_demod_wfm_process_mono(&r->wfm, r->samplebuf, r->demodout, (int)ns);
        /* :901, :1000-1001 resampRate = m_demodWfmSampleRate / m_audioOutRate */
        return (uint32_t)po_resampler_process(r->resamp, (int)ns, ((double)r->wfm_rate * 1.0) / ((double)r->audio_rate * 1.0), r->demodout, audio);
    }
    /* :935-938 gain restore 10^(2*stages/20) */
    double g = pow(10, (double)(po_decimator_dec_by2_stages(r->dec) * 2) / 20.0);
    for (uint32_t i = 0; i < 2 * ns; i++) r->samplebuf[i] = r->samplebuf[i] * g;
    /* :950 band-pass.  With the stock 2048/1025 sizes and n=2048 this returns exactly n samples per
     * call; the parametrised 8192/4097 variant returns 0 or 4096 alternately, and the chain below then
     * runs on however many samples came out (the stock BandPassFilter::process ignores the count,
     * bandpassfilter.cpp:53-56, which only works for the stock sizes). */
    int nb = po_fastfir_process(r->bp, (int)ns, r->samplebuf, r->bpout);
    if (nb <= 0) return 0;
    if (r->squelch_db > -120.0 && r->last_spec) { /* :959-965: m_avgDb < m_squelchDb -> return, nothing behind it runs */
        if (po_fd_estimate(r->last_spec, (int)r->spec_bins, r->fs, (float)r->bp_lo, (float)r->bp_hi, r->mixer_freq, NULL) < r->squelch_db) return 0;
    }
    /* :968-971 "Tune only mode, no demod or output": clearCPX(m_audioBuf) and return -- before NoiseFilter, AGC, Demod, the
     * resampler and the audio callback, so nothing is delivered and their states stay where they were */
    if (r->mode == PO_NONE) return 0;
    /* :974 NoiseFilter (ANF) */
    if (r->anf_on) po_anf_process(&r->anf, r->bpout, r->bpout, nb);
    /* :983 AGC, written back over the band-pass output buffer */
    po_agc_process(r->agc, r->bpout, r->bpout, nb);
    /* :987 demod */
    double *dst = r->audio_rate ? r->demodout : audio;
    if (r->mode == PO_AM) po_demod_am_process(&r->am, r->bpout, dst, nb);
    else if (r->mode == PO_SAM) po_demod_sam_process(&r->sam, r->bpout, dst, nb);
    else if (r->mode == PO_FMN) po_demod_nfm_process(&r->nfm, r->bpout, dst, nb);
    else memcpy(dst, r->bpout, (size_t)nb * 2 * sizeof(double)); /* SSB/CW/DIG/DSB pass-through, demod.cpp:127-138 */
    if (!r->audio_rate) return (uint32_t)nb;
    /* :994, :1000-1001 */
    return (uint32_t)po_resampler_process(r->resamp, nb, ((double)r->demod_rate * 1.0) / ((double)r->audio_rate * 1.0), r->demodout, audio);
}

void po_receiver_set_conditioners(po_receiver *r, int flags, double gain_factor, double phase_factor)
{
    if ((flags & 4) && !(r->cond_flags & 4)) po_nb_enable(&r->nb, 1);
    if ((flags & 8) && !(r->cond_flags & 8)) po_nb_enable(&r->nb, 2);
    r->cond_flags = flags; r->iq_gain = gain_factor; r->iq_phase = phase_factor;
}
void po_receiver_set_anf(po_receiver *r, int on) { r->anf_on = on; }
void po_receiver_set_squelch(po_receiver *r, double squelch_db) { r->squelch_db = squelch_db; } /* :704-707 */
void po_receiver_set_agc(po_receiver *r, int mode, int threshold) { po_agc_set_mode(r->agc, mode, threshold); }
void po_receiver_set_audio_rate(po_receiver *r, uint32_t audio_rate) { r->audio_rate = audio_rate; }
