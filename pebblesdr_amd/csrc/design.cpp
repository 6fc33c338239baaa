// design.cpp -- host-side fp64 filter/chain design (see design.h).
#include "design.h"
#include <cmath>
#include "hb_taps.inc"
#include "dc_taps.inc"

namespace pg {
namespace design {

// The reference's ladder tries cic3, hb11, hb15 ... hb59 in that order and takes the first whose
// wPass protects the signal at the current rate; a repeat of the previous pick widens that stage's
// stride instead of adding a stage (pebblelib/decimator.cpp:74-146).
Chain build_chain(uint32_t fs_in, uint32_t protect_bw, uint32_t fs_out_min)
{
    Chain c;
    c.rate = (float)fs_in;
    const float floor_rate = (float)(fs_out_min ? fs_out_min : 15000u);  // decimator.h:245
    const double bw = (double)protect_bw;
    while (c.rate > floor_rate) {
        int pick = -1;
        for (int k = 0; k < PEBBLE_HB_NDESIGNS && pick < 0; k++)
            if ((double)c.rate >= bw / pebble_hb_designs[k].wpass) pick = k;
        if (pick < 0) break;
        c.dec_by2++;
        c.total *= 2;
        if (!c.stages.empty() && c.stages.back().ntaps == pebble_hb_designs[pick].ntaps)
            c.stages.back().stride *= 2;
        else
            c.stages.push_back(Stage{pick, pebble_hb_designs[pick].ntaps, 2u});
        c.rate /= 2;
    }
    return c;
}

const double *halfband_taps(int design) { return pebble_hb_designs[design].h; }

DcChain downconvert_chain(double in_rate, double max_bw, bool simple)
{
    DcChain c;
    double f = in_rate;
    if (simple) {  // SetDataRateSimple, downconvert.cpp:224-229
        while (f > 400000.0) {
            c.stages.push_back(PEBBLE_DC_NDESIGNS - 1);
            f /= 2.0;
        }
    } else {
        const double last = pebble_dc_designs[PEBBLE_DC_NDESIGNS - 1].max_a - pebble_dc_designs[PEBBLE_DC_NDESIGNS - 1].max_b;
        while (f > (max_bw / last) && f > (7900.0 * 2.0)) {  // :152, MIN_OUTPUT_RATE :58
            for (int k = 0; k < PEBBLE_DC_NDESIGNS; k++)
                if (f >= max_bw / (pebble_dc_designs[k].max_a - pebble_dc_designs[k].max_b)) {  // the ladder, :154-203
                    c.stages.push_back(k);
                    break;
                }
            f /= 2.0;
        }
    }
    c.out_rate = f;
    return c;
}
int downconvert_stage_taps(int design) { return pebble_dc_designs[design].ntaps; }
std::vector<double> downconvert_stage_response(int design)
{
    if (design == 0) return {0.125, 0.375, 0.375, 0.125};  // CCicN3DecimateBy2::DecBy2, :524-526
    const int T = pebble_dc_designs[design].ntaps;
    std::vector<double> h(pebble_dc_designs[design].h, pebble_dc_designs[design].h + T);
    if (design != 1) h[0] += pebble_dc_designs[design].h[0];  // CHalfBandDecimateBy2::DecBy2: the accumulator starts from tap 0, the loop adds it again
    return h;
}

void mixer_amplitudes(float *tab, int n, float *a_inf)
{
    double a = 1.0;
    for (int i = 0; i < n; i++) {
        tab[i] = (float)a;
        a = a * (1.95 - a * a);
    }
    *a_inf = (float)std::sqrt(0.95);
}

void fft(std::vector<std::complex<double>> &x, int dir)
{
    const size_t n = x.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(x[i], x[j]);
    }
    const double sgn = dir >= 0 ? -1.0 : 1.0;
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len >> 1;
        for (size_t k = 0; k < half; k++) {
            const double ang = sgn * kTwoPi * (double)k / (double)len;
            const std::complex<double> w(std::cos(ang), std::sin(ang));
            for (size_t s = k; s < n; s += len) {
                const std::complex<double> t = x[s + half] * w;
                x[s + half] = x[s] - t;
                x[s] += t;
            }
        }
    }
}

bool fastfir_design(uint32_t fft_size, uint32_t fir_size, double lo, double hi, double offset, double fs,
                    std::vector<std::complex<double>> &H)
{
    lo += offset;
    hi += offset;
    if (lo >= hi || lo >= fs / 2.0 || lo <= -fs / 2.0 || hi >= fs / 2.0 || hi <= -fs / 2.0) return false;
    const double nfl = lo / fs, nfh = hi / fs;
    const double nfc = (nfh - nfl) / 2.0;            // prototype low-pass cutoff
    const double nfs = kTwoPi * (nfh + nfl) / 2.0;   // heterodyne to the band centre
    const double centre = 0.5 * (double)(fir_size - 1);
    std::vector<std::complex<double>> h(fft_size, std::complex<double>(0, 0));
    for (uint32_t i = 0; i < fir_size; i++) {
        // Blackman-Nuttall window, fastfir.cpp:102-111
        const double w = 0.3635819 - 0.4891775 * std::cos((kTwoPi * i) / (fir_size - 1)) +
                         0.1365995 * std::cos((2.0 * kTwoPi * i) / (fir_size - 1)) -
                         0.0106411 * std::cos((3.0 * kTwoPi * i) / (fir_size - 1));
        const double x = (double)i - centre;
        const double z = ((double)i == centre) ? 2.0 * nfc : std::sin(kTwoPi * x * nfc) / (kPi * x) * w;
        h[i] = std::complex<double>(z * std::cos(nfs * x) / (double)fft_size, z * std::sin(nfs * x) / (double)fft_size);
    }
    fft(h, +1);
    H.swap(h);
    return true;
}

static double bessel_i0(double x)  // fir.cpp:494-512, series to 1e-9 relative
{
    const double x2 = x / 2.0;
    double sum = 1.0, ds = 1.0, di = 1.0;
    do {
        double t = x2 / di;
        t *= t;
        ds *= t;
        sum += ds;
        di += 1.0;
    } while (ds >= 1e-9 * sum);
    return sum;
}

std::vector<double> fir_lowpass(int force_taps, double scale, double astop, double fpass, double fstop, double fs)
{
    const double nfp = fpass / fs, nfs = fstop / fs, nfc = (nfs + nfp) / 2.0;
    double beta;
    if (astop < 20.96) beta = 0;
    else if (astop >= 50.0) beta = .1102 * (astop - 8.71);
    else beta = .5842 * std::pow(astop - 20.96, 0.4) + .07886 * (astop - 20.96);
    int ntaps = (int)((astop - 8.0) / (2.285 * kTwoPi * (nfs - nfp)) + 1);
    if (ntaps > 75) ntaps = 75;
    if (ntaps < 3) ntaps = 3;
    if (force_taps) ntaps = force_taps;
    std::vector<double> h(ntaps);
    const double centre = .5 * (double)(ntaps - 1), izb = bessel_i0(beta);
    for (int n = 0; n < ntaps; n++) {
        double x = (double)n - centre;
        const double c = ((double)n == centre) ? 2.0 * nfc : std::sin(kTwoPi * x * nfc) / (kPi * x);
        x = ((double)n - ((double)ntaps - 1.0) / 2.0) / (((double)ntaps - 1.0) / 2.0);
        h[n] = scale * c * bessel_i0(beta * std::sqrt(1 - x * x)) / izb;
    }
    return h;
}

static Biquad rbj(double f0, double q, double fs, int kind)
{
    const double w0 = kTwoPi * f0 / fs, alpha = std::sin(w0) / (2.0 * q), A = 1.0 / (1.0 + alpha);
    Biquad b;
    b.a1 = A * (-2.0 * std::cos(w0));
    b.a2 = A * (1.0 - alpha);
    if (kind == 0) {  // low-pass, iir.cpp:88-103
        b.b0 = A * ((1.0 - std::cos(w0)) / 2.0);
        b.b1 = A * (1.0 - std::cos(w0));
        b.b2 = b.b0;
    } else if (kind == 3) {  // band-pass, iir.cpp:131-146
        b.b0 = A * alpha;
        b.b1 = 0.0;
        b.b2 = A * -alpha;
    } else if (kind == 2) {  // high-pass, iir.cpp:110-125
        b.b0 = A * ((1.0 + std::cos(w0)) / 2.0);
        b.b1 = -A * (1.0 + std::cos(w0));
        b.b2 = b.b0;
    } else {          // band-reject, iir.cpp:152-167
        b.b0 = A;
        b.b1 = A * (-2.0 * std::cos(w0));
        b.b2 = A;
    }
    return b;
}
Biquad biquad_lowpass(double f0, double q, double fs) { return rbj(f0, q, fs, 0); }
Biquad biquad_notch(double f0, double q, double fs) { return rbj(f0, q, fs, 1); }
Biquad biquad_highpass(double f0, double q, double fs) { return rbj(f0, q, fs, 2); }
Biquad biquad_bandpass(double f0, double q, double fs) { return rbj(f0, q, fs, 3); }

// HILBLP_H, demod_wfm.cpp:79-98 (data): the 61-tap symmetric low-pass prototype the Hilbert pair is shifted from; first 31 taps
static const double kHilbHalf[31] = {
    -0.000389631665953405, 0.000115430826670992, 0.000945331102222503, 0.001582460677684605,
    0.001370803713784687, -0.000000000000000002, -0.002077413537668161, -0.003656132107176520,
    -0.003372610825000167, -0.000649815020884706, 0.003583263233560064, 0.006997162933343487,
    0.006990985399916562, 0.002383133886438500, -0.005324501734543406, -0.012092135317628615,
    -0.013212201698221963, -0.006168904735839018, 0.007082277142635906, 0.020017841466263672,
    0.024271835962039127, 0.014255112728911837, -0.008597071392140753, -0.034478282954624850,
    -0.048147195828726633, -0.035409729589347565, 0.009623663461671806, 0.080084441681677138,
    0.157278883310078170, 0.217148915611638180, 0.239688166538436750
};
WfmPilotDesign wfm_pilot_design(double fs)
{
    WfmPilotDesign d;
    for (int n = 0; n < 61; n++) {  // CFir::GenerateHBFilter(42000), fir.cpp:212-243: the prototype shifted up, I and Q taps
        const double h = kHilbHalf[n <= 30 ? n : 60 - n];
        const double a = (kTwoPi * 42000.0 / fs) * ((double)n - 30.0);
        d.hilb[n] = 2.0 * h * std::cos(a);
        d.hilb[61 + n] = 2.0 * h * std::sin(a);
    }
    d.bp = biquad_bandpass(19000.0, 500, fs);  // demod_wfm.cpp:171
    const double norm = kTwoPi / fs;
    d.nco_freq0 = -19000.0;                    // :374, as written: hertz until the loop's first clamp
    d.nco_lo = (d.nco_freq0 - 20.0) * norm;    // PILOTPLL_RANGE
    d.nco_hi = (d.nco_freq0 + 20.0) * norm;
    d.alpha = 2.0 * .707 * 10.0 * norm;        // PILOTPLL_ZETA, PILOTPLL_BW
    d.beta = (d.alpha * d.alpha) / (4.0 * .707 * .707);
    d.err_alpha = 1.0 - std::exp(-1.0 / (fs * .5));  // LOCK_TIMECONST
    d.phase_adjust = -7.267e-6 * fs + 3.677;   // PHASE_ADJ_M, PHASE_ADJ_B, :60-61, :161
    return d;
}

RdsDesign rds_design(double fs)
{
    RdsDesign d;
    const DcChain c = downconvert_chain(fs, 8000.0, false);   // demod_wfm.cpp:187
    d.rate = c.out_rate;
    d.stages = c.stages;
    d.osc_turns = 57000.0 / fs;                                // :188 with downconvert.cpp:103-108
    d.lp = fir_lowpass(0, 1.0, 40.0, 2400.0, 1.3 * 2400.0, d.rate);  // :496
    const double norm = kTwoPi / d.rate;
    d.nco_lo = (0.0 - 12.0) * norm;                            // RDSPLL_RANGE, :500-501
    d.nco_hi = (0.0 + 12.0) * norm;
    d.alpha = 2.0 * .707 * 1.0 * norm;                         // RDSPLL_ZETA, RDSPLL_BW
    d.beta = (d.alpha * d.alpha) / (4.0 * .707 * .707);
    const double bitrate = 57000.0 / 48.0;
    int len = (int)(d.rate / bitrate);                         // m_MatchCoefLength, an int (:507)
    std::vector<double> co(2 * (size_t)len + 1, 0.0);
    for (int i = 0; i <= len; i++) {                           // :508-517 as written: i = 0 goes through 1 / infinity
        const double t = (double)i / d.rate, x = t * bitrate, x64 = 64.0 * x;
        const double v = .75 * std::cos(2.0 * kTwoPi * x) * ((1.0 / (1.0 / x - x64)) - (1.0 / (9.0 / x - x64)));
        co[(size_t)(i + len)] = v;
        co[(size_t)(len - i)] = -v;
    }
    len *= 2;                                                  // :518; InitConstFir clips at MAX_NUMCOEF (fir.cpp:180-183)
    if (len > 75) len = 75;
    d.matched.assign(co.begin(), co.begin() + len);
    d.bitsync = biquad_bandpass(bitrate, 500, d.rate);         // :522
    return d;
}
void oscillator_amplitudes(double *tab, int n)
{
    double a = 1.0;
    for (int i = 0; i < n; i++) {
        tab[i] = a;
        a = a * (1.95 - a * a);
    }
}

double blackman_harris(uint32_t n, std::vector<double> &w)
{
    // float constants and a float 2*pi, exactly as the reference declares them (windowfunction.cpp:49-51,218-222)
    const float two_pi = (float)kTwoPi, a0 = 0.35875F, a1 = 0.48829F, a2 = 0.14128F, a3 = 0.01168F;
    w.resize(n);
    double sum = 0;
    const int N = (int)n;
    for (int i = 0; i < N; i++) {
        w[i] = a0 - a1 * std::cos(two_pi * (i + 0.5) / N) + a2 * std::cos(2.0 * two_pi * (i + 0.5) / N) -
               a3 * std::cos(3.0 * two_pi * (i + 0.5) / N);
        sum += w[i];
    }
    return sum / N;
}

M2 m2_mul(const M2 &x, const M2 &y)
{
    return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
M2 m2_pow(M2 x, uint64_t e)
{
    M2 r{1, 0, 0, 1};
    while (e) {
        if (e & 1) r = m2_mul(r, x);
        x = m2_mul(x, x);
        e >>= 1;
    }
    return r;
}
double spectral_radius(const M2 &x)
{
    const double tr = x.a + x.d, det = x.a * x.d - x.b * x.c, disc = tr * tr / 4 - det;
    if (disc < 0) return std::sqrt(det);
    const double r = std::sqrt(disc);
    return std::fmax(std::fabs(tr / 2 + r), std::fabs(tr / 2 - r));
}

bool cascade_impulse(const std::vector<double> &fir_prefix, const double *one_pole_avg_a, const std::vector<Biquad> &biquads, double tol,
                     int max_len, std::vector<double> &h)
{
    const int n = max_len + 4096;  // run well past max_len so the tail sum is known
    std::vector<double> y((size_t)n, 0.0);
    if (fir_prefix.empty()) y[0] = 1.0;
    else for (size_t i = 0; i < fir_prefix.size() && i < (size_t)n; i++) y[i] = fir_prefix[i];
    if (one_pole_avg_a) {
        const double a = *one_pole_avg_a;
        double s0 = 0;
        for (int i = 0; i < n; i++) { s0 = (1.0 - a) * s0 + a * y[i]; y[i] = s0 * 2.0; }
    }
    for (const Biquad &q : biquads) {
        double w1 = 0, w2 = 0;
        for (int i = 0; i < n; i++) {
            const double w0 = y[i] - q.a1 * w1 - q.a2 * w2;
            y[i] = q.b0 * w0 + q.b1 * w1 + q.b2 * w2;
            w2 = w1;
            w1 = w0;
        }
    }
    double total = 0;
    for (double v : y) total += std::fabs(v);
    if (!(total > 0)) return false;
    // residual beyond the simulated window: bounded by the geometric decay of the last stretch
    double tail = 0;
    for (int i = n - 1; i >= max_len; i--) tail += std::fabs(y[i]);
    double last = 0;
    for (int i = n - 256; i < n; i++) last += std::fabs(y[i]);
    if (last > tol * total * 1e-3) return false;  // still ringing at the end of the window: slow poles
    int len = max_len;
    if (tail > tol * total) return false;
    while (len > 1 && tail + std::fabs(y[len - 1]) <= tol * total) { tail += std::fabs(y[len - 1]); len--; }
    h.assign(y.begin(), y.begin() + len);
    return true;
}

}  // namespace design
}  // namespace pg
