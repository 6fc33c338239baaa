// design.h -- host-side (fp64) parameter design for the receive chain: which decimation stages,
// which taps, which biquads.  Runs once per control change, never per sample; the results are
// rounded to fp32 and uploaded.  Each routine names the reference code whose numbers it must match.
#pragma once
#include <complex>
#include <cstdint>
#include <vector>

namespace pg {
namespace design {

constexpr double kPi = 3.14159265358979323846264338328;     // pebblelib/cpx.h:16
constexpr double kTwoPi = 6.28318530717958647692528676656;  // pebblelib/cpx.h:18

struct Stage {
    int design;       // index into the halfband design table (hb_taps.inc)
    int ntaps;        // 0 => CIC3
    uint32_t stride;  // decimate-by after merging identical consecutive picks
};
struct Chain {
    std::vector<Stage> stages;
    uint32_t dec_by2 = 0;   // Decimator::decBy2Stages()
    uint32_t total = 1;     // D
    float rate = 0;         // achieved rate (float, as decimator.h:251)
};
// Decimator::buildDecimationChain, pebblelib/decimator.cpp:64-149
Chain build_chain(uint32_t fs_in, uint32_t protect_bw, uint32_t fs_out_min);
const double *halfband_taps(int design);  // length = ntaps of that design

// CDownConvert::SetDataRate / SetDataRateSimple (pebblelib/downconvert.cpp:139-237): the decimate-by-2 stages from the input rate down,
// each an index into the stage table (dc_taps.inc: 0 = CIC3, 1 = the fixed 11-tap halfband, 2.. = hb15 .. hb51); returns the output rate
struct DcChain {
    std::vector<int> stages;
    double out_rate = 0;
};
DcChain downconvert_chain(double in_rate, double max_bw, bool simple);
int downconvert_stage_taps(int design);          // 0 for the CIC3
// the taps a stage applies, oldest sample first: CIC3 -> (1 3 3 1) / 8 (ending on the pair's odd sample); the fixed 11-tap class -> its
// table; the generic class -> its table with tap 0 counted twice (downconvert.cpp:368-376)
std::vector<double> downconvert_stage_response(int design);

// Mixer amplitude sequence a_0 = 1, a_{n+1} = a_n (1.95 - a_n^2)  (pebblelib/mixer.cpp:65-67); a_inf = sqrt(.95)
void mixer_amplitudes(float *tab, int n, float *a_inf);

// in-place radix-2 FFT, fp64; dir +1: e^{-j}, -1: e^{+j}; unscaled (Accelerate semantics, fftaccelerate.cpp:62,95)
void fft(std::vector<std::complex<double>> &x, int dir);

// CFastFIR::SetupParameters, pebblelib/fastfir.cpp:191-272: frequency-domain H (fft_size), 1/fft_size folded in.
// returns false on "Filter Parameter error" (H untouched).
bool fastfir_design(uint32_t fft_size, uint32_t fir_size, double lo, double hi, double offset, double fs,
                    std::vector<std::complex<double>> &H);

// CFir::InitLPFilter, pebblelib/fir.cpp:246-337 (Kaiser-windowed sinc, clamp 3..75 taps)
std::vector<double> fir_lowpass(int force_taps, double scale, double astop, double fpass, double fstop, double fs);

// CIir::InitLP / InitBR, pebblelib/iir.cpp:88-103,152-167.  Direct form 2: w = x - a1 w1 - a2 w2; y = b0 w + b1 w1 + b2 w2
struct Biquad { double b0, b1, b2, a1, a2; };
Biquad biquad_lowpass(double f0, double q, double fs);
Biquad biquad_notch(double f0, double q, double fs);
Biquad biquad_highpass(double f0, double q, double fs);
Biquad biquad_bandpass(double f0, double q, double fs);   // CIir::InitBP, iir.cpp:131-146

// Demod_WFM's stereo members at demodulator rate fs (demod_wfm.cpp:161-171 setSampleRate, :371-386 initPilotPll): the 61-tap Hilbert
// pair [I taps | Q taps] (InitConstFir(HILB_LENGTH, HILBLP_H) + GenerateHBFilter(42000), fir.cpp:176-243), the pilot band-pass and the
// PLL's constants
struct WfmPilotDesign {
    double hilb[2 * 61];
    Biquad bp;
    double nco_lo, nco_hi, alpha, beta, err_alpha, phase_adjust, nco_freq0;
};
WfmPilotDesign wfm_pilot_design(double fs);

// Demod_WFM's RDS members at demodulator rate fs (demod_wfm.cpp:187-191 setSampleRate, :490-537 initRds): m_RdsDownConvert's chain
// (SetDataRate(fs, 8000); SetFrequency(-57000), whose argument CDownConvert negates: the oscillator turns at +57000 Hz), the 2400 Hz
// low-pass behind it, the PLL's constants, the biphase matched filter and the bit-rate resonator
struct RdsDesign {
    double rate = 0;                 // behind the chain
    std::vector<int> stages;         // indices into the CDownConvert stage table (downconvert_stage_response)
    double osc_turns = 0;            // oscillator increment, turns per input sample
    std::vector<double> lp;          // y[i] = sum_k lp[k] x[i - k]
    std::vector<double> matched;     // y[i] = sum_k matched[k] x[i - k]: the first 2 len of the 2 len + 1 values initRds computes
    Biquad bitsync;
    double nco_lo = 0, nco_hi = 0, alpha = 0, beta = 0;
};
RdsDesign rds_design(double fs);
// |m_Osc1| of CDownConvert's (and Mixer's) oscillator before sample n: a_0 = 1, a_{n+1} = a_n (1.95 - a_n^2) -> sqrt(0.95), in double
void oscillator_amplitudes(double *tab, int n);

// WindowFunction BLACKMANHARRIS, pebblelib/windowfunction.cpp:214-235; returns coherentGain = sum/N
double blackman_harris(uint32_t n, std::vector<double> &w);

// Impulse response of  [optional FIR prefix] -> [one-pole average: s = (1-a) s + a u, y = 2 s] -> [biquads DF2 ...],
// run in fp64 and truncated where the remaining absolute tail sum drops below tol * (total absolute sum).  Used to turn
// fast-decaying IIR cascades into one FIR (every output then depends only on input history: no carried filter state,
// fully parallel).  Returns false if the response is still above tol after max_len samples (slow poles).
bool cascade_impulse(const std::vector<double> &fir_prefix, const double *one_pole_avg_a, const std::vector<Biquad> &biquads, double tol,
                     int max_len, std::vector<double> &h);

// 2x2 real matrix helpers for the chunked recurrence scans (state transition powers)
struct M2 { double a, b, c, d; };
M2 m2_mul(const M2 &x, const M2 &y);
M2 m2_pow(M2 x, uint64_t e);
double spectral_radius(const M2 &x);

}  // namespace design
}  // namespace pg
