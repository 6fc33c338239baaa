/*
 * pebble_oracle.h -- CPU restatement (fp64, scalar C) of PebbleSDR's per-frame IQ receive chain.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared against, and the
 * "port" CPU baseline that bench.py times.  Nothing under pebblesdr_amd/ or include/ may include,
 * link or call it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY PINNING STATUS: the reference (Qt5 + Apple Accelerate, macOS/Windows only) cannot be
 * built in this image without writing stand-in Qt/vDSP headers, which the build rules forbid, so
 * there is no oracle/_ref.  The reference ships no tests and no golden vectors.
 * The oracle is pinned by (1) the one worked known-answer table the reference holds
 * (pebblelib/fft.cpp:363-369, -10 dB tone -> spectrum peak per FFT size), (2) the outputs of
 * the reference itself recorded when it was executed at survey time (chain
 * tables, tap counts, oscillator fixed point, FastFIR pass-band gain, spectrum peaks), and
 * (3) closed-form / independent-implementation cross-checks (numpy.fft, scipy.signal).  Stages with
 * none of these say "parity unpinned" in their tests.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * Complex buffers are interleaved (re, im) doubles == CPX = std::complex<double> (pebblelib/cpx.h:96).
 */
#ifndef PEBBLE_ORACLE_H
#define PEBBLE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- DemodMode numeric values: pebblelib/device_interfaces.h:124-138 ---- */
enum po_demod_mode { PO_AM = 0, PO_SAM, PO_FMN, PO_FMM, PO_FMS, PO_DSB, PO_LSB, PO_USB, PO_CWL, PO_CWU,
                     PO_DIGL, PO_DIGU, PO_NONE };

/* ---- Mixer: pebblelib/mixer.cpp:25-81 ---- */
typedef struct {
    double fs, freq, inc, osc_cos, osc_sin, last_re, last_im;
} po_mixer;
void po_mixer_init(po_mixer *m, double fs);
void po_mixer_set_frequency(po_mixer *m, double f);                 /* mixer.cpp:25-40 */
/* returns 1 and fills out[]; returns 0 (out untouched) when frequency==0 (mixer.cpp:51-53: returns in) */
int po_mixer_process(po_mixer *m, const double *in, double *out, uint32_t n);

/* ---- Decimator: pebblelib/decimator.cpp:64-226, 593-659, 695-753 ---- */
typedef struct po_decimator po_decimator;
po_decimator *po_decimator_new(void);
void po_decimator_free(po_decimator *d);
/* buildDecimationChain(fs_in, protect_bw, fs_out) -> achieved rate (decimator.cpp:64-149) */
double po_decimator_build(po_decimator *d, uint32_t fs_in, uint32_t protect_bw, uint32_t fs_out_min);
int po_decimator_chain_len(const po_decimator *d);
uint32_t po_decimator_dec_by2_stages(const po_decimator *d);
/* stage i: ntaps (0 => CIC3), stride (m_decimate after merging), design index into pebble_hb_designs */
void po_decimator_stage(const po_decimator *d, int i, int *ntaps, uint32_t *stride, int *design);
/* Decimator::process (vDSP path): returns number of output samples (decimator.cpp:152-226) */
uint32_t po_decimator_process(po_decimator *d, const double *in, double *out, uint32_t n);

/* ---- CDownConvert: pebblelib/downconvert.cpp:63-535, filtercoef.h (the alternate mixer + decimator; "parity unpinned": the
 * reference's tests hold no vector for it -- pinned only by the thresholds its own comments work out, downconvert.cpp:124-134) ---- */
typedef struct po_downconvert po_downconvert;
po_downconvert *po_downconvert_new(void);
void po_downconvert_free(po_downconvert *d);
void po_downconvert_set_frequency(po_downconvert *d, double f);      /* SetFrequency, :100-112 */
void po_downconvert_set_cw_offset(po_downconvert *d, double offset); /* SetCwOffset, downconvert.h */
/* SetDataRate (:139-206) / SetDataRateSimple (:213-237): returns the output rate */
double po_downconvert_set_data_rate(po_downconvert *d, double in_rate, double max_bw, int simple);
int po_downconvert_chain_len(const po_downconvert *d);
int po_downconvert_stage_taps(const po_downconvert *d, int i);      /* 0 => CIC3 */
/* ProcessData (:250-335): mixes a COPY of in[] (the reference mixes in place), returns the output count */
int po_downconvert_process(po_downconvert *d, int n, const double *in, double *out);

/* ---- plain DFT used by FastFIR and spectrum: Accelerate semantics, fftaccelerate.cpp:42-105 ----
 * dir=+1 forward e^{-j}, dir=-1 inverse e^{+j}; both unscaled; n power of two; in place. */
void po_fft(double *x, uint32_t n, int dir);

/* ---- CFastFIR: pebblelib/fastfir.cpp:77-334 (fft_size/fir_size are #defines 2048/1025 there) ---- */
typedef struct po_fastfir po_fastfir;
po_fastfir *po_fastfir_new(uint32_t fft_size, uint32_t fir_size);
void po_fastfir_free(po_fastfir *f);
/* SetupParameters (fastfir.cpp:191-272): 0 ok / unchanged, -1 "Filter Parameter error" (old taps kept) */
int po_fastfir_setup(po_fastfir *f, double lo, double hi, double offset, double fs);
/* ProcessData (fastfir.cpp:281-319): returns samples written to out */
int po_fastfir_process(po_fastfir *f, int n, const double *in, double *out);
const double *po_fastfir_coef(const po_fastfir *f);  /* frequency-domain H, fft_size complex */

/* ---- FFT::fftSpectrum: fft.cpp:67-118,129-157,183-225,324-399; windowfunction.cpp:214-235 ---- */
typedef struct po_spectrum po_spectrum;
/* window_type: 0 = BLACKMANHARRIS (SignalSpectrum, signalspectrum.cpp:58), 1 = NONE.
 * lift_clamp!=0 lifts the reference's m_maxFFTSize=65535 clamp (fft.h:21) -- documented deviation. */
po_spectrum *po_spectrum_new(uint32_t fft_size, uint32_t samples_per_buffer, int window_type, int lift_clamp);
void po_spectrum_free(po_spectrum *s);
uint32_t po_spectrum_bins(const po_spectrum *s);
double po_spectrum_coherent_gain(const po_spectrum *s);
const double *po_spectrum_window(const po_spectrum *s);
/* returns overload flag; out has bins doubles (dB amplitude, -f..+f) */
int po_spectrum_process(po_spectrum *s, const double *in, uint32_t n, double *out_db);

/* ---- CFir: pebblelib/fir.cpp:106-132, 246-337, 494-512 ---- */
typedef struct {
    int ntaps, state;
    double fs;
    double coef[2 * 75];              /* m_Coef (prototype low-pass) */
    double icoef[2 * 75], qcoef[2 * 75]; /* m_ICoef / m_QCoef: what the complex ProcessFilter uses */
    double zre[75], zim[75];
} po_fir;
void po_fir_init_const(po_fir *f, double fs); /* the 61-tap Hilbert prototype, demod_wfm.cpp:79-98,167 */
int po_fir_init_lp(po_fir *f, int ntaps, double scale, double astop, double fpass, double fstop, double fs);
void po_fir_generate_hb(po_fir *f, double freq_offset);                  /* CFir::GenerateHBFilter, fir.cpp:454-468 */
void po_fir_process_cpx(po_fir *f, int n, const double *in, double *out); /* in may == out */

/* ---- CIir: pebblelib/iir.cpp:88-207 ---- */
typedef struct { double a1, a2, b0, b1, b2, w1a, w2a, w1b, w2b; } po_iir;
void po_iir_init_lp(po_iir *q, double f0, double Q, double fs);
void po_iir_init_hp(po_iir *q, double f0, double Q, double fs);
void po_iir_init_bp(po_iir *q, double f0, double Q, double fs);
void po_iir_init_br(po_iir *q, double f0, double Q, double fs);
void po_iir_process_cpx(po_iir *q, int n, const double *in, double *out);

/* ---- Demod_AM::processBlockFiltered: application/demod/demod_am.cpp:17-64 ---- */
typedef struct { double fs, dc, dc_last; po_fir lp; } po_demod_am;
void po_demod_am_init(po_demod_am *d, double fs);
void po_demod_am_set_bandwidth(po_demod_am *d, double bw);
void po_demod_am_process(po_demod_am *d, const double *in, double *out, int n);

/* ---- Demod_NFM::processBlockNCO: application/demod/demod_nfm.cpp:44-66,225-257.  State members are `float` in the
 * reference (demod_nfm.h:27-40) and sin/cos of a float argument resolve to the float overloads in C++ ---- */
typedef struct {
    double fs;
    float err_dc, nco_freq, nco_lo, nco_hi, phase, alpha, beta, dc_alpha, out_gain;
    po_fir lp;
} po_demod_nfm;
void po_demod_nfm_init(po_demod_nfm *d, double fs);
void po_demod_nfm_process(po_demod_nfm *d, const double *in, double *out, int n);

/* ---- Demod_SAM::processBlock / pll: application/demod/demod_sam.cpp:5-112 (float PLL state, demod_sam.h:19-25) ---- */
typedef struct {
    double fs;
    float lo, hi, freq, phase, alpha, beta;
    double dc_re, dc_re_last, dc_im, dc_im_last;
    po_fir bp;
} po_demod_sam;
void po_demod_sam_init(po_demod_sam *d, double fs);
void po_demod_sam_process(po_demod_sam *d, const double *in, double *out, int n);

/* ---- Demod_WFM::processDataMono / processDataStereo (audio and the RDS branch up to the group queue; the text decoder
 * behind it, rdsdecode.cpp, is GUI): application/demod/demod_wfm.cpp:154-232, 255-365, 371-429, 451-485, 488-821 ---- */
typedef struct {
    double fs, d1_re, d1_im, deemph_alpha, deemph_re, deemph_im;
    po_iir mono_lp, notch;
    po_fir lp;
    /* stereo: Hilbert pair, pilot band-pass and pilot PLL (:161-171, :371-429) */
    po_fir hilbert;
    po_iir pilot_bp;
    double nco_phase, nco_freq, nco_lo, nco_hi, pll_alpha, pll_beta, err_ave, err_alpha, phase_adjust;
    int pilot_locked;
    struct po_rds *rds;   /* the RDS branch of processDataStereo (:296-357), allocated by init */
} po_demod_wfm;
typedef struct { uint16_t a, b, c, d; } po_rds_group; /* tRDS_GROUPS, rbdsconstants.h */
void po_demod_wfm_init(po_demod_wfm *d, double fs);
void po_demod_wfm_free(po_demod_wfm *d);              /* releases what init allocated (the RDS state) */
/* RDS branch (demod_wfm.cpp:296-357, 488-761): rate behind m_RdsDownConvert, the matched filter's output (m_RdsData) and the bit-sync
 * resonator's output of the LAST processDataStereo call, the bits handed to processNewRdsBit and the groups put into m_RdsGroupQueue since
 * the last drain (a cleared queue shows as the all-zero group the reference stuffs in), and getNextRdsGroupData itself */
double po_demod_wfm_rds_rate(const po_demod_wfm *d);
int po_demod_wfm_rds_last(const po_demod_wfm *d, double *data, double *sync, int cap);
int po_demod_wfm_rds_bits(po_demod_wfm *d, uint8_t *bits, int cap);
int po_demod_wfm_rds_pushed(po_demod_wfm *d, po_rds_group *g, int cap);
int po_demod_wfm_next_rds_group(po_demod_wfm *d, po_rds_group *g, int *changed); /* returns 0 when the queue is empty */
/* out = (left, right); returns the pilot-lock flag of this block (m_PilotLocked) */
int po_demod_wfm_process_stereo(po_demod_wfm *d, const double *in, double *out, int n);
/* in is const here; the reference overwrites its input (demod_wfm.cpp:212) */
void po_demod_wfm_process_mono(po_demod_wfm *d, const double *in, double *out, int n);

/* ---- Receiver::processIQData, DSP skeleton only: application/receiver.cpp:116-281, 758-1009 ----
 * Steps that are default-off / identity / GUI are omitted exactly as SURVEY.md 8(a-1) scopes them:
 * DCRemoval, IQBalance, NoiseBlanker, NoiseFilter, AGC, squelch and resampler are default-off / identity until their
 * setters below are used; audio out is the caller's. */
/* ------------------------------------------------------------------------------------------------
 * AGC -- application/agc.{h,cpp}.  modes: 0 AGC_OFF, 1 ACG_FAST, 2 AGC_MED, 3 AGC_SLOW, 4 AGC_LONG
 * (agc.h enum AgcMode).  Parity unpinned: the reference holds no recorded values for this class.
 * ---------------------------------------------------------------------------------------------- */
typedef struct po_agc po_agc;
po_agc *po_agc_new(double sample_rate);                       /* AGC::AGC, agc.cpp:15-32 */
void po_agc_free(po_agc *a);
void po_agc_set_mode(po_agc *a, int mode, int threshold);     /* setAgcMode + setParameters, agc.cpp:53-82,237-300 */
void po_agc_process(po_agc *a, const double *in, double *out, int n); /* processBlock, agc.cpp:84-235 */

/* ------------------------------------------------------------------------------------------------
 * CFractResampler (complex version) -- pebblelib/fractresampler.cpp:87-195.  Parity unpinned (no
 * recorded values in the reference).
 * ---------------------------------------------------------------------------------------------- */
typedef struct po_resampler po_resampler;
po_resampler *po_resampler_new(int max_input);                /* Init, fractresampler.cpp:87-140 */
void po_resampler_free(po_resampler *r);
/* Resample(InLength, Rate, pIn, pOut): returns the number of output samples */
int po_resampler_process(po_resampler *r, int n, double rate, const double *in, double *out);
double po_resampler_time(const po_resampler *r);              /* m_FloatTime, for tests */

/* ------------------------------------------------------------------------------------------------
 * Pre-chain conditioners and the noise filter (all default-off in the reference; parity unpinned: no
 * recorded values).  DCRemoval is po_iir_init_hp(10, 0.7071, fs) + po_iir_process_cpx (dcremoval.cpp:3-19).
 * ---------------------------------------------------------------------------------------------- */
/* IQBalance::ProcessBlock, application/iqbalance.cpp:65-86 (t1, t2 restart at zero every block) */
void po_iq_balance(double gain_factor, double phase_factor, const double *in, double *out, int n);
/* NoiseBlanker::ProcessBlock / ProcessBlock2, application/noiseblanker.cpp:45-97 */
typedef struct {
    float nb_avg_mag, nb2_avg_mag; int spike_count; double nb2_avg[2];
    double delay[16]; int head, last;   /* DelayLine(8, 2), pebblelib/delayline.cpp */
} po_nb;
void po_nb_init(po_nb *b);
void po_nb_enable(po_nb *b, int which);  /* setNbEnabled(true) / setNb2Enabled(true) resets */
void po_nb1_process(po_nb *b, const double *in, double *out, int n);
void po_nb2_process(po_nb *b, const double *in, double *out, int n);
/* NoiseFilter::ProcessBlock (ANF, LMS), application/noisefilter.cpp:31-88 */
typedef struct { double coeff[2 * 45]; double delay[2 * 512]; int head, last; } po_anf;
void po_anf_init(po_anf *a);
void po_anf_process(po_anf *a, const double *in, double *out, int n);

/* SignalStrength::fdEstimate (application/signalstrength.cpp:287-380) on one dB spectrum, update-timer gate forced
 * open.  out[4] = peakDb, avgDb, snrDb, floorDb.  Returns avgDb.  Integer bin width (quint32 / int) as written. */
double po_fd_estimate(const double *spectrum, int bins, uint32_t spectrum_rate, float bp_lo, float bp_hi, double mixer_freq, double *out);

typedef struct po_receiver po_receiver;
po_receiver *po_receiver_new(uint32_t fs, uint32_t frames_per_buffer, uint32_t spectrum_bins,
                             uint32_t fastfir_fft, uint32_t fastfir_taps);
void po_receiver_free(po_receiver *r);
void po_receiver_set_mode(po_receiver *r, int mode);
void po_receiver_set_mixer(po_receiver *r, double f);                 /* receiver.cpp:709 */
int po_receiver_set_filter(po_receiver *r, double lo, double hi);     /* receiver.cpp:658 */
double po_receiver_demod_rate(const po_receiver *r, int wfm);
uint32_t po_receiver_dec_stages(const po_receiver *r, int wfm);
/* dmFMS: the groups Demod::fmStereo popped (one getNextRdsGroupData per frame, demod.cpp:207-219) since the last drain, and whether each
 * differed from the one before it (only those reach the text decoder) */
int po_receiver_rds_polled(po_receiver *r, po_rds_group *g, uint8_t *changed, int cap);
/* one frame in; returns number of audio samples written (0 while accumulating, else frames_per_buffer);
 * spectrum_db (may be NULL) receives the unprocessed spectrum of this frame (every frame, no timer gate).
 * audio must hold max(frames_per_buffer, fastfir_fft) complex samples (the 8192/4097 FastFIR variant
 * emits 0 or 4096 samples per call). */
uint32_t po_receiver_process(po_receiver *r, const double *in, uint32_t n, double *audio, double *spectrum_db);
/* AGC::setAgcMode on the narrow branch (receiver.cpp:983) and the audio resampler (receiver.cpp:994-1003):
 * audio_rate 0 leaves the audio at the demod rate (the resampRate == 1 branch) */
void po_receiver_set_agc(po_receiver *r, int mode, int threshold);
void po_receiver_set_audio_rate(po_receiver *r, uint32_t audio_rate);
/* flags: 1 DCRemoval, 2 IQBalance (with gain/phase factors), 4 NoiseBlanker 1, 8 NoiseBlanker 2 (receiver.cpp:814-823,
 * in that order, before the spectrum and the mixer); anf: NoiseFilter on the narrow branch (receiver.cpp:974) */
void po_receiver_set_conditioners(po_receiver *r, int flags, double gain_factor, double phase_factor);
void po_receiver_set_anf(po_receiver *r, int on);
void po_receiver_set_squelch(po_receiver *r, double squelch_db);       /* receiver.cpp:704-707; gate at :893-897, :962-965 */

#ifdef __cplusplus
}
#endif
#endif
