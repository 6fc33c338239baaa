// receiver.h -- host-side objects behind the C ABI (include/pebblegpu.h).
//
// Five device "cores", one per reference class on the hot path, each owning its device state and
// launching its kernels on a caller-supplied stream:
//   DecimCore   <- Mixer + Decimator           (pebblelib/mixer.cpp, decimator.cpp)
//   FastFirCore <- CFastFIR / BandPassFilter   (pebblelib/fastfir.cpp, application/bandpassfilter.cpp)
//   AmCore      <- Demod_AM                    (application/demod/demod_am.cpp)
//   WfmCore     <- Demod_WFM (mono)            (application/demod/demod_wfm.cpp)
//   SpectrumCore<- FFT::fftSpectrum            (pebblelib/fft.cpp)
// Receiver composes them the way Receiver::processIQData does; the stand-alone steps (steps.hip) wrap one
// core each behind the reference's per-class call shapes.
#pragma once
#include <algorithm>
#include <complex>
#include <mutex>
#include <vector>
#include "../../include/pebblegpu.h"
#include "common.h"
#include "design.h"
#include "params.h"

namespace pg {

// [channel][hist | data] complex buffer; data(c) points at the first new sample, data(c)[-hist..-1] is history.
struct HistBuf {
    float2 *base = nullptr;
    long long pitch = 0;  // samples per channel row (hist + capacity, even)
    int hist = 0;
    long long cap = 0;
    int chans = 0;
    float2 *data(int c = 0) const { return base + (long long)c * pitch + hist; }
    int alloc(int channels, int hist_len, long long capacity);
    void release();
};

void fill_scan_section(ScanSection &s, int type, const double *c);
int scan_warm_subchunks(const ScanSection *secs, int nsec, double tol);
int make_twiddles(int n, float2 **d_tw);
int make_twiddles_t128(float2 **d_tw);
int make_twiddles_t128q(float2 **d_tw);  // four tables, the pruned transform's per-work-item factor folded in (k_spectrum_t128)
int run_save_tails(hipStream_t s, const std::vector<TailJob> &jobs, uint32_t channels, const OscAdvance *oa = nullptr);
int run_nap(hipStream_t s, unsigned ticks_100mhz);  // one sleeping wave (k_nap)
int run_fir_dec(hipStream_t s, const float2 *in, long long in_pitch, float2 *out, long long out_pitch, long long n_out, int stride,
                const float *d_taps, int ntaps, uint32_t channels);
int fill_tail_jobs(TailJobs &tj, const std::vector<TailJob> &jobs, const OscAdvance *oa);  // 0, or a failure code (too many / too deep)
int run_normalize_iq(int fmt, int order, double gain, const void *d_src, long long n, float2 *d_dst, hipStream_t s, bool wait, const float *final_scale = nullptr);
int run_gate_eval(hipStream_t s, const float4 *d_smeter, long long smeter_pitch, int frames_per_sf, int k, const float *d_squelch, unsigned char *d_gate,
                  int stride, uint32_t channels);
int run_gate_zero(hipStream_t s, float2 *audio, long long pitch, long long spf, const unsigned char *d_gate, int stride, uint32_t channels, int k);
int run_signal_strength(hipStream_t s, const float *d_spec, long long stream_pitch, int bins, long long n_frames, const SmBins *d_bins,
                        float4 *d_out, long long out_pitch, uint32_t channels);

// ---- oscillator bank (Mixer state for C channels) ----
struct OscBank {
    struct Ctl { double freq = 0, inc = 0, phase0 = 0; uint64_t n0 = 0; bool dirty = true; };
    double fs = 0;
    uint32_t C = 0;
    std::vector<Ctl> ctl;
    std::vector<ChanOsc> h_osc;               // host master copy
    struct Dyn { double phase0; uint32_t n0, mix_on; };
    OscDynInline inline_dyn = {};             // filled by upload() when allow_inline and C <= kOscInline
    bool allow_inline = false;                // set by owners whose kernels take OscDynInline
    Dyn *h_dyn[2] = {nullptr, nullptr};       // pinned staging of the per-call fields, ping-pong: no host sync per call
    hipEvent_t h_done[2] = {nullptr, nullptr};
    int h_idx = 0;
    ChanOsc *d_osc = nullptr;
    float *d_amp = nullptr;
    float a_inf = 0;
    // Banks too large for kernel-argument transport (C > kOscInline): with device_advance set by the owner, the per-call fields
    // are advanced ON the device at the end of each call (k_save_tails, OscAdvance) instead of being copied up before the next;
    // the owner passes advance_job(n) to its tail-refresh launch.  d_adv holds frac(n * inc) per channel for the current call
    // length, rebuilt (long double, as advance() computes) when the length or a frequency changes.
    bool device_advance = false;
    bool dev_dyn_valid = false;    // the device blocks hold the per-call fields of the coming call
    double *d_adv = nullptr;
    uint64_t adv_n = 0;
    bool adv_stale = true;
    uint64_t dyn_epoch = 0;        // counts host writes of the device blocks' per-call fields (a retune's block, the per-call upload)
    int advance_job(hipStream_t s, uint64_t n, OscAdvance *oa);  // fills *oa (osc == nullptr when the device is not advancing)
    int init(uint32_t channels, double sample_rate);
    void release();
    void retune(uint32_t ch, double f);           // Mixer::setFrequency, mixer.cpp:25-40
    void retune_keep(uint32_t ch, double f);      // CDownConvert::SetFrequency, downconvert.cpp:100-112: phase and amplitude carry on
    bool force_mix = false;                       // no "frequency 0 returns the input" exit (CDownConvert always multiplies)
    int upload(hipStream_t s);                     // refresh the device blocks (async, from pinned staging)
    void advance(uint64_t n);                      // after a call consumed n samples
    bool any_transient() const;                    // some oscillator is inside its amplitude transient (n0 < kAmpTab)
};

// ---- Mixer + Decimator ----
bool bank_variant(int np, int t1, int t2, int t3, int *hy, int *nstate, int *minw);  // a k_mix_dec_mfma instance exists for this front and triple

struct DecimCore {
    design::Chain chain;
    uint32_t C = 0;
    FirTaps first;                   // stage 0 (fused with the mixer)
    CascadeParams casc;              // stages 1.. (one fused kernel)
    size_t casc_lds_bytes = 0;
    HistBuf buf0, fin;               // stage-0 output (head-room = look-back of its reader), final output (absent for 1-stage chains)
    // A second stage with a stride >= 8 (merged halfbands at high input rates) would make the fused cascade's tiles mostly
    // halo: it runs as its own strided FIR (k_fir_dec) from buf0 into buf1, and the fused rest reads buf1.
    bool wide = false;
    bool fused_front = false;        // merged CIC3 + wide halfband in one kernel (k_mix_cic_hb): nothing is written at the CIC rate
    FrontTaps wide_fir;              // the wide stage's taps as kernel arguments (fused_front)
    bool bank_front = false;         // hb11 first stage in registers (k_mix_hb11_bank): a >= 16-channel bank off a shared stream, or one channel when asked
    bool want_lds_free = false;      // set per call by the owner: the first kernel should leave LDS alone (it runs beside the display transform)
    bool front_is_lds_free() const { return C == 1 && (fused_front || bank_front); }
    FrontTaps bank_taps;
    // the whole decimator in one kernel (k_mix_dec_fused): a >= 16-channel bank off one shared stream whose chain is hb11 x S
    // followed by hb15, hb19, hb31; calls inside an oscillator transient take the two-kernel route (both keep each other's history)
    bool fused_all = false, last_fused = false;
    struct FusedDecParams *fused_p = nullptr;   // host copy of the kernel's parameter block
    float2 *d_xhist[2] = {nullptr, nullptr};    // [16] raw input tail of the previous call, ping-pong with hist_parity
    float2 *d_y0stage = nullptr;                // [C][HY] the call's last first-stage outputs, copied into buf0's head-room by the tail refresh (k_mix_dec_fused: = d_y0stage2[0])
    float2 *d_y0stage2[2] = {nullptr, nullptr}; // k_mix_dec_mfma stages them alternately and reads the previous launch's directly (y0_cur: the one written last)
    int y0_cur = 0;
    int fused_hy = 0, fused_L = 0;
    // the same chain with its first stage on the matrix pipe, one wave per (32 channels, two chunks): k_mix_dec_mfma (kernels_bank_dec.h),
    // the default route of such a bank; PEBBLEGPU_BANK_DEC=0 keeps the four-wave pipeline above
    bool bank_mfma = false;
    int bank_nstate = 0, bank_minw = 2;          // running sums per channel of the chain's instance; waves per SIMD it admits
    int run_bank_mfma(hipStream_t s, const float2 *d_in, long long n, const OscBank &osc, bool had_state, const OscAdvance *oa);
    int bank_waves = 0;                          // target waves per SIMD of a launch (PEBBLEGPU_BANK_WAVES, default 1)
    int xh_depth = 16;                           // samples of raw-input tail kept in d_xhist
    float2 *d_bank_state[2] = {nullptr, nullptr};  // [C][38] the halfbands' running sums where the last k_mix_dec_mfma call ended (ping-pong)
    int bank_state_parity = 0;
    bool bank_state_valid = false;               // the previous call took that route (else the next one warms up from the first-stage history)
    int wide_taps = 0, wide_stride = 1;
    float *d_wide_taps = nullptr;
    HistBuf buf1;
    long long len1 = 0;
    long long len0 = 0, len_out = 0; // lengths produced by the last run
    const char *front_name = "";     // the kernel the last run() used for the mixer + first stage (bench / profiling labels)
    const char *rest_name = "";      // ... and for the remaining stages
    float2 *d_hist_mixed[2] = {nullptr, nullptr};  // [C][kMaxTaps]: mixed-sample history of stage 0 (read one, write the other)
    // One channel through hb11 x 8, hb15, hb23, hb47 beside an 8192-bin display transform: k_spectrum_t128<.., DEC> computes the whole
    // decimator from the frames it holds in LDS (DecFuse, kernels_spectrum.h).  Its look-back is the previous call's last frame, kept
    // WINDOWED (the transform's workgroups park windowed frames): written by that kernel, or by k_window_tail behind a call that took
    // the general kernels, so either route can follow the other.
    float2 *d_xtail_w[2] = {nullptr, nullptr};     // [2048] ping-pong
    int xtail_parity = 0;
    const float *fuse_window = nullptr;            // the display transform's window (set by the owner; nullptr: never fused)
    float2 *d_c0tab = nullptr;                     // [7][256] first-stage taps over that window times the oscillator's step per tap (DecFuse::c0tab)
    std::vector<float> h_r0;                       // ... the real part of it: h0[d] / w[n]
    std::vector<float2> h_c0;
    bool c0_valid = false;
    double c0_inc = 0;
    int c0_mix = 0;
    float2 *d_ph_scratch = nullptr;                // DecFuse::ph_scratch
    size_t ph_cap = 0;
    int set_fuse_window(const float *d_window, const std::vector<float> &w);  // the owner's display transform uses this window
    bool shape_for_spectrum() const;               // the chain is the one the kernel is built for
    bool spectrum_can_run(const OscBank &osc) const { return shape_for_spectrum() && fuse_window && want_lds_free && !osc.any_transient(); }
    // fills the kernel's parameter block for a call of n samples (before the transform is launched)
    int fill_dec_fuse(hipStream_t s, DecFuse *df, const OscBank &osc, long long n);
    // what is left for the chain's stream in such a call: the mixed-sample history for a later general call (two workgroups)
    int run_beside_spectrum(hipStream_t s, const float2 *d_in, long long in_pitch, bool shared_input, long long n, const OscBank &osc, const RawSrc *raw);
    int hist_parity = 0;
    // last_hist: head-room of the final buffer (what the consumer looks back at); last_gain: folded into the final stage
    int init(uint32_t channels, const design::Chain &c, long long max_in, int last_hist, float last_gain);
    void release();
    // n must be a multiple of chain.total; any such n streams exactly (no minimum frame length)
    // oa (from OscBank::advance_job for this call, or nullptr): when the route's kernel can advance the oscillators itself it does, and
    // osc_advanced says so (the caller then leaves the advance out of its tail launch)
    int run(hipStream_t s, const float2 *d_in, long long in_pitch, bool shared_input, long long n, const OscBank &osc,
            hipEvent_t after_first = nullptr, const RawSrc *raw = nullptr, const OscAdvance *oa = nullptr);
    bool osc_advanced = false;
    bool last_mfma = false;                 // the last run() was a k_mix_dec_mfma launch
    int flush_y0(hipStream_t s);
    OscDyn *d_dyn[2] = {nullptr, nullptr};  // [C] the oscillators' per-call fields as k_mix_dec_mfma hands them from call to call (ping-pong)
    int dyn_parity = 0;
    bool dyn_valid = false;                 // d_dyn[dyn_parity] holds this call's fields (the previous call was such a launch and nothing was uploaded since)
    uint64_t dyn_epoch_seen = 0;
    bool y0_pending = false;                // d_y0stage holds first-stage history the stage-0 head-room has not received yet (copied when a call needs it there)
    // the first kernels this call would run read raw device-format samples themselves (k_mix_hb11_lean + its edge launch)
    bool raw_ready(const OscBank &osc) const { return bank_front && C == 1 && want_lds_free && !osc.any_transient() && !(fused_all || bank_mfma); }
    void tail_jobs(std::vector<TailJob> &jobs) const;  // after run(): what must be refreshed before the next call
    // Two output buffers, written by alternate calls, so that whatever reads a call's output (the band-pass) may run on another stream
    // beside the NEXT call's decimator: the consumer's look-back (the head-room) is carried from the buffer just written into the other
    // one's head-room.  tail_jobs() = tail_jobs_dec() (the decimator's own histories: its stream) + tail_job_out() (the consumer's stream)
    hipEvent_t done_event = nullptr;         // set by the caller before run(): an event to complete WITH the call's last launch when that is the bank kernel ...
    bool done_recorded = false;              // ... and whether run() did so (else the caller records it)
    bool rotate3 = true;                     // this call rotates three output buffers (set by the caller before run(); ignored without fin3)
    // a call of this many input samples is "long": chunks of 128 outputs or more, where two output buffers do as well as three
    bool long_call(long long n) const { const long long lo = n / (long long)chain.total; return (lo + 2 * std::max(1LL, 1024LL / ((C + 31) / 32)) - 1) / (2 * std::max(1LL, 1024LL / ((C + 31) / 32))) >= 128; }
    HistBuf fin2, fin3;                      // fin: this call's; fin2: the next call's (its head-room filled by this call's consumer); fin3: a third, written by the call after that,
                                             // so that a call's decimator waits for the consumer of the call THREE back (long over) instead of two (often still running)
    int enable_double_out();                 // after init(); fails for a single-stage chain (its output is the first stage's buffer)
    bool double_out() const { return fin2.base != nullptr; }
    void tail_jobs_dec(std::vector<TailJob> &jobs) const;
    void tail_job_out(std::vector<TailJob> &jobs) const;
    const HistBuf &out() const { return casc.nst > 0 ? fin : buf0; }
    long long out_len() const { return len_out; }
};

// ---- CFastFIR ----
struct FastFirCore {
    uint32_t C = 0, fft_n = 2048, taps = 1025;
    float2 *d_H = nullptr, *d_tw = nullptr;
    float2 *d_tw128 = nullptr;   // 2048-point case: table of the two-wave transform (fft_t128.h)
    int init(uint32_t channels, uint32_t fft_size, uint32_t fir_size);
    void release();
    long long block_len() const { return (long long)fft_n - (taps - 1); }
    // *ok = false (and H left alone) on the reference's "Filter Parameter error"
    int design(hipStream_t s, uint32_t ch, double lo, double hi, double offset, double rate, bool *ok);
    // in: buffer with taps-1 head-room; n multiple of block_len()
    int run(hipStream_t s, const HistBuf &in, long long n, float2 *out, long long out_pitch);
    // caller-owned rows without head-room; d_tail [C][taps-1] carries the overlap between calls and is refreshed here
    // d_tail_next (2048-point plan only): a second [C][taps-1] buffer that receives the next call's overlap straight from the
    // kernel (the caller alternates the two) instead of a copy launched behind it
    int run_ext(hipStream_t s, const float2 *in, long long in_pitch, float2 *d_tail, long long n, float2 *out, long long out_pitch, float2 *d_tail_next = nullptr);
};

// ---- Demod_AM ----
struct AmCore {
    uint32_t C = 0;
    double rate = 0;
    HistBuf tmp;                    // DC-blocked magnitude, head-room for the audio FIR
    float *d_taps = nullptr;        // [C][kMaxTaps]
    int *d_ntaps = nullptr, *d_list = nullptr;
    double *d_state = nullptr;      // [C][1][2][2]
    ScanParams<1> scan;
    std::vector<int> list;
    int init(uint32_t channels, double demod_rate, long long max_n);
    void release();
    int set_bandwidth(hipStream_t s, uint32_t ch, double bw);   // Demod_AM::setBandwidth, demod_am.cpp:17-21
    int set_list(hipStream_t s, const std::vector<int> &am_channels);
    // in/out rows may be the same buffer (the scan reads `in`, the FIR writes `out`)
    int run(hipStream_t s, const float2 *in, long long in_pitch, float2 *out, long long out_pitch, long long n, Gate gate = Gate{nullptr, 0, 0});
    // defer_tail: run() leaves the refresh of tmp's head-room to the caller's tail launch (tail_jobs: every row, also those of channels
    // that are not AM -- harmless; not for gated calls, whose closed channels keep their history)
    bool defer_tail = false;
    long long last_n = 0;
    void tail_jobs(std::vector<TailJob> &jobs) const { if (defer_tail && !list.empty() && last_n > 0) jobs.push_back(TailJob{tmp.data(), tmp.pitch, last_n, tmp.hist, 0, nullptr, 0}); }
};

// ---- Demod_NFM / Demod_SAM (PLL demodulators) ----
struct PllCore {
    uint32_t C = 0;
    double rate = 0;
    int mode = 0, ntaps = 0;        // 0 NFM, 1 SAM
    PllParams pp;
    HistBuf tmp;                    // PLL output, head-room for the CFir
    float *d_taps_i = nullptr, *d_taps_q = nullptr;
    PllState *d_state = nullptr;
    int *d_list = nullptr;
    std::vector<int> list;
    int init(uint32_t channels, double demod_rate, long long max_n, int which);
    void release();
    int set_list(hipStream_t s, const std::vector<int> &channels);
    int run(hipStream_t s, const float2 *in, long long in_pitch, float2 *out, long long out_pitch, long long n, Gate gate = Gate{nullptr, 0, 0});
};

// ---- Demod_WFM mono ----
// ---- the RDS branch of Demod_WFM::processDataStereo (demod_wfm.cpp:296-357, 488-786) for the dmFMS channels of a WfmCore ----
// Device side (kernels_rds.h): discriminator -> Hilbert pair -> m_RdsDownConvert -> 2400 Hz low-pass -> PLL -> matched filter -> bit-rate
// resonator, slicer, block synchroniser; every completed group goes into a per-channel log with the number of the processDataStereo
// call (frame) it fell in.  Host side: m_RdsGroupQueue (a ring of RDS_Q_SIZE with the reference's head / tail arithmetic) replayed from
// that log together with its consumer, Demod::fmStereo (demod.cpp:196-226: ONE getNextRdsGroupData per frame) -- groups() hands out
// what that consumer popped, each with getNextRdsGroupData's return value (the group differs from the one before it).
struct RdsGroup { uint16_t a, b, c, d; };
struct RdsCore {
    uint32_t C = 0;
    long long cap = 0;               // demodulator-rate samples per call at most
    bool on = false;                 // allocated (first dmFMS channel of the owner)
    design::RdsDesign des;
    int D = 1;                       // the chain's decimation
    RdsParams pp{};
    HistBuf raw, mag;                // rows of double
    std::vector<HistBuf> st;         // rows of double2: st[j] the input of stage j, st[nst] the low-pass's input
    double2 *d_lp = nullptr;         // [C][cap / D]
    double *d_data = nullptr;        // [C][cap / D] m_RdsData of the last call
    RdsState *d_state = nullptr;
    RdsEvent *d_log = nullptr;       // [C][log_cap]
    double *d_amp = nullptr, *d_matched = nullptr, *d_lptaps = nullptr;
    std::vector<double *> d_taps;    // per stage, oldest sample first
    std::vector<int> ntaps, newest;
    long long last_len = 0;
    struct Host {                    // m_RdsGroupQueue and its consumer
        RdsGroup q[100];
        int head = 0, tail = 0;
        RdsGroup last{0, 0, 0, 0};
        unsigned long long seen = 0; // log entries replayed
        long long frames = 0;        // frames whose pop has been replayed
        std::vector<RdsGroup> out;
        std::vector<unsigned char> changed;
        unsigned long long lost = 0; // log entries overwritten before a collect() read them
    };
    std::vector<Host> host;
    int init(uint32_t channels, double demod_rate, long long max_n);
    void release();
    // queues the branch for the listed channels; block: demodulator-rate samples per processDataStereo call of the reference
    int check(long long n, int block) const;   // the sizes run() accepts (the owner asks before it queues anything of the call)
    int run(hipStream_t s, const float2 *in, long long in_pitch, long long n, const double *d_hilb, const int *d_list, int n_list, int block);
    int collect(hipStream_t s, uint32_t ch);   // waits for the stream, replays the channel's new log entries
    int groups(hipStream_t s, uint32_t ch, RdsGroup *g, unsigned char *changed, uint32_t cap_out, uint32_t *n_out);
    int signal(hipStream_t s, uint32_t ch, double *data, uint32_t cap_out, uint32_t *n_out);  // m_RdsData of the last call
};

struct WfmCore {
    uint32_t C = 0;
    double rate = 0;
    HistBuf a, b, c;                // low-passed IQ (hist 1), discriminator (hist FIR), FIR out
    float *d_taps = nullptr;
    int ntaps = 0;
    bool lp_on = false;
    ScanParams<1> lp;
    ScanParams<2> dn;
    int warm_lp = -1, warm_dn = -1, parity = 0;
    double *d_lp_state[2] = {nullptr, nullptr}, *d_dn_state[2] = {nullptr, nullptr};
    bool fused = false;             // single-kernel FIR-ised path (k_wfm_fir); else the multi-kernel sequential fallback
    int L4 = 0, Llp = 1;            // fused: combined audio response (padded to 16) and low-pass response lengths
    float *d_h = nullptr;           // [L4 + 16]
    float *d_hlp = nullptr;         // [Llp]
    float2 *d_xtail[2] = {nullptr, nullptr};  // [C][L4 + Llp] input history, ping-pong
    const float2 *deferred_in = nullptr;      // set by run(): the history copy is left to tail_jobs()
    long long deferred_pitch = 0;
    // dmFMS as the reference delivers it: processDataStereo's pilot PLL does not hold lock, so the block
    // copies the discriminator output to both channels -- processDataMono without its 75 kHz pre-filter (demod_wfm.cpp:255-293)
    std::vector<unsigned char> stereo;        // per channel: 1 = dmFMS
    unsigned char *d_stereo = nullptr;        // device copy, uploaded by run() when dirty; null until a channel asks for it
    bool stereo_dirty = false;
    int set_stereo(uint32_t ch, bool on);
    // ... and before it drops out (at most the first blocks of a stream): k_wfm_pilot runs the discriminator, the Hilbert pair, the
    // pilot band-pass and the PLL serially per dmFMS channel and leaves the (L - R) contribution of the blocks that end locked in `lm`;
    // k_wfm_lmr_fir sends it through the audio response and adds / subtracts it.  Allocated when a channel first asks for dmFMS.
    HistBuf lm;                               // [C] rows, .x = lmr, head-room = the audio response's look-back
    struct WfmPilotState *d_pilot = nullptr;  // [C]
    double *d_hilb = nullptr;                 // [2][61]
    int *d_stereo_list = nullptr;             // the dmFMS channels
    int n_stereo = 0;
    long long max_n_ = 0;
    int stereo_block = 2048;                  // samples per processDataStereo call in the reference (the owner's frame length)
    WfmPilotParams pilot;
    RdsCore rds;                              // the RDS branch (PEBBLEGPU_RDS=0 leaves it out)
    // Demod_WFM::getStereoLock (demod_wfm.cpp:436-447): m_PilotLocked after the last block, and whether it differs from the value the
    // previous call of this function saw (m_LastPilotLocked starts as the opposite of m_PilotLocked: the first call reports a change)
    std::vector<char> stereo_ran, last_lock;
    int stereo_lock(hipStream_t s, uint32_t ch, int *lock, int *changed);
    bool rds_enabled = true;
    int init(uint32_t channels, double demod_rate, long long max_n);
    void release();
    // more_tails / oa: the caller's other tail-refresh jobs and oscillator advance; when the single-kernel path runs it carries
    // them (and its own history copy) in extra workgroups of its launch and sets *carried -- the caller then skips its own
    // tail-refresh launch (one launch and ~5 us fewer on the call's critical path)
    int run(hipStream_t s, const float2 *in, long long in_pitch, float2 *out, long long out_pitch, long long n,
            const std::vector<TailJob> *more_tails = nullptr, const OscAdvance *oa = nullptr, bool *carried = nullptr);
    void tail_jobs(std::vector<TailJob> &jobs) const;
    long long last_n = 0;
};

// ---- AGC (application/agc.cpp), on the band-passed samples of the narrow branch ----
struct AgcCore {
    uint32_t C = 0;
    double rate = 0;
    struct AgcState *d_state = nullptr;
    int *d_list = nullptr;
    std::vector<int> list;          // channels the kernel visits: every mode but OFF-with-unit-gain
    struct Host { int mode = 0, threshold = 1, use_hang = 0, thr = 0, decay = 0; double slope = 0, sample_rate = 100.0, manual = 1.0; bool dirty = false; };
    std::vector<Host> host;
    int init(uint32_t channels, double demod_rate);
    void release();
    int set_mode(uint32_t ch, int mode, int threshold);      // AGC::setAgcMode, agc.cpp:53-82
    std::vector<char> muted;        // channels the owner keeps out of the list (dmNONE: the reference returns before the AGC)
    void set_muted(uint32_t ch, bool m) { if (muted.size() != C) muted.assign(C, 0); if ((muted[ch] != 0) != m) { muted[ch] = m; list_dirty = true; } }
    int apply(hipStream_t s);                                // upload changed parameters and the channel list
    int run(hipStream_t s, float2 *buf, long long pitch, long long n, Gate gate = Gate{nullptr, 0, 0});
    bool list_dirty = false;
};

// ---- DCRemoval, IQBalance, NoiseBlanker on the input streams (receiver.cpp:814-823) and NoiseFilter/ANF (receiver.cpp:974) ----
struct ConditionCore {
    uint32_t S = 0, nf = 0;
    double fs = 0;
    long long cap = 0;
    struct Host { int flags = 0; double gain = 1, phase = 0; };
    std::vector<Host> host;
    bool any = false, dirty = false;
    float2 *d_buf = nullptr;          // [S][cap] conditioned copy of the input (allocated on first enable)
    double *d_dc_state = nullptr;     // [S][1][2][2]
    int *d_dc_list = nullptr;
    std::vector<int> dc_list;
    double2 *d_iq = nullptr;          // [S] (gain, phase), gain < 0: off
    struct NbState *d_nb = nullptr;   // [S]
    ScanParams<1> dc;
    bool iq_any = false, nb_any = false;
    int init(uint32_t streams, uint32_t frame, double sample_rate, long long max_n);
    void release();
    int set(uint32_t stream, int flags, double gain, double phase);
    int apply(hipStream_t s);
    // copies d_in to the internal buffer and runs the enabled steps on it; *out = what the rest of the call should read
    int run(hipStream_t s, const float2 *d_in, long long in_pitch, long long n, const float2 **out, long long *out_pitch);
};
struct AnfCore {
    uint32_t C = 0;
    struct AnfState *d_state = nullptr;
    int *d_list = nullptr;
    std::vector<int> list;
    std::vector<char> on, muted;   // muted: dmNONE channels (the reference returns before the noise filter)
    bool dirty = false;
    void set_muted(uint32_t ch, bool m) { if (muted.size() != C) muted.assign(C, 0); if ((muted[ch] != 0) != m) { muted[ch] = m; dirty = true; } }
    int init(uint32_t channels);
    void release();
    int set(uint32_t ch, bool enable);
    int apply(hipStream_t s);
    int run(hipStream_t s, float2 *buf, long long pitch, long long n, Gate gate = Gate{nullptr, 0, 0});
};

// ---- CFractResampler (complex), pebblelib/fractresampler.cpp ----
struct ResampCore {
    uint32_t C = 0, nf = 0;
    double dt = 1.0, float_time = 0.0;     // Rate = input rate / output rate; m_FloatTime
    float *d_sinc = nullptr;
    float2 *d_hist[2] = {nullptr, nullptr};  // [C][28] last inputs of the previous call, ping-pong
    struct ResampFrame *d_frames = nullptr, *h_frames[2] = {nullptr, nullptr};
    hipEvent_t h_done[2] = {nullptr, nullptr};
    int parity = 0, pin = 0;
    uint32_t max_frames = 0;
    int init(uint32_t channels, uint32_t frame, double rate, uint32_t frames_per_call);
    void release();
    long long max_out(long long n) const { return (long long)((double)n / dt) + (long long)(n / (nf ? nf : 1)) + 8; }
    // n: multiple of the frame; returns the output count of this call in *n_out (no device sync)
    int run(hipStream_t s, const float2 *in, long long in_pitch, long long n, float2 *out, long long out_pitch, long long *n_out);
};

// ---- FFT::fftSpectrum ----
struct SpectrumCore {
    uint32_t S = 0, nf = 2048, bins = 0;
    float *d_prev[2] = {nullptr, nullptr};
    float *d_window = nullptr;
    float2 *d_btab = nullptr, *d_tw_nf = nullptr;  // btab: [bins/nf][32] wave-uniform pre-twiddle factors
    float2 *d_btab128 = nullptr, *d_tw128 = nullptr;  // the same for the two-wave transform (fft_t128.h), 8192 bins
    float2 *d_ftab = nullptr;         // [bins/nf][nf] window[n] * W_bins^{n q}: the one factor per point of k_spectrum_q128
    std::vector<float> h_window;      // host copy of the window (the decimator's taps against windowed samples)
    bool last_fullc = false;          // the last run used k_spectrum_t128's register-held twiddles (nothing ran beside it)
    bool use_w64 = false;             // 8192 bins on k_spectrum_w64 (PEBBLEGPU_SPECTRUM_W64=1 when the core is created)
    int stagger = 0, pad_lds = 0;     // k_spectrum_t128: barrier intervals between the two halves of a 1024-item workgroup (0: 512-item workgroups)
    bool per_q = false;               // k_spectrum_q128 (one transform per 128-item workgroup) instead of the shared-frame kernels
    // 65536-sample frames / 65536 bins (four-step, kernels_spectrum.h): the [S][F][32][2048] intermediate
    bool big = false;
    float2 *d_Y = nullptr;
    size_t y_cap = 0;
    float scale = 0;
    int parity = 0;
    // any other frame length, and frames shorter than samplesPerBuffer (un-windowed): k_spectrum_any
    bool any = false;                 // the frame length / bin count has none of the kernels above: every call takes the general kernel
    int any_M = 0, any_logM = 0, any_zp_log2 = 0;
    float2 *d_twM = nullptr;          // W_M^k, k < M / 2
    int run_any(hipStream_t s, const float2 *d_in, long long in_pitch, long long n_frames, float *d_out, int n_in, bool windowed);
    int init(uint32_t streams, uint32_t frame, uint32_t fft_size);
    void release();
    int run(hipStream_t s, const float2 *d_in, long long in_pitch, long long n_frames, float *d_out, const RawSrc *raw = nullptr, const DecFuse *df = nullptr, bool nothing_beside = false);
    bool dec_ready() const { return !big && !per_q && bins == 8192 && !use_w64; }  // k_spectrum_t128<.., DEC> exists for this plan
    bool raw_ready() const { return !big && !per_q && bins == 8192; }  // k_spectrum_t128 converts in its loads
};

// HIP events around each kernel group, kept for the last kRing calls so a caller can run calls back to back
// (no host sync per call) and read the per-kernel durations afterwards.
struct Timers {
    static constexpr int kRing = 64;
    hipEvent_t ev[kRing][8] = {};
    hipEvent_t start_ev[kRing] = {};  // where the call began (its ev[0])
    hipEvent_t end_ev[kRing] = {};    // where it ended: its ev[6], or the next call's ev[0] (a side-by-side call records no end of its own)
    int open_slot = -1;               // the last call's end is not known yet (Receiver::close_timing)
    bool detailed[kRing] = {};   // per-kernel events (2..5) were recorded for that call
    bool has_mid[kRing] = {};    // event 1 (behind the display transform) was recorded: calls without a spectrum skip it unless profiling
    uint64_t calls = 0;
    hipEvent_t *slot() { return ev[calls % kRing]; }
};

class Receiver {
public:
    int create(const pebblegpu_config *cfg);
    ~Receiver();
    int set_mixer(uint32_t ch, double f);
    int set_bandpass(uint32_t ch, double lo, double hi);
    int set_mode(uint32_t ch, int mode);
    int set_agc(uint32_t ch, int mode, int threshold);
    int set_conditioners(uint32_t stream, int flags, double iq_gain, double iq_phase);
    int set_noise_filter(uint32_t ch, bool on);
    int process_raw(int fmt, int order, double gain, const void *d_raw, uint64_t n);  // normalizeIQ on the library's stream, then process()
    // host ingest through library-owned pinned buffers (two slots): the device plugin writes its raw samples into a slot, the upload
    // of one slot travels on a copy stream while the call on the other computes
    int ingest_acquire(uint32_t slot, uint64_t bytes, void **host_ptr);
    int ingest_submit(uint32_t slot, uint64_t bytes);
    int process_ingested(uint32_t slot, int fmt, int order, double gain, uint64_t n);
    int set_squelch(uint32_t ch, double squelch_db);   // Receiver::squelchChanged, receiver.cpp:704-707
    // dmFMS channels of a WFM bank: what Demod::fmStereo took from the RDS group queue since the last call (waits for queued work)
    int rds_groups(uint32_t ch, RdsGroup *g, unsigned char *changed, uint32_t cap, uint32_t *n);
    int stereo_lock(uint32_t ch, int *lock, int *changed);
    int process(const float2 *d_iq, uint64_t n, bool with_spectrum, bool with_chain, const RawSrc *raw = nullptr);
    int process_iq(const double *iq, uint16_t n, double *audio, uint32_t *n_audio, double *spectrum_db);
    int sync();
    int close_timing();  // records the end event a side-by-side call left out (no-op otherwise)
    const char *kernel_name(int which) const;  // the kernels behind pebblegpu_receiver_last_ms's groups, as last run

    int device = 0;
    double fs = 0;
    uint32_t nf = 2048, C = 1, S = 1, bins = 0, ff_n = 2048, ff_taps = 1025, max_sf = 1;
    bool shared_input = false, wfm = false;
    design::Chain chain;
    uint32_t demod_rate_int = 0;
    uint64_t superframe = 0;
    uint64_t last_audio_n = 0, last_spec_frames = 0;
    Timers tm;
    HistBuf audio;   // [C][k*nf]
    float *d_spec = nullptr;
    // SignalSpectrum::zoomed (signalspectrum.cpp:89-113): the display transform of every decimated frame of every channel
    uint32_t zoom_bins = 0;
    float *d_zoom = nullptr;          // [C][max_sf * superframe / (D * nf)][zoom_bins]
    uint64_t last_zoom_frames = 0;
    // SignalStrength::fdEstimate per frame (S-meter): enabled on request, needs the spectrum
    bool smeter_on = false;
    float4 *d_smeter = nullptr;       // [C][max frames]
    SmBins *d_sm_bins = nullptr;
    long long smeter_pitch = 0;
    int enable_smeter(bool on);
    double squelch_db_ = -120.0;      // DB::minDb: the gate never closes (receiverwidget.cpp:82); the one-channel, one-super-frame shape
    // banks (or calls of several super-frames): a per-channel threshold, decided on the device per (channel, super-frame)
    std::vector<float> squelch_;      // per channel, -120 = never closes
    bool bank_gate_ = false, squelch_dirty_ = false;
    float *d_squelch = nullptr;
    unsigned char *d_gate = nullptr;  // [C][max_sf]
    float4 *h_gate_ = nullptr;        // pinned: the S-meter value the gate reads back
    uint64_t squelched_calls = 0;
    bool failed_ = false;             // a process call failed after it had started queueing work: the handle is refused from then on
    bool profile_detail = false;      // record the per-kernel events too (pebblegpu_receiver_set_profiling)
    uint32_t audio_rate = 0;          // 0: audio stays at the demod rate (the resampRate == 1 branch, receiver.cpp:1000-1003)
    float2 *d_audio_rs = nullptr;     // [C][rs_pitch] resampled audio
    long long rs_pitch = 0;
    const float2 *audio_ptr() const { return audio_rate ? d_audio_rs : audio.data(0); }
    long long audio_pitch() const { return audio_rate ? rs_pitch : audio.pitch; }

private:
    struct ChanCtl {
        int mode = 0;
        double lo = 0, hi = 0, am_bw = 16000;  // Demod_AM ctor default (demod_am.cpp:9)
        bool bp_valid = false, bp_dirty = false, am_dirty = true;
    };
    int apply_controls(hipStream_t osc_stream);
    std::mutex mu_;
    hipStream_t stream_ = nullptr;
    // The display transform is arithmetic-bound and the front of the chain memory-bound: when the chain's first kernel
    // needs no LDS (the register front ends) the two run side by side, the chain on its own stream between a fork and a
    // join event.
    hipStream_t chain_stream_ = nullptr;
    // No events of its own: the chain stream waits for the call's start event, the call's end event is recorded on the chain
    // stream once it has also seen the transform's end event, and whatever next touches the main stream (the next call, a
    // synchronise) first waits for that end event.  Every event record costs the stream ~5 us, so none is spent on the fork/join.
    hipEvent_t spec_end_ = nullptr;   // pipelined calls: the last display transform queued on the main stream (for the chain's stream to wait on at a join)
    bool pipeline_ = false;           // successive side-by-side calls overlap (PEBBLEGPU_PIPELINE=1 when the receiver is created)
    bool touched_ = true;             // a setter ran since the last call
    bool bank_pipe_ok_ = false;       // no display transform: the call's two stages (decimator | band-pass .. resampler) on the two streams, stage 2 beside the next call's stage 1
    hipEvent_t f_end_[3] = {nullptr, nullptr, nullptr};  // where stage 2 of the last three such calls ended
    std::vector<std::pair<const void *, hipEvent_t>> out_reader_;  // per decimator output buffer: where the second stage that last read it ended
    hipEvent_t d_end_prev_ = nullptr;           // where the last call ended, if that was a two-stage call (the next one is timed from there)
    hipEvent_t sync_ev_[4] = {nullptr, nullptr, nullptr, nullptr};  // stage 1 -> stage 2 hand-over events (no timing), a ring
    hipEvent_t pipe_ev_ = nullptr;
    bool fuse_dec_ = false;           // the one-channel decimator inside the display transform's kernel (PEBBLEGPU_FUSE_DEC=1 at creation)
    hipEvent_t chain_end_ = nullptr;  // set when a two-stream call failed half-way: what was queued on the chain stream, for the main stream to wait on
    std::vector<ChanCtl> ctl_;
    bool am_list_dirty_ = true, sm_dirty_ = true;
    long long pll_cap_ = 0;
    OscBank osc_;
    DecimCore dec_;
    FastFirCore ff_;
    AmCore am_;
    PllCore nfm_, sam_;
    WfmCore wfmc_;
    AgcCore agc_;
    ConditionCore cond_;
    AnfCore anf_;
    ResampCore resamp_;
    SpectrumCore spec_, zoom_;
    float2 *d_stage_in_ = nullptr;
    float2 *d_raw_stage_ = nullptr;   // process_raw: the normalised copy of a raw device-format call (allocated on first use)
    struct IngestSlot {
        void *h = nullptr, *d = nullptr;      // pinned host buffer and its device twin
        size_t cap = 0, submitted = 0;
        hipEvent_t uploaded = nullptr, done_main = nullptr, done_chain = nullptr;
        bool in_flight = false;               // a call that reads the device twin has been queued and not waited for
    } ingest_[2];
    hipStream_t copy_stream_ = nullptr;
    std::vector<float> h_frame_, h_out_;
    uint64_t acc_frames_ = 0;
};

}  // namespace pg
