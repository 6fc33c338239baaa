// params.h -- plain structs shared by the host objects and the kernels (passed by value or uploaded).
#pragma once
#include "common.h"

namespace pg {

// Per-channel oscillator block, rewritten by the host before every call (a few hundred bytes/channel).
// Mixer::processBlock (pebblelib/mixer.cpp:48-81) restated in closed form:
//   osc_i = a_{n0+i} * exp(j*2*pi*(phase0 + (i+1)*inc)),  a_0 = 1, a_{k+1} = a_k*(1.95 - a_k^2)
struct ChanOsc {
    // --- first 16 bytes change every call (uploaded alone, asynchronously) ---
    double phase0;          // cycles; accumulated phase at the end of the previous call (0 after a retune)
    uint32_t n0;            // samples since the last retune, saturated at kAmpTab
    uint32_t mix_on;        // 0 => f == 0: Mixer returns its input untouched (mixer.cpp:51-53)
    // --- the rest changes only on a retune ---
    double inc;             // cycles per sample = -f/Fs (the reference negates f, mixer.cpp:31)
    float2 step[kMaxTaps];  // exp(j*2*pi*d*inc), d = 0..kMaxTaps-1, rounded from fp64
    float2 step512;         // exp(j*2*pi*512*inc): advance of one 256-lane float4 sweep
    float2 pad_;
};

struct FirTaps {
    int ntaps;
    int stride;
    int cic3;   // 1: CIC3 decimate-by-`stride` in the reference's merged form (decimator.cpp:719-737)
    float gain; // applied to the output (gain restore on the last stage, receiver.cpp:935-938), else 1
    float h[kMaxTaps];
};

// the wide hb11 stage behind a merged CIC3, as kernel arguments of k_mix_cic_hb
constexpr int kFrontT1 = 11;
struct FrontTaps { float h[kFrontT1]; int stride; };

// later decimation stages fused in one kernel (k_cascade): taps live in the kernel-argument segment
constexpr int kMaxCascade = 6;
struct CascadeParams {
    int nst, outb;            // fused stages, final outputs per workgroup
    int lds_half, pad_;       // float2 slots of the first ping-pong buffer
    float gain;               // applied to the final stage's output
    int ntaps[kMaxCascade], stride[kMaxCascade];
    float h[kMaxCascade][60];
};

constexpr size_t kChanOscDynBytes = 16;
// The per-call part of ChanOsc for small banks travels in the kernel arguments instead of a host-to-device copy in the
// stream (which costs a ~10 us bubble per call); use == 0: read it from the ChanOsc block.
constexpr int kOscInline = 8;
struct OscDyn { double phase0; uint32_t n0, mix_on; };
struct OscDynInline {
    OscDyn d[kOscInline];
    int use, pad_;
};

constexpr int kSeg = 8;               // consecutive samples one lane runs serially
constexpr int kSub = 64 * kSeg;       // samples one wave scans at a time

enum ScanType { kOnePoleDiff = 0, kOnePoleAvg = 1, kBiquadDf2 = 2 };

struct ScanSection {
    int type;
    int pad_;
    double c[5];      // kOnePoleDiff: c0 = alpha.  kOnePoleAvg: c0 = alpha.  kBiquadDf2: b0 b1 b2 a1 a2
    double P[6][4];   // M^(kSeg * 2^k), row-major 2x2, k = 0..5
};
template <int NSEC> struct ScanParams { ScanSection sec[NSEC]; };

// PLL demodulators (k_pll_demod): loop constants and per-channel state, floats as the reference declares them
struct PllParams {
    int mode;                 // 0: NFM, 1: SAM
    float lo, hi, alpha, beta;
    float dc_alpha, out_gain; // NFM only
    int pad_;
};
struct PllState {             // per channel
    float freq, phase, err_dc, pad_;
    double dc_re_last, dc_im_last;  // SAM DC removal (doubles, demod_sam.h:27-31)
};

// history-tail refresh jobs, one launch for all buffers: for every channel c, dst[c][j] = data[c][n - hist + j], j < hist,
// where dst is the buffer's own head-room (data[c][-hist + j]) unless a separate history buffer is given
struct TailJob {
    float2 *data;
    long long pitch, n;
    int hist, pad_;
    float2 *dst;           // nullptr: the head-room in front of data
    long long dst_pitch;
};
constexpr int kMaxTailJobs = 12;
// The oscillators' per-call fields can ride on the same launch: osc[c].phase0 += adv[c] (mod 1), n0 += adv_n (saturating at
// kAmpTab) for c < osc_count -- OscBank::advance restated on the device, so a bank too large for kernel-argument transport
// needs no host-to-device copy per call (a DMA-engine copy in the stream cost ~20 us of idle GPU each call).
struct OscAdvance {
    ChanOsc *osc;        // nullptr: nothing to advance
    const double *adv;   // [osc_count] frac(n * inc), computed on the host in long double for this call length
    uint32_t adv_n, osc_count;
};
struct TailJobs {
    int count, pad_;
    OscAdvance oa;
    TailJob job[kMaxTailJobs];
};

// Demod_WFM::processDataStereo before its pilot PLL drops out (application/demod/demod_wfm.cpp:255-297, :392-429): the constants of one
// demodulator rate and the state one channel carries.  hilb: [2][61] the 61-tap Hilbert pair (I taps, Q taps) in device memory.
struct WfmPilotParams {
    double b0, b2, a1, a2;                 // pilot band-pass biquad (b1 = 0), iir.cpp:131-146
    double nco_lo, nco_hi, alpha, beta;    // initPilotPll, :371-386
    double err_alpha, phase_adjust;
    int block;                             // samples per processDataStereo call (the lock decision is per block)
    int L4;                                // length of the audio response the (L - R) part will go through
};
struct WfmPilotState {
    double d1_re, d1_im;                   // the discriminator's previous sample
    double z[61];                          // the Hilbert filter's delay line (discriminator values), circular
    double w1a, w2a, w1b, w2b;             // pilot band-pass
    double nco_phase, nco_freq, err_ave;
    int zpos;
    int dropped;                           // a block has ended without lock: from here on the block copies the mono signal (see WfmCore)
    long long quiet;                       // (L - R) samples that have been zero at the end of the stream so far, saturating
    int skip, pad_;                        // this call adds nothing to the output (set per call for the FIR behind)
};

// The RDS branch of Demod_WFM::processDataStereo (application/demod/demod_wfm.cpp:296-357, 488-757): constants of one demodulator
// rate, the state one dmFMS channel carries and the record of a group put into m_RdsGroupQueue
struct RdsParams {
    double osc_turns;                      // m_RdsDownConvert's oscillator, turns per demodulator-rate sample
    double nco_lo, nco_hi, alpha, beta;    // processRdsPll, :499-503
    double b0, b2, a1, a2;                 // bit-rate resonator (b1 = 0), :522
    int block;                             // RDS-rate samples per processDataStereo call of the reference
    int mtaps;                             // matched filter
    int log_cap;                           // entries of a channel's group log (a ring)
    int pad_;
};
struct RdsEvent {
    unsigned short a, b, c, d;             // tRDS_GROUPS; all zero: the queue was cleared (:642-647)
    unsigned flags;                        // 1: this entry is the clear + zero group of a lost signal
    long long frame;                       // processDataStereo call (counted from the object's first) that produced it
};
struct RdsState {
    double prev_re, prev_im;               // the discriminator's previous sample
    double nco_phase, nco_freq;
    double w1, w2;                         // resonator
    double last_sync, last_slope, last_data;
    long long n0;                          // samples the down-converter's oscillator has produced
    long long frames;                      // processDataStereo calls so far
    unsigned long long n_events;           // groups logged so far (the log is a ring of log_cap entries)
    unsigned in_bits;                      // m_InBitStream
    int last_bit, bit_pos, cur_block, state, bgroup, block_errors;
    unsigned short block[4];
    int pad_;
};

// A stream still in the device's own sample format (DeviceInterfaceBase::normalizeIQ, pebblelib/deviceinterfacebase.cpp:648-838, done
// in the first loads of the kernels that take it instead of a separate pass): base == nullptr: the float2 pointer is the input.
//   fmt 0 CPX8 int8 pairs, 1 CPXU8 (v - 128), 2 CPX16, 3 CPXFLOAT, 4 WAV PCM16; order 0 IQ, 1 QI, 2 I only, 3 Q only; scale includes the gain
struct RawSrc {
    const void *base;
    int fmt, order;
    float scale;
    int pad_;
};
__device__ __forceinline__ float2 raw_load(const RawSrc &r, long long i)
{
    float a, b;
    if (r.fmt == 0) {
        const char2 v = reinterpret_cast<const char2 *>(r.base)[i];
        a = (float)v.x; b = (float)v.y;
    } else if (r.fmt == 1) {
        const uchar2 v = reinterpret_cast<const uchar2 *>(r.base)[i];
        a = (float)v.x - 128.0f; b = (float)v.y - 128.0f;
    } else if (r.fmt == 2 || r.fmt == 4) {
        const short2 v = reinterpret_cast<const short2 *>(r.base)[i];
        a = (float)v.x; b = (float)v.y;
    } else {
        const float2 v = reinterpret_cast<const float2 *>(r.base)[i];
        a = v.x; b = v.y;
    }
    a *= r.scale;
    b *= r.scale;
    return make_float2((r.order & 1) == 0 ? a : b, (r.order == 1 || r.order == 2) ? a : b);
}

// four consecutive samples i .. i + 3 (i a multiple of 4) in one or two wide loads: 8 bytes for the int8 formats, 16 for int16, 32 for float.
// In two steps, so that a kernel can hold the words as they came (2, 4 or 8 registers) and convert where it uses the samples: a
// conversion behind the load makes the compiler wait for the load there.
template <int FMT>
struct RawQuad {
    uint4 a, b;  // 8-bit formats: a.x, a.y; 16-bit: a; float: a, b
};
template <int FMT>
__device__ __forceinline__ void raw_fetch4(const RawSrc &r, long long i, RawQuad<FMT> &w)
{
    if (FMT == 0 || FMT == 1) {
        const uint2 v = reinterpret_cast<const uint2 *>(r.base)[i >> 2];
        w.a.x = v.x; w.a.y = v.y;
    } else if (FMT == 2 || FMT == 4) {
        w.a = reinterpret_cast<const uint4 *>(r.base)[i >> 2];
    } else {
        w.a = reinterpret_cast<const uint4 *>(r.base)[i >> 1];
        w.b = reinterpret_cast<const uint4 *>(r.base)[(i >> 1) + 1];
    }
}
template <int FMT>
__device__ __forceinline__ void raw_convert4(const RawSrc &r, const RawQuad<FMT> &w, float2 (&o)[4])
{
    float v[8];
    if (FMT == 0 || FMT == 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned word = k < 4 ? w.a.x : w.a.y;
            const int sh = 8 * (k & 3);
            v[k] = FMT == 0 ? (float)((int)(word << (24 - sh)) >> 24) : (float)((word >> sh) & 0xFFu) - 128.0f;
        }
    } else if (FMT == 2 || FMT == 4) {
        const unsigned ww[4] = {w.a.x, w.a.y, w.a.z, w.a.w};
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (float)((int)(ww[k >> 1] << (16 - 16 * (k & 1))) >> 16);
    } else {
        v[0] = __uint_as_float(w.a.x); v[1] = __uint_as_float(w.a.y); v[2] = __uint_as_float(w.a.z); v[3] = __uint_as_float(w.a.w);
        v[4] = __uint_as_float(w.b.x); v[5] = __uint_as_float(w.b.y); v[6] = __uint_as_float(w.b.z); v[7] = __uint_as_float(w.b.w);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float a = v[2 * k] * r.scale, b = v[2 * k + 1] * r.scale;
        o[k] = make_float2((r.order & 1) == 0 ? a : b, (r.order == 1 || r.order == 2) ? a : b);
    }
}
template <int FMT>
__device__ __forceinline__ void raw_load4(const RawSrc &r, long long i, float2 (&o)[4])
{
    RawQuad<FMT> w;
    raw_fetch4<FMT>(r, i, w);
    raw_convert4<FMT>(r, w, o);
}

template <int FMT>
__device__ __forceinline__ void raw_load4_even(const RawSrc &r, long long i, float2 (&o)[4])
{
    float v[8];
    if (FMT == 0 || FMT == 1) {
        unsigned w[2];
        __builtin_memcpy(w, reinterpret_cast<const unsigned char *>(r.base) + 2 * i, 8);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned word = w[k >> 2];
            const int sh = 8 * (k & 3);
            v[k] = FMT == 0 ? (float)((int)(word << (24 - sh)) >> 24) : (float)((word >> sh) & 0xFFu) - 128.0f;
        }
    } else if (FMT == 2 || FMT == 4) {
        unsigned w[4];
        __builtin_memcpy(w, reinterpret_cast<const unsigned char *>(r.base) + 4 * i, 16);
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (float)((int)(w[k >> 1] << (16 - 16 * (k & 1))) >> 16);
    } else {
        const float4 a = reinterpret_cast<const float4 *>(r.base)[i >> 1], b = reinterpret_cast<const float4 *>(r.base)[(i >> 1) + 1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float a = v[2 * k] * r.scale, b = v[2 * k + 1] * r.scale;
        o[k] = make_float2((r.order & 1) == 0 ? a : b, (r.order == 1 || r.order == 2) ? a : b);
    }
}

// Per-channel squelch gate of a bank (receiver.cpp:959-965 per channel): open[c * stride + j] != 0 <=> channel c's super-frame j of
// this call passes.  The kernels behind the band-pass take one super-frame j at a time and leave a closed channel alone
// (no output, no state change).  open == nullptr: no gate.
struct Gate {
    const unsigned char *open;
    int stride, j;
    __host__ __device__ bool closed(int c) const { return open != nullptr && open[(long long)c * stride + j] == 0; }
};

// fdEstimate's bin windows for one channel: noise [nlo, nhi] around the band-pass [lo, hi]; stream = which spectrum it reads
struct SmBins { int nlo, lo, hi, nhi, bp_bins, stream, pad_[2]; };

// k_spectrum_t128<.., DEC = true>: the one-channel mixer + decimator hb11 x 8, hb15, hb23, hb47 (20 Msps -> 312.5 kHz; the halfbands'
// coefficients are literals of the kernel, hb_const.h) computed by the
// display transform's workgroups from the frames they hold in LDS anyway -- the stream crosses HBM once instead of twice, and no
// first-stage buffer is written or read (kernels_spectrum.h)
struct DecFuse {
    float2 *y;               // final outputs, 32 per frame: y[32 f + i]
    float2 *y0_tail;         // the call's last 256 first-stage outputs (for a later call that takes the general kernels)
    const float2 *xtail;     // the WINDOWED frame in front of the call (the previous call's last frame; zeros at the very start)
    float2 *xtail_next;      // this call's last frame, windowed: the next call's xtail
    double phase0, inc;      // oscillator at input sample n of the call: a_inf e^{j 2 pi (phase0 + (n + 1) inc)}
    float a_inf, gain0, gain_last;
    int mix_on;
    int dbg;                 // timing experiments only (PEBBLEGPU_FUSE_DBG): 1 no table / oscillator loads, 2 no first stage, 4 no halfbands
    float2 wfr;              // e^{j 2 pi 2048 inc}: an output's oscillator from one frame to the next
    float2 *ph_scratch;      // [chains][256]: the first stage's oscillator per output, carried from frame to frame of a chain (L2-resident)
    const float2 *c0tab;     // [7][256]: step[d] h0[d] / w[8 jf - 10 + d] for the seven non-zero taps d = 0 2 4 5 6 8 10: the first stage's taps against
                             // WINDOWED samples with the oscillator's advance over the window folded in (rebuilt on a retune)
};

struct SpectrumParams {
    long long in_pitch;      // samples between streams
    long long n_frames;      // frames in this call (per stream)
    int frames_per_group;    // G
    float scale;             // 1 / (coherentGain * maxBinPower), maxBinPower = NF (fft.cpp:84)
    long long out_pitch;     // floats between streams (= n_frames*bins)
};

}  // namespace pg
