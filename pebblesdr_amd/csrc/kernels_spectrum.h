// kernels_spectrum.h -- SignalSpectrum::unprocessed / FFT::fftSpectrum on the device.
//
// Per frame of NF = 2048 samples (framesPerBuffer): BlackmanHarris window, zero-pad to
// bins = ZP*NF, forward FFT, unfold to -f..+f, |X|/(coherentGain*NF), average with the PREVIOUS
// frame's linear amplitude, 20*log10, clip to [-120, 0] dB
// (pebblelib/fft.cpp:129-157, 207-213, 324-399; fftaccelerate.cpp:106-119).
//
// Zero-padding is pruned instead of transformed: with x[n] = 0 for n >= NF,
//   X[ZP*j + q] = FFT_NF( x[n] * w[n] * W_bins^{n*q} )[j],  q = 0..ZP-1
// i.e. ZP independent NF-point transforms of the windowed frame times a per-q twiddle -- log2(ZP) butterfly
// passes over zeros are never executed.
//
// Three kernels share this scheme: k_spectrum<2> below (4096 bins: one wave per transform), k_spectrum_t128 (8192 bins: two
// waves per transform, four waves per SIMD; measured ~8 % faster than the one-wave form at ZP = 4 and ~5 % slower at
// ZP = 2, hence the split) and k_spectrum_1to1 (2048 bins: a wave owns whole frames).
//
// Mapping of k_spectrum<ZP>: one WAVE per (frame, q); the ZP waves of a frame form a group, a 256-item workgroup holds 4/ZP groups.
//   * the group's waves load the NEXT frame once, cooperatively (each wave a 1/ZP slice, coalesced, issued before
//     the current frame's transform so HBM latency hides under it), apply the window and park it in LDS;
//   * each wave gathers the whole windowed frame from LDS in the FFT's strided register layout and applies its
//     twiddle as W^{lane*q} (one per-lane constant) times W^{64*m*q} (wave-uniform, scalar registers);
//   * the 2048-point transform runs inside the wave (32 points per lane, wave-private LDS exchange, no
//     s_barrier -- fft_lds.h);
//   * the wave keeps its q-slice of the previous frame's amplitudes in registers, and parks its 2048 dB values
//     in its own LDS region; after a workgroup barrier the group interleaves the q-slices so each lane stores
//     ZP adjacent bins as one 16-byte vector, fully coalesced.
// A group walks G consecutive frames and recomputes one extra frame (the one before its first) to seed the
// average.  Three workgroup barriers per frame; 72 KiB of exchange regions + 8 KiB twiddle table = 80 KiB LDS, i.e.
// exactly two workgroups per CU.
//
// Bound: HBM (8*NF B in + 4*bins B out per frame) with LDS / fp32 ALU close behind at ZP = 4 (~0.5 Mflop/frame).
// Algorithmic bytes per frame: 8*NF + 4*bins.
#pragma once
#include "fft_lds.h"
#include "fft_t128.h"
#include "fft_w64.h"
#include "hb_const.h"
#include "params.h"

namespace pg {

template <int ZP>
__global__ __launch_bounds__(256, 2) void k_spectrum(const float2 *__restrict__ in, float *__restrict__ out,
                                                      const float *__restrict__ window, const float2 *__restrict__ btab,
                                                      const float2 *__restrict__ tw_nf, const float *__restrict__ prev_in,
                                                      float *__restrict__ prev_out, SpectrumParams sp)
{
    constexpr int NF = 2048, E = NF / 64, BINS = NF * ZP, GROUPS = 4 / ZP, TG = 64 * ZP;
    constexpr int SL = NF / ZP, EL = E / ZP;  // slice of the frame one wave loads, points per lane of it
    // a wave's LDS region: FFT exchange image [0, kSlots), aliased by its dB slice (floats, first 8 KiB) and, behind
    // that, by the windowed slice of the next frame it parks for the group ([XOFF, XOFF+SL))
    constexpr int XOFF = NF / 2;
    constexpr int REGION = (XOFF + SL > FftLds<NF>::kSlots) ? XOFF + SL : FftLds<NF>::kSlots;
    __shared__ float2 lds[4][REGION];
    __shared__ float2 tw_lds[fft_tw_off(NF, NF)];  // per-pass twiddle tables (fft_lds.h), 7 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave / ZP, q = wave % ZP, s = blockIdx.y;
    const int G = sp.frames_per_group;
    const long long f0 = ((long long)blockIdx.x * GROUPS + g) * G;
    const float2 *x = in + (long long)s * sp.in_pitch;
    float *y = out + (long long)s * sp.out_pitch;
    float2 *my = lds[wave];
    float *stage = reinterpret_cast<float *>(my);  // this wave's dB slice, [j] for bins ZP*j + q

    for (int i = threadIdx.x; i < fft_tw_off(NF, NF); i += 256) tw_lds[i] = tw_nf[i];

    // window values of the slice this wave loads: constant over frames, kept in registers when the slice is short
    // (ZP = 4: 8 values); for longer slices they are re-read (L1-resident) when the slice is parked
    constexpr bool WIN_REGS = EL <= 8;
    float win[WIN_REGS ? EL : 1];
    if (WIN_REGS) {
#pragma unroll
        for (int i = 0; i < EL; i++) win[i] = window[q * SL + lane + 64 * i];
    }
    const float *wq = window + q * SL + lane;
    const float2 tw_lane = cis_cycles(-(double)(lane * q) / (double)BINS);  // W_bins^{lane q}
    const float2 *bq = btab + q * E;                                         // W_bins^{64 m q}, m < E (wave-uniform)

    float pa[E];  // previous frame's |X| (unscaled) of bins ZP*(lane+64m)+q
    const float db_off = 6.02059991327962f * __builtin_amdgcn_logf(0.5f * sp.scale);
    float2 xn[EL];
    // prologue: park the first frame this group transforms (f0 - 1, or frame 0 at the call boundary)
    {
        const long long ff = f0 > 0 ? f0 - 1 : 0;
        if (ff < sp.n_frames) {
#pragma unroll
            for (int i = 0; i < EL; i++) xn[i] = x[ff * NF + q * SL + lane + 64 * i];
#pragma unroll
            for (int i = 0; i < EL; i++) my[XOFF + lane + 64 * i] = cscale(xn[i], WIN_REGS ? win[i] : wq[64 * i]);
        }
    }
    __syncthreads();

    for (int it = -1; it < G; it++) {
        const long long f = f0 + it;
        const bool live = f < sp.n_frames;  // wave-uniform
        const bool xform = live && f >= 0;
        // Everything below is addressed from `ln`, an opaque copy of the lane id: without it the compiler hoists
        // ~200 loop-invariant LDS/global addresses and twiddles out of the frame loop and then spills them.
        int ln = lane;
        opaque(ln);
        // the frame the next iteration transforms: issue its loads now so HBM latency hides under this transform
        // (when f0 == 0 iteration -1 transforms nothing and frame 0 is already parked)
        const bool fetch = it + 1 < G && f + 1 < sp.n_frames && f >= 0;
        if (fetch) {
            const float2 *xp = x + (f + 1) * NF + q * SL + ln;
#pragma unroll
            for (int i = 0; i < EL; i++) xn[i] = xp[64 * i];
        }
        float2 v[E];
        if (xform) {
            // sample n = ln + 64 m is parked in the region of wave n / SL = (64 m) / SL of this group
            const float2 *gp = &lds[g * ZP][XOFF + ln];
#pragma unroll
            for (int m = 0; m < E; m++) {
                const float2 xv = gp[((64 * m) / SL) * REGION + (64 * m) % SL];
                v[m] = ZP == 1 ? xv : cmul(bq[m], cmul(tw_lane, xv));  // constants first (see cmul)
            }
        }
        __syncthreads();  // A: every wave holds the frame in registers; regions may be overwritten
        if (live && f < 0) {
            // the frame before this call: amplitudes saved by the previous call (zeros on the first)
            const float *pp = prev_in + (long long)s * BINS + ZP * ln + q;
#pragma unroll
            for (int m = 0; m < E; m++) pa[m] = pp[ZP * 64 * m];
        } else if (xform) {
            fft_regs<NF, +1, 64>(v, my, tw_lds, ln);
            float *st = stage + ln;
            // amplitude = |X| * scale, averaged with the previous frame's (fft.cpp:379-381): 0.5*scale*(|X| + |X'|).
            // pa[] and the prev buffers carry the unscaled |X|; the factor moves into the log as a constant.
            // 20*log10(a) = 6.0206*log2(a); a == 0 gives -inf which the clip turns into -120 (db.h:24-26,44-48).
            // One operation at a time over all 32 bins: sqrt and log are long-latency, so the 32 independent
            // instances of each have to be in flight together (a wave has nothing else to issue meanwhile).
            float mag[E];
#pragma unroll
            for (int m = 0; m < E; m++) mag[m] = v[m].x * v[m].x + v[m].y * v[m].y;
#pragma unroll
            for (int m = 0; m < E; m++) mag[m] = __builtin_amdgcn_sqrtf(mag[m]);
            sched_fence();
#pragma unroll
            for (int m = 0; m < E; m++) {
                const float a = mag[m] + pa[m];
                pa[m] = mag[m];
                mag[m] = a;
            }
#pragma unroll
            for (int m = 0; m < E; m++) mag[m] = __builtin_amdgcn_logf(mag[m]);
            sched_fence();
#pragma unroll
            for (int m = 0; m < E; m++)
                st[64 * m] = fminf(fmaxf(fmaf(6.02059991327962f, mag[m], db_off), -120.f), 0.f);  // for it == -1 nobody reads this slice
            if (f == sp.n_frames - 1) {
                float *pp = prev_out + (long long)s * BINS + ZP * ln + q;
#pragma unroll
                for (int m = 0; m < E; m++) pp[ZP * 64 * m] = pa[m];
            }
        }
        __syncthreads();  // B: dB slices parked, exchanges finished
        if (fetch) {
            float2 *xs = my + XOFF + ln;
#pragma unroll
            for (int i = 0; i < EL; i++) xs[64 * i] = cscale(xn[i], WIN_REGS ? win[i] : wq[64 * i]);
        }
        if (it >= 0 && live) {
            // the ZP waves of this frame interleave their slices: lane tg owns sub-bins j = tg + TG*i
            const int tg = ln + 64 * q;
            const float *sp0 = reinterpret_cast<const float *>(lds[g * ZP]) + tg;
            float *yf = y + f * (long long)BINS;
#pragma unroll
            for (int i = 0; i < NF / TG; i++) {
                const int j = tg + TG * i;
                float db[ZP];
#pragma unroll
                for (int qq = 0; qq < ZP; qq++) db[qq] = sp0[qq * (REGION * 2) + TG * i];
                // bin k = ZP*j + qq unfolds to (k + BINS/2) mod BINS (fft.cpp:207-213); the ZP bins stay adjacent
                const int u = (ZP * j + BINS / 2) & (BINS - 1);
                if (ZP == 4) store_stream(reinterpret_cast<float4 *>(yf + u), make_float4(db[0], db[1], db[2], db[3]));
                else if (ZP == 2) store_stream(reinterpret_cast<float2 *>(yf + u), make_float2(db[0], db[1]));
                else store_stream(yf + u, db[0]);
            }
        }
        __syncthreads();  // C: next frame parked, dB slices consumed
    }
}

// ------------------------------------------------------------------------------------------------
// 8192 bins on the two-wave transform (fft_t128.h): a workgroup of 512 = 4 transforms (q = 0..3) x 2 waves handles one
// frame at a time; 16 points and 16 previous amplitudes per work-item instead of 32 + 32, so four waves fit a SIMD
// (LDS: the same four 18 KiB exchange images, two workgroups per CU).  Seven workgroup barriers per frame.
// grid (ceil(F / G) / HALVES, S), block 512 * HALVES.
// ------------------------------------------------------------------------------------------------
//
// HALVES = 2: a 1024-item workgroup = two such halves, each with its own frame chain and its own four images (one workgroup
// per CU, 152 KiB of LDS).  Two 512-item workgroups sharing a CU run in lockstep -- both in their arithmetic phase, then both
// in their LDS phase: a workgroup alone takes 0.33 ms for the bench batch, two per CU 0.30 (profiles/r02_spectrum_sq_before.txt:
// vector ALU busy 63 %, LDS 41 %, the frame time their SUM).  Here the barriers are common to both halves and the second half
// runs `shift` barrier intervals behind the first, so that in every interval one half's butterflies meet the other's exchange:
// the offset is fixed by construction, not left to the dispatcher.
// FMT >= 0: the frames are read in the device's own sample format (RawSrc, pebblegpu_iq_format FMT) and converted in the load --
// normalizeIQ without a float2 copy of the stream (2 instead of 8 bytes per sample from HBM for HackRF / RTL int8 pairs).  A
// work-item then takes four CONSECUTIVE samples (one 8- or 16-byte load) instead of four samples 512 apart.
// DEC: the workgroup also runs the one-channel mixer + decimator (hb11 x 8, hb15, hb23, hb47) over the frames it parks in LDS for
// the transforms -- the stream crosses HBM once instead of twice and nothing is written at the intermediate rates.  A half's chain
// of frames is a chain for the decimator as well: the frame in front of it (loaded anyway for the previous amplitudes; for the call's
// first chain the previous call's last frame, DecFuse::xtail) is the cascade's whole look-back (1946 samples).  The stages run in the
// barrier intervals the transform already has, each on other waves: first stage from the parked (windowed) frame with taps h[d] / w[n]
// before barrier A, hb15 / hb23 / hb47 after the transform's first three exchanges barriers, through three small LDS arrays whose
// heads hold the previous frame's last T - 1 outputs.
struct DecLds {
    float2 z0[14 + 256], z1[22 + 128], z2[2][46 + 64], xt[2][10];  // (z2 twice: hb47 runs a frame behind hb23, which is writing the next one)
};

// FULLC: pass C's fifteen twiddles per work-item formed once and held (126 registers: four such waves fill a SIMD's register file --
// for calls whose first decimator stage does not run beside this kernel, Receiver::process).
template <int HALVES, int FMT, bool DEC = false, bool FULLC = false>
static __global__ __launch_bounds__(512 * HALVES, 2 * HALVES) void k_spectrum_t128(const float2 *__restrict__ in, float *__restrict__ out,
                                                                 const float *__restrict__ window, const float2 *__restrict__ btab128,
                                                                 const float2 *__restrict__ tw128, const float *__restrict__ prev_in,
                                                                 float *__restrict__ prev_out, SpectrumParams sp, int shift, RawSrc raw,
                                                                 DecFuse df)
{
    constexpr int NF = 2048, ZP = 4, BINS = NF * ZP, E = 16, SL = NF / ZP;
    constexpr int REGION = FftLds<NF>::kSlots;
    // products of pass C held beside the seven table entries: what fits under 112 registers (the chain's first stage needs the other
    // 64 of a SIMD's 512); raw float samples wait for their conversion in eight registers, not four
    constexpr int HELD = (DEC || FULLC) ? 0 : (FMT == 3 ? 5 : FMT >= 0 ? 7 : 6);  // (the integer formats wait in two or four)
    // The parked frame: sample n sits in region n >> 9 at slot XOFF + m (m = n & 511).  DEC: at XOFF + m + (m >> 3) -- one pad slot
    // per eight samples, so that the first decimator stage's stride-8 reads (all of one residue mod 8: four banks of a plain layout)
    // spread over the banks with an address that stays affine in the lane -- and each region's last ten samples once more in front
    // of the next region's first (m = -10 .. -1), so a window never straddles two regions.
    constexpr int XOFF = NF / 2 + (DEC ? 16 : 0), RSTEP = DEC ? 144 : 128;
    auto fslot = [](int m) { return XOFF + m + (DEC ? (m >> 3) : 0); };
    constexpr bool RAW = FMT >= 0;
    __shared__ float2 lds_all[HALVES][4][REGION];
    __shared__ DecLds dec_all[DEC ? HALVES : 1];
    __builtin_amdgcn_s_setprio(3);  // in front of the first stage's waves beside it: that kernel has slack, this one is the call's length
    const int tid = threadIdx.x & 511, lane = tid & 63;
    const int half = HALVES > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 9) : 0;
    float2 (*lds)[REGION] = lds_all[half];
    DecLds &dl = dec_all[DEC ? half : 0];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wave >> 1, s = blockIdx.y;
    const int G = sp.frames_per_group;
    const long long f0 = ((long long)blockIdx.x * HALVES + half) * G;
    const float2 *x = in + (long long)s * sp.in_pitch;
    float *y = out + (long long)s * sp.out_pitch;
    float2 *my = lds[q];
    float *stage = reinterpret_cast<float *>(my);
    float win[4];
#pragma unroll
    for (int i = 0; i < 4; i++) win[i] = RAW ? window[4 * tid + i] : window[tid + 512 * i];
    const int t0 = (wave & 1) * 64 + lane;                                   // work-item of its transform, 0..127
    // The pre-twiddle W_bins^{n q}, n = t + 128 m: its wave-uniform part W_bins^{128 m q} multiplies the samples; its per-work-item part
    // W_bins^{t q} costs nothing -- it is folded into the transform's own twiddles (tw128: one table per q, make_twiddles_t128q), which
    // a work-item reads once and keeps
    const float2 *bq = btab128 + q * E;                                      // W_bins^{128 m q}, m < 16 (wave-uniform)
    Tw128Regs twr;
    {
        const float2 *tq = tw128 + q * kTw128Count;
        const int kb = t0 & 15;
        twr.b1 = tq[kTw128B + kb]; twr.b2 = tq[kTw128B + 16 + kb]; twr.b4 = tq[kTw128B + 32 + kb];
        twr.c1 = tq[kTw128C + t0]; twr.c2 = tq[kTw128C + 128 + t0]; twr.c4 = tq[kTw128C + 256 + t0]; twr.c8 = tq[kTw128C + 384 + t0];
        if (FULLC || HELD > 0) tw128_fill_cx(twr);
    }
    const float db_off = 6.02059991327962f * __builtin_amdgcn_logf(0.5f * sp.scale);
    float pa[E];
    float2 xn[4];
    RawQuad<FMT < 0 ? 0 : FMT> xr;  // RAW: the next frame's samples as fetched; converted where they are parked (park)
    constexpr int kD[7] = {0, 2, 4, 5, 6, 8, 10};  // the hb11's non-zero taps (DEC)
    if (DEC) {
        for (int i = tid; i < (int)(sizeof(DecLds) / sizeof(float2)); i += 512) reinterpret_cast<float2 *>(&dl)[i] = make_float2(0.f, 0.f);
    }
    // the first stage's taps (against windowed samples, times the oscillator's step per tap) and its oscillator for the coming frame:
    // requested one barrier interval ahead, by the four waves that use them (the others get zeros: a conditional request alone would
    // keep the old values alive through the transform)
    float2 c0n[7], phn;
    auto dec_prefetch = [&](int td2) {
#pragma unroll
        for (int i = 0; i < 7; i++) c0n[i] = make_float2(0.f, 0.f);
        phn = make_float2(0.f, 0.f);
        if (td2 < 256 && !(df.dbg & 1)) {
#pragma unroll
            for (int i = 0; i < 7; i++) c0n[i] = df.c0tab[256 * i + td2];
            phn = df.ph_scratch[((long long)blockIdx.x * HALVES + half) * 256 + td2];
        }
    };
    if (DEC) dec_prefetch(tid);
    auto park = [&](int tt) {  // sample n = tt + 512 i goes to region i, slot XOFF + tt (RAW: n = 4 tt + i: region tt / 128, four adjacent slots)
        if (RAW) raw_convert4<FMT < 0 ? 0 : FMT>(raw, xr, xn);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int reg = RAW ? tt >> 7 : i, m = RAW ? (4 * tt + i) & 511 : tt;
            const float2 v = cscale(xn[i], win[i]);
            lds[reg][fslot(m)] = v;
            if (DEC && m >= 502 && reg < 3) lds[reg + 1][fslot(m - 512)] = v;
        }
    };
    {
        const long long ff = f0 > 0 ? f0 - 1 : 0;
        if (DEC && f0 == 0) {
            // the call's first chain: the frame in front of the call, kept windowed by the previous call's last workgroup
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float2 v = df.xtail[tid + 512 * i];
                lds[i][fslot(tid)] = v;
                if (tid >= 502 && i < 3) lds[i + 1][fslot(tid - 512)] = v;
            }
        } else if (ff < sp.n_frames) {
            if (RAW) {
                raw_fetch4<FMT < 0 ? 0 : FMT>(raw, (long long)s * sp.in_pitch + ff * NF + 4 * tid, xr);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) xn[i] = x[ff * NF + tid + 512 * i];
            }
            park(tid);
        }
    }
    __syncthreads();
    if (HALVES > 1 && half == 1)
        for (int i = 0; i < shift; i++) __syncthreads();  // the second half runs `shift` intervals behind (same barrier count overall)
    // One iteration; `pa` holds the previous frame's magnitudes on entry and `pn` takes this frame's.  The loop below calls it twice per
    // round with the two arrays exchanged: a magnitude is then produced in the registers it is read from a frame later, and the
    // sixteen register moves a single carried array costs per frame are gone (DEC: one array, in place -- its loop is the longer one).
    auto iteration = [&](const int it, float (&pa)[E], float (&pn)[E]) __attribute__((always_inline)) {
        const long long f = f0 + it;
        const bool live = f < sp.n_frames;   // workgroup-uniform
        const bool xform = live && f >= 0;
        int t = t0, td = tid;
        opaque(t);
        opaque(td);
        // (DEC: the call's first chain starts on the frame in front of the call, so its frame 0 is fetched like any other)
        const bool fetch = it + 1 < G && f + 1 < sp.n_frames && (f >= 0 || DEC);
        auto fetch_next = [&]() {
            if (!fetch) return;
            if (RAW) {
                raw_fetch4<FMT < 0 ? 0 : FMT>(raw, (long long)s * sp.in_pitch + (f + 1) * NF + 4 * td, xr);
            } else {
                const float2 *xp = x + (f + 1) * NF + td;
#pragma unroll
                for (int i = 0; i < 4; i++) xn[i] = xp[512 * i];
            }
        };
        // (DEC: the first decimator stage below waits for its prefetched taps; vector-memory results come back in issue order, so the
        // next frame is requested behind it -- in front it would make that wait last until the frame has come in from HBM)
        if (!DEC) fetch_next();
        float2 v[E];
        if (xform) {
            const float2 *gp = &lds[0][fslot(t)];
#pragma unroll
            for (int m = 0; m < E; m++) {
                const float2 xv = gp[(m / 4) * REGION + (m % 4) * RSTEP];
                v[m] = m == 0 ? xv : cmul(bq[m], xv);  // (bq[0] = 1)
            }
        }
        const bool dec_run = DEC && live;          // (f == -1 included: the look-back frame)
        const bool dec_emit = dec_run && it >= 0;  // outputs of the chain's own frames leave the kernel
        const int par = (int)(f & 1);
        if (dec_run && !(df.dbg & 2)) {
            if (td < 256) {
                // first stage: y0[jf] = ph * sum_d (h[d] step[d]) x[8 jf - 10 + d], the samples read windowed and un-windowed by the tap
                // (taps and oscillator come from memory every frame: nothing of the decimator is held in registers through the transform)
                float2 ph = make_float2(df.gain0, 0.f);
                if (df.mix_on) {
                    // (exact every eighth frame, one rotation per frame in between; carried in memory, not in registers)
                    ph = ((it + 1) & 7) == 0 ? cscale(cis_cycles(df.phase0 + (double)((long long)NF * f + 8 * td - 9) * df.inc), df.a_inf * df.gain0) : cmul(df.wfr, phn);
                    df.ph_scratch[((long long)blockIdx.x * HALVES + half) * 256 + td] = ph;
                }
                // lane l of wave w: samples 512 w + 8 l + c, c = d - 10 <= 0: region w, slot XOFF + 9 l + c + floor(c / 8) -- affine, also
                // for the few in front of the region (its copy of the previous region's end; for w = 0 the previous frame's: xt)
                const float2 *fb = &lds[td >> 6][XOFF + 9 * (td & 63)];
                v2f_t acc = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    const int c = kD[i] - 10, n = 8 * td + c;
                    const float2 xs = n >= 0 ? fb[c + (c >> 3)] : dl.xt[par ^ 1][n + 10];
                    acc = cmac_pk(acc, v2f_t{c0n[i].x, c0n[i].y}, v2f_t{xs.x, xs.y});
                }
                const float2 y0 = cmul(ph, make_float2(acc.x, acc.y));
                dl.z0[14 + td] = y0;
                if (dec_emit && f == sp.n_frames - 1) df.y0_tail[td] = y0;
            } else if (td < 266) {
                dl.xt[par][td - 256] = lds[3][fslot(502 + td - 256)];
            }
        }
        // The decimator's later stages in the transform's barrier intervals, each on waves of its own.  Interval 1: hb15 of this frame
        // (waves 4, 5) and hb47 of the PREVIOUS frame (wave 7: its input was finished by hb23 an interval later, last frame);
        // interval 2: hb23 (wave 6) and the first array's head for the next frame; interval 3: the second array's head.
        auto hb47 = [&](long long fr, bool emit) {  // 32 outputs of frame fr, the 24 + 1 taps of each split over two lanes 32 apart
            // (the filter is symmetric: the upper lane walks its 12 taps 46, 44 .. 24 with the lower lane's coefficients 0, 2 .. 22 -- literals)
            const int m = (td - 448) & 31, part = (td - 448) >> 5;
            const float2 *z = dl.z2[(int)(fr & 1)];
            const float2 *w = z + 2 * m + 46 * part;
            const int sg = 1 - 2 * part;
            float2 acc = cscale(z[2 * m + 23], part ? 0.f : hb_tap<47>(23));  // the centre with the lower half
#pragma unroll
            for (int p = 0; p < 24; p += 2) acc = cadd(acc, cscale(w[sg * p], hb_tap<47>(p)));
            acc.x += __shfl_down(acc.x, 32);
            acc.y += __shfl_down(acc.y, 32);
            if (emit && part == 0) df.y[32 * fr + m] = cscale(acc, df.gain_last);
        };
        auto dec_slot = [&](int k) {
            if (df.dbg & 4) return;
            if (k == 1) {
                if (dec_run && td >= 256 && td < 384) {  // hb15 over z0
                    const int m = td - 256;
                    const float2 *w = dl.z0 + 2 * m;
                    float2 acc = cscale(w[7], hb_tap<15>(7));
#pragma unroll
                    for (int p = 0; p < 15; p += 2) acc = cadd(acc, cscale(w[p], hb_tap<15>(p)));
                    dl.z1[22 + m] = acc;
                }
                // (frame f - 1 went through hb23 an iteration ago; it >= 1: it is one of this chain's own frames)
                if (DEC && td >= 448 && it >= 1 && f - 1 < sp.n_frames) hb47(f - 1, true);
            } else if (k == 2) {
                if (!dec_run) return;
                if (td >= 384 && td < 448) {  // hb23 over z1, into this frame's z2 and (its last 46) the head of the next frame's
                    const int m = td - 384;
                    const float2 *w = dl.z1 + 2 * m;
                    float2 acc = cscale(w[11], hb_tap<23>(11));
#pragma unroll
                    for (int p = 0; p < 23; p += 2) acc = cadd(acc, cscale(w[p], hb_tap<23>(p)));
                    dl.z2[par][46 + m] = acc;
                    if (m >= 18) dl.z2[par ^ 1][m - 18] = acc;
                } else if (td >= 448 && td < 462) {
                    dl.z0[td - 448] = dl.z0[256 + td - 448];  // its head for the next frame
                }
            } else if (k == 3) {
                if (dec_run && td >= 384 && td < 406) dl.z1[td - 384] = dl.z1[128 + td - 384];  // its head for the next frame
            }
        };
        if (DEC) fetch_next();
        __syncthreads();  // A
        if (live && f < 0) {
            const float *pp = prev_in + (long long)s * BINS + ZP * t + q;
#pragma unroll
            for (int m = 0; m < E; m++) pa[m] = pp[ZP * 128 * m];
        }
        // every work-item of the workgroup takes the same path below (xform is uniform): the barriers inside match
        if (xform) {
            int slot = 0;
            auto bar = [&] {
                __syncthreads();
                if (DEC) dec_slot(++slot);
            };
            fft2048_t128<decltype(bar), true, true, true, FULLC, HELD>(v, my, tw128, t, bar, twr);
            float *st = stage + t;
            float mag[E];
#pragma unroll
            for (int m = 0; m < E; m++) mag[m] = v[m].x * v[m].x + v[m].y * v[m].y;
#pragma unroll
            for (int m = 0; m < E; m++) mag[m] = __builtin_amdgcn_sqrtf(mag[m]);
#pragma unroll
            for (int m = 0; m < E; m++) {
                const float a = mag[m] + pa[m];
                pn[m] = mag[m];
                mag[m] = a;
            }
#pragma unroll
            for (int m = 0; m < E; m++) mag[m] = __builtin_amdgcn_logf(mag[m]);
#pragma unroll
            for (int m = 0; m < E; m++) st[128 * m] = fminf(fmaxf(fmaf(6.02059991327962f, mag[m], db_off), -120.f), 0.f);
            if (f == sp.n_frames - 1) {
                float *pp = prev_out + (long long)s * BINS + ZP * t + q;
#pragma unroll
                for (int m = 0; m < E; m++) pp[ZP * 128 * m] = pn[m];
            }
        } else {
            if (HALVES > 1 || DEC) {  // the halves share every barrier: an iteration without a transform still passes its four
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    __syncthreads();
                    if (DEC) dec_slot(i + 1);
                }
            }
            if (!DEC) {
#pragma unroll
                for (int m = 0; m < E; m++) pn[m] = pa[m];
            }
        }
        __syncthreads();  // B
        if (DEC) dec_prefetch(td);
        if (fetch) park(td);
        if (DEC && fetch && f + 1 == sp.n_frames - 1) {
            // the call's last frame, as parked: the next call's look-back (its samples are this work-item's own four)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int n = RAW ? 4 * td + i : td + 512 * i;
                df.xtail_next[n] = cscale(xn[i], win[i]);
            }
        }
        if (it >= 0 && live) {
            float *yf = y + f * (long long)BINS;
            // (an index of its own per store: the four reads of one store then pair up across the regions, (x, y) and (z, w), and
            //  land in the store's registers; from one base the compiler pairs them across the stores and spends three moves
            //  per store on sorting them)
            int j[NF / 512];
#pragma unroll
            for (int i = 0; i < NF / 512; i++) {
                j[i] = td + 512 * i;
                opaque(j[i]);
            }
            float4 d[NF / 512];
#pragma unroll
            for (int i = 0; i < NF / 512; i++) {
                const float *spi = reinterpret_cast<const float *>(lds[0]) + j[i];
                d[i].x = spi[0 * (REGION * 2)];
                d[i].y = spi[1 * (REGION * 2)];
                d[i].z = spi[2 * (REGION * 2)];
                d[i].w = spi[3 * (REGION * 2)];
            }
#pragma unroll
            for (int i = 0; i < NF / 512; i++) {
                const int u = (ZP * j[i] + BINS / 2) & (BINS - 1);
                store_stream(reinterpret_cast<float4 *>(yf + u), d[i]);
            }
        }
        __syncthreads();  // C
    };
    if (DEC) {
        for (int it = -1; it < G; it++) iteration(it, pa, pa);
    } else {
        float pb[E];
        for (int it = -1; it < G; it += 2) {
            iteration(it, pa, pb);
            if (it + 1 >= G) break;
            iteration(it + 1, pb, pa);
        }
    }
    if (DEC) {  // hb47 of the chain's last frame (its input complete since the last iteration's second interval, several barriers ago)
        const long long fl = f0 + G - 1;
        if (tid >= 448 && fl < sp.n_frames) {
            const int m = (tid - 448) & 31, part = (tid - 448) >> 5;
            const float2 *z = dl.z2[(int)(fl & 1)];
            const float2 *w = z + 2 * m + 46 * part;
            const int sg = 1 - 2 * part;
            float2 acc = cscale(z[2 * m + 23], part ? 0.f : hb_tap<47>(23));
#pragma unroll
            for (int p = 0; p < 24; p += 2) acc = cadd(acc, cscale(w[sg * p], hb_tap<47>(p)));
            acc.x += __shfl_down(acc.x, 32);
            acc.y += __shfl_down(acc.y, 32);
            if (part == 0) df.y[32 * fl + m] = cscale(acc, df.gain_last);
        }
    }
    if (HALVES > 1 && half == 0)
        for (int i = 0; i < shift; i++) __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// 8192 bins on the ONE-wave transform of fft_w64.h (radix 32 * 4 * 16: the middle pass on lane swaps, one LDS exchange).
//
// A 256-item workgroup = the four transforms (q = wave) of one frame at a time, a chain of G frames.  Per frame and wave: 32
// samples from the parked frame, the wave-uniform pre-twiddle W_128^{n1 q} (scalar operands), the transform in the wave's own
// 8.5 KiB image (no barrier inside), amplitude / average / dB, the dB values parked in that same image; then all four waves
// store 16-byte groups of the four q.  TWO workgroup barriers per frame against seven in k_spectrum_t128, 5.2 M LDS instructions
// per bench step against 8.6 M, the same 77 M vector instructions -- and the same time: at 1.3 kW the chip is at its power limit
// under either kernel, and the clock gives back whatever the schedule gains (profiles/r02_power_clock.txt).
// Opt-in (PEBBLEGPU_SPECTRUM_W64=1): two workgroups per CU (51 KiB of LDS, 164 registers) leave room for the chain's LDS
// kernels beside them, which k_spectrum_t128's 152 KiB do not.  grid (ceil(F / G), S), block 256.
// ------------------------------------------------------------------------------------------------
template <int FMT>
static __global__ __launch_bounds__(256, 2) void k_spectrum_w64(const float2 *__restrict__ in, float *__restrict__ out, const float *__restrict__ window,
                                                                 const float2 *__restrict__ btab, const float *__restrict__ prev_in,
                                                                 float *__restrict__ prev_out, SpectrumParams sp, RawSrc raw)
{
    constexpr int NF = 2048, ZP = 4, BINS = NF * ZP, E = 32, QSTRIDE = kW64ImageSlots * 2;
    constexpr bool RAW = FMT >= 0;
    __shared__ float2 frame[NF];
    __shared__ float2 img[ZP][kW64ImageSlots];
    __shared__ float2 stab[ZP][32];  // W_128^{n1 q}: the wave-uniform part of the pre-twiddle W_8192^{n q}, n = 64 n1 + lane
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6), s = blockIdx.y;
    const int G = sp.frames_per_group;
    const long long f0 = (long long)blockIdx.x * G;
    const float2 *x = in + (long long)s * sp.in_pitch;
    float *y = out + (long long)s * sp.out_pitch;
    // a work-item parks samples tid + 256 i (RAW: 4 tid + 1024 (i >> 2) + (i & 3): two runs of four adjacent samples)
    const W64Consts c = w64_consts(lane, cis_cycles(-(double)(lane * q) / (double)BINS));
    const float db_off = 6.02059991327962f * __builtin_amdgcn_logf(0.5f * sp.scale);
    const int kb = w64_kbase(lane);
    if (tid < ZP * 32) stab[tid >> 5][tid & 31] = btab[tid];
    float pa[E];
    float2 xn[8];
    float win[8];
#ifdef PG_W64_PROFILE  // tools/ubench/spectrum_phases.hip: cycles per phase of the frame loop, summed per wave (raw.base = the output table)
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = 0;
#define PG_W64_TICK(i) { const long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tl; tl = t_; }
#else
#define PG_W64_TICK(i)
#endif  // fetched with the frame (L1 hits) rather than held through the transform: the registers are needed there
    auto fetch_window = [&](int td) {
#pragma unroll
        for (int i = 0; i < 8; i++) win[i] = RAW ? window[4 * td + 1024 * (i >> 2) + (i & 3)] : window[td + 256 * i];
    };
    auto fetch_frame = [&](long long ff, int td) {
        if (RAW) {
            const long long i0 = (long long)s * sp.in_pitch + ff * NF + 4 * td;
            float2 a[4], b[4];
            raw_load4<FMT < 0 ? 0 : FMT>(raw, i0, a);
            raw_load4<FMT < 0 ? 0 : FMT>(raw, i0 + 1024, b);
#pragma unroll
            for (int i = 0; i < 4; i++) { xn[i] = a[i]; xn[4 + i] = b[i]; }
        } else {
            const float2 *xp = x + ff * NF + td;
#pragma unroll
            for (int i = 0; i < 8; i++) xn[i] = xp[256 * i];
        }
    };
    auto park = [&](int td) {
#pragma unroll
        for (int i = 0; i < 8; i++) frame[RAW ? 4 * td + 1024 * (i >> 2) + (i & 3) : td + 256 * i] = cscale(xn[i], win[i]);
    };
    {
        const long long ff = f0 > 0 ? f0 - 1 : 0;
        if (ff < sp.n_frames) {
            fetch_frame(ff, tid);
            fetch_window(tid);
            park(tid);
        }
    }
    __syncthreads();
#ifdef PG_W64_PROFILE
    tl = __builtin_readcyclecounter();
#endif
    for (int it = -1; it < G; it++) {
        const long long f = f0 + it;
        const bool live = f < sp.n_frames;   // workgroup-uniform
        const bool xform = live && f >= 0;
        int ln = lane, td = tid, qq = q;
        opaque(ln);
        opaque(td);
        asm volatile("" : "+s"(qq));
        const bool fetch = it + 1 < G && f + 1 < sp.n_frames && f >= 0;
        // the next frame is requested whether or not it will be used (a clamped index, no branch): a conditional request is a
        // block of its own that the compiler moves up into the transform, where its registers do not exist
        const long long fnext = f + 1 < sp.n_frames ? (f + 1 > 0 ? f + 1 : 0) : sp.n_frames - 1;
        fetch_frame(fnext, td);  // (164 registers: two such workgroups leave a SIMD 176 for the chain's kernels beside them)
        if (live && f < 0) {
            int kq = ZP * kb + q;
            opaque(kq);  // (addresses of the two rare paths are not worth 64 registers across the loop)
            const float *pp = prev_in + (long long)s * BINS + kq;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int k3 = 0; k3 < 16; k3++) pa[16 * h + perm16(k3)] = pp[ZP * (32 * h + 128 * k3)];
        }
        if (xform) {
            float2 v[E];
            const float2 *gp = frame + ln;
            const float2 *sq = stab[qq];  // one address for the whole wave: a broadcast read, two registers until the product
            v[0] = gp[0];
#pragma unroll
            for (int m = 1; m < E; m++) v[m] = cmul_pk(sq[m], gp[64 * m]);
            PG_W64_TICK(0)
            fft2048_w64(v, img[q], c, ln);
            PG_W64_TICK(1)
            sched_fence();
            fetch_window(td);  // (L1 hits; not held through the transform: the registers are needed there)
            sched_fence();
            float *st = reinterpret_cast<float *>(img[q]) + w64_kbase(ln);
#pragma unroll
            for (int g = 0; g < 4; g++) {  // eight values at a time: they are in flight together, not all 32
                constexpr int NG = 8;
                const int h = g >> 1, kk = (g & 1) * NG;
                float mag[NG];
#pragma unroll
                for (int m = 0; m < NG; m++) {
                    const float2 z = v[16 * h + perm16(kk + m)];
                    mag[m] = z.x * z.x + z.y * z.y;
                }
#pragma unroll
                for (int m = 0; m < NG; m++) mag[m] = __builtin_amdgcn_sqrtf(mag[m]);
#pragma unroll
                for (int m = 0; m < NG; m++) {
                    const float a = mag[m] + pa[16 * h + perm16(kk + m)];
                    pa[16 * h + perm16(kk + m)] = mag[m];
                    mag[m] = a;
                }
#pragma unroll
                for (int m = 0; m < NG; m++) mag[m] = __builtin_amdgcn_logf(mag[m]);
#pragma unroll
                for (int m = 0; m < NG; m++)
                    st[32 * h + 128 * (kk + m)] = fminf(fmaxf(fmaf(6.02059991327962f, mag[m], db_off), -120.f), 0.f);
                sched_fence();
            }
            if (f == sp.n_frames - 1) {
                int kq = ZP * kb + q;
                opaque(kq);
                float *pp = prev_out + (long long)s * BINS + kq;
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int k3 = 0; k3 < 16; k3++) pp[ZP * (32 * h + 128 * k3)] = pa[16 * h + perm16(k3)];
            }
        }
        PG_W64_TICK(2)
        __syncthreads();  // B: dB values parked, the frame consumed
        PG_W64_TICK(3)
        if (fetch) park(td);
        if (it >= 0 && live) {
            // bins j, j + 1 (j = 2 td + 512 i) of the four q: four 8-byte reads, two adjacent 16-byte stores
            const float2 *sp0 = reinterpret_cast<const float2 *>(img[0]) + td;
            float *yf = y + f * (long long)BINS;
#pragma unroll
            for (int i = 0; i < NF / 512; i++) {
                const int j = 2 * td + 512 * i;
                const float2 d0 = sp0[0 * (QSTRIDE / 2) + 256 * i], d1 = sp0[1 * (QSTRIDE / 2) + 256 * i];
                const float2 d2 = sp0[2 * (QSTRIDE / 2) + 256 * i], d3 = sp0[3 * (QSTRIDE / 2) + 256 * i];
                // bin k = ZP*j + q unfolds to (k + BINS/2) mod BINS (fft.cpp:207-213); j is even: the pair never straddles the fold
                float4 *dst = reinterpret_cast<float4 *>(yf + ((ZP * j + BINS / 2) & (BINS - 1)));
                store_stream(dst, make_float4(d0.x, d1.x, d2.x, d3.x));
                store_stream(dst + 1, make_float4(d0.y, d1.y, d2.y, d3.y));
            }
        }
        PG_W64_TICK(4)
        __syncthreads();  // C: next frame parked, dB values consumed
        PG_W64_TICK(5)
    }
#ifdef PG_W64_PROFILE
    if (lane == 0) {
        long long *o = reinterpret_cast<long long *>(const_cast<void *>(raw.base)) + ((long long)blockIdx.x * 4 + q) * 8;
        for (int i = 0; i < 8; i++) o[i] = tacc[i];
    }
#endif
#undef PG_W64_TICK
}

// ------------------------------------------------------------------------------------------------
// One (frame chain, q) per 128-item workgroup -- any bins = ZP * 2048, ZP = 1, 2, 4, 8, 16.
//
// The 512-item kernel above shares a loaded frame between its four transforms through LDS and interleaves their bins
// through LDS again; that costs it seven workgroup-wide barriers per frame, and its counters (profiles/r02_spectrum_sq_before.txt)
// show the two workgroups of a CU moving in lockstep: vector ALU busy 63 %, LDS 41 %, the frame time their SUM.  Here a
// workgroup is ONE transform: the two waves that share its exchange image are the only ones its barriers involve, eight
// such workgroups (from different chains and phases) fill a CU, and nothing is shared between the ZP transforms of a frame:
//   * each reads the frame itself (coalesced 8-byte loads; the ZP workgroups of a chain sit on one XCD under round-robin
//     placement -- blockIdx = 8 * (ZP * chain_hi + q) + chain_lo -- so the re-reads are L2 hits; placement is speed only);
//   * window and pre-twiddle are ONE factor per point, F[q][n] = w[n] * W_bins^{n q} (host table, fp64-rounded), held in
//     registers for the whole chain: one complex product per point instead of a scale and two products;
//   * a work-item stores its 16 dB values itself, bin ZP * j + q (4-byte stores, ZP * 4 bytes apart: the ZP workgroups of
//     the frame fill each line within microseconds of each other and the XCD's L2 merges them before write-back).
// grid (8 * ZP * ceil(chains / 8), S), block 128; LDS 18 KiB.
// ------------------------------------------------------------------------------------------------
template <bool F_REGS /* hold the per-point factors in registers for the whole chain (else re-read them per frame: L1/L2 hits) */>
static __global__ __launch_bounds__(128, 4) void k_spectrum_q128(const float2 *__restrict__ in, float *__restrict__ out,
                                                                 const float2 *__restrict__ ftab, const float2 *__restrict__ tw128,
                                                                 const float *__restrict__ prev_in, float *__restrict__ prev_out,
                                                                 SpectrumParams sp, int zp_log2)
{
    constexpr int NF = 2048, E = 16;
    __shared__ float2 lds[FftLds<NF>::kSlots];
    const int t0 = threadIdx.x, s = blockIdx.y;
    const int ZP = 1 << zp_log2, BINS = NF << zp_log2;
    const int r = blockIdx.x >> 3;
    const int q = r & (ZP - 1);
    const long long chain = (long long)(r >> zp_log2) * 8 + (blockIdx.x & 7);
    const int G = sp.frames_per_group;
    const long long f0 = chain * G;
    if (f0 >= sp.n_frames) return;  // workgroup-uniform (grid padding)
    const float2 *x = in + (long long)s * sp.in_pitch;
    float *y = out + (long long)s * sp.out_pitch;
    float2 F[F_REGS ? E : 1];
    if (F_REGS) {
#pragma unroll
        for (int m = 0; m < E; m++) F[m] = ftab[q * NF + t0 + 128 * m];
    }
    const float db_off = 6.02059991327962f * __builtin_amdgcn_logf(0.5f * sp.scale);
    // bin ZP*j + q of j = t + 128 m unfolds to (k + BINS/2) mod BINS (fft.cpp:207-213): m < 8 lands in the upper half
    float pa[E];
#pragma unroll
    for (int m = 0; m < E; m++) pa[m] = 0.f;
    for (int it = -1; it < G; it++) {
        const long long f = f0 + it;
        if (f >= sp.n_frames) break;  // workgroup-uniform
        int t = t0;
        opaque(t);
        if (f < 0) {  // the frame before this call: amplitudes saved by the previous call (zeros on the first)
            const float *pp = prev_in + (long long)s * BINS + (t << zp_log2) + q;
#pragma unroll
            for (int m = 0; m < E; m++) pa[m] = pp[(128 * m) << zp_log2];
            continue;
        }
        float2 v[E];
        {
            const float2 *xp = x + f * NF + t;
#pragma unroll
            for (int m = 0; m < E; m++) v[m] = xp[128 * m];
            if (F_REGS) {
#pragma unroll
                for (int m = 0; m < E; m++) v[m] = cmul(F[m], v[m]);
            } else {
                const float2 *fp = ftab + q * NF + t;
                float2 fm[E];
#pragma unroll
                for (int m = 0; m < E; m++) fm[m] = fp[128 * m];
#pragma unroll
                for (int m = 0; m < E; m++) v[m] = cmul(fm[m], v[m]);
            }
        }
        fft2048_t128(v, lds, tw128, t, [] { __syncthreads(); });
        float mag[E];
#pragma unroll
        for (int m = 0; m < E; m++) mag[m] = v[m].x * v[m].x + v[m].y * v[m].y;
#pragma unroll
        for (int m = 0; m < E; m++) mag[m] = __builtin_amdgcn_sqrtf(mag[m]);
#pragma unroll
        for (int m = 0; m < E; m++) {
            const float a = mag[m] + pa[m];
            pa[m] = mag[m];
            mag[m] = a;
        }
        if (it >= 0) {
#pragma unroll
            for (int m = 0; m < E; m++) mag[m] = __builtin_amdgcn_logf(mag[m]);
            float *yp = y + f * (long long)BINS + ((t << zp_log2) + q);
#pragma unroll
            for (int m = 0; m < E; m++) {
                const int u = m < 8 ? ((128 * m + 1024) << zp_log2) : ((128 * (m - 8)) << zp_log2);
                yp[u] = fminf(fmaxf(fmaf(6.02059991327962f, mag[m], db_off), -120.f), 0.f);
            }
        }
        if (f == sp.n_frames - 1) {
            float *pp = prev_out + (long long)s * BINS + (t << zp_log2) + q;
#pragma unroll
            for (int m = 0; m < E; m++) pp[(128 * m) << zp_log2] = pa[m];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bins == frame length (2048, no zero padding): every wave owns whole frames, so nothing is shared inside the workgroup --
// no parked slices, no interleave, no barrier in the frame loop.  A wave loads its next frame into registers while it
// transforms the current one and stores its dB bins itself (lane-contiguous).  grid (ceil(F / (4 G)), S), block 256.
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256, 2) void k_spectrum_1to1(const float2 *__restrict__ in, float *__restrict__ out,
                                                                 const float *__restrict__ window, const float2 *__restrict__ tw_nf,
                                                                 const float *__restrict__ prev_in, float *__restrict__ prev_out, SpectrumParams sp)
{
    constexpr int NF = 2048, E = NF / 64;
    __shared__ float2 lds[4][FftLds<NF>::kSlots];
    __shared__ float2 tw_lds[fft_tw_off(NF, NF)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = blockIdx.y, G = sp.frames_per_group;
    const long long f0 = ((long long)blockIdx.x * 4 + wave) * G;
    const float2 *x = in + (long long)s * sp.in_pitch;
    float *y = out + (long long)s * sp.out_pitch;
    float2 *my = lds[wave];
    for (int i = threadIdx.x; i < fft_tw_off(NF, NF); i += 256) tw_lds[i] = tw_nf[i];
    __syncthreads();
    const float db_off = 6.02059991327962f * __builtin_amdgcn_logf(0.5f * sp.scale);
    float pa[E];
    float2 nxt[E];
    {   // the first frame this wave transforms: f0 - 1 (its amplitudes seed the average), or frame 0 at the call boundary
        const long long ff = f0 > 0 ? f0 - 1 : 0;
        if (ff < sp.n_frames) {
#pragma unroll
            for (int m = 0; m < E; m++) nxt[m] = x[ff * NF + lane + 64 * m];
        }
    }
    for (int it = -1; it < G; it++) {
        const long long f = f0 + it;
        const bool live = f < sp.n_frames;
        int ln = lane;
        opaque(ln);
        if (live && f < 0) {
            const float *pp = prev_in + (long long)s * NF + ln;
#pragma unroll
            for (int m = 0; m < E; m++) pa[m] = pp[64 * m];
            continue;
        }
        if (!live) break;
        float2 v[E];
        {
            const float *wp = window + ln;
#pragma unroll
            for (int m = 0; m < E; m++) v[m] = cscale(nxt[m], wp[64 * m]);
        }
        if (it + 1 < G && f + 1 < sp.n_frames) {
            const float2 *xp = x + (f + 1) * NF + ln;
#pragma unroll
            for (int m = 0; m < E; m++) nxt[m] = xp[64 * m];
        }
        fft_regs<NF, +1, 64>(v, my, tw_lds, ln);
        float mag[E];
#pragma unroll
        for (int m = 0; m < E; m++) mag[m] = v[m].x * v[m].x + v[m].y * v[m].y;
#pragma unroll
        for (int m = 0; m < E; m++) mag[m] = __builtin_amdgcn_sqrtf(mag[m]);
        sched_fence();
#pragma unroll
        for (int m = 0; m < E; m++) {
            const float a = mag[m] + pa[m];
            pa[m] = mag[m];
            mag[m] = a;
        }
#pragma unroll
        for (int m = 0; m < E; m++) mag[m] = __builtin_amdgcn_logf(mag[m]);
        sched_fence();
        if (it >= 0) {
            float *yf = y + f * (long long)NF;
#pragma unroll
            for (int m = 0; m < E; m++)  // bin k = ln + 64 m unfolds to (k + NF/2) mod NF (fft.cpp:207-213)
                store_stream(yf + ((ln + 64 * m + NF / 2) & (NF - 1)), fminf(fmaxf(fmaf(6.02059991327962f, mag[m], db_off), -120.f), 0.f));
        }
        if (f == sp.n_frames - 1) {
            float *pp = prev_out + (long long)s * NF + ln;
#pragma unroll
            for (int m = 0; m < E; m++) pp[64 * m] = pa[m];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 65536-point spectrum (BASELINE config 5; beyond the reference's own m_maxFFTSize = 65535 clamp, fft.h:21 -- the
// oracle lifts the clamp, everything else is FFT::fftSpectrum unchanged).  Frame = 65536 samples, window 65536, no
// zero padding.  Four-step split N = 32 x 2048 with n = 2048*n1 + n2, k = k1 + 32*k2:
//   X[k1 + 32 k2] = sum_{n2} W_2048^{n2 k2} * [ W_N^{n2 k1} * sum_{n1} w[n] x[n] W_32^{n1 k1} ]
//   pass A (k_big_cols): one work-item per n2 does the 32-point DFT over n1 in registers (every load and store is
//           lane-contiguous, no LDS), applies window and the W_N^{n2 k1} twiddle, writes Y[k1][n2];
//   pass B (k_big_rows): one wave per row k1 runs the 2048-point wave transform (fft_lds.h), amplitude, previous-frame
//           average (kept in registers over the frame loop), dB; four waves (k1 = 4a..4a+3) interleave their bins.
// Bound: HBM; algorithmic bytes per frame 8*N + 4*N, plus the 2 x 8*N intermediate (written once, read once).
// ------------------------------------------------------------------------------------------------
constexpr int kBigN = 65536, kBigR = 32, kBigM = 2048;

// grid (frames * 8, S), block 256.  Y layout: [stream][frame][k1][n2]
static __global__ __launch_bounds__(256) void k_big_cols(const float2 *__restrict__ in, long long in_pitch, float2 *__restrict__ Y,
                                                         const float *__restrict__ window,
                                                         long long n_frames)
{
    const int s = blockIdx.y;
    const long long f = blockIdx.x >> 3;
    const int n2 = ((blockIdx.x & 7) << 8) + threadIdx.x;
    const float2 *x = in + (long long)s * in_pitch + f * kBigN + n2;
    float2 u[kBigR];
#pragma unroll
    for (int n1 = 0; n1 < kBigR; n1++) u[n1] = cscale(x[(long long)kBigM * n1], window[kBigM * n1 + n2]);
    dft32(u);
    float2 *y = Y + ((long long)s * n_frames + f) * kBigN + n2;
    // W_N^{n2 k1}, k1 < 32, from ONE exact phasor w = W_N^{n2}: k1 = 4 hi + lo, tw = w^(4 hi) * w^lo from two small tables built with
    // ten products (at most four deep), then one product per output -- instead of a fp64 range reduction and a sincos per output
    float2 lo[4], hi[8];
    lo[0] = make_float2(1.f, 0.f);
    lo[1] = cis_cycles(-(double)n2 / (double)kBigN);
    lo[2] = cmul(lo[1], lo[1]);
    lo[3] = cmul(lo[2], lo[1]);
    hi[0] = lo[0];
    hi[1] = cmul(lo[2], lo[2]);
    hi[2] = cmul(hi[1], hi[1]);
    hi[3] = cmul(hi[2], hi[1]);
    hi[4] = cmul(hi[2], hi[2]);
    hi[5] = cmul(hi[4], hi[1]);
    hi[6] = cmul(hi[4], hi[2]);
    hi[7] = cmul(hi[4], hi[3]);
#pragma unroll
    for (int k1 = 0; k1 < kBigR; k1++) {
        const float2 v = u[perm32(k1)];
        const float2 tw = (k1 & 3) == 0 ? hi[k1 >> 2] : (k1 >> 2) == 0 ? lo[k1 & 3] : cmul(hi[k1 >> 2], lo[k1 & 3]);
        y[(long long)kBigM * k1] = k1 == 0 ? v : cmul(tw, v);
    }
}

// grid (ceil(n_frames / G) * 8, S), block 256 = 4 waves = rows k1 = 4a .. 4a+3 of one frame at a time
static __global__ __launch_bounds__(256, 2) void k_big_rows(const float2 *__restrict__ Y, float *__restrict__ out,
                                                             const float2 *__restrict__ tw_nf, const float *__restrict__ prev_in,
                                                             float *__restrict__ prev_out, SpectrumParams sp)
{
    constexpr int E = 32;
    __shared__ float2 lds[4][FftLds<kBigM>::kSlots];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = blockIdx.y, a = blockIdx.x & 7;
    const int k1 = 4 * a + wave;
    const int G = sp.frames_per_group;
    const long long f0 = (long long)(blockIdx.x >> 3) * G;
    float2 *my = lds[wave];
    float *stage = reinterpret_cast<float *>(my);
    float pa[E];
    const float db_off = 6.02059991327962f * __builtin_amdgcn_logf(0.5f * sp.scale);
    for (int it = -1; it < G; it++) {
        const long long f = f0 + it;
        const bool live = f < sp.n_frames;
        int ln = lane;
        opaque(ln);
        if (live && f < 0) {
            const float *pp = prev_in + (long long)s * kBigN + k1 + 32 * ln;
#pragma unroll
            for (int m = 0; m < E; m++) pa[m] = pp[32 * 64 * m];
        } else if (live) {
            const float2 *row = Y + (((long long)s * sp.n_frames + f) * kBigR + k1) * kBigM + ln;
            float2 v[E];
#pragma unroll
            for (int m = 0; m < E; m++) v[m] = row[64 * m];
            fft_regs<kBigM, +1, 64>(v, my, tw_nf, ln);
            float *st = stage + ln;
            float mag[E];
#pragma unroll
            for (int m = 0; m < E; m++) mag[m] = v[m].x * v[m].x + v[m].y * v[m].y;
#pragma unroll
            for (int m = 0; m < E; m++) mag[m] = __builtin_amdgcn_sqrtf(mag[m]);
            sched_fence();
#pragma unroll
            for (int m = 0; m < E; m++) {
                const float a = mag[m] + pa[m];
                pa[m] = mag[m];
                mag[m] = a;
            }
#pragma unroll
            for (int m = 0; m < E; m++) mag[m] = __builtin_amdgcn_logf(mag[m]);
            sched_fence();
#pragma unroll
            for (int m = 0; m < E; m++) st[64 * m] = fminf(fmaxf(fmaf(6.02059991327962f, mag[m], db_off), -120.f), 0.f);
            if (f == sp.n_frames - 1) {
                float *pp = prev_out + (long long)s * kBigN + k1 + 32 * ln;
#pragma unroll
                for (int m = 0; m < E; m++) pp[32 * 64 * m] = pa[m];
            }
        }
        __syncthreads();
        if (it >= 0 && live) {
            // bins k = 4a + w + 32*j, w = 0..3: each lane stores 4 adjacent bins of one j
            float *yf = out + (long long)s * sp.out_pitch + f * (long long)kBigN;
#pragma unroll
            for (int i = 0; i < kBigM / 256; i++) {
                const int j = tid + 256 * i;
                float4 d;
                d.x = reinterpret_cast<const float *>(lds[0])[j];
                d.y = reinterpret_cast<const float *>(lds[1])[j];
                d.z = reinterpret_cast<const float *>(lds[2])[j];
                d.w = reinterpret_cast<const float *>(lds[3])[j];
                const int u = (4 * a + 32 * j + kBigN / 2) & (kBigN - 1);  // unfold, fft.cpp:207-213
                *reinterpret_cast<float4 *>(yf + u) = d;  // (16-byte pieces 128 B apart: not nontemporal -- partial lines, 0.52 -> 0.68 ms per call)
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The same spectrum as 256 x 256 (n = 256 n1 + n2, k = k1 + 256 k2): every global access of both passes is a whole 128- or
// 256-byte segment.  In the 32 x 2048 split above pass B's bins k1 + 32 k2 of one row lie 128 bytes apart: four rows per
// workgroup made 16-byte pieces, eight workgroups wrote into every line, and the pass ran 229 us for 402 MB (pass A: 95 us for
// 536 MB).
//   X[k1 + 256 k2] = sum_{n2} W_256^{n2 k2} * [ W_N^{n2 k1} * sum_{n1} w[n] x[n] W_256^{n1 k1} ]
//   pass A (k_big256_cols): a workgroup takes 32 adjacent columns n2 (256-byte row segments); work-item (column c, j) runs the
//           256-point transform of its column as 16 x 16: two DFT16 over a (n1 = 16 a + b, b = j, j + 8), twiddle W_256^{b cc},
//           exchange through LDS [cc][b][column], two DFT16 over b (cc = j, j + 8), the four-step twiddle W_N^{n2 k1} as a power
//           series from three exact phasors, Y[k1][n2] stored as 256-byte segments;
//   pass B (k_big256_rows): a workgroup takes 32 adjacent rows k1, sixteen at a time, sixteen work-items per row (128-byte
//           loads): DFT16 over a (n2 = 16 a + b), twiddle (per work-item constants), exchange inside the row's sixteen lanes
//           (wave-private LDS, padded), DFT16 over b, amplitude / previous-frame average (registers, along a chain of frames) /
//           dB, parked as floats [k2][k1] in LDS; then every bin line k1 .. k1 + 31 of one k2 leaves as one 128-byte segment
//           (unfolded: k2 ^ 128), nontemporal.
// ------------------------------------------------------------------------------------------------
constexpr int kBig256 = 256;

// grid (frames * 8, S), block 256.  Y layout: [stream][frame][k1][n2]
static __global__ __launch_bounds__(256, 2) void k_big256_cols(const float2 *__restrict__ in, long long in_pitch, float2 *__restrict__ Y,
                                                               const float *__restrict__ window, long long n_frames)
{
    __shared__ float2 T[256 * 32];   // [cc * 16 + b][column]
    __shared__ float2 tw[256];       // W_256^m
    const int t = threadIdx.x, c = t & 31, j = t >> 5;
    const int s = blockIdx.y;
    const long long f = blockIdx.x >> 3;
    const int n2 = ((blockIdx.x & 7) << 5) + c;
    tw[t] = cis_cycles(-(double)t / 256.0);
    const float2 *x = in + (long long)s * in_pitch + f * kBigN + n2;
    const float *w = window + n2;
    float2 u[2][16];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int a = 0; a < 16; a++) {
            const int n1 = 16 * a + j + 8 * h;
            u[h][a] = cscale(x[(long long)kBig256 * n1], w[kBig256 * n1]);
        }
    __syncthreads();  // (the twiddle table)
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int b = j + 8 * h;
        dft16(u[h]);
#pragma unroll
        for (int cc = 0; cc < 16; cc++) {
            const float2 v = u[h][perm16(cc)];
            T[(cc * 16 + b) * 32 + c] = cc == 0 ? v : cmul(tw[(b * cc) & 255], v);
        }
    }
    __syncthreads();
    // W_N^{n2 k1}, k1 = cc + 16 d: exact phasors for cc = j, j + 8 and for 16, then a power series over d
    const float2 w16 = cis_cycles(-(double)(16 * n2) / (double)kBigN);
    float2 *y = Y + ((long long)s * n_frames + f) * kBigN + n2;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int cc = j + 8 * h;
        float2 v[16];
#pragma unroll
        for (int b = 0; b < 16; b++) v[b] = T[(cc * 16 + b) * 32 + c];
        dft16(v);
        float2 ph = cis_cycles(-(double)(cc * n2) / (double)kBigN);
#pragma unroll
        for (int d = 0; d < 16; d++) {
            y[(long long)kBig256 * (cc + 16 * d)] = cmul(ph, v[perm16(d)]);
            ph = cmul(w16, ph);
        }
    }
}

// grid (ceil(n_frames / G) * 8, S), block 256: rows k1 = 32 (blockIdx.x & 7) + 16 pass + (t >> 4), sixteen work-items per row
static __global__ __launch_bounds__(256, 2) void k_big256_rows(const float2 *__restrict__ Y, float *__restrict__ out, const float *__restrict__ prev_in,
                                                               float *__restrict__ prev_out, SpectrumParams sp)
{
    __shared__ float2 T[16 * 16 * 17];  // [row][cc][b], a pad slot per run of sixteen
    __shared__ float db[256 * 33];      // [k2][k1 - first], a pad slot per line
    const int t = threadIdx.x, bb = t & 15, rl = t >> 4;
    const int s = blockIdx.y, rt = blockIdx.x & 7;
    const int G = sp.frames_per_group;
    const long long f0 = (long long)(blockIdx.x >> 3) * G;
    const float db_off = 6.02059991327962f * __builtin_amdgcn_logf(0.5f * sp.scale);
    // W_256^{bb cc}, cc = 1 .. 15
    float2 twr[16];
    twr[0] = make_float2(1.f, 0.f);
    twr[1] = cis_cycles(-(double)bb / 256.0);
#pragma unroll
    for (int cc = 2; cc < 16; cc++) twr[cc] = cis_cycles(-(double)(bb * cc) / 256.0);
    float pa[2][16];
    float2 *Tr = T + rl * 16 * 17;
    for (int it = -1; it < G; it++) {
        const long long f = f0 + it;
        const bool live = f < sp.n_frames;
        if (live && f < 0) {
            const float *pp = prev_in + (long long)s * kBigN + (long long)rt * 8192 + t;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int d = 0; d < 16; d++) pa[h][d] = pp[(h * 16 + d) * 256];
        } else if (live) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k1 = 32 * rt + 16 * h + rl;
                const float2 *row = Y + (((long long)s * sp.n_frames + f) * kBig256 + k1) * kBig256 + bb;
                float2 u[16];
#pragma unroll
                for (int a = 0; a < 16; a++) u[a] = row[16 * a];
                dft16(u);
                wave_sync();  // (the row's sixteen lanes have read what the previous pass parked)
#pragma unroll
                for (int cc = 0; cc < 16; cc++) Tr[cc * 17 + bb] = cc == 0 ? u[perm16(0)] : cmul(twr[cc], u[perm16(cc)]);
                wave_sync();
                float2 v[16];
#pragma unroll
                for (int b = 0; b < 16; b++) v[b] = Tr[bb * 17 + b];  // (this work-item's cc is its lane index in the row)
                dft16(v);
                float mag[16];
#pragma unroll
                for (int d = 0; d < 16; d++) {
                    const float2 z = v[perm16(d)];
                    mag[d] = __builtin_amdgcn_sqrtf(z.x * z.x + z.y * z.y);
                }
#pragma unroll
                for (int d = 0; d < 16; d++) {
                    const float a = mag[d] + pa[h][d];
                    pa[h][d] = mag[d];
                    mag[d] = __builtin_amdgcn_logf(a);
                }
                if (it >= 0) {
#pragma unroll
                    for (int d = 0; d < 16; d++)  // k2 = bb + 16 d
                        db[(bb + 16 * d) * 33 + 16 * h + rl] = fminf(fmaxf(fmaf(6.02059991327962f, mag[d], db_off), -120.f), 0.f);
                }
            }
            if (f == sp.n_frames - 1) {
                float *pp = prev_out + (long long)s * kBigN + (long long)rt * 8192 + t;
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int d = 0; d < 16; d++) pp[(h * 16 + d) * 256] = pa[h][d];
            }
        }
        __syncthreads();
        if (it >= 0 && live) {
            // lines of 32 bins: k = 32 rt + lane + 256 k2, unfolded to k2 ^ 128 (fft.cpp:207-213)
            float *yf = out + (long long)s * sp.out_pitch + f * (long long)kBigN + 32 * rt + (t & 31);
#pragma unroll 8
            for (int i = 0; i < 32; i++) {
                const int k2 = 8 * i + (t >> 5);
                store_stream(yf + 256 * (k2 ^ 128), db[k2 * 33 + (t & 31)]);
            }
        }
        __syncthreads();
    }
}

// (Pass B was also built on the two-wave transform of fft_t128.h -- 512-item workgroups, four rows x two waves, four waves per
// SIMD -- and measured slower, 0.39 ms against 0.32 for the bench shard: its rows wait on workgroup-wide barriers and on their
// own loads, where here every wave runs alone.)
// ------------------------------------------------------------------------------------------------
// Any other frame length: FFT::fftSpectrum for framesPerBuffer != 2048 (settings.cpp:57 makes it a setting) and for the branch of
// m_applyWindow that takes fewer samples than samplesPerBuffer -- copied, zero-padded, NOT windowed (fft.cpp:129-157).  Functional
// path, not the bench's: one (frame chain, q) per 256-item workgroup, the pruned zero-padded transform as bins / M transforms of M =
// the frame length rounded up to a power of two (<= 16384: 128 KiB of LDS), each a plain radix-2 decimation-in-time transform in LDS
// (bit-reversed load, log2 M passes, twiddles W_M^k from a table), previous-frame amplitudes carried in registers along the chain.
//   X[ZP k + q] = sum_{n < n_in} (w[n] x[n] W_bins^{n q}) W_M^{n k}
// grid (chains * ZP, S), block 256, dynamic LDS M * 8 bytes.
// ------------------------------------------------------------------------------------------------
struct AnySpecParams {
    int n_in;        // samples taken from every frame (== frame length, or fewer: the un-windowed branch)
    int frame;       // samples from one frame to the next in the input
    int M, logM;     // transform length and its log2
    int zp_log2;     // bins = M << zp_log2
    int windowed;    // multiply by window[n] (numSamples == samplesPerBuffer)
};
static __global__ __launch_bounds__(256) void k_spectrum_any(const float2 *__restrict__ in, float *__restrict__ out, const float *__restrict__ window,
                                                            const float2 *__restrict__ twM, const float *__restrict__ prev_in, float *__restrict__ prev_out,
                                                            SpectrumParams sp, AnySpecParams ap)
{
    HIP_DYNAMIC_SHARED(float2, buf)
    constexpr int EMAX = 64;  // M / 256 <= 64 points per work-item
    const int tid = threadIdx.x, s = blockIdx.y;
    const int ZP = 1 << ap.zp_log2, q = blockIdx.x & (ZP - 1);
    const int M = ap.M, logM = ap.logM, bins = M << ap.zp_log2;
    const int E = M >= 256 ? M >> 8 : 1;
    const int G = sp.frames_per_group;
    const long long f0 = (long long)(blockIdx.x >> ap.zp_log2) * G;
    const float db_off = 6.02059991327962f * __builtin_amdgcn_logf(0.5f * sp.scale);
    float pa[EMAX];
#pragma unroll
    for (int m = 0; m < EMAX; m++) pa[m] = 0.f;
    for (int it = -1; it < G; it++) {
        const long long f = f0 + it;
        if (f >= sp.n_frames) break;  // (uniform)
        if (f < 0) {
#pragma unroll
            for (int m = 0; m < EMAX; m++)
                if (m < E && tid + 256 * m < M) pa[m] = prev_in[(long long)s * bins + ((tid + 256 * m) << ap.zp_log2) + q];
            continue;
        }
        const float2 *x = in + (long long)s * sp.in_pitch + f * ap.frame;
        for (int i = tid; i < M; i += 256) {
            float2 v = make_float2(0.f, 0.f);
            if (i < ap.n_in) {
                v = x[i];
                if (ap.windowed) v = cscale(v, window[i]);
                if (q) v = cmul(cis_cycles(-(double)(((long long)i * q) & (bins - 1)) / (double)bins), v);
            }
            buf[__brev((unsigned)i) >> (32 - logM)] = v;
        }
        __syncthreads();
        for (int st = 1; st <= logM; st++) {
            const int half = 1 << (st - 1);
            for (int b = tid; b < (M >> 1); b += 256) {
                const int j = b & (half - 1), i0 = ((b >> (st - 1)) << st) + j, i1 = i0 + half;
                const float2 t = cmul(twM[j << (logM - st)], buf[i1]), a = buf[i0];
                buf[i0] = cadd(a, t);
                buf[i1] = csub(a, t);
            }
            __syncthreads();
        }
        float *yf = out + (long long)s * sp.out_pitch + f * (long long)bins;
#pragma unroll
        for (int m = 0; m < EMAX; m++) {
            const int k = tid + 256 * m;
            if (m < E && k < M) {
                const float2 z = buf[k];
                const float mag = __builtin_amdgcn_sqrtf(z.x * z.x + z.y * z.y);
                const float a = mag + pa[m];
                pa[m] = mag;
                if (it >= 0) {
                    const int u = (((k << ap.zp_log2) + q) + (bins >> 1)) & (bins - 1);  // unfold, fft.cpp:207-213
                    yf[u] = fminf(fmaxf(fmaf(6.02059991327962f, __builtin_amdgcn_logf(a), db_off), -120.f), 0.f);
                }
            }
        }
        if (f == sp.n_frames - 1) {
#pragma unroll
            for (int m = 0; m < EMAX; m++)
                if (m < E && tid + 256 * m < M) prev_out[(long long)s * bins + ((tid + 256 * m) << ap.zp_log2) + q] = pa[m];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// SignalStrength::fdEstimate (application/signalstrength.cpp:287-380) on every frame of the unprocessed spectrum: peak and
// average power inside the band-pass window around the mixer frequency, average power of one window width either side
// (noise), all from the dB bins (power = 10^(dB/10)).  One wave per (frame, channel); out[channel][frame] =
// (peakDb, avgDb, snrDb, floorDb).  The bin indices come from the host (integer bin width etc. as the reference has it).
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(64) void k_signal_strength(const float *__restrict__ spec, long long stream_pitch, int bins, long long n_frames,
                                                               const SmBins *__restrict__ sb, float4 *__restrict__ out, long long out_pitch)
{
    const int c = blockIdx.y, lane = threadIdx.x;
    const long long f = blockIdx.x;
    const SmBins b = sb[c];
    const float *sp = spec + (long long)b.stream * stream_pitch + f * bins;
    double peak = 0.0, total = 0.0, noise = 0.0;
    int nn = 0;
    const int last = b.nhi < bins - 1 ? b.nhi : bins - 1;  // the reference's loop ends at i == noiseHighBin or the last bin
    for (int i = b.nlo + lane; i <= last; i += 64) {
        const double pwr = exp2((double)sp[i] * 0.33219280948873623479);  // 10^(dB/10)
        if (i >= b.lo && i <= b.hi) {
            total += pwr;
            peak = pwr > peak ? pwr : peak;
        } else {
            noise += pwr;
            nn++;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double p2 = __shfl_xor(peak, d), t2 = __shfl_xor(total, d), n2 = __shfl_xor(noise, d);
        const int c2 = __shfl_xor(nn, d);
        peak = p2 > peak ? p2 : peak;
        total += t2;
        noise += n2;
        nn += c2;
    }
    if (lane != 0) return;
    auto p2db = [](double p) { return p == 0.0 ? -120.0 : 10.0 * log10(p); };
    auto clip = [](double d) { return d < -120.0 ? -120.0 : (d > 0.0 ? 0.0 : d); };
    const double avg = total / (double)b.bp_bins, navg = noise / (double)nn;
    double snr = (navg == 0.0 || peak == 0.0) ? -120.0 : 10.0 * log10(peak / navg);
    snr = snr < 0.0 ? 0.0 : (snr > 120.0 ? 120.0 : snr);
    out[(long long)c * out_pitch + f] = make_float4((float)clip(p2db(peak)), (float)clip(p2db(avg)), (float)snr, (float)clip(p2db(navg)));
}

}  // namespace pg
