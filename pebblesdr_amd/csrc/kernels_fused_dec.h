// kernels_fused_dec.h -- mixer + the WHOLE decimator in one kernel for a bank of channels tuned off one shared stream:
// Mixer::processBlock + Decimator::process (pebblelib/mixer.cpp:48-81, decimator.cpp:152-226, 593-659) for chains of the
// form  hb11 x S (one 11-tap halfband evaluated every S-th sample), then three halfbands at stride 2  -- e.g. the
// reference's 2.048 Msps / 30 kHz chain hb11x4, hb15, hb19, hb31 (BASELINE configs[2]).
//
// Why: the two-kernel route (k_mix_hb11_bank -> stage-0 buffer -> k_cascade) writes and re-reads every channel's stream at
// Fs/S: 268 MB each way for 256 channels x 0.5 M samples, 15x the bytes the call has to move (4 MB in, 34 MB out), and both
// kernels sat on that buffer.  Here nothing above the demodulator rate is written.
//
// Mapping: the lanes of a wave are 64 channels of the bank (as in k_mix_hb11_bank<UNIFORM>): the input window is the same
// address in every lane, so it is fetched with scalar loads, and a lane walks ITS channel through time: per block of 8
// first-stage outputs it computes them in the oscillator-factored form of k_mix_hb11_lean
//     y0[j] = pa(j) * sum_d (h[d] step[d]) x[S j - 10 + d],   pa(j) = a_inf e^{j 2 pi (phase0 + (S j - 9) inc)}
// and pushes each one, as it appears, through the three halfbands held ENTIRELY IN REGISTERS in transposed form: a stage
// keeps the partial sums of its PENDING outputs instead of its past inputs,
//     y[m] = sum_p h[p] x[2m - (T-1) + p]:  an even-indexed input adds h[p] x to the (T+1)/2 outputs it reaches through the
//     even taps (the first of them starts there, the last is completed by it and moves on to the next stage); an
//     odd-indexed input meets only the centre tap (T = 4k + 3)
// -- 11 + 11 + 16 running sums for 15/19/31 taps (76 registers; the stored-input form needs 146), every index static, and
// 31 register moves per block to re-base them.  No barrier and no memory traffic inside the loop except the block's input
// window (below); a wave's 16 x 64 results leave through a wave-private LDS tile as whole 128-byte row segments
// (front_store_rows' layout).
//
// Time is cut into chunks of L final outputs, one wave each.  A chunk cannot inherit the registers of the wave before it,
// so it first runs `warm` = halo / 8 blocks whose results it throws away (halo = (T1-1) + 2 (T2-1) + 4 (T3-1) first-stage
// outputs is the look-back of the cascade: 170 -> 21 blocks for 15/19/31); after them every carried value is exact.  Chunk 0
// of a call takes its warm-up first-stage outputs from the previous call's tail (the stage-0 buffer's head-room, which
// this kernel also refreshes -- through a small staging buffer, because chunk 0 of the same launch still reads the old
// one -- together with the mixed-sample history of the two-kernel route, so a call can go either way: the host sends
// calls inside an oscillator's amplitude transient down the two-kernel route).
//
// Bound: fp32 VALU (~27 packed operations per first-stage output per lane); HBM traffic is the compulsory 8 B per input
// sample per stream + 8 B per final output per channel.
#pragma once
#include "kernels_frontend.h"
#include "params.h"

namespace pg {

// The halfband coefficient tables (generated data, hb_taps.inc) as compile-time constants: with a constant design and tap index
// the load folds to a literal, so the ~45 coefficients of a chain need no registers (held in scalar registers they spilled).
namespace hbc {
#include "hb_taps.inc"
}
template <int T> __device__ __forceinline__ float hb_tap(int p) { return (float)hbc::pebble_hb_designs[(T - 7) / 4].h[p]; }

struct FusedDecParams {
    long long n_out;          // final outputs per channel in this call
    long long out_pitch;      // floats2 per row of `out`
    long long y0_pitch;       // row pitch of the stage-0 history (float2)
    int S;                    // first-stage stride
    int L;                    // final outputs per chunk (multiple of 16)
    int n_chan;
    int hist_pitch;           // mixed-history row pitch (kMaxTaps)
    float a_inf, gain0, gain; // oscillator amplitude, first-stage gain (1), gain on the final output
};

// HY: depth of the first-stage history a call leaves (and chunk 0 reads) = 8 * warm + 8
template <int T1, int T2, int T3>
struct FusedDecGeom {
    static constexpr int halo = (T1 - 1) + 2 * (T2 - 1) + 4 * (T3 - 1);
    static constexpr int warm = halo / 8;
    static constexpr int HY = 8 * warm + 8;
    // running sums per stage: pending outputs at a block's start plus the ones its inputs start
    static constexpr int N1 = (T1 + 1) / 2 + 3, N2 = (T2 + 1) / 2 + 1, N3 = (T3 + 1) / 2;
    static_assert(T1 % 4 == 3 && T2 % 4 == 3 && T3 % 4 == 3, "halfbands have 4k + 3 taps (odd centre)");
};

// x_hist: [>= 16] raw input samples preceding the call (x[-16 .. -1]); xh_out receives the call's last 16.
// y0_hist: data pointer of the stage-0 history rows: y0_hist[c * y0_pitch - HY .. -1] = previous call's last HY first-stage outputs.
// y0_stage: [C][HY] receives this call's last HY first-stage outputs (the host's tail refresh copies them into the head-room).
// grid (ceil(chunks / 4), ceil(C / 64)), block 256 = four independent waves (four consecutive chunks).
template <int T1, int T2, int T3>
static __global__ __launch_bounds__(256, 2) void k_mix_dec_fused(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                                 const ChanOsc *__restrict__ osc, OscDynInline dyn,
                                                                 const float2 *__restrict__ x_hist, float2 *__restrict__ xh_out,
                                                                 const float2 *__restrict__ y0_hist, float2 *__restrict__ y0_stage,
                                                                 float2 *__restrict__ mixed_hist_out, FusedDecParams P)
{
    using G = FusedDecGeom<T1, T2, T3>;
    constexpr int N1 = G::N1, N2 = G::N2, N3 = G::N3, HY = G::HY;
    constexpr int PC1 = (T1 - 1) / 2, PC2 = (T2 - 1) / 2, PC3 = (T3 - 1) / 2;  // centre taps
    constexpr int XB = 128;  // samples per window buffer: a block's raw span is 7 S + 11 <= 123 for S <= 16
    __shared__ float2 tiles[4][16 * 65];
    __shared__ float2 xwin[4][2][XB];
    __shared__ float2 htile[4][8 * 65];  // a block's eight first-stage outputs x 64 channels on their way to the history rows
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *tile = tiles[wv];
    const int cbase = blockIdx.y * 64;
    const bool live = cbase + lane < P.n_chan;
    const int c = live ? cbase + lane : P.n_chan - 1;
    const int S = P.S;
    const long long chunk = (long long)blockIdx.x * 4 + wv;
    const long long o0 = chunk * P.L;
    if (o0 >= P.n_out) return;  // wave-uniform
    const long long o1 = o0 + P.L < P.n_out ? o0 + P.L : P.n_out;
    const bool last_chunk = o1 == P.n_out;
    const long long len0 = 8 * P.n_out;      // first-stage outputs of the call
    const long long n_in = (long long)S * len0;

    // ---- per-channel constants ----
    const ChanOsc *oc = &osc[c];
    const double inc = oc->inc;
    double phase0 = oc->phase0;
    uint32_t mix_on = oc->mix_on;
    if (dyn.use) {
#pragma unroll
        for (int k = 0; k < kOscInline; k++)
            if (c == k) { phase0 = dyn.d[k].phase0; mix_on = dyn.d[k].mix_on; }
    }
    const bool mix = mix_on != 0;
    // c_d = h[d] * step[d] for the window's used samples d = 0 2 4 5 6 8 10 (c_0 is real: step[0] = 1)
    float2 c2 = make_float2(hb_tap<11>(2), 0.f), c4 = make_float2(hb_tap<11>(4), 0.f), c5 = make_float2(hb_tap<11>(5), 0.f), c6 = make_float2(hb_tap<11>(6), 0.f),
           c8 = make_float2(hb_tap<11>(8), 0.f), c10 = make_float2(hb_tap<11>(10), 0.f);
    float2 rot = make_float2(1.f, 0.f), rot8 = make_float2(1.f, 0.f);
    if (mix) {
        c2 = cscale(oc->step[2], hb_tap<11>(2)); c4 = cscale(oc->step[4], hb_tap<11>(4)); c5 = cscale(oc->step[5], hb_tap<11>(5));
        c6 = cscale(oc->step[6], hb_tap<11>(6)); c8 = cscale(oc->step[8], hb_tap<11>(8)); c10 = cscale(oc->step[10], hb_tap<11>(10));
        rot = cis_cycles((double)S * inc);        // one first-stage output to the next
        rot8 = cis_cycles((double)(8 * S) * inc); // one block to the next
    }
    const float h00 = hb_tap<11>(0);
    const float amp = mix ? P.a_inf * P.gain0 : P.gain0;  // f == 0: the mixer returns its input untouched (mixer.cpp:51-53)

    // running sums: a1[i] <-> y1[4 o - 3 + i], a2[i] <-> y2[2 o - 1 + i], a3[i] <-> y3[o + i] at the start of block o
    float2 a1[N1], a2[N2], a3[N3];
#pragma unroll
    for (int i = 0; i < N1; i++) a1[i] = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < N2; i++) a2[i] = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < N3; i++) a3[i] = make_float2(0.f, 0.f);

    const float2 *yh = y0_hist + (long long)c * P.y0_pitch;  // yh[j], j in [-HY, -1]
    float2 *ys = y0_stage + (long long)c * HY;
    const long long o_start = o0 - G::warm;
    // the last chunk runs one block more: first-stage outputs 8 n_out - 7 .. 8 n_out - 1 belong to no output of this call but
    // to the next call's history
    const long long o_end = last_chunk ? o1 + 1 : o1;
    float2 pa_blk = make_float2(amp, 0.f);

    // The raw samples a block needs, x[S (8 o - 7) - 10 ...] (7 S + 11 of them), are the same for every lane: the wave fetches
    // them one block ahead with ONE coalesced vector load per 64 samples (lane = sample; samples before the call's start come
    // from the previous call's tail), parks them in its own LDS window and every lane then reads the taps at the same
    // addresses (broadcast reads).  Scalar loads did this without LDS, but a wave then waited out a scalar-cache miss per
    // output: 5000 clocks per block against ~1400 of arithmetic.
    const int nl = 7 * S + 11 > 64 ? 2 : 1;
    auto fetch = [&](long long o, float2 (&r)[2]) {
        const long long b = (long long)S * (8 * o - 7) - 10;
        if (b >= 0 && b + 128 <= n_in) {  // wave-uniform: the whole span lies inside the call (all but the edge blocks)
            r[0] = in[b + lane];
            if (nl > 1) r[1] = in[b + lane + 64];
            return;
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (q < nl) {
                long long i = b + lane + 64 * q;
                i = i < n_in ? i : n_in - 1;
                r[q] = i < 0 ? (i >= -16 ? x_hist[16 + i] : make_float2(0.f, 0.f)) : in[i];
            }
        }
    };
    // window pipeline: block o's samples are fetched during block o - 2 (registers), parked in LDS in the middle of block
    // o - 1 (the buffer block o - 2 read) and read back tap by tap during block o: a fetch has a whole block to land
    const long long ob0 = o_start >= 0 ? o_start : 0;  // blocks before the call's start take their first-stage outputs from the history
    float2 xn[2];
    auto park = [&](long long o) {
        xwin[wv][o & 1][lane] = xn[0];
        if (nl > 1) xwin[wv][o & 1][lane + 64] = xn[1];
    };
    fetch(ob0, xn);
    park(ob0);
    fetch(ob0 + 1, xn);
    wave_sync();

    float2 y3 = make_float2(0.f, 0.f);
    // first-stage output number k of a block (j = 8 o - 7 + k) enters the cascade; k is a constant after unrolling
    auto feed = [&](int k, float2 y0) {
        auto bc = [](float h) { return make_float2(h, h); };
        if (k & 1) {  // j even: the even taps of stage 1; y1[4 o - 3 + b] is complete
            const int b = (k - 1) / 2;
#pragma unroll
            for (int t = 0; t < (T1 + 1) / 2; t++) {
                const float2 v = y0 * bc(hb_tap<T1>(T1 - 1 - 2 * t));
                a1[b + t] = t == (T1 + 1) / 2 - 1 ? v : a1[b + t] + v;
            }
            const float2 y1 = a1[b];
            if (b & 1) {  // m even
                const int b2 = (b - 1) / 2;
#pragma unroll
                for (int t = 0; t < (T2 + 1) / 2; t++) {
                    const float2 v = y1 * bc(hb_tap<T2>(T2 - 1 - 2 * t));
                    a2[b2 + t] = t == (T2 + 1) / 2 - 1 ? v : a2[b2 + t] + v;
                }
                const float2 y2 = a2[b2];
                if (b2 & 1) {  // q even: y3[o] is complete
#pragma unroll
                    for (int t = 0; t < (T3 + 1) / 2; t++) {
                        const float2 v = y2 * bc(hb_tap<T3>(T3 - 1 - 2 * t));
                        a3[t] = t == (T3 + 1) / 2 - 1 ? v : a3[t] + v;
                    }
                    y3 = a3[0];
                } else {
                    a3[(PC3 - 1) / 2] = a3[(PC3 - 1) / 2] + y2 * bc(hb_tap<T3>(PC3));
                }
            } else {
                a2[(b + PC2 - 1) / 2] = a2[(b + PC2 - 1) / 2] + y1 * bc(hb_tap<T2>(PC2));
            }
        } else {
            a1[(k + PC1 - 1) / 2] = a1[(k + PC1 - 1) / 2] + y0 * bc(hb_tap<T1>(PC1));
        }
    };

    struct Win { float2 x0, x2, x6, x8, x10; float4 x45; };
    // window of a block's output k: xw[S k + d], d = 0 2 4 5 6 8 10
    auto read_win = [&](const float2 *xw, int k) {
        const float2 *p = xw + S * k;
        Win w;
        w.x0 = p[0]; w.x2 = p[2]; w.x6 = p[6]; w.x8 = p[8]; w.x10 = p[10];
        w.x45 = *reinterpret_cast<const float4 *>(p + 4);
        return w;
    };
    auto stage0 = [&](const Win &w, float2 pa) {
        // a tree, not a chain: six dependent additions in a row leave a lone wave nothing to issue between them
        const float2 t0 = cadd(cscale(w.x0, h00), cmul(c2, w.x2));
        const float2 t1 = cadd(cmul(c4, make_float2(w.x45.x, w.x45.y)), cmul(c5, make_float2(w.x45.z, w.x45.w)));
        const float2 t2 = cadd(cmul(c6, w.x6), cmul(c8, w.x8));
        const float2 acc = cadd(cadd(t0, t1), cadd(t2, cmul(c10, w.x10)));
        return cmul(pa, acc);
    };

    for (long long o = o_start; o < o_end; o++) {  // wave-uniform
        const float2 *xw = xwin[wv][o & 1];
        // in the middle of block o (o >= ob0): park block o + 1's samples, start fetching block o + 2's
        auto advance = [&]() {
            if (o >= ob0) {
                park(o + 1);
                wave_sync();
                fetch(o + 2, xn);
            }
        };
        // ---- the block's eight first-stage outputs j = 8 o - 7 .. 8 o ----
        const long long j0 = 8 * o - 7;
        if (o < 0) {  // chunk 0's warm-up: the previous call's tail
#pragma unroll
            for (int k = 0; k < 8; k++) feed(k, yh[j0 + k]);
            advance();
        } else {
            if (((o - o_start) & 7) == 0 || o == 0) {
                // exact phase every eighth block, a constant rotation in between (and from output to output inside a block)
                if (mix) pa_blk = cscale(cis_cycles(phase0 + (double)((long long)S * j0 - 9) * inc), amp);
            } else {
                pa_blk = cmul(rot8, pa_blk);  // (1, 0) for a channel that does not mix
            }
            float2 pa = pa_blk;
            const bool keep = last_chunk && j0 + 7 >= len0 - HY;  // among the call's last HY: the next call's history
            if (o == 0) {  // block 0: only output 0 is new, the seven before it are the previous call's
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    float2 y0 = stage0(read_win(xw, k), pa);
                    pa = cmul(rot, pa);
                    if (k < 7) y0 = yh[j0 + k];
                    if (keep) htile[wv][k * 65 + lane] = y0;
                    feed(k, y0);
                    sched_fence();
                    if (k == 3) advance();
                }
            } else {
                // the window of output k + 1 is read while output k is computed
                Win w = read_win(xw, 0);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    Win wn = w;
                    if (k < 7) wn = read_win(xw, k + 1);
                    const float2 y0 = stage0(w, pa);
                    pa = cmul(rot, pa);
                    if (keep) htile[wv][k * 65 + lane] = y0;
                    feed(k, y0);
                    w = wn;
                    sched_fence();  // one window ahead, not eight: the scheduler otherwise hoists them all (spills)
                    if (k == 3) advance();
                }
            }
            if (keep) {  // eight outputs x 64 channels -> 64-byte row segments of the staging rows
                wave_sync();
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int idx = it * 64 + lane;
                    const int ch = idx >> 3, k = idx & 7;
                    const long long j = j0 + k;
                    if (cbase + ch < P.n_chan && j >= len0 - HY && j < len0)
                        y0_stage[(long long)(cbase + ch) * HY + (j - (len0 - HY))] = htile[wv][k * 65 + ch];
                }
                wave_sync();
            }
        }
        if (o >= o0 && o < o1) {
            const int jt = (int)(o - o0) & 15;
            tile[jt * 65 + lane] = cscale(y3, P.gain);
            if (jt == 15 || o == o1 - 1) {  // 16 outputs x 64 channels -> whole 128-byte row segments
                const long long ob = o - jt;
                wave_sync();
#pragma unroll
                for (int it = 0; it < 16; it++) {
                    const int idx = it * 64 + lane;
                    const int ch = idx >> 4, t = idx & 15;
                    if (cbase + ch < P.n_chan && t <= jt) out[(long long)(cbase + ch) * P.out_pitch + ob + t] = tile[t * 65 + ch];
                }
                wave_sync();
            }
        }
        // ---- re-base the running sums on the next block ----
#pragma unroll
        for (int i = 0; i + 4 < N1; i++) a1[i] = a1[i + 4];
#pragma unroll
        for (int i = 0; i + 2 < N2; i++) a2[i] = a2[i + 2];
#pragma unroll
        for (int i = 0; i + 1 < N3; i++) a3[i] = a3[i + 1];
    }
    if (last_chunk) {
        // the next call's raw-input tail and, for the two-kernel route, the mixed-sample history m[n-10 .. n-1] (each with its exact phase)
        if (blockIdx.y == 0 && lane < 16) xh_out[lane] = in[n_in - 16 + lane];
        if (live && mixed_hist_out != nullptr) {
            float2 *hp = mixed_hist_out + (long long)c * P.hist_pitch;
#pragma unroll 1
            for (int q = 0; q < 10; q++) {
                const long long i = n_in - 10 + q;
                float2 v = in[i];
                if (mix) v = cmul(cscale(cis_cycles(phase0 + (double)(i + 1) * inc), P.a_inf), v);
                hp[q] = v;
            }
        }
    }
}

}  // namespace pg
