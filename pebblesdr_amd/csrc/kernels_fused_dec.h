// kernels_fused_dec.h -- mixer + the WHOLE decimator in one kernel for a bank of channels tuned off one shared stream:
// Mixer::processBlock + Decimator::process (pebblelib/mixer.cpp:48-81, decimator.cpp:152-226, 593-659) for chains of the
// form  hb11 x S (one 11-tap halfband evaluated every S-th sample), then three halfbands at stride 2  -- e.g. the
// reference's 2.048 Msps / 30 kHz chain hb11x4, hb15, hb19, hb31 (BASELINE configs[2]).
//
// Why: the two-kernel route (k_mix_hb11_bank -> stage-0 buffer -> k_cascade) writes and re-reads every channel's stream at
// Fs/S: 268 MB each way for 256 channels x 0.5 M samples, 15x the bytes the call has to move (4 MB in, 34 MB out), and both
// kernels sat on that buffer.  Here nothing above the demodulator rate is written.
//
// Mapping: the lanes of a wave are 64 channels of the bank (as in k_mix_hb11_bank<UNIFORM>) and a lane walks ITS channel
// through time in blocks of 8 first-stage outputs, computed in the oscillator-factored form of k_mix_hb11_lean
//     y0[j] = pa(j) * sum_d (h[d] step[d]) x[S j - 10 + d],   pa(j) = a_inf e^{j 2 pi (phase0 + (S j - 9) inc)}
// and pushed, as they appear, through the three halfbands held ENTIRELY IN REGISTERS in transposed form: a stage keeps the
// partial sums of its PENDING outputs instead of its past inputs,
//     y[m] = sum_p h[p] x[2m - (T-1) + p]:  an even-indexed input adds h[p] x to the (T+1)/2 outputs it reaches through the
//     even taps (the first of them starts there, the last is completed by it and moves on to the next stage); an
//     odd-indexed input meets only the centre tap (T = 4k + 3)
// -- 11 + 11 + 16 running sums for 15/19/31 taps (76 registers; the stored-input form needs 146), every index static.
//
// A chunk of L final outputs is the work of a FOUR-WAVE workgroup, a pipeline with one workgroup barrier per block: waves 0 and 1
// produce a block's first-stage outputs 0..3 and 4..7 into an LDS ring, wave 2 runs halfband 1 one block behind them and
// hands its four outputs on through a second ring, wave 3 runs halfbands 2 and 3 two blocks behind and writes the results.
// (One wave per chunk doing all of it ran the bank as 1024 lone waves: 0.138 ms per configs[2] call, VALU active 49 % of a
// wave's life; three waves -- both halfband stages in one -- 0.094; this form 0.092.)
// The first stage uses the hb11's symmetry: its seven taps pair up around the centre,
//     h[5+e] (step[5+e] x[5+e] + step[5-e] x[5-e]) = step[5] h[5+e] (cos(e t) (x[5+e] + x[5-e]) + j sin(e t) (x[5+e] - x[5-e])),
// t = 2 pi inc, e = 1 3 5, x[d] = x[S j - 10 + d], and the sums and (rotated) differences are the same for every channel:
// wave 0 forms them once per output, two blocks ahead and straight from global memory (56 lanes = 8 outputs x 7 quantities),
// and parks them in LDS; a producer reads its 28 at the same address in every lane and spends six packed FMAs with REAL
// per-channel coefficients per output (fourteen in the tap-by-tap complex form: 0.102 ms).  The packed instructions take their
// broadcasts and half-negations as operand modifiers (common.h cmul_pk, fma_lo_pk).  The chunk's results leave through an LDS
// tile as whole 128-byte row segments (front_store_rows' layout).
//
// A chunk cannot inherit the registers of the chunk before it, so it first runs `warm` = halo / 8 blocks whose results it
// throws away (halo = (T1-1) + 2 (T2-1) + 4 (T3-1) first-stage outputs is the look-back of the cascade: 170 -> 21 blocks for
// 15/19/31); after them every carried value is exact.  Chunk 0 of a call takes its warm-up first-stage outputs from the
// previous call's tail (the stage-0 buffer's head-room, which this kernel also refreshes -- through a small staging buffer,
// because chunk 0 of the same launch still reads the old one -- together with the mixed-sample history of the two-kernel
// route, so a call can go either way: the host sends calls inside an oscillator's amplitude transient down the two-kernel
// route).
//
// Bound: VALU issue plus LDS time, which add rather than overlap here (as in k_spectrum_t128): per block of 8 x 64
// first-stage outputs ~310 wave instructions over the four waves (v_pk_fma_f32 retires one per ~5.2 clocks per SIMD whatever the
// occupancy) and ~230 clocks of the CU's LDS pipe (the producers' broadcast reads cost their full 64 x 16 bytes each, a
// 16-byte write ~13 clocks), 20-25 % of it all warm-up.  HBM traffic is the compulsory 8 B per input sample per stream + 8 B
// per final output per channel, plus the raw samples of the warm-up blocks (L2 hits: every chunk of every 64-channel group
// reads the same 4 MB).
// Measured alternative (round 2, not kept): first stage with the lanes along TIME (raw samples per lane, channel constants
// scalar) transposed to this layout through a 64 x 64 LDS tile, ten waves per workgroup -- parity-clean but 0.143 ms: its
// 78 KB of LDS admit one workgroup per CU and the halfband wave, alone on its SIMD, needs ~760 clocks per block.
#pragma once
#include "kernels_frontend.h"
#include "params.h"
#include "hb_const.h"

namespace pg {

// (hb_tap<T>(p): the halfband coefficients as literals, hb_const.h -- held in scalar registers the ~45 of a chain spilled)

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
// grid (chunks, ceil(C / 64)), block 256.
template <int T1, int T2, int T3>
static __global__ __launch_bounds__(256) void k_mix_dec_fused(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                            const ChanOsc *__restrict__ osc, OscDynInline dyn,
                                                            const float2 *__restrict__ x_hist, float2 *__restrict__ xh_out,
                                                            const float2 *__restrict__ y0_hist, float2 *__restrict__ y0_stage,
                                                            float2 *__restrict__ mixed_hist_out, FusedDecParams P)
{
    using G = FusedDecGeom<T1, T2, T3>;
    constexpr int N1 = G::N1, N2 = G::N2, N3 = G::N3, HY = G::HY;
    constexpr int PC1 = (T1 - 1) / 2, PC2 = (T2 - 1) / 2, PC3 = (T3 - 1) / 2;
    __shared__ float2 tile[16 * 65];
    __shared__ __attribute__((aligned(16))) float2 srw[2][64];  // [block parity][7 k + q]: the shared sums of a block's eight outputs
    __shared__ float4 ring[2][4 * 64];   // [block parity][output pair][channel lane]
    __shared__ float4 ring1[2][2 * 64];  // the same for halfband 1's four outputs of a block
    __shared__ float2 htile[8 * 65];
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0, 1: first-stage outputs 0..3 / 4..7; 2: halfband 1; 3: halfbands 2 and 3
    const int cbase = blockIdx.y * 64;
    const bool live = cbase + lane < P.n_chan;
    const int c = live ? cbase + lane : P.n_chan - 1;
    const int S = P.S;
    const long long o0 = (long long)blockIdx.x * P.L;
    if (o0 >= P.n_out) return;  // workgroup-uniform
    const long long o1 = o0 + P.L < P.n_out ? o0 + P.L : P.n_out;
    const bool last_chunk = o1 == P.n_out;
    const long long len0 = 8 * P.n_out;
    const long long n_in = (long long)S * len0;
    const long long o_start = o0 - G::warm;
    const long long o_end = last_chunk ? o1 + 1 : o1;  // the last chunk runs one more block (o == n_out) whose first-stage outputs only feed the history
    const long long ob0 = o_start >= 0 ? o_start : 0;
    const int nb = (int)(o_end - o_start);
    const int n_iter = nb + 2;            // halfband 1 runs one block behind the producers, halfbands 2 and 3 two
    const int it0 = (int)(ob0 - o_start);  // the first iteration that produces (chunk 0's warm-up blocks come from the history)
    const float2 *yh = y0_hist + (long long)c * P.y0_pitch;

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

    if (role < 2) {
        // ------------------------------ producers ------------------------------
        // The hb11's seven taps pair up around the centre: h[5+e] (step[5+e] x[5+e] + step[5-e] x[5-e]) =
        //   step[5] h[5+e] (cos(e t) (x[5+e] + x[5-e]) + j sin(e t) (x[5+e] - x[5-e])),  t = 2 pi inc, e = 1 3 5, x[d] = x[S j - 10 + d],
        // so  y0[j] = pa5(j) (h5 x5 + sum_e hc_e s_e + hs_e r_e),  s_e = x[5+e] + x[5-e],  r_e = j (x[5+e] - x[5-e]),
        //     pa5(j) = amp e^{j 2 pi (phase0 + (S j - 4) inc)}:
        // the sums s_e and rotated differences r_e do not depend on the channel -- wave 0 forms them once per output and block
        // (56 lanes: 8 outputs x {h5 x5, s1, r1, s3, r3, s5, r5}, two raw samples each, straight from global memory) and parks them
        // in LDS; a producer then spends six packed FMAs with REAL per-channel coefficients per output instead of fourteen
        // (the complex products of the tap-by-tap form), and holds 6 coefficient registers instead of 56.
        const float amp = mix ? P.a_inf * P.gain0 : P.gain0;
        v2f_t hcs[3];  // (hc_e, hs_e) = h[5+e] (cos, sin)(e t)
        v2f_t rot = {1.f, 0.f}, rot8 = {1.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const float hq = q == 0 ? hb_tap<11>(6) : q == 1 ? hb_tap<11>(8) : hb_tap<11>(10);
            const float2 e = mix ? oc->step[2 * q + 1] : make_float2(1.f, 0.f);
            hcs[q] = v2f_t{hq * e.x, hq * e.y};
        }
        if (mix) {
            const float2 r1 = cis_cycles((double)S * inc), r8 = cis_cycles((double)(8 * S) * inc);
            rot = v2f_t{r1.x, r1.y};
            rot8 = v2f_t{r8.x, r8.y};
        }
        const int k0 = 4 * role;  // this wave's outputs of a block: k0 .. k0 + 3
        // feeder lane l < 56: output k = l / 7 of the block, quantity q = l % 7 (0: h5 x5; 1 3 5: s_e; 2 4 6: r_e; e = 1 1 3 3 5 5).
        // Always two loads per lane (samples before the call's start come from the previous call's tail; lanes before that tail
        // or past the end load a valid address and are never used): with a fixed number of loads per fetch the compiler can count,
        // and the wait before a park leaves the younger fetch in flight
        const int fl = lane < 56 ? lane : 55, fk = fl / 7, fq = fl - 7 * fk, fe = fq == 0 ? 0 : 2 * ((fq - 1) >> 1) + 1;
        auto fetch = [&](long long o, float2 (&r)[2]) {
            const long long b = (long long)S * (8 * o - 7 + fk) - 5;  // the centre tap's sample
#pragma unroll
            for (int q = 0; q < 2; q++) {
                long long i = b + (q == 0 ? fe : -fe);
                i = i < n_in ? i : n_in - 1;
                const float2 *p = i >= 0 ? in + i : x_hist + (i >= -16 ? 16 + i : 0);
                r[q] = *p;
            }
        };
        // two blocks in flight, in two register sets that swap roles from block to block (the loop is unrolled by two: a
        // register move would have to wait for the load it moves)
        float2 xn[2], xn2[2];
        auto park = [&](long long o, const float2 (&r)[2]) {
            const float h5 = hb_tap<11>(5);
            const float2 sum = cadd(r[0], r[1]), dif = csub(r[0], r[1]);
            float2 v = (fq & 1) ? sum : make_float2(-dif.y, dif.x);
            if (fq == 0) v = cscale(r[0], h5);
            srw[o & 1][lane] = v;
        };
        if (role == 0) {
            fetch(ob0, xn);
            park(ob0, xn);
            fetch(ob0 + 1, xn);
            fetch(ob0 + 2, xn2);
        }
        __syncthreads();
        v2f_t pa_blk = {amp, 0.f};
        // one block; FEEDER (wave 0) also parks block o + 1's sums (the buffer the block before this one read) and starts
        // the fetch of block o + 3 into the registers that held them.  No condition around the fetch, and one copy of the loop per
        // role: only then does the compiler know that exactly one younger fetch is in flight when it waits for the parked one.
        auto produce = [&](auto feeder, long long o, float2 (&xq)[2]) {
            const long long j0 = 8 * o - 7;
            if (((o - o_start) & 7) == 0 || o == 0) {
                // exact phase every eighth block (each producer for its own first output), a constant rotation in between
                if (mix) {
                    const float2 e = cscale(cis_cycles(phase0 + (double)((long long)S * (j0 + k0) - 4) * inc), amp);
                    pa_blk = v2f_t{e.x, e.y};
                }
            } else {
                pa_blk = cmul_pk(rot8, pa_blk);
            }
            // this wave's 4 x 7 shared quantities: 224 consecutive bytes, the same address in every lane
            const float4 *sp = reinterpret_cast<const float4 *>(srw[o & 1] + 7 * k0);
            float4 s4[14];
#pragma unroll
            for (int m = 0; m < 14; m++) s4[m] = sp[m];
            v2f_t y[4];
            v2f_t pa = pa_blk;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                auto sv = [&](int q) { const int i = 7 * k + q; return (i & 1) ? v2f_t{s4[i / 2].z, s4[i / 2].w} : v2f_t{s4[i / 2].x, s4[i / 2].y}; };
                v2f_t acc = sv(0);
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    acc = fma_lo_pk(acc, hcs[q], sv(1 + 2 * q));
                    acc = fma_hi_pk(acc, hcs[q], sv(2 + 2 * q));
                }
                y[k] = cmul_pk(pa, acc);
                if (k < 3) pa = cmul_pk(rot, pa);
            }
            float4 *rg = ring[o & 1] + (k0 / 2) * 64 + lane;
            rg[0] = make_float4(y[0].x, y[0].y, y[1].x, y[1].y);
            rg[64] = make_float4(y[2].x, y[2].y, y[3].x, y[3].y);
            if (decltype(feeder)::value) {
                park(o + 1, xq);
                fetch(o + 3, xq);
            }
            __syncthreads();
        };
        auto run = [&](auto feeder) {
            int it = 0;
            for (; it < it0; it++) __syncthreads();
            for (; it + 1 < nb; it += 2) {
                produce(feeder, o_start + it, xn);
                produce(feeder, o_start + it + 1, xn2);
            }
            if (it < nb) produce(feeder, o_start + it, xn);
            __syncthreads();  // (the halfbands' last two blocks)
            __syncthreads();
        };
        if (role == 0) run(std::true_type{});
        else run(std::false_type{});
        if (last_chunk) {
            // the next call's raw-input tail and, for the two-kernel route, the mixed-sample history m[n-10 .. n-1] (each with its exact phase)
            if (role == 0 && blockIdx.y == 0 && lane < 16) xh_out[lane] = in[n_in - 16 + lane];
            if (role == 1 && live && mixed_hist_out != nullptr) {
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
        return;
    }

    // ------------------------------ the halfbands: wave 2 stage 1 (one block behind), wave 3 stages 2 and 3 (two behind) ------------------------------
    auto bc = [](float h) { return make_float2(h, h); };
    if (role == 2) {
        float2 a1[N1];
#pragma unroll
        for (int i = 0; i < N1; i++) a1[i] = make_float2(0.f, 0.f);
        __syncthreads();  // (the producers' prologue barrier)
        for (int it = 0; it < n_iter; it++) {
            const long long o = o_start + it - 1;  // the block this wave works on in this iteration
            if (it >= 1 && it <= nb) {
                const long long j0 = 8 * o - 7;
                float2 y0[8];
                if (o <= 0) {
#pragma unroll
                    for (int k = 0; k < 7; k++) y0[k] = yh[j0 + k];  // chunk 0's warm-up: the previous call's tail (block 0: only output 0 is new)
                    if (o < 0) {
                        y0[7] = yh[j0 + 7];
                    } else {
                        const float4 v = ring[0][3 * 64 + lane];
                        y0[7] = make_float2(v.z, v.w);
                    }
                } else {
                    const float4 *rg = ring[o & 1] + lane;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float4 v = rg[k * 64];
                        y0[2 * k] = make_float2(v.x, v.y);
                        y0[2 * k + 1] = make_float2(v.z, v.w);
                    }
                }
                const bool keep = last_chunk && o >= 0 && j0 + 7 >= len0 - HY;
                if (keep) {  // eight outputs x 64 channels -> 64-byte row segments of the history staging rows
#pragma unroll
                    for (int k = 0; k < 8; k++) htile[k * 65 + lane] = y0[k];
                    wave_sync();
#pragma unroll 1
                    for (int i8 = 0; i8 < 8; i8++) {
                        const int idx = i8 * 64 + lane;
                        const int ch = idx >> 3, k = idx & 7;
                        const long long j = j0 + k;
                        if (cbase + ch < P.n_chan && j >= len0 - HY && j < len0)
                            y0_stage[(long long)(cbase + ch) * HY + (j - (len0 - HY))] = htile[k * 65 + ch];
                    }
                    wave_sync();
                }
                if (o != P.n_out) {
                    float2 y1[4];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (k & 1) {  // j even: the even taps of stage 1; y1[4 o - 3 + b] is complete
                            const int b = (k - 1) / 2;
#pragma unroll
                            for (int t = 0; t < (T1 + 1) / 2; t++) {
                                const float2 v = y0[k] * bc(hb_tap<T1>(T1 - 1 - 2 * t));
                                a1[b + t] = t == (T1 + 1) / 2 - 1 ? v : a1[b + t] + v;
                            }
                            y1[b] = a1[b];
                        } else {
                            a1[(k + PC1 - 1) / 2] = a1[(k + PC1 - 1) / 2] + y0[k] * bc(hb_tap<T1>(PC1));
                        }
                    }
                    float4 *wr = ring1[o & 1] + lane;
                    wr[0] = make_float4(y1[0].x, y1[0].y, y1[1].x, y1[1].y);
                    wr[64] = make_float4(y1[2].x, y1[2].y, y1[3].x, y1[3].y);
#pragma unroll
                    for (int i = 0; i + 4 < N1; i++) a1[i] = a1[i + 4];
                }
            }
            __syncthreads();
        }
        return;
    }
    float2 a2[N2], a3[N3];
#pragma unroll
    for (int i = 0; i < N2; i++) a2[i] = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < N3; i++) a3[i] = make_float2(0.f, 0.f);
    float2 y3 = make_float2(0.f, 0.f);
    __syncthreads();  // (the producers' prologue barrier)
    for (int it = 0; it < n_iter; it++) {
        const long long o = o_start + it - 2;
        if (it >= 2 && o != P.n_out) {  // (the history-only block has no outputs)
            const float4 *rd = ring1[o & 1] + lane;
            const float4 v01 = rd[0], v23 = rd[64];
            const float2 y1[4] = {make_float2(v01.x, v01.y), make_float2(v01.z, v01.w), make_float2(v23.x, v23.y), make_float2(v23.z, v23.w)};
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (b & 1) {  // m even
                    const int b2 = (b - 1) / 2;
#pragma unroll
                    for (int t = 0; t < (T2 + 1) / 2; t++) {
                        const float2 v = y1[b] * bc(hb_tap<T2>(T2 - 1 - 2 * t));
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
                    a2[(b + PC2 - 1) / 2] = a2[(b + PC2 - 1) / 2] + y1[b] * bc(hb_tap<T2>(PC2));
                }
            }
            if (o >= o0 && o < o1) {
                const int jt = (int)(o - o0) & 15;
                tile[jt * 65 + lane] = cscale(y3, P.gain);
                if (jt == 15 || o == o1 - 1) {
                    const long long ob = o - jt;
                    wave_sync();
#pragma unroll 1  // (unrolled, the sixteen row pointers are carried, and stepped, through every block)
                    for (int i16 = 0; i16 < 16; i16++) {
                        const int idx = i16 * 64 + lane;
                        const int ch = idx >> 4, t = idx & 15;
                        if (cbase + ch < P.n_chan && t <= jt) out[(long long)(cbase + ch) * P.out_pitch + ob + t] = tile[t * 65 + ch];
                    }
                    wave_sync();
                }
            }
#pragma unroll
            for (int i = 0; i + 2 < N2; i++) a2[i] = a2[i + 2];
#pragma unroll
            for (int i = 0; i + 1 < N3; i++) a3[i] = a3[i + 1];
        }
        __syncthreads();
    }
}

}  // namespace pg
