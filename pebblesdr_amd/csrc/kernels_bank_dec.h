// kernels_bank_dec.h -- mixer + the WHOLE decimator of a bank of channels tuned off one shared stream, in one kernel of
// independent single-wave workgroups: no LDS hand-offs between waves, no workgroup barrier, first stage on the matrix pipe.
// Mixer::processBlock + Decimator::process (pebblelib/mixer.cpp:48-81, decimator.cpp:152-226, 593-659, 695-737).
//
// Successor of k_mix_dec_fused (kernels_fused_dec.h: four-wave pipeline with one barrier per block, 0.086 ms per configs[2]
// call at 2.7 waves per SIMD, VALU active a third of the time, 21 M vector instructions, an LDS ring between the stages).
//
// The first stage is a contraction.  With the oscillator factored out of a stage's window (k_mix_hb11_lean) and the
// window's symmetry used (k_mix_dec_fused),
//     y0[c][j] = pa_c(j) * sum_p ( g_p cos(th_cp) s_p(j) + g_p sin(th_cp) r_p(j) ),    th_cp = 2 pi inc_c e_p,
//     s_p(j) = x[ctr(j) + e_p] + x[ctr(j) - e_p],   r_p(j) = j (x[ctr(j) + e_p] - x[ctr(j) - e_p]),
// where the pairs p run over the symmetric taps of the first stage (hb11 x S: e = 0 1 3 5; a merged CIC3 in front of it:
// twelve pairs), s_p and r_p do not depend on the channel and the coefficients do not depend on time: a [time x 2 NP] by
// [2 NP x channel] product of REAL matrices for the real and for the imaginary part.  v_mfma_f32_32x32x2_f32 does it in exact
// fp32 (a k-ordered fmaf chain, bit for bit) at the vector pipe's own rate but beside it: NP matrix instructions per block of
// 8 outputs x 32 channels x 2 chunks replace 6 NP / 4 packed FMAs per output and channel, and the shared sums need no LDS
// (every lane forms ITS element of the A operand from two 4-byte loads that hit L1/L2).
//
// Mapping.  D = A B has its column on the lane (l & 31) and its rows in the registers, 4 (l >> 5) + (r & 3) + 8 (r >> 2): the
// columns are 32 CHANNELS of the bank, the rows are [chunk l >> 5][output k = r >> 1][re / im = r & 1] -- each lane ends up
// with the eight complex first-stage outputs of a block for one channel and ONE OF TWO time chunks the wave works on.  From
// there on a lane walks its (channel, chunk) through time exactly as in k_mix_dec_fused: y0 = pa acc (the oscillator at the
// window's centre: exact phase every eighth block, rotations in between), then the three halfbands in TRANSPOSED form with
// every running sum in registers.  One wave does all of it, so nothing is handed over and nothing waits: a workgroup is one
// wave, the launch puts one (or a few) on every SIMD, and the matrix pipe works on block o + 1 while the vector pipe runs the
// halfbands of block o (the block's code places one matrix instruction between two halfband stages).  The running sums shift
// by renaming: every one of them is written in every block, so the compiler writes it to its next place (no register moves).
//
// Chunks, warm-up and histories are k_mix_dec_fused's (same FusedDecGeom, same stage-0 history rows, same raw-input tail and
// mixed-sample history left for the other routes), so a call may take any of the three routes.  What a call LEAVES (its last HY
// first-stage outputs, the raw tail, the mixed-sample history) is the work of a few extra waves of the same launch that
// recompute those 22 blocks' first stage and nothing else: the main waves carry no stores besides their results (a store in
// front of a fetch makes the wait for the fetch a wait for the store -- one counter, in order).
// Workgroup -> (chunk pair, channel group): consecutive workgroup ids are dealt to the eight XCDs round-robin, so the id's low
// three bits select the chunk pair's residue and all channel groups of one stretch of the stream run on ONE XCD: the shared
// stream is fetched once into one L2 instead of once into each (k_mix_dec_fused: 21.9 MB read for a 4.2 MB stream).
#pragma once
#include "kernels_fused_dec.h"

namespace pg {

typedef float v16f_t __attribute__((ext_vector_type(16)));

template <int NP>
struct BankDecParams {
    long long n_out;          // final outputs per channel in this call
    long long out_pitch;      // float2 per row of `out`
    long long y0_pitch;       // row pitch of the stage-0 history (float2)
    long long n_in;           // input samples of this call = S * 8 * n_out
    int S;                    // input samples per first-stage output
    int L;                    // final outputs per chunk (multiple of 16)
    int n_chan, n_chunks, n_groups;  // n_groups = ceil(n_chan / 32)
    int hist_pitch;           // mixed-history row pitch (kMaxTaps)
    int xh;                   // depth of the raw-input tail x_hist (samples in front of the call)
    int min_off, max_off;     // smallest / largest sample offset of the front's window relative to S j
    float a_inf, gain;        // oscillator amplitude, gain on the final output (the first stage's own gain is 1 in such a chain)
    double ctr1;              // the window's centre + 1 relative to S j: pa(j) = amp e^{j 2 pi (phase0 + (S j + ctr1) inc)}
    int oa[NP], ob[NP];       // pair p = samples S j + oa[p], S j + ob[p]
    float g[NP];              // its tap (a centre tap as a pair with itself and half the weight)
    float e[NP];              // (oa - ob) / 2
    const float2 *state_in;   // [n_chan][N1 + N2 + N3] the halfbands' running sums where the previous call ended (nullptr: that call took another route:
                              // chunk 0 then warms up from the first-stage history)
    float2 *state_out;        // the same for the next call
    int hist_split;           // history waves per channel group
    int cic_s0;               // 0: the front is hb11 x S (the two-kernel route keeps ten mixed samples); else the merged CIC3's stride S0 in front of
                              // the hb11 (that route keeps the call's last twelve mixed sample pairs S0 P, S0 P + 1)
    // The oscillators' per-call fields (OscBank::advance restated on the device) ride on this launch's history waves when the caller
    // hands them over: dyn_out[c] receives channel c's fields advanced by this call (adv[c] = frac(n inc), adv_n samples).  dyn_in, when
    // not null, is what the previous such launch wrote for THIS call: every wave then reads phase and mixer switch from it, and the
    // history waves also advance osc_rw[c] in place (no wave of this launch reads it), so that kernels of the other routes find current
    // values -- and no tail launch is needed for the oscillators.  With dyn_in == nullptr the waves read osc[c]; the caller's tail launch
    // advances it in place as before.
    const OscDyn *dyn_in;
    OscDyn *dyn_out;
    ChanOsc *osc_rw;
    const double *adv;
    unsigned adv_n, pad_;
    unsigned long long *clk;  // diagnosis (PEBBLEGPU_BANK_CLK): per wave {shader clocks, 100 MHz ticks, blocks} of its block loop; nullptr otherwise
};

// grid: n_main = 8 * ceil(pairs / 8) * ceil(n_groups / 4) workgroups of four INDEPENDENT waves (four channel groups of one chunk
// pair; pairs = ceil(n_chunks / 2)), then ceil(n_groups / 4) workgroups of history waves.  (Single-wave workgroups were measured
// first: the dispatcher put all four of a CU on one SIMD -- 1900 clocks per block instead of 1250; the four waves of one workgroup
// go to the four SIMDs.)
template <int NP, int T1, int T2, int T3, int DBG = 0, int MINW = 2, int EARLY = (NP == 4 && MINW == 2 ? 1 : 0)>
static __global__ __launch_bounds__(256, MINW) void k_mix_dec_mfma(const float2 *__restrict__ in, float2 *__restrict__ out, const ChanOsc *__restrict__ osc,
                                                               OscDynInline dyn, const float2 *__restrict__ x_hist, float2 *__restrict__ xh_out,
                                                               const float2 *__restrict__ y0_hist, float2 *__restrict__ y0_stage,
                                                               float2 *__restrict__ mixed_hist_out, BankDecParams<NP> P)
{
    using G = FusedDecGeom<T1, T2, T3>;
    constexpr int H1 = (T1 + 1) / 2, H2 = (T2 + 1) / 2, H3 = (T3 + 1) / 2;
    constexpr int N1 = G::N1, N2 = G::N2, N3 = G::N3, HY = G::HY, WARM = G::warm;
    constexpr int PC1 = (T1 - 1) / 2, PC2 = (T2 - 1) / 2, PC3 = (T3 - 1) / 2;
    constexpr bool kEarlyStore = EARLY != 0;  // where a block's store is issued (see the block's code)
    constexpr int kStoreAux = (DBG & 128) != 0 ? 2 : (DBG & 256) != 0 ? 1 : (DBG & 512) != 0 ? 17 : 0;  // (A/B: nontemporal / sc0 / sc0 sc1 result stores)
    __shared__ float2 tiles[4][2][16 * 65];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pairs = (P.n_chunks + 1) >> 1;
    const int gq = (P.n_groups + 3) >> 2;
    const int n_main = 8 * ((pairs + 7) >> 3) * gq;
    const bool hist_wave = (int)blockIdx.x >= n_main;  // (uniform)
    // The main waves are the launch's length: they issue in front of whatever shares their SIMD -- the previous call's band-pass (two-stage
    // calls) and this launch's own history waves (the main waves that had one of those beside them ran 1418 clocks per block against 1172
    // and ended the kernel 8 us after the rest)
    if (!hist_wave) __builtin_amdgcn_s_setprio(3);
    const int wq = blockIdx.x >> 3;
    const int hq = hist_wave ? (int)blockIdx.x - n_main : 0;  // history workgroups: [channel-group quad][part]
    const int hpart = hq % P.hist_split;
    const int pair = hist_wave ? 0 : (wq / gq) * 8 + (blockIdx.x & 7);
    const int grp = hist_wave ? (hq / P.hist_split) * 4 + wv : (wq % gq) * 4 + wv;
    if (pair >= pairs || grp >= P.n_groups) return;  // wave-uniform (no workgroup barrier anywhere below)
    if constexpr ((DBG & 64) != 0) { if (hist_wave) return; }  // (timing experiment: no history waves)

    // ---- D side: this lane's channel and chunk ----
    const int S = P.S, L = P.L;
    const int ch_raw = grp * 32 + (lane & 31), chunk_raw = 2 * pair + (lane >> 5);
    const int c = ch_raw < P.n_chan ? ch_raw : P.n_chan - 1;
    const int chunk = chunk_raw < P.n_chunks ? chunk_raw : P.n_chunks - 1;
    const long long o0 = (long long)chunk * L;
    const long long len0 = 8 * P.n_out;
    const float2 *yh = y0_hist + (long long)c * P.y0_pitch;

    const ChanOsc *oc = &osc[c];
    const double inc = oc->inc;
    double phase0 = oc->phase0;
    uint32_t mix_on = oc->mix_on;
    if (P.dyn_in != nullptr) { phase0 = P.dyn_in[c].phase0; mix_on = P.dyn_in[c].mix_on; }
    if (dyn.use) {
#pragma unroll
        for (int k = 0; k < kOscInline; k++)
            if (c == k) { phase0 = dyn.d[k].phase0; mix_on = dyn.d[k].mix_on; }
    }
    const bool mix = mix_on != 0;
    const float amp = mix ? P.a_inf : 1.f;  // (first-stage outputs stay unscaled: they are the history the other routes share)

    // ---- B operand: lane l holds B[k = l >> 5][column l & 31] of every pair's 2 x 32 slice: g_p cos / g_p sin of this lane's channel ----
    const int kq = lane >> 5;
    float bco[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const float2 e = mix ? cis_cycles((double)P.e[p] * inc) : make_float2(1.f, 0.f);
        bco[p] = P.g[p] * (kq == 0 ? e.x : e.y);
    }
    v2f_t r1 = {1.f, 0.f}, r2 = r1, r4 = r1, r8 = r1;
    if (mix) {
        const float2 a = cis_cycles((double)S * inc), b = cis_cycles((double)(2 * S) * inc), d = cis_cycles((double)(4 * S) * inc),
                     f = cis_cycles((double)(8 * S) * inc);
        r1 = v2f_t{a.x, a.y}; r2 = v2f_t{b.x, b.y}; r4 = v2f_t{d.x, d.y}; r8 = v2f_t{f.x, f.y};
    }
    // c * x as ONE asm statement (between two statements the compiler pads a wait state: an issue slot each, and a lone wave pays
    // every slot)
    auto cmul2 = [](v2f_t cc, v2f_t x) {
        v2f_t r;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "=&v"(r) : "v"(cc), "v"(x));
        return r;
    };
    // the oscillator at the centres of a block's eight windows from the first one's
    auto osc8 = [&](v2f_t p0, v2f_t (&pa)[8]) {
        pa[0] = p0;
        pa[1] = cmul2(r1, pa[0]);
        pa[2] = cmul2(r2, pa[0]); pa[3] = cmul2(r2, pa[1]);
#pragma unroll
        for (int k = 0; k < 4; k++) pa[4 + k] = cmul2(r4, pa[k]);
    };
    auto osc_exact = [&](long long j0) {
        const float2 e = mix ? cscale(cis_cycles(phase0 + ((double)S * (double)j0 + P.ctr1) * inc), amp) : make_float2(amp, 0.f);
        return v2f_t{e.x, e.y};
    };
    // y0 = pa * D.  The first of a product's two instructions is the compiler's own, so it pads the matrix result's read hazard (it
    // does not model what is inside an asm statement); the second takes the half negation as an operand modifier
    auto first_stage_out = [&](const v16f_t &D, const v2f_t (&pa)[8], float2 (&y0)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const v2f_t d = {D[2 * k], D[2 * k + 1]};
            const v2f_t t = d.yx * pa[k].yy;  // (d.y p.y, d.x p.y)
            v2f_t y;
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(y) : "v"(d), "v"(pa[k]), "v"(t));  // (d.x p.x - t.x, d.y p.x + t.y)
            y0[k] = make_float2(y.x, y.y);
        }
    };

    // ---- A side: lane l holds A[row l & 31][k = l >> 5]; row = [half (l >> 2) & 1][output ((l >> 1) & 1) + 2 ((l >> 3) & 3)][re / im l & 1] ----
    const int iA = lane & 31, compA = iA & 1, kA = ((iA >> 1) & 1) + 2 * (iA >> 3), halfA = (iA >> 2) & 1;
    // k = 0: s_p = xa + xb in this row's component; k = 1: r_p = j (xa - xb): re = xb.im - xa.im, im = xa.re - xb.re
    const int compL = kq == 0 ? compA : 1 - compA;
    const float ca = (kq == 1 && compA == 0) ? -1.f : 1.f, cb = (kq == 1 && compA == 1) ? -1.f : 1.f;
    const float *inf = reinterpret_cast<const float *>(in);
    const float *xhf = reinterpret_cast<const float *>(x_hist);
    // Sample fetches.  A lane's two samples of a pair sit at one 32-bit byte offset `voff` (this lane's element of the window's first
    // sample; 64 S bytes further per block) plus the pair's place in the window as the instruction's scalar offset: no address
    // arithmetic in the loop.  The offset is clamped into the buffer: windows that reach in front of the call or past its end belong
    // to outputs nobody uses (chunk 0's warm-up blocks take the previous call's first-stage history; the output one past the call's
    // end) -- except the first o_safe blocks of the call proper, whose windows straddle the call's first sample: those go through
    // the edge variant of the fetch, which requests every value from the call and from the previous call's raw tail and selects.
    const unsigned vmax = (unsigned)(8 * (P.n_in - 1 - (P.max_off - P.min_off)) + 4 * compL);  // the last window that ends inside the call
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(inf), 0, (int)(unsigned)(P.n_in * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_xh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xhf), 0, (int)(unsigned)(P.xh * 8), 0x00020000);
    const int o_safe = (int)(((P.min_off < 0 ? ((long long)(-P.min_off) + S - 1) / S : 0) + 14) / 8);  // S (8 o - 7) + min_off >= 0 from block o_safe on
    float ld[2 * NP], ld_alt[2 * NP];  // the fetched samples of the next block (two sets: the main loop alternates them, no copies)
    const v16f_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (hist_wave) {
        // ------------------------------------------------------------------------------------------------------------------------
        // What the call leaves: its last HY first-stage outputs (blocks n_out - 22 .. n_out, twelve per lane half), the raw-input
        // tail, and for the two-kernel route the mixed-sample history m[n-10 .. n-1] (each with its exact phase)
        // ------------------------------------------------------------------------------------------------------------------------
        constexpr int HB = (HY / 8 + 2) / 2;  // blocks per half
        const long long ohA = P.n_out - 2 * HB + 1 + (long long)halfA * HB, ohD = P.n_out - 2 * HB + 1 + (long long)(lane >> 5) * HB;
        auto hfetch = [&](int i) {  // block i's samples (the window's first sample lies inside the call, or its outputs are not kept)
            const long long sj = (long long)S * (8 * (ohA + i) - 7 + kA) + P.min_off;
            long long v8 = 8 * sj + 4 * compL;
            v8 = v8 < 0 ? 0 : v8;
            const unsigned vc = (unsigned long long)v8 < (unsigned long long)vmax ? (unsigned)v8 : vmax;
#pragma unroll
            for (int p = 0; p < NP; p++) {
                ld[2 * p] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vc, 8 * (P.oa[p] - P.min_off), 0));
                ld[2 * p + 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vc, 8 * (P.ob[p] - P.min_off), 0));
            }
        };
        const int per = (HB + P.hist_split - 1) / P.hist_split, i_lo = hpart * per, i_hi = i_lo + per < HB ? i_lo + per : HB;
        if (i_lo < i_hi) hfetch(i_lo);
        for (int i = i_lo; i < i_hi; i++) {
            float av[NP];
#pragma unroll
            for (int p = 0; p < NP; p++) av[p] = __builtin_fmaf(ca, ld[2 * p], cb * ld[2 * p + 1]);
            hfetch(i + 1 < i_hi ? i + 1 : i);  // the next block's samples travel while this one is worked on
            v16f_t acc = zero16;
#pragma unroll
            for (int p = 0; p < NP; p++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], bco[p], acc, 0, 0, 0);
            const long long j0 = 8 * (ohD + i) - 7;
            v2f_t pa[8];
            osc8(osc_exact(j0), pa);
            float2 y0[8];
            first_stage_out(acc, pa, y0);
            if (ch_raw < P.n_chan) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const long long j = j0 + k;
                    if (j >= len0 - HY && j < len0) y0_stage[(long long)c * HY + (j - (len0 - HY))] = y0[k];
                }
            }
        }
        if (P.dyn_out != nullptr && hpart == 0 && lane < 32 && ch_raw < P.n_chan) {
            // OscBank::advance for this channel (the arithmetic of save_tails_block, tail_refresh.h)
            double pn = phase0 + P.adv[c];
            pn -= floor(pn);
            pn = pn >= 1.0 ? 0.0 : pn;
            const uint32_t n_old = P.dyn_in != nullptr ? P.dyn_in[c].n0 : oc->n0;
            uint32_t n_new = n_old + P.adv_n;
            n_new = n_new > (uint32_t)kAmpTab ? (uint32_t)kAmpTab : n_new;
            P.dyn_out[c] = OscDyn{pn, n_new, mix_on};
            if (P.dyn_in != nullptr) {
                P.osc_rw[c].phase0 = pn;
                P.osc_rw[c].n0 = n_new;
            }
        }
        if (grp == 0 && hpart == 0) {
            for (int i = lane; i < P.xh; i += 64) {
                const long long src = P.n_in - P.xh + i;
                xh_out[i] = src >= 0 ? in[src] : x_hist[P.xh + src];
            }
        }
        if (hpart == P.hist_split - 1 && ch_raw < P.n_chan && lane < 32 && mixed_hist_out != nullptr) {
            float2 *hp = mixed_hist_out + (long long)c * P.hist_pitch;
            if (P.cic_s0 == 0) {
                float2 xv[10];
#pragma unroll
                for (int q = 0; q < 10; q++) xv[q] = in[P.n_in - 10 + q];
                // (exact phase at the first of the ten, a rotation per sample from there: ten steps)
                const float2 st = mix ? cis_cycles(inc) : make_float2(1.f, 0.f);
                float2 ph = mix ? cscale(cis_cycles(phase0 + (double)(P.n_in - 9) * inc), P.a_inf) : make_float2(1.f, 0.f);
#pragma unroll
                for (int q = 0; q < 10; q++) {
                    hp[q] = mix ? cmul(ph, xv[q]) : xv[q];
                    ph = cmul(st, ph);
                }
            } else {
                // k_mix_cic_hb's history: the last twelve sample pairs, each sample with its exact phase
                const long long P0 = P.n_in / P.cic_s0 - 12;
                float4 xv[12];
#pragma unroll
                for (int q = 0; q < 12; q++) xv[q] = *reinterpret_cast<const float4 *>(in + (P0 + q) * (long long)P.cic_s0);
                const float2 st = mix ? cis_cycles(inc) : make_float2(1.f, 0.f);
#pragma unroll
                for (int q = 0; q < 12; q++) {
                    const float2 ph = mix ? cscale(cis_cycles(phase0 + (double)((P0 + q) * (long long)P.cic_s0 + 1) * inc), P.a_inf) : make_float2(1.f, 0.f);
                    hp[2 * q] = mix ? cmul(ph, make_float2(xv[q].x, xv[q].y)) : make_float2(xv[q].x, xv[q].y);
                    hp[2 * q + 1] = mix ? cmul(cmul(st, ph), make_float2(xv[q].z, xv[q].w)) : make_float2(xv[q].z, xv[q].w);
                }
            }
        }
        return;
    }

    // ----------------------------------------------------------------------------------------------------------------------------
    // main waves
    // ----------------------------------------------------------------------------------------------------------------------------
    const int nb = L + WARM;
    int chunkA = 2 * pair + halfA;
    if (chunkA >= P.n_chunks) chunkA = P.n_chunks - 1;
    const long long oA_start = (long long)chunkA * L - WARM;
    // (modulo 2^32: exact wherever it is used unclamped -- the host sends calls of 4 GiB or more down the other routes)
    unsigned voff = (unsigned)(4 * (2LL * S * (8 * oA_start - 7 + kA) + compL + 2LL * P.min_off));
    // the raw samples of block `it` of both chunks (block index relative to the chunks' first block); `voff` is that block's
    typedef unsigned v3u_t __attribute__((ext_vector_type(3)));
    auto fetch_all = [&](int it, auto edgec, float (&ld)[2 * NP]) {
        if constexpr (!decltype(edgec)::value && (DBG & 1) != 0) return;  // (timing experiment: no sample fetches in the plain blocks)
        if constexpr (decltype(edgec)::value && NP == 4) {
            // The blocks at the call's start (chunk pair 0 only): every value on its own -- from the call (index clamped into it) AND
            // from the previous call's raw tail (clamped into that), selected by the sign of its index.  No branch: with a branch
            // around a slower path the compiler knew no count of the requests in flight and every one of these blocks drained them all,
            // the ones it had just issued included.
            const int sj = S * (8 * (int)(oA_start + it) - 7 + kA);
            const int n_in_i = (int)P.n_in;  // (the host keeps calls under 4 GiB)
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int i = sj + (q == 0 ? P.oa[p] : P.ob[p]);
                    int ia = i < 0 ? 0 : i;
                    ia = ia < n_in_i ? ia : n_in_i - 1;
                    int ib = P.xh + i;
                    ib = ib < 0 ? 0 : ib;
                    ib = ib < P.xh ? ib : P.xh - 1;
                    const float va = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, 8u * (unsigned)ia + 4u * (unsigned)compL, 0, 0));
                    const float vb = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_xh, 8u * (unsigned)ib + 4u * (unsigned)compL, 0, 0));
                    ld[2 * p + q] = i < 0 ? vb : va;
                }
            }
        } else if (decltype(edgec)::value && pair == 0 &&
                   ((it >= WARM && it < WARM + o_safe) || (it >= WARM - L && it < WARM - L + o_safe))) {
            // (the CIC front keeps the branch: twenty-four values a block, each with two requests, cost its edge blocks more than the
            // drain does -- configs[3] on one stream 0.117 -> 0.134 ms with the branch-free form)
            const long long sj = (long long)S * (8 * (oA_start + it) - 7 + kA);
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    long long i = sj + (q == 0 ? P.oa[p] : P.ob[p]);
                    i = i < P.n_in ? i : P.n_in - 1;
                    long long ih = (long long)P.xh + i;  // in front of the call: the previous call's tail (never used when older than that)
                    ih = ih > 0 ? ih : 0;
                    ld[2 * p + q] = i >= 0 ? inf[2 * i + compL] : xhf[2 * ih + compL];
                }
            }
        } else {
            const unsigned vc = voff < vmax ? voff : vmax;
            if constexpr (NP == 12) {
                // A merged CIC3 in front: its window is twelve sample pairs x[S0 t], x[S0 t + 1] (t = 0 .. 11 from the window's first sample),
                // one 128-byte line each at S0 = 16 -- and each line serves two of the twelve symmetric pairs: (od_{11-i}, ev_i) and
                // (ev_{11-i}, od_i).  One 12-byte load per line and lane ([ev.c, the other component, od.c]) touches every line ONCE; with
                // a 4-byte load per operand each line was touched four times by a wave whose working set (192 lines per block, four
                // waves) does not fit the CU's vector cache: 4.5 GB through L2 per configs[3] call, 0.158 ms against 0.094 (two 4-byte
                // loads per line, back to back: 0.161 -- the second does not meet the first one's miss)
                v3u_t ln[12];
#pragma unroll
                for (int t = 0; t < 12; t++) ln[t] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, vc, 8 * P.cic_s0 * t, 0);
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    ld[4 * i] = __uint_as_float(ln[11 - i].z);      // pair 2 i: (od of line 11 - i, ev of line i)
                    ld[4 * i + 1] = __uint_as_float(ln[i].x);
                    ld[4 * i + 2] = __uint_as_float(ln[11 - i].x);  // pair 2 i + 1: (ev of line 11 - i, od of line i)
                    ld[4 * i + 3] = __uint_as_float(ln[i].z);
                }
            } else {
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    ld[2 * p] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vc, 8 * (P.oa[p] - P.min_off), 0));
                    ld[2 * p + 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vc, 8 * (P.ob[p] - P.min_off), 0));
                }
            }
        }
    };
    v16f_t D;

    float2 a1[N1], a2[N2], a3[N3];
#pragma unroll
    for (int i = 0; i < N1; i++) a1[i] = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < N2; i++) a2[i] = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < N3; i++) a3[i] = make_float2(0.f, 0.f);
    auto bc = [](float h) { return make_float2(h, h); };

    // prologue: block 0 through the matrix pipe, block 1's samples on their way
    {
        v16f_t acc = zero16;
        if (pair == 0) fetch_all(0, std::true_type{}, ld);
        else fetch_all(0, std::false_type{}, ld);
#pragma unroll
        for (int p = 0; p < NP; p++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_fmaf(ca, ld[2 * p], cb * ld[2 * p + 1]), bco[p], acc, 0, 0, 0);
        D = acc;
        voff += 64u * S;
        if (pair == 0) fetch_all(1, std::true_type{}, ld);
        else fetch_all(1, std::false_type{}, ld);
        voff += 64u * S;
    }

    // Results leave as whole 128-byte row segments, one store instruction per block: a block parks its 64 results in one of two LDS
    // tiles [16 outputs][64 (channel, chunk) lanes] and stores one sixteenth of the tile the sixteen blocks before it filled: two
    // channels' rows x both chunks x 16 outputs (lane = [chunk][row][output]).  The lane part of the address never changes, the rest
    // is the instruction's scalar offset, and everything that moves from block to block is a running value (one add each): every
    // instruction of any kind is an issue slot of a wave that has its SIMD to itself.
    // (Every store is issued: the ones that must not land carry an out-of-range lane offset, which the buffer's range check drops --
    // the scalar offset takes no part in that check.  A store inside a branch would leave the compiler no static count of the memory
    // operations in flight, and the wait for the next block's samples would become a wait for the store.)
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(unsigned)((unsigned long long)P.n_chan * (unsigned long long)P.out_pitch * 8ull), 0x00020000);
    constexpr unsigned kNoStore = 0xFFFFFF00u;
    const int st_t = lane & 15, st_row = (lane >> 4) & 1, st_half = lane >> 5;
    const int ch_lim = P.n_chan - grp * 32;  // channels of this group that exist
    const bool ragged = ch_lim < 32;         // (uniform)
    // 32-bit offsets: the host keeps the result rows under 4 GiB
    const unsigned ovoff = 2 * pair + st_half < P.n_chunks ? 8u * ((unsigned)st_row * (unsigned)P.out_pitch + (unsigned)st_half * (unsigned)L + (unsigned)st_t) : kNoStore;
    const unsigned so_grp = 8u * (unsigned)(grp * 32) * (unsigned)P.out_pitch + 8u * (unsigned)(2 * pair) * (unsigned)L;
    const unsigned so_row2 = 8u * 2u * (unsigned)P.out_pitch;
    typedef v2f_t __attribute__((address_space(3))) lds_f2;  // (the native vector type: float2's operators do not exist in that address space)
    const unsigned tb0 = (unsigned)(uintptr_t)(lds_f2 *)(v2f_t *)tiles[wv][0], tb1 = (unsigned)(uintptr_t)(lds_f2 *)(v2f_t *)tiles[wv][1];  // (LDS byte addresses)
    const unsigned lds_wr0 = tb0 + 8u * lane, lds_wr1 = tb1 + 8u * lane;
    const unsigned lds_rd0 = tb0 + 8u * (st_t * 65 + st_half * 32 + st_row), lds_rd1 = tb1 + 8u * (st_t * 65 + st_half * 32 + st_row);
    auto lds_at = [](unsigned a) { return (lds_f2 *)(uintptr_t)a; };
    // running values of a round of sixteen blocks (set by round_start)
    unsigned twr = lds_wr0 + 8u * 65u * (unsigned)((-WARM) & 15), trd = lds_rd1, so_run = 0, vo_cur = kNoStore;
    int row_run = st_row;  // (ragged groups) the channel row this lane stores next
    v2f_t pa_blk = {amp, 0.f};
    // the first block of a round (rel a multiple of 16): exact oscillator phase, the tiles change roles, the store's running offsets
    auto round_start = [&](int rel, long long j0) {
        pa_blk = osc_exact(j0);
        const bool odd = ((rel >> 4) & 1) != 0;
        twr = odd ? lds_wr1 : lds_wr0;
        trd = odd ? lds_rd0 : lds_rd1;
        so_run = so_grp + 8u * (unsigned)(rel - 16);
        vo_cur = rel >= 16 ? ovoff : kNoStore;
        row_run = st_row;
    };
    // the block's result into the tile being filled; one sixteenth of the other tile to memory
    auto park = [&](float2 y) {
        *lds_at(twr) = v2f_t{y.x, y.y};
        twr += 8u * 65u;
    };
    auto store_sixteenth = [&](float2 sv) {
        unsigned vo = vo_cur;
        if (ragged) {
            vo = row_run < ch_lim ? vo : kNoStore;
            row_run += 2;
        }
        // (the scalar offset IS uniform; said so explicitly, or the compiler wraps the store in a loop over the lanes' values)
        __builtin_amdgcn_raw_buffer_store_b64(v2f_t{sv.x, sv.y}, orsrc, vo, __builtin_amdgcn_readfirstlane((int)so_run), kStoreAux);
        so_run += so_row2;
    };

    // One block; `it` = its index in the chunk (0 = first warm-up block).  On entry D holds the block's matrix products and ld the raw
    // samples of block it + 1.  EDGE: the variant for the blocks at the call's start (first-stage outputs in front of the call from the
    // previous call's tail `yq`, requested a block ahead; the slow sample fetches of the straddling blocks); the plain variant is one
    // straight run of code.
    auto block = [&](int it, auto edgec, float2 (&yq)[8], float (&ld_use)[2 * NP], float (&ld_next)[2 * NP]) {
        constexpr bool EDGE = decltype(edgec)::value;
        const int rel = it - WARM;               // block o = o0 + rel = final output o
        const long long o = o0 + rel;
        const long long j0 = 8 * o - 7;          // its first-stage outputs j0 .. j0 + 7
        // the oscillator at the centre of output j0's window: exact at every round's start, a constant rotation in between
        if ((rel & 15) == 0) round_start(rel, j0);
        else pa_blk = cmul2(r8, pa_blk);
        // the tile read of this block's store: first thing, the store follows the sample requests below
        float2 sv = make_float2(0.f, 0.f);
        if constexpr ((DBG & 8) == 0 || EDGE) {
            const v2f_t t = *lds_at(trd);
            sv = make_float2(t.x, t.y);
            trd += 16u;
        }
        v2f_t pa[8];
        float2 y0[8];
        if constexpr ((DBG & 16) != 0 && !EDGE) {  // (timing experiment: no oscillator, no product)
#pragma unroll
            for (int k = 0; k < 8; k++) y0[k] = make_float2(D[2 * k] + pa_blk.x, D[2 * k + 1]);
        } else {
            osc8(pa_blk, pa);
            first_stage_out(D, pa, y0);
        }
        // The matrix instructions of block it + 1 go BETWEEN the halfband stages of this block.  Nothing but data dependences keeps
        // them there (the compiler sinks and hoists across everything else, and a wave issues in order: back to back, every
        // dependent matrix instruction would hold the wave for its 64 cycles), so the order is spelled out as dependences through
        // empty asm statements: a step's A operand "depends" on everything the stage in front of it wrote, and the inputs of the stage
        // behind it "depend" on the step's accumulator.
        // the A values of block it + 1 at once, then all the requests for block it + 2 (a full block ahead of their use, one wait)
        float av[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) av[p] = __builtin_fmaf(ca, ld_use[2 * p], cb * ld_use[2 * p + 1]);
        fetch_all(it + 2, edgec, ld_next);
        voff += 64u * S;
        // The block's store goes out right BEHIND those requests, not at the block's end: loads and stores retire through ONE in-order
        // counter, so the wait for a block's samples is also a wait for every store issued in front of their requests.  At the block's
        // end the store sat one block-time in front of the wait it blocks and cost 440 of the block's 1540 clocks (with either the
        // fetches or the stores switched off the block took 1100); here it is younger than the requests the next wait is for and has
        // two block-times to land.  (Its data is a sixteenth of the OTHER tile: it does not depend on this block's arithmetic.)
        // (not in the edge blocks: behind their branching fetch the compiler drains every request in front of the store -- there it stays at the end)
        if constexpr ((DBG & 8) == 0 && kEarlyStore && !EDGE) store_sixteenth(sv);
        v16f_t acc = zero16;
        // Eight places between the stages take the block's NP matrix instructions, ceil-spread (NP = 4: places 0 2 4 6; NP = 12: two,
        // one, two, one ...); `after` ties the first one of a place to what the stage in front of it wrote
        auto matrix_slot = [&](auto slotc, auto after) {
            constexpr int i = decltype(slotc)::value, p_lo = (i * NP + 7) / 8, p_hi = ((i + 1) * NP + 7) / 8;
            if constexpr (p_lo < p_hi) after(av[p_lo]);
#pragma unroll
            for (int p = p_lo; p < p_hi; p++) {
                if constexpr ((DBG & 2) != 0 && !EDGE) acc[p] += av[p] * bco[p];  // (timing experiment: no matrix instructions in the plain blocks)
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], bco[p], acc, 0, 0, 0);
            }
        };
        // (an instance that is given the whole register file keeps its accumulator in the AGPRs: a pin that asked for it in VGPRs cost
        // sixteen v_accvgpr_read each)
        auto behind_matrix = [&](float2 &x) {
            if constexpr (MINW == 1) asm volatile("" : "+v"(x.x) : "a"(acc));
            else asm volatile("" : "+v"(x.x) : "v"(acc));
        };
        matrix_slot(std::integral_constant<int, 0>{}, [&](float &l) {
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("" : "+v"(l) : "v"(y0[k].x));
        });
        if constexpr (EDGE) {
            // first-stage outputs in front of the call's start come from the previous call's history: every warm-up block of a chunk
            // that starts there when that call left no running sums (it took another route), else only block o = 0 (its outputs 0..6;
            // output 7 is the call's first).  Then the request for the next block's
            const bool h7 = o < 0, h = o <= 0;
#pragma unroll
            for (int k = 0; k < 7; k++) y0[k] = h ? yq[k] : y0[k];
            y0[7] = h7 ? yq[7] : y0[7];
#pragma unroll
            for (int k = 0; k < 8; k++) yq[k] = yh[j0 + 8 + k];  // (the next block's)
        }
        // halfband 1 (transposed form, kernels_fused_dec.h): outputs 0..3 then 4..7 of the block
        float2 y1[4];
        auto hb1 = [&](int k) {
            if (k & 1) {
                const int b = (k - 1) / 2;
#pragma unroll
                for (int t = 0; t < H1; t++) {
                    const float2 v = y0[k] * bc(hb_tap<T1>(T1 - 1 - 2 * t));
                    a1[b + t] = t == H1 - 1 ? v : a1[b + t] + v;
                }
                y1[b] = a1[b];
            } else {
                a1[(k + PC1 - 1) / 2] = a1[(k + PC1 - 1) / 2] + y0[k] * bc(hb_tap<T1>(PC1));
            }
        };
        auto after_a1 = [&](float &l) {
#pragma unroll
            for (int i = 0; i < N1; i++) asm volatile("" : "+v"(l) : "v"(a1[i].x));
        };
#pragma unroll
        for (int k = 0; k < 8; k++) behind_matrix(y0[k]);
        hb1(0); hb1(1);
        matrix_slot(std::integral_constant<int, 1>{}, after_a1);
        if constexpr ((1 * NP + 7) / 8 < (2 * NP + 7) / 8) {
#pragma unroll
            for (int k = 2; k < 8; k++) behind_matrix(y0[k]);
        }
        hb1(2); hb1(3);
        matrix_slot(std::integral_constant<int, 2>{}, after_a1);
        if constexpr ((2 * NP + 7) / 8 < (3 * NP + 7) / 8) {
#pragma unroll
            for (int k = 4; k < 8; k++) behind_matrix(y0[k]);
        }
        hb1(4); hb1(5);
        matrix_slot(std::integral_constant<int, 3>{}, after_a1);
        if constexpr ((3 * NP + 7) / 8 < (4 * NP + 7) / 8) { behind_matrix(y0[6]); behind_matrix(y0[7]); }
        hb1(6); hb1(7);
        matrix_slot(std::integral_constant<int, 4>{}, after_a1);
        // halfband 2
#pragma unroll
        for (int b = 0; b < 4; b++) behind_matrix(y1[b]);
        float2 y2[2];
        auto hb2 = [&](int b) {
            if (b & 1) {
                const int b2 = (b - 1) / 2;
#pragma unroll
                for (int t = 0; t < H2; t++) {
                    const float2 v = y1[b] * bc(hb_tap<T2>(T2 - 1 - 2 * t));
                    a2[b2 + t] = t == H2 - 1 ? v : a2[b2 + t] + v;
                }
                y2[b2] = a2[b2];
            } else {
                a2[(b + PC2 - 1) / 2] = a2[(b + PC2 - 1) / 2] + y1[b] * bc(hb_tap<T2>(PC2));
            }
        };
        auto after_a2 = [&](float &l) {
#pragma unroll
            for (int i = 0; i < N2; i++) asm volatile("" : "+v"(l) : "v"(a2[i].x));
        };
        if constexpr ((DBG & 4) != 0 && !EDGE) {  // (timing experiment: no halfband 2)
            y2[0] = y1[1]; y2[1] = y1[3];
        } else {
            hb2(0); hb2(1);
        }
        matrix_slot(std::integral_constant<int, 5>{}, after_a2);
        if constexpr ((5 * NP + 7) / 8 < (6 * NP + 7) / 8) { behind_matrix(y1[2]); behind_matrix(y1[3]); }
        if constexpr ((DBG & 4) == 0 || EDGE) { hb2(2); hb2(3); }
        matrix_slot(std::integral_constant<int, 6>{}, after_a2);
        behind_matrix(y2[0]);
        behind_matrix(y2[1]);
        // halfband 3: the centre tap's input, then the last place, then the even taps' input
        a3[(PC3 - 1) / 2] = a3[(PC3 - 1) / 2] + y2[0] * bc(hb_tap<T3>(PC3));
        matrix_slot(std::integral_constant<int, 7>{}, [&](float &l) { asm volatile("" : "+v"(l) : "v"(a3[(PC3 - 1) / 2].x)); });
        if constexpr ((7 * NP + 7) / 8 < (8 * NP + 7) / 8) behind_matrix(y2[1]);
        D = acc;
#pragma unroll
        for (int t = 0; t < H3; t++) {
            const float2 v = y2[1] * bc(hb_tap<T3>(T3 - 1 - 2 * t));
            a3[t] = t == H3 - 1 ? v : a3[t] + v;
        }
        const float2 y3 = a3[0];
        // the sums still pending move up by 4 / 2 / 1 places (renamed, not moved: every one of them is written in every block)
#pragma unroll
        for (int i = 0; i + 4 < N1; i++) a1[i] = a1[i + 4];
#pragma unroll
        for (int i = 0; i + 2 < N2; i++) a2[i] = a2[i + 2];
#pragma unroll
        for (int i = 0; i + 1 < N3; i++) a3[i] = a3[i + 1];
        if constexpr ((DBG & 8) == 0 || EDGE) {
            park(cscale(y3, P.gain));  // (the warm-up blocks' values land in rows that are rewritten before their tile is stored)
            if constexpr (!kEarlyStore || EDGE) store_sixteenth(sv);
        } else {
            asm volatile("" :: "v"(y3.x), "v"(y3.y));  // (timing experiment: no tile, no store)
        }
    };

    pa_blk = osc_exact(8 * (o0 - WARM - 1) - 7);  // (the block's in front of block 0: every block but a round's first opens with one rotation)
    // A store that is dropped (out-of-range offset), so that the loops are ENTERED as they are re-entered: with a store behind the sample
    // requests in flight.  The compiler merges the two states at a loop's head and waits for the worse one -- without this the wait for
    // the last requested sample was a wait for everything, the previous block's store included, in every block of the instances that store
    // at a block's end
    if constexpr (!kEarlyStore) __builtin_amdgcn_raw_buffer_store_b64(v2f_t{0.f, 0.f}, orsrc, kNoStore, 0, 0);
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    int it = 0;
    float2 none[8];
    if (pair == 0 && (DBG & 32) == 0) {  // (uniform; DBG 32: timing experiment without the edge blocks)
        // Edge blocks: all blocks up to the last straddling one when the chunks warm up from the first-stage history; with the
        // running sums restored, only the blocks around the restore -- the warm-up blocks in front run plain and what they leave in a
        // chunk that starts at the call's start is overwritten
        int e0 = 0, e_lo = WARM + 1;  // (the last block that needs the edge variant is o = 0: the straddling windows of blocks o = 0 .. o_safe - 1 are requested two blocks earlier)
        if (e_lo > nb) e_lo = nb;
        if (P.state_in != nullptr) {
            e0 = (L < WARM ? WARM - L : WARM) - 2;  // (L = 16: chunk 1 starts in front of the call as well)
            e0 = e0 < 0 ? 0 : e0;
        }
        for (; it < e0; it++) block(it, std::false_type{}, none, ld, ld);
        float2 yq[8];
#pragma unroll
        for (int k = 0; k < 8; k++) yq[k] = yh[8 * (o0 - WARM + it) - 7 + k];
        for (; it < e_lo; it++) {
            if (P.state_in != nullptr && (it == WARM || it == WARM - L)) {
                // the running sums where the previous call ended, in the lanes whose chunk reaches the call's start with this block
                const float2 *st = P.state_in + (long long)c * (N1 + N2 + N3);
                const bool mine = o0 - WARM + it == 0;
#pragma unroll
                for (int i = 0; i < N1; i++) { const float2 v = st[i]; a1[i] = mine ? v : a1[i]; }
#pragma unroll
                for (int i = 0; i < N2; i++) { const float2 v = st[N1 + i]; a2[i] = mine ? v : a2[i]; }
#pragma unroll
                for (int i = 0; i < N3; i++) { const float2 v = st[N1 + N2 + i]; a3[i] = mine ? v : a3[i]; }
            }
            block(it, std::true_type{}, yq, ld, ld);
        }
    }
    // (two blocks per turn with the two sample sets exchanged: from one set the compiler fetched into fixed registers and copied them at
    // the top of every block)
    for (; it + 1 < nb; it += 2) {
        block(it, std::false_type{}, none, ld, ld_alt);
        block(it + 1, std::false_type{}, none, ld_alt, ld);
    }
    if (it < nb) { block(it, std::false_type{}, none, ld, ld); it++; }
    // the running sums where this call ends, for the next call's chunk 0 (L divides n_out: the last chunk is a whole one)
    if (P.state_out != nullptr && 2 * pair + 1 >= P.n_chunks - 1 && chunk_raw == P.n_chunks - 1 && ch_raw < P.n_chan) {
        float2 *st = P.state_out + (long long)c * (N1 + N2 + N3);
#pragma unroll
        for (int i = 0; i < N1; i++) st[i] = a1[i];
#pragma unroll
        for (int i = 0; i < N2; i++) st[N1 + i] = a2[i];
#pragma unroll
        for (int i = 0; i < N3; i++) st[N1 + N2 + i] = a3[i];
    }
    // the last tile
    {
        wave_sync();
        const bool odd = (((L - 1) >> 4) & 1) != 0;
        trd = odd ? lds_rd1 : lds_rd0;
        so_run = so_grp + 8u * (unsigned)((L - 1) & ~15);
        vo_cur = ovoff;
        row_run = st_row;
        for (int g = 0; g < 16; g++) {
            const v2f_t sv = *lds_at(trd);
            trd += 16u;
            unsigned vo = vo_cur;
            if (ragged) {
                vo = row_run < ch_lim ? vo : kNoStore;
                row_run += 2;
            }
            __builtin_amdgcn_raw_buffer_store_b64(v2f_t{sv.x, sv.y}, orsrc, vo, (int)so_run, kStoreAux);
            so_run += so_row2;
        }
    }
    if (P.clk != nullptr && lane == 0) {
        unsigned long long *w = P.clk + 4 * ((size_t)blockIdx.x * 4 + wv);
        w[0] = __builtin_amdgcn_s_memtime() - clk0;
        w[1] = __builtin_amdgcn_s_memrealtime() - rt0;
        w[2] = (unsigned long long)nb;
        w[3] = (unsigned long long)pair;
    }
}

}  // namespace pg
