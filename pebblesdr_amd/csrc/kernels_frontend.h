// kernels_frontend.h -- full-rate kernels: mixer, fused mixer + first decimation stage, later
// decimation stages / CFir (one generic strided FIR), history tails.
//
// Data layout in HBM: complex samples are float2, time-contiguous per stream/channel
// ([channel][time], pitch in samples).  Every stage-to-stage buffer keeps `hist` samples of
// head-room in front of its data pointer holding the tail of the previous call, so a consumer indexes
// data[i - hist .. ] without branching (the reference's "[numTaps-1 delay samples | new samples]"
// buffer, pebblelib/decimator.cpp:289-293, kept on the device).
//
// Bound: HBM.  Algorithmic bytes: 8 B read per input sample per stream + 8/stride B written per
// channel for the first stage; later stages (8 + 8/stride) B per their own input sample.
#pragma once
#include "params.h"
#include "tail_refresh.h"

namespace pg {


__device__ __forceinline__ float osc_amp(const float *__restrict__ amp_tab, float a_inf, uint32_t n0, long long i)
{
    unsigned long long k = (unsigned long long)n0 + (unsigned long long)i;
    return k < (unsigned long long)kAmpTab ? amp_tab[k] : a_inf;
}

// Stand-alone mixer (Mixer::processBlock shape): out[c][i] = osc_c(i) * in[s(c)][i].
// grid (ceil(n/(256*4)), C); each work-item does 4 consecutive samples from one accurate phasor.
static __global__ __launch_bounds__(256) void k_mixer(const float2 *__restrict__ in, long long in_pitch, int shared_input,
                                                float2 *__restrict__ out, long long out_pitch, long long n,
                                                const ChanOsc *__restrict__ osc, const float *__restrict__ amp_tab,
                                                float a_inf)
{
    const int c = blockIdx.y;
    const ChanOsc *o = &osc[c];
    const float2 *x = in + (shared_input ? 0 : (long long)c * in_pitch);
    float2 *y = out + (long long)c * out_pitch;
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    if (!o->mix_on) {
        for (int d = 0; d < 4 && i0 + d < n; d++) y[i0 + d] = x[i0 + d];
        return;
    }
    const float2 p0 = cis_cycles(o->phase0 + (double)(i0 + 1) * o->inc);
    for (int d = 0; d < 4 && i0 + d < n; d++) {
        const float2 ph = cmul(p0, o->step[d]);
        const float a = osc_amp(amp_tab, a_inf, o->n0, i0 + d);
        y[i0 + d] = cmul(cscale(ph, a), x[i0 + d]);
    }
}

// Fused mixer + first decimation stage, LDS-tiled.  A workgroup produces 256 consecutive outputs:
//   halfband:  y[o] = sum_p m[o*S + p - (T-1)] * h[p]          (decimator.cpp:637-648, vDSP_zrdesampD)
//   CIC3:      y[o] = .125*(m[oS+1] + m[(o-1)S] + 3*(m[(o-1)S+1] + m[oS]))   (decimator.cpp:727-731)
// with m[i] = osc(i)*x[i] for i >= 0 and, for i < 0, the mixed samples kept from the previous call (`hist`).
// The 256*S + look-back input samples are fetched once with 16-byte coalesced loads, rotated by the oscillator
// on the way in (one exact fp64-phase sincos per lane every fourth sweep, a constant rotation exp(j*2*pi*512*inc)
// in between) and parked in LDS with one pad slot per S samples, so the stride-S tap reads are conflict-free.
// grid (ceil(n_out/256), C); dynamic LDS = pad(256*S + look-back) float2.
static __global__ __launch_bounds__(256) void k_mix_dec1(const float2 *__restrict__ in, long long in_pitch, int shared_input,
                                                          float2 *__restrict__ out, long long out_pitch, long long n_out,
                                                          const ChanOsc *__restrict__ osc, const float2 *__restrict__ hist,
                                                          int hist_pitch, const float *__restrict__ amp_tab, float a_inf,
                                                          FirTaps taps, float2 *__restrict__ hist_out, OscDynInline dyn, int chan_group,
                                                          int n_chan)
{
    HIP_DYNAMIC_SHARED(float2, tile)
    __shared__ float ht[kMaxTaps];  // taps out of the kernel-argument segment: the tap loop must not wait on scalar loads
    const int c = blockIdx.y, t = threadIdx.x;
    const ChanOsc *oc = &osc[c];
    const float2 *x = in + (shared_input ? 0 : (long long)c * in_pitch);
    const float2 *hs = hist + (long long)c * hist_pitch;
    const int S = taps.stride, T = taps.ntaps;
    if (t < kMaxTaps) ht[t] = t < T ? taps.h[t] : 0.f;
    const int ls = __ffs(S) - 1;                      // S is a power of two
    const int H = taps.cic3 ? S : T - 1;              // look-back in input samples (even)
    const long long o0 = (long long)blockIdx.x * 256;
    const int nout = (int)((n_out - o0) < 256 ? (n_out - o0) : 256);
    const long long i0 = o0 * S - H;                  // input index of tile slot 0 (even)
    const int span = nout * S + H;                    // even
    const double phase0 = dyn.use ? dyn.d[c].phase0 : oc->phase0;
    const uint32_t n0 = dyn.use ? dyn.d[c].n0 : oc->n0;
    const bool mix = (dyn.use ? dyn.d[c].mix_on : oc->mix_on) != 0;
    const bool settled = n0 >= (uint32_t)kAmpTab; // amplitude transient over: a_n == sqrt(0.95) to fp32
    if (taps.cic3 && S > 2) {
        // Merged CIC3 at stride S reads only the pair m[oS], m[oS+1] of every S inputs (decimator.cpp:719-737 as merged):
        // fetch and mix just those pairs -- pair p of this tile is input (o0 + p - 1)*S, p = 0 .. nout.  With a shared
        // input one workgroup serves chan_group channels from ONE fetch of the pairs (each pair is 16 bytes of a 128-byte
        // line: per-channel fetches would pull the whole stream through L2 once per channel).
        float4 *pairs = reinterpret_cast<float4 *>(tile);
        const int c0 = blockIdx.y * chan_group;
        const int ng = (n_chan - c0) < chan_group ? (n_chan - c0) : chan_group;
        for (int p = t; p <= nout; p += 256) {
            const long long i = (o0 + p - 1) * (long long)S;
            float4 xx = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i >= 0 && shared_input) xx = *reinterpret_cast<const float4 *>(in + i);
            for (int g = 0; g < ng; g++) {
                const int cg = c0 + g;
                const ChanOsc *og = &osc[cg];
                float2 a, b;
                if (i < 0) {
                    a = hist[(long long)cg * hist_pitch];
                    b = hist[(long long)cg * hist_pitch + 1];
                } else {
                    if (!shared_input) xx = *reinterpret_cast<const float4 *>(in + (long long)cg * in_pitch + i);
                    a = make_float2(xx.x, xx.y);
                    b = make_float2(xx.z, xx.w);
                    const bool gmix = (dyn.use ? dyn.d[cg].mix_on : og->mix_on) != 0;
                    if (gmix) {
                        const double gph0 = dyn.use ? dyn.d[cg].phase0 : og->phase0;
                        const uint32_t gn0 = dyn.use ? dyn.d[cg].n0 : og->n0;
                        const bool gset = gn0 >= (uint32_t)kAmpTab;
                        const float2 ph = cis_cycles(gph0 + (double)(i + 1) * og->inc);
                        const float2 ph1 = cmul(ph, og->step[1]);
                        const float aa = gset ? a_inf : osc_amp(amp_tab, a_inf, gn0, i);
                        const float ab = gset ? a_inf : osc_amp(amp_tab, a_inf, gn0, i + 1);
                        a = cmul(cscale(ph, aa), a);
                        b = cmul(cscale(ph1, ab), b);
                    }
                }
                pairs[g * 258 + p] = make_float4(a.x, a.y, b.x, b.y);
            }
        }
        __syncthreads();
        for (int g = 0; g < ng; g++) {
            const int cg = c0 + g;
            const float4 *pg = pairs + g * 258;
            if (hist_out != nullptr && o0 + nout == n_out && t < 2) {
                const float4 last = pg[nout];  // m[n-S], m[n-S+1]: the next call's previous pair
                hist_out[(long long)cg * hist_pitch + t] = t == 0 ? make_float2(last.x, last.y) : make_float2(last.z, last.w);
            }
            if (t < nout) {
                const float4 pv = pg[t], cu = pg[t + 1];
                float2 acc;
                acc.x = .125f * (cu.z + pv.x + 3.0f * (pv.z + cu.x));
                acc.y = .125f * (cu.w + pv.y + 3.0f * (pv.w + cu.y));
                out[(long long)cg * out_pitch + o0 + t] = cscale(acc, taps.gain);
            }
        }
        return;
    }
    float2 ph = make_float2(1.f, 0.f);
    int sweep = 0;
    // Loads are issued kLoadBatch at a time before any of them is consumed: with one 16-byte load in flight per
    // work-item the kernel sat at ~4 TB/s (latency x bytes in flight), well under the copy ceiling.
    constexpr int kLoadBatch = 5;
    for (int vb = t; vb < span / 2; vb += 256 * kLoadBatch) {
        float4 xx[kLoadBatch];
#pragma unroll
        for (int k = 0; k < kLoadBatch; k++) {
            const int v = vb + 256 * k;
            const long long i = i0 + 2 * v;
            xx[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < span / 2 && i >= 0) xx[k] = *reinterpret_cast<const float4 *>(x + i);
        }
#pragma unroll
        for (int k = 0; k < kLoadBatch; k++) {
            const int v = vb + 256 * k;
            if (v < span / 2) {
                const long long i = i0 + 2 * v;
                float2 a, b;
                if (i < 0) {
                    if (taps.cic3) {  // only m[-S], m[-S+1] are ever read (hist[0], hist[1])
                        a = i == -S ? hs[0] : make_float2(0.f, 0.f);
                        b = i == -S ? hs[1] : make_float2(0.f, 0.f);
                    } else {
                        a = hs[H + i];
                        b = hs[H + i + 1];
                    }
                    sweep = -1;  // next sweep must take a fresh phase
                } else {
                    a = make_float2(xx[k].x, xx[k].y);
                    b = make_float2(xx[k].z, xx[k].w);
                    if (mix) {
                        if ((sweep & 3) == 0) ph = cis_cycles(phase0 + (double)(i + 1) * oc->inc);
                        else ph = cmul(ph, oc->step512);
                        const float2 ph1 = cmul(ph, oc->step[1]);
                        const float aa = settled ? a_inf : osc_amp(amp_tab, a_inf, n0, i);
                        const float ab = settled ? a_inf : osc_amp(amp_tab, a_inf, n0, i + 1);
                        a = cmul(cscale(ph, aa), a);
                        b = cmul(cscale(ph1, ab), b);
                    }
                }
                const int s0 = 2 * v;
                tile[s0 + (s0 >> ls)] = a;
                tile[s0 + 1 + ((s0 + 1) >> ls)] = b;
                sweep++;
            }
        }
    }
    __syncthreads();
    // The block that reaches the end of the call leaves the mixed-sample history for the next one in the OTHER history
    // buffer (block 0 of this launch may still be reading `hist`): hist_out[c][j] = m[n - H + j], the tile's last H slots
    // (for CIC3 the pair m[n-S], m[n-S+1]).
    if (hist_out != nullptr && o0 + nout == n_out) {
        const int Hh = taps.cic3 ? 2 : T - 1;
        if (t < Hh) {
            const int sl = taps.cic3 ? span - S + t : span - Hh + t;
            hist_out[(long long)c * hist_pitch + t] = tile[sl + (sl >> ls)];
        }
    }
    if (t >= nout) return;
    float2 acc;
    if (taps.cic3) {
        const int e = t * S;  // previous pair at e, e+1; current pair at e+S, e+S+1
        const float2 pev = tile[e + (e >> ls)], pod = tile[e + 1 + ((e + 1) >> ls)];
        const float2 ev = tile[e + S + ((e + S) >> ls)], od = tile[e + S + 1 + ((e + S + 1) >> ls)];
        acc.x = .125f * (od.x + pev.x + 3.0f * (pod.x + ev.x));
        acc.y = .125f * (od.y + pev.y + 3.0f * (pod.y + ev.y));
    } else {
        // halfband (T = 4k+3): non-zero taps sit at even p and at the odd centre (T-1)/2
        const int e = t * S, cc = (T - 1) >> 1;
        const float2 mc = tile[e + cc + ((e + cc) >> ls)];
        acc = cscale(mc, ht[cc]);
        for (int p = 0; p < T; p += 2) {
            const float h = ht[p];
            const float2 m = tile[e + p + ((e + p) >> ls)];
            acc.x = fmaf(m.x, h, acc.x);
            acc.y = fmaf(m.y, h, acc.y);
        }
    }
    out[(long long)c * out_pitch + o0 + t] = cscale(acc, taps.gain);
}

// All decimation stages after the first, fused: a workgroup produces `outb` final outputs, walking the
// stages through two LDS ping-pong buffers (y_s[j] = sum_p y_{s-1}[j*S_s + p - (T_s-1)] * h_s[p]).  `in` points at the
// first stage-0 output of this call; what lies before it (head-room) is the previous call's tail, deep enough for
// the whole cascade: halo0 = sum_s (T_s - 1) * prod_{r<s} S_r.
// grid (ceil(n_out/outb), C); dynamic LDS = (count_0 + count_1) float2.
// F0, F1, F2 > 0: a chain of exactly three stride-2 stages with these tap counts (the one-channel WFM chain hb15, hb23, hb47 runs
// behind the display transform on an idle GPU, so its length is the call's): the tap loops unroll and the taps come straight out
// of the kernel arguments as scalars -- the generic form reads every tap from LDS next to every sample, and its three stages took
// 13 of the kernel's 24 us.  F0 = 0: any chain.
template <int F0, int F1, int F2>
static __global__ __launch_bounds__(256) void k_cascade(const float2 *__restrict__ in, long long in_pitch,
                                                         float2 *__restrict__ out, long long out_pitch, long long n_out,
                                                         CascadeParams cp)
{
    HIP_DYNAMIC_SHARED(float2, buf)
    __shared__ float ht[kMaxCascade][64];  // taps out of the kernel-argument segment
    const int c = blockIdx.y, t = threadIdx.x;
    for (int i = t; i < kMaxCascade * 64; i += 256) {
        const int s = i >> 6, p = i & 63;
        ht[s][p] = (s < cp.nst && p < cp.ntaps[s]) ? cp.h[s][p] : 0.f;
    }
    const long long o0 = (long long)blockIdx.x * cp.outb;
    const int nfin = (int)((n_out - o0) < cp.outb ? (n_out - o0) : cp.outb);
    int cnt[kMaxCascade + 1];
    long long first = o0;
#pragma unroll
    for (int s = kMaxCascade; s >= 0; s--) cnt[s] = 0;
#pragma unroll
    for (int s = kMaxCascade; s >= 1; s--) {
        if (s == cp.nst) cnt[s] = nfin;
        if (s <= cp.nst) {
            cnt[s - 1] = cnt[s] * cp.stride[s - 1] + cp.ntaps[s - 1] - 1;
            first = first * cp.stride[s - 1] - (cp.ntaps[s - 1] - 1);
        }
    }
    const float2 *x = in + (long long)c * in_pitch + first;
    float2 *src = buf, *dst = buf + cp.lds_half;
    // all of this tile's loads in flight together (as in k_mix_dec1): batches of kBatch, issued before any LDS store
    // (16-byte lanes: `first`, the row pitch and cnt[0] are all even, so the tile is a whole number of aligned sample pairs)
    {
        constexpr int kBatch = 5;
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        float4 *s4 = reinterpret_cast<float4 *>(src);
        const int n4 = cnt[0] >> 1;
        for (int jb = t; jb < n4; jb += 256 * kBatch) {
            float4 v[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                const int j = jb + 256 * k;
                v[k] = j < n4 ? x4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                const int j = jb + 256 * k;
                if (j < n4) s4[j] = v[k];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kMaxCascade; s++) {
        if (s < cp.nst) {
            const int S = cp.stride[s], T = cp.ntaps[s], ns = cnt[s + 1];
            const bool last = s == cp.nst - 1;
            const int cc = (T - 1) >> 1;  // halfband: even taps + the odd centre
            if (F0 > 0 && s < 3) {
                constexpr int TF[3] = {F0 > 0 ? F0 : 3, F1 > 0 ? F1 : 3, F2 > 0 ? F2 : 3};
                auto fixed = [&](auto tc) {
                    constexpr int TT = decltype(tc)::value, CC = (TT - 1) >> 1;
                    for (int j = t; j < ns; j += 256) {
                        const float2 *w = src + j * 2;
                        float2 acc = cscale(w[CC], cp.h[s][CC]);
#pragma unroll
                        for (int p = 0; p < TT; p += 2) {
                            const float h = cp.h[s][p];
                            const float2 m = w[p];
                            acc.x = fmaf(m.x, h, acc.x);
                            acc.y = fmaf(m.y, h, acc.y);
                        }
                        if (last) out[(long long)c * out_pitch + o0 + j] = cscale(acc, cp.gain);
                        else dst[j] = acc;
                    }
                };
                if (s == 0) fixed(std::integral_constant<int, TF[0]>{});
                else if (s == 1) fixed(std::integral_constant<int, TF[1]>{});
                else fixed(std::integral_constant<int, TF[2]>{});
            } else
            for (int j = t; j < ns; j += 256) {
                const float2 *w = src + j * S;
                float2 acc = cscale(w[cc], ht[s][cc]);
                for (int p = 0; p < T; p += 2) {
                    const float h = ht[s][p];
                    const float2 m = w[p];
                    acc.x = fmaf(m.x, h, acc.x);
                    acc.y = fmaf(m.y, h, acc.y);
                }
                if (last) out[(long long)c * out_pitch + o0 + j] = cscale(acc, cp.gain);
                else dst[j] = acc;
            }
            __syncthreads();
            float2 *tmp = src;
            src = dst;
            dst = tmp;
        }
    }
}

// Mixer + merged CIC3 (stride S0) + wide halfband (hb11, stride S1 >= 8) in one pass -- the front of every chain the
// reference builds for narrow channels at >= 5 Msps and for WFM at >= 100 Msps (cic3 x S0, hb11 x 16, ...).  Output j of
// the halfband reads CIC outputs S1 j - (T1-1) .. S1 j; CIC output k reads the sample pairs k-1 and k (pair P = samples
// P S0, P S0 + 1): a window of T1 + 1 pairs out of every S1.  Only those pairs are fetched and mixed, nothing at the CIC
// rate is written -- the unfused route writes and re-reads 8 B per CIC output per channel, which is what a wide bank is
// bound by.
//   One work-item owns whole outputs of one channel: it fetches its window's 12 pairs (16-byte loads), mixes them with
//   an oscillator phase it rotates pair to pair in registers, and runs the CIC3 + halfband arithmetic on the spot: no
//   LDS, no barrier, so waves are limited by registers only.  Lanes of a wave are `CL` channels x 64/CL outputs; with a
//   shared input stream all channels of an output load the same address (one line per load instruction).
//   hist: [channel][2 (T1 + 1)] the previous call's last T1 + 1 mixed pairs (pairs -(T1+1) .. -1 of this call); one
//   more "virtual output" per channel mixes the call's final T1 + 1 pairs into hist_out for the next call.
// grid (ceil((n_out + 1) / (4 R 64/CL)), ceil(C / CL)), block 256 (four independent waves, R outputs per lane each).
// Row stores for the lanes-are-channels kernels below.  A lane finishing output j of channel c would store 8 bytes into
// row c: 64 lanes = 64 different lines per store instruction, and the write path, not the arithmetic, set the pace
// (33 M such stores took 0.25 ms).  Instead each wave parks its R x 64 results in a private LDS tile [output][channel]
// (row pad 1: both the lane-wise writes and the transposed reads are conflict-free) and then writes whole row segments:
// consecutive lanes = consecutive outputs of one channel, R*OL*8 bytes per channel (a full 128-byte line at R*OL = 16).
__host__ __device__ inline int front_tile_slots(int cl_log2, int R) { return R * (64 >> cl_log2) * ((1 << cl_log2) + 1); }
__device__ __forceinline__ void front_store_rows(const float2 *__restrict__ tile, float2 *__restrict__ out, long long out_pitch, long long n_out,
                                                 int cbase, int n_chan, long long jb, int cl_log2, int R, int lane)
{
    const int CL = 1 << cl_log2, OL = 64 >> cl_log2;
    const int per_chan = R * OL;                 // outputs of one channel in the wave's tile (a power of two)
    const int pl = __ffs(per_chan) - 1;
    wave_sync();
    for (int it = 0; it < R; it++) {
        const int idx = it * 64 + lane;
        const int ch = idx >> pl, jt = idx & (per_chan - 1);
        const long long j = jb + jt;
        if (cbase + ch < n_chan && j < n_out) out[(long long)(cbase + ch) * out_pitch + j] = tile[jt * (CL + 1) + ch];
    }
}

// TRANSIENT: some oscillator of the bank is inside its amplitude transient (the first kAmpTab samples after a reset):
// per-sample amplitudes from the table; otherwise every amplitude is a_inf.  The host picks the variant per call.
// UNIFORM: 64 channels across the lanes and one shared stream -- a wave's lanes then all read the same window, so its
// address is wave-uniform and the loads go through the scalar cache (a vector load hands every lane its own copy: 16
// cycles of L1 return path per 16-byte load whatever the addresses, which was ~40 % of this kernel's time).
template <bool TRANSIENT, bool UNIFORM>
static __global__ __launch_bounds__(256) void k_mix_cic_hb(const float2 *__restrict__ in, long long in_pitch, int shared_input,
                                                            float2 *__restrict__ out, long long out_pitch, long long n_out,
                                                            const ChanOsc *__restrict__ osc, const float2 *__restrict__ hist,
                                                            float2 *__restrict__ hist_out, int hist_pitch, const float *__restrict__ amp_tab,
                                                            float a_inf, FrontTaps hb /* stage 1: hb11 */, int S0, float out_gain, OscDynInline dyn,
                                                            int cl_log2, int n_chan, int R)
{
    constexpr int T1 = kFrontT1, NP = T1 + 1;
    HIP_DYNAMIC_SHARED(float2, tiles)
    if (UNIFORM) cl_log2 = 6;
    const int lane = threadIdx.x & 63;
    const int wv = UNIFORM ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
    const int CL = 1 << cl_log2, OL = 64 >> cl_log2;
    const int cl = lane & (CL - 1), ol = lane >> cl_log2;
    const int cbase = blockIdx.y * CL;
    const bool live = cbase + cl < n_chan;       // idle lanes still help with the row stores
    const int c = live ? cbase + cl : n_chan - 1;
    float2 *tile = tiles + wv * front_tile_slots(cl_log2, R);
    const int S1 = hb.stride;
    const ChanOsc *oc = &osc[c];
    const double inc = oc->inc;
    double phase0 = oc->phase0;
    uint32_t n0 = oc->n0, mix_on = oc->mix_on;
    if (dyn.use) {  // per-call oscillator state rides in the kernel arguments (scalar registers): select, don't index
#pragma unroll
        for (int k = 0; k < kOscInline; k++)
            if (c == k) { phase0 = dyn.d[k].phase0; n0 = dyn.d[k].n0; mix_on = dyn.d[k].mix_on; }
    }
    const float2 step1 = oc->step[1];
    const float2 rot = cis_cycles((double)S0 * inc);  // pair to pair
    const float2 *in_c = (UNIFORM || shared_input) ? in : in + (long long)c * in_pitch;
    const float2 *hist_c = hist + (long long)c * hist_pitch;
    // outputs 0 .. n_out-1, plus the virtual output n_out: the call's last NP pairs, mixed into hist_out for the next call
    const long long n_work = hist_out != nullptr ? n_out + 1 : n_out;
    const long long jb = ((long long)blockIdx.x * 4 + wv) * R * OL;
    auto first_pair = [&](long long jj) { return jj == n_out ? (long long)S1 * n_out - NP : (long long)S1 * jj - T1; };
    // all of a window's loads are issued together; pairs before the call's start are clamped here and patched from `hist`
    auto load_window = [&](long long P0, float4 (&w)[NP]) {
        if (P0 >= 0) {  // one 64-bit address, twelve 32-bit offsets
            const float2 *p = in_c + P0 * (long long)S0;
#pragma unroll
            for (int q = 0; q < NP; q++) w[q] = *reinterpret_cast<const float4 *>(p + q * S0);
        } else {
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const long long P = P0 + q > 0 ? P0 + q : 0;
                w[q] = *reinterpret_cast<const float4 *>(in_c + P * (long long)S0);
            }
        }
    };
    const float2 rot_out = cis_cycles((double)S0 * (double)(S1 * OL) * inc);  // this lane's output to its next one
    float2 ph0 = make_float2(1.f, 0.f);
    for (int r = 0; r < R; r++) {
        const long long j = jb + (long long)r * OL + ol;
        if (!live || j >= n_work) continue;
        const bool virt = j == n_out;
        const long long P0 = first_pair(j);
        float4 m[NP];
        load_window(P0, m);
        if (mix_on != 0) {
            // exact fp64 phase at the lane's first output, a constant rotation from one output to the next (the virtual
            // window sits one pair off that grid: exact phase again)
            ph0 = (r == 0 || virt) ? cis_cycles(phase0 + (double)(P0 * (long long)S0 + 1) * inc) : cmul(rot_out, ph0);
            float2 ph = ph0;
#pragma unroll
            for (int q = 0; q < NP; q++) {
                float aa = a_inf, ab = a_inf;
                if (TRANSIENT) {
                    const long long i = (P0 + q > 0 ? P0 + q : 0) * (long long)S0;
                    aa = osc_amp(amp_tab, a_inf, n0, i);
                    ab = osc_amp(amp_tab, a_inf, n0, i + 1);
                }
                const float2 a = cmul(cscale(ph, aa), make_float2(m[q].x, m[q].y));
                const float2 b = cmul(cscale(cmul(step1, ph), ab), make_float2(m[q].z, m[q].w));
                m[q] = make_float4(a.x, a.y, b.x, b.y);
                ph = cmul(rot, ph);
            }
        }
        if (P0 < 0) {  // the first output of a call: its leading pairs are the previous call's, already mixed
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const int P = (int)P0 + q;
                if (P < 0) {
                    const float2 a = hist_c[2 * (NP + P)], b = hist_c[2 * (NP + P) + 1];
                    m[q] = make_float4(a.x, a.y, b.x, b.y);
                }
            }
        }
        if (virt) {
            float2 *hp = hist_out + (long long)c * hist_pitch;
#pragma unroll
            for (int q = 0; q < NP; q++) {
                hp[2 * q] = make_float2(m[q].x, m[q].y);
                hp[2 * q + 1] = make_float2(m[q].z, m[q].w);
            }
            continue;
        }
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int q = 1; q <= T1; q++) {
            // CIC3 output k = S1 j - T1 + q from pairs k-1 and k: .125 (od + pev + 3 (pod + ev))
            const float4 pv = m[q - 1], cu = m[q];
            const float cx = .125f * (cu.z + pv.x + 3.0f * (pv.z + cu.x));
            const float cy = .125f * (cu.w + pv.y + 3.0f * (pv.w + cu.y));
            acc.x = fmaf(cx, hb.h[q - 1], acc.x);
            acc.y = fmaf(cy, hb.h[q - 1], acc.y);
        }
        if (cl_log2 == 0) out[(long long)c * out_pitch + j] = cscale(acc, out_gain);  // lanes = consecutive outputs of one channel
        else tile[(r * OL + ol) * (CL + 1) + cl] = cscale(acc, out_gain);
    }
    if (cl_log2 != 0) front_store_rows(tile, out, out_pitch, n_out, cbase, n_chan, jb, cl_log2, R, lane);
}

// Mixer + first halfband stage (hb11, stride S) for a BANK of channels tuned off one shared stream -- the register form of
// k_mix_dec1.  There a workgroup mixes one channel's tile into LDS: every channel re-reads the stream through L2 and
// pays a load -> mix -> barrier -> FIR latency chain per 256 outputs, which is what a wide bank was bound by (0.29 ms
// for 256 channels x 0.5 M samples, far off both the HBM and the VALU floor).  Here the lanes of a wave are 64 channels
// of the bank and a work-item owns whole outputs of its channel: the window's six 16-byte loads are the same address in
// every lane (one line per load instruction for 64 channels), the seven samples a halfband touches (even taps + centre)
// are mixed with phases ph(i0) * step[d] from the channel's step table, the oscillator is rotated output to output in
// registers (exact fp64 phase once per work-item), and the FIR is seven FMAs: no LDS, no barrier.
//   y[o] = sum_p m[o S + p - 10] h[p]  (decimator.cpp:637-648), m = osc * x; m[i < 0] from `hist` ([channel][10] mixed)
//   the virtual output o = n_out mixes the call's last ten samples into hist_out for the next call.
// grid (ceil((n_out + 1) / (4 R 64/CL)), ceil(C / CL)), block 256 (four independent waves, R outputs per lane each).
template <bool TRANSIENT, bool UNIFORM /* as in k_mix_cic_hb */, bool RAW = false /* raw device-format input (RawSrc), one stream */>
static __global__ __launch_bounds__(256) void k_mix_hb11_bank(const float2 *__restrict__ in, long long in_pitch, int shared_input,
                                                               float2 *__restrict__ out, long long out_pitch, long long n_out,
                                                               const ChanOsc *__restrict__ osc, const float2 *__restrict__ hist,
                                                               float2 *__restrict__ hist_out, int hist_pitch, const float *__restrict__ amp_tab,
                                                               float a_inf, FrontTaps hb /* stage 0: hb11 */, float out_gain, OscDynInline dyn,
                                                               int cl_log2, int n_chan, int R,
                                                               long long edges_below /* >= 0: only outputs j < edges_below and the history */,
                                                               RawSrc raw = RawSrc{nullptr, 0, 0, 0.f, 0})
{
    constexpr int T = kFrontT1, H = T - 1;
    HIP_DYNAMIC_SHARED(float2, tiles)
    if (UNIFORM) cl_log2 = 6;
    const int lane = threadIdx.x & 63;
    const int wv = UNIFORM ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
    const int CL = 1 << cl_log2, OL = 64 >> cl_log2;
    const int cl = lane & (CL - 1), ol = lane >> cl_log2;
    const int cbase = blockIdx.y * CL;
    const bool live = cbase + cl < n_chan;
    const int c = live ? cbase + cl : n_chan - 1;
    float2 *tile = tiles + wv * front_tile_slots(cl_log2, R);
    const int S = hb.stride;
    const ChanOsc *oc = &osc[c];
    const double inc = oc->inc;
    double phase0 = oc->phase0;
    uint32_t n0 = oc->n0, mix_on = oc->mix_on;
    if (dyn.use) {
#pragma unroll
        for (int k = 0; k < kOscInline; k++)
            if (c == k) { phase0 = dyn.d[k].phase0; n0 = dyn.d[k].n0; mix_on = dyn.d[k].mix_on; }
    }
    // the window's used samples sit at offsets 0 2 4 5 6 8 10 from its first sample
    const float2 st2 = oc->step[2], st4 = oc->step[4], st5 = oc->step[5], st6 = oc->step[6], st8 = oc->step[8], st10 = oc->step[10];
    const float2 rot = cis_cycles((double)(S * OL) * inc);  // this lane's output to its next one
    const float2 *in_c = (UNIFORM || shared_input) ? in : in + (long long)c * in_pitch;
    const float2 *hist_c = hist + (long long)c * hist_pitch;
    const long long n_work = hist_out != nullptr ? n_out + 1 : n_out;
    // edges only: a two-block launch, block 0 for the first outputs and the block that holds output n_out for the history
    const long long bx = edges_below >= 0 && blockIdx.x != 0 ? n_out / (4LL * R * OL) : (long long)blockIdx.x;
    const long long jb = (bx * 4 + wv) * R * OL;
    float2 ph = make_float2(1.f, 0.f);
    // the six 16-byte loads of a window; pairs before the call's start are clamped here and patched from `hist` below
    auto load_window = [&](long long jj, float4 (&w)[6]) {
        const long long i0 = (long long)S * jj - H;
        if (RAW) {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const long long i = i0 + 2 * k > 0 ? i0 + 2 * k : 0;
                const float2 a = raw_load(raw, i), b = raw_load(raw, i + 1);
                w[k] = make_float4(a.x, a.y, b.x, b.y);
            }
        } else if (i0 >= 0) {  // one address, six immediate offsets
            const float4 *p = reinterpret_cast<const float4 *>(in_c + i0);
#pragma unroll
            for (int k = 0; k < 6; k++) w[k] = p[k];
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const long long i = i0 + 2 * k > 0 ? i0 + 2 * k : 0;
                w[k] = *reinterpret_cast<const float4 *>(in_c + i);
            }
        }
    };
    for (int r = 0; r < R; r++) {
        const long long j = jb + (long long)r * OL + ol;
        if (!live || j >= n_work) continue;
        if (edges_below >= 0 && j >= edges_below && j != n_out) continue;  // the rest comes from k_mix_hb11_lean
        if (j == n_out) {  // the next call's history: m[n - 10 .. n - 1], each with its exact phase
            const long long n = n_out * S;
            float2 *hp = hist_out + (long long)c * hist_pitch;
#pragma unroll 1
            for (int q = 0; q < H; q++) {
                const long long i = n - H + q;
                float2 v = RAW ? raw_load(raw, i) : in_c[i];
                if (mix_on != 0) v = cmul(cscale(cis_cycles(phase0 + (double)(i + 1) * inc), osc_amp(amp_tab, a_inf, n0, i)), v);
                hp[q] = v;
            }
            continue;
        }
        const long long i0 = (long long)S * j - H;  // even
        float4 x[6];
        load_window(j, x);
        float2 m0 = make_float2(x[0].x, x[0].y), m2 = make_float2(x[1].x, x[1].y), m4 = make_float2(x[2].x, x[2].y), m5 = make_float2(x[2].z, x[2].w),
               m6 = make_float2(x[3].x, x[3].y), m8 = make_float2(x[4].x, x[4].y), m10 = make_float2(x[5].x, x[5].y);
        if (mix_on != 0) {
            ph = r == 0 ? cis_cycles(phase0 + (double)(i0 + 1) * inc) : cmul(rot, ph);
            if (TRANSIENT) {
                auto amp = [&](int d) { const long long i = i0 + d; return osc_amp(amp_tab, a_inf, n0, i > 0 ? i : 0); };
                m0 = cmul(cscale(ph, amp(0)), m0);
                m2 = cmul(cscale(cmul(st2, ph), amp(2)), m2);
                m4 = cmul(cscale(cmul(st4, ph), amp(4)), m4);
                m5 = cmul(cscale(cmul(st5, ph), amp(5)), m5);
                m6 = cmul(cscale(cmul(st6, ph), amp(6)), m6);
                m8 = cmul(cscale(cmul(st8, ph), amp(8)), m8);
                m10 = cmul(cscale(cmul(st10, ph), amp(10)), m10);
            } else {
                const float2 pa = cscale(ph, a_inf);
                m0 = cmul(pa, m0);
                m2 = cmul(cmul(st2, pa), m2);
                m4 = cmul(cmul(st4, pa), m4);
                m5 = cmul(cmul(st5, pa), m5);
                m6 = cmul(cmul(st6, pa), m6);
                m8 = cmul(cmul(st8, pa), m8);
                m10 = cmul(cmul(st10, pa), m10);
            }
        }
        if (i0 < 0) {  // output 0 of the call (S <= 10 would add output 1): its first ten samples are the previous call's
            if (i0 + 0 < 0) m0 = hist_c[H + (int)i0 + 0];
            if (i0 + 2 < 0) m2 = hist_c[H + (int)i0 + 2];
            if (i0 + 4 < 0) m4 = hist_c[H + (int)i0 + 4];
            if (i0 + 5 < 0) m5 = hist_c[H + (int)i0 + 5];
            if (i0 + 6 < 0) m6 = hist_c[H + (int)i0 + 6];
            if (i0 + 8 < 0) m8 = hist_c[H + (int)i0 + 8];
            if (i0 + 10 < 0) m10 = hist_c[H + (int)i0 + 10];
        }
        float2 acc = cscale(m0, hb.h[0]);
        acc.x = fmaf(m2.x, hb.h[2], acc.x);   acc.y = fmaf(m2.y, hb.h[2], acc.y);
        acc.x = fmaf(m4.x, hb.h[4], acc.x);   acc.y = fmaf(m4.y, hb.h[4], acc.y);
        acc.x = fmaf(m5.x, hb.h[5], acc.x);   acc.y = fmaf(m5.y, hb.h[5], acc.y);
        acc.x = fmaf(m6.x, hb.h[6], acc.x);   acc.y = fmaf(m6.y, hb.h[6], acc.y);
        acc.x = fmaf(m8.x, hb.h[8], acc.x);   acc.y = fmaf(m8.y, hb.h[8], acc.y);
        acc.x = fmaf(m10.x, hb.h[10], acc.x); acc.y = fmaf(m10.y, hb.h[10], acc.y);
        if (cl_log2 == 0) out[(long long)c * out_pitch + j] = cscale(acc, out_gain);  // lanes = consecutive outputs of one channel
        else tile[(r * OL + ol) * (CL + 1) + cl] = cscale(acc, out_gain);
    }
    if (cl_log2 != 0) front_store_rows(tile, out, out_pitch, n_out, cbase, n_chan, jb, cl_log2, R, lane);
}

// The one-channel form of the above that runs BESIDE the spectrum kernel (receiver.hip, two-stream call).  That kernel's
// workgroups hold 4 x 112 of a SIMD's 512 registers per lane and 144 of a CU's 160 KiB of LDS for the whole launch, so a
// neighbour has to live in 64 registers and no LDS or it only gets onto a CU by displacing one of them.  Lean by
// construction: lanes = consecutive outputs (coalesced stores, no transpose tile); only outputs whose window lies inside
// the call (j >= j_first; the first one or two and the mixed history for the next call come from a small launch of
// k_mix_hb11_bank with edges_only); settled oscillator amplitude only (the host sends a call inside the transient to the
// general kernels); and the oscillator factored out of the window, y[j] = gain * pa(j) * sum_d (h[d] step[d]) x[S j - 10 + d]
// with pa(j) = a_inf e^{j 2 pi (phase0 + (S j - 9) inc)}: seven complex MACs and one product per output instead of
// thirteen products and seven MACs.
// grid ceil(n_out / (4 R 64)) + 1 (the last workgroup: the call's edges), block 256 (four independent waves, R outputs per lane each).
template <int FMT /* >= 0: raw device-format input (RawSrc, that sample format) converted in the window loads; -1: float2 `in` */>
static __global__ __launch_bounds__(256, 8) void k_mix_hb11_lean(const float2 *__restrict__ in, float2 *__restrict__ out, long long n_out,
                                                                  const ChanOsc *__restrict__ osc, float a_inf, FrontTaps hb, float out_gain,
                                                                  OscDynInline dyn, int R, long long j_first, RawSrc raw,
                                                                  const float2 *__restrict__ hist, float2 *__restrict__ hist_out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = hb.stride;
    const ChanOsc *oc = &osc[0];
    const double inc = oc->inc;
    const double phase0 = dyn.use ? dyn.d[0].phase0 : oc->phase0;
    const bool mix = (dyn.use ? dyn.d[0].mix_on : oc->mix_on) != 0;
    const bool own_edges = j_first >= 0;  // (A/B: a negative j_first leaves the edges to a launch of k_mix_hb11_bank)
    if (!own_edges) j_first = -j_first;
    if (own_edges && blockIdx.x == gridDim.x - 1) {
        // The launch's last workgroup does what the outputs' lanes leave out -- the first j_first outputs, whose windows reach back into
        // the previous call's mixed samples (`hist`, ten of them), and the ten mixed samples this call leaves for the next (`hist_out`) --
        // with k_mix_hb11_bank's arithmetic (that kernel used to be launched behind this one for it, two workgroups that did not fit
        // beside the display transform's: 60 us on the chain's critical path for a microsecond of work).
        const int t = threadIdx.x;
        if (t < (int)j_first) {
            const long long i0 = (long long)S * t - 10;
            const float2 pa = mix ? cscale(cis_cycles(phase0 + (double)(i0 + 1) * inc), a_inf) : make_float2(1.f, 0.f);
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 7; q++) {
                const int d = q < 3 ? 2 * q : q == 3 ? 5 : 2 * q - 2;  // 0 2 4 5 6 8 10
                const long long i = i0 + d;
                float2 m;
                if (i < 0) {
                    m = hist[10 + i];
                } else {
                    m = FMT >= 0 ? raw_load(raw, i) : in[i];
                    if (mix) m = cmul(d == 0 ? pa : cmul(oc->step[d], pa), m);
                }
                if (q == 0) acc = cscale(m, hb.h[0]);
                else { acc.x = fmaf(m.x, hb.h[d], acc.x); acc.y = fmaf(m.y, hb.h[d], acc.y); }
            }
            out[t] = cscale(acc, out_gain);
        }
        if (t >= 64 && t < 74) {
            const long long i = n_out * S - 10 + (t - 64);
            float2 v = FMT >= 0 ? raw_load(raw, i) : in[i];
            if (mix) v = cmul(cscale(cis_cycles(phase0 + (double)(i + 1) * inc), a_inf), v);
            hist_out[t - 64] = v;
        }
        return;
    }
    float2 c2 = make_float2(hb.h[2], 0.f), c4 = make_float2(hb.h[4], 0.f), c5 = make_float2(hb.h[5], 0.f), c6 = make_float2(hb.h[6], 0.f),
           c8 = make_float2(hb.h[8], 0.f), c10 = make_float2(hb.h[10], 0.f);
    if (mix) {
        c2 = cscale(oc->step[2], hb.h[2]); c4 = cscale(oc->step[4], hb.h[4]); c5 = cscale(oc->step[5], hb.h[5]);
        c6 = cscale(oc->step[6], hb.h[6]); c8 = cscale(oc->step[8], hb.h[8]); c10 = cscale(oc->step[10], hb.h[10]);
    }
    const float h0 = hb.h[0];
    const float2 rot = cis_cycles((double)(S * 64) * inc);  // this lane's output to its next one
    const long long jb = ((long long)blockIdx.x * 4 + wv) * R * 64;
    float2 ph = make_float2(a_inf * out_gain, 0.f);
    for (int r = 0; r < R; r++) {
        const long long j = jb + (long long)r * 64 + lane;
        const long long i0 = (long long)S * j - 10;  // even
        // (before the range test: a lane that sits out its first output must still carry the phase to its later ones)
        if (mix) ph = r == 0 ? cscale(cis_cycles(phase0 + (double)(i0 + 1) * inc), a_inf * out_gain) : cmul(rot, ph);
        if (j < j_first || j >= n_out) continue;
        float2 s0, s2, s4, s5, s6, s8, s10;
        if (FMT >= 0) {
            // the window's 11 samples as three wide loads of four (one 2-byte load per tap had this kernel run twice as long
            // as its float2 form beside the display transform)
            float2 w[12];
            raw_load4_even<FMT>(raw, i0, reinterpret_cast<float2 (&)[4]>(w[0]));
            raw_load4_even<FMT>(raw, i0 + 4, reinterpret_cast<float2 (&)[4]>(w[4]));
            raw_load4_even<FMT>(raw, i0 + 8, reinterpret_cast<float2 (&)[4]>(w[8]));
            s0 = w[0]; s2 = w[2]; s4 = w[4]; s5 = w[5]; s6 = w[6]; s8 = w[8]; s10 = w[10];
        } else {
            const float4 *p = reinterpret_cast<const float4 *>(in + i0);
            const float4 x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3], x4 = p[4], x5 = p[5];
            s0 = make_float2(x0.x, x0.y); s2 = make_float2(x1.x, x1.y); s4 = make_float2(x2.x, x2.y); s5 = make_float2(x2.z, x2.w);
            s6 = make_float2(x3.x, x3.y); s8 = make_float2(x4.x, x4.y); s10 = make_float2(x5.x, x5.y);
        }
        float2 acc = cscale(s0, h0);
        acc = cadd(acc, cmul(c2, s2));
        acc = cadd(acc, cmul(c4, s4));
        acc = cadd(acc, cmul(c5, s5));
        acc = cadd(acc, cmul(c6, s6));
        acc = cadd(acc, cmul(c8, s8));
        acc = cadd(acc, cmul(c10, s10));
        const float2 y = mix ? cmul(ph, acc) : cscale(acc, out_gain);
        // float2 input: past L2 -- beside the display transform the input's lines are worth more there than these, which the cascade reads
        // a hundred microseconds later (-0.9 % per call; with a raw format the input is a quarter of the size and the hint costs 0.9 %)
        if (FMT < 0) store_stream(out + j, y);
        else out[j] = y;
    }
}

// Generic real-tap FIR on complex data with decimation: the CFir post-demod filters (stride 1) and any stand-alone
// strided FIR.  `in` points at the data start; in[-(T-1)..-1] is history.
//   y[c][o].re = gain * sum_p in[c][o*S + p - (T-1)].re * hI_c[p],   y.im likewise with hQ (CFir::ProcessFilter complex,
//   pebblelib/fir.cpp:106-132: separate I and Q coefficient sets; equal except after GenerateHBFilter)
// taps: one set for all channels (taps_pitch == 0) or per channel, in device memory; taps_q == nullptr => hQ = hI.
// sideband_mix: out = (re + im, re - im), the last loop of Demod_SAM::processBlock (demod_sam.cpp:103-110).
// grid (ceil(n_out/256), C or listed channels).
static __global__ __launch_bounds__(256) void k_fir_dec(const float2 *__restrict__ in, long long in_pitch,
                                                  float2 *__restrict__ out, long long out_pitch, long long n_out,
                                                  int stride, const float *__restrict__ taps, const float *__restrict__ taps_q,
                                                  int taps_pitch, const int *__restrict__ ntaps_per_chan, int ntaps_all,
                                                  float gain, int sideband_mix, const int *__restrict__ chan_list, Gate gate)
{
    __shared__ float hi[kMaxTaps], hq[kMaxTaps];
    const int c = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    if (gate.closed(c)) return;  // workgroup-uniform
    const int T = ntaps_per_chan ? ntaps_per_chan[c] : ntaps_all;
    if (threadIdx.x < kMaxTaps) {
        const int p = threadIdx.x;
        const float a = p < T ? taps[(long long)c * taps_pitch + p] : 0.f;
        hi[p] = a;
        hq[p] = (taps_q && p < T) ? taps_q[(long long)c * taps_pitch + p] : a;
    }
    __syncthreads();
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const float2 *x = in + (long long)c * in_pitch + o * stride - (T - 1);
    float2 acc = make_float2(0.f, 0.f);
    for (int p = 0; p < T; p++) {
        const float2 v = x[p];
        acc.x = fmaf(v.x, hi[p], acc.x);
        acc.y = fmaf(v.y, hq[p], acc.y);
    }
    acc = cscale(acc, gain);
    if (sideband_mix) acc = make_float2(acc.x + acc.y, acc.x - acc.y);
    out[(long long)c * out_pitch + o] = acc;
}

// Copy the last `hist` samples of each channel's data into its head-room: buf[c][-hist + j] = buf[c][n - hist + j].
// Requires n >= hist (true for every buffer: calls are whole super-frames).  grid (ceil(hist/256), C).
static __global__ __launch_bounds__(256) void k_save_tail(float2 *__restrict__ data, long long pitch, long long n, int hist,
                                                    const int *__restrict__ chan_list, Gate gate)
{
    const int c = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    if (gate.closed(c)) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= hist) return;
    float2 *b = data + (long long)c * pitch;
    b[-hist + j] = b[n - hist + j];
}

// DeviceInterfaceBase::normalizeIQ on the device (pebblelib/deviceinterfacebase.cpp:648-838) and WavFile::ReadSamples'
// PCM16 scaling (wavfile.cpp:299-300): raw device samples -> float2, gain and IQ order applied.  Uploading the raw
// 2 or 4 bytes per sample instead of converted doubles cuts the host->device traffic 4-8x.
//   fmt 0: CPX8  int8 pairs,  v * (1/128) * gain          fmt 1: CPXU8 uint8 pairs, (v - 128) * (1/128) * gain
//   fmt 2: CPX16 int16 pairs, v * (1/32768) * gain        fmt 3: CPXFLOAT, v * gain       fmt 4: WAV PCM16, v / 32767 * gain
//   order 0 IQ, 1 QI, 2 I only, 3 Q only  (DeviceInterface::IQOrder, device_interfaces.h:140-145)
static __global__ __launch_bounds__(256) void k_normalize_iq(const void *__restrict__ src, float2 *__restrict__ dst, long long n, int fmt,
                                                              int order, float scale)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float a, b;
        if (fmt == 0) {
            const char2 v = reinterpret_cast<const char2 *>(src)[i];
            a = (float)v.x; b = (float)v.y;
        } else if (fmt == 1) {
            const uchar2 v = reinterpret_cast<const uchar2 *>(src)[i];
            a = (float)v.x - 128.0f; b = (float)v.y - 128.0f;
        } else if (fmt == 2 || fmt == 4) {
            const short2 v = reinterpret_cast<const short2 *>(src)[i];
            a = (float)v.x; b = (float)v.y;
        } else {
            const float2 v = reinterpret_cast<const float2 *>(src)[i];
            a = v.x; b = v.y;
        }
        a *= scale;
        b *= scale;
        float2 o;
        if (order == 0) o = make_float2(a, b);
        else if (order == 1) o = make_float2(b, a);
        else if (order == 2) o = make_float2(a, a);
        else o = make_float2(b, b);
        dst[i] = o;
    }
}

// Bandwidth probes for the roofline (bench.py): plain streaming copies with 16-byte and 8-byte lanes, four loads in flight
// per work-item before the first store (one load in flight per work-item read 4.9 TB/s where the part sustains ~6.3).
template <class V>
__device__ __forceinline__ void probe_copy_body(const V *__restrict__ src, V *__restrict__ dst, long long n)
{
    // a workgroup walks 1024-element tiles (16 KiB / 8 KiB, contiguous), every work-item with four loads in flight
    const long long tiles = n / 1024;
    for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long long i = t * 1024 + threadIdx.x;
        const V a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
    }
    for (long long i = tiles * 1024 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
static __global__ __launch_bounds__(256) void k_probe_copy16(const float4 *__restrict__ src, float4 *__restrict__ dst, long long n)
{
    probe_copy_body(src, dst, n);
}
static __global__ __launch_bounds__(256) void k_probe_copy8(const float2 *__restrict__ src, float2 *__restrict__ dst, long long n)
{
    probe_copy_body(src, dst, n);
}

// The per-channel squelch of a bank.  k_gate_eval: open[c][j] = avgDb of the super-frame's last raw frame >= the channel's threshold
// (m_avgDb < m_squelchDb closes, receiver.cpp:962-965); k_gate_zero clears the audio of the closed (channel, super-frame) pairs --
// what a host sees where the reference would have delivered nothing.
static __global__ __launch_bounds__(256) void k_gate_eval(const float4 *__restrict__ smeter, long long smeter_pitch, int frames_per_sf, int k,
                                                          const float *__restrict__ squelch_db, unsigned char *__restrict__ open, int stride, int n_chan)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_chan * k) return;
    const int c = i / k, j = i % k;
    open[(long long)c * stride + j] = smeter[(long long)c * smeter_pitch + (long long)(j + 1) * frames_per_sf - 1].y < squelch_db[c] ? 0 : 1;
}
static __global__ __launch_bounds__(256) void k_gate_zero(float2 *__restrict__ audio, long long pitch, long long spf, const unsigned char *__restrict__ open, int stride)
{
    const int c = blockIdx.y, j = blockIdx.z;
    if (open[(long long)c * stride + j]) return;
    float2 *a = audio + (long long)c * pitch + (long long)j * spf;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < spf; i += (long long)gridDim.x * 256) a[i] = make_float2(0.f, 0.f);
}

// All history tails of a call in one launch: buf[c][-hist + j] = buf[c][n - hist + j], j < hist, for every buffer.
// n may be shorter than hist (then part of the old history is kept, shifted), so a workgroup first reads every value
// it will write.  grid (1, C, jobs), hist <= 256*32.
// One wave that sleeps for `ticks` of the 100 MHz wall clock (bounded: at most 4096 naps whatever the clock reads).  Queued in front of
// the second stage of a two-stage call so that the NEXT call's decimator, which becomes ready at the same moment on the other stream, has
// its workgroups placed first (Receiver::process).
static __global__ __launch_bounds__(64) void k_nap(unsigned ticks)
{
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < 4096 && wall_clock64() - t0 < (unsigned long long)ticks; i++) __builtin_amdgcn_s_sleep(8);
}

static __global__ __launch_bounds__(256) void k_save_tails(TailJobs jobs)
{
    save_tails_block(jobs, (int)blockIdx.z, (int)blockIdx.y, (int)threadIdx.x);
}

}  // namespace pg
