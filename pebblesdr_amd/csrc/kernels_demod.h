// kernels_demod.h -- demod-rate kernels: linear recurrences as wave-parallel scans, FM discriminator.
//
// The reference's demodulators are serial per sample: AM DC-block (demod_am.cpp:52-57), CIir
// biquads (pebblelib/iir.cpp:176-207) and the one-pole de-emphasis (demod_wfm.cpp:476-485).  All are
// LINEAR recurrences, so a wave evaluates kSub = 512 samples at once: lane l owns kSeg = 8 consecutive samples,
// runs them from a zero state, the 64 end states are combined with a Hillis-Steele scan using the
// precomputed transition powers M^(kSeg*2^k), and each lane re-runs its samples from its true entry state.
// Recurrence arithmetic is fp64 (the DC-block pole 0.9999 needs it); samples are stored fp32.
//
// Across workgroups a long single-channel stream is cut into chunks; a chunk that does not start at
// the call boundary warms up on `warm_sub` preceding sub-chunks from a zero state (host picks warm_sub
// so the pole radius^samples < 1e-13, else it selects the exact sequential mode, warm_sub < 0).
#pragma once
#include "params.h"
#include "tail_refresh.h"

namespace pg {


__device__ __forceinline__ int spad(int i) { return i + i / kSeg; }  // lane stride kSeg -> kSeg+1 dwords: conflict-free

__device__ __forceinline__ double sec_step(const ScanSection &S, double &s0, double &s1, double u)
{
    double y;
    if (S.type == kOnePoleDiff) {        // m_amDc = ALPHA*m_amDcLast + mag; out = m_amDc - m_amDcLast
        const double dn = (S.c[0] * s0) + u;
        y = dn - s0;
        s0 = dn;
    } else if (S.type == kOnePoleAvg) {  // ave = (1-a)*ave + a*in; out = 2*ave
        s0 = (1.0 - S.c[0]) * s0 + S.c[0] * u;
        y = s0 * 2.0;
    } else {                              // w0 = in - A1*w1 - A2*w2; out = B0*w0 + B1*w1 + B2*w2
        const double w0 = u - S.c[3] * s0 - S.c[4] * s1;
        y = S.c[0] * w0 + S.c[1] * s0 + S.c[2] * s1;
        s1 = s0;
        s0 = w0;
    }
    return y;
}

// One section over one sub-chunk held in LDS (padded float array), in place.  c0/c1: state entering
// the sub-chunk, replaced by the state after its last valid sample.  Must be called by all 64 lanes.
__device__ __forceinline__ void scan_sub(const ScanSection &S, float *buf, int nv, double &c0, double &c1, int lane)
{
    const int base = lane * kSeg;
    int cnt = nv - base;
    cnt = cnt < 0 ? 0 : (cnt > kSeg ? kSeg : cnt);
    double s0 = lane == 0 ? c0 : 0.0, s1 = lane == 0 ? c1 : 0.0;
#pragma unroll
    for (int i = 0; i < kSeg; i++) {
        const double u = i < cnt ? (double)buf[spad(base + i)] : 0.0;
        (void)sec_step(S, s0, s1, u);
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double t0 = __shfl_up(s0, 1u << k), t1 = __shfl_up(s1, 1u << k);
        if (lane >= (1 << k)) {
            s0 += S.P[k][0] * t0 + S.P[k][1] * t1;
            s1 += S.P[k][2] * t0 + S.P[k][3] * t1;
        }
    }
    double e0 = __shfl_up(s0, 1u), e1 = __shfl_up(s1, 1u);
    if (lane == 0) { e0 = c0; e1 = c1; }
#pragma unroll
    for (int i = 0; i < kSeg; i++) {
        if (i < cnt) {
            const int a = spad(base + i);
            buf[a] = (float)sec_step(S, e0, e1, (double)buf[a]);
        }
    }
    const int lf = (nv - 1) / kSeg;
    c0 = __shfl(e0, lf);
    c1 = __shfl(e1, lf);
}

// MODE 0: complex in -> complex out, both components filtered independently (CIir complex, iir.cpp:191-207)
// MODE 1: real: in.x -> (y, y)
// MODE 2: magnitude: |in| -> (y, y)      (Demod_AM::processBlockFiltered front half)
// state layout: [channel][section][component(2)][2] doubles.  grid (n_blocks, n_listed_channels), block 64.
template <int MODE, int NSEC>
static __global__ __launch_bounds__(64) void k_iir_scan(const float2 *__restrict__ in, long long in_pitch,
                                                  float2 *__restrict__ out, long long out_pitch, long long n,
                                                  ScanParams<NSEC> sp, const double *__restrict__ state_in,
                                                  double *__restrict__ state_out, int sub_per_block, int warm_sub,
                                                  const int *__restrict__ chan_list, Gate gate)
{
    __shared__ float re[kSub + kSub / kSeg + 1];
    __shared__ float im[kSub + kSub / kSeg + 1];
    const int lane = threadIdx.x;
    const int c = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    if (gate.closed(c)) return;  // workgroup-uniform
    const long long nsub = (n + kSub - 1) / kSub;
    const long long first_out = (long long)blockIdx.x * sub_per_block;
    long long last = first_out + sub_per_block;
    if (last > nsub) last = nsub;
    long long start = warm_sub < 0 ? 0 : first_out - warm_sub;
    if (start < 0) start = 0;

    double st[NSEC][2][2];
#pragma unroll
    for (int s = 0; s < NSEC; s++)
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
            const double *p = state_in + (((long long)c * NSEC + s) * 2 + comp) * 2;
            st[s][comp][0] = start == 0 ? p[0] : 0.0;
            st[s][comp][1] = start == 0 ? p[1] : 0.0;
        }

    const float2 *x = in + (long long)c * in_pitch;
    float2 *y = out + (long long)c * out_pitch;
    for (long long sub = start; sub < last; sub++) {
        const long long off = sub * kSub;
        int nv = (int)((n - off) < kSub ? (n - off) : kSub);
        for (int j = lane; j < nv; j += 64) {
            const float2 v = x[off + j];
            if (MODE == 0) { re[spad(j)] = v.x; im[spad(j)] = v.y; }
            else if (MODE == 1) re[spad(j)] = v.x;
            else re[spad(j)] = sqrtf(v.x * v.x + v.y * v.y);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NSEC; s++) {
            scan_sub(sp.sec[s], re, nv, st[s][0][0], st[s][0][1], lane);
            if (MODE == 0) scan_sub(sp.sec[s], im, nv, st[s][1][0], st[s][1][1], lane);
        }
        __syncthreads();
        if (sub >= first_out) {
            for (int j = lane; j < nv; j += 64) {
                const float a = re[spad(j)];
                y[off + j] = make_float2(a, MODE == 0 ? im[spad(j)] : a);
            }
        }
        __syncthreads();
    }
    if (last == nsub && lane == 0) {
#pragma unroll
        for (int s = 0; s < NSEC; s++)
#pragma unroll
            for (int comp = 0; comp < 2; comp++) {
                double *p = state_out + (((long long)c * NSEC + s) * 2 + comp) * 2;
                p[0] = st[s][comp][0];
                p[1] = st[s][comp][1];
            }
    }
}

// Demod_WFM::processDataMono (application/demod/demod_wfm.cpp:207-232) in ONE kernel with NO carried filter state:
//   biquad low-pass on I and Q (if rate >= 150 kHz)  ->  0.25*atan2 discriminator  ->  75-tap CFir  ->  de-emphasis  ->  19 kHz notch
// The recurrences of this path have fast poles (radius <= ~0.97 at every WFM rate), so the host folds each cascade into
// its truncated impulse response (design::cascade_impulse, fp64, tail below 1e-11 of the total): `hlp` for the low-pass
// and ONE combined response `h` for CFir * de-emphasis * notch.  Every output then depends only on input history --
// the previous call's last Lx input samples (`xtail`) -- and all outputs are computed independently: no scans, no
// warm-up runs, no sequential carries.  (The scan kernels above remain for slow poles: AM's DC block, other rates.)
//
// One workgroup per kWfmOutB = 1024 outputs.  x -> LDS (one pad slot per 8 samples); low-pass (fp32, Llp taps): each
// work-item owns 8 consecutive outputs and slides a register window over the padded samples -> LDS; discriminator ->
// LDS as fp32 in blocks of 4; audio FIR: each work-item owns 4 consecutive outputs, slides a two-block window through
// the history (one 16-byte LDS read per 4 taps against 16 FMAs), multiplies in fp32 and folds every 16-tap partial
// sum into an fp64 accumulator (fp64 FMA runs at a fraction of the fp32 rate here; the fold keeps the rounding of a
// 600..1500-tap sum at the 1e-7 level).
constexpr int kWfmOutB = 1024;    // outputs per workgroup
constexpr int kWfmIrMax = 1536;   // longest combined audio response (taps, multiple of 16) that fits the LDS budget
constexpr int kWfmLpMax = 64;     // longest low-pass response

struct WfmFirParams {
    int L4;        // combined audio response length, zero-padded to a multiple of 16
    int Llp;       // low-pass response length (1 with tap 1.0 when the low-pass is off)
    float gain;    // FMDEMOD_GAIN
    int pad_;
};
__host__ __device__ inline size_t wfm_fir_lds_bytes(int L4, int Llp)
{
    const size_t nd = (size_t)kWfmOutB + L4, nl = (nd + 1 + 7) & ~(size_t)7, nx = nl + Llp - 1;
    return (nx + nx / 8 + 1) * sizeof(float2) + (nl + 1) * sizeof(float2) + nd * sizeof(float);
}

// xtail: [channel][L4 + Llp] input samples preceding in[0] (zeros before the first call)
static __global__ __launch_bounds__(512) void k_wfm_fir(const float2 *__restrict__ in, long long in_pitch, const float2 *__restrict__ xtail,
                                                        float2 *__restrict__ out, long long out_pitch, long long n, WfmFirParams wp,
                                                        const float *__restrict__ h, const float *__restrict__ hlp,
                                                        const unsigned char *__restrict__ no_prefilter /* [channel] or null: dmFMS, see WfmCore */,
                                                        int n_fir_groups, TailJobs tails /* run by the workgroups behind the first n_fir_groups: the call's tail refresh rides on this launch */)
{
    if ((int)blockIdx.x >= n_fir_groups) {  // (none of what these jobs write is read by the demodulator's workgroups)
        save_tails_block(tails, (int)blockIdx.x - n_fir_groups, (int)blockIdx.y, (int)threadIdx.x);
        return;
    }
    HIP_DYNAMIC_SHARED(float2, dyn)
    const int L4 = wp.L4, Llp = wp.Llp;
    const int ND = kWfmOutB + L4, NL = (ND + 1 + 7) & ~7, NX = NL + Llp - 1, Lx = L4 + Llp;
    float2 *xs = dyn;                                                    // NX samples, position j + j/8
    float2 *lpb = xs + (NX + NX / 8 + 1);                                // NL (+1)
    float *db = reinterpret_cast<float *>(lpb + NL + 1);                 // ND
    __shared__ float hl[kWfmLpMax];
    const int tid = threadIdx.x, c = blockIdx.y;
    const long long s = (long long)blockIdx.x * kWfmOutB;
    const float2 *x = in + (long long)c * in_pitch;
    const float2 *xt = xtail + (long long)c * Lx;
    const bool raw_iq = no_prefilter != nullptr && no_prefilter[c] != 0;  // processDataStereo has no 75 kHz low-pass: the identity response
    if (tid < kWfmLpMax) hl[tid] = tid < Llp ? (raw_iq ? (tid == 0 ? 1.f : 0.f) : hlp[tid]) : 0.f;
    // ---- 1. input samples x0 .. x0+NX-1, x0 = s - L4 - Llp (history from the tail, nothing past n) ----
    const long long x0 = s - Lx;
    for (int j = tid; j < NX; j += 512) {
        const long long g = x0 + j;
        float2 v = make_float2(0.f, 0.f);
        if (g < 0) v = xt[Lx + g];
        else if (g < n) v = x[g];
        xs[j + (j >> 3)] = v;
    }
    __syncthreads();
    // ---- 2. low-pass: lpb[k] = sum_m hlp[m] * xs[k + Llp - 1 - m], k < NL.  Work-item: outputs 8a .. 8a+7; with the
    //         pad its 9-slot stride spreads a wave's 8-byte reads over all banks. ----
    for (int a8 = tid * 8; a8 < NL; a8 += 512 * 8) {
        float2 w[8], acc[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            acc[r] = make_float2(0.f, 0.f);
            const int j = a8 + r + Llp - 1;
            w[r] = xs[j + (j >> 3)];  // tap 0 operands: w[r] = x for output a8 + r
        }
        for (int m = 0; m < Llp; m++) {
            const float t = hl[m];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                acc[r].x = fmaf(t, w[r].x, acc[r].x);
                acc[r].y = fmaf(t, w[r].y, acc[r].y);
            }
#pragma unroll
            for (int r = 7; r > 0; r--) w[r] = w[r - 1];
            const int j = a8 + Llp - 2 - m;  // next tap's operand of output a8 (may run below 0 on the last turn)
            w[0] = j >= 0 ? xs[j + (j >> 3)] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) lpb[a8 + r] = acc[r];
    }
    __syncthreads();
    // ---- 3. discriminator d[s - L4 + i], i < ND (demod_wfm.cpp:217) ----
    for (int i = tid; i < ND; i += 512) {
        const float2 c0 = lpb[i + 1], c1 = lpb[i];
        db[i] = wp.gain * atan2f(c1.x * c0.y - c0.x * c1.y, c1.x * c0.x + c1.y * c0.y);
    }
    __syncthreads();
    // ---- 4. y[s + 4 tid + r] = sum_p h[p] * d[s + 4 tid + r - p] ----
    //         Taps are fetched 16 at a time, one chunk (64 FMAs per work-item) ahead of their use, with VECTOR loads of
    //         a lane-invariant address: scalar loads share the LDS wait counter and return out of order, which would
    //         force a full wait at every LDS read; taps parked in LDS cost four 16-byte broadcast reads per chunk and per
    //         wave (measured: 0.027 -> 0.040 ms).  The next history block is read from LDS one group ahead.
    //         The workgroup's two halves split the taps: work-items 0..255 run p in [0, L4/2) rounded to 16, 256..511 the rest, and
    //         the halves' fp64 partial sums meet in LDS (this kernel runs on an otherwise idle GPU behind the display transform:
    //         its own serial length is the call's time).
    const int half = tid >> 8, ot = tid & 255;
    const int p_split = ((L4 >> 1) + 15) & ~15;
    const int p_lo = half ? p_split : 0, p_hi = half ? L4 : p_split;
    const float4 *blk = reinterpret_cast<const float4 *>(db) + ot + (L4 >> 2) - (p_lo >> 2);
    float4 cur = blk[0], nxt = blk[-1];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int vz = 0;
    opaque(vz);  // a zero the compiler must treat as per-lane: keeps the tap loads on the vector memory path
    const float4 *hv = reinterpret_cast<const float4 *>(h) + vz;  // (16-byte loads: four per chunk)
    float4 hn[4];
#pragma unroll
    for (int u = 0; u < 4; u++) hn[u] = hv[(p_lo >> 2) + u];
    for (int p0 = p_lo; p0 < p_hi; p0 += 16) {  // L4 is a multiple of 16; h carries 16 zeros past it
        float ht[16];
#pragma unroll
        for (int u = 0; u < 4; u++) { ht[4 * u] = hn[u].x; ht[4 * u + 1] = hn[u].y; ht[4 * u + 2] = hn[u].z; ht[4 * u + 3] = hn[u].w; }
#pragma unroll
        for (int u = 0; u < 4; u++) hn[u] = hv[((p0 + 16) >> 2) + u];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            blk -= 1;
            const float4 prev = nxt;
            nxt = blk[-1];  // on the very last group this reads the 16 bytes in front of db (the end of lpb): in bounds, unused
            const float h0 = ht[4 * g], h1 = ht[4 * g + 1], h2 = ht[4 * g + 2], h3 = ht[4 * g + 3];
            a0 = fmaf(h0, cur.x, a0); a1 = fmaf(h0, cur.y, a1); a2 = fmaf(h0, cur.z, a2); a3 = fmaf(h0, cur.w, a3);
            a0 = fmaf(h1, prev.w, a0); a1 = fmaf(h1, cur.x, a1); a2 = fmaf(h1, cur.y, a2); a3 = fmaf(h1, cur.z, a3);
            a0 = fmaf(h2, prev.z, a0); a1 = fmaf(h2, prev.w, a1); a2 = fmaf(h2, cur.x, a2); a3 = fmaf(h2, cur.y, a3);
            a0 = fmaf(h3, prev.y, a0); a1 = fmaf(h3, prev.z, a1); a2 = fmaf(h3, prev.w, a2); a3 = fmaf(h3, cur.x, a3);
            cur = prev;
        }
        acc[0] += (double)a0; acc[1] += (double)a1; acc[2] += (double)a2; acc[3] += (double)a3;
    }
    // the second half's sums through LDS (xs is free: every work-item is past the low-pass), added in the first half
    __syncthreads();
    double *part = reinterpret_cast<double *>(xs);
    if (half) {
#pragma unroll
        for (int r = 0; r < 4; r++) part[r * 256 + ot] = acc[r];
    }
    __syncthreads();
    if (half) return;
    float2 *y = out + (long long)c * out_pitch + s + 4 * ot;
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (s + 4 * ot + r < n) {
            const float v = (float)(acc[r] + part[r * 256 + ot]);
            y[r] = make_float2(v, v);  // mono: left = right (demod_wfm.cpp:229-230)
        }
}

// FM stereo while the reference's pilot PLL still reports lock (Demod_WFM::processDataStereo, demod_wfm.cpp:255-297; processPilotPll
// :392-429; arctan2 :792-821): the discriminator, the 61-tap Hilbert pair, the pilot band-pass and the PLL are one serial loop per
// channel -- nonlinear feedback -- in double as the reference declares it; one lane per channel.  The lock decision is taken at the
// END of a block and applies to the whole block: a locked block contributes lmr[i] = 2 raw[i] sin(2 (nco_phase_i + adjust)) to
// left - right, an unlocked one nothing.  The first block that ends without lock ends the loop for good (`dropped`): the lock
// average, once above its threshold, needs seconds of a quiet phase detector to come back, and this detector never settles
// (tests/test_oracle_pins.py::test_wfm_stereo_pilot_pll_of_the_reference_does_not_hold_lock) -- from there on dmFMS is what k_wfm_fir
// delivers by itself.  lm: [channel][lm_pitch] float2 rows with the audio response's look-back as head-room, .x = lmr (zeroed by the
// caller before the launch: only locked blocks are written).
static __global__ __launch_bounds__(64) void k_wfm_pilot(const float2 *__restrict__ in, long long in_pitch, long long n, WfmPilotParams pp,
                                                         const double *__restrict__ hilb, WfmPilotState *__restrict__ st,
                                                         float2 *__restrict__ lm, long long lm_pitch, const int *__restrict__ chan_list, int nlist)
{
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nlist) return;
    const int c = chan_list[li];
    WfmPilotState *s = &st[c];
    const long long quiet0 = s->quiet;
    if (s->dropped) {
        s->skip = quiet0 >= (long long)pp.L4 ? 1 : 0;
        const long long q = quiet0 + n;
        s->quiet = q > (1LL << 40) ? (1LL << 40) : q;
        const float2 v = in[(long long)c * in_pitch + n - 1];
        s->d1_re = (double)v.x;
        s->d1_im = (double)v.y;
        return;
    }
    s->skip = 0;
    const double kTwoPi = 6.28318530717958647692528676656, kPiD = 3.14159265358979323846;
    const float2 *x = in + (long long)c * in_pitch;
    float2 *l = lm + (long long)c * lm_pitch;
    double d1r = s->d1_re, d1i = s->d1_im, w1a = s->w1a, w2a = s->w2a, w1b = s->w1b, w2b = s->w2b;
    double nco_phase = s->nco_phase, nco_freq = s->nco_freq, err_ave = s->err_ave;
    int zpos = s->zpos;
    long long quiet = quiet0;
    const double *hI = hilb, *hQ = hilb + 61;
    for (long long b0 = 0; b0 < n; b0 += pp.block) {
        const long long be = b0 + pp.block < n ? b0 + pp.block : n;
        for (long long i = b0; i < be; i++) {
            const double xr = (double)x[i].x, xi = (double)x[i].y;
            const double raw = 0.25 * atan2(d1r * xi - xr * d1i, d1r * xr + d1i * xi);  // :258-263 (FMDEMOD_GAIN)
            d1r = xr; d1i = xi;
            // CFir::ProcessFilter, real in / complex out (fir.cpp:106-154): y = sum_k h[k] raw[i - k]
            zpos = zpos == 0 ? 60 : zpos - 1;
            s->z[zpos] = raw;
            double hr = 0.0, hi = 0.0;
            int q = zpos;
            for (int k = 0; k < 61; k++) {
                const double zv = s->z[q];
                hr += hI[k] * zv;
                hi += hQ[k] * zv;
                q = q == 60 ? 0 : q + 1;
            }
            // pilot band-pass, direct form 2 (iir.cpp:191-207)
            const double w0a = hr - pp.a1 * w1a - pp.a2 * w2a;
            const double pr = pp.b0 * w0a + pp.b2 * w2a;
            w2a = w1a; w1a = w0a;
            const double w0b = hi - pp.a1 * w1b - pp.a2 * w2b;
            const double pi = pp.b0 * w0b + pp.b2 * w2b;
            w2b = w1b; w1b = w0b;
            // processPilotPll, :392-429
            const double sn = sin(nco_phase), cs = cos(nco_phase);
            const double tr = cs * pr - sn * pi, ti = cs * pi + sn * pr;
            double ang;  // Demod_WFM::arctan2(ti, tr), :792-821, with its constants as written
            if (tr == 0.0) ang = ti > 0.0 ? kTwoPi : (ti == 0.0 ? 0.0 : -kTwoPi);
            else {
                const double zq = ti / tr;
                if (fabs(zq) < 1.0) {
                    ang = zq / (1.0 + 0.2854 * zq * zq);
                    if (tr < 0.0) ang = ti < 0.0 ? ang - kPiD : ang + kPiD;
                } else {
                    ang = kTwoPi - zq / (zq * zq + 0.2854);
                    if (ti < 0.0) ang -= kPiD;
                }
            }
            const double err = -ang;
            nco_freq += pp.beta * err;
            if (nco_freq > pp.nco_hi) nco_freq = pp.nco_hi;
            else if (nco_freq < pp.nco_lo) nco_freq = pp.nco_lo;
            nco_phase += nco_freq + pp.alpha * err;
            err_ave = (1.0 - pp.err_alpha) * err_ave + pp.err_alpha * err * err;
            l[i].x = (float)(2.0 * raw * sin((nco_phase + pp.phase_adjust) * 2.0));  // kept if the block ends locked
        }
        nco_phase = fmod(nco_phase, kTwoPi);
        if (!(err_ave < 0.05)) {  // LOCK_MAG_THRESHOLD: the block copies the mono signal
            for (long long i = b0; i < be; i++) l[i].x = 0.f;
            s->dropped = 1;
            quiet += n - b0;
            break;
        }
        quiet = 0;
    }
    s->d1_re = (double)x[n - 1].x; s->d1_im = (double)x[n - 1].y;
    s->w1a = w1a; s->w2a = w2a; s->w1b = w1b; s->w2b = w2b;
    s->nco_phase = nco_phase; s->nco_freq = nco_freq; s->err_ave = err_ave;
    s->zpos = zpos;
    s->quiet = quiet;
}

// out[c][i] += (w, -w), w = sum_p h[p] lmr[i - p]: the (L - R) part through the audio response k_wfm_fir applies to L + R
// (demod_wfm.cpp:359-361 are linear: FIR, de-emphasis, notch).  grid (ceil(n / 256), listed channels).
static __global__ __launch_bounds__(256) void k_wfm_lmr_fir(const float2 *__restrict__ lm, long long lm_pitch, const float *__restrict__ h, int L4,
                                                           float2 *__restrict__ out, long long out_pitch, long long n,
                                                           const WfmPilotState *__restrict__ st, const int *__restrict__ chan_list)
{
    const int c = chan_list[blockIdx.y];
    if (st[c].skip) return;  // workgroup-uniform
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 *l = lm + (long long)c * lm_pitch + i;
    double acc = 0.0;
    for (int p0 = 0; p0 < L4; p0 += 16) {
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < 16; u++) a = fmaf(h[p0 + u], l[-(p0 + u)].x, a);
        acc += (double)a;
    }
    float2 *y = out + (long long)c * out_pitch + i;
    const float w = (float)acc;
    const float2 v = *y;
    *y = make_float2(v.x + w, v.y - w);
}

// The PLL demodulators -- Demod_NFM::processBlockNCO (application/demod/demod_nfm.cpp:225-257) and Demod_SAM::pll /
// processBlock (demod_sam.cpp:41-101) -- are NON-linear feedback loops: serial in time, parallel only across channels.
// One lane per channel walks the call; the loop state is `float` exactly as the reference declares it
// (demod_nfm.h:27-40, demod_sam.h:19-25), trigonometry of the float phase uses the float functions (what sin(float)
// resolves to in C++), everything else is double as written.  Output goes to a buffer with FIR head-room; the
// CFir that follows is k_fir_dec.
static __global__ __launch_bounds__(64) void k_pll_demod(const float2 *__restrict__ in, long long in_pitch, float2 *__restrict__ out,
                                                         long long out_pitch, long long n, PllParams pp, PllState *__restrict__ st,
                                                         const int *__restrict__ chan_list, int nlist, Gate gate)
{
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nlist) return;
    const int c = chan_list ? chan_list[li] : li;
    if (gate.closed(c)) return;
    const float2 *x = in + (long long)c * in_pitch;
    float2 *y = out + (long long)c * out_pitch;
    PllState s = st[c];
    const double kTwoPi = 6.28318530717958647692528676656, kPiD = 3.14159265358979323846;
    if (pp.mode == 0) {
        for (long long i = 0; i < n; i++) {
            const double nco_sin = (double)sinf(s.phase), nco_cos = (double)cosf(s.phase);
            const float2 v = x[i];
            const double tr = nco_cos * (double)v.x - nco_sin * (double)v.y;
            const double ti = nco_cos * (double)v.y + nco_sin * (double)v.x;
            const double phzerror = -atan2(ti, tr);
            s.freq = (float)((double)s.freq + ((double)pp.beta * phzerror));
            if (s.freq > pp.hi) s.freq = pp.hi;
            else if (s.freq < pp.lo) s.freq = pp.lo;
            s.phase = (float)((double)s.phase + ((double)s.freq + (double)pp.alpha * phzerror));
            s.err_dc = (float)((1.0 - (double)pp.dc_alpha) * (double)s.err_dc + (double)pp.dc_alpha * (double)s.freq);
            y[i] = make_float2(__fmul_rn(__fsub_rn(s.freq, s.err_dc), pp.out_gain), 0.f);  // float arithmetic; CPX = real: imag 0
        }
        s.phase = (float)fmod((double)s.phase, kTwoPi);  // "keep radian counter bounded", once per block
    } else {
        for (long long i = 0; i < n; i++) {
            const float2 v = x[i];
            const double sr = (double)v.x, si = (double)v.y;
            const double zr = (double)cosf(s.phase), zi = (double)sinf(s.phase);
            const double pr = zr * sr - zi * si, pi = zr * si + zi * sr;
            double ph = atan(pi / ((pr == 0) ? 1e-200 : pr));  // CpxUtil::phaseCpx, cpx.cpp:5-21
            if (pr < 0 && pi < 0) ph -= kPiD;
            else if (pr < 0 && pi >= 0) ph += kPiD;
            const float diff = (float)(sqrt(sr * sr + si * si) * ph);
            s.freq = __fadd_rn(s.freq, __fmul_rn(pp.beta, diff));  // float arithmetic, no FMA contraction (as an x86-64 build)
            if (s.freq < pp.lo) s.freq = pp.lo;
            if (s.freq > pp.hi) s.freq = pp.hi;
            s.phase = __fadd_rn(s.phase, __fadd_rn(s.freq, __fmul_rn(pp.alpha, diff)));
            while ((double)s.phase >= kTwoPi) s.phase = (float)((double)s.phase - kTwoPi);
            while (s.phase < 0) s.phase = (float)((double)s.phase + kTwoPi);
            const double dre = ((double)0.9999f * s.dc_re_last) + pr, dim = ((double)0.9999f * s.dc_im_last) + pi;
            y[i] = make_float2((float)(dre - s.dc_re_last), (float)(dim - s.dc_im_last));
            s.dc_re_last = dre;
            s.dc_im_last = dim;
        }
    }
    st[c] = s;
}

// FM discriminator, demod_wfm.cpp:214-220: out = gain * atan2(I1*Q0 - I0*Q1, I1*I0 + Q1*Q0), written to
// both components.  in[-1] is the previous call's last sample (head-room 1).  grid (ceil(n/256), C).
static __global__ __launch_bounds__(256) void k_discrim(const float2 *__restrict__ in, long long in_pitch,
                                                  float2 *__restrict__ out, long long out_pitch, long long n, float gain)
{
    const int c = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 *x = in + (long long)c * in_pitch;
    const float2 d1 = x[i - 1], d0 = x[i];
    const float v = gain * atan2f(d1.x * d0.y - d0.x * d1.y, d1.x * d0.x + d1.y * d0.y);
    out[(long long)c * out_pitch + i] = make_float2(v, v);
}

// copy (the WFM IIR low-pass is skipped below 150 kHz, demod_wfm.cpp:210-212).  grid (ceil(n/256), C)
static __global__ __launch_bounds__(256) void k_copy(const float2 *__restrict__ in, long long in_pitch,
                                               float2 *__restrict__ out, long long out_pitch, long long n)
{
    const int c = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[(long long)c * out_pitch + i] = in[(long long)c * in_pitch + i];
}

// ------------------------------------------------------------------------------------------------
// AGC::processBlock (application/agc.cpp:84-235): delay line, sliding-window peak of log magnitudes, attack/decay
// averagers, gain from the knee curve.  Non-linear feedback with data-dependent control flow: serial in time, one lane
// per listed channel, fp64 state as the reference declares it; in place on the band-passed samples.
// ------------------------------------------------------------------------------------------------
constexpr int kAgcMaxDelayBuf = 2048;  // agc.h MAX_DELAY_BUF
struct AgcState {        // per channel, device resident
    int mode, use_hang, delay_samples, window_samples, hang_time, hang_timer, sig_ptr, mag_pos;
    double manual_gain, decay_avg, attack_avg, attack_rise, attack_fall, decay_rise, decay_fall, fixed_gain, knee, gain_slope, peak;
    float2 sig[kAgcMaxDelayBuf];
    double mag[kAgcMaxDelayBuf];
};

static __global__ __launch_bounds__(64) void k_agc(float2 *__restrict__ buf, long long pitch, long long n, AgcState *__restrict__ st,
                                                   const int *__restrict__ chan_list, int nlist, Gate gate)
{
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nlist) return;
    const int c = chan_list[li];
    if (gate.closed(c)) return;
    float2 *x = buf + (long long)c * pitch;
    AgcState *a = st + c;
    if (a->mode == 0) {  // manual gain, agc.cpp:86-95
        const double g = a->manual_gain;
        for (long long i = 0; i < n; i++) x[i] = make_float2((float)(g * (double)x[i].x), (float)(g * (double)x[i].y));
        return;
    }
    const float out_scale = 0.7f, min_const = 1e-8f, max_amp = 1.0f;  // agc.h: float constants
    int sig_ptr = a->sig_ptr, mag_pos = a->mag_pos, hang_timer = a->hang_timer;
    const int delay_samples = a->delay_samples, window_samples = a->window_samples, hang_time = a->hang_time, use_hang = a->use_hang;
    double peak = a->peak, attack_avg = a->attack_avg, decay_avg = a->decay_avg;
    const double ar = a->attack_rise, af = a->attack_fall, dr = a->decay_rise, df = a->decay_fall;
    const double knee = a->knee, gain_slope = a->gain_slope, fixed_gain = a->fixed_gain;
    for (long long i = 0; i < n; i++) {
        const float2 in = x[i];
        const float2 delayed = a->sig[sig_ptr];
        a->sig[sig_ptr++] = in;
        if (sig_ptr >= delay_samples) sig_ptr = 0;
        double mag = fabs((double)in.x);
        const double mim = fabs((double)in.y);
        if (mim > mag) mag = mim;
        mag = log10(mag + (double)min_const) - log10((double)max_amp);
        double tmp = a->mag[mag_pos];
        a->mag[mag_pos++] = mag;
        if (mag_pos >= window_samples) mag_pos = 0;
        if (mag > peak) {
            peak = mag;
        } else if (tmp == peak) {
            peak = -8.0;
            for (int k = 0; k < window_samples; k++) {
                tmp = a->mag[k];
                if (tmp > peak) peak = tmp;
            }
        }
        if (peak > attack_avg) attack_avg = (1.0 - ar) * attack_avg + ar * peak;
        else attack_avg = (1.0 - af) * attack_avg + af * peak;
        if (use_hang) {
            if (peak > decay_avg) { decay_avg = (1.0 - dr) * decay_avg + dr * peak; hang_timer = 0; }
            else if (hang_timer < hang_time) hang_timer++;
            else decay_avg = (1.0 - df) * decay_avg + df * peak;
        } else {
            if (peak > decay_avg) decay_avg = (1.0 - dr) * decay_avg + dr * peak;
            else decay_avg = (1.0 - df) * decay_avg + df * peak;
        }
        mag = attack_avg > decay_avg ? attack_avg : decay_avg;
        const double gain = mag <= knee ? fixed_gain : (double)out_scale * pow(10.0, mag * (gain_slope - 1.0));
        x[i] = make_float2((float)((double)delayed.x * gain), (float)((double)delayed.y * gain));
    }
    a->sig_ptr = sig_ptr; a->mag_pos = mag_pos; a->hang_timer = hang_timer;
    a->peak = peak; a->attack_avg = attack_avg; a->decay_avg = decay_avg;
}

// ------------------------------------------------------------------------------------------------
// CFractResampler::Resample, complex version (pebblelib/fractresampler.cpp:149-195): every output convolves 28 input
// samples with a windowed sinc read from a 280 001-point table at floor((j - t) * 10000).  The output times t are the
// reference's running fp64 sum (m_FloatTime += dt per output, -= InLength per frame); the host replays that sum per
// call -- it also yields the output count without a device sync -- and hands each frame's start time to the kernel,
// which re-adds dt in the same order, so every table index is the reference's own.
// grid (frames * sub_blocks, channels), block 256.
// ------------------------------------------------------------------------------------------------
constexpr int kSincPeriodPts = 10000, kSincPeriods = 28, kSincLength = kSincPeriods * kSincPeriodPts + 1;
struct ResampFrame { double t_start; int out_offset, nout; };

static __global__ __launch_bounds__(256) void k_resample(const float2 *__restrict__ in, long long in_pitch, const float2 *__restrict__ hist,
                                                         float2 *__restrict__ out, long long out_pitch, int nf, int sub_blocks, double dt,
                                                         const ResampFrame *__restrict__ frames, const float *__restrict__ sinc)
{
    __shared__ double tk[256];
    const int f = blockIdx.x / sub_blocks, sb = blockIdx.x % sub_blocks, c = blockIdx.y, tid = threadIdx.x;
    const ResampFrame fr = frames[f];
    if (sb * 256 >= fr.nout) return;
    if (tid == 0) {
        double t = fr.t_start;
        for (int k = 0; k < sb * 256; k++) t += dt;
        for (int k = 0; k < 256; k++) { tk[k] = t; t += dt; }
    }
    __syncthreads();
    const int k = sb * 256 + tid;
    if (k >= fr.nout) return;
    const double t = tk[tid];
    const int it = (int)t;
    // m_pInputBuf[j], j = it + i: [28 samples of history | this frame]; frame f of the call starts at in[f * nf]
    const float2 *x = in + (long long)c * in_pitch + (long long)f * nf - kSincPeriods;
    const float2 *h = hist + (long long)c * kSincPeriods;
    double ar = 0.0, ai = 0.0;
#pragma unroll 4
    for (int i = 1; i <= kSincPeriods; i++) {
        const int j = it + i;
        const int sindx = (int)(((double)j - t) * (double)kSincPeriodPts);
        const float2 v = (f == 0 && j < kSincPeriods) ? h[j] : x[j];
        const double w = (double)sinc[sindx];
        ar = ar + (double)v.x * w;
        ai = ai + (double)v.y * w;
    }
    out[(long long)c * out_pitch + fr.out_offset + k] = make_float2((float)ar, (float)ai);
}

// ------------------------------------------------------------------------------------------------
// Pre-chain conditioners (receiver.cpp:814-823) and the noise filter (receiver.cpp:974).  All are default-off in the
// reference and serial in time by construction (adaptive / thresholded feedback); they run one lane per independent
// unit so that banks of streams parallelise, and exist for functional parity, not for the throughput path.
// ------------------------------------------------------------------------------------------------
// IQBalance::ProcessBlock (application/iqbalance.cpp:65-86): the adaptive terms t1, t2 restart at zero every block,
// so every (stream, frame) is independent: one lane each.  grid (ceil(frames/64), streams), block 64.
static __global__ __launch_bounds__(64) void k_iq_balance(float2 *__restrict__ buf, long long pitch, int nf, long long n_frames,
                                                          const double2 *__restrict__ factors /* [stream] (gain, phase); gain < 0: off */)
{
    const long long f = (long long)blockIdx.x * 64 + threadIdx.x;
    const int s = blockIdx.y;
    if (f >= n_frames) return;
    const double2 gp = factors[s];
    if (gp.x < 0) return;
    float2 *x = buf + (long long)s * pitch + f * nf;
    double t1r = 0, t1i = 0, t2r = 0, t2i = 0;
    const float mu = 0.0025f;
    const double sc = 1.0 - mu * 0.000001;
    for (int i = 0; i < nf; i++) {
        const float2 v = x[i];
        const double orr = (double)v.x * gp.x;
        const double oi = (double)v.y + ((double)v.x * gp.y);
        t1r = orr + (t2r * orr + t2i * oi);
        t1i = oi + (t2i * orr - t2r * oi);
        const double sqr = t1r * t1r - t1i * t1i, sqi = 2.0 * t1r * t1i;
        t2r = t2r * sc - sqr * mu;
        t2i = t2i * sc - sqi * mu;
        x[i] = make_float2((float)t1r, (float)t1i);
    }
}

// NoiseBlanker::ProcessBlock then ProcessBlock2 (application/noiseblanker.cpp:45-97), state carried across calls.
struct NbState {
    float nb_avg_mag, nb2_avg_mag;
    int spike_count, head, last, flags;  // flags: 1 NB1 on, 2 NB2 on
    double nb2_avg[2];
    float2 delay[8];                     // DelayLine(8, 2)
};
static __global__ __launch_bounds__(64) void k_noise_blank(float2 *__restrict__ buf, long long pitch, long long n, NbState *__restrict__ st, int n_streams)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_streams) return;
    NbState b = st[s];
    if (!(b.flags & 3)) return;
    float2 *x = buf + (long long)s * pitch;
    const double threshold = 3.3;
    for (long long i = 0; i < n; i++) {
        float2 v = x[i];
        if (b.flags & 1) {
            const float mag = (float)sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y);
            b.delay[b.head] = v;
            b.last = b.head;
            b.head = b.head == 0 ? 7 : b.head - 1;
            b.nb_avg_mag = (float)((0.999 * (double)b.nb_avg_mag) + (0.001 * (double)mag));
            if (b.spike_count == 0 && (double)mag > ((double)b.nb_avg_mag * threshold)) b.spike_count = 7;
            if (b.spike_count > 0) {
                v = make_float2(0.f, 0.f);
                b.spike_count--;
            } else {
                v = b.delay[(b.last + 2) % 8];
            }
        }
        if (b.flags & 2) {
            const float mag = (float)sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y);
            b.nb2_avg[0] = b.nb2_avg[0] * 0.75 + (double)v.x * 0.25;
            b.nb2_avg[1] = b.nb2_avg[1] * 0.75 + (double)v.y * 0.25;
            b.nb2_avg_mag = (float)(0.999 * (double)b.nb2_avg_mag + 0.001 * (double)mag);
            if ((double)mag > (threshold * (double)b.nb2_avg_mag)) v = make_float2((float)b.nb2_avg[0], (float)b.nb2_avg[1]);
        }
        x[i] = v;
    }
    st[s] = b;
}

// NoiseFilter::ProcessBlock (ANF: 45-tap leaky LMS predictor on a 64-sample delay, application/noisefilter.cpp:31-88).
// One WAVE per listed channel: lane j < 45 owns coefficient j and reads delayed sample j; the two dot products are
// wave reductions.  The delay line lives in the state block (global memory, L2-resident).
constexpr int kAnfTaps = 45, kAnfDelaySize = 512, kAnfDelay = 64;
struct AnfState {
    double coeff[2 * kAnfTaps];
    float2 delay[kAnfDelaySize];
    int head, last, pad_[2];
};
static __global__ __launch_bounds__(64) void k_anf(float2 *__restrict__ buf, long long pitch, long long n, AnfState *__restrict__ st,
                                                   const int *__restrict__ chan_list, Gate gate)
{
    const int c = chan_list[blockIdx.x], lane = threadIdx.x;
    if (gate.closed(c)) return;  // wave-uniform
    float2 *x = buf + (long long)c * pitch;
    AnfState *a = st + c;
    const bool tap = lane < kAnfTaps;
    double cr = tap ? a->coeff[2 * lane] : 0.0, ci = tap ? a->coeff[2 * lane + 1] : 0.0;
    int head = a->head, last = a->last;
    const double rate = 0.01, leakage = 0.00001, scl1 = 1.0 - rate * leakage;
    for (long long i = 0; i < n; i++) {
        const float2 in = x[i];
        if (lane == 0) a->delay[head] = in;
        last = head;
        head = head == 0 ? kAnfDelaySize - 1 : head - 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float2 d = tap ? a->delay[(last + kAnfDelay + lane) % kAnfDelaySize] : make_float2(0.f, 0.f);
        const double dr = (double)d.x, di = (double)d.y;
        double sosr = dr * dr, sosi = di * di, accr = cr * dr, acci = ci * di;
        // the reference sums j = 0..44 in order; a tree sum differs in the last bits only (the result is stored as float)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            sosr += __shfl_xor(sosr, m); sosi += __shfl_xor(sosi, m);
            accr += __shfl_xor(accr, m); acci += __shfl_xor(acci, m);
        }
        if (lane == 0) x[i] = make_float2((float)(accr * 1.25), (float)(acci * 1.25));
        const double er = ((double)in.x - accr) * (rate / (sosr + 1e-10));
        const double ei = ((double)in.y - acci) * (rate / (sosi + 1e-10));
        cr = cr * scl1 + er * dr;
        ci = ci * scl1 + ei * di;
    }
    if (tap) { a->coeff[2 * lane] = cr; a->coeff[2 * lane + 1] = ci; }
    if (lane == 0) { a->head = head; a->last = last; }
}

}  // namespace pg
