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
                                                  const int *__restrict__ chan_list)
{
    __shared__ float re[kSub + kSub / kSeg + 1];
    __shared__ float im[kSub + kSub / kSeg + 1];
    const int lane = threadIdx.x;
    const int c = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
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

// Demod_WFM::processDataMono (application/demod/demod_wfm.cpp:207-232) in ONE kernel, one wave per 512 output samples:
//   biquad low-pass on I and Q (if rate >= 150 kHz)  ->  0.25*atan2 discriminator  ->  75-tap CFir  ->  de-emphasis  ->  19 kHz notch
// The two recurrences are wave scans (scan_sub).  A workgroup that does not start at the call boundary rebuilds their
// state by running the filters over a warm-up stretch from zero state (host sizes it so pole^samples < 1e-13):
//   notch/de-emphasis warm-up Wd, then 74 samples of FIR look-back + 1 of discriminator look-back, then the low-pass
//   warm-up Wl  ->  up to Wd + Wl + 512 + 75 input samples per 512 outputs, all kept in LDS.
// At the call boundary the exact state of the previous call is used instead: recurrence states, the last low-passed
// sample and the last 74 discriminator outputs (ping-pong buffers: block 0 reads while the last block writes).
// grid (ceil(n/512), C), block 256 (wave 0 scans I and later the audio, wave 1 scans Q; all four do the rest).
struct WfmParams {
    ScanSection lp;       // biquad low-pass (applied to I and Q independently)
    ScanSection dn[2];    // de-emphasis, notch
    int lp_on, ntaps;     // FIR length (<= 75)
    int warm_lp, warm_dn; // warm-up lengths in samples (multiples of kSub)
    float gain;           // FMDEMOD_GAIN
    int pad_;
};
struct WfmState {         // per channel
    double lp[2][2];      // [component][2]
    double dn[2][2];      // [section][2]
    float2 lp_last;       // last low-passed sample of the previous call (discriminator look-back)
    float dtail[kMaxTaps];  // last ntaps-1 discriminator outputs, oldest first (FIR look-back)
    float pad_[2];
};

constexpr int kWfmMaxWarm = 4 * kSub;  // per recurrence

static __global__ __launch_bounds__(256) void k_wfm_mono(const float2 *__restrict__ in, long long in_pitch, float2 *__restrict__ out,
                                                         long long out_pitch, long long n, WfmParams wp,
                                                         const float *__restrict__ taps, const WfmState *__restrict__ st_in,
                                                         WfmState *__restrict__ st_out)
{
    constexpr int kPad = kSub + kSub / kSeg + 1;
    constexpr int kLp = kWfmMaxWarm + kSub + kMaxTaps;   // low-passed samples a block may keep
    __shared__ float re[kPad], im[kPad];
    __shared__ float2 lpbuf[kLp];
    __shared__ float dbuf[kLp + 8 * 256];                 // + slack so the register-blocked FIR may read past the end
    __shared__ float fbuf[kWfmMaxWarm + kSub];
    __shared__ float ht[kMaxTaps];
    const int tid = threadIdx.x, lane = tid & 63, c = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = wp.ntaps;
    const float2 *x = in + (long long)c * in_pitch;
    const WfmState *si = st_in + c;
    const long long s = (long long)blockIdx.x * kSub;                    // first output of this block
    const long long e = (s + kSub) < n ? (s + kSub) : n;                 // one past its last output
    const bool last_block = e == n;
    if (tid < kMaxTaps) ht[tid] = tid < T ? taps[tid] : 0.f;

    long long ds = s - wp.warm_dn;                                       // where the de-emphasis/notch run starts
    const bool dn_exact = ds <= 0;
    if (ds < 0) ds = 0;
    const long long d0 = ds - (T - 1);                                   // first discriminator sample the FIR reads (may be < 0)
    long long lk = d0 - 1;                                               // first low-passed sample kept (discriminator look-back)
    if (lk < 0) lk = 0;
    long long ls = lk - wp.warm_lp;                                      // where the low-pass run starts
    const bool lp_exact = ls <= 0 || !wp.lp_on;
    if (ls < 0 || !wp.lp_on) ls = wp.lp_on ? 0 : lk;

    // ---- 1. low-pass over [ls, e), keeping [lk, e) in lpbuf.  Wave 0 scans I, wave 1 scans Q. ----
    double q0 = 0, q1 = 0;  // biquad state of this wave's component
    if (lp_exact && wp.lp_on && wave < 2) { q0 = si->lp[wave][0]; q1 = si->lp[wave][1]; }
    for (long long off = ls; off < e; off += kSub) {
        const int nv = (int)((e - off) < kSub ? (e - off) : kSub);
        for (int j = tid; j < nv; j += 256) {
            const float2 v = x[off + j];
            re[spad(j)] = v.x;
            im[spad(j)] = v.y;
        }
        __syncthreads();
        if (wp.lp_on) {
            if (wave == 0) scan_sub(wp.lp, re, nv, q0, q1, lane);
            else if (wave == 1) scan_sub(wp.lp, im, nv, q0, q1, lane);
        }
        __syncthreads();
        for (int j = tid; j < nv; j += 256) {
            const long long idx = off + j;
            if (idx >= lk) lpbuf[idx - lk] = make_float2(re[spad(j)], im[spad(j)]);
        }
        __syncthreads();
    }
    // ---- 2. discriminator over [max(d0,0), e); older samples come from the previous call's tail ----
    const int nd = (int)(e - d0);                                        // dbuf[i] <-> discriminator sample d0 + i
    for (int i = tid; i < nd; i += 256) {
        const long long idx = d0 + i;
        float v;
        if (idx < 0) {
            v = si->dtail[(T - 1) + idx];
        } else {
            const float2 c0 = lpbuf[idx - lk];
            const float2 c1 = idx == 0 ? si->lp_last : lpbuf[idx - 1 - lk];
            v = wp.gain * atan2f(c1.x * c0.y - c0.x * c1.y, c1.x * c0.x + c1.y * c0.y);  // demod_wfm.cpp:217
        }
        dbuf[i] = v;
    }
    __syncthreads();
    // ---- 3. CFir over [ds, e): y[i] = sum_p d[i - (T-1) + p] * h[p]; 10 outputs per work-item in flight ----
    const int nf = (int)(e - ds);
    {
        constexpr int RB = (kWfmMaxWarm + kSub) / 256;  // 10
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; r++) acc[r] = 0.f;
        const float *dp = dbuf + tid;
        for (int p = 0; p < T; p++) {
            const float h = ht[p];
#pragma unroll
            for (int r = 0; r < RB; r++) acc[r] = fmaf(dp[p + 256 * r], h, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < RB; r++)
            if (tid + 256 * r < nf) fbuf[tid + 256 * r] = acc[r];
    }
    __syncthreads();
    // ---- 4. de-emphasis + notch over [ds, e) on wave 0, emitting [s, e) ----
    double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    if (dn_exact) { a0 = si->dn[0][0]; a1 = si->dn[0][1]; b0 = si->dn[1][0]; b1 = si->dn[1][1]; }
    float2 *y = out + (long long)c * out_pitch;
    for (long long off = ds; off < e; off += kSub) {
        const int nv = (int)((e - off) < kSub ? (e - off) : kSub);
        for (int j = tid; j < nv; j += 256) re[spad(j)] = fbuf[off - ds + j];
        __syncthreads();
        if (wave == 0) {
            scan_sub(wp.dn[0], re, nv, a0, a1, lane);
            scan_sub(wp.dn[1], re, nv, b0, b1, lane);
        }
        __syncthreads();
        if (off + nv > s) {
            for (int j = tid; j < nv; j += 256) {
                const long long idx = off + j;
                if (idx >= s) {
                    const float v = re[spad(j)];
                    y[idx] = make_float2(v, v);
                }
            }
        }
        __syncthreads();
    }
    // ---- 5. the block that reaches the end of the call leaves the exact state for the next one ----
    if (last_block) {
        WfmState *so = st_out + c;
        if (lane == 0 && wave < 2) { so->lp[wave][0] = q0; so->lp[wave][1] = q1; }
        if (tid == 0) {
            so->dn[0][0] = a0; so->dn[0][1] = a1; so->dn[1][0] = b0; so->dn[1][1] = b1;
            so->lp_last = lpbuf[(n - 1) - lk];
        }
        for (int j = tid; j < T - 1; j += 256) so->dtail[j] = dbuf[(n - (T - 1) + j) - d0];
    }
}

// The PLL demodulators -- Demod_NFM::processBlockNCO (application/demod/demod_nfm.cpp:225-257) and Demod_SAM::pll /
// processBlock (demod_sam.cpp:41-101) -- are NON-linear feedback loops: serial in time, parallel only across channels.
// One lane per channel walks the call; the loop state is `float` exactly as the reference declares it
// (demod_nfm.h:27-40, demod_sam.h:19-25), trigonometry of the float phase uses the float functions (what sin(float)
// resolves to in C++), everything else is double as written.  Output goes to a buffer with FIR head-room; the
// CFir that follows is k_fir_dec.
static __global__ __launch_bounds__(64) void k_pll_demod(const float2 *__restrict__ in, long long in_pitch, float2 *__restrict__ out,
                                                         long long out_pitch, long long n, PllParams pp, PllState *__restrict__ st,
                                                         const int *__restrict__ chan_list, int nlist)
{
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nlist) return;
    const int c = chan_list ? chan_list[li] : li;
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

}  // namespace pg
