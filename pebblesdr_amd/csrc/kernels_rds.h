// kernels_rds.h -- the RDS branch of Demod_WFM::processDataStereo (application/demod/demod_wfm.cpp:296-357, 488-757) for the dmFMS
// channels of a WfmCore, in double throughout as the reference declares it: the wanted signal reaches the decoder only through the
// stop band of the Hilbert pair (CDownConvert negates the -57 kHz it is handed, downconvert.cpp:103-108, so the oscillator moves the
// multiplex UP and the decoder lives on the image) and sits 60-70 dB under the multiplex -- single precision in front of the
// down-converter would leave it three digits.  Sample-parallel stages (discriminator, Hilbert pair + oscillator, the decimate-by-2
// chain, the 2400 Hz low-pass, the matched filter) are ordinary grids over (sample, channel); the two feedback loops (the PLL; the
// resonator, bit slicer and block synchroniser) are one lane per channel.  All rows are 8-byte elements in HistBuf rows: a row of
// double has the element count of a float2 row, a row of double2 twice that (pitch_d2 = pitch / 2).
#pragma once
#include "params.h"

namespace pg {

constexpr int kRdsAmpTab = 1024;  // the oscillator's amplitude has converged to sqrt(0.95) in double long before (ratio -0.9 per sample)

// m_RawFm, :258-263: raw[i] = FMDEMOD_GAIN atan2(...) of consecutive samples; row `raw` has the Hilbert pair's look-back in front
static __global__ __launch_bounds__(256) void k_rds_discrim(const float2 *__restrict__ in, long long in_pitch, long long n, const RdsState *__restrict__ st,
                                                            double *__restrict__ raw, long long raw_pitch, const int *__restrict__ chan_list)
{
    const int c = chan_list[blockIdx.y];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 *x = in + (long long)c * in_pitch;
    const double xr = (double)x[i].x, xi = (double)x[i].y;
    double pr, pi;
    if (i == 0) { pr = st[c].prev_re; pi = st[c].prev_im; }
    else { pr = (double)x[i - 1].x; pi = (double)x[i - 1].y; }
    raw[(long long)c * raw_pitch + i] = 0.25 * atan2(pr * xi - xr * pi, pr * xr + pi * xi);
}

// m_HilbertFilter.ProcessFilter (real in, complex out; fir.cpp:143-170) and the in-place product of CDownConvert::ProcessData
// (downconvert.cpp:283-310): osc_n = a_n e^{j (n + 1) inc}, a_n the amplitude recurrence's value before sample n
static __global__ __launch_bounds__(256) void k_rds_hilbert_mix(const double *__restrict__ raw, long long raw_pitch, long long n, const double *__restrict__ hilb,
                                                                const double *__restrict__ amp, const RdsState *__restrict__ st, double osc_turns,
                                                                double2 *__restrict__ mix, long long mix_pitch, const int *__restrict__ chan_list)
{
    __shared__ double hI[61], hQ[61];
    if (threadIdx.x < 61) { hI[threadIdx.x] = hilb[threadIdx.x]; hQ[threadIdx.x] = hilb[61 + threadIdx.x]; }
    __syncthreads();
    const int c = chan_list[blockIdx.y];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *r = raw + (long long)c * raw_pitch + i;
    double yr = 0.0, yi = 0.0;
#pragma unroll 4
    for (int k = 0; k < 61; k++) {
        const double v = r[-k];
        yr = fma(hI[k], v, yr);
        yi = fma(hQ[k], v, yi);
    }
    const long long m = st[c].n0 + i;
    // (m + 1) inc in turns: the integer part of the product is split off exactly (osc_turns < 1, m + 1 < 2^52)
    const double t = (double)(m + 1) * osc_turns;
    const double lo = fma((double)(m + 1), osc_turns, -t);
    double fr = (t - floor(t)) + lo;
    double sn, cs;
    sincospi(2.0 * fr, &sn, &cs);
    const double a = amp[m < kRdsAmpTab ? m : kRdsAmpTab - 1];
    const double orr = a * cs, oi = a * sn;
    mix[(long long)c * mix_pitch + i] = make_double2(yr * orr - yi * oi, yr * oi + yi * orr);
}

// one decimate-by-2 stage of CDownConvert (or, stride 1, the complex CFir behind it): y[m] = sum_k h[k] x[stride m + newest - (T - 1) + k],
// h oldest sample first, `newest` 1 for the CIC3 (its last tap is the pair's odd sample, downconvert.cpp:517-533), else 0
static __global__ __launch_bounds__(256) void k_rds_fir(const double2 *__restrict__ in, long long in_pitch, double2 *__restrict__ out, long long out_pitch,
                                                        long long n_out, int stride, int newest, const double *__restrict__ h, int T,
                                                        const int *__restrict__ chan_list)
{
    __shared__ double hs[80];
    if ((int)threadIdx.x < T) hs[threadIdx.x] = h[threadIdx.x];
    __syncthreads();
    const int c = chan_list[blockIdx.y];
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= n_out) return;
    const double2 *x = in + (long long)c * in_pitch + (long long)stride * m + newest - (T - 1);
    double ar = 0.0, ai = 0.0;
    for (int k = 0; k < T; k++) {
        const double2 v = x[k];
        ar = fma(hs[k], v.x, ar);
        ai = fma(hs[k], v.y, ai);
    }
    out[(long long)c * out_pitch + m] = make_double2(ar, ai);
}

__device__ __forceinline__ double rds_arctan2(double y, double x)  // Demod_WFM::arctan2, :792-821, constants as written
{
    const double kTwoPi = 6.28318530717958647692528676656, kPiD = 3.14159265358979323846;
    if (x == 0.0) return y > 0.0 ? kTwoPi : (y == 0.0 ? 0.0 : -kTwoPi);
    const double z = y / x;
    double ang;
    if (fabs(z) < 1.0) {
        ang = z / (1.0 + 0.2854 * z * z);
        if (x < 0.0) ang = y < 0.0 ? ang - kPiD : ang + kPiD;
    } else {
        ang = kTwoPi - z / (z * z + 0.2854);
        if (y < 0.0) ang -= kPiD;
    }
    return ang;
}

// processRdsPll, :542-569: one lane per channel; mag[i] = the de-rotated sample's imaginary part; the phase is folded at the end of
// every block of pp.block samples as the reference folds it at the end of every call
static __global__ __launch_bounds__(64) void k_rds_pll(const double2 *__restrict__ lp, long long lp_pitch, long long len, RdsParams pp, RdsState *__restrict__ st,
                                                       double *__restrict__ mag, long long mag_pitch, const int *__restrict__ chan_list, int nlist)
{
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nlist) return;
    const int c = chan_list[li];
    RdsState *s = &st[c];
    const double kTwoPi = 6.28318530717958647692528676656;
    const double2 *x = lp + (long long)c * lp_pitch;
    double *m = mag + (long long)c * mag_pitch;
    double ph = s->nco_phase, fq = s->nco_freq;
    for (long long b0 = 0; b0 < len; b0 += pp.block) {
        const long long be = b0 + pp.block < len ? b0 + pp.block : len;
        for (long long i = b0; i < be; i++) {
            const double sn = sin(ph), cs = cos(ph);
            const double2 v = x[i];
            const double tr = cs * v.x - sn * v.y, ti = cs * v.y + sn * v.x;
            const double err = -rds_arctan2(ti, tr);
            fq += pp.beta * err;
            if (fq > pp.nco_hi) fq = pp.nco_hi;
            else if (fq < pp.nco_lo) fq = pp.nco_lo;
            ph += fq + pp.alpha * err;
            m[i] = ti;
        }
        ph = fmod(ph, kTwoPi);
    }
    s->nco_phase = ph;
    s->nco_freq = fq;
}

// m_RdsMatchedFilter.ProcessFilter (real; fir.cpp:77-95): data[i] = sum_k h[k] mag[i - k]
static __global__ __launch_bounds__(256) void k_rds_matched(const double *__restrict__ mag, long long mag_pitch, long long len, const double *__restrict__ h, int T,
                                                            double *__restrict__ data, long long data_pitch, const int *__restrict__ chan_list)
{
    __shared__ double hs[80];
    if ((int)threadIdx.x < T) hs[threadIdx.x] = h[threadIdx.x];
    __syncthreads();
    const int c = chan_list[blockIdx.y];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    const double *x = mag + (long long)c * mag_pitch + i;
    double acc = 0.0;
    for (int k = 0; k < T; k++) acc = fma(hs[k], x[-k], acc);
    data[(long long)c * data_pitch + i] = acc;
}

// Demod_WFM::checkBlock, :708-757: syndrome of the last 26 bits against a block's offset; with use_fec the Meggitt decoder corrects a
// burst of up to five bits in place
__device__ inline unsigned rds_check_block(unsigned &in_bits, unsigned offset, bool use_fec)
{
    const unsigned parckh[16] = {0x2DC, 0x16E, 0x0B7, 0x287, 0x39F, 0x313, 0x355, 0x376, 0x1BB, 0x201, 0x3DC, 0x1EE, 0x0F7, 0x2A7, 0x38F, 0x31B};
    unsigned test = in_bits & 0x3FFFFFFu;
    unsigned syn = test >> 16;
    for (int i = 0; i < 16; i++) {
        if (test & 0x8000u) syn ^= parckh[i];
        test <<= 1;
    }
    syn ^= offset;
    if (syn && use_fec) {
        unsigned mask = 1u << 25;
        for (int i = 0; i < 16; i++) {
            if (syn & 0x200u) {
                if ((syn & 0x1Fu) == 0) {
                    in_bits ^= mask;
                    syn <<= 1;
                } else {
                    syn = (syn << 1) ^ 0x5B9u;
                }
            } else {
                syn <<= 1;
            }
            mask >>= 1;
        }
        syn &= 0x3FFu;
    }
    return syn;
}

__device__ inline void rds_log(RdsState *s, RdsEvent *log, int log_cap, unsigned short a, unsigned short b, unsigned short c, unsigned short d, unsigned flags,
                               long long frame)
{
    RdsEvent e;
    e.a = a; e.b = b; e.c = c; e.d = d; e.flags = flags; e.frame = frame;
    log[s->n_events % (unsigned long long)log_cap] = e;
    s->n_events++;
}

// processNewRdsBit, :576-700
__device__ inline void rds_new_bit(RdsState *s, int bit, RdsEvent *log, int log_cap, long long frame)
{
    const unsigned offs[8] = {0x3D8, 0x3D4, 0x25C, 0x258, 0x3D8, 0x3D4, 0x3CC, 0x258};  // BLK_OFFSET_TBL
    s->in_bits = (s->in_bits << 1) | (unsigned)bit;
    if (s->state == 0) {  // STATE_BITSYNC: every position until a block A checks out
        if (rds_check_block(s->in_bits, 0x3D8, false) == 0) {
            s->bit_pos = 0;
            s->bgroup = 0;
            s->block[0] = (unsigned short)(s->in_bits >> 10);
            s->cur_block = 1;
            s->state = 1;
        }
        return;
    }
    s->bit_pos++;
    if (s->bit_pos < 26) return;
    s->bit_pos = 0;
    if (s->state == 3) {  // STATE_GROUPRESYNC: skip to the next group
        s->cur_block++;
        if (s->cur_block > 3) {
            s->cur_block = 0;
            s->state = 2;
        }
        return;
    }
    const bool decode = s->state == 2;  // STATE_GROUPDECODE (with FEC) against STATE_BLOCKSYNC (without)
    if (rds_check_block(s->in_bits, offs[s->cur_block + s->bgroup], decode)) {
        if (!decode) {
            s->state = 0;
            return;
        }
        s->block_errors++;
        if (s->block_errors > 5) {  // BLOCK_ERROR_LIMIT: the queue is cleared and a zero group stuffed in
            rds_log(s, log, log_cap, 0, 0, 0, 0, 1u, frame);
            s->state = 0;
        } else {
            s->cur_block++;
            if (s->cur_block > 3) s->cur_block = 0;
            if (s->cur_block != 0) s->state = 3;
        }
        return;
    }
    s->block[s->cur_block] = (unsigned short)(s->in_bits >> 10);
    s->bgroup = (s->cur_block == 1 && (s->block[1] & 0x0800)) ? 4 : 0;  // GROUPB_BIT
    if (s->cur_block >= 3) {
        rds_log(s, log, log_cap, s->block[0], s->block[1], s->block[2], s->block[3], 0u, frame);
        s->cur_block = 0;
        s->block_errors = 0;
        s->state = 2;
    } else {
        s->cur_block++;
    }
}

// :312-353: the squared data through the bit-rate resonator (CIir real, iir.cpp:173-182), a bit at every positive peak of it (the
// sample before), differential decoding, the block synchroniser; also closes the call: the discriminator's previous sample, the
// oscillator's sample count, the frame count
static __global__ __launch_bounds__(64) void k_rds_bits(const double *__restrict__ data, long long data_pitch, long long len, RdsParams pp, RdsState *__restrict__ st,
                                                        RdsEvent *__restrict__ logs, const float2 *__restrict__ in, long long in_pitch, long long n,
                                                        const int *__restrict__ chan_list, int nlist)
{
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nlist) return;
    const int c = chan_list[li];
    RdsState *s = &st[c];
    RdsEvent *log = logs + (long long)c * pp.log_cap;
    const double *d = data + (long long)c * data_pitch;
    double w1 = s->w1, w2 = s->w2, last_sync = s->last_sync, last_slope = s->last_slope, last_data = s->last_data;
    int last_bit = s->last_bit;
    long long frame = s->frames;
    for (long long b0 = 0; b0 < len; b0 += pp.block, frame++) {
        const long long be = b0 + pp.block < len ? b0 + pp.block : len;
        for (long long i = b0; i < be; i++) {
            const double v = d[i];
            const double w0 = v * v - pp.a1 * w1 - pp.a2 * w2;
            const double sync = pp.b0 * w0 + pp.b2 * w2;
            w2 = w1; w1 = w0;
            const double slope = sync - last_sync;
            last_sync = sync;
            if (slope < 0.0 && last_slope * slope < 0.0) {
                const int bit = last_data >= 0 ? 1 : 0;
                rds_new_bit(s, bit ^ last_bit, log, pp.log_cap, frame);
                last_bit = bit;
            }
            last_data = v;
            last_slope = slope;
        }
    }
    s->w1 = w1; s->w2 = w2; s->last_sync = last_sync; s->last_slope = last_slope; s->last_data = last_data;
    s->last_bit = last_bit;
    s->frames = frame;
    s->n0 += n;
    const float2 v = in[(long long)c * in_pitch + n - 1];
    s->prev_re = (double)v.x;
    s->prev_im = (double)v.y;
}

}  // namespace pg
