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

// Fused mixer + first decimation stage.  One work-item per output sample o:
//   halfband:  y[o] = sum_p m[o*S + p - (T-1)] * h[p]          (decimator.cpp:637-648, vDSP_zrdesampD)
//   CIC3:      y[o] = .125*(m[oS+1] + m[(o-1)S] + 3*(m[(o-1)S+1] + m[oS]))   (decimator.cpp:727-731)
// with m[i] = osc(i)*x[i] for i >= 0 and m[i] = hist[T-1+i] (mixed samples kept from the previous call).
// Only taps that are non-zero are mixed at all, so for stride > taps most input samples cost nothing.
// grid (ceil(n_out/256), C).
static __global__ __launch_bounds__(256) void k_mix_dec1(const float2 *__restrict__ in, long long in_pitch, int shared_input,
                                                   float2 *__restrict__ out, long long out_pitch, long long n_out,
                                                   const ChanOsc *__restrict__ osc, const float2 *__restrict__ hist,
                                                   int hist_pitch, const float *__restrict__ amp_tab, float a_inf,
                                                   FirTaps taps)
{
    const int c = blockIdx.y;
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const ChanOsc *oc = &osc[c];
    const float2 *x = in + (shared_input ? 0 : (long long)c * in_pitch);
    const float2 *hs = hist + (long long)c * hist_pitch;
    const int S = taps.stride;
    float2 acc;
    if (taps.cic3) {
        // history slots: hist[0] = m[-S] (even of the previous stride), hist[1] = m[-S+1] (odd)
        const long long ie = o * S;
        const float2 p0 = oc->mix_on ? cis_cycles(oc->phase0 + (double)(ie + 1) * oc->inc) : make_float2(1.f, 0.f);
        float2 ev = x[ie], od = x[ie + 1], pev, pod;
        if (oc->mix_on) {
            ev = cmul(cscale(p0, osc_amp(amp_tab, a_inf, oc->n0, ie)), ev);
            od = cmul(cscale(cmul(p0, oc->step[1]), osc_amp(amp_tab, a_inf, oc->n0, ie + 1)), od);
        }
        if (o == 0) {
            pev = hs[0];
            pod = hs[1];
        } else {
            const long long ip = ie - S;
            pev = x[ip];
            pod = x[ip + 1];
            if (oc->mix_on) {
                const float2 q0 = cis_cycles(oc->phase0 + (double)(ip + 1) * oc->inc);
                pev = cmul(cscale(q0, osc_amp(amp_tab, a_inf, oc->n0, ip)), pev);
                pod = cmul(cscale(cmul(q0, oc->step[1]), osc_amp(amp_tab, a_inf, oc->n0, ip + 1)), pod);
            }
        }
        acc.x = .125f * (od.x + pev.x + 3.0f * (pod.x + ev.x));
        acc.y = .125f * (od.y + pev.y + 3.0f * (pod.y + ev.y));
    } else {
        const int T = taps.ntaps;
        const long long i0 = o * S - (T - 1);           // time index of tap 0
        const long long ib = i0 < 0 ? 0 : i0;            // first index that comes from this call's input
        float2 p0 = make_float2(1.f, 0.f);
        if (oc->mix_on) p0 = cis_cycles(oc->phase0 + (double)(ib + 1) * oc->inc);
        acc = make_float2(0.f, 0.f);
        for (int p = 0; p < T; p++) {
            const float h = taps.h[p];
            if (h == 0.f) continue;  // halfband zeros: uniform branch, nothing to load or mix
            const long long i = i0 + p;
            float2 m;
            if (i < 0) {
                m = hs[(T - 1) + i];
            } else {
                m = x[i];
                if (oc->mix_on) {
                    const float2 ph = cmul(p0, oc->step[(int)(i - ib)]);
                    m = cmul(cscale(ph, osc_amp(amp_tab, a_inf, oc->n0, i)), m);
                }
            }
            acc.x = fmaf(m.x, h, acc.x);
            acc.y = fmaf(m.y, h, acc.y);
        }
    }
    out[(long long)c * out_pitch + o] = cscale(acc, taps.gain);
}

// Mixed-sample history for the next call: hist[c][j] = m[n - H + j], j < H (H = T-1, or for CIC3
// the pair m[n-S], m[n-S+1]).  grid (C), block 64.  Runs after k_mix_dec1 of the same call.
static __global__ __launch_bounds__(64) void k_mix_tail(const float2 *__restrict__ in, long long in_pitch, int shared_input,
                                                  long long n, const ChanOsc *__restrict__ osc, float2 *__restrict__ hist,
                                                  int hist_pitch, const float *__restrict__ amp_tab, float a_inf,
                                                  int ntaps, int stride, int cic3)
{
    const int c = blockIdx.x, j = threadIdx.x;
    const ChanOsc *oc = &osc[c];
    const float2 *x = in + (shared_input ? 0 : (long long)c * in_pitch);
    long long i;
    if (cic3) {
        if (j >= 2) return;
        i = n - stride + j;
    } else {
        if (j >= ntaps - 1) return;
        i = n - (ntaps - 1) + j;
    }
    float2 m = x[i];
    if (oc->mix_on) {
        const float2 ph = cis_cycles(oc->phase0 + (double)(i + 1) * oc->inc);
        m = cmul(cscale(ph, osc_amp(amp_tab, a_inf, oc->n0, i)), m);
    }
    hist[(long long)c * hist_pitch + j] = m;
}

// Generic real-tap FIR on complex data with decimation: later halfband stages (stride 2^k) and the
// CFir post-demod filters (stride 1).  `in` points at the data start; in[-(T-1)..-1] is history.
//   y[c][o] = gain * sum_p in[c][o*S + p - (T-1)] * h_c[p]
// taps: either one set for all channels (taps_pitch == 0) or per channel, in device memory.
// grid (ceil(n_out/256), C).
static __global__ __launch_bounds__(256) void k_fir_dec(const float2 *__restrict__ in, long long in_pitch,
                                                  float2 *__restrict__ out, long long out_pitch, long long n_out,
                                                  int stride, const float *__restrict__ taps, int taps_pitch,
                                                  const int *__restrict__ ntaps_per_chan, int ntaps_all, float gain,
                                                  const int *__restrict__ chan_list)
{
    const int c = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const int T = ntaps_per_chan ? ntaps_per_chan[c] : ntaps_all;
    const float *h = taps + (long long)c * taps_pitch;
    const float2 *x = in + (long long)c * in_pitch + o * stride - (T - 1);
    float2 acc = make_float2(0.f, 0.f);
    for (int p = 0; p < T; p++) {
        const float hp = h[p];
        if (hp == 0.f) continue;
        const float2 v = x[p];
        acc.x = fmaf(v.x, hp, acc.x);
        acc.y = fmaf(v.y, hp, acc.y);
    }
    out[(long long)c * out_pitch + o] = cscale(acc, gain);
}

// Copy the last `hist` samples of each channel's data into its head-room: buf[c][-hist + j] = buf[c][n - hist + j].
// Requires n >= hist (true for every buffer: calls are whole super-frames).  grid (ceil(hist/256), C).
static __global__ __launch_bounds__(256) void k_save_tail(float2 *__restrict__ data, long long pitch, long long n, int hist,
                                                    const int *__restrict__ chan_list)
{
    const int c = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= hist) return;
    float2 *b = data + (long long)c * pitch;
    b[-hist + j] = b[n - hist + j];
}

// All history tails of a call in one launch.  grid (ceil(max hist/256), C, jobs).
static __global__ __launch_bounds__(256) void k_save_tails(TailJobs jobs)
{
    const TailJob &t = jobs.job[blockIdx.z];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= t.hist) return;
    float2 *b = t.data + (long long)blockIdx.y * t.pitch;
    b[-t.hist + j] = b[t.n - t.hist + j];
}

}  // namespace pg
