// kernels_spectrum.h -- SignalSpectrum::unprocessed / FFT::fftSpectrum on the device.
//
// Per frame of NF = 2048 samples (framesPerBuffer): BlackmanHarris window, zero-pad to
// bins = ZP*NF, forward FFT, unfold to -f..+f, |X|/(coherentGain*NF), average with the PREVIOUS
// frame's linear amplitude, 20*log10, clip to [-120, 0] dB
// (pebblelib/fft.cpp:129-157, 207-213, 324-399; fftaccelerate.cpp:106-119).
//
// Zero-padding is pruned instead of transformed: with x[n] = 0 for n >= NF,
//   X[ZP*j + q] = FFT_NF( x[n] * W_bins^{n*q} )[j],  q = 0..ZP-1
// i.e. ZP independent NF-point transforms of the twiddled frame -- log2(ZP) butterfly passes over
// zeros are never executed, and each work-item ends up holding ZP adjacent bins (one vector store).
//
// A workgroup walks G consecutive frames of one stream so the previous frame's amplitudes stay in
// registers; it recomputes one extra frame (the one before its first) for the average.
//
// Bound: HBM first (8*NF B in, 4*bins B out per frame), fp32 ALU/LDS close behind at ZP = 4
// (about 0.5 Mflop per frame).  Algorithmic bytes per frame: 8*NF + 4*bins.
#pragma once
#include "fft_lds.h"
#include "params.h"

namespace pg {


template <int ZP>
__global__ __launch_bounds__(256) void k_spectrum(const float2 *__restrict__ in, float *__restrict__ out,
                                                   const float *__restrict__ window, const float2 *__restrict__ tw_nf,
                                                   const float2 *__restrict__ tw_bins,
                                                   const float *__restrict__ prev_in, float *__restrict__ prev_out,
                                                   SpectrumParams sp)
{
    constexpr int NF = 2048, E = NF / 256, BINS = NF * ZP;
    __shared__ float2 lds[FftLds<NF>::kSlots];
    const int tid = threadIdx.x, s = blockIdx.y;
    const long long f0 = (long long)blockIdx.x * sp.frames_per_group;
    long long f1 = f0 + sp.frames_per_group;
    if (f1 > sp.n_frames) f1 = sp.n_frames;
    const float2 *x = in + (long long)s * sp.in_pitch;
    float *y = out + (long long)s * sp.out_pitch;

    float w[E];
#pragma unroll
    for (int m = 0; m < E; m++) w[m] = window[tid + 256 * m];

    float pa[ZP][E];  // previous frame's linear amplitudes, element (q, m) <-> bin ZP*(tid+256m)+q
    for (long long f = f0 - 1; f < f1; f++) {
        float amp[ZP][E];
        if (f < 0) {
            // frame before the call: amplitudes saved by the previous call (zeros on the first)
#pragma unroll
            for (int q = 0; q < ZP; q++)
#pragma unroll
                for (int m = 0; m < E; m++) pa[q][m] = prev_in[(long long)s * BINS + ZP * (tid + 256 * m) + q];
            continue;
        }
        float2 xin[E];
#pragma unroll
        for (int m = 0; m < E; m++) xin[m] = cscale(x[f * NF + tid + 256 * m], w[m]);
#pragma unroll
        for (int q = 0; q < ZP; q++) {
            float2 v[E];
#pragma unroll
            for (int m = 0; m < E; m++) {
                if (q == 0) v[m] = xin[m];
                else v[m] = cmul(xin[m], tw_bins[(tid + 256 * m) * q]);  // W_bins^{n q}, n*q < bins
            }
            fft_regs<NF, +1>(v, lds, tw_nf, tid);
#pragma unroll
            for (int m = 0; m < E; m++) amp[q][m] = sqrtf(v[m].x * v[m].x + v[m].y * v[m].y) * sp.scale;
        }
        if (f >= f0) {
            float *yf = y + f * (long long)BINS;
#pragma unroll
            for (int m = 0; m < E; m++) {
                float db[ZP];
#pragma unroll
                for (int q = 0; q < ZP; q++) {
                    const float a = 0.5f * (amp[q][m] + pa[q][m]);   // fft.cpp:379-381
                    float d = a == 0.f ? -120.f : 20.f * log10f(a);   // db.h:44-48
                    d = fminf(fmaxf(d, -120.f), 0.f);                 // db.h:24-26
                    db[q] = d;
                }
                // bin k = ZP*j+q unfolds to (k + BINS/2) mod BINS (fft.cpp:207-213); ZP adjacent bins stay adjacent
                const int k = ZP * (tid + 256 * m);
                const int u = (k + BINS / 2) & (BINS - 1);
                if (ZP == 4) *reinterpret_cast<float4 *>(yf + u) = make_float4(db[0], db[1], db[2], db[3]);
                else if (ZP == 2) *reinterpret_cast<float2 *>(yf + u) = make_float2(db[0], db[1]);
                else
#pragma unroll
                    for (int q = 0; q < ZP; q++) yf[u + q] = db[q];
            }
        }
#pragma unroll
        for (int q = 0; q < ZP; q++)
#pragma unroll
            for (int m = 0; m < E; m++) pa[q][m] = amp[q][m];
        if (f == sp.n_frames - 1) {
#pragma unroll
            for (int q = 0; q < ZP; q++)
#pragma unroll
                for (int m = 0; m < E; m++) prev_out[(long long)s * BINS + ZP * (tid + 256 * m) + q] = amp[q][m];
        }
    }
}

}  // namespace pg
