// kernels_fastfir.h -- CFastFIR::ProcessData (pebblelib/fastfir.cpp:281-334) as batched overlap-save.
//
// One workgroup per (block, channel): gather N = L + (taps-1) samples [overlap | L new] straight into
// the strided register layout, forward FFT, multiply by the channel's H (1/N already folded into the
// taps, fastfir.cpp:244-245), inverse FFT, keep samples taps-1 .. N-1.  Forward-last and inverse-first
// passes share the register layout (fft_lds.h), so the product needs no exchange.
//
// The input buffer carries taps-1 samples of head-room holding the previous call's tail, which is
// exactly m_pFFTOverlapBuf (zero before the first call, fastfir.cpp:104-105).  A caller-owned buffer
// without head-room passes `tail` instead ([channel][taps-1], refreshed by the host side after the call).
//
// Bound: HBM.  Algorithmic bytes per demod-rate sample: 8 read + 8 written (+ H once per block: with
// taps-1 = N/2 the overlap doubles the read to 16 B unless L2 serves the second touch).
#pragma once
#include "fft_lds.h"
#include "fft_t128.h"

namespace pg {

template <int N>
__global__ __launch_bounds__(256) void k_fastfir(const float2 *__restrict__ in, long long in_pitch,
                                                  float2 *__restrict__ out, long long out_pitch,
                                                  const float2 *__restrict__ H, const float2 *__restrict__ tw,
                                                  int overlap /* taps-1 */, const float2 *__restrict__ tail)
{
    constexpr int E = N / 256;
    __shared__ float2 lds[FftLds<N>::kSlots];
    const int tid = threadIdx.x, c = blockIdx.y;
    const int L = N - overlap;
    const long long b = blockIdx.x;
    const float2 *x = in + (long long)c * in_pitch + b * L - overlap;  // first sample of [overlap | new]
    const float2 *h = H + (long long)c * N;
    float2 v[E];
    if (tail != nullptr && b == 0) {  // caller's buffer has no head-room: the overlap of block 0 lives in `tail` [c][overlap]
        const float2 *t = tail + (long long)c * overlap;
#pragma unroll
        for (int m = 0; m < E; m++) {
            const int i = tid + 256 * m;
            v[m] = i < overlap ? t[i] : x[i];
        }
    } else {
#pragma unroll
        for (int m = 0; m < E; m++) v[m] = x[tid + 256 * m];
    }
    fft_regs<N, +1>(v, lds, tw, tid);
#pragma unroll
    for (int m = 0; m < E; m++) v[m] = cmul(h[tid + 256 * m], v[m]);  // CpxMpy, fastfir.cpp:325-334
    fft_regs<N, -1>(v, lds, tw, tid);
    float2 *y = out + (long long)c * out_pitch + b * L;
#pragma unroll
    for (int m = 0; m < E; m++) {
        const int i = tid + 256 * m;
        if (i >= overlap) y[i - overlap] = v[m];
    }
}

// The 2048/1025 case (the reference's stock sizes) on the two-wave transform: a 128-item workgroup per block, 16 points
// per work-item, barriers that only involve its own two waves.  The inverse transform is the forward one between two
// conjugations.  grid (n / L, channels), block 128.
template <bool TWLDS = true>
static __global__ __launch_bounds__(128) void k_fastfir_t128(const float2 *__restrict__ in, long long in_pitch,
                                                             float2 *__restrict__ out, long long out_pitch,
                                                             const float2 *__restrict__ H, const float2 *__restrict__ tw128,
                                                             int overlap /* taps-1 */, const float2 *__restrict__ tail,
                                                             float2 *__restrict__ tail_out /* or null: [c][overlap] receives the call's last `overlap` input samples */,
                                                             int nb, int nchan)
{
    constexpr int N = 2048, E = 16;
    __shared__ float2 lds[FftLds<N>::kSlots];
    __shared__ float2 tw_lds_[TWLDS ? kTw128Count : 1];
    const float2 *tw_lds = TWLDS ? tw_lds_ : tw128;  // (A/B: the twiddle table read through the vector cache instead of a copy per workgroup)
    const int t = threadIdx.x;
    const int L = N - overlap;
    // Workgroup -> (block, channel).  A block's window repeats the `overlap` samples in front of it, which are the previous block's new
    // samples, and every block of a channel reads the channel's 16 KiB of H: with (block, channel) = (blockIdx.x, blockIdx.y) consecutive
    // blocks go to consecutive XCDs -- eight separate L2s -- and both come from HBM again.  nchan > 0 (a one-dimensional grid): the
    // workgroup id's low three bits are the XCD; each XCD walks ITS eighth of the blocks in (channel, block) order, so a block's neighbour
    // and its H are in the same L2 a moment before it.
    int c;
    long long b;
    if (nchan > 0) {
        const long long total = (long long)nb * nchan, per = (total + 7) >> 3;
        const long long q = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if ((long long)(blockIdx.x >> 3) >= per || q >= total) return;  // (uniform: the whole workgroup)
        c = (int)(q / nb);
        b = q - (long long)c * nb;
    } else {
        c = blockIdx.y;
        b = blockIdx.x;
        nb = (int)gridDim.x;
    }
    const float2 *x = in + (long long)c * in_pitch + b * L - overlap;  // first sample of [overlap | new]
    const float2 *h = H + (long long)c * N;
    if (TWLDS) for (int i = t; i < kTw128Count; i += 128) tw_lds_[i] = tw128[i];
    float2 v[E];
    if (tail != nullptr && b == 0) {
        const float2 *tl = tail + (long long)c * overlap;
#pragma unroll
        for (int m = 0; m < E; m++) {
            const int i = t + 128 * m;
            v[m] = i < overlap ? tl[i] : x[i];
        }
    } else {
#pragma unroll
        for (int m = 0; m < E; m++) v[m] = x[t + 128 * m];
    }
    if (tail_out != nullptr && b == (long long)nb - 1) {
        // m_pFFTOverlapBuf for the next call (fastfir.cpp:312-316): the last block's window ends with them.  A different buffer
        // from `tail`, which block 0 of this launch may still be reading
        float2 *to = tail_out + (long long)c * overlap;
#pragma unroll
        for (int m = 0; m < E; m++) {
            const int i = t + 128 * m;
            if (i >= L) to[i - L] = v[m];
        }
    }
    __syncthreads();
    fft2048_t128(v, lds, tw_lds, t, [] { __syncthreads(); });
#pragma unroll
    for (int m = 0; m < E; m++) {
        const float2 p = cmul(h[t + 128 * m], v[m]);  // CpxMpy, fastfir.cpp:325-334
        v[m] = make_float2(p.x, -p.y);                // conj: IFFT(z) = conj(FFT(conj(z))), unscaled (1/N is in the taps)
    }
    fft2048_t128(v, lds, tw_lds, t, [] { __syncthreads(); });  // (the forward transform ends behind a barrier: the image is free)
    float2 *y = out + (long long)c * out_pitch + b * L;
#pragma unroll
    for (int m = 0; m < E; m++) {
        const int i = t + 128 * m;
        if (i >= overlap) y[i - overlap] = make_float2(v[m].x, -v[m].y);
    }
}

}  // namespace pg
