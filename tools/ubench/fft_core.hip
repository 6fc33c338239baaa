// DEVELOPER-ONLY micro-benchmark: throughput of the two-wave 2048-point transform (fft_t128.h) with nothing around it --
// the whole transform, its LDS exchanges alone and its butterflies alone, at 8 / 4 / 2 workgroups of 128 per CU.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipebblesdr_amd/csrc tools/ubench/fft_core.hip -o gpurun_out/fft_core
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fft_t128.h"
using namespace pg;

template <bool DO_LDS, bool DO_MATH>
__global__ __launch_bounds__(128, 4) void k_core(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ tw128, int reps)
{
    __shared__ float2 lds[FftLds<2048>::kSlots];
    __shared__ float2 tw_lds[kTw128Count];
    extern __shared__ char pad[];
    const int t = threadIdx.x;
    for (int i = t; i < kTw128Count; i += 128) tw_lds[i] = tw128[i];
    float2 v[16];
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = in[(size_t)blockIdx.x * 2048 + t + 128 * m];
    __syncthreads();
    for (int r = 0; r < reps; r++) {
        int tt = t;
        asm volatile("" : "+v"(tt));
        auto sync = [] { __syncthreads(); };
        fft2048_t128<decltype(sync), DO_LDS, DO_MATH>(v, lds, tw_lds, tt, sync);
#pragma unroll
        for (int m = 0; m < 16; m++) v[m] = cscale(v[m], 1.0f / 64.0f);
    }
#pragma unroll
    for (int m = 0; m < 16; m++) out[(size_t)blockIdx.x * 2048 + t + 128 * m] = v[m];
    if (pad[0] == 77) out[0] = v[0];
}

int main()
{
    const int WG = 2048, reps = 64;
    std::vector<float2> h((size_t)WG * 2048), tw(kTw128Count);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_float2((float)(i % 97) * 0.01f, (float)(i % 89) * 0.01f);
    for (int k = 0; k < 16; k++) for (int e = 0; e < 3; e++) { double a = -6.283185307179586 * ((16 * k) << e) / 2048.0; tw[kTw128B + e * 16 + k] = make_float2((float)cos(a), (float)sin(a)); }
    for (int k = 0; k < 128; k++) for (int e = 0; e < 4; e++) { double a = -6.283185307179586 * ((long long)k << e) / 2048.0; tw[kTw128C + e * 128 + k] = make_float2((float)cos(a), (float)sin(a)); }
    float2 *d_in, *d_out, *d_tw;
    hipMalloc(&d_in, h.size() * 8); hipMalloc(&d_out, h.size() * 8); hipMalloc(&d_tw, tw.size() * 8);
    hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name, size_t padlds, int per_cu) {
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(WG), dim3(128), padlds, 0, d_in, d_out, d_tw, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double ffts = (double)WG * reps;
        printf("%-28s %d WG/CU: %.3f ms for %.0f transforms -> %.2f ns each chip-wide, %.0f CU-clk per transform at 2.1 GHz\n", name, per_cu, best, ffts, best * 1e6 / ffts,
               best * 1e-3 * 2.1e9 / (ffts / 256));
    };
    for (size_t pad : {(size_t)0, (size_t)20000, (size_t)60000}) {
        const int per_cu = pad == 0 ? 7 : pad == 20000 ? 3 : 1;
        run(k_core<true, true>, "whole transform", pad, per_cu);
        run(k_core<true, false>, "LDS exchanges only", pad, per_cu);
        run(k_core<false, true>, "butterflies only", pad, per_cu);
    }
    return 0;
}
