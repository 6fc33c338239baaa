// DEVELOPER-ONLY micro-benchmark: the one-wave 2048-point transform of fft_w64.h (radix 32 * 4 * 16, the middle pass on
// v_permlane{16,32}_swap, one LDS exchange) -- checked against a host DFT of the zero-padded frame (8192 bins as four
// pre-twiddled transforms, the way the spectrum kernel uses it), then timed bare at several occupancies next to its
// exchange alone and its arithmetic alone.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipebblesdr_amd/csrc tools/ubench/fft_w64.hip -o gpurun_out/fft_w64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "fft_w64.h"
using namespace pg;

// stab[q][n1] = W_128^{n1 q} (wave-uniform: scalar loads)
template <bool DO_LDS, bool DO_MATH>
__global__ __launch_bounds__(256) void k_core(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ stab, int reps, int check)
{
    __shared__ float2 img[4][kW64ImageSlots];
    extern __shared__ char pad[];
    const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const W64Consts c = w64_consts(lane, cis_cycles(-(double)(lane * q) / 8192.0));
    float2 v[32];
    const float2 *src = in + (size_t)blockIdx.x * 2048;
#pragma unroll
    for (int m = 0; m < 32; m++) v[m] = src[lane + 64 * m];
    for (int r = 0; r < reps; r++) {
        int ll = lane, qq = q;
        asm volatile("" : "+v"(ll));
        asm volatile("" : "+s"(qq));
        const float2 *sq = stab + 32 * qq;
        if (DO_MATH) {
#pragma unroll
            for (int m = 1; m < 32; m++) v[m] = cmul_pk(sq[m], v[m]);
        }
        fft2048_w64<DO_LDS, DO_MATH>(v, img[q], c, ll);
        if (!check) {
#pragma unroll
            for (int m = 0; m < 32; m++) v[m] = cscale(v[m], 1.0f / 64.0f);
        }
    }
    if (check) {
        // bin 4 k + q of the 8192-bin spectrum
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int k3 = 0; k3 < 16; k3++) out[(size_t)blockIdx.x * 8192 + 4 * (w64_kbase(lane) + 32 * h + 128 * k3) + q] = v[16 * h + perm16(k3)];
    } else {
#pragma unroll
        for (int m = 0; m < 32; m++) out[(size_t)blockIdx.x * 8192 + q * 2048 + lane + 64 * m] = v[m];
    }
    if (pad[0] == 77) out[0] = v[0];
}

__global__ void k_swap_probe(unsigned *o)
{
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + threadIdx.x] = s[0];
    o[192 + threadIdx.x] = s[1];
}

int main()
{
    const int WG = 1024, reps = 64;
    const double PI2 = 6.283185307179586476925286766559;
    std::vector<float2> h((size_t)WG * 2048), st(4 * 32);
    unsigned seed = 12345;
    for (size_t i = 0; i < h.size(); i++) {
        seed = seed * 1664525u + 1013904223u;
        const float a = (float)((seed >> 8) & 0xffff) / 65536.0f - 0.5f;
        seed = seed * 1664525u + 1013904223u;
        const float b = (float)((seed >> 8) & 0xffff) / 65536.0f - 0.5f;
        h[i] = make_float2(a, b);
    }
    for (int q = 0; q < 4; q++)
        for (int n1 = 0; n1 < 32; n1++) st[q * 32 + n1] = make_float2((float)cos(-PI2 * n1 * q / 128.0), (float)sin(-PI2 * n1 * q / 128.0));
    float2 *d_in, *d_out, *d_st;
    unsigned *d_probe;
    hipMalloc(&d_in, h.size() * 8);
    hipMalloc(&d_out, (size_t)WG * 8192 * 8);
    hipMalloc(&d_st, st.size() * 8);
    hipMalloc(&d_probe, 256 * 4);
    hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_st, st.data(), st.size() * 8, hipMemcpyHostToDevice);

    // ---- swap semantics ----
    {
        hipLaunchKernelGGL(k_swap_probe, dim3(1), dim3(64), 0, 0, d_probe);
        std::vector<unsigned> p(256);
        hipMemcpy(p.data(), d_probe, 1024, hipMemcpyDeviceToHost);
        bool ok = true;
        for (int l = 0; l < 64; l++) {
            const unsigned e0 = l < 32 ? l : 100 + (l - 32), e1 = l < 32 ? 32 + l : 100 + l;
            ok = ok && p[l] == e0 && p[64 + l] == e1;
            const int row = l >> 4, c = l & 15;
            const unsigned f0 = (row & 1) ? 100 + 16 * (row - 1) + c : l, f1 = (row & 1) ? 100 + l : 16 * (row + 1) + c;
            ok = ok && p[128 + l] == f0 && p[192 + l] == f1;
        }
        printf("permlane swap semantics: %s\n", ok ? "as assumed" : "DIFFERENT");
        if (!ok) {
            for (int l = 0; l < 64; l += 8) printf("  lane %2d: swap32 -> %3u %3u   swap16 -> %3u %3u\n", l, p[l], p[64 + l], p[128 + l], p[192 + l]);
        }
    }
    // ---- correctness: frame 0, 8192 bins ----
    {
        hipLaunchKernelGGL((k_core<true, true>), dim3(1), dim3(256), 0, 0, d_in, d_out, d_st, 1, 1);
        std::vector<float2> got(8192);
        hipMemcpy(got.data(), d_out, 8192 * 8, hipMemcpyDeviceToHost);
        std::vector<double> cs(8192), sn(8192);
        for (int i = 0; i < 8192; i++) { cs[i] = cos(PI2 * i / 8192.0); sn[i] = -sin(PI2 * i / 8192.0); }
        double err = 0, ref = 0, worst = 0;
        for (int k = 0; k < 8192; k++) {
            double re = 0, im = 0;
            for (int n = 0; n < 2048; n++) {
                const int m = (int)(((long long)n * k) & 8191);
                re += h[n].x * cs[m] - h[n].y * sn[m];
                im += h[n].x * sn[m] + h[n].y * cs[m];
            }
            const double dx = got[k].x - re, dy = got[k].y - im;
            err += dx * dx + dy * dy;
            ref += re * re + im * im;
            worst = fmax(worst, sqrt(dx * dx + dy * dy));
        }
        printf("8192 bins against the host DFT: relative RMS error %.3e, worst bin %.3e (rms bin %.3e)\n", sqrt(err / ref), worst, sqrt(ref / 8192));
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name, size_t padlds, int per_cu) {
        float best = 1e9f;
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)padlds);
        for (int it = 0; it < 5; it++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(WG), dim3(256), padlds, 0, d_in, d_out, d_st, reps, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double ffts = (double)WG * 4 * reps;
        printf("%-28s %d waves/CU: %.3f ms for %.0f transforms -> %.2f ns each chip-wide (the two-wave transform: 2.42)\n", name, per_cu, best, ffts, best * 1e6 / ffts);
    };
    // 256-item workgroups with 34 KiB of images: pad the LDS so that 4 / 3 / 2 / 1 of them fit a CU
    for (size_t pad : {(size_t)0, (size_t)17000, (size_t)40000, (size_t)90000}) {
        const int per_cu = pad == 0 ? 16 : pad == 17000 ? 12 : pad == 40000 ? 8 : 4;
        run(k_core<true, true>, "whole transform", pad, per_cu);
        run(k_core<true, false>, "exchange + swaps only", pad, per_cu);
        run(k_core<false, true>, "arithmetic only", pad, per_cu);
    }
    return 0;
}
