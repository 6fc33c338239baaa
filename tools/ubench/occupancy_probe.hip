// DEVELOPER-ONLY: do two workgroups of a kernel really share a CU's SIMDs at the same time?  Each wave records
// (start, end, HW_ID, XCC_ID) around a fixed chain of v_pk_fma_f32; the host prints overlap per (xcc, se, cu, simd).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

template <int LDS_BYTES>
__global__ __launch_bounds__(256) void k_probe(long long *out, float seed)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    __shared__ char lds[LDS_BYTES];
    lds[threadIdx.x] = (char)seed;
    __syncthreads();
    v2 a = {seed, seed + 1}, b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    long long t0 = clock64();
    for (int it = 0; it < 4096; it++)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2\nv_pk_fma_f32 %0, %0, %1, %2\nv_pk_fma_f32 %0, %0, %1, %2\nv_pk_fma_f32 %0, %0, %1, %2\n"
                     "v_pk_fma_f32 %0, %0, %1, %2\nv_pk_fma_f32 %0, %0, %1, %2\nv_pk_fma_f32 %0, %0, %1, %2\nv_pk_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(a) : "v"(b), "v"(c));
    long long t1 = clock64();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        long long *o = out + ((long long)blockIdx.x * 4 + threadIdx.x / 64) * 4;
        o[0] = t0; o[1] = t1; o[2] = hw; o[3] = xcc;
    }
    if (a.x == 12345.f) out[0] = lds[5];
}

template <int LDS_BYTES> static void run(int blocks, long long *d)
{
    hipLaunchKernelGGL(k_probe<LDS_BYTES>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_probe<LDS_BYTES>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)blocks * 16);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<long long, long long>>> per_simd;
    double mean = 0;
    long long tmin = h[0], tmax = h[1];
    for (int w = 0; w < blocks * 4; w++) {
        const long long t0 = h[w * 4], t1 = h[w * 4 + 1];
        const unsigned hw = (unsigned)h[w * 4 + 2], xcc = (unsigned)h[w * 4 + 3] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, se = (hw >> 13) & 7;
        per_simd[((unsigned long long)xcc << 24) | (se << 16) | (cu << 8) | simd].push_back({t0, t1});
        mean += (double)(t1 - t0);
        tmin = std::min(tmin, t0); tmax = std::max(tmax, t1);
    }
    mean /= blocks * 4;
    int maxw = 0, overl = 0, pairs = 0;
    for (auto &kv : per_simd) {
        auto &v = kv.second;
        maxw = std::max(maxw, (int)v.size());
        for (size_t i = 0; i < v.size(); i++)
            for (size_t j = i + 1; j < v.size(); j++) {
                pairs++;
                const long long lo = std::max(v[i].first, v[j].first), hi = std::min(v[i].second, v[j].second);
                if (hi - lo > (v[i].second - v[i].first) / 2) overl++;
            }
    }
    printf("LDS %6d B, %4d blocks: wall %.1f us; mean in-kernel %.0f ticks/wave (%.2f per instr); span %lld ticks; %zu distinct SIMDs, max %d waves on one; "
           "%d of %d same-SIMD pairs overlap >50%%\n", LDS_BYTES, blocks, ms * 1e3, mean, mean / (4096.0 * 8), tmax - tmin, per_simd.size(), maxw, overl, pairs);
}

int main()
{
    long long *d;
    hipMalloc(&d, 8 * 16 * 4096);
    run<1024>(256, d);
    run<1024>(512, d);
    run<1024>(1024, d);
    run<32768>(256, d);
    run<32768>(512, d);
    run<81920>(256, d);
    run<81920>(512, d);
    return 0;
}
