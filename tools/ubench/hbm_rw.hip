// DEVELOPER-ONLY: what the HBM path of this part sustains for a read-only, a write-only and a read+write stream, and for
// the read:write mix of the 8192-bin spectrum kernel (1 byte read per 2 written).  1 GiB per direction, 16-byte lanes,
// grid-stride over 256 x 8 workgroups.  Prints GB/s per pattern (best of 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ a, float4 *__restrict__ sink, long long n)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float4 v = a[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(float4 *__restrict__ b, long long n, float s)
{
    const float4 v = make_float4(s, s + 1, s + 2, s + 3);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = v;
}
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = a[i];
}
// one read per two writes (8 B in, 16 B out per unit -- the spectrum kernel's 8*N + 4*bins with bins = 4 N)
__global__ __launch_bounds__(256) void k_r1w2(const float4 *__restrict__ a, float4 *__restrict__ b, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float4 v = a[i];
        b[2 * i] = v;
        b[2 * i + 1] = make_float4(v.w, v.z, v.y, v.x);
    }
}

template <class F> static float best_ms(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        hipEventRecord(e0, 0);
        f();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it) best = std::min(best, ms);
    }
    return best;
}

int main()
{
    const long long bytes = 1LL << 30, n = bytes / 16;
    float4 *a, *b, *sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, 2 * bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(a, 1, bytes); hipMemset(b, 0, 2 * bytes);
    const dim3 grid(256 * 8), block(256);
    const float r = best_ms([&] { hipLaunchKernelGGL(k_read, grid, block, 0, 0, a, sink, n); });
    const float w = best_ms([&] { hipLaunchKernelGGL(k_write, grid, block, 0, 0, b, n, 1.f); });
    const float c = best_ms([&] { hipLaunchKernelGGL(k_copy, grid, block, 0, 0, a, b, n); });
    const float m = best_ms([&] { hipLaunchKernelGGL(k_r1w2, grid, block, 0, 0, a, b, n / 2); });
    printf("read-only   1 GiB: %.3f ms  %.0f GB/s\n", r, bytes / (r * 1e-3) / 1e9);
    printf("write-only  1 GiB: %.3f ms  %.0f GB/s\n", w, bytes / (w * 1e-3) / 1e9);
    printf("copy    1+1 GiB: %.3f ms  %.0f GB/s total\n", c, 2.0 * bytes / (c * 1e-3) / 1e9);
    printf("read 0.5 GiB + write 1 GiB (1:2): %.3f ms  %.0f GB/s total, %.0f GB/s written\n", m, 1.5 * bytes / (m * 1e-3) / 1e9, bytes / (m * 1e-3) / 1e9);
    return 0;
}
