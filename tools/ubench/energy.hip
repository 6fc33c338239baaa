// DEVELOPER-ONLY: socket power under sustained streams of one instruction class -- what a wave64 instruction costs in energy on
// this part, and how far the shader clock is pulled down while it runs.  Each class runs ~3 s (launches of ~20 ms back to back);
// power from rocm-smi (sampled three times in the last second), instruction rate from the launch times.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/energy.hip -o tools/ubench/energy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                                                         \
    __global__ __launch_bounds__(256) void name(long long *out, float seed, int iters)                           \
    {                                                                                                             \
        typedef float v2 __attribute__((ext_vector_type(2)));                                                     \
        v2 a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        v2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};                                                             \
        float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3, f4 = seed + 4, f5 = seed + 5, f6 = seed + 6, f7 = seed + 7, fb = 1.0001f, fc = 0.5f; \
        __shared__ float lds[4096];                                                                               \
        int addr = threadIdx.x * 8;                                                                               \
        lds[threadIdx.x] = seed;                                                                                  \
        __syncthreads();                                                                                          \
        const long long t0 = clock64();                                                                           \
        for (int it = 0; it < iters; it++) {                                                                      \
            asm volatile(REP8(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(c), "v"(addr), "v"(fb), "v"(fc)); \
        }                                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                                    \
        const long long t1 = clock64();                                                                           \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                \
        if (a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 12345.f) out[1000] = 1; \
    }

// eight instructions per body line, eight lines per iteration = 64 instructions per iteration
KERNEL(k_pk_fma, "v_pk_fma_f32 %0, %0, %16, %17\nv_pk_fma_f32 %1, %1, %16, %17\nv_pk_fma_f32 %2, %2, %16, %17\nv_pk_fma_f32 %3, %3, %16, %17\nv_pk_fma_f32 %4, %4, %16, %17\nv_pk_fma_f32 %5, %5, %16, %17\nv_pk_fma_f32 %6, %6, %16, %17\nv_pk_fma_f32 %7, %7, %16, %17\n")
KERNEL(k_pk_add, "v_pk_add_f32 %0, %0, %16\nv_pk_add_f32 %1, %1, %16\nv_pk_add_f32 %2, %2, %16\nv_pk_add_f32 %3, %3, %16\nv_pk_add_f32 %4, %4, %16\nv_pk_add_f32 %5, %5, %16\nv_pk_add_f32 %6, %6, %16\nv_pk_add_f32 %7, %7, %16\n")
KERNEL(k_fma, "v_fma_f32 %8, %8, %19, %20\nv_fma_f32 %9, %9, %19, %20\nv_fma_f32 %10, %10, %19, %20\nv_fma_f32 %11, %11, %19, %20\nv_fma_f32 %12, %12, %19, %20\nv_fma_f32 %13, %13, %19, %20\nv_fma_f32 %14, %14, %19, %20\nv_fma_f32 %15, %15, %19, %20\n")
KERNEL(k_add, "v_add_f32 %8, %8, %19\nv_add_f32 %9, %9, %19\nv_add_f32 %10, %10, %19\nv_add_f32 %11, %11, %19\nv_add_f32 %12, %12, %19\nv_add_f32 %13, %13, %19\nv_add_f32 %14, %14, %19\nv_add_f32 %15, %15, %19\n")
KERNEL(k_sqrt, "v_sqrt_f32 %8, %8\nv_sqrt_f32 %9, %9\nv_sqrt_f32 %10, %10\nv_sqrt_f32 %11, %11\nv_sqrt_f32 %12, %12\nv_sqrt_f32 %13, %13\nv_sqrt_f32 %14, %14\nv_sqrt_f32 %15, %15\n")
KERNEL(k_ds_read, "ds_read_b64 %0, %18\nds_read_b64 %1, %18 offset:512\nds_read_b64 %2, %18 offset:1024\nds_read_b64 %3, %18 offset:1536\nds_read_b64 %4, %18 offset:2048\nds_read_b64 %5, %18 offset:2560\nds_read_b64 %6, %18 offset:3072\nds_read_b64 %7, %18 offset:3584\n")
KERNEL(k_ds_write, "ds_write_b64 %18, %0\nds_write_b64 %18, %1 offset:512\nds_write_b64 %18, %2 offset:1024\nds_write_b64 %18, %3 offset:1536\nds_write_b64 %18, %4 offset:2048\nds_write_b64 %18, %5 offset:2560\nds_write_b64 %18, %6 offset:3072\nds_write_b64 %18, %7 offset:3584\n")
KERNEL(k_swap, "v_permlane32_swap_b32 %8, %9\ns_nop 1\nv_permlane32_swap_b32 %10, %11\ns_nop 1\nv_permlane32_swap_b32 %12, %13\ns_nop 1\nv_permlane32_swap_b32 %14, %15\ns_nop 1\nv_permlane16_swap_b32 %8, %9\ns_nop 1\nv_permlane16_swap_b32 %10, %11\ns_nop 1\nv_permlane16_swap_b32 %12, %13\ns_nop 1\nv_permlane16_swap_b32 %14, %15\ns_nop 1\n")
KERNEL(k_nop, "s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\n")

static double read_power()
{
    FILE *p = popen("rocm-smi --showpower 2>/dev/null", "r");
    if (!p) return -1;
    char line[512];
    double w = -1;
    while (fgets(line, sizeof line, p)) {
        const char *s = strstr(line, "(W):");
        if (s) w = atof(s + 4);
    }
    pclose(p);
    return w;
}

template <class K> static void run(const char *name, K kern, long long *d, int cus, int iters)
{
    using clk = std::chrono::steady_clock;
    const int blocks = cus * 4;  // four waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0f, iters);
    hipDeviceSynchronize();
    const auto t0 = clk::now();
    long launches = 0;
    double pw[3] = {0, 0, 0};
    int np = 0;
    // keep eight launches queued; sample the power three times after two seconds
    while (true) {
        for (int i = 0; i < 8; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0f, iters);
        launches += 8;
        const double el = std::chrono::duration<double>(clk::now() - t0).count();
        if (el > 2.0 && np < 3) pw[np++] = read_power();
        hipDeviceSynchronize();
        if (np == 3) break;
    }
    const double el = std::chrono::duration<double>(clk::now() - t0).count();
    long long ticks = 0;
    hipMemcpy(&ticks, d, 8, hipMemcpyDeviceToHost);
    const double instr = (double)launches * blocks * 4 /*waves*/ * (double)iters * 64.0;  // wave64 instructions
    const double p = (pw[0] + pw[1] + pw[2]) / 3;
    printf("%-22s %7.0f W   %8.2f G wave-instr/s   %6.2f shader-clock ticks per instruction and wave (4 waves/SIMD)\n", name, p, instr / el * 1e-9, (double)ticks / ((double)iters * 64.0));
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    long long *d;
    hipMalloc(&d, 8 * 4096);
    printf("%s, %d CUs; idle %.0f W\n", pr.name, pr.multiProcessorCount, read_power());
    const int cus = pr.multiProcessorCount;
    run("s_nop 7 (idle waves)", k_nop, d, cus, 40000);
    run("v_pk_fma_f32", k_pk_fma, d, cus, 40000);
    run("v_pk_add_f32", k_pk_add, d, cus, 40000);
    run("v_fma_f32", k_fma, d, cus, 80000);
    run("v_add_f32", k_add, d, cus, 80000);
    run("v_sqrt_f32", k_sqrt, d, cus, 30000);
    run("v_permlane{32,16}_swap", k_swap, d, cus, 40000);
    run("ds_read_b64", k_ds_read, d, cus, 30000);
    run("ds_write_b64", k_ds_write, d, cus, 10000);
    return 0;
}
