// DEVELOPER-ONLY micro-benchmark: issue cost (cycles per wave64 instruction) of the VALU/LDS instructions the FFT
// kernels lean on, on whatever gfx9 device runs it.  One wave per SIMD unless WAVES > 1.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/isa_rates.hip -o gpurun_out/isa_rates && ./gpurun_out/isa_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(name, asm_line)                                                                              \
    __global__ void name(long long *out, float seed)                                                     \
    {                                                                                                     \
        typedef float v2 __attribute__((ext_vector_type(2)));                                             \
        v2 a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        v2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};                                                      \
        float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3, f4 = seed + 4, f5 = seed + 5, f6 = seed + 6, f7 = seed + 7, fb = 1.0001f, fc = 0.5f; \
        __shared__ float lds[8192];                                                                       \
        int addr = threadIdx.x * 8;                                                                       \
        lds[threadIdx.x] = seed;                                                                          \
        __syncthreads();                                                                                  \
        long long t0 = clock64();                                                                         \
        for (int it = 0; it < 1024; it++) {                                                                 \
            asm volatile(REP16(asm_line) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(c), "v"(addr), "v"(fb), "v"(fc)); \
        }                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                            \
        long long t1 = clock64();                                                                         \
        if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;                     \
        if (a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 12345.f) out[1000] = 1;              \
    }

// 8 independent chains per asm line group: each line below is 8 instructions
#define L8(op, ...) op " %0, " __VA_ARGS__ "\n" op " %1, " __VA_ARGS__ "\n" op " %2, " __VA_ARGS__ "\n" op " %3, " __VA_ARGS__ "\n" \
                    op " %4, " __VA_ARGS__ "\n" op " %5, " __VA_ARGS__ "\n" op " %6, " __VA_ARGS__ "\n" op " %7, " __VA_ARGS__ "\n"

BODY(k_pk_fma, "v_pk_fma_f32 %0, %0, %16, %17\nv_pk_fma_f32 %1, %1, %16, %17\nv_pk_fma_f32 %2, %2, %16, %17\nv_pk_fma_f32 %3, %3, %16, %17\nv_pk_fma_f32 %4, %4, %16, %17\nv_pk_fma_f32 %5, %5, %16, %17\nv_pk_fma_f32 %6, %6, %16, %17\nv_pk_fma_f32 %7, %7, %16, %17\n")
BODY(k_pk_add, "v_pk_add_f32 %0, %0, %16\nv_pk_add_f32 %1, %1, %16\nv_pk_add_f32 %2, %2, %16\nv_pk_add_f32 %3, %3, %16\nv_pk_add_f32 %4, %4, %16\nv_pk_add_f32 %5, %5, %16\nv_pk_add_f32 %6, %6, %16\nv_pk_add_f32 %7, %7, %16\n")
BODY(k_pk_add_dep, "v_pk_add_f32 %0, %0, %16\nv_pk_add_f32 %0, %0, %16\nv_pk_add_f32 %0, %0, %16\nv_pk_add_f32 %0, %0, %16\nv_pk_add_f32 %0, %0, %16\nv_pk_add_f32 %0, %0, %16\nv_pk_add_f32 %0, %0, %16\nv_pk_add_f32 %0, %0, %16\n")
BODY(k_fma, "v_fma_f32 %8, %8, %19, %20\nv_fma_f32 %9, %9, %19, %20\nv_fma_f32 %10, %10, %19, %20\nv_fma_f32 %11, %11, %19, %20\nv_fma_f32 %12, %12, %19, %20\nv_fma_f32 %13, %13, %19, %20\nv_fma_f32 %14, %14, %19, %20\nv_fma_f32 %15, %15, %19, %20\n")
BODY(k_fma_dep, "v_fma_f32 %8, %8, %19, %20\nv_fma_f32 %8, %8, %19, %20\nv_fma_f32 %8, %8, %19, %20\nv_fma_f32 %8, %8, %19, %20\nv_fma_f32 %8, %8, %19, %20\nv_fma_f32 %8, %8, %19, %20\nv_fma_f32 %8, %8, %19, %20\nv_fma_f32 %8, %8, %19, %20\n")
BODY(k_sqrt, "v_sqrt_f32 %8, %8\nv_sqrt_f32 %9, %9\nv_sqrt_f32 %10, %10\nv_sqrt_f32 %11, %11\nv_sqrt_f32 %12, %12\nv_sqrt_f32 %13, %13\nv_sqrt_f32 %14, %14\nv_sqrt_f32 %15, %15\n")
BODY(k_log, "v_log_f32 %8, %8\nv_log_f32 %9, %9\nv_log_f32 %10, %10\nv_log_f32 %11, %11\nv_log_f32 %12, %12\nv_log_f32 %13, %13\nv_log_f32 %14, %14\nv_log_f32 %15, %15\n")
BODY(k_sqrt_mix, "v_sqrt_f32 %8, %8\nv_fma_f32 %9, %9, %19, %20\nv_fma_f32 %10, %10, %19, %20\nv_fma_f32 %11, %11, %19, %20\nv_sqrt_f32 %12, %12\nv_fma_f32 %13, %13, %19, %20\nv_fma_f32 %14, %14, %19, %20\nv_fma_f32 %15, %15, %19, %20\n")
BODY(k_mov, "v_mov_b32 %8, %19\nv_mov_b32 %9, %19\nv_mov_b32 %10, %19\nv_mov_b32 %11, %19\nv_mov_b32 %12, %19\nv_mov_b32 %13, %19\nv_mov_b32 %14, %19\nv_mov_b32 %15, %19\n")
BODY(k_ds_read_b64, "ds_read_b64 %0, %18\nds_read_b64 %1, %18 offset:512\nds_read_b64 %2, %18 offset:1024\nds_read_b64 %3, %18 offset:1536\nds_read_b64 %4, %18 offset:2048\nds_read_b64 %5, %18 offset:2560\nds_read_b64 %6, %18 offset:3072\nds_read_b64 %7, %18 offset:3584\n")
BODY(k_ds_write_b64, "ds_write_b64 %18, %0\nds_write_b64 %18, %1 offset:512\nds_write_b64 %18, %2 offset:1024\nds_write_b64 %18, %3 offset:1536\nds_write_b64 %18, %4 offset:2048\nds_write_b64 %18, %5 offset:2560\nds_write_b64 %18, %6 offset:3072\nds_write_b64 %18, %7 offset:3584\n")


// ---- LDS variants (bytes per lane: b32 4, b64 8, b128 16, *2 forms twice that) ----
#define LDSK(name, line) BODY(name, line)
typedef float v4f __attribute__((ext_vector_type(4)));
#define BODY4(name, asm_line)                                                                             \
    __global__ void name(long long *out, float seed)                                                     \
    {                                                                                                     \
        v4f a0 = {seed, seed + 1, seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;               \
        __shared__ float lds[8192];                                                                       \
        int addr = threadIdx.x * 16;                                                                      \
        lds[threadIdx.x] = seed;                                                                          \
        __syncthreads();                                                                                  \
        for (int it = 0; it < 1024; it++) {                                                               \
            asm volatile(REP16(asm_line) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(addr));           \
        }                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                            \
        if (a0.x + a1.x + a2.x + a3.x == 12345.f) out[1000] = 1;                                          \
    }
// 4 instructions per line (so n = 1024*16*4)
BODY4(k_ds_write_b128, "ds_write_b128 %4, %0\nds_write_b128 %4, %1 offset:4096\nds_write_b128 %4, %2 offset:8192\nds_write_b128 %4, %3 offset:12288\n")
BODY4(k_ds_read_b128, "ds_read_b128 %0, %4\nds_read_b128 %1, %4 offset:4096\nds_read_b128 %2, %4 offset:8192\nds_read_b128 %3, %4 offset:12288\n")
BODY(k_ds_write2_b64, "ds_write2_b64 %18, %0, %1 offset1:64\nds_write2_b64 %18, %2, %3 offset0:128 offset1:192\nds_write2_b64 %18, %4, %5 offset1:64\nds_write2_b64 %18, %6, %7 offset0:128 offset1:192\nds_write2_b64 %18, %0, %1 offset1:64\nds_write2_b64 %18, %2, %3 offset0:128 offset1:192\nds_write2_b64 %18, %4, %5 offset1:64\nds_write2_b64 %18, %6, %7 offset0:128 offset1:192\n")
BODY(k_ds_write_b32, "ds_write_b32 %18, %8\nds_write_b32 %18, %9 offset:512\nds_write_b32 %18, %10 offset:1024\nds_write_b32 %18, %11 offset:1536\nds_write_b32 %18, %12 offset:2048\nds_write_b32 %18, %13 offset:2560\nds_write_b32 %18, %14 offset:3072\nds_write_b32 %18, %15 offset:3584\n")
BODY(k_ds_read_b32, "ds_read_b32 %8, %18\nds_read_b32 %9, %18 offset:512\nds_read_b32 %10, %18 offset:1024\nds_read_b32 %11, %18 offset:1536\nds_read_b32 %12, %18 offset:2048\nds_read_b32 %13, %18 offset:2560\nds_read_b32 %14, %18 offset:3072\nds_read_b32 %15, %18 offset:3584\n")


// overlap tests: 8 instructions per line again (2 LDS + 6 VALU, or 4 + 4)
BODY(k_mix_w_fma, "ds_write_b64 %18, %0\nv_pk_fma_f32 %1, %1, %16, %17\nv_pk_fma_f32 %2, %2, %16, %17\nv_pk_fma_f32 %3, %3, %16, %17\nds_write_b64 %18, %4 offset:512\nv_pk_fma_f32 %5, %5, %16, %17\nv_pk_fma_f32 %6, %6, %16, %17\nv_pk_fma_f32 %7, %7, %16, %17\n")
BODY(k_mix_r_fma, "ds_read_b64 %0, %18\nv_pk_fma_f32 %1, %1, %16, %17\nv_pk_fma_f32 %2, %2, %16, %17\nv_pk_fma_f32 %3, %3, %16, %17\nds_read_b64 %4, %18 offset:512\nv_pk_fma_f32 %5, %5, %16, %17\nv_pk_fma_f32 %6, %6, %16, %17\nv_pk_fma_f32 %7, %7, %16, %17\n")
BODY(k_mix_w_sfma, "ds_write_b64 %18, %0\nv_fma_f32 %9, %9, %19, %20\nv_fma_f32 %10, %10, %19, %20\nv_fma_f32 %11, %11, %19, %20\nds_write_b64 %18, %4 offset:512\nv_fma_f32 %13, %13, %19, %20\nv_fma_f32 %14, %14, %19, %20\nv_fma_f32 %15, %15, %19, %20\n")

static int g_cus = 256;
static double g_ghz = 2.4;

template <class K> static void run(const char *name, K kernel, int waves_per_simd, long long *d, double n = 1024.0 * 16 * 8)
{
    // 256-thread workgroups (one wave per SIMD of a CU each); the 32 KiB of LDS per workgroup caps residency at 5/CU.
    // grid = CUs * waves_per_simd workgroups -> that many waves per SIMD when the dispatcher spreads them evenly.
    const int blocks = g_cus * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(16);
    hipMemcpy(h.data(), d, 8 * 16, hipMemcpyDeviceToHost);
    printf("%-14s %d wave/SIMD: in-kernel %.2f ticks/instr/wave; wall %.1f us -> %.2f SIMD-clk per instr at %.2f GHz (incl. launch)\n", name,
           waves_per_simd, h[0] / n, ms * 1e3, ms * 1e-3 * g_ghz * 1e9 / (n * waves_per_simd), g_ghz);
}

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    g_cus = pr.multiProcessorCount;
    g_ghz = pr.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", pr.name, g_cus, g_ghz);
    long long *d;
    hipMalloc(&d, 8 * 16 * 4096);
    for (int w = 1; w <= 4; w *= 2) {
        run("v_pk_fma_f32", k_pk_fma, w, d);
        run("v_pk_add_f32", k_pk_add, w, d);
        run("v_pk_add dep", k_pk_add_dep, w, d);
        run("v_fma_f32", k_fma, w, d);
        run("v_fma dep", k_fma_dep, w, d);
        run("v_mov_b32", k_mov, w, d);
        run("v_sqrt_f32", k_sqrt, w, d);
        run("v_log_f32", k_log, w, d);
        run("sqrt+3fma mix", k_sqrt_mix, w, d);
        run("ds_read_b64", k_ds_read_b64, w, d);
        run("ds_write_b64", k_ds_write_b64, w, d);
        run("2w_b64+6pkfma", k_mix_w_fma, w, d);
        run("2r_b64+6pkfma", k_mix_r_fma, w, d);
        run("2w_b64+6fma", k_mix_w_sfma, w, d);
        run("ds_write_b32", k_ds_write_b32, w, d);
        run("ds_read_b32", k_ds_read_b32, w, d);
        run("ds_write2_b64", k_ds_write2_b64, w, d);
        run("ds_write_b128", k_ds_write_b128, w, d, 1024.0 * 16 * 4);
        run("ds_read_b128", k_ds_read_b128, w, d, 1024.0 * 16 * 4);
    }
    return 0;
}
