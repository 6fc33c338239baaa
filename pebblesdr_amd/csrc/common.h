// common.h -- shared definitions for libpebblegpu (gfx950 only; no CPU path, no other GPU back end).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace pg {

constexpr int kWave = 64;           // CDNA wavefront
constexpr int kBlock = 256;         // default workgroup: 4 waves, one per SIMD
constexpr int kMaxTaps = 80;        // >= 75 (CFir MAX_NUMCOEF) and >= 59 (largest halfband)
constexpr int kAmpTab = 512;        // oscillator amplitude transient table length (decays as 0.9^n)
constexpr int kMaxStages = 16;

// error plumbing: thread-local message, negative status codes (include/pebblegpu.h)
std::string &last_error();
int fail(int code, const char *fmt, ...);

#define PG_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return pg::fail(-3, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// kernel launch through a function pointer: the name may be a template-id with commas
template <class... KA, class... A>
inline void launch(void (*kernel)(KA...), dim3 grid, dim3 block, hipStream_t stream, A... args)
{
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, static_cast<KA>(args)...);
}

template <class... KA, class... A>
inline void launch_lds(void (*kernel)(KA...), dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream, A... args)
{
    hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, static_cast<KA>(args)...);
}

// ---- small complex helpers on float2 ----
// Written with float2's own vector operators (native <2 x float> underneath) so that a complex value stays one
// aligned register pair and the arithmetic becomes v_pk_add/mul/fma_f32 with op_sel/neg modifiers.  Component-wise
// scalar code makes the compiler scatter re/im over unrelated registers and gather them again with v_mov before
// every packed instruction (a third of the VALU stream in the FFT kernels).
__device__ __forceinline__ float2 cswap(float2 a) { return make_float2(a.y, a.x); }
// Complex product as two packed instructions and nothing else: on the native two-element vector type the half negation
// (-a.y, a.y) folds into the multiply's neg_lo modifier and the swap into its op_sel, also when `a` changes from use to use
// (through float2's operators the compiler built (-a.y, a.y) with a v_xor and a v_mov per product of a varying operand).
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    const v2f_t aa = {a.x, a.x}, na = {-a.y, a.y}, bb = {b.x, b.y}, bs = {b.y, b.x};
    const v2f_t d = __builtin_elementwise_fma(aa, bb, na * bs);
    return make_float2(d.x, d.y);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b)  // a * conj(b)
{
    const v2f_t bx = {b.x, b.x}, nb = {b.y, -b.y}, av = {a.x, a.y}, as = {a.y, a.x};
    const v2f_t d = __builtin_elementwise_fma(bx, av, nb * as);
    return make_float2(d.x, d.y);
}
// c * x and acc + c * x in two packed instructions with c as it lies in its register pair: the broadcasts and the half
// negation are operand modifiers.  (The compiler's own cmul materialises {c.x, c.x} and {-c.y, c.y} -- two more register pairs
// per loop-invariant constant, or two more instructions per product when c varies.)
__device__ __forceinline__ v2f_t cmul_pk(v2f_t c, v2f_t x)
{
    v2f_t r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(r) : "v"(c), "v"(x));  // (-c.y x.y, c.y x.x)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(r) : "v"(c), "v"(x));                       // + (c.x x.x, c.x x.y)
    return r;
}
// acc + c.x * v and acc + c.y * v (a real coefficient out of either half of a register pair, broadcast by the operand selects)
__device__ __forceinline__ v2f_t fma_lo_pk(v2f_t acc, v2f_t c, v2f_t v)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(c), "v"(v));
    return acc;
}
__device__ __forceinline__ v2f_t fma_hi_pk(v2f_t acc, v2f_t c, v2f_t v)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(c), "v"(v));
    return acc;
}
__device__ __forceinline__ float2 cmul_pk(float2 c, float2 x)
{
    const v2f_t r = cmul_pk(v2f_t{c.x, c.y}, v2f_t{x.x, x.y});
    return make_float2(r.x, r.y);
}
// the same with a wave-uniform c held in a scalar register pair (one scalar operand per instruction: the constant bus limit)
__device__ __forceinline__ float2 cmul_pk_s(float2 c, float2 x)
{
    const v2f_t cc = {c.x, c.y}, xx = {x.x, x.y};
    v2f_t r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(r) : "s"(cc), "v"(xx));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(r) : "s"(cc), "v"(xx));
    return make_float2(r.x, r.y);
}
__device__ __forceinline__ v2f_t cmac_pk(v2f_t acc, v2f_t c, v2f_t x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(c), "v"(x));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(c), "v"(x));
    return acc;
}
// Stores of results nothing on the device reads again soon (the display transforms' dB values: half a gigabyte per call): marked
// nontemporal they stream past L2 instead of evicting the input two kernels are reading (k_spectrum_t128: -4.7 % per call).
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(float4 *p, float4 v) { __builtin_nontemporal_store(v4f_t{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_t *>(p)); }
__device__ __forceinline__ void store_stream(float2 *p, float2 v) { __builtin_nontemporal_store(v2f_t{v.x, v.y}, reinterpret_cast<v2f_t *>(p)); }
__device__ __forceinline__ void store_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return a + b; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return a - b; }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return a * make_float2(s, s); }

// Wave-private LDS hand-off: lanes of one wave run in lockstep and its LDS operations retire in order, so
// this only has to stop the compiler from moving LDS accesses across the point (no s_barrier is emitted).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// compiler-only helpers: hide a value from loop-invariant code motion / pin the instruction schedule at a point
__device__ __forceinline__ void opaque(int &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// e^{j 2 pi c}, c in cycles (fp64), reduced to [-0.5, 0.5) before the fp32 sincos
__device__ __forceinline__ float2 cis_cycles(double c)
{
    c -= rint(c);
    float s, co;
    sincospif(2.0f * (float)c, &s, &co);
    return make_float2(co, s);
}

}  // namespace pg
