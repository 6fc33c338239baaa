// hip_emu.h -- DEVELOPER-ONLY single-threaded emulator of the small HIP subset libpebblegpu uses.
//
// Purpose: debug kernel indexing (FFT passes, halos, scans) in the GPU-less build container before
// spending GPU-box minutes.  It is NOT a CPU fallback: the package (pebblesdr_amd/) never builds, loads
// or knows about it; only tools/hipemu/check.py does, by explicit path.  Work-items of one workgroup run as
// ucontext fibers on one OS thread; __syncthreads and the wave shuffles are cooperative yield points.
// Timing, occupancy, bank conflicts and memory coalescing are not modelled.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define PEBBLE_HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
static inline float2 operator+(float2 a, float2 b) { return float2{a.x + b.x, a.y + b.y}; }
static inline float2 operator-(float2 a, float2 b) { return float2{a.x - b.x, a.y - b.y}; }
static inline float2 operator*(float2 a, float2 b) { return float2{a.x * b.x, a.y * b.y}; }
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct uint2 { unsigned x, y; };
struct char2 { signed char x, y; };
struct uchar2 { unsigned char x, y; };
struct short2 { short x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

namespace hipemu {
struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    int state = 0;  // 0 runnable, 1 waiting block barrier, 2 waiting wave sync, 3 done
};
struct State {
    dim3 tIdx, bIdx, bDim, gDim;
    std::vector<Fiber> fibers;
    ucontext_t sched;
    int cur = -1;
    std::function<void()> body;
    uint64_t slot[1024];
};
inline State &S() { static State s; return s; }
inline void yield_as(int st)
{
    State &s = S();
    s.fibers[s.cur].state = st;
    swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}
inline void trampoline()
{
    State &s = S();
    s.body();
    s.fibers[s.cur].state = 3;
    swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}
inline void run_block(unsigned nthreads)
{
    State &s = S();
    const size_t STK = 256 * 1024;
    if (s.fibers.size() < nthreads) {
        size_t old = s.fibers.size();
        s.fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; i++) s.fibers[i].stack = (char *)malloc(STK);
    }
    for (unsigned i = 0; i < nthreads; i++) {
        Fiber &f = s.fibers[i];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STK;
        f.ctx.uc_link = &s.sched;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
        f.state = 0;
    }
    for (;;) {
        for (unsigned i = 0; i < nthreads; i++) {
            if (s.fibers[i].state != 0) continue;
            s.cur = (int)i;
            s.tIdx = dim3(i % s.bDim.x, (i / s.bDim.x) % s.bDim.y, i / (s.bDim.x * s.bDim.y));
            swapcontext(&s.sched, &s.fibers[i].ctx);
        }
        unsigned live = 0, at_block = 0;
        for (unsigned i = 0; i < nthreads; i++) {
            if (s.fibers[i].state != 3) live++;
            if (s.fibers[i].state == 1) at_block++;
        }
        if (!live) break;
        // a wave whose live lanes all wait on a wave-level sync proceeds
        bool released = false;
        for (unsigned w = 0; w * 64 < nthreads; w++) {
            unsigned lo = w * 64, hi = lo + 64 < nthreads ? lo + 64 : nthreads;
            unsigned wl = 0, ww = 0;
            for (unsigned i = lo; i < hi; i++) {
                if (s.fibers[i].state != 3) wl++;
                if (s.fibers[i].state == 2) ww++;
            }
            if (ww && ww == wl) {
                for (unsigned i = lo; i < hi; i++) if (s.fibers[i].state == 2) s.fibers[i].state = 0;
                released = true;
            }
        }
        if (released) continue;
        if (at_block == live) {
            for (unsigned i = 0; i < nthreads; i++) if (s.fibers[i].state == 1) s.fibers[i].state = 0;
            continue;
        }
        fprintf(stderr, "hipemu: deadlock (divergent barrier/shuffle) in block (%u,%u)\n", s.bIdx.x, s.bIdx.y);
        abort();
    }
}
template <class F>
inline void launch(dim3 grid, dim3 block, F f)
{
    State &s = S();
    s.gDim = grid;
    s.bDim = block;
    s.body = f;
    unsigned nt = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                s.bIdx = dim3(bx, by, bz);
                run_block(nt);
            }
}
inline unsigned flat_tid() { State &s = S(); return s.tIdx.x + s.bDim.x * (s.tIdx.y + s.bDim.y * s.tIdx.z); }
template <class T>
inline T shfl_idx(T v, int src_lane)
{
    static_assert(sizeof(T) <= 8, "shfl width");
    State &s = S();
    unsigned t = flat_tid(), base = t & ~63u;
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    s.slot[t] = raw;
    yield_as(2);
    unsigned nthreads = s.bDim.x * s.bDim.y * s.bDim.z;
    unsigned src = base + ((unsigned)src_lane & 63u);
    T r = v;
    if (src < nthreads) memcpy(&r, &s.slot[src], sizeof(T));
    yield_as(2);
    return r;
}
}  // namespace hipemu

#define threadIdx (hipemu::S().tIdx)
#define blockIdx (hipemu::S().bIdx)
#define blockDim (hipemu::S().bDim)
#define gridDim (hipemu::S().gDim)

static inline void __syncthreads() { hipemu::yield_as(1); }
template <class T> static inline T __shfl(T v, int src, int width = 64)
{
    int lane = (int)(hipemu::flat_tid() & 63u);
    int s = (lane & ~(width - 1)) + (src & (width - 1));
    return hipemu::shfl_idx(v, s);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64)
{
    int lane = (int)(hipemu::flat_tid() & 63u);
    int s = lane - (int)d;
    if (s < (lane & ~(width - 1))) s = lane;
    return hipemu::shfl_idx(v, s);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64)
{
    int lane = (int)(hipemu::flat_tid() & 63u);
    int s = lane + (int)d;
    if (s > (lane | (width - 1))) s = lane;
    return hipemu::shfl_idx(v, s);
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64)
{
    (void)width;
    int lane = (int)(hipemu::flat_tid() & 63u);
    return hipemu::shfl_idx(v, lane ^ m);
}

// wave-level builtins used by common.h's wave_sync()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { hipemu::yield_as(2); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }

// device math the kernels use
static inline void sincospif(float x, float *s, float *c)
{
    double a = 3.14159265358979323846 * (double)x;
    *s = (float)sin(a);
    *c = (float)cos(a);
}
static inline void sincospi(double x, double *s, double *c)
{
    double a = 3.14159265358979323846 * x;
    *s = sin(a);
    *c = cos(a);
}
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

// ---- host runtime subset ----
typedef int hipError_t;
typedef void *hipStream_t;
struct hipEmuEvent { double t; };
typedef hipEmuEvent *hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
#define hipStreamNonBlocking 1
static inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipFree(void *p) { free(p); return 0; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void *p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, int, hipStream_t)
{
    for (size_t r = 0; r < h; r++) memmove((char *)d + r * dp, (const char *)s + r * sp, w);
    return 0;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return 0; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipEmuEvent{0}; return 0; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new hipEmuEvent{0}; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = 0; return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }

// kernels are launched through pg_launch() in csrc/common.h, which maps onto hipemu::launch here.
