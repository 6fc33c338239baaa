// fft_lds.h -- cooperative FFT for N in {2048, 4096, 8192}: T work-items (a 256-item workgroup, or ONE
// 64-lane wave for N = 2048), N/T points per work-item held in registers, Stockham autosort passes
// (radix 8, with one leading radix-4 or radix-2 pass) exchanged through LDS.  fp32, twiddles from an
// fp64-rounded per-pass table (fft_tw_off below): for the pass whose earlier radices multiply to P, three runs of P
// entries W^{i}, W^{2i}, W^{4i} with i = k*N/(P*R), indexed by k -- consecutive work-items read consecutive entries
// (no LDS bank conflicts when the table is staged in LDS, coalesced when it is read from L1/L2).
//
// With T = 64 the exchange is wave-private: a wave's LDS operations retire in issue order, so the passes
// need no s_barrier at all -- four waves of a workgroup run four independent transforms.
//
// Register layout ("strided"): x[m] holds element  tid + T*m.  The first pass reads its butterfly
// inputs from exactly those slots and the last pass leaves its outputs in exactly those slots, so a
// forward transform, a pointwise product and an inverse transform chain with no extra exchange
// (the overlap-save band-pass does that), and global loads/stores of x[] are coalesced.
//
// LDS image: element i lives at lpad(i) = i + 4*(i>>5) (float2 units).  The 4-slot pad per 32
// elements makes the stride-4 / stride-32 scatter of the early passes land on distinct banks
// for ds_write_b64 (16-lane groups, 32 dword banks) while unit-stride gathers stay conflict-free.
#pragma once
#include "common.h"

namespace pg {

__device__ __forceinline__ int lpad(int i) { return i + ((i >> 5) << 2); }
// The exchange after a leading radix-4 pass uses its own padding, one slot per 16: a work-item scatters 4 adjacent
// elements there (lane stride 32 B), which the 4-per-32 pad leaves on half of the banks.
__device__ __forceinline__ int lpad4(int i) { return i + (i >> 4); }
// ... and after the wave transform's leading radix-32 pass (32 adjacent elements per lane, lane stride 256 B) one slot
// per 32: lane stride 33 slots = 66 dwords, 16 lanes cover the 32 banks.
__device__ __forceinline__ int lpad1(int i) { return i + (i >> 5); }
enum { kPad4per32 = 0, kPad1per16 = 1, kPad1per32 = 2 };
template <int A> __device__ __forceinline__ int lpad_sel(int i) { return A == kPad1per16 ? lpad4(i) : A == kPad1per32 ? lpad1(i) : lpad(i); }
template <int N> struct FftLds { static constexpr int kSlots = N + (N >> 5) * 4; };

// the pass plan (shared by the kernels and the host code that fills the twiddle table)
__host__ __device__ constexpr int fft_passes(int n) { return n == 8192 ? 5 : 4; }
__host__ __device__ constexpr int fft_radix(int n, int pass) { return n == 2048 ? (pass == 0 ? 4 : 8) : n == 4096 ? 8 : (pass == 0 ? 2 : 8); }
// offset (float2 entries) of the table of the pass with prefix product P; P == n gives the table length
__host__ __device__ constexpr int fft_tw_off(int n, int P)
{
    int off = 0, p = 1;
    for (int i = 0; i < fft_passes(n); i++) {
        if (p == P) return off;
        if (p > 1) off += 3 * p;
        p *= fft_radix(n, i);
    }
    return off;
}

// DIR = +1: forward (e^{-j}), DIR = -1: inverse (e^{+j}); unscaled both ways.
template <int DIR> __device__ __forceinline__ float2 mul_mj(float2 a)  // a * (-j*DIR)
{
    return DIR > 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}
template <int DIR> __device__ __forceinline__ float2 twid(float2 w)  // table holds forward twiddles
{
    return DIR > 0 ? w : make_float2(w.x, -w.y);
}

template <int DIR> __device__ __forceinline__ void bfly2(float2 *u)
{
    float2 a = u[0], b = u[1];
    u[0] = cadd(a, b);
    u[1] = csub(a, b);
}
// a + (-j*DIR) b and a - (-j*DIR) b as one packed FMA each: (-j*DIR) b = swap(b) * (DIR, -DIR).  (A swap folds into
// the instruction's op_sel; a half negation does not and would cost two extra moves per use.)
template <int DIR> __device__ __forceinline__ float2 add_mj(float2 a, float2 b)
{
    return cswap(b) * make_float2((float)DIR, (float)-DIR) + a;
}
template <int DIR> __device__ __forceinline__ float2 sub_mj(float2 a, float2 b)
{
    return cswap(b) * make_float2((float)-DIR, (float)DIR) + a;
}
template <int DIR> __device__ __forceinline__ void bfly4(float2 *u)
{
    float2 t0 = cadd(u[0], u[2]), t1 = csub(u[0], u[2]);
    float2 t2 = cadd(u[1], u[3]), d = csub(u[1], u[3]);
    u[0] = cadd(t0, t2);
    u[2] = csub(t0, t2);
    u[1] = add_mj<DIR>(t1, d);
    u[3] = sub_mj<DIR>(t1, d);
}
template <int DIR> __device__ __forceinline__ void bfly8(float2 *u)
{
    float2 e[4] = {u[0], u[2], u[4], u[6]};
    float2 o[4] = {u[1], u[3], u[5], u[7]};
    bfly4<DIR>(e);
    bfly4<DIR>(o);
    const float c = 0.70710678118654752440f;
    // W8^1 = c(1 - j*DIR), W8^2 = -j*DIR, W8^3 = c(-1 - j*DIR):  W8^1 o = c (o + mj o),  W8^3 o = -c (o - mj o)
    const float2 o1 = cscale(add_mj<DIR>(o[1], o[1]), c);
    const float2 o3 = cscale(sub_mj<DIR>(o[3], o[3]), c);
    u[0] = cadd(e[0], o[0]); u[4] = csub(e[0], o[0]);
    u[1] = cadd(e[1], o1);   u[5] = csub(e[1], o1);
    u[2] = add_mj<DIR>(e[2], o[2]); u[6] = sub_mj<DIR>(e[2], o[2]);
    u[3] = csub(e[3], o3);   u[7] = cadd(e[3], o3);
}
// 32-point forward DFT in registers as 8 x 4 (n = 4*na + nb): DFT8 over na, twiddle W32^{nb*ka}, DFT4 over nb.
// in: u[n]; out: X[ka + 8*kb] is left in u[4*ka + kb], i.e. X[k] = u[perm32(k)].
__host__ __device__ constexpr int perm32(int k) { return 4 * (k & 7) + (k >> 3); }
__device__ __forceinline__ void dft32(float2 *u)
{
    // W32^m = exp(-2 pi i m / 32), m = nb*ka <= 21
    constexpr float kC[22] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                              0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.f, -0.19509032201612826785f,
                              -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                              -0.92387953251128675613f, -0.98078528040323044913f, -1.f, -0.98078528040323044913f, -0.92387953251128675613f,
                              -0.83146961230254523708f, -0.70710678118654752440f, -0.55557023301960222474f};
    constexpr float kS[22] = {0.f, -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f,
                              -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f, -1.f, -0.98078528040323044913f,
                              -0.92387953251128675613f, -0.83146961230254523708f, -0.70710678118654752440f, -0.55557023301960222474f,
                              -0.38268343236508977173f, -0.19509032201612826785f, 0.f, 0.19509032201612826785f, 0.38268343236508977173f,
                              0.55557023301960222474f, 0.70710678118654752440f, 0.83146961230254523708f};
#pragma unroll
    for (int nb = 0; nb < 4; nb++) {
        float2 t[8];
#pragma unroll
        for (int na = 0; na < 8; na++) t[na] = u[nb + 4 * na];
        bfly8<+1>(t);
#pragma unroll
        for (int ka = 0; ka < 8; ka++) {
            const int m = nb * ka;
            u[4 * ka + nb] = m == 0 ? t[ka] : m == 8 ? mul_mj<+1>(t[ka]) : cmul(make_float2(kC[m], kS[m]), t[ka]);
        }
    }
#pragma unroll
    for (int ka = 0; ka < 8; ka++) bfly4<+1>(u + 4 * ka);  // over nb: u[4*ka + kb] = X[ka + 8*kb]
}

template <int R, int DIR> __device__ __forceinline__ void bfly(float2 *u)
{
    if (R == 2) bfly2<DIR>(u);
    else if (R == 4) bfly4<DIR>(u);
    else bfly8<DIR>(u);
}

template <int T> __device__ __forceinline__ void fft_sync()
{
    if (T == 64) wave_sync();
    else __syncthreads();
}

// One Stockham pass.  P = product of the radices of the passes before this one.
template <int N, int T, int R, int P, int DIR, bool FIRST, bool LAST>
__device__ __forceinline__ void fft_pass(float2 (&x)[N / T], float2 *lds, const float2 *__restrict__ tw, int tid)
{
    constexpr int E = N / T, Q = E / R;
    static_assert(E % R == 0 && Q >= 1, "points per work-item must be a multiple of the radix");
    // the image written by a leading radix-4 pass / by the wave transform's leading radix-32 pass
    constexpr int PAD_IN = (!FIRST && P == 4) ? kPad1per16 : (!FIRST && T == 64 && P == 32) ? kPad1per32 : kPad4per32;
    constexpr int PAD_OUT = (FIRST && R == 4) ? kPad1per16 : kPad4per32;
    // Twiddle base powers w1, w2, w4 come from the table; when P <= T the index k = b & (P-1) is the same for all Q
    // butterflies of this work-item, so they are fetched once per pass.  They are fetched BEFORE the gather: LDS
    // returns data in issue order, and a twiddle read queued behind the gather would make the first butterfly wait
    // for all E elements instead of its own R.
    float2 w1 = make_float2(1.f, 0.f), w2 = w1, w4 = w1;
    auto fetch_tw = [&](int k) {
        // Twiddles are loop-invariant in a kernel that transforms frame after frame; hoisting all of them costs
        // dozens of VGPRs (spills at 2 waves/SIMD).  Keep them as table loads next to their use instead.
        if (T == 64) opaque(k);
        const float2 *t = tw + fft_tw_off(N, P) + k;
        w1 = twid<DIR>(t[0]);
        if (R >= 4) w2 = twid<DIR>(t[P]);
        if (R == 8) w4 = twid<DIR>(t[2 * P]);
    };
    if (P > 1) fetch_tw(tid & (P - 1));  // for P > T this is butterfly 0's (k = tid); the others fetch theirs below
    if (!FIRST) {
        {
            // lpad(tid + T*m) == lpad(tid) + lpad(T*m) because T is a multiple of 32: one base, immediate offsets
            const float2 *rp = lds + lpad_sel<PAD_IN>(tid);
            // in butterfly order, two butterflies at a time (their elements are adjacent: one ds_read2 serves both)
#pragma unroll
            for (int q = 0; q < Q; q += 2) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    x[q + r * Q] = rp[lpad_sel<PAD_IN>(T * (q + r * Q))];
                    if (q + 1 < Q) x[q + 1 + r * Q] = rp[lpad_sel<PAD_IN>(T * (q + 1 + r * Q))];
                }
                if (T == 64) sched_fence();
            }
        }
        fft_sync<T>();  // all gathers done before any work-item scatters again
    }
    const int jbase = (P <= T) ? (tid - (tid & (P - 1))) * R + (tid & (P - 1)) : tid;
    float2 *wbase = lds + lpad_sel<PAD_OUT>(jbase);
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int b = tid + T * q;
        float2 u[R];
#pragma unroll
        for (int r = 0; r < R; r++) u[r] = x[q + r * Q];
        const int k = b & (P - 1);
        if (P > 1) {
            if (P > T && q > 0) fetch_tw(k);
            // twiddle first: cmul builds (-a.y, a.y) from its first argument, which is per-pass constant when P <= T
            u[1] = cmul(w1, u[1]);
            if (R >= 4) {
                const float2 w3 = cmul(w1, w2);
                u[2] = cmul(w2, u[2]);
                u[3] = cmul(w3, u[3]);
                if (R == 8) {
                    u[4] = cmul(w4, u[4]);
                    u[5] = cmul(cmul(w4, w1), u[5]);
                    u[6] = cmul(cmul(w4, w2), u[6]);
                    u[7] = cmul(cmul(w4, w3), u[7]);
                }
            }
        }
        bfly<R, DIR>(u);
        if (LAST) {
            // P == N/R here, so k == b and output r lands on element b + r*N/R = tid + T*(q + r*Q)
#pragma unroll
            for (int r = 0; r < R; r++) x[q + r * Q] = u[r];
        } else {
            // lpad(j + r*P) == lpad(j) + lpad(r*P): (j mod 32) + (r*P mod 32) never carries for power-of-two P
            // (j = P*R*c + k with k < P), so the R scatters share one base register and use immediate offsets.
            // j itself splits into a per-pass base (wbase, from tid) plus a per-q constant that is a multiple of T.
            const int cq = (P <= T) ? T * R * q : R * (T * q - (T * q) % P) + (T * q) % P;
            float2 *wp = wbase + lpad_sel<PAD_OUT>(cq);
#pragma unroll
            for (int r = 0; r < R; r++) wp[lpad_sel<PAD_OUT>(r * P)] = u[r];
        }
        if (T == 64) sched_fence();  // one butterfly at a time: interleaving all Q of them multiplies the live set
    }
    if (!LAST) fft_sync<T>();
}

// Leading pass of the 2048-point wave transform (T = 64): the 32 strided elements a lane holds are exactly one radix-32
// butterfly (inputs tid + 64 r), no twiddles; output k goes to element 32*tid + k.  One exchange fewer than 4*8*8*8 --
// an LDS write costs a wave 24 cycles per 8 bytes per lane on this part (the VGPR-to-LDS path), three times a read.
__device__ __forceinline__ void fft_first32(float2 (&x)[32], float2 *lds, int tid)
{
    dft32(x);
    float2 *wp = lds + lpad1(32 * tid);  // lpad1(32*tid + k) == lpad1(32*tid) + k for k < 32
#pragma unroll
    for (int k = 0; k < 32; k++) wp[k] = x[perm32(k)];
    wave_sync();
}

// Whole transform, strided registers in and out.  lds: FftLds<N>::kSlots float2 (private to the T work-items).
template <int N, int DIR, int T = 256>
__device__ __forceinline__ void fft_regs(float2 (&x)[N / T], float2 *lds, const float2 *__restrict__ tw, int tid)
{
    static_assert((T == 256 && (N == 2048 || N == 4096 || N == 8192)) || (T == 64 && N == 2048), "supported sizes");
    if (N == 2048 && T == 64) {
        static_assert(!(N == 2048 && T == 64) || DIR == +1, "the wave transform is forward only");
        fft_first32(reinterpret_cast<float2 (&)[32]>(x), lds, tid);  // radix 32 * 8 * 8
        fft_pass<N, T, 8, 32, DIR, false, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 256, DIR, false, true>(x, lds, tw, tid);
    } else if (N == 2048) {
        fft_pass<N, T, 4, 1, DIR, true, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 4, DIR, false, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 32, DIR, false, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 256, DIR, false, true>(x, lds, tw, tid);
    } else if (N == 4096) {
        fft_pass<N, T, 8, 1, DIR, true, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 8, DIR, false, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 64, DIR, false, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 512, DIR, false, true>(x, lds, tw, tid);
    } else {
        fft_pass<N, T, 2, 1, DIR, true, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 2, DIR, false, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 16, DIR, false, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 128, DIR, false, false>(x, lds, tw, tid);
        fft_pass<N, T, 8, 1024, DIR, false, true>(x, lds, tw, tid);
    }
}

}  // namespace pg
