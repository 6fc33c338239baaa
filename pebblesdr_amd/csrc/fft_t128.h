// fft_t128.h -- 2048-point forward FFT by 128 work-items (two waves), 16 points each, radix 16 * 8 * 16.
//
// Same Stockham scheme as fft_lds.h (x[m] holds element t + 128*m on entry and on exit, two LDS exchanges), but half
// the registers per work-item of the one-wave transform: a kernel built on it fits four waves per SIMD where the
// one-wave transform fits two.  The price is a workgroup barrier at each exchange (the two waves of a transform run in
// a larger workgroup).  Twiddles: a table of 560 entries -- pass B: W^{16k}, W^{32k}, W^{64k} for k < 16; pass C:
// W^{k}, W^{2k}, W^{4k}, W^{8k} for k < 128 (make_twiddles_t128 on the host).
#pragma once
#include "fft_lds.h"

namespace pg {

constexpr int kTw128B = 0, kTw128C = 48, kTw128Count = 48 + 512;

__host__ __device__ constexpr int perm16(int k) { return 4 * (k & 3) + (k >> 2); }

// 16-point forward DFT in registers as 4 x 4 (n = 4*na + nb): DFT4 over na, twiddle W16^{nb*ka}, DFT4 over nb.
// X[ka + 4*kb] is left in u[4*ka + kb], i.e. X[k] = u[perm16(k)].
__device__ __forceinline__ void dft16(float2 *u)
{
    constexpr float kC[10] = {1.f, 0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f, 0.f,
                              -0.38268343236508977173f, -0.70710678118654752440f, -0.92387953251128675613f, -1.f, -0.92387953251128675613f};
    constexpr float kS[10] = {0.f, -0.38268343236508977173f, -0.70710678118654752440f, -0.92387953251128675613f, -1.f,
                              -0.92387953251128675613f, -0.70710678118654752440f, -0.38268343236508977173f, 0.f, 0.38268343236508977173f};
#pragma unroll
    for (int nb = 0; nb < 4; nb++) {
        float2 t[4];
#pragma unroll
        for (int na = 0; na < 4; na++) t[na] = u[nb + 4 * na];
        bfly4<+1>(t);
#pragma unroll
        for (int ka = 0; ka < 4; ka++) {
            const int m = nb * ka;
            u[4 * ka + nb] = m == 0 ? t[ka] : m == 4 ? mul_mj<+1>(t[ka]) : cmul(make_float2(kC[m], kS[m]), t[ka]);
        }
    }
#pragma unroll
    for (int ka = 0; ka < 4; ka++) bfly4<+1>(u + 4 * ka);
}

// sync(): a barrier over (at least) the 128 work-items of this transform, executed by all of them.
// DO_LDS / DO_MATH exist for tools/ubench/fft_core.hip only (the exchanges alone, the butterflies alone); kernels use the defaults.
// The seven table entries of a work-item, held by the caller (REGS: nothing is read from tw).  A caller that transforms
// x[n] W_M^{n q} (the pruned zero-padded transform) can fold the per-work-item part of that factor, W_M^{t q}, into these: behind
// pass A it is W^{q r} per butterfly input r of pass B times a factor common to the butterfly, which is W^{q m} per register m of
// pass C -- both power series like the twiddles themselves (kernels_spectrum.h, make_twiddles_t128q).
struct Tw128Regs {
    float2 b1, b2, b4, c1, c2, c4, c8;
    float2 cx[16];  // FULLC: all fifteen twiddles of pass C (cx[m] for register m: the very products the pass would form), formed once by the caller
};
__device__ __forceinline__ void tw128_fill_cx(Tw128Regs &w)
{
    w.cx[0] = make_float2(1.f, 0.f);
    w.cx[1] = w.c1; w.cx[2] = w.c2; w.cx[3] = cmul_pk(w.c1, w.c2); w.cx[4] = w.c4;
    w.cx[5] = cmul_pk(w.c4, w.c1); w.cx[6] = cmul_pk(w.c4, w.c2); w.cx[7] = cmul_pk(w.c4, w.cx[3]); w.cx[8] = w.c8;
#pragma unroll
    for (int i = 1; i < 8; i++) w.cx[8 + i] = cmul_pk(w.c8, w.cx[i]);
}

// HELD (REGS, not FULLC): the first HELD of pass C's seven products w8 * w_i come from wr.cx[9 ..] (formed once by the caller) -- as many
// as the caller's register budget allows
template <class Sync, bool DO_LDS = true, bool DO_MATH = true, bool REGS = false, bool FULLC = false, int HELD = 0>
__device__ __forceinline__ void fft2048_t128(float2 (&x)[16], float2 *lds, const float2 *__restrict__ tw, int t, Sync sync, Tw128Regs wr = Tw128Regs())
{
    // ---- pass A: radix 16, the 16 strided elements of a work-item are one butterfly; output k -> element 16 t + k ----
    if (DO_MATH) dft16(x);
    if (DO_LDS) {
        float2 *wp = lds + lpad4(16 * t);  // one pad slot per 16: work-item stride 17 slots, 16 of them cover all banks
#pragma unroll
        for (int k = 0; k < 16; k++) wp[k] = x[perm16(k)];
    }
    sync();
    // ---- pass B: radix 8, P = 16, two butterflies ----
    {
        const int k = t & 15;
        float2 w1, w2, w4;
        if (REGS) { w1 = wr.b1; w2 = wr.b2; w4 = wr.b4; }
        else { w1 = tw[kTw128B + k]; w2 = tw[kTw128B + 16 + k]; w4 = tw[kTw128B + 32 + k]; }
        if (DO_LDS) {
            const float2 *rp = lds + lpad4(t);  // lpad4(t + 128 m) = lpad4(t) + 136 m
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = rp[136 * m];
        }
        sync();  // every gather done before anyone scatters into the same image
        const float2 w3 = cmul_pk(w1, w2), w5 = cmul_pk(w4, w1), w6 = cmul_pk(w4, w2), w7 = cmul_pk(w4, w3);
        float2 *wbase = lds + lpad((t - k) * 8 + k);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            float2 u[8];
#pragma unroll
            for (int r = 0; r < 8; r++) u[r] = x[q + 2 * r];
            if (DO_MATH) {
                // (cmul_pk: the twiddle as it lies in its registers; cmul would build (-w.y, w.y) with two more instructions each)
                u[1] = cmul_pk(w1, u[1]); u[2] = cmul_pk(w2, u[2]); u[3] = cmul_pk(w3, u[3]); u[4] = cmul_pk(w4, u[4]);
                u[5] = cmul_pk(w5, u[5]); u[6] = cmul_pk(w6, u[6]); u[7] = cmul_pk(w7, u[7]);
                bfly8<+1>(u);
            }
            float2 *wp = wbase + lpad(1024 * q);
            if (DO_LDS) {
#pragma unroll
                for (int r = 0; r < 8; r++) wp[lpad(16 * r)] = u[r];  // (j mod 32) + (16 r mod 32) never carries: j mod 32 < 16
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) x[q + 2 * r] = u[r];
            }
        }
    }
    sync();
    // ---- pass C: radix 16, P = 128 = T, k = t; output r -> element t + 128 r (the register layout) ----
    {
        float2 w1, w2, w4, w8;
        if (REGS) { w1 = wr.c1; w2 = wr.c2; w4 = wr.c4; w8 = wr.c8; }
        else { w1 = tw[kTw128C + t]; w2 = tw[kTw128C + 128 + t]; w4 = tw[kTw128C + 256 + t]; w8 = tw[kTw128C + 384 + t]; }
        if (DO_LDS) {
            const float2 *rp = lds + lpad(t);  // lpad(t + 128 m) = lpad(t) + 144 m
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = rp[144 * m];
        }
        sync();  // the image may be overwritten (the caller parks its results there)
        if (!DO_MATH) return;
        if (REGS && FULLC) {
#pragma unroll
            for (int m = 1; m < 16; m++) x[m] = cmul_pk(wr.cx[m], x[m]);
        } else {
        const float2 w3 = cmul_pk(w1, w2), w5 = cmul_pk(w4, w1), w6 = cmul_pk(w4, w2), w7 = cmul_pk(w4, w3);
        x[1] = cmul_pk(w1, x[1]); x[2] = cmul_pk(w2, x[2]); x[3] = cmul_pk(w3, x[3]); x[4] = cmul_pk(w4, x[4]);
        x[5] = cmul_pk(w5, x[5]); x[6] = cmul_pk(w6, x[6]); x[7] = cmul_pk(w7, x[7]); x[8] = cmul_pk(w8, x[8]);
        x[9] = cmul_pk(REGS && HELD >= 1 ? wr.cx[9] : cmul_pk(w8, w1), x[9]);
        x[10] = cmul_pk(REGS && HELD >= 2 ? wr.cx[10] : cmul_pk(w8, w2), x[10]);
        x[11] = cmul_pk(REGS && HELD >= 3 ? wr.cx[11] : cmul_pk(w8, w3), x[11]);
        x[12] = cmul_pk(REGS && HELD >= 4 ? wr.cx[12] : cmul_pk(w8, w4), x[12]);
        x[13] = cmul_pk(REGS && HELD >= 5 ? wr.cx[13] : cmul_pk(w8, w5), x[13]);
        x[14] = cmul_pk(REGS && HELD >= 6 ? wr.cx[14] : cmul_pk(w8, w6), x[14]);
        x[15] = cmul_pk(REGS && HELD >= 7 ? wr.cx[15] : cmul_pk(w8, w7), x[15]);
        }
        dft16(x);
        float2 y[16];
#pragma unroll
        for (int r = 0; r < 16; r++) y[r] = x[perm16(r)];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = y[r];
    }
}

}  // namespace pg
