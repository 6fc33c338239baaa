// fft_w64.h -- 2048-point forward FFT by ONE wave, 32 points per lane, radix 32 * 4 * 16 with a single LDS exchange.
//
// The two-wave transform of fft_t128.h pays two LDS exchanges (an 8-byte LDS write costs a SIMD 24 cycles, three times a
// read) and a workgroup barrier at each.  Here the middle radix-4 pass runs ACROSS lanes 16 and 32 apart on gfx950's
// v_permlane16_swap / v_permlane32_swap: one swap turns "lane bit" into "register bit" for a register pair, after which the
// butterfly is an ordinary lane-wise add/subtract -- two swaps and no LDS for two index bits.  One exchange is left (lane
// bits 0-3 against four register bits, i.e. a 16 x 16 transpose inside each row of 16 lanes), wave-private, no barrier.
//
//   n = 64 n1 + 16 n2 + n3   (n1 < 32 in registers, n2 = lane >> 4, n3 = lane & 15 on entry: x[n1] = element 64 n1 + lane)
//   k = k1 + 32 k2 + 128 k3
//   pass 1  A[k1]      = sum_n1 x[n1] W32^{n1 k1}                      dft32, in registers
//   twiddle A[k1]     *= wq * W2048^{lane k1}                           (wq: the caller's per-lane factor, e.g. W_{8192}^{lane q})
//   pass 2  B[k1, k2]  = sum_n2 A[k1; n2] W4^{n2 k2}                    two swap stages; afterwards lane bit 5 = k1 bit 4,
//                                                                       lane bit 4 = k1 bit 3, register 4 a + c: a = k1 & 7, k2 = bitrev2(c)
//   twiddle B         *= W64^{n3 k2}
//   exchange (LDS)     lane bits 0-3 (n3) <-> (a, c & 1); half h = c >> 1 at a time through one 8.5 KiB image
//   pass 3  X          = sum_n3 B W16^{n3 k3}                           dft16, twice
// On exit x[16 h + perm16(k3)] = X[w64_kbase(lane) + 32 h + 128 k3].
#pragma once
#include "fft_t128.h"

namespace pg {

constexpr int kW64ImageSlots = 4 * 16 * 17;  // float2 slots: four rows of 16 lanes, 16 x 16 values each, one pad slot per 16

__host__ __device__ constexpr int w64_kbase(int lane)
{
    return (lane & 7) | (((lane >> 4) & 1) << 3) | (((lane >> 5) & 1) << 4) | (((lane >> 3) & 1) << 6);
}

// per-lane constants of the transform (lane l): w1..w4 = W2048^{l}, ^{2l}, ^{3l}, ^{4l}; t1..t3 = W64^{(l & 15) k2}, k2 = 1..3
struct W64Consts {
    float2 wq, w1, w2, w3, w4, t1, t2, t3;
};

__device__ __forceinline__ W64Consts w64_consts(int lane, float2 wq)
{
    W64Consts c;
    c.wq = wq;
    c.w1 = cis_cycles(-(double)lane / 2048.0);
    c.w2 = cis_cycles(-(double)(2 * lane) / 2048.0);
    c.w3 = cis_cycles(-(double)(3 * lane) / 2048.0);
    c.w4 = cis_cycles(-(double)(4 * lane) / 2048.0);
    const int n3 = lane & 15;
    c.t1 = cis_cycles(-(double)n3 / 64.0);
    c.t2 = cis_cycles(-(double)(2 * n3) / 64.0);
    c.t3 = cis_cycles(-(double)(3 * n3) / 64.0);
    return c;
}

// p keeps its lanes 0-31 and receives q's lanes 0-31 in its lanes 32-63; q receives p's lanes 32-63 in its lanes 0-31
__device__ __forceinline__ void swap_lanes32(float2 &p, float2 &q)
{
    const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(p.x), __float_as_uint(q.x), false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(p.y), __float_as_uint(q.y), false, false);
    p = make_float2(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
    q = make_float2(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}
// the same inside each half: rows of 16 lanes, p's odd rows against q's even rows
__device__ __forceinline__ void swap_lanes16(float2 &p, float2 &q)
{
    const auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(p.x), __float_as_uint(q.x), false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(p.y), __float_as_uint(q.y), false, false);
    p = make_float2(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
    q = make_float2(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}

// DO_LDS / DO_MATH: tools/ubench/fft_w64.hip only
template <bool DO_LDS = true, bool DO_MATH = true>
__device__ __forceinline__ void fft2048_w64(float2 (&x)[32], float2 *img, const W64Consts &c, int lane)
{
    if (DO_MATH) {
        dft32(x);  // A[k1] in x[perm32(k1)]
        // the 32 twiddles wq * W2048^{lane k1} are built on the fly from four powers (31 products; a [q][k1][lane] table read
        // through L2 instead cost more time than the products: 64 KiB per workgroup and frame)
        float2 g = c.wq;
#pragma unroll
        for (int hi = 0; hi < 8; hi++) {
            const int k = 4 * hi;
            x[perm32(k)] = cmul_pk(g, x[perm32(k)]);
            x[perm32(k + 1)] = cmul_pk(cmul_pk(g, c.w1), x[perm32(k + 1)]);
            x[perm32(k + 2)] = cmul_pk(cmul_pk(g, c.w2), x[perm32(k + 2)]);
            x[perm32(k + 3)] = cmul_pk(cmul_pk(g, c.w3), x[perm32(k + 3)]);
            if (hi < 7) g = cmul_pk(g, c.w4);
        }
    }
    // pass 2, first stage: lanes 32 apart (n2 bit 1) against k1 bit 4
#pragma unroll
    for (int a = 0; a < 16; a++) {
        float2 p = x[perm32(a)], q = x[perm32(a + 16)];
        swap_lanes32(p, q);
        if (DO_MATH) {
            x[perm32(a)] = cadd(p, q);
            x[perm32(a + 16)] = csub(p, q);  // still owes W4^{n2 & 1} = -j in the odd rows of 16 lanes: taken in the next stage
        } else {
            x[perm32(a)] = p;
            x[perm32(a + 16)] = q;
        }
    }
    // second stage: lanes 16 apart (n2 bit 0) against k1 bit 3
#pragma unroll
    for (int e = 0; e < 2; e++) {
#pragma unroll
        for (int a = 0; a < 8; a++) {
            float2 u = x[perm32(a + 16 * e)], v = x[perm32(a + 8 + 16 * e)];
            swap_lanes16(u, v);
            if (DO_MATH) {
                // after the swap v holds what sat in odd rows and u what sat in even rows: the owed -j of the differences
                // (e = 1) rides on the butterfly's packed FMAs
                x[perm32(a + 16 * e)] = e ? add_mj<+1>(u, v) : cadd(u, v);
                x[perm32(a + 8 + 16 * e)] = e ? sub_mj<+1>(u, v) : csub(u, v);
            } else {
                x[perm32(a + 16 * e)] = u;
                x[perm32(a + 8 + 16 * e)] = v;
            }
        }
    }
    // register 4 a + c now holds (k1 = a + 8 lane_bit4 + 16 lane_bit5, k2 = bitrev2(c)) for n3 = lane & 15
    if (DO_MATH) {
#pragma unroll
        for (int a = 0; a < 8; a++) {
            x[4 * a + 1] = cmul_pk(c.t2, x[4 * a + 1]);
            x[4 * a + 2] = cmul_pk(c.t1, x[4 * a + 2]);
            x[4 * a + 3] = cmul_pk(c.t3, x[4 * a + 3]);
        }
    }
    const int row = lane >> 4, n3 = lane & 15;
    float2 *wp = img + row * 272 + n3;
    const float2 *rp = img + row * 272 + 17 * n3;
    float2 y[2][16];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (DO_LDS) {
#pragma unroll
            for (int a = 0; a < 8; a++) {
                wp[17 * a] = x[4 * a + 2 * h];
                wp[17 * (a + 8)] = x[4 * a + 2 * h + 1];
            }
            wave_sync();
#pragma unroll
            for (int i = 0; i < 16; i++) y[h][i] = rp[i];
            wave_sync();
        } else {
#pragma unroll
            for (int a = 0; a < 8; a++) {
                y[h][a] = x[4 * a + 2 * h];
                y[h][a + 8] = x[4 * a + 2 * h + 1];
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (DO_MATH) dft16(y[h]);
#pragma unroll
        for (int i = 0; i < 16; i++) x[16 * h + i] = y[h][i];
    }
}

}  // namespace pg
