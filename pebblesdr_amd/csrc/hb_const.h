// hb_const.h -- the halfband coefficient tables (generated data, hb_taps.inc) as compile-time constants: with a constant design and
// tap index the load folds to a literal, so the coefficients of a chain need no registers and no loads.
#pragma once
#include <hip/hip_runtime.h>

namespace pg {
namespace hbc {
#include "hb_taps.inc"
}
// (the table runs cic3, hb11, hb15 ... hb51 in steps of four taps, then hb59)
template <int T> __device__ __forceinline__ float hb_tap(int p)
{
    static_assert(T == 59 || (T >= 11 && T <= 51 && (T - 7) % 4 == 0), "no such halfband in the reference's table");
    return (float)hbc::pebble_hb_designs[T == 59 ? 12 : (T - 7) / 4].h[p];
}
}  // namespace pg
