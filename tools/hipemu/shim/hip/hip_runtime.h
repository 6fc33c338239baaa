// DEVELOPER-ONLY shim: lets g++ compile pebblesdr_amd/csrc/*.hip against tools/hipemu/hip_emu.h.
// Only tools/hipemu/build_emu.sh puts this directory on the include path; hipcc never sees it.
#pragma once
#include "../../hip_emu.h"
namespace hipemu { static char dyn_shared[160 * 1024] __attribute__((aligned(16))); }
#define HIP_DYNAMIC_SHARED(type, var) type *var = reinterpret_cast<type *>(hipemu::dyn_shared);
static inline int __ffs(int v) { return __builtin_ffs(v); }
template <class... KA, class... A>
static inline void hipLaunchKernelGGL(void (*k)(KA...), dim3 g, dim3 b, size_t lds, hipStream_t, A... a)
{
    if (lds > sizeof(hipemu::dyn_shared)) { fprintf(stderr, "hipemu: %zu bytes of dynamic LDS\n", lds); abort(); }
    hipemu::launch(g, b, [=]() { k(static_cast<KA>(a)...); });
}
