// DEVELOPER-ONLY shim: lets g++ compile pebblesdr_amd/csrc/*.hip against tools/hipemu/hip_emu.h.
// Only tools/hipemu/build_emu.sh puts this directory on the include path; hipcc never sees it.
#pragma once
#include "../../hip_emu.h"
template <class... KA, class... A>
static inline void hipLaunchKernelGGL(void (*k)(KA...), dim3 g, dim3 b, size_t, hipStream_t, A... a)
{
    hipemu::launch(g, b, [=]() { k(static_cast<KA>(a)...); });
}
