// tail_refresh.h -- the body of the tail-refresh launch (k_save_tails), shared with kernels that carry it in extra workgroups.
#pragma once
#include "params.h"

namespace pg {

// All history tails of a call: buf[c][-hist + j] = buf[c][n - hist + j], j < hist, for job z and channel y (or into a separate
// destination).  n may be shorter than hist (then part of the old history is kept, shifted), so a workgroup first reads every
// value it will write.  Executed by ALL work-items of a workgroup (one barrier); the first 256 do the work; hist <= 256 * 32.
// z == 0 also advances the oscillators when the job list carries an OscAdvance.
__device__ __forceinline__ void save_tails_block(const TailJobs &jobs, int z, int y, int tid)
{
    if (jobs.oa.osc != nullptr && z == 0 && tid == 0 && (uint32_t)y < jobs.oa.osc_count) {
        // OscBank::advance (Mixer's carried phase, mixer.cpp:48-81 in closed form): every later kernel of the stream sees it
        ChanOsc &o = jobs.oa.osc[y];
        double p = o.phase0 + jobs.oa.adv[y];
        p -= floor(p);
        o.phase0 = p >= 1.0 ? 0.0 : p;
        const uint32_t n0 = o.n0 + jobs.oa.adv_n;
        o.n0 = n0 > (uint32_t)kAmpTab ? (uint32_t)kAmpTab : n0;
    }
    if (z >= jobs.count) return;  // (an advance-only launch; uniform over the workgroup)
    const TailJob &tj = jobs.job[z];
    float2 *b = tj.data + (long long)y * tj.pitch;
    float2 keep[32];
    if (tid < 256) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const int j = tid + 256 * k;
            if (j < tj.hist) keep[k] = b[tj.n - tj.hist + j];
        }
    }
    __syncthreads();
    if (tid < 256) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const int j = tid + 256 * k;
            if (j < tj.hist) {
                if (tj.dst) tj.dst[(long long)y * tj.dst_pitch + j] = keep[k];
                else b[-tj.hist + j] = keep[k];
            }
        }
    }
}

}  // namespace pg
