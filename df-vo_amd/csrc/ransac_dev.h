// Device helpers shared by the RANSAC kernels (solver_ransac.hip, solver_pnp.hip).
#pragma once
#include "dfvo_common.h"
#include "solver.h"
#include "solver_math.h"

namespace dfvo {

// ------------------------------------------------------------------------------------------------
// wave64 integer sum: ds_swizzle butterflies inside 32-lane halves, then one cross-half exchange
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum(int v) {
    // xor 1,2,4,8,16 within 32 lanes (BitMode swizzle: and_mask 0x1f, or 0, xor k)
    v += __builtin_amdgcn_ds_swizzle(v, 0x041F);
    v += __builtin_amdgcn_ds_swizzle(v, 0x081F);
    v += __builtin_amdgcn_ds_swizzle(v, 0x101F);
    v += __builtin_amdgcn_ds_swizzle(v, 0x201F);
    v += __builtin_amdgcn_ds_swizzle(v, 0x401F);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// replay of RANSACPointSetRegistrator::run over iterations [it0, it1)
__device__ inline void ransac_replay(RansacState* st, const int* __restrict__ nmodels, const int* __restrict__ counts,
                              int max_models, int it0, int it1, int count, int model_points, double confidence) {
    if (st->done) return;
    int iter = st->iter;
    int niters = st->niters;
    int max_good = st->max_good;
    for (; iter < it1 && iter < niters; ++iter) {
        if (st->subset_fail_at >= 0 && iter >= st->subset_fail_at) {
            niters = iter;  // getSubset failed: the CPU loop breaks here
            break;
        }
        const int nm = nmodels[iter];
        for (int m = 0; m < nm; ++m) {
            const int good = counts[iter * max_models + m];
            const int lim = max_good > model_points - 1 ? max_good : model_points - 1;
            if (good > lim) {
                max_good = good;
                st->best_iter = iter;
                st->best_model = m;
                niters = sm::ransac_update_num_iters(confidence, (double)(count - good) / count, model_points, niters);
            }
        }
    }
    st->iter = iter;
    st->niters = niters;
    st->max_good = max_good;
    if (iter >= niters) {
        st->done = 1;
        st->found = max_good > 0 ? 1 : 0;
    }
    (void)it0;
}
// hypothesis chunks enqueued back to back: [0,128) [128,max_iters).  The first chunk almost always ends the run
// (RANSACUpdateNumIters drops below 128 with the first good model); every further chunk is a chain of dependent
// launches that each wait for a free compute unit while the nets run, so there is only one of them.
static inline void chunk_bounds(int max_iters, int* b) {
    b[0] = 0;
    b[1] = max_iters < 128 ? max_iters : 128;
    b[2] = max_iters;
    b[3] = max_iters;
}

}  // namespace dfvo
