// Wavefront-parallel RANSAC for the essential matrix (five-point) and the homography (4-point DLT),
// plus recoverPose / triangulation.  Replaces the OpenCV CPU calls of
//   /root/reference/libs/tracker/E_tracker.py:199-205 (cv2.findHomography), :231-239
//   (cv2.findEssentialMat), :292-295 (cv2.recoverPose), libs/geometry/ops_3d.py:63 (triangulatePoints).
//
// OpenCV's RANSAC loop is sequential and adaptive (niters shrinks when a better model appears).  The
// parity-preserving parallelisation used here:
//   1. the subset sequence comes from one sequential cv::RNG stream -> one lane generates the subsets
//      of a chunk of iterations (kernels k_*_subsets);
//   2. every hypothesis of the chunk is solved in parallel, ONE HYPOTHESIS PER LANE for the minimal
//      solver (k_e_stage1, k_h_solve: pure register/scratch f64 arithmetic; the polynomial + root stage of the
//      five-point solver, k_e_poly_stage3, runs one ROOT per lane, sixteen lanes per hypothesis), then
//      ONE HYPOTHESIS PER WAVEFRONT for inlier scoring (k_e_score/k_h_score: 64 lanes stride over the
//      correspondences, counts reduced across the wave with ds_swizzle/DPP butterflies);
//   3. a single lane replays OpenCV's sequential "goodCount > max(best, modelPoints-1) -> update best,
//      niters = RANSACUpdateNumIters(...)" scan over the per-hypothesis counts (k_replay), which yields
//      the same winner and the same stopping iteration as the CPU loop;
//   4. chunks (128, 384, rest) are enqueued back to back; once the replay has passed `niters` the
//      remaining chunks exit at once (device-side flag, no host round trip).
// All arithmetic is f64 (scores are cast to float exactly where OpenCV casts) and this file is built
// with -ffp-contract=off, so masks are bit-identical to the sequential algorithm.
#include <cstring>

#include "dfvo_common.h"
#include "h_refine_dev.h"  // jacobi_eigen_coop, HRefineShared / HRefineWork, h_refit_refine_block
#include "ransac_dev.h"
#include "solver.h"
#include "solver_math.h"
#include "solver_poly_lanes.h"
#include "tracker.h"

namespace dfvo {

// ------------------------------------------------------------------------------------------------
// shared RANSAC bookkeeping
// ------------------------------------------------------------------------------------------------
__global__ void k_ransac_init(RansacState* st, int max_iters, uint64_t seed) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->rng_state = seed ? seed : 0xffffffffULL;
    st->niters = max_iters > 1 ? max_iters : 1;
    st->iter = 0;
    st->max_good = 0;
    st->best_iter = -1;
    st->best_model = -1;
    st->done = 0;
    st->subset_fail_at = -1;
    st->found = 0;
}

__global__ void k_replay(RansacState* st, const int* __restrict__ nmodels, const int* __restrict__ counts,
                         int max_models, int it0, int it1, int count, const int* __restrict__ d_n, int model_points,
                         double confidence) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ransac_replay(st, nmodels, counts, max_models, it0, it1, d_n ? *d_n : count, model_points, confidence);
}

// ================================================================================================
// essential matrix.  The kernels take a batch of independent RANSAC problems over the same number of
// correspondences (the `repeat` shuffled runs of compute_pose_2d2d): blockIdx.y = problem, so one launch
// sequence on one stream serves them all.  cv::RANSAC seeds its RNG with the same constant in every
// call and the five-point sampler has no data-dependent subset check, so the subset indices depend
// only on the point count: they are drawn once and shared by the batch.
// ================================================================================================
__global__ void k_e_normalise(const double* __restrict__ pts, int n, double a, double bx, double by,
                              double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i * 2] = pts[i * 2] * a + bx;
    out[i * 2 + 1] = pts[i * 2 + 1] * a + by;
}

struct ERep {
    RansacState* state;
    const double *pts1, *pts2;  // inputs
    double *norm_a, *norm_b;
    double* ws;
    int* ok;
    double* models;
    int* nmodels;
    int* counts;
    uint8_t* mask;
    double* out;
};
struct EBatch {
    int nrep;
    int* idx;  // shared subset indices [iters][5]
    ERep r[MAX_E_BATCH];
};

// rng_pre (optional): the subsets of the first chunk were drawn ahead (k_e_subsets_pre): the stream continues behind them
__global__ void k_e_init_normalise(const EBatch B, int n, int max_iters, double a, double bx, double by,
                                   const unsigned long long* __restrict__ rng_pre) {
    const ERep& R = B.r[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        RansacState* st = R.state;
        st->rng_state = rng_pre ? *rng_pre : 0xffffffffffffffffULL;
        st->niters = max_iters > 1 ? max_iters : 1;
        st->iter = 0;
        st->max_good = 0;
        st->best_iter = -1;
        st->best_model = -1;
        st->done = 0;
        st->subset_fail_at = -1;
        st->found = 0;
    }
    if (i >= n) return;
    R.norm_a[i * 2] = R.pts1[i * 2] * a + bx;
    R.norm_a[i * 2 + 1] = R.pts1[i * 2 + 1] * a + by;
    R.norm_b[i * 2] = R.pts2[i * 2] * a + bx;
    R.norm_b[i * 2 + 1] = R.pts2[i * 2 + 1] * a + by;
}

// The FIRST chunk's subsets ahead of the call (round 6).  cv::RANSAC seeds its RNG with the same constant in every call and the
// five-point sampler has no data-dependent subset check: the indices are a function of the point COUNT alone.  So they are drawn
// as soon as the keypoint count exists -- on the stream of the RandomState-independent half, beside the homography chain --
// instead of as the fourth dependent launch of the RandomState-ordered chain (50 us of one lane on the path to every pose).
// idx [it1][5]; *rng_out = the stream's state behind them (k_e_init_normalise hands it to every problem of the batch).
__global__ void k_e_subsets_pre(int* __restrict__ idx, unsigned long long* __restrict__ rng_out, const int* __restrict__ d_n, int n_arg,
                                int it1) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int count = d_n ? *d_n : n_arg;
    sm::CvRng rng;
    rng.state = 0xffffffffffffffffULL;
    if (count >= 5)
        for (int it = 0; it < it1; ++it) {
            int* id = idx + it * 5;
            for (int i = 0; i < 5;) {
                int v, j;
                for (;;) {
                    v = id[i] = sm::cvrng_uniform(rng, 0, count);
                    for (j = 0; j < i; j++)
                        if (v == id[j]) break;
                    if (j == i) break;
                }
                i++;
            }
        }
    *rng_out = rng.state;
}

// one lane: subsets of iterations [it0, it1) from the sequential cv::RNG stream, shared by the batch
__global__ void k_e_subsets(const EBatch B, int count, int it0, int it1) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool all_done = true;
    for (int r = 0; r < B.nrep; ++r) all_done = all_done && B.r[r].state->done;
    if (all_done) return;
    sm::CvRng rng;
    rng.state = B.r[0].state->rng_state;
    for (int it = it0; it < it1; ++it) {
        int* id = B.idx + it * 5;
        for (int i = 0; i < 5;) {
            int v, j;
            for (;;) {
                v = id[i] = sm::cvrng_uniform(rng, 0, count);
                for (j = 0; j < i; j++)
                    if (v == id[j]) break;
                if (j == i) break;
            }
            i++;
        }
    }
    for (int r = 0; r < B.nrep; ++r) B.r[r].state->rng_state = rng.state;
}

// one hypothesis per lane, 16 lanes per block: the ~3.7 KB of dense work (9x9 SVD rows, the 10x20
// constraint matrix, the 10x10 LU with its right-hand sides) of each lane sits in an LDS slice of odd
// stride, so the pivoting / Jacobi subscripts never touch scratch memory
constexpr int E_STAGE1_LANES = 16;
constexpr int E_STAGE1_STRIDE = sm::FIVE_POINT_WS + 1;
__global__ __launch_bounds__(E_STAGE1_LANES) void k_e_stage1(const EBatch B, int it0, int it1) {
    __shared__ double s_ws[E_STAGE1_LANES * E_STAGE1_STRIDE];
    const ERep& R = B.r[blockIdx.y];
    const RansacState* st = R.state;
    const int* idx = B.idx;
    const double *p1 = R.norm_a, *p2 = R.norm_b;
    double* ws = R.ws;
    int* ok = R.ok;
    const int it = it0 + blockIdx.x * E_STAGE1_LANES + threadIdx.x;
    if (st->done || it >= it1) return;
    double q1[10], q2[10];
    for (int i = 0; i < 5; i++) {
        const int k = idx[it * 5 + i];
        q1[i * 2] = p1[k * 2];
        q1[i * 2 + 1] = p1[k * 2 + 1];
        q2[i * 2] = p2[k * 2];
        q2[i * 2 + 1] = p2[k * 2 + 1];
    }
    double* w = ws + (size_t)it * E_WS;  // [EE 36 | b 39 | c 11 | roots 20]
    ok[it] = sm::five_point_stage1_ws(q1, q2, w, w + 36, w + 75, s_ws + threadIdx.x * E_STAGE1_STRIDE) ? 1 : 0;
}

// Polynomial + stage 3 in one launch, sixteen lanes (one DPP row) per hypothesis, one root per lane from the first sweep
// of cv::solvePoly to the root's E (solver_poly_lanes.h: the Gauss-Seidel sweep with the updated roots handed over the
// row by DPP broadcasts -- same operands in the same order as the sequential loop, a third of its instruction stream).
// The roots never leave the registers; ws[86..105] is still written for the trace / tests that read it.
__global__ __launch_bounds__(64) void k_e_poly_stage3(const EBatch B, int it0, int it1) {
    const ERep& R = B.r[blockIdx.y];
    const RansacState* st = R.state;
    double* ws = R.ws;
    const int* ok = R.ok;
    double* models = R.models;
    int* nmodels = R.nmodels;
    const int grp = threadIdx.x >> 4, root = threadIdx.x & 15;
    const int it = it0 + blockIdx.x * 4 + grp;
    if (st->done) return;  // uniform over the block
    const bool active = it < it1;
    const bool good = active && ok[it];
    double* w = ws + (size_t)(active ? it : it0) * E_WS;
    double c[11];
#pragma unroll
    for (int i = 0; i < 11; i++) c[i] = good ? w[75 + i] : 1.0;
    int n = 10;
    for (; n > 1; n--)
        if (fabs(c[n]) + 0.0 > DBL_EPSILON) break;
    double rre = 0, rim = 0;
    sm::solve_poly10_row(c, root, good && n == 10, rre, rim);
    if (good && n != 10) {  // leading coefficients ~ 0: the run-time-degree path, first lane of the row (rare)
        if (root == 0) {
            double r_re[10], r_im[10];
            sm::solve_poly10(c, r_re, r_im);
            for (int i = 0; i < 10; i++) {
                w[86 + i] = r_re[i];
                w[96 + i] = r_im[i];
            }
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        if (root < 10) {
            rre = ((volatile double*)w)[86 + root];
            rim = ((volatile double*)w)[96 + root];
        }
    } else if (good && root < 10) {
        w[86 + root] = rre;
        w[96 + root] = rim;
    }
    bool valid = false;
    double Ev[9];
    if (good && root < 10) valid = sm::five_point_root_to_E(w, w + 36, rre, rim, Ev);
    const unsigned long long m = __ballot(valid);
    const unsigned grp_mask = (unsigned)((m >> (grp * 16)) & 0xffffull);
    if (valid) {
        const int slot = __popc(grp_mask & ((1u << root) - 1u));
        double* dst = models + (size_t)it * 90 + slot * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) dst[k] = Ev[k];
    }
    if (active && root == 0) nmodels[it] = __popc(grp_mask);
}

// one wavefront per hypothesis: Sampson error of every correspondence under each of its models
__global__ __launch_bounds__(256) void k_e_score(const EBatch B, int it0, int it1, int n, float thr2) {
    __shared__ double sE[4][90];
    const ERep& R = B.r[blockIdx.y];
    const RansacState* st = R.state;
    const double* models = R.models;
    const int* nmodels = R.nmodels;
    const double *p1 = R.norm_a, *p2 = R.norm_b;
    int* counts = R.counts;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int it = it0 + blockIdx.x * 4 + wave;
    const bool active = !st->done && it < it1;
    const int nm = active ? nmodels[it] : 0;
    for (int k = lane; k < nm * 9; k += 64) sE[wave][k] = models[(size_t)it * 90 + k];
    __syncthreads();
    if (nm <= 0) return;
    int cnt[10];
#pragma unroll
    for (int m = 0; m < 10; m++) cnt[m] = 0;
    for (int i = lane; i < n; i += 64) {
        const double x1 = p1[i * 2], y1 = p1[i * 2 + 1], x2 = p2[i * 2], y2 = p2[i * 2 + 1];
#pragma unroll
        for (int m = 0; m < 10; m++) {
            if (m < nm) {
                const float e = sm::essential_error(&sE[wave][m * 9], x1, y1, x2, y2);
                cnt[m] += (e <= thr2) ? 1 : 0;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 10; m++) {
        if (m < nm) {
            const int s = wave_sum(cnt[m]);
            if (lane == 0) counts[it * 10 + m] = s;
        }
    }
}

__global__ void k_e_replay(const EBatch B, int it0, int it1, int count, double confidence) {
    if (threadIdx.x != 0) return;
    const ERep& R = B.r[blockIdx.x];
    ransac_replay(R.state, R.nmodels, R.counts, 10, it0, it1, count, 5, confidence);
}

__global__ void k_e_mask(const EBatch B, int n, float thr2) {
    const ERep& R = B.r[blockIdx.y];
    const RansacState* st = R.state;
    const double* models = R.models;
    const double *p1 = R.norm_a, *p2 = R.norm_b;
    uint8_t* mask = R.mask;
    double* E_out = R.out;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (!st->found) {
        if (i < n) mask[i] = 0;
        return;
    }
    const double* E = models + (size_t)st->best_iter * 90 + st->best_model * 9;
    if (i < 9) E_out[i] = E[i];
    if (i >= n) return;
    const float e = sm::essential_error(E, p1[i * 2], p1[i * 2 + 1], p2[i * 2], p2[i * 2 + 1]);
    mask[i] = e <= thr2 ? 1 : 0;
}

int RansacWorkspace::ensure(int n, int max_iters) {
    if (n <= cap_n && max_iters <= cap_iters) return DFVO_OK;
    release();
    cap_n = n > cap_n ? n : cap_n;
    cap_iters = max_iters > cap_iters ? max_iters : cap_iters;
    DFVO_HIP_CHECK(hipMalloc((void**)&state, sizeof(RansacState)));
    DFVO_HIP_CHECK(hipMalloc((void**)&pts_a, sizeof(double) * 2 * cap_n));
    DFVO_HIP_CHECK(hipMalloc((void**)&pts_b, sizeof(double) * 2 * cap_n));
    DFVO_HIP_CHECK(hipMalloc((void**)&norm_a, sizeof(double) * 2 * cap_n));
    DFVO_HIP_CHECK(hipMalloc((void**)&norm_b, sizeof(double) * 2 * cap_n));
    DFVO_HIP_CHECK(hipMalloc((void**)&f_a, sizeof(float) * 2 * cap_n));
    DFVO_HIP_CHECK(hipMalloc((void**)&f_b, sizeof(float) * 2 * cap_n));
    DFVO_HIP_CHECK(hipMalloc((void**)&idx, sizeof(int) * 5 * cap_iters));
    DFVO_HIP_CHECK(hipMalloc((void**)&ws, sizeof(double) * E_WS * cap_iters));
    DFVO_HIP_CHECK(hipMalloc((void**)&ok, sizeof(int) * cap_iters));
    DFVO_HIP_CHECK(hipMalloc((void**)&models, sizeof(double) * 90 * cap_iters));
    DFVO_HIP_CHECK(hipMalloc((void**)&nmodels, sizeof(int) * cap_iters));
    DFVO_HIP_CHECK(hipMalloc((void**)&counts, sizeof(int) * 10 * cap_iters));
    DFVO_HIP_CHECK(hipMalloc((void**)&mask, cap_n));
    DFVO_HIP_CHECK(hipMalloc((void**)&out, sizeof(double) * 64));
    DFVO_HIP_CHECK(hipMalloc((void**)&lm, sizeof(double) * (size_t)(2 * cap_n) * 10 + 4096));
    DFVO_HIP_CHECK(hipMalloc((void**)&cidx, sizeof(int) * (cap_n + 16)));
    return DFVO_OK;
}

void RansacWorkspace::release() {
    void* ptrs[] = {state, pts_a, pts_b, norm_a, norm_b, f_a, f_b, idx, ws, ok, models, nmodels, counts, mask, out, lm, cidx};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    state = nullptr;
    pts_a = pts_b = norm_a = norm_b = nullptr;
    f_a = f_b = nullptr;
    idx = ok = nmodels = counts = cidx = nullptr;
    ws = models = out = lm = nullptr;
    mask = nullptr;
    cap_n = cap_iters = 0;
}


// batch of `nrep` problems (workspaces w[r], inputs d_pts1[r]/d_pts2[r], all with n correspondences).
// Results per problem: w[r].state (RansacState), w[r].out[0..8] = E, w[r].mask[n]
// rng_pre (optional): enqueue_e_subsets_prefetch(w[0], ...) ran for this call's point count: the first chunk's subsets are in
// w[0].idx already, *rng_pre is the sampler's state behind them.
int enqueue_find_essential_batch(RansacWorkspace* w, const double* const* d_pts1, const double* const* d_pts2, int nrep,
                                 int n, double focal, double ppx, double ppy, double prob, double threshold,
                                 int max_iters, hipStream_t s, const unsigned long long* rng_pre) {
    DFVO_ARG_CHECK(n >= 0 && max_iters >= 1 && nrep >= 1 && nrep <= MAX_E_BATCH, "find_essential: bad sizes");
    EBatch B;
    memset(&B, 0, sizeof(B));
    B.nrep = nrep;
    for (int r = 0; r < nrep; ++r) {
        int rc = w[r].ensure(n > 8 ? n : 8, max_iters);
        if (rc != DFVO_OK) return rc;
        ERep& R = B.r[r];
        R.state = w[r].state;
        R.pts1 = d_pts1[r];
        R.pts2 = d_pts2[r];
        R.norm_a = w[r].norm_a;
        R.norm_b = w[r].norm_b;
        R.ws = w[r].ws;
        R.ok = w[r].ok;
        R.models = w[r].models;
        R.nmodels = w[r].nmodels;
        R.counts = w[r].counts;
        R.mask = w[r].mask;
        R.out = w[r].out;
    }
    B.idx = w[0].idx;
    const double a = 1. / focal, bx = -ppx * a, by = -ppy * a;
    threshold /= (focal + focal) / 2;
    const float thr2 = (float)(threshold * threshold);
    const unsigned R = (unsigned)nrep;
    hipLaunchKernelGGL(k_e_init_normalise, dim3(cdiv(n > 0 ? n : 1, 256), R), dim3(256), 0, s, B, n, max_iters, a, bx, by, rng_pre);
    if (n >= 5) {  // count < modelPoints: no model (state->found stays 0)
        // count == modelPoints would run the kernel once on all points; DF-VO never gets there (N >= 10
        // is required upstream), treat it through the generic loop with the single possible subset order.
        int cb[4];
        chunk_bounds(max_iters, cb);
        for (int c = 0; c < 3; ++c) {
            const int it0 = cb[c], it1 = cb[c + 1];
            if (it1 <= it0) continue;
            const int nh = it1 - it0;
            if (!(c == 0 && rng_pre)) hipLaunchKernelGGL(k_e_subsets, dim3(1), dim3(1), 0, s, B, n, it0, it1);
            hipLaunchKernelGGL(k_e_stage1, dim3(cdiv(nh, E_STAGE1_LANES), R), dim3(E_STAGE1_LANES), 0, s, B, it0, it1);
            // polynomial + stage 3: one root per lane (round 4; one lane per hypothesis + a separate stage-3 launch before:
            // 1.10 -> 0.70 ms per pair, profiles/r4b_kernel_stats_poly_*.csv)
            hipLaunchKernelGGL(k_e_poly_stage3, dim3(cdiv(nh, 4), R), dim3(64), 0, s, B, it0, it1);
            hipLaunchKernelGGL(k_e_score, dim3(cdiv(nh, 4), R), dim3(256), 0, s, B, it0, it1, n, thr2);
            hipLaunchKernelGGL(k_e_replay, dim3(R), dim3(1), 0, s, B, it0, it1, n, prob);
        }
    }
    hipLaunchKernelGGL(k_e_mask, dim3(cdiv(n > 9 ? n : 9, 256), R), dim3(256), 0, s, B, n, thr2);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

int enqueue_find_essential(RansacWorkspace& w, const double* d_pts1, const double* d_pts2, int n, double focal,
                           double ppx, double ppy, double prob, double threshold, int max_iters, hipStream_t s) {
    return enqueue_find_essential_batch(&w, &d_pts1, &d_pts2, 1, n, focal, ppx, ppy, prob, threshold, max_iters, s, nullptr);
}

int enqueue_e_subsets_prefetch(RansacWorkspace& w0, const int* d_n, int n_bound, int max_iters, unsigned long long* d_rng_pre,
                               hipStream_t s) {
    DFVO_ARG_CHECK(d_rng_pre && max_iters >= 1 && n_bound >= 0, "e_subsets_prefetch: bad argument");
    int rc = w0.ensure(n_bound > 8 ? n_bound : 8, max_iters);
    if (rc != DFVO_OK) return rc;
    int cb[4];
    chunk_bounds(max_iters, cb);
    hipLaunchKernelGGL(k_e_subsets_pre, dim3(1), dim3(1), 0, s, w0.idx, d_rng_pre, d_n, n_bound, cb[1]);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ================================================================================================
// homography
// ================================================================================================
__global__ void k_to_float(const double* __restrict__ a, int n, float* __restrict__ o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = (float)a[i];
}

// findHomography prologue in one launch: RANSAC state reset + both point sets converted to float (blockIdx.y)
// d_n (optional): the point count lives on the device (fused pipeline, chain enqueued before the host knows it);
// then n_arg is only the launch bound.  Fewer than 5 points: no model (state done, found = 0).
__global__ void k_h_init_to_float(RansacState* st, int max_iters, const double* __restrict__ a, const double* __restrict__ b,
                                  int n_arg, const int* __restrict__ d_n, float* __restrict__ fa, float* __restrict__ fb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = d_n ? *d_n : n_arg;
    const int n2 = 2 * n;
    if (i == 0 && blockIdx.y == 0) {
        st->rng_state = 0xffffffffffffffffULL;
        st->niters = max_iters > 1 ? max_iters : 1;
        st->iter = 0;
        st->max_good = 0;
        st->best_iter = -1;
        st->best_model = -1;
        st->done = n < 5 ? 1 : 0;
        st->subset_fail_at = -1;
        st->found = 0;
    }
    if (i >= n2) return;
    if (blockIdx.y == 0)
        fa[i] = (float)a[i];
    else
        fb[i] = (float)b[i];
}

// one lane: subsets with HomographyEstimatorCallback::checkSubset, up to 10000 attempts each
// (Round 6: the two point sets copied into LDS by 256 threads before lane 0 walks the sampler -- no dependent global reads per
// draw -- measured no faster: 167 vs 150 us for the first chunk of 128 subsets, profiles/r6t_h_subsets_lds.txt.  The ~1.2 us per
// subset are the sampler's own dependent arithmetic: four 64-bit modulo draws and checkSubset's collinearity / orientation tests
// in double.  Reverted.)
__global__ void k_h_subsets(RansacState* st, int* __restrict__ idx, const float* __restrict__ src,
                            const float* __restrict__ dst, int count_arg, const int* __restrict__ d_n, int it0, int it1) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->done || st->subset_fail_at >= 0) return;
    const int count = d_n ? *d_n : count_arg;
    sm::CvRng rng;
    rng.state = st->rng_state;
    for (int it = it0; it < it1; ++it) {
        int* id = idx + it * 4;
        float ms1[8], ms2[8];
        int i = 0, iters = 0;
        const int maxAttempts = 10000;
        for (; iters < maxAttempts; iters++) {
            for (i = 0; i < 4 && iters < maxAttempts;) {
                int v, j;
                for (;;) {
                    v = id[i] = sm::cvrng_uniform(rng, 0, count);
                    for (j = 0; j < i; j++)
                        if (v == id[j]) break;
                    if (j == i) break;
                }
                ms1[i * 2] = src[v * 2];
                ms1[i * 2 + 1] = src[v * 2 + 1];
                ms2[i * 2] = dst[v * 2];
                ms2[i * 2 + 1] = dst[v * 2 + 1];
                i++;
            }
            if (i == 4 && !sm::homography_check_subset(ms1, ms2)) continue;
            break;
        }
        if (!(i == 4 && iters < maxAttempts)) {
            st->subset_fail_at = it;
            break;
        }
    }
    st->rng_state = rng.state;
}



// one 4-point hypothesis per wavefront: normalisation constants on lane 0, the 45 LtL entries one per lane, the
// 9 x 9 eigen decomposition cooperatively (jacobi_eigen_coop), de-normalisation on lane 0
__global__ __launch_bounds__(64) void k_h_solve(const RansacState* st, const int* __restrict__ idx,
                                                 const float* __restrict__ src, const float* __restrict__ dst, int it0,
                                                 int it1, double* __restrict__ models, int* __restrict__ nmodels) {
    __shared__ double s_LtL[81], s_W[9], s_V[81], s_norm[8];
    __shared__ int s_ok;
    const int it = it0 + blockIdx.x, lane = threadIdx.x;
    if (st->done || it >= it1) return;
    if (st->subset_fail_at >= 0 && it >= st->subset_fail_at) {
        if (lane == 0) nmodels[it] = 0;
        return;
    }
    float M[8], m[8];
    for (int i = 0; i < 4; i++) {
        const int k = idx[it * 4 + i];
        M[i * 2] = src[k * 2];
        M[i * 2 + 1] = src[k * 2 + 1];
        m[i * 2] = dst[k * 2];
        m[i * 2 + 1] = dst[k * 2 + 1];
    }
    if (lane == 0) {  // HomographyEstimatorCallback::runKernel normalisation (sequential sums over the 4 points)
        sm::HNorm h;
        h.cMx = h.cMy = h.cmx = h.cmy = h.sMx = h.sMy = h.smx = h.smy = 0;
        for (int i = 0; i < 4; i++) {
            h.cmx += m[i * 2];
            h.cmy += m[i * 2 + 1];
            h.cMx += M[i * 2];
            h.cMy += M[i * 2 + 1];
        }
        h.cmx /= 4;
        h.cmy /= 4;
        h.cMx /= 4;
        h.cMy /= 4;
        for (int i = 0; i < 4; i++) {
            h.smx += fabs(m[i * 2] - h.cmx);
            h.smy += fabs(m[i * 2 + 1] - h.cmy);
            h.sMx += fabs(M[i * 2] - h.cMx);
            h.sMy += fabs(M[i * 2 + 1] - h.cMy);
        }
        const bool deg = fabs(h.smx) < DBL_EPSILON || fabs(h.smy) < DBL_EPSILON || fabs(h.sMx) < DBL_EPSILON ||
                         fabs(h.sMy) < DBL_EPSILON;
        s_ok = deg ? 0 : 1;
        if (!deg) {
            h.smx = 4 / h.smx;
            h.smy = 4 / h.smy;
            h.sMx = 4 / h.sMx;
            h.sMy = 4 / h.sMy;
        }
        s_norm[0] = h.cMx; s_norm[1] = h.cMy; s_norm[2] = h.cmx; s_norm[3] = h.cmy;
        s_norm[4] = h.sMx; s_norm[5] = h.sMy; s_norm[6] = h.smx; s_norm[7] = h.smy;
    }
    __syncthreads();
    if (!s_ok) {
        if (lane == 0) nmodels[it] = 0;
        return;
    }
    sm::HNorm h;
    h.cMx = s_norm[0]; h.cMy = s_norm[1]; h.cmx = s_norm[2]; h.cmy = s_norm[3];
    h.sMx = s_norm[4]; h.sMy = s_norm[5]; h.smx = s_norm[6]; h.smy = s_norm[7];
    if (lane < 45) {  // upper-triangle entry (lj, lk): sequential sum over the points (homography_accumulate)
        int lj = 0, rem = lane;
        while (rem >= 9 - lj) {
            rem -= 9 - lj;
            lj++;
        }
        const int lk = lj + rem;
        double acc = 0;
        for (int i = 0; i < 4; i++) {
            const double x = (m[i * 2] - h.cmx) * h.smx, y = (m[i * 2 + 1] - h.cmy) * h.smy;
            const double X = (M[i * 2] - h.cMx) * h.sMx, Y = (M[i * 2 + 1] - h.cMy) * h.sMy;
            const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
            const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
            acc += Lx[lj] * Lx[lk] + Ly[lj] * Ly[lk];
        }
        s_LtL[lj * 9 + lk] = acc;
        s_LtL[lk * 9 + lj] = acc;
    }
    __syncthreads();
    jacobi_eigen_coop<9>(s_LtL, s_W, s_V, lane);
    __syncthreads();
    if (lane == 0) {
        double model[9];
        sm::homography_denormalise(h, s_V + 72, model);
        for (int i = 0; i < 9; i++) models[(size_t)it * 9 + i] = model[i];
        nmodels[it] = 1;
    }
}

__global__ __launch_bounds__(256) void k_h_score(const RansacState* st, int it0, int it1,
                                                  const double* __restrict__ models, const int* __restrict__ nmodels,
                                                  const float* __restrict__ src, const float* __restrict__ dst, int n_arg,
                                                  const int* __restrict__ d_n, float thr2, int* __restrict__ counts) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int it = it0 + blockIdx.x * 4 + wave;
    if (st->done || it >= it1) return;
    if (nmodels[it] <= 0) return;
    const int n = d_n ? *d_n : n_arg;
    float Hf[8];
#pragma unroll
    for (int k = 0; k < 8; k++) Hf[k] = (float)models[(size_t)it * 9 + k];
    int cnt = 0;
    for (int i = lane; i < n; i += 64) {
        const float e = sm::homography_error(Hf, src[i * 2], src[i * 2 + 1], dst[i * 2], dst[i * 2 + 1]);
        cnt += (e <= thr2) ? 1 : 0;
    }
    const int s = wave_sum(cnt);
    if (lane == 0) counts[it] = s;
}

__global__ void k_h_mask(const RansacState* st, const double* __restrict__ models, const float* __restrict__ src,
                         const float* __restrict__ dst, int n, float thr2, uint8_t* __restrict__ mask,
                         double* __restrict__ H_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (!st->found) {
        if (i < n) mask[i] = 0;
        return;
    }
    const double* H = models + (size_t)st->best_iter * 9;
    if (i < 9) H_out[i] = H[i];
    if (i >= n) return;
    float Hf[8];
#pragma unroll
    for (int k = 0; k < 8; k++) Hf[k] = (float)H[k];
    mask[i] = sm::homography_error(Hf, src[i * 2], src[i * 2 + 1], dst[i * 2], dst[i * 2 + 1]) <= thr2 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_h_refine(const RansacState* st, const float* __restrict__ src,
                                                   const float* __restrict__ dst, int n_arg, const int* __restrict__ d_n,
                                                   uint8_t* __restrict__ mask, int* __restrict__ cidx,
                                                   double* __restrict__ lm, double* __restrict__ H_io, int pts_cap,
                                                   const double* __restrict__ models, float thr2) {
    __shared__ HRefineShared sh;
    extern __shared__ float s_allpts[];  // Mx My mx my of every inlier when they fit (pts_cap points)
    __shared__ HRefineWork w;
    long long* const trace = reinterpret_cast<long long*>(lm);  // null unless DFVO_HREFINE_TRACE (see h_refit_refine_block)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (trace && t == 0) trace[0] = wall_clock64();
    const int n = d_n ? *d_n : n_arg;
    if (!st->found) {  // no model: all-zero mask (what k_h_mask writes in the stand-alone path)
        for (int i = t; i < n; i += 256) mask[i] = 0;
        return;
    }
    // ---- inlier mask of the winning model (findInliers on the best hypothesis) + ordered compaction of its indices
    float Hf[8];
    {
        const double* Hb = models + (size_t)st->best_iter * 9;
#pragma unroll
        for (int k = 0; k < 8; k++) Hf[k] = (float)Hb[k];
        if (t < 9) H_io[t] = Hb[t];
    }
    if (t == 0) sh.base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + t;
        bool f = false;
        if (i < n) {
            f = sm::homography_error(Hf, src[i * 2], src[i * 2 + 1], dst[i * 2], dst[i * 2 + 1]) <= thr2;
            mask[i] = f ? 1 : 0;
        }
        const unsigned long long b = __ballot(f);
        if (lane == 0) sh.wave_cnt[wave] = __popcll(b);
        __syncthreads();
        int off = sh.base;
        for (int w = 0; w < wave; w++) off += sh.wave_cnt[w];
        if (f) cidx[off + __popcll(b & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (t == 0) sh.base += sh.wave_cnt[0] + sh.wave_cnt[1] + sh.wave_cnt[2] + sh.wave_cnt[3];
        __syncthreads();
    }
    const int np = sh.base;
    if (np <= 0) return;
    const bool cached = np <= pts_cap;
    if (cached) {
        for (int i = t; i < np; i += 256) {
            const int p = cidx[i];
            s_allpts[i * 4 + 0] = src[p * 2];
            s_allpts[i * 4 + 1] = src[p * 2 + 1];
            s_allpts[i * 4 + 2] = dst[p * 2];
            s_allpts[i * 4 + 3] = dst[p * 2 + 1];
        }
        __syncthreads();
    }
    // points of the chunk starting at c0: the cached copy, or a fresh load into sh.pts
    auto chunk_pts = [&](int c0) -> const float* {
        if (cached) return s_allpts + (size_t)c0 * 4;
        h_chunk_points(sh, src, dst, cidx, c0, np);
        __syncthreads();
        return sh.pts;
    };
    h_refit_refine_block(sh, w, np, chunk_pts, H_io, true, trace);
    if (t < 8) H_io[t] = w.x[t];
    if (t == 8) H_io[8] = w.h[8];
}

// d_n (optional): device-resident point count (n is then the upper bound used for launch sizes)
int enqueue_find_homography(RansacWorkspace& w, const double* d_pts1, const double* d_pts2, int n, double thr,
                            int max_iters, double confidence, hipStream_t s, const int* d_n) {
    DFVO_ARG_CHECK(n >= 0 && max_iters >= 1, "find_homography: bad sizes");
    int rc = w.ensure(n > 8 ? n : 8, max_iters);
    if (rc != DFVO_OK) return rc;
    if (thr <= 0) thr = 3;
    const float thr2 = (float)(thr * thr);
    if (n < 5 && !d_n) {  // n < 4: no model; n == 4 is not reachable from DF-VO (kp count > 10 is checked upstream)
        hipLaunchKernelGGL(k_ransac_init, dim3(1), dim3(1), 0, s, w.state, max_iters, (uint64_t)-1);
        hipLaunchKernelGGL(k_h_mask, dim3(cdiv(n > 9 ? n : 9, 256)), dim3(256), 0, s, w.state, w.models, w.f_a, w.f_b, n,
                           thr2, w.mask, w.out);
        DFVO_HIP_CHECK(hipGetLastError());
        return DFVO_OK;
    }
    hipLaunchKernelGGL(k_h_init_to_float, dim3(cdiv(2 * (n > 0 ? n : 1), 256), 2), dim3(256), 0, s, w.state, max_iters, d_pts1,
                       d_pts2, n, d_n, w.f_a, w.f_b);
    int cb[4];
    chunk_bounds(max_iters, cb);
    for (int c = 0; c < 3; ++c) {
        const int it0 = cb[c], it1 = cb[c + 1];
        if (it1 <= it0) continue;
        const int nh = it1 - it0;
        hipLaunchKernelGGL(k_h_subsets, dim3(1), dim3(1), 0, s, w.state, w.idx, w.f_a, w.f_b, n, d_n, it0, it1);
        hipLaunchKernelGGL(k_h_solve, dim3(nh), dim3(64), 0, s, w.state, w.idx, w.f_a, w.f_b, it0, it1, w.models, w.nmodels);
        hipLaunchKernelGGL(k_h_score, dim3(cdiv(nh, 4)), dim3(256), 0, s, w.state, it0, it1, w.models, w.nmodels, w.f_a,
                           w.f_b, n, d_n, thr2, w.counts);
        hipLaunchKernelGGL(k_replay, dim3(1), dim3(1), 0, s, w.state, w.nmodels, w.counts, 1, it0, it1, n, d_n, 4,
                           confidence);
    }
    // every inlier's coordinates stay in LDS across the LM passes when they fit (16 B / point)
    const int pts_cap = n <= 6144 ? n : 0;
    if (pts_cap)
        if (int rc_lds = ensure_dyn_lds((const void*)k_h_refine, 6144 * 16)) return rc_lds;
    // inlier mask of the winner + refit + LM in one launch
    // DFVO_HREFINE_TRACE=1 (diagnostics): phase timestamps of the launch, printed after a stream synchronisation
    static const bool trace_on = getenv("DFVO_HREFINE_TRACE") != nullptr;
    double* const d_trace = trace_on ? w.lm + (size_t)20 * w.cap_n + 256 : nullptr;
    if (trace_on) DFVO_HIP_CHECK(hipMemsetAsync(d_trace, 0, 24 * sizeof(long long), s));
    hipLaunchKernelGGL(k_h_refine, dim3(1), dim3(256), (size_t)pts_cap * 16, s, w.state, w.f_a, w.f_b, n, d_n, w.mask, w.cidx,
                       d_trace, w.out, pts_cap, w.models, thr2);
    DFVO_HIP_CHECK(hipGetLastError());
    if (trace_on) {
        long long tr[24];
        DFVO_HIP_CHECK(hipMemcpyAsync(tr, d_trace, sizeof(tr), hipMemcpyDeviceToHost, s));
        DFVO_HIP_CHECK(hipStreamSynchronize(s));
        auto us = [&](int a, int b) { return tr[b] && tr[a] ? (tr[b] - tr[a]) * 0.01 : -1.0; };
        fprintf(stderr,
                "k_h_refine trace: inliers %lld | mask+compaction+cache %.1f us | centroids+scales %.1f | LtL sums %.1f | eigen 9x9 %.1f "
                "(%lld rotations) | first J pass %.1f | LM: %lld iterations, eigen 8x8 %.1f (%lld rotations), back-subst %.1f, trial "
                "passes %.1f, step logic %.1f | total %.1f us\n",
                tr[15], us(0, 1), us(1, 2), us(2, 3), us(3, 4), tr[13], us(4, 5), tr[12], tr[8] * 0.01, tr[14], tr[9] * 0.01,
                tr[10] * 0.01, tr[11] * 0.01, us(0, 6));
    }
    return DFVO_OK;
}

// ================================================================================================
// recoverPose / triangulation
// ================================================================================================
// out layout: [0..8] R1, [9..17] R2, [18..20] t, then 4 x 12 projection matrices from 24
__device__ void pose_candidates(const double* __restrict__ E, double* __restrict__ out) {
    double R1[9], R2[9], t[3];
    sm::decompose_essential(E, R1, R2, t);
    for (int i = 0; i < 9; i++) {
        out[i] = R1[i];
        out[9 + i] = R2[i];
    }
    for (int i = 0; i < 3; i++) out[18 + i] = t[i];
    for (int c = 0; c < 4; c++) {
        const double* R = (c & 1) ? R2 : R1;
        const bool neg = c >= 2;
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) out[24 + c * 12 + r * 4 + k] = R[r * 3 + k];
            out[24 + c * 12 + r * 4 + 3] = neg ? -t[r] : t[r];
        }
    }
}

// one lane per (correspondence, candidate): triangulate against the candidate (blockIdx.y), cheirality flag
__global__ void k_cheirality(const double* __restrict__ cand, const double* __restrict__ p1,
                             const double* __restrict__ p2, int n, uint8_t* __restrict__ flags,
                             int* __restrict__ good) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    int f = 0;
    if (i < n) {
        const double* P = cand + 24 + c * 12;
        double X4[4];
        sm::triangulate_point(P0, P, p1[i * 2], p1[i * 2 + 1], p2[i * 2], p2[i * 2 + 1], X4);
        f = sm::cheirality_ok(P, X4, 50.0) ? 1 : 0;
        flags[c * n + i] = f ? 255 : 0;
    }
    const int s = wave_sum(f);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&good[c], s);
}

// recoverPose prologue in one launch: both point sets normalised (blockIdx.y), the four (R, t) candidates of E and
// the zeroed vote counters on thread 0 of block (0, 0)
__global__ void k_pose_prepare(const double* __restrict__ E, const double* __restrict__ p1, const double* __restrict__ p2,
                               int n, double a, double bx, double by, double* __restrict__ n1, double* __restrict__ n2,
                               double* __restrict__ cand, int* __restrict__ good) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double* src = blockIdx.y == 0 ? p1 : p2;
        double* dst = blockIdx.y == 0 ? n1 : n2;
        dst[i * 2] = src[i * 2] * a + bx;
        dst[i * 2 + 1] = src[i * 2 + 1] * a + by;
    }
    if (i == 0 && blockIdx.y == 0) {
        good[0] = good[1] = good[2] = good[3] = 0;
        pose_candidates(E, cand);
    }
}

// fin (optional): PoseState of the fused pipeline; its recoverPose bookkeeping (E_tracker.py:292-300) and the
// inverse pose T21 for the scale stage are written by the same thread that selects the winner
__global__ void k_pose_select(const double* __restrict__ cand, const int* __restrict__ good,
                              const uint8_t* __restrict__ flags, int n, double* __restrict__ out,
                              uint8_t* __restrict__ mask, PoseFinish fin) {
    // out: [0..8] R, [9..11] t, [12] good count (as double)
    const int g0 = good[0], g1 = good[1], g2 = good[2], g3 = good[3];
    int sel;
    if (g0 >= g1 && g0 >= g2 && g0 >= g3)
        sel = 0;
    else if (g1 >= g0 && g1 >= g2 && g1 >= g3)
        sel = 1;
    else if (g2 >= g0 && g2 >= g1 && g2 >= g3)
        sel = 2;
    else
        sel = 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mask[i] = flags[sel * n + i];
    if (i == 0) {
        const double* R = cand + ((sel & 1) ? 9 : 0);
        for (int k = 0; k < 9; k++) out[k] = R[k];
        for (int k = 0; k < 3; k++) out[9 + k] = sel >= 2 ? -cand[18 + k] : cand[18 + k];
        out[12] = (double)good[sel];
        if (fin.ps && fin.ps->major_valid) {
            PoseState* ps = fin.ps;
            const int g = good[sel];
            ps->cheirality = g;
            if ((double)g > (double)ps->n * 0.1) {
                for (int k = 0; k < 9; k++) ps->R[k] = out[k];
                for (int k = 0; k < 3; k++) ps->t[k] = out[9 + k];
            }
        }
        if (fin.ps && fin.T21) {  // T21 = inverse of [R t; 0 1] (E_pose.inv_pose, E_tracker.py:504)
            const double* R = fin.ps->R;
            const double* tt = fin.ps->t;
            double* T = fin.T21;
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) T[r * 4 + c] = R[c * 3 + r];
                T[r * 4 + 3] = -(R[0 * 3 + r] * tt[0] + R[1 * 3 + r] * tt[1] + R[2 * 3 + r] * tt[2]);
            }
            T[12] = T[13] = T[14] = 0;
            T[15] = 1;
        }
    }
}

// d_E: device 9 doubles; points device [n][2].  Results in w.out[16..28] (R, t, count), w.mask
int enqueue_recover_pose(RansacWorkspace& w, const double* d_E, const double* d_pts1, const double* d_pts2, int n,
                         double focal, double ppx, double ppy, hipStream_t s, PoseFinish fin) {
    int rc = w.ensure(n > 8 ? n : 8, w.cap_iters > 0 ? w.cap_iters : 16);
    if (rc != DFVO_OK) return rc;
    DFVO_ARG_CHECK(4 * (size_t)n <= sizeof(double) * 10 * 2 * (size_t)w.cap_n, "recover_pose: workspace");
    const double a = 1. / focal, bx = -ppx * a, by = -ppy * a;
    double* cand = w.lm;                        // 72 doubles
    int* good = (int*)(w.lm + 80);              // 4 ints
    uint8_t* flags = (uint8_t*)(w.lm + 96);     // 4 * n bytes
    hipLaunchKernelGGL(k_pose_prepare, dim3(cdiv(n > 0 ? n : 1, 256), 2), dim3(256), 0, s, d_E, d_pts1, d_pts2, n, a, bx, by,
                       w.norm_a, w.norm_b, cand, good);
    hipLaunchKernelGGL(k_cheirality, dim3(cdiv(n > 0 ? n : 1, 64), 4), dim3(64), 0, s, cand, w.norm_a, w.norm_b, n, flags,
                       good);
    hipLaunchKernelGGL(k_pose_select, dim3(cdiv(n > 0 ? n : 1, 256)), dim3(256), 0, s, cand, good, flags, n, w.out + 16,
                       w.mask, fin);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

__global__ void k_triangulate(const double* __restrict__ P, const double* __restrict__ x1,
                              const double* __restrict__ x2, int n, double* __restrict__ X4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double X[4];
    sm::triangulate_point(P, P + 12, x1[i], x1[n + i], x2[i], x2[n + i], X);
    for (int k = 0; k < 4; k++) X4[k * n + i] = X[k];
}

// d_P: device 24 doubles (P1 | P2); x1, x2 device [2][n]; X4 device [4][n]
int enqueue_triangulate(const double* d_P, const double* d_x1, const double* d_x2, int n, double* d_X4,
                        hipStream_t s) {
    if (n <= 0) return DFVO_OK;
    hipLaunchKernelGGL(k_triangulate, dim3(cdiv(n, 256)), dim3(256), 0, s, d_P, d_x1, d_x2, n, d_X4);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

}  // namespace dfvo
