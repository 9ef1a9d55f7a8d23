// PnP fallback of the tracker on the device: PnpTracker.compute_pose_3d2d
// (/root/reference/libs/tracker/pnp_tracker.py:45-125) = keypoint filtering + unprojection
// (ops_3d.py:70-94), `repeat` x [np.random.shuffle, cv2.solvePnPRansac(iter, reproj_thre)], best by inlier
// count, cv2.Rodrigues.  The numerics follow oracle/cv3_pnp.c statement for statement (see pnp_math.h);
// the parallel decomposition is the one of the five-point path:
//   * the `repeat` problems form one batch (blockIdx.y = repeat) on one stream, subset draws shared;
//   * one EPnP hypothesis per lane (k_pnp_solve; dense 12x12 work in per-lane LDS slices);
//   * one hypothesis per wavefront for reprojection scoring (k_pnp_score), ds_swizzle count reduction;
//   * single-lane replay of the sequential accept / RANSACUpdateNumIters scan;
//   * refinement on the inliers (cvFindExtrinsicCameraParams2: DLT + CvLevMarq) by one workgroup per
//     repeat: points are expanded to rows in LDS by all lanes, every sum that the CPU forms sequentially is
//     owned by one lane that walks the rows in the same order (k_pnp_refine).
// Built with -ffp-contract=off.
#include <cstring>

#include "h_refine_dev.h"  // the homography refit + LM block (planar initialisation)
#include "pnp_math.h"
#include "ransac_dev.h"
#include "tracker.h"

namespace dfvo {

struct PRep {
    RansacState* state;
    const float *obj, *img;  // shuffled float32 points [n][3], [n][2]
    double* models;          // [iters][6] = rvec | tvec
    int *nmodels, *counts;   // [iters]
    uint8_t* mask;           // [n]
    float* pts5;             // compacted inliers [n][5] = X Y Z u v
    PnpRepOut* out;
};
struct PBatch {
    int nrep;
    const int* n_ptr;  // filtered keypoint count (device)
    int* idx;          // shared subset indices [iters][5]
    double K4[4];
    PRep r[MAX_REP];
};

// ------------------------------------------------------------------------------------------------
// filtering + unprojection: ordered compaction of the keypoints that survive the three reference masks
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pnp_filter(const double* __restrict__ kp1, const double* __restrict__ kp2,
                                                     const int* __restrict__ n_in_ptr, int n_in_host,
                                                     const double* __restrict__ depth, int H, int W, PnpConfig cfg,
                                                     double* __restrict__ fk1, double* __restrict__ fk2,
                                                     double* __restrict__ xyz, uint8_t* __restrict__ keep,
                                                     int* __restrict__ info, int depth_per_kp) {
    // depth_per_kp: `depth` holds, per keypoint, the map's value at that keypoint's (truncated, wrapped) kp1 pixel -- [n]
    __shared__ int s_base, s_wave[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = n_in_ptr ? *n_in_ptr : n_in_host;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + t;
        bool f = false;
        double x1 = 0, y1 = 0, x2 = 0, y2 = 0, d = 0;
        if (i < n) {
            x1 = kp1[i * 2];
            y1 = kp1[i * 2 + 1];
            x2 = kp2[i * 2];
            y2 = kp2[i * 2 + 1];
            f = (x2 >= 0) && (x2 < W) && (y2 >= 0) && (y2 < H);
            if (f) {
                int xi = (int)x1, yi = (int)y1;  // astype(int): truncation; python wraps negative indices
                if (xi < 0) xi += W;
                if (yi < 0) yi += H;
                // (a NaN coordinate has no pixel: numpy's astype(int) makes it INT64_MIN and the reference's indexing raises; dropped)
                f = xi >= 0 && xi < W && yi >= 0 && yi < H && x1 == x1 && y1 == y1;
                if (f) {
                    d = depth[depth_per_kp ? (size_t)i : (size_t)yi * W + xi];
                    f = (d != 0) && (d < cfg.max_depth) && (d > cfg.min_depth);
                }
            }
        }
        if (i < n) keep[i] = f ? 1 : 0;
        const unsigned long long b = __ballot(f);
        if (lane == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        if (f) {
            const int o = off + __popcll(b & ((1ull << lane) - 1ull));
            fk1[o * 2] = x1;
            fk1[o * 2 + 1] = y1;
            fk2[o * 2] = x2;
            fk2[o * 2 + 1] = y2;
            // unprojection_kp: (inv_K @ [x, y, 1]) * depth, row by row, left to right
            const double* iK = cfg.inv_K;
            const double X = iK[0] * x1 + iK[1] * y1 + iK[2] * 1.0;
            const double Y = iK[3] * x1 + iK[4] * y1 + iK[5] * 1.0;
            const double Z = iK[6] * x1 + iK[7] * y1 + iK[8] * 1.0;
            xyz[o * 3] = X * d;
            xyz[o * 3 + 1] = Y * d;
            xyz[o * 3 + 2] = Z * d;
        }
        __syncthreads();
        if (t == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (t == 0) info[0] = s_base;
}

// new_XYZ = XYZ[perm], new_kp2 = kp2[perm], converted to float32 as solvePnPRansac does (blockIdx.y = repeat)
__global__ void k_pnp_permute(const int* __restrict__ n_ptr, const int* __restrict__ perm, int perm_stride,
                              const double* __restrict__ xyz, const double* __restrict__ fk2, float* __restrict__ obj,
                              float* __restrict__ img, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_ptr) return;
    const int p = perm[(size_t)blockIdx.y * perm_stride + i];
    float* o = obj + ((size_t)blockIdx.y * cap + i) * 3;
    float* m = img + ((size_t)blockIdx.y * cap + i) * 2;
    o[0] = (float)xyz[p * 3];
    o[1] = (float)xyz[p * 3 + 1];
    o[2] = (float)xyz[p * 3 + 2];
    m[0] = (float)fk2[p * 2];
    m[1] = (float)fk2[p * 2 + 1];
}

// ------------------------------------------------------------------------------------------------
// RANSAC
// ------------------------------------------------------------------------------------------------
__global__ void k_pnp_init(const PBatch B, int max_iters) {
    if (threadIdx.x != 0) return;
    RansacState* st = B.r[blockIdx.x].state;
    const int n = *B.n_ptr;
    st->rng_state = 0xffffffffffffffffULL;
    st->niters = max_iters > 1 ? max_iters : 1;
    st->iter = 0;
    st->max_good = 0;
    st->best_iter = -1;
    st->best_model = -1;
    st->done = n > 4 ? 0 : 1;  // the reference only calls solvePnPRansac with more than 4 points
    st->subset_fail_at = -1;
    st->found = 0;
}

// one lane: the subset draws of iterations [it0, it1), shared by the batch (same seed, same count)
__global__ void k_pnp_subsets(const PBatch B, int it0, int it1) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool all_done = true;
    for (int r = 0; r < B.nrep; ++r) all_done = all_done && B.r[r].state->done;
    if (all_done) return;
    const int count = *B.n_ptr;
    if (count == 5) {  // count == modelPoints: the kernel runs once on all points, nothing is drawn
        for (int i = 0; i < 5; i++) B.idx[i] = i;
        return;
    }
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

constexpr int PNP_SOLVE_LANES = 8;
constexpr int PNP_SOLVE_STRIDE = sm::EPNP_WS + 1;
__global__ __launch_bounds__(PNP_SOLVE_LANES) void k_pnp_solve(const PBatch B, int it0, int it1) {
    __shared__ double s_ws[PNP_SOLVE_LANES * PNP_SOLVE_STRIDE];
    const PRep& R = B.r[blockIdx.y];
    const int it = it0 + blockIdx.x * PNP_SOLVE_LANES + threadIdx.x;
    if (R.state->done || it >= it1) return;
    if (*B.n_ptr == 5 && it > 0) return;  // count == modelPoints: one kernel run
    float obj[15], img[10];
    for (int i = 0; i < 5; i++) {
        const int k = B.idx[it * 5 + i];
        obj[i * 3] = R.obj[k * 3];
        obj[i * 3 + 1] = R.obj[k * 3 + 1];
        obj[i * 3 + 2] = R.obj[k * 3 + 2];
        img[i * 2] = R.img[k * 2];
        img[i * 2 + 1] = R.img[k * 2 + 1];
    }
    double rvec[3], tvec[3];
    sm::epnp_kernel(B.K4, obj, img, rvec, tvec, s_ws + threadIdx.x * PNP_SOLVE_STRIDE);
    double* m = R.models + (size_t)it * 6;
    for (int i = 0; i < 3; i++) {
        m[i] = rvec[i];
        m[3 + i] = tvec[i];
    }
    R.nmodels[it] = 1;
}

// one wavefront per hypothesis: squared reprojection error of every correspondence
__global__ __launch_bounds__(256) void k_pnp_score(const PBatch B, int it0, int it1, float thr2) {
    const PRep& R = B.r[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int it = it0 + blockIdx.x * 4 + wave;
    if (R.state->done || it >= it1) return;
    const int n = *B.n_ptr;
    if (n == 5) return;  // count == modelPoints: accepted without scoring
    const double* m = R.models + (size_t)it * 6;
    double Rm[9];
    sm::rodrigues_v2m(m, Rm, nullptr);
    const double tv[3] = {m[3], m[4], m[5]};
    int cnt = 0;
    for (int i = lane; i < n; i += 64) {
        const float e = sm::pnp_error(Rm, tv, B.K4, R.obj + (size_t)i * 3, R.img + (size_t)i * 2);
        cnt += (e <= thr2) ? 1 : 0;
    }
    const int s = wave_sum(cnt);
    if (lane == 0) R.counts[it] = s;
}

__global__ void k_pnp_replay(const PBatch B, int it0, int it1, double confidence) {
    if (threadIdx.x != 0) return;
    const PRep& R = B.r[blockIdx.x];
    if (*B.n_ptr == 5) {  // count == modelPoints: the single model is accepted, every point is an inlier
        RansacState* st = R.state;
        if (st->done) return;
        st->best_iter = 0;
        st->best_model = 0;
        st->max_good = 5;
        st->iter = 1;
        st->done = 1;
        st->found = 1;
        return;
    }
    ransac_replay(R.state, R.nmodels, R.counts, 1, it0, it1, *B.n_ptr, 5, confidence);
}

__global__ void k_pnp_mask(const PBatch B, float thr2) {
    const PRep& R = B.r[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = *B.n_ptr;
    if (i >= n) return;
    if (!R.state->found) {
        R.mask[i] = 0;
        return;
    }
    if (n == 5) {
        R.mask[i] = 1;
        return;
    }
    const double* m = R.models + (size_t)R.state->best_iter * 6;
    double Rm[9];
    sm::rodrigues_v2m(m, Rm, nullptr);
    const double tv[3] = {m[3], m[4], m[5]};
    const float e = sm::pnp_error(Rm, tv, B.K4, R.obj + (size_t)i * 3, R.img + (size_t)i * 2);
    R.mask[i] = e <= thr2 ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// refinement on the inliers: cvFindExtrinsicCameraParams2 (one workgroup per repeat)
// ------------------------------------------------------------------------------------------------
constexpr int PR_CHUNK = 256;
constexpr int PR_ROW = 24;  // doubles per point in the row buffer
constexpr int PR_UNROLL = 8;

// a += buf[p][i0]*buf[p][i1]; a += buf[p][i2]*buf[p][i3]   per point p, sequentially (products batched)
__device__ __forceinline__ double pr_acc_two(const double* buf, int cnt, int i0, int i1, int i2, int i3, double a) {
    int k = 0;
    for (; k + PR_UNROLL <= cnt; k += PR_UNROLL) {
        double p0[PR_UNROLL], p1[PR_UNROLL];
#pragma unroll
        for (int u = 0; u < PR_UNROLL; ++u) {
            const double* o = buf + (k + u) * PR_ROW;
            p0[u] = o[i0] * o[i1];
            p1[u] = o[i2] * o[i3];
        }
#pragma unroll
        for (int u = 0; u < PR_UNROLL; ++u) {
            a += p0[u];
            a += p1[u];
        }
    }
    for (; k < cnt; k++) {
        const double* o = buf + k * PR_ROW;
        a += o[i0] * o[i1];
        a += o[i2] * o[i3];
    }
    return a;
}

// DFVO_PNP_TRACE (diagnostics): wall_clock64 ticks (100 MHz) of repeat 0's workgroup -- [0] start [1] inliers compacted [2] centroid +
// covariance + 3 x 3 SVD [3] DLT sums (or the planar branch) [4] DLT solved (12 x 12 SVD on one lane) [5] end; sums over the LM loop:
// [8] passes with Jacobian [9] passes without [10] lane-0 steps (6 x 6 SVD solve); [11] passes with J [12] passes without [13] steps
// [14] inliers [15] planar
__device__ long long g_pnp_trace[16];

__global__ __launch_bounds__(256) void k_pnp_refine(const PBatch B, int trace_on) {
    __shared__ double s_buf[PR_CHUNK * PR_ROW];  // per-point rows (DLT: 2 x 12, LM: J 2 x 6 + err 2)
    __shared__ float s_pts[PR_CHUNK * 5];
    __shared__ double s_ws[sm::PNP_DLT_WS];
    __shared__ double s_Mc[3], s_MM[9], s_LL[144], s_JtJ[36], s_JtErr[6], s_param[6], s_prev[6], s_R[9], s_dRdr[27];
    __shared__ double s_errnorm2;
    __shared__ int s_base, s_wave[4], s_flag;
    const PRep& R = B.r[blockIdx.x];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    PnpRepOut* out = R.out;
    const bool tr = trace_on && blockIdx.x == 0 && t == 0;
    auto mark = [&](int i) {
        if (tr) g_pnp_trace[i] = wall_clock64();
    };
    if (tr)
        for (int i = 0; i < 16; i++) g_pnp_trace[i] = 0;
    mark(0);
    if (t == 0) {
        out->flag = 0;
        out->n_inliers = 0;
        out->status = 0;
        out->lm_iters = 0;
    }
    if (!R.state->found) return;
    const int n = *B.n_ptr;
    // ---- ordered compaction of the inliers (float points, as solvePnPRansac hands them on)
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + t;
        const bool f = i < n && R.mask[i] != 0;
        const unsigned long long b = __ballot(f);
        if (lane == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        if (f) {
            float* p = R.pts5 + (size_t)(off + __popcll(b & ((1ull << lane) - 1ull))) * 5;
            p[0] = R.obj[(size_t)i * 3];
            p[1] = R.obj[(size_t)i * 3 + 1];
            p[2] = R.obj[(size_t)i * 3 + 2];
            p[3] = R.img[(size_t)i * 2];
            p[4] = R.img[(size_t)i * 2 + 1];
        }
        __syncthreads();
        if (t == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    const int np = s_base;
    __threadfence_block();
    mark(1);
    if (tr) g_pnp_trace[14] = np;
    auto load_chunk = [&](int c0) {  // points [c0, c0 + 256) -> s_pts
        const int cnt = np - c0 < PR_CHUNK ? np - c0 : PR_CHUNK;
        for (int k = t; k < cnt * 5; k += 256) s_pts[k] = R.pts5[(size_t)c0 * 5 + k];
        return cnt;
    };
    const double fx = B.K4[0], fy = B.K4[1], cx = B.K4[2], cy = B.K4[3], ifx = 1. / fx, ify = 1. / fy;
    // ---- centroid (channel sums times 1/n) and covariance of the object points
    double acc = 0;
    for (int c0 = 0; c0 < np; c0 += PR_CHUNK) {
        const int cnt = load_chunk(c0);
        __syncthreads();
        if (t < 3)
            for (int k = 0; k < cnt; k++) acc += (double)s_pts[k * 5 + t];
        __syncthreads();
    }
    if (t < 3) s_Mc[t] = acc * (1. / np);
    __syncthreads();
    int ma = 0, mb = 0;  // upper-triangle entry of the 3 x 3 covariance owned by lane t < 6
    if (t < 6) {
        const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
        ma = ia[t];
        mb = ib[t];
    }
    acc = 0;
    for (int c0 = 0; c0 < np; c0 += PR_CHUNK) {
        const int cnt = load_chunk(c0);
        __syncthreads();
        if (t < 6) {
            const double ca = s_Mc[ma], cb = s_Mc[mb];
            for (int k = 0; k < cnt; k++) acc += ((double)s_pts[k * 5 + ma] - ca) * ((double)s_pts[k * 5 + mb] - cb);
        }
        __syncthreads();
    }
    if (t < 6) {
        s_MM[ma * 3 + mb] = acc;
        s_MM[mb * 3 + ma] = acc;
    }
    __syncthreads();
    if (t == 0) {
        double* W = s_ws;
        sm::svd_square_t<3>(s_MM, W, s_ws + 3, s_ws + 12);
        s_flag = (W[2] / W[1] < 1e-3 || np < 4) ? 1 : 0;
    }
    __syncthreads();
    const bool planar = s_flag != 0;
    mark(2);
    if (tr) g_pnp_trace[15] = planar;
    if (planar) {
        // Coplanar object points: cvFindExtrinsicCameraParams2's planar initialisation (calibration.cpp; oracle:
        // cv3_find_extrinsic_guess).  The points are rotated into their principal plane (R_transform = V^T of the SVD above,
        // made right-handed; T_transform = -R_transform Mc), cvFindHomography (method 0: DLT over all inliers + LM,
        // h_refit_refine_block -- the block findHomography's own tail runs on) maps the in-plane coordinates to the
        // normalised image points, and the pose is read off H.  The LDS of the row buffer is free at this point.
        __shared__ double s_pl[12 + 9];  // R_transform | T_transform | zero matrix (the model of a rejected point set)
        HRefineShared& hs = *reinterpret_cast<HRefineShared*>(s_buf);
        HRefineWork& hw = *reinterpret_cast<HRefineWork*>(s_buf + 5200);
        static_assert(sizeof(HRefineShared) <= 5200 * sizeof(double) && 5200 * sizeof(double) + sizeof(HRefineWork) <= sizeof(s_buf),
                      "the homography block does not fit the row buffer");
        if (t == 0) {
            const double* V = s_ws + 12;  // V^T: rows are the right singular vectors
            double Rt[9];
            for (int i = 0; i < 9; i++) Rt[i] = V[i];
            if (V[2] * V[2] + V[5] * V[5] < 1e-10) {
                for (int i = 0; i < 9; i++) Rt[i] = 0.;
                Rt[0] = Rt[4] = Rt[8] = 1.;
            }
            if (sm::det3(Rt) < 0)
                for (int i = 0; i < 9; i++) Rt[i] *= -1.;
            for (int i = 0; i < 9; i++) s_pl[i] = Rt[i];
            for (int i = 0; i < 3; i++) s_pl[9 + i] = (Rt[i * 3] * s_Mc[0] + Rt[i * 3 + 1] * s_Mc[1] + Rt[i * 3 + 2] * s_Mc[2]) * -1.;
            for (int i = 0; i < 9; i++) s_pl[12 + i] = 0.;
        }
        __syncthreads();
        auto plane_pts = [&](int c0) -> const float* {  // (in-plane x, y | normalised image point) of inliers [c0, c0 + 256)
            const int cnt = load_chunk(c0);
            __syncthreads();
            if (t < cnt) {
                const double X = s_pts[t * 5], Y = s_pts[t * 5 + 1], Z = s_pts[t * 5 + 2];
                const double Mx = s_pl[0] * X + s_pl[1] * Y + s_pl[2] * Z + s_pl[9];
                const double My = s_pl[3] * X + s_pl[4] * Y + s_pl[5] * Z + s_pl[10];
                hs.pts[t * 4 + 0] = (float)Mx;
                hs.pts[t * 4 + 1] = (float)My;
                hs.pts[t * 4 + 2] = (float)(((double)s_pts[t * 5 + 3] - cx) * ifx);
                hs.pts[t * 4 + 3] = (float)(((double)s_pts[t * 5 + 4] - cy) * ify);
            }
            __syncthreads();
            return hs.pts;
        };
        h_refit_refine_block(hs, hw, np, plane_pts, s_pl + 12, /*refine_on_reject=*/false);
        __syncthreads();
        if (t == 0) {
            double h[9];
            for (int i = 0; i < 8; i++) h[i] = hw.x[i];
            h[8] = hw.h[8];
            bool finite = true;
            for (int i = 0; i < 9; i++) finite = finite && isfinite(h[i]);
            double Rm[9], tt[3] = {0., 0., 0.};
            if (finite) {
                const double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
                const double h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
                const double s1 = 1. / (h1n > DBL_EPSILON ? h1n : DBL_EPSILON), s2 = 1. / (h2n > DBL_EPSILON ? h2n : DBL_EPSILON);
                const double s3 = 2. / (h1n + h2n > DBL_EPSILON ? h1n + h2n : DBL_EPSILON);
                for (int i = 0; i < 3; i++) {
                    h[i * 3] *= s1;
                    h[i * 3 + 1] *= s2;
                    tt[i] = h[i * 3 + 2] * s3;
                }
                h[2] = h[3] * h[7] - h[6] * h[4];
                h[5] = h[6] * h[1] - h[0] * h[7];
                h[8] = h[0] * h[4] - h[3] * h[1];
                double r3[3], Hr[9];
                sm::rodrigues_m2v(h, r3, s_ws);
                sm::rodrigues_v2m(r3, Hr, nullptr);
                const double* Rt = s_pl;
                const double* Tt = s_pl + 9;
                for (int i = 0; i < 3; i++) tt[i] = (Hr[i * 3] * Tt[0] + Hr[i * 3 + 1] * Tt[1] + Hr[i * 3 + 2] * Tt[2]) + tt[i];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) Rm[i * 3 + j] = Hr[i * 3] * Rt[j] + Hr[i * 3 + 1] * Rt[3 + j] + Hr[i * 3 + 2] * Rt[6 + j];
            } else {
                for (int i = 0; i < 9; i++) Rm[i] = 0.;
                Rm[0] = Rm[4] = Rm[8] = 1.;
            }
            sm::rodrigues_m2v(Rm, s_param, s_ws);
            for (int i = 0; i < 3; i++) s_param[3 + i] = tt[i];
        }
    }
    __syncthreads();
    // ---- DLT: LL = L^T L, L = 2 rows per point; lane t < 78 owns upper-triangle entry (la, lb)
    int la = 0, lb = 0;
    if (t < 78) {
        int rem = t;
        while (rem >= 12 - la) {
            rem -= 12 - la;
            la++;
        }
        lb = la + rem;
    }
    acc = 0;
    for (int c0 = 0; c0 < np && !planar; c0 += PR_CHUNK) {
        const int cnt = load_chunk(c0);
        __syncthreads();
        if (t < cnt) {
            const double X = s_pts[t * 5], Y = s_pts[t * 5 + 1], Z = s_pts[t * 5 + 2];
            const double x = -(((double)s_pts[t * 5 + 3] - cx) * ifx), y = -(((double)s_pts[t * 5 + 4] - cy) * ify);
            double* o = s_buf + t * PR_ROW;
            o[0] = X; o[1] = Y; o[2] = Z; o[3] = 1.; o[4] = 0.; o[5] = 0.; o[6] = 0.; o[7] = 0.;
            o[8] = x * X; o[9] = x * Y; o[10] = x * Z; o[11] = x;
            o[12] = 0.; o[13] = 0.; o[14] = 0.; o[15] = 0.; o[16] = X; o[17] = Y; o[18] = Z; o[19] = 1.;
            o[20] = y * X; o[21] = y * Y; o[22] = y * Z; o[23] = y;
        }
        __syncthreads();
        if (t < 78) acc = pr_acc_two(s_buf, cnt, la, lb, 12 + la, 12 + lb, acc);
        __syncthreads();
    }
    if (t < 78 && !planar) {
        s_LL[la * 12 + lb] = acc;
        s_LL[lb * 12 + la] = acc;
    }
    __syncthreads();
    mark(3);
    if (t == 0 && !planar) sm::pnp_dlt_finish(s_LL, s_param, s_ws);
    __syncthreads();
    mark(4);
    // ---- CvLevMarq(6 parameters, 2 np residuals, 20 iterations, FLT_EPSILON)
    int ja = 0, jb = 0;  // lane t < 21: JtJ entry (ja, jb)
    if (t < 21) {
        int rem = t;
        while (rem >= 6 - ja) {
            rem -= 6 - ja;
            ja++;
        }
        jb = ja + rem;
    }
    // residuals (and Jacobian rows) at s_param: row buffer [0..5] du/dp, [6..11] dv/dp, [12] eu, [13] ev
    auto pass = [&](bool with_j) {
        if (t == 0) sm::rodrigues_v2m(s_param, s_R, with_j ? s_dRdr : nullptr);
        __syncthreads();
        double a = 0;
        for (int c0 = 0; c0 < np; c0 += PR_CHUNK) {
            const int cnt = load_chunk(c0);
            __syncthreads();
            if (t < cnt) {
                double u, v, jr[6], jt[6];
                sm::project_point(s_R, s_param + 3, B.K4, s_pts[t * 5], s_pts[t * 5 + 1], s_pts[t * 5 + 2], &u, &v, s_dRdr,
                                  with_j ? jr : nullptr, with_j ? jt : nullptr);
                double* o = s_buf + t * PR_ROW;
                o[12] = u - (double)s_pts[t * 5 + 3];
                o[13] = v - (double)s_pts[t * 5 + 4];
                if (with_j)
                    for (int j = 0; j < 3; j++) {
                        o[j] = jr[j];
                        o[3 + j] = jt[j];
                        o[6 + j] = jr[3 + j];
                        o[9 + j] = jt[3 + j];
                    }
            }
            __syncthreads();
            if (with_j && t < 21) {
                a = pr_acc_two(s_buf, cnt, ja, jb, 6 + ja, 6 + jb, a);
            } else if (with_j && t >= 64 && t < 70) {
                a = pr_acc_two(s_buf, cnt, t - 64, 12, 6 + t - 64, 13, a);
            } else if (t == 128) {  // |err|^2, four residuals (two points) per step as cv::norm does
                int k = 0;
                for (; k + 1 < cnt; k += 2) {
                    const double v0 = s_buf[k * PR_ROW + 12], v1 = s_buf[k * PR_ROW + 13];
                    const double v2 = s_buf[(k + 1) * PR_ROW + 12], v3 = s_buf[(k + 1) * PR_ROW + 13];
                    a += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
                }
                for (; k < cnt; k++) {  // only the very last point of an odd-sized inlier set
                    const double v0 = s_buf[k * PR_ROW + 12], v1 = s_buf[k * PR_ROW + 13];
                    a += v0 * v0;
                    a += v1 * v1;
                }
            }
            __syncthreads();
        }
        if (with_j && t < 21) {
            s_JtJ[ja * 6 + jb] = a;
            s_JtJ[jb * 6 + ja] = a;
        } else if (with_j && t >= 64 && t < 70) {
            s_JtErr[t - 64] = a;
        } else if (t == 128) {
            s_errnorm2 = a;
        }
        __syncthreads();
    };
    double prevErrNorm = DBL_MAX;
    int lambdaLg10 = -3, iters = 0;
    bool calc_j = true;
    for (;;) {
        long long tr0 = tr ? wall_clock64() : 0;
        pass(calc_j);
        if (tr) {
            g_pnp_trace[calc_j ? 8 : 9] += wall_clock64() - tr0;
            g_pnp_trace[calc_j ? 11 : 12] += 1;
            tr0 = wall_clock64();
        }
        if (calc_j) {
            if (t == 0) {
                for (int i = 0; i < 6; i++) s_prev[i] = s_param[i];
                sm::pnp_lm_step(s_JtJ, s_JtErr, lambdaLg10, s_prev, s_param, s_ws);
            }
            if (tr) {
                g_pnp_trace[10] += wall_clock64() - tr0;
                g_pnp_trace[13] += 1;
            }
            if (iters == 0) prevErrNorm = sqrt(s_errnorm2);
            calc_j = false;
            __syncthreads();
            continue;
        }
        const double errNorm = sqrt(s_errnorm2);
        if (errNorm > prevErrNorm) {
            if (++lambdaLg10 <= 16) {
                if (t == 0) sm::pnp_lm_step(s_JtJ, s_JtErr, lambdaLg10, s_prev, s_param, s_ws);
                if (tr) {
                    g_pnp_trace[10] += wall_clock64() - tr0;
                    g_pnp_trace[13] += 1;
                }
                __syncthreads();
                continue;
            }
        }
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        if (++iters >= 20 || sm::pnp_rel_change6(s_param, s_prev) < FLT_EPSILON) break;
        prevErrNorm = errNorm;
        calc_j = true;
    }
    mark(5);
    if (t == 0) {
        out->flag = 1;
        out->n_inliers = np;
        out->status = 1;
        out->lm_iters = iters;
        for (int i = 0; i < 3; i++) {
            out->rvec[i] = s_param[i];
            out->tvec[i] = s_param[3 + i];
        }
    }
}

// best repeat by inlier count (first one wins ties), cv2.Rodrigues of its rotation vector
__global__ void k_pnp_select(const PBatch B, PnpResult* __restrict__ res) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int best = -1, best_inlier = 0, status = 0;
    for (int r = 0; r < B.nrep; ++r) {
        const PnpRepOut* o = B.r[r].out;
        if (o->status < 0) status = o->status;
        if (o->flag && o->n_inliers > best_inlier) {
            best = r;
            best_inlier = o->n_inliers;
        }
    }
    res->found = best >= 0 ? 1 : 0;
    res->best_inliers = best_inlier;
    res->n_filtered = *B.n_ptr;
    res->status = status;
    for (int i = 0; i < 3; i++) res->rvec[i] = res->tvec[i] = 0;
    for (int i = 0; i < 9; i++) res->R[i] = (i % 4 == 0) ? 1. : 0.;
    if (best >= 0) {
        const PnpRepOut* o = B.r[best].out;
        for (int i = 0; i < 3; i++) {
            res->rvec[i] = o->rvec[i];
            res->tvec[i] = o->tvec[i];
        }
        sm::rodrigues_v2m(o->rvec, res->R, nullptr);
    }
}

// ================================================================================================
int PnpBuffers::ensure(int n, int iters) {
    if (n <= cap && iters <= iters_cap) return DFVO_OK;
    release();
    cap = n > cap ? n : cap;
    iters_cap = iters > iters_cap ? iters : iters_cap;
    const size_t c = (size_t)cap, it = (size_t)iters_cap;
    DFVO_HIP_CHECK(hipMalloc((void**)&info, sizeof(int) * 4));
    DFVO_HIP_CHECK(hipMalloc((void**)&fk1, sizeof(double) * 2 * c));
    DFVO_HIP_CHECK(hipMalloc((void**)&fk2, sizeof(double) * 2 * c));
    DFVO_HIP_CHECK(hipMalloc((void**)&xyz, sizeof(double) * 3 * c));
    DFVO_HIP_CHECK(hipMalloc((void**)&perm, sizeof(int) * MAX_REP * (c + 8)));
    DFVO_HIP_CHECK(hipMalloc((void**)&obj, sizeof(float) * MAX_REP * 3 * c));
    DFVO_HIP_CHECK(hipMalloc((void**)&img, sizeof(float) * MAX_REP * 2 * c));
    DFVO_HIP_CHECK(hipMalloc((void**)&state, sizeof(RansacState) * MAX_REP));
    DFVO_HIP_CHECK(hipMalloc((void**)&idx, sizeof(int) * 5 * it));
    DFVO_HIP_CHECK(hipMalloc((void**)&models, sizeof(double) * MAX_REP * 6 * it));
    DFVO_HIP_CHECK(hipMalloc((void**)&nmodels, sizeof(int) * MAX_REP * it));
    DFVO_HIP_CHECK(hipMalloc((void**)&counts, sizeof(int) * MAX_REP * it));
    DFVO_HIP_CHECK(hipMalloc((void**)&mask, MAX_REP * c));
    DFVO_HIP_CHECK(hipMalloc((void**)&keep, c));
    DFVO_HIP_CHECK(hipMalloc((void**)&pts5, sizeof(float) * MAX_REP * 5 * c));
    DFVO_HIP_CHECK(hipMalloc((void**)&rep_out, sizeof(PnpRepOut) * MAX_REP));
    DFVO_HIP_CHECK(hipMalloc((void**)&result, sizeof(PnpResult)));
    return DFVO_OK;
}

void PnpBuffers::release() {
    void* ptrs[] = {info, fk1, fk2, xyz, perm, obj, img, state, idx, models, nmodels, counts, mask, keep, pts5, rep_out, result};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    info = perm = idx = nmodels = counts = nullptr;
    fk1 = fk2 = xyz = models = nullptr;
    obj = img = pts5 = nullptr;
    state = nullptr;
    mask = keep = nullptr;
    rep_out = nullptr;
    result = nullptr;
    cap = iters_cap = 0;
}

// kp1 / kp2: device [n][2] doubles (n = *d_n when d_n != nullptr, else n_host); depth: device f64 [H][W].
// Results: pb.result (PnpResult), pb.fk1 / pb.fk2 (filtered keypoints, pb.info[0] of them)
int enqueue_compute_pose_3d2d(PnpBuffers& pb, uint32_t* mt_state, const double* d_kp1, const double* d_kp2,
                              const int* d_n, int n_host, const double* d_depth, int H, int W, const PnpConfig& cfg,
                              hipStream_t s, bool depth_per_kp) {
    DFVO_ARG_CHECK(n_host >= 0 && cfg.repeat >= 1 && cfg.repeat <= MAX_REP && cfg.iters >= 1, "compute_pose_3d2d: bad sizes");
    int rc = pb.ensure(n_host > 8 ? n_host : 8, cfg.iters);
    if (rc != DFVO_OK) return rc;
    const int cap = pb.cap;
    hipLaunchKernelGGL(k_pnp_filter, dim3(1), dim3(256), 0, s, d_kp1, d_kp2, d_n, n_host, d_depth, H, W, cfg, pb.fk1,
                       pb.fk2, pb.xyz, pb.keep, pb.info, depth_per_kp ? 1 : 0);
    // the shuffles are drawn for every repeat, whatever the point count (pnp_tracker.py:90-92)
    rc = enqueue_mt_shuffle(mt_state, pb.info, n_host, cfg.repeat, cap + 8, pb.perm, s);
    if (rc != DFVO_OK) return rc;
    const int nb = cdiv(n_host > 0 ? n_host : 1, 256);
    const unsigned R = (unsigned)cfg.repeat;
    hipLaunchKernelGGL(k_pnp_permute, dim3(nb, R), dim3(256), 0, s, pb.info, pb.perm, cap + 8, pb.xyz, pb.fk2, pb.obj,
                       pb.img, cap);
    PBatch B;
    memset(&B, 0, sizeof(B));
    B.nrep = cfg.repeat;
    B.n_ptr = pb.info;
    B.idx = pb.idx;
    B.K4[0] = cfg.fx;
    B.K4[1] = cfg.fy;
    B.K4[2] = cfg.cx;
    B.K4[3] = cfg.cy;
    for (int r = 0; r < cfg.repeat; ++r) {
        PRep& P = B.r[r];
        P.state = pb.state + r;
        P.obj = pb.obj + (size_t)r * cap * 3;
        P.img = pb.img + (size_t)r * cap * 2;
        P.models = pb.models + (size_t)r * pb.iters_cap * 6;
        P.nmodels = pb.nmodels + (size_t)r * pb.iters_cap;
        P.counts = pb.counts + (size_t)r * pb.iters_cap;
        P.mask = pb.mask + (size_t)r * cap;
        P.pts5 = pb.pts5 + (size_t)r * cap * 5;
        P.out = pb.rep_out + r;
    }
    const float thr2 = (float)(cfg.reproj_thre * cfg.reproj_thre);
    hipLaunchKernelGGL(k_pnp_init, dim3(R), dim3(1), 0, s, B, cfg.iters);
    int cb[4];
    chunk_bounds(cfg.iters, cb);
    for (int c = 0; c < 3; ++c) {
        const int it0 = cb[c], it1 = cb[c + 1];
        if (it1 <= it0) continue;
        const int nh = it1 - it0;
        hipLaunchKernelGGL(k_pnp_subsets, dim3(1), dim3(1), 0, s, B, it0, it1);
        hipLaunchKernelGGL(k_pnp_solve, dim3(cdiv(nh, PNP_SOLVE_LANES), R), dim3(PNP_SOLVE_LANES), 0, s, B, it0, it1);
        hipLaunchKernelGGL(k_pnp_score, dim3(cdiv(nh, 4), R), dim3(256), 0, s, B, it0, it1, thr2);
        hipLaunchKernelGGL(k_pnp_replay, dim3(R), dim3(1), 0, s, B, it0, it1, 0.99);
    }
    hipLaunchKernelGGL(k_pnp_mask, dim3(nb, R), dim3(256), 0, s, B, thr2);
    static const bool trace_on = getenv("DFVO_PNP_TRACE") != nullptr;
    hipLaunchKernelGGL(k_pnp_refine, dim3(R), dim3(256), 0, s, B, trace_on ? 1 : 0);
    hipLaunchKernelGGL(k_pnp_select, dim3(1), dim3(1), 0, s, B, pb.result);
    DFVO_HIP_CHECK(hipGetLastError());
    if (trace_on) {
        long long tr[16];
        DFVO_HIP_CHECK(hipStreamSynchronize(s));
        DFVO_HIP_CHECK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_pnp_trace), sizeof(tr)));
        auto us = [&](int a, int b) { return tr[a] && tr[b] ? (tr[b] - tr[a]) * 0.01 : -1.0; };
        fprintf(stderr,
                "k_pnp_refine trace (repeat 0): inliers %lld planar %lld | compaction %.1f us | centroid + covariance %.1f | DLT sums %.1f | "
                "DLT solve %.1f | LM: %lld passes with J %.1f us, %lld passes without %.1f, %lld steps %.1f | total %.1f us\n",
                tr[14], tr[15], us(0, 1), us(1, 2), us(2, 3), us(3, 4), tr[11], tr[8] * 0.01, tr[12], tr[9] * 0.01, tr[13], tr[10] * 0.01,
                us(0, 5));
    }
    return DFVO_OK;
}

}  // namespace dfvo
