// Device code shared by the homography refinement of findHomography (k_h_refine, solver_ransac.hip) and the planar
// initialisation of cvFindExtrinsicCameraParams2 (k_pnp_refine, solver_pnp.hip): the cooperative Jacobi eigen solver and
// the "refit on the inliers + Levenberg-Marquardt" block, one 256-thread workgroup, one lane per sequentially-summed
// accumulator (same rounding as the CPU loops of oracle/cv3_calib3d.c: h_run_kernel / h_refine_lm).
#pragma once
#include <cfloat>

#include "solver_math.h"

namespace dfvo {

// ------------------------------------------------------------------------------------------------
// Jacobi eigen decomposition run by ONE WAVEFRONT on a matrix in LDS (same arithmetic, element for element, as
// sm::jacobi_eigen_ws: the rotation sequence is data dependent and stays sequential, but inside a rotation the
// pivot search is a (value, scan order) max-reduction over <= 16 lanes, the element pairs of the two rows/columns
// are rotated one per lane, and the four pivot-table rescans run on four lanes).  ~10x the single-lane rate.
// ------------------------------------------------------------------------------------------------
#define WAVE_LDS_SYNC()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
    } while (0)

// 64-bit / 32-bit values through one DPP control (VALU moves inside a row of sixteen lanes, no LDS round trip)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
// one exchange of the pivot search: (key, scan order) maximum with the lower scan order winning ties; p = the signed element
// and code = order << 16 | k << 8 | l travel with the key
template <int CTRL>
__device__ __forceinline__ void pivot_exchange(double& key, double& p, int& code) {
    const double ok = dpp_f64<CTRL>(key), op = dpp_f64<CTRL>(p);
    const int oc = dpp_i32<CTRL>(code);
    if (ok > key || (ok == key && oc < code)) {
        key = ok;
        p = op;
        code = oc;
    }
}
__device__ __forceinline__ double uniform_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double lane_f64(double v, int src_lane) {  // src_lane wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
                            __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

// Round 5 form (profiles/r5j_h_refine_trace.txt: the eigen problems were 1.0 of the 1.4 ms of k_h_refine at 2.1 - 2.6 us per
// rotation).  What a rotation no longer does: the pivot tables and the diagonal live in REGISTERS of their owner lanes (lane r <
// N - 1: row r's table entry, lane N - 2 + c: column c's, lane i < N: W[i]) instead of LDS; the pivot maximum is four DPP
// exchanges (quad xor 1, xor 2, half mirror, row mirror -- an all-reduce over the sixteen candidate lanes) that carry the signed
// element and its (k, l) along, instead of twelve ds_bpermute round trips plus three more for the winner's coordinates and an LDS
// read of the element; W[k], W[l] come through v_readlane.  With nothing but A and V left in LDS, and every lane of the
// rotation touching its own elements only, ONE wave-level LDS synchronisation per rotation remains (after the rotation, before
// the owner lanes rescan rows / columns k and l) instead of three.  The rescans and the rotation are single code paths with
// per-lane addresses (they were up to four / three divergent branches).  Same operations on the same operands in the same
// order as sm::jacobi_eigen_ws -- the pivot search now also has the sequential scan's NaN behaviour (a NaN first candidate
// sticks, any other NaN is never selected).
template <int N>
__device__ int jacobi_eigen_coop(double* A, double* W, double* V, int lane) {
    static_assert(2 * N - 2 <= 16, "the pivot candidates must fit one DPP row");
    const double eps = DBL_EPSILON;
    // ---- ownership: lane r < N - 1 owns row r's pivot-table entry, lane N - 2 + c (c = 1 .. N - 1) column c's
    const bool own_row = lane < N - 1, own_col = lane >= N - 1 && lane < 2 * N - 2;
    const int oc = lane - (N - 2);
    // the owner's scan as (first address, stride, count, index of the first element): row r: A[r][r + 1 ..], column c: A[0 .. c - 1][c]
    const int sc_base = own_row ? N * lane + lane + 1 : oc, sc_stride = own_row ? 1 : N, sc_cnt = own_row ? N - 1 - lane : oc,
              sc_m0 = own_row ? lane + 1 : 0;
    auto rescan = [&]() {  // first maximum of |A| over the owner's row / column part
        int m = sc_m0;
        double mv = fabs(A[sc_base]);
        for (int j = 1; j < sc_cnt; j++) {
            const double val = fabs(A[sc_base + j * sc_stride]);
            if (mv < val) mv = val, m = sc_m0 + j;
        }
        return m;
    };
    for (int idx = lane; idx < N * N; idx += 64) V[idx] = (idx / N == idx % N) ? 1. : 0.;
    double myW = lane < N ? A[(N + 1) * lane] : 0.;
    int myind = 0;
    if (own_row || own_col) myind = rescan();
    WAVE_LDS_SYNC();
    const int maxIters = N * N * 30;
    int iters = 0;
    for (; iters < maxIters; iters++) {
        // pivot candidates in the sequential scan order: rows 0..N-2, then columns 1..N-1
        double key = -1., p = 0.;
        int code = 0x7fffffff;
        if (own_row || own_col) {
            const int ck = own_row ? lane : myind, cl = own_row ? myind : oc;
            p = A[N * ck + cl];
            key = fabs(p);
            if (key != key) key = lane == 0 ? __builtin_inf() : -1.;
            code = lane << 16 | ck << 8 | cl;
        }
        pivot_exchange<0xB1>(key, p, code);   // quad_perm [1 0 3 2]
        pivot_exchange<0x4E>(key, p, code);   // quad_perm [2 3 0 1]
        pivot_exchange<0x141>(key, p, code);  // row_half_mirror
        pivot_exchange<0x140>(key, p, code);  // row_mirror
        code = __builtin_amdgcn_readfirstlane(code);
        p = uniform_f64(p);
        const int k = (code >> 8) & 0xff, l = code & 0xff;
        if (fabs(p) <= eps) break;
        const double y = (lane_f64(myW, l) - lane_f64(myW, k)) * 0.5;
        double t = fabs(y) + sm::hypot_p(p, y);
        double sn = sm::hypot_p(p, t);
        const double c = t / sn;
        sn = p / sn;
        t = (p / t) * p;
        if (y < 0) sn = -sn, t = -t;
        if (lane == k) myW -= t;
        if (lane == l) myW += t;
        if (lane == 0) A[N * k + l] = 0;
        if (lane < N) {
            const int i = lane;
            if (i != k && i != l) {  // the stored (upper-triangle) copies of A(i, k) and A(i, l)
                const int ak = i < k ? N * i + k : N * k + i, al = i < l ? N * i + l : N * l + i;
                const double a0 = A[ak], b0 = A[al];
                A[ak] = a0 * c - b0 * sn;
                A[al] = a0 * sn + b0 * c;
            }
            const double a0 = V[N * k + i], b0 = V[N * l + i];
            V[N * k + i] = a0 * c - b0 * sn;
            V[N * l + i] = a0 * sn + b0 * c;
        }
        WAVE_LDS_SYNC();
        if ((own_row && (lane == k || lane == l)) || (own_col && (oc == k || oc == l))) myind = rescan();
    }
    if (lane < N) W[lane] = myW;
    WAVE_LDS_SYNC();
    if (lane == 0) {  // descending selection sort of the eigenvalues with their vectors
        for (int k = 0; k < N - 1; k++) {
            int m = k;
            for (int i = k + 1; i < N; i++)
                if (W[m] < W[i]) m = i;
            if (k != m) {
                double t = W[m];
                W[m] = W[k];
                W[k] = t;
                for (int i = 0; i < N; i++) {
                    t = V[N * m + i];
                    V[N * m + i] = V[N * k + i];
                    V[N * k + i] = t;
                }
            }
        }
    }
    WAVE_LDS_SYNC();
    return iters;  // rotations applied (diagnostics only)
}

// ------------------------------------------------------------------------------------------------
// post-RANSAC refinement on the inliers (findHomography tail): one 256-thread block.
//   * ordered (ballot) compaction of the inlier indices,
//   * least-squares refit: every LtL entry / centroid / scale is a sequential sum over the inliers in
//     index order, so ONE LANE OWNS ONE ACCUMULATOR (45 + 4 lanes) -> same rounding as the CPU loop;
//     the per-point terms are produced 256 points at a time by all threads into LDS,
//   * Levenberg-Marquardt (<= 10 iterations): residuals and the two Jacobian rows of 256 points per chunk
//     in parallel into LDS, then J^T J (36 unique entries), J^T r (8), |r|^2 and |r|_inf again one lane
//     per accumulator over the chunk; the 8x8 eigen-solve on lane 0.
// ------------------------------------------------------------------------------------------------
struct HRefineShared {
    double buf[256 * 18];  // per-point Lx|Ly (refit) or J rows (16) + r (2) (LM)
    float pts[256 * 4];    // Mx My mx my of the chunk
    int wave_cnt[4];
    int base;
};

__device__ __forceinline__ void h_chunk_points(HRefineShared& sh, const float* src, const float* dst, const int* cidx,
                                               int c0, int np) {
    const int t = threadIdx.x;
    if (c0 + t < np) {
        const int p = cidx[c0 + t];
        sh.pts[t * 4 + 0] = src[p * 2];
        sh.pts[t * 4 + 1] = src[p * 2 + 1];
        sh.pts[t * 4 + 2] = dst[p * 2];
        sh.pts[t * 4 + 3] = dst[p * 2 + 1];
    }
}

// residuals (and Jacobian rows) of the chunk at parameters h into sh.buf: [t*18 + 0..7] row a, [8..15] row b,
// [16],[17] residuals
__device__ __forceinline__ void h_chunk_lm(HRefineShared& sh, const float* pts, const double* h, int cnt, bool with_j) {
    const int t = threadIdx.x;
    if (t >= cnt) return;
    const double Mx = pts[t * 4], My = pts[t * 4 + 1];
    double ww = h[6] * Mx + h[7] * My + 1.;
    ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
    const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
    const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
    double* o = sh.buf + t * 18;
    o[16] = xi - pts[t * 4 + 2];
    o[17] = yi - pts[t * 4 + 3];
    if (with_j) {
        o[0] = Mx * ww;
        o[1] = My * ww;
        o[2] = ww;
        o[3] = o[4] = o[5] = 0.;
        o[6] = -Mx * ww * xi;
        o[7] = -My * ww * xi;
        o[8] = o[9] = o[10] = 0.;
        o[11] = Mx * ww;
        o[12] = My * ww;
        o[13] = ww;
        o[14] = -Mx * ww * yi;
        o[15] = -My * ww * yi;
    }
}

// Sequential (reference-order) sums over the points of a chunk.  The products are independent, only the adds
// form the chain; batches of 8 points are loaded and multiplied first so the LDS latency is paid once per batch.
constexpr int H_UNROLL = 8;
// a += o[i0]*o[i1]; a += o[i2]*o[i3]   per point
__device__ __forceinline__ double seq_acc_two(const double* buf, int cnt, int i0, int i1, int i2, int i3, double a) {
    int k = 0;
    for (; k + H_UNROLL <= cnt; k += H_UNROLL) {
        double p0[H_UNROLL], p1[H_UNROLL];
#pragma unroll
        for (int u = 0; u < H_UNROLL; ++u) {
            const double* o = buf + (k + u) * 18;
            p0[u] = o[i0] * o[i1];
            p1[u] = o[i2] * o[i3];
        }
#pragma unroll
        for (int u = 0; u < H_UNROLL; ++u) {
            a += p0[u];
            a += p1[u];
        }
    }
    for (; k < cnt; k++) {
        const double* o = buf + k * 18;
        a += o[i0] * o[i1];
        a += o[i2] * o[i3];
    }
    return a;
}
// a += o[i0]*o[i1] + o[i2]*o[i3]   per point
__device__ __forceinline__ double seq_acc_pair(const double* buf, int cnt, int i0, int i1, int i2, int i3, double a) {
    int k = 0;
    for (; k + H_UNROLL <= cnt; k += H_UNROLL) {
        double p[H_UNROLL];
#pragma unroll
        for (int u = 0; u < H_UNROLL; ++u) {
            const double* o = buf + (k + u) * 18;
            p[u] = o[i0] * o[i1] + o[i2] * o[i3];
        }
#pragma unroll
        for (int u = 0; u < H_UNROLL; ++u) a += p[u];
    }
    for (; k < cnt; k++) {
        const double* o = buf + k * 18;
        a += o[i0] * o[i1] + o[i2] * o[i3];
    }
    return a;
}
// |r|^2 in groups of two points (four residual rows), the order the oracle's norm loop uses
__device__ __forceinline__ double seq_acc_sq(const double* buf, int cnt, double a) {
    int k = 0;
    for (; k + H_UNROLL <= cnt; k += H_UNROLL) {
        double p[H_UNROLL / 2];
#pragma unroll
        for (int u = 0; u < H_UNROLL / 2; ++u) {
            const double v0 = buf[(k + 2 * u) * 18 + 16], v1 = buf[(k + 2 * u) * 18 + 17];
            const double v2 = buf[(k + 2 * u + 1) * 18 + 16], v3 = buf[(k + 2 * u + 1) * 18 + 17];
            p[u] = v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
        }
#pragma unroll
        for (int u = 0; u < H_UNROLL / 2; ++u) a += p[u];
    }
    for (; k + 1 < cnt; k += 2) {
        const double v0 = buf[k * 18 + 16], v1 = buf[k * 18 + 17];
        const double v2 = buf[(k + 1) * 18 + 16], v3 = buf[(k + 1) * 18 + 17];
        a += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    for (; k < cnt; k++) {  // only at the very end of the inlier list (chunks hold an even count otherwise)
        const double v0 = buf[k * 18 + 16], v1 = buf[k * 18 + 17];
        a += v0 * v0;
        a += v1 * v1;
    }
    return a;
}

__device__ double seq_dot8(const double* a, const double* b) {
    double r = 0;
    r += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    r += a[4] * b[4] + a[5] * b[5] + a[6] * b[6] + a[7] * b[7];
    return r;
}

// everything the refit + LM block keeps in LDS besides HRefineShared
struct HRefineWork {
    double h[9], x[8], xd[8], d[8], v[8], A[64], D[8], norm[8], LtL[81];
    double S, Sd, rinf, lambda, lc;
    double Ap[64], eig[2 * 81 + 2 * 9];  // lane 0's dense 8x8 / 9x9 work (LDS, not scratch)
    double At[64], vt[8], rinft;         // J^T J, J^T r, |r|_inf at the trial point x - d (see full_pass)
    int flag;
};

// cv::findHomography's tail on `np` point pairs: the normalised DLT over all of them (HomographyEstimatorCallback::
// runKernel), then the Levenberg-Marquardt refinement (<= 10 iterations).  chunk_pts(c0) returns the points
// [c0, c0 + 256) as Mx My mx my floats (and may synchronise the workgroup).  When the DLT rejects the point set
// (degenerate scales) the model is H_fallback.  All 256 threads must call; the result is in w.x[0..7] (h[8] = w.h[8]).
// refine_on_reject: run the LM from H_fallback even when the DLT rejected the points (findHomography's RANSAC tail calls
// runKernel and the LM unconditionally); false = leave H_fallback as the answer (cv::findHomography method 0 returns
// an empty matrix there, which cvFindHomography turns into zeros).
// trace (optional, diagnostics: DFVO_HREFINE_TRACE): 24 long long in global memory, written by thread 0 -- wall_clock64 ticks
// (100 MHz) at [1] entry [2] centroids + scales done [3] LtL sums done [4] 9 x 9 eigen + de-normalisation done [5] first
// Jacobian pass done [6] end; sums over the LM loop [8] 8 x 8 eigen [9] back-substitution [10] trial passes [11] lane-0 step
// logic; [12] LM iterations [13] rotations of the 9 x 9 problem [14] rotations of all 8 x 8 problems [15] point count.
template <class ChunkPts>
__device__ __forceinline__ void h_refit_refine_block(HRefineShared& sh, HRefineWork& w, int np, ChunkPts chunk_pts,
                                                     const double* H_fallback, bool refine_on_reject = true,
                                                     long long* trace = nullptr) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    auto mark = [&](int i) {
        if (trace && t == 0) trace[i] = wall_clock64();
    };
    auto add_since = [&](int i, long long t0) {
        if (trace && t == 0) trace[i] += wall_clock64() - t0;
    };
    if (trace && t == 0) {
        for (int i = 8; i < 15; i++) trace[i] = 0;
        trace[15] = np;
    }
    mark(1);
    double* const s_h = w.h;
    double* const s_x = w.x;
    double* const s_xd = w.xd;
    double* const s_d = w.d;
    double* const s_v = w.v;
    double* const s_A = w.A;
    double* const s_D = w.D;
    double* const s_norm = w.norm;
    double* const s_LtL = w.LtL;
    double* const s_Ap = w.Ap;
    double* const s_eig = w.eig;
    double& s_S = w.S;
    double& s_Sd = w.Sd;
    double& s_rinf = w.rinf;
    double& s_lambda = w.lambda;
    double& s_lc = w.lc;
    int& s_flag = w.flag;
    const double* const H_io = H_fallback;
    // ---- refit: centroids (4 sequential sums), then scales (4 sequential sums)
    double acc = 0;
    for (int c0 = 0; c0 < np; c0 += 256) {
        const int cnt = np - c0 < 256 ? np - c0 : 256;
        const float* cp = chunk_pts(c0);
        if (t < 4) {  // 0: cm.x 1: cm.y 2: cM.x 3: cM.y
            const int comp = t < 2 ? 2 + t : t - 2;
            int k = 0;
            for (; k + 8 <= cnt; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = cp[(k + u) * 4 + comp];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; k < cnt; k++) acc += cp[k * 4 + comp];
        }
        __syncthreads();
    }
    if (t < 4) s_norm[t] = acc / np;
    __syncthreads();
    acc = 0;
    for (int c0 = 0; c0 < np; c0 += 256) {
        const int cnt = np - c0 < 256 ? np - c0 : 256;
        const float* cp = chunk_pts(c0);
        if (t < 4) {
            const int comp = t < 2 ? 2 + t : t - 2;
            const double c = s_norm[t];
            int k = 0;
            for (; k + 8 <= cnt; k += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = fabs(cp[(k + u) * 4 + comp] - c);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; k < cnt; k++) acc += fabs(cp[k * 4 + comp] - c);
        }
        __syncthreads();
    }
    if (t < 4) s_norm[4 + t] = acc;
    __syncthreads();
    mark(2);
    sm::HNorm hn;
    hn.cmx = s_norm[0];
    hn.cmy = s_norm[1];
    hn.cMx = s_norm[2];
    hn.cMy = s_norm[3];
    const bool degenerate = fabs(s_norm[4]) < DBL_EPSILON || fabs(s_norm[5]) < DBL_EPSILON ||
                            fabs(s_norm[6]) < DBL_EPSILON || fabs(s_norm[7]) < DBL_EPSILON;
    if (!degenerate) {
        hn.smx = np / s_norm[4];
        hn.smy = np / s_norm[5];
        hn.sMx = np / s_norm[6];
        hn.sMy = np / s_norm[7];
        int lj = 0, lk = 0;
        if (t < 45) {  // lane t owns upper-triangular entry (lj, lk)
            int rem = t;
            while (rem >= 9 - lj) {
                rem -= 9 - lj;
                lj++;
            }
            lk = lj + rem;
        }
        acc = 0;
        for (int c0 = 0; c0 < np; c0 += 256) {
            const int cnt = np - c0 < 256 ? np - c0 : 256;
            const float* cp = chunk_pts(c0);
            if (t < cnt) {
                const double x = (cp[t * 4 + 2] - hn.cmx) * hn.smx, y = (cp[t * 4 + 3] - hn.cmy) * hn.smy;
                const double X = (cp[t * 4 + 0] - hn.cMx) * hn.sMx, Y = (cp[t * 4 + 1] - hn.cMy) * hn.sMy;
                double* o = sh.buf + t * 18;
                o[0] = X; o[1] = Y; o[2] = 1; o[3] = 0; o[4] = 0; o[5] = 0; o[6] = -x * X; o[7] = -x * Y; o[8] = -x;
                o[9] = 0; o[10] = 0; o[11] = 0; o[12] = X; o[13] = Y; o[14] = 1; o[15] = -y * X; o[16] = -y * Y; o[17] = -y;
            }
            __syncthreads();
            if (t < 45) acc = seq_acc_pair(sh.buf, cnt, lj, lk, 9 + lj, 9 + lk, acc);
            __syncthreads();
        }
        if (t < 81) s_LtL[t] = 0;
        __syncthreads();
        if (t < 45) s_LtL[lj * 9 + lk] = acc;
        __syncthreads();
        if (t < 81) {  // symmetric completion, then the 9 x 9 eigen decomposition on wave 0
            const int j = t / 9, k = t % 9;
            if (k < j) s_LtL[t] = s_LtL[k * 9 + j];
        }
        __syncthreads();
        mark(3);
        if (wave == 0) {
            const int rot = jacobi_eigen_coop<9>(s_LtL, s_eig, s_eig + 9, lane);
            if (trace && t == 0) trace[13] = rot;
        }
        __syncthreads();
        if (t == 0) sm::homography_denormalise(hn, s_eig + 9 + 72, s_h);
    } else if (t == 0) {
        for (int i = 0; i < 9; i++) s_h[i] = H_io[i];
    }
    __syncthreads();
    if (degenerate && !refine_on_reject) {
        if (t < 8) s_x[t] = s_h[t];
        __syncthreads();
        return;
    }
    // ---- Levenberg-Marquardt on the 8 free parameters
    if (t < 8) s_x[t] = s_h[t];
    __syncthreads();
    mark(4);
    // accumulator roles: t < 36 -> JtJ(i,j); 64..71 -> Jtr(i); 128 -> |r|^2 (4-row groups); 192 -> |r|_inf
    int ai = 0, aj = 0;
    if (t < 36) {
        int rem = t;
        while (rem >= 8 - ai) {
            rem -= 8 - ai;
            ai++;
        }
        aj = ai + rem;
    }
    // residuals + Jacobian at h -> (A, v, S, rinf) of the CURRENT point (s_A, s_v, s_S, s_rinf) or, trial = true, of the TRIAL
    // point (w.At, w.vt, s_Sd, w.rinft).  The trial form replaces the reference loop's "residual norm at x - d first, then --
    // if the step is accepted -- everything again at the new x": the norm is the same ordered sum either way (seq_acc_sq over
    // the same residuals), and the Jacobian sums of an accepted step are exactly what the second evaluation would compute, so
    // one pass per iteration yields the same bits as the two; a rejected step discards the trial set (its Jacobian sums were
    // wasted: LM on RANSAC inliers rejects rarely).  Saves one ordered pass over the points -- ~13 us of dependent fp64 adds
    // plus sixteen workgroup barriers -- per accepted iteration.
    auto full_pass = [&](const double* h, bool trial) {
        double a = 0;
        for (int c0 = 0; c0 < np; c0 += 256) {
            const int cnt = np - c0 < 256 ? np - c0 : 256;
            const float* cp = chunk_pts(c0);
            h_chunk_lm(sh, cp, h, cnt, true);
            __syncthreads();
            if (t < 36) {
                // One loop for all 36 lanes.  (Until round 5 the lanes whose row-a or row-b factor is structurally zero -- row a
                // is non-zero in columns {0,1,2,6,7}, row b in {3,4,5,6,7} -- skipped that term in loops of their own: three
                // divergent loops run one after the other by the same wavefront, 60 us per pass over 590 inliers instead of 20.)
                // The extra term is an exact +-0 product, and x + (+-0) == x for every x the running sum can hold (it starts at
                // +0 and can never become -0): the sums are the same bits, and the same operations the reference's GEMM performs.
                a = seq_acc_two(sh.buf, cnt, ai, aj, 8 + ai, 8 + aj, a);
            } else if (t >= 64 && t < 72) {
                a = seq_acc_two(sh.buf, cnt, t - 64, 16, 8 + t - 64, 17, a);
            } else if (t == 128) {
                a = seq_acc_sq(sh.buf, cnt, a);
            } else if (t == 192) {
                int k = 0;
                for (; k + 8 <= cnt; k += 8) {
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        v[2 * u] = fabs(sh.buf[(k + u) * 18 + 16]);
                        v[2 * u + 1] = fabs(sh.buf[(k + u) * 18 + 17]);
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) a = a > v[u] ? a : v[u];
                }
                for (; k < cnt; k++) {
                    const double v0 = fabs(sh.buf[k * 18 + 16]), v1 = fabs(sh.buf[k * 18 + 17]);
                    a = a > v0 ? a : v0;
                    a = a > v1 ? a : v1;
                }
            }
            __syncthreads();
        }
        double* const dA = trial ? w.At : s_A;
        double* const dv = trial ? w.vt : s_v;
        if (t < 36) {
            dA[ai * 8 + aj] = a;
            dA[aj * 8 + ai] = a;
        } else if (t >= 64 && t < 72) {
            dv[t - 64] = a;
        } else if (t == 128) {
            if (trial)
                s_Sd = a;
            else
                s_S = a;
        } else if (t == 192) {
            if (trial)
                w.rinft = a;
            else
                s_rinf = a;
        }
        __syncthreads();
    };
    full_pass(s_x, false);
    mark(5);
    if (t < 8) s_D[t] = s_A[t * 8 + t];
    if (t == 0) {
        s_lambda = 1;
        s_lc = 0.75;
    }
    __syncthreads();
    const double Rlo = 0.25, Rhi = 0.75;
    const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    int iter = 0;
    for (;;) {
        if (t < 64) s_Ap[t] = s_A[t];
        __syncthreads();
        if (t < 8) s_Ap[t * 8 + t] += s_lambda * s_D[t];
        __syncthreads();
        // solve(Ap, v, d, DECOMP_EIG): eigen decomposition on wave 0 (in place), back-substitution on lane 0
        long long tr0 = trace ? wall_clock64() : 0;
        if (wave == 0) {
            const int rot = jacobi_eigen_coop<8>(s_Ap, s_eig + 64, s_eig, lane);
            if (trace && t == 0) trace[14] += rot;
        }
        __syncthreads();
        add_since(8, tr0);
        tr0 = trace ? wall_clock64() : 0;
        if (t == 0) {
            sm::svbksb_eig_vec<8>(s_eig + 64, s_eig, s_v, s_d);
            for (int i = 0; i < 8; i++) s_xd[i] = s_x[i] - s_d[i];
        }
        __syncthreads();
        add_since(9, tr0);
        tr0 = trace ? wall_clock64() : 0;
        full_pass(s_xd, true);  // |r(x - d)|^2 -> s_Sd, and the Jacobian sums the step needs if it is accepted
        add_since(10, tr0);
        tr0 = trace ? wall_clock64() : 0;
        if (t == 0) {
            const double Sd = s_Sd;
            double temp_d[8], d[8], v[8];
            for (int i = 0; i < 8; i++) {
                d[i] = s_d[i];
                v[i] = s_v[i];
            }
            for (int i = 0; i < 8; i++) {
                double sacc = 0;
                for (int k = 0; k < 8; k++) sacc += s_A[i * 8 + k] * d[k];
                temp_d[i] = sacc * -1. + v[i] * 2.;
            }
            const double dS = seq_dot8(d, temp_d);
            const double S = s_S;
            const double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
            double lambda = s_lambda, lc = s_lc;
            if (R > Rhi) {
                lambda *= 0.5;
                if (lambda < lc) lambda = 0;
            } else if (R < Rlo) {
                const double tt = seq_dot8(d, v);
                double nu = (Sd - S) / (fabs(tt) > DBL_EPSILON ? tt : 1) + 2;
                nu = nu > 2. ? nu : 2.;
                nu = nu < 10. ? nu : 10.;
                if (lambda == 0) {
                    double* Ai = s_Ap;
                    sm::invert_eig_ws<8>(s_A, Ai, s_eig);
                    double maxval = DBL_EPSILON;
                    for (int i = 0; i < 8; i++) maxval = maxval > fabs(Ai[i * 8 + i]) ? maxval : fabs(Ai[i * 8 + i]);
                    lambda = lc = 1. / maxval;
                    nu *= 0.5;
                }
                lambda *= nu;
            }
            s_lambda = lambda;
            s_lc = lc;
            s_flag = Sd < S ? 1 : 0;
        }
        __syncthreads();
        add_since(11, tr0);
        if (s_flag) {
            if (t < 8) {
                const double tx = s_x[t];
                s_x[t] = s_xd[t];
                s_xd[t] = tx;
            }
            // the trial set becomes the current one (what a second evaluation at the new x would have produced)
            if (t < 64) s_A[t] = w.At[t];
            if (t >= 64 && t < 72) s_v[t - 64] = w.vt[t - 64];
            if (t == 128) s_S = s_Sd;
            if (t == 192) s_rinf = w.rinft;
            __syncthreads();
        }
        iter++;
        double dinf = 0;
        for (int i = 0; i < 8; i++) dinf = dinf > fabs(s_d[i]) ? dinf : fabs(s_d[i]);
        const bool proceed = iter < 10 && dinf >= epsx && s_rinf >= epsf;
        __syncthreads();
        if (!proceed) break;
    }
    if (trace && t == 0) trace[12] = iter;
    mark(6);
}

}  // namespace dfvo
