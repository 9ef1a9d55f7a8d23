// Scalar f64 building blocks of the pose solvers (device functions; one hypothesis per lane).
//
// These follow the OpenCV 3.4.3 routines behind cv2.findEssentialMat / findHomography / recoverPose /
// triangulatePoints (call sites: /root/reference/libs/tracker/E_tracker.py:199-205,231-239,292-295,
// libs/geometry/ops_3d.py:63): the Jacobi SVD / eigen solvers, LU inverse, Durand-Kerner polynomial
// solver, the Nister five-point kernel, the normalised-DLT homography kernel and their error
// functions, written so that every floating-point operation happens in the same order as in the
// sequential CPU algorithm (the file is compiled with -ffp-contract=off): results are bit-identical
// to the CPU oracle, which is what makes the RANSAC inlier masks reproducible.
//
// No HIP intrinsics here: the kernels in solver_*.hip call these functions per lane; tests also build
// this header for the host (tests/host_harness) to check it against the oracle without a GPU.
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define SM_HD __host__ __device__ __forceinline__
#define SM_HD_NOINLINE __host__ __device__ __noinline__ inline
#else
#define SM_HD inline
#define SM_HD_NOINLINE inline
#endif

namespace sm {

// ------------------------------------------------------------------------------------------------
// cv::RNG (multiply-with-carry)
// ------------------------------------------------------------------------------------------------
struct CvRng {
    uint64_t state;
};
SM_HD void cvrng_init(CvRng& r, uint64_t seed) { r.state = seed ? seed : 0xffffffffULL; }
SM_HD unsigned cvrng_next(CvRng& r) {
    r.state = (uint64_t)(unsigned)r.state * 4164903690U + (unsigned)(r.state >> 32);
    return (unsigned)r.state;
}
SM_HD int cvrng_uniform(CvRng& r, int a, int b) { return a == b ? a : (int)(cvrng_next(r) % (unsigned)(b - a) + a); }

// hypot from IEEE-exact operations only (+ * / sqrt): libm's hypot differs between glibc and the
// device library in the last bit, which would break bit-reproducibility of the Jacobi rotations.
SM_HD double hypot_p(double x, double y) {
    double a = fabs(x), b = fabs(y);
    if (a < b) {
        const double t = a;
        a = b;
        b = t;
    }
    if (a == 0) return 0;
    const double r = b / a;
    return a * sqrt(1 + r * r);
}

SM_HD double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
SM_HD double matx_det3(const double* a) {
    return a[0] * (a[4] * a[8] - a[7] * a[5]) - a[1] * (a[3] * a[8] - a[6] * a[5]) + a[2] * (a[3] * a[7] - a[6] * a[4]);
}
SM_HD void mul33(const double* a, const double* b, double* d) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            t[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
    for (int i = 0; i < 9; i++) d[i] = t[i];
}

// ------------------------------------------------------------------------------------------------
// One-sided Jacobi SVD on the rows of At (n rows of length m, row stride astep).
// W: n values (scratch + output), Vt: n x n or nullptr, n1: rows of At normalised on output; rows
// n..n1-1 are completed with OpenCV's deterministic pseudo-random orthogonal basis.
// ------------------------------------------------------------------------------------------------
SM_HD void jacobi_svd_impl(double* At, int astep, double* W, double* Vt, int vstep, int m, int n, int n1) {
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    int i, j, k, iter;
    const int max_iter = m > 30 ? m : 30;
    double c, s, sd;
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            const double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }
    for (iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot_p(p, beta);
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    const double t0 = c * Ai[k] + s * Aj[k];
                    const double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                if (Vt) {
                    double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        const double t0 = c * Vi[k] + s * Vj[k];
                        const double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            const double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            const double tw = W[i];
            W[i] = W[j];
            W[j] = tw;
            if (Vt) {
                for (k = 0; k < m; k++) {
                    const double t = At[i * astep + k];
                    At[i * astep + k] = At[j * astep + k];
                    At[j * astep + k] = t;
                }
                for (k = 0; k < n; k++) {
                    const double t = Vt[i * vstep + k];
                    Vt[i * vstep + k] = Vt[j * vstep + k];
                    Vt[j * vstep + k] = t;
                }
            }
        }
    }
    if (!Vt) return;
    CvRng rng;
    cvrng_init(rng, 0x12345678);
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / m;
            for (k = 0; k < m; k++) At[i * astep + k] = (cvrng_next(rng) & 256) != 0 ? val0 : -val0;
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) {
                        const double t = At[i * astep + k] - sd * At[j * astep + k];
                        At[i * astep + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++) At[i * astep + k] *= asum;
                }
            }
            sd = 0;
            for (k = 0; k < m; k++) {
                const double t = At[i * astep + k];
                sd += t * t;
            }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
}

SM_HD_NOINLINE void jacobi_svd(double* At, int astep, double* W, double* Vt, int vstep, int m, int n, int n1) {
    jacobi_svd_impl(At, astep, W, Vt, vstep, m, n, n1);
}

// SVD of a square N x N matrix the way cv::SVD::compute(src, w, u, vt) does it (flags = 0):
// rows of `at` = columns of src on input; on output u = at^T (columns), vt as is.
template <int N>
SM_HD void svd_square(const double* src, double* w, double* u, double* vt) {
    double at[N * N];
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) at[j * N + i] = src[i * N + j];
    jacobi_svd(at, N, w, vt, N, N, N, N);
    if (u)
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) u[j * N + i] = at[i * N + j];
}

// ------------------------------------------------------------------------------------------------
// Jacobi eigen decomposition of a symmetric n x n matrix (rows of V = eigenvectors, descending)
// ------------------------------------------------------------------------------------------------
// `ind` = 2N ints of scratch (the row/column pivot tables); callers on the GPU hand in LDS so that the
// data-dependent indexing does not go through scratch memory.
template <int N>
SM_HD void jacobi_eigen_ws(double* A, double* W, double* V, int* ind) {
    const double eps = DBL_EPSILON;
    const int n = N, astep = N, vstep = N;
    int i, j, k, m;
    int *indR = ind, *indC = ind + N;
    for (i = 0; i < n; i++) {
        for (j = 0; j < n; j++) V[i * vstep + j] = 0;
        V[i * vstep + i] = 1;
    }
    const int maxIters = n * n * 30;
    double mv = 0;
    for (k = 0; k < n; k++) {
        W[k] = A[(astep + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = fabs(A[astep * k + m]), i = k + 2; i < n; i++) {
                const double val = fabs(A[astep * k + i]);
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(A[k]), i = 1; i < k; i++) {
                const double val = fabs(A[astep * i + k]);
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    if (n > 1)
        for (int iters = 0; iters < maxIters; iters++) {
            for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
                const double val = fabs(A[astep * i + indR[i]]);
                if (mv < val) mv = val, k = i;
            }
            int l = indR[k];
            for (i = 1; i < n; i++) {
                const double val = fabs(A[astep * indC[i] + i]);
                if (mv < val) mv = val, k = indC[i], l = i;
            }
            const double p = A[astep * k + l];
            if (fabs(p) <= eps) break;
            const double y = (W[l] - W[k]) * 0.5;
            double t = fabs(y) + hypot_p(p, y);
            double s = hypot_p(p, t);
            const double c = t / s;
            s = p / s;
            t = (p / t) * p;
            if (y < 0) s = -s, t = -t;
            A[astep * k + l] = 0;
            W[k] -= t;
            W[l] += t;
            double a0, b0;
#define SM_ROTATE(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
            for (i = 0; i < k; i++) SM_ROTATE(A[astep * i + k], A[astep * i + l]);
            for (i = k + 1; i < l; i++) SM_ROTATE(A[astep * k + i], A[astep * i + l]);
            for (i = l + 1; i < n; i++) SM_ROTATE(A[astep * k + i], A[astep * l + i]);
            for (i = 0; i < n; i++) SM_ROTATE(V[vstep * k + i], V[vstep * l + i]);
#undef SM_ROTATE
            for (j = 0; j < 2; j++) {
                const int idx = j == 0 ? k : l;
                if (idx < n - 1) {
                    for (m = idx + 1, mv = fabs(A[astep * idx + m]), i = idx + 2; i < n; i++) {
                        const double val = fabs(A[astep * idx + i]);
                        if (mv < val) mv = val, m = i;
                    }
                    indR[idx] = m;
                }
                if (idx > 0) {
                    for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; i++) {
                        const double val = fabs(A[astep * i + idx]);
                        if (mv < val) mv = val, m = i;
                    }
                    indC[idx] = m;
                }
            }
        }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++)
            if (W[m] < W[i]) m = i;
        if (k != m) {
            double t = W[m];
            W[m] = W[k];
            W[k] = t;
            for (i = 0; i < n; i++) {
                t = V[vstep * m + i];
                V[vstep * m + i] = V[vstep * k + i];
                V[vstep * k + i] = t;
            }
        }
    }
}

template <int N>
SM_HD_NOINLINE void jacobi_eigen(double* A, double* W, double* V) {
    int ind[2 * N];
    jacobi_eigen_ws<N>(A, W, V, ind);
}

// solve(A, b, x, DECOMP_EIG) / invert(A, DECOMP_EIG) for symmetric A (SVBkSb back-substitution)
// ws: 2*N*N + 2*N doubles
// SVBkSb for solve(DECOMP_EIG): x = sum_i (v_i . b / w_i) v_i over the eigenvector rows v_i
template <int N>
SM_HD void svbksb_eig_vec(const double* w, const double* v, const double* b, double* x);
template <int N>
SM_HD void solve_eig_ws(const double* A, const double* b, double* x, double* ws) {
    double *a = ws, *v = ws + N * N, *w = ws + 2 * N * N;
    for (int i = 0; i < N * N; i++) a[i] = A[i];
    jacobi_eigen_ws<N>(a, w, v, reinterpret_cast<int*>(ws + 2 * N * N + N));
    svbksb_eig_vec<N>(w, v, b, x);
}
template <int N>
SM_HD void svbksb_eig_vec(const double* w, const double* v, const double* b, double* x) {
    const double eps = DBL_EPSILON * 2;
    double threshold = 0;
    for (int i = 0; i < N; i++) x[i] = 0;
    for (int i = 0; i < N; i++) threshold += w[i];
    threshold *= eps;
    for (int i = 0; i < N; i++) {
        const double* ui = v + i * N;
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < N; j++) s += ui[j] * b[j];
        s *= wi;
        for (int j = 0; j < N; j++) x[j] = x[j] + s * ui[j];
    }
}
template <int N>
SM_HD void solve_eig(const double* A, const double* b, double* x) {
    double ws[2 * N * N + 2 * N];
    solve_eig_ws<N>(A, b, x, ws);
}
template <int N>
SM_HD void invert_eig_ws(const double* A, double* dst, double* ws) {
    double *a = ws, *v = ws + N * N, *w = ws + 2 * N * N;
    for (int i = 0; i < N * N; i++) a[i] = A[i];
    jacobi_eigen_ws<N>(a, w, v, reinterpret_cast<int*>(ws + 2 * N * N + N));
    const double eps = DBL_EPSILON * 2;
    double threshold = 0;
    for (int i = 0; i < N * N; i++) dst[i] = 0;
    for (int i = 0; i < N; i++) threshold += w[i];
    threshold *= eps;
    for (int i = 0; i < N; i++) {
        const double* ui = v + i * N;
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        for (int j = 0; j < N; j++) {
            double s = ui[j];
            s *= wi;
            for (int k = 0; k < N; k++) dst[k * N + j] = dst[k * N + j] + s * ui[k];
        }
    }
}

template <int N>
SM_HD void invert_eig(const double* A, double* dst) {
    double ws[2 * N * N + 2 * N];
    invert_eig_ws<N>(A, dst, ws);
}

// ------------------------------------------------------------------------------------------------
// LU with partial pivoting, in place; b is m x n right-hand sides.  Returns 0 when singular.
// ------------------------------------------------------------------------------------------------
SM_HD int lu_solve_impl(double* A, int astep, int m, double* b, int bstep, int n) {
    const double eps = DBL_EPSILON * 100;
    int i, j, k, p = 1;
    for (i = 0; i < m; i++) {
        k = i;
        for (j = i + 1; j < m; j++)
            if (fabs(A[j * astep + i]) > fabs(A[k * astep + i])) k = j;
        if (fabs(A[k * astep + i]) < eps) return 0;
        if (k != i) {
            for (j = i; j < m; j++) {
                const double t = A[i * astep + j];
                A[i * astep + j] = A[k * astep + j];
                A[k * astep + j] = t;
            }
            for (j = 0; j < n; j++) {
                const double t = b[i * bstep + j];
                b[i * bstep + j] = b[k * bstep + j];
                b[k * bstep + j] = t;
            }
            p = -p;
        }
        const double d = -1 / A[i * astep + i];
        for (j = i + 1; j < m; j++) {
            const double alpha = A[j * astep + i] * d;
            for (k = i + 1; k < m; k++) A[j * astep + k] += alpha * A[i * astep + k];
            for (k = 0; k < n; k++) b[j * bstep + k] += alpha * b[i * bstep + k];
        }
    }
    for (i = m - 1; i >= 0; i--)
        for (j = 0; j < n; j++) {
            double s = b[i * bstep + j];
            for (k = i + 1; k < m; k++) s -= A[i * astep + k] * b[k * bstep + j];
            b[i * bstep + j] = s / A[i * astep + i];
        }
    return p;
}
SM_HD_NOINLINE int lu_solve(double* A, int astep, int m, double* b, int bstep, int n) {
    return lu_solve_impl(A, astep, m, b, bstep, n);
}

// ------------------------------------------------------------------------------------------------
// Durand-Kerner roots of sum_i c[i] x^i, degree 10 (cv::solvePoly, 300 iterations), register resident
// ------------------------------------------------------------------------------------------------
template <int N>
SM_HD void solve_poly_fixed(const double* c, double* rre, double* rim) {
    // N is a compile-time constant so that every index below is static after unrolling: the 2N root
    // components and N+1 coefficients stay in registers on the GPU (300 x N x N complex steps per call)
    double cr[N + 1], xr[N], xi[N];
#pragma unroll
    for (int i = 0; i <= N; i++) cr[i] = c[i];
    {
        double pre = 1, pim = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            xr[i] = pre;
            xi[i] = pim;
            const double tre = pre * 1.0 - pim * 1.0, tim = pre * 1.0 + pim * 1.0;
            pre = tre;
            pim = tim;
        }
    }
    for (int iter = 0; iter < 300; iter++) {
        double maxDiff = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const double pre = xr[i], pim = xi[i];
            double nre = cr[N], nim = 0, dre = cr[N], dim = 0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                double tre = nre * pre - nim * pim, tim = nre * pim + nim * pre;
                nre = tre + cr[N - j - 1];
                nim = tim + 0.0;
                if (j != i) {
                    const double qre = pre - xr[j], qim = pim - xi[j];
                    tre = dre * qre - dim * qim;
                    tim = dre * qim + dim * qre;
                    dre = tre;
                    dim = tim;
                }
            }
            const double t = 1. / (dre * dre + dim * dim);
            const double qre = (nre * dre + nim * dim) * t, qim = (-nre * dim + nim * dre) * t;
            xr[i] = pre - qre;
            xi[i] = pim - qim;
            const double an = qre * qre + qim * qim;  // |q|^2: only `maxDiff <= 0` is tested, and sqrt keeps 0 / NaN / order
            maxDiff = maxDiff > an ? maxDiff : an;
        }
        if (maxDiff <= 0) break;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        rre[i] = xr[i];
        rim[i] = fabs(xi[i]) < 1e-100 ? 0 : xi[i];
    }
}

// generic degree (leading coefficients ~ 0): same algorithm with run-time n
SM_HD_NOINLINE void solve_poly_generic(const double* c, int n, double* rre, double* rim) {
    double pre = 1, pim = 0;
    for (int i = 0; i < n; i++) {
        rre[i] = pre;
        rim[i] = pim;
        const double tre = pre * 1.0 - pim * 1.0, tim = pre * 1.0 + pim * 1.0;
        pre = tre;
        pim = tim;
    }
    for (int iter = 0; iter < 300; iter++) {
        double maxDiff = 0;
        for (int i = 0; i < n; i++) {
            pre = rre[i];
            pim = rim[i];
            double nre = c[n], nim = 0, dre = c[n], dim = 0;
            for (int j = 0; j < n; j++) {
                double tre = nre * pre - nim * pim, tim = nre * pim + nim * pre;
                nre = tre + c[n - j - 1];
                nim = tim + 0.0;
                if (j != i) {
                    const double qre = pre - rre[j], qim = pim - rim[j];
                    tre = dre * qre - dim * qim;
                    tim = dre * qim + dim * qre;
                    dre = tre;
                    dim = tim;
                }
            }
            const double t = 1. / (dre * dre + dim * dim);
            const double qre = (nre * dre + nim * dim) * t, qim = (-nre * dim + nim * dre) * t;
            rre[i] = pre - qre;
            rim[i] = pim - qim;
            const double an = qre * qre + qim * qim;  // |q|^2: only `maxDiff <= 0` is tested, and sqrt keeps 0 / NaN / order
            maxDiff = maxDiff > an ? maxDiff : an;
        }
        if (maxDiff <= 0) break;
    }
    for (int i = 0; i < n; i++)
        if (fabs(rim[i]) < 1e-100) rim[i] = 0;
}

SM_HD_NOINLINE void solve_poly10(const double* c, double* rre, double* rim) {
    int n = 10;
    for (; n > 1; n--)
        if (fabs(c[n]) + 0.0 > DBL_EPSILON) break;
    if (n == 10) {
        solve_poly_fixed<10>(c, rre, rim);
        return;
    }
    solve_poly_generic(c, n, rre, rim);
    for (; n < 10; n++) {
        rre[n] = rre[n - 1];
        rim[n] = rim[n - 1];
    }
}

// ------------------------------------------------------------------------------------------------
// five-point kernel
// ------------------------------------------------------------------------------------------------
// monomial tables (see oracle/cv3_calib3d.c for the derivation): linear [x y z 1], quadratic
// [x2 y2 z2 xy xz yz x y z 1], cubic columns [x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1]
SM_HD int qidx(int i, int j) {
    const int t[16] = {0, 3, 4, 6, 3, 1, 5, 7, 4, 5, 2, 8, 6, 7, 8, 9};
    return t[i * 4 + j];
}
SM_HD int cidx(int q, int j) {
    const int t[40] = {0,  2,  4,  5,  3,  1,  6,  7,  10, 13, 16, 17, 2,  3,  8,  9,  4,  8,  10, 11,
                       8,  6,  13, 14, 5,  9,  11, 12, 9,  7,  14, 15, 11, 14, 17, 18, 12, 15, 18, 19};
    return t[q * 4 + j];
}
SM_HD void lin_mul_acc(const double* a, const double* b, double* out) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[qidx(i, j)] = out[qidx(i, j)] + a[i] * b[j];
}
SM_HD void quad_mul_acc(const double* q, const double* l, bool plus, double* out) {
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 4; j++) {
            const double p = q[i] * l[j];
            out[cidx(i, j)] = plus ? out[cidx(i, j)] + p : out[cidx(i, j)] - p;
        }
}

// stage 1: 5 correspondences -> null-space basis EE[4][9] and the 3 x 13 polynomial matrix b, plus
// the degree-10 coefficients c[11].  Returns false if the 10x10 block is singular.
// ws layout (FIVE_POINT_WS doubles; on the GPU a per-lane LDS slice, see k_e_stage1):
//   [0,200)   A (10 x 20)
//   [200,400) phase 1: Vt[81] W[5] V5[25]; phase 2: L[36] m0 m1 m2 t[40] EEt[90] tr[10]; phase 3: A1[100] inv[100]
//   [400,460) Ar (6 x 10)
constexpr int FIVE_POINT_WS = 460;
SM_HD bool five_point_stage1_ws(const double* q1, const double* q2, double* EE, double* b, double* c, double* ws) {
    double *Vt = ws + 200, *W = ws + 281;
    for (int i = 0; i < 81; i++) Vt[i] = 0;
    for (int i = 0; i < 5; i++) {
        const double x1 = q1[i * 2], y1 = q1[i * 2 + 1], x2 = q2[i * 2], y2 = q2[i * 2 + 1];
        double* r = Vt + i * 9;
        r[0] = x1 * x2;
        r[1] = y1 * x2;
        r[2] = x2;
        r[3] = x1 * y2;
        r[4] = y1 * y2;
        r[5] = y2;
        r[6] = x1;
        r[7] = y1;
        r[8] = 1.0;
    }
    jacobi_svd_impl(Vt, 9, W, ws + 286, 5, 9, 5, 9);  // SVD::compute(Q, FULL_UV): m < n -> works on Q's rows
    for (int i = 0; i < 36; i++) EE[i] = Vt[45 + i];
    // ---- coefficient matrix A (10 x 20)
    double* A = ws;
    for (int i = 0; i < 200; i++) A[i] = 0;
    double(*L)[4] = reinterpret_cast<double(*)[4]>(ws + 200);
    for (int e = 0; e < 9; e++)
        for (int v = 0; v < 4; v++) L[e][v] = EE[v * 9 + e];
    {
        double *m0 = ws + 236, *m1 = ws + 246, *m2 = ws + 256, *t = ws + 266;
        for (int k = 0; k < 10; k++) m0[k] = m1[k] = m2[k] = t[k] = 0;
        lin_mul_acc(L[4], L[8], m0);
        lin_mul_acc(L[5], L[7], t);
        for (int k = 0; k < 10; k++) m0[k] = m0[k] - t[k], t[k] = 0;
        lin_mul_acc(L[3], L[8], m1);
        lin_mul_acc(L[5], L[6], t);
        for (int k = 0; k < 10; k++) m1[k] = m1[k] - t[k], t[k] = 0;
        lin_mul_acc(L[3], L[7], m2);
        lin_mul_acc(L[4], L[6], t);
        for (int k = 0; k < 10; k++) m2[k] = m2[k] - t[k];
        quad_mul_acc(m0, L[0], true, A);
        quad_mul_acc(m1, L[1], false, A);
        quad_mul_acc(m2, L[2], true, A);
    }
    {
        double(*EEt)[10] = reinterpret_cast<double(*)[10]>(ws + 276);
        double* tr = ws + 366;
        for (int i = 0; i < 9; i++)
            for (int k = 0; k < 10; k++) EEt[i][k] = 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) lin_mul_acc(L[i * 3 + k], L[j * 3 + k], EEt[i * 3 + j]);
        for (int k = 0; k < 10; k++) tr[k] = (EEt[0][k] + EEt[4][k]) + EEt[8][k];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double* row = A + (1 + i * 3 + j) * 20;
                for (int k = 0; k < 3; k++) quad_mul_acc(EEt[i * 3 + k], L[k * 3 + j], true, row);
                for (int cc = 0; cc < 20; cc++) row[cc] = 2.0 * row[cc];
                quad_mul_acc(tr, L[i * 3 + j], false, row);
            }
    }
    // ---- A(:,0:10)^-1 via LU on the identity, then times A(:,10:20)
    double *A1 = ws + 200, *inv = ws + 300;
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            A1[i * 10 + j] = A[i * 20 + j];
            inv[i * 10 + j] = i == j ? 1.0 : 0.0;
        }
    if (!lu_solve_impl(A1, 10, 10, inv, 10, 10)) return false;
    double* Ar = ws + 400;  // rows 4..9 only
    for (int i = 4; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            double s = 0;
            for (int k = 0; k < 10; k++) s += inv[i * 10 + k] * A[k * 20 + 10 + j];
            Ar[(i - 4) * 10 + j] = s;
        }
    for (int i = 0; i < 3; i++) {
        const double* r1 = Ar + (i * 2) * 10;
        const double* r2 = Ar + (i * 2 + 1) * 10;
        double row1[13], row2[13];
        for (int k = 0; k < 13; k++) row1[k] = row2[k] = 0;
        for (int k = 0; k < 3; k++) {
            row1[1 + k] = r1[k];
            row1[5 + k] = r1[3 + k];
            row2[k] = r2[k];
            row2[4 + k] = r2[3 + k];
        }
        for (int k = 0; k < 4; k++) {
            row1[9 + k] = r1[6 + k];
            row2[8 + k] = r2[6 + k];
        }
        for (int k = 0; k < 13; k++) b[i * 13 + k] = row1[k] - row2[k];
    }
    // ---- determinant polynomial
    double p[9][5];
    for (int j = 0; j < 3; j++) {
        const double* br = b + j * 13;
        for (int k = 0; k < 4; k++) {
            p[j * 3 + 0][k] = br[3 - k];
            p[j * 3 + 1][k] = br[7 - k];
        }
        p[j * 3 + 0][4] = p[j * 3 + 1][4] = 0;
        for (int k = 0; k < 5; k++) p[j * 3 + 2][k] = br[12 - k];
    }
    double t1[8], t2[8], mm[8], pr[11];
    for (int k = 0; k < 11; k++) c[k] = 0;
    auto pmul = [](const double* a, int na, const double* bb, int nb, double* out) {
        for (int i = 0; i < na + nb - 1; i++) out[i] = 0;
        for (int i = 0; i < na; i++)
            for (int j = 0; j < nb; j++) out[i + j] = out[i + j] + a[i] * bb[j];
    };
    pmul(p[4], 4, p[8], 5, t1);
    pmul(p[5], 5, p[7], 4, t2);
    for (int k = 0; k < 8; k++) mm[k] = t1[k] - t2[k];
    pmul(p[0], 4, mm, 8, pr);
    for (int k = 0; k < 11; k++) c[k] = c[k] + pr[k];
    pmul(p[3], 4, p[8], 5, t1);
    pmul(p[5], 5, p[6], 4, t2);
    for (int k = 0; k < 8; k++) mm[k] = t1[k] - t2[k];
    pmul(p[1], 4, mm, 8, pr);
    for (int k = 0; k < 11; k++) c[k] = c[k] - pr[k];
    pmul(p[3], 4, p[7], 4, t1);
    pmul(p[4], 4, p[6], 4, t2);
    for (int k = 0; k < 7; k++) mm[k] = t1[k] - t2[k];
    pmul(p[2], 5, mm, 7, pr);
    for (int k = 0; k < 11; k++) c[k] = c[k] + pr[k];
    return true;
}

SM_HD_NOINLINE bool five_point_stage1(const double* q1, const double* q2, double* EE, double* b, double* c) {
    double ws[FIVE_POINT_WS];
    return five_point_stage1_ws(q1, q2, EE, b, c, ws);
}

// stage 3 for ONE root: false when the root is complex or its (x, y, 1) null vector degenerates; else E (9 doubles,
// unit Frobenius norm).  five_point_stage3 below is this in root order with the survivors compacted; on the GPU the
// ten roots of a hypothesis run on ten lanes (k_e_stage3).
SM_HD bool five_point_root_to_E(const double* EE, const double* b, double rre, double rim, double* Ev) {
    if (fabs(rim) > 1e-10) return false;
    const double z1 = rre;
    const double z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
    double bz[9];
    for (int j = 0; j < 3; j++) {
        const double* br = b + j * 13;
        bz[j * 3 + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
        bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
        bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
    }
    double w3[3], vt3[9];
    svd_square<3>(bz, w3, nullptr, vt3);
    const double* xy1 = vt3 + 6;
    if (fabs(xy1[2]) < 1e-10) return false;
    const double xs = xy1[0] / xy1[2], ys = xy1[1] / xy1[2], zs = z1;
    for (int k = 0; k < 9; k++) {
        const double t = EE[k] * xs + EE[9 + k] * ys;
        const double u = t + EE[18 + k] * zs;
        Ev[k] = u + EE[27 + k];
    }
    double s = 0;
    s += Ev[0] * Ev[0] + Ev[1] * Ev[1] + Ev[2] * Ev[2] + Ev[3] * Ev[3];
    s += Ev[4] * Ev[4] + Ev[5] * Ev[5] + Ev[6] * Ev[6] + Ev[7] * Ev[7];
    s += Ev[8] * Ev[8];
    const double inv = 1. / sqrt(s);
    for (int k = 0; k < 9; k++) Ev[k] = Ev[k] * inv;
    return true;
}

// stage 3: real roots -> essential matrices (up to 10 x 9 doubles); returns their number
SM_HD_NOINLINE int five_point_stage3(const double* EE, const double* b, const double* rre, const double* rim,
                                     double* E_out) {
    int count = 0;
    for (int i = 0; i < 10; i++)
        if (five_point_root_to_E(EE, b, rre[i], rim[i], E_out + count * 9)) count++;
    return count;
}

// Sampson error of one correspondence under E (EMEstimatorCallback::computeError), as float
SM_HD float essential_error(const double* E, double x1x, double x1y, double x2x, double x2y) {
    const double x1[3] = {x1x, x1y, 1.}, x2[3] = {x2x, x2y, 1.};
    double Ex1[3], Etx2[3];
    for (int r = 0; r < 3; r++) {
        double s = 0, s2 = 0;
        for (int k = 0; k < 3; k++) {
            s += E[r * 3 + k] * x1[k];
            s2 += E[k * 3 + r] * x2[k];
        }
        Ex1[r] = s;
        Etx2[r] = s2;
    }
    double x2tEx1 = 0;
    for (int k = 0; k < 3; k++) x2tEx1 += x2[k] * Ex1[k];
    const double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
    return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

// ------------------------------------------------------------------------------------------------
// homography
// ------------------------------------------------------------------------------------------------
SM_HD bool have_collinear_points(const float* pts, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = pts[j * 2] - pts[i * 2];
        const double dy1 = pts[j * 2 + 1] - pts[i * 2 + 1];
        for (int k = 0; k < j; k++) {
            const double dx2 = pts[k * 2] - pts[i * 2];
            const double dy2 = pts[k * 2 + 1] - pts[i * 2 + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return true;
        }
    }
    return false;
}

SM_HD bool homography_check_subset(const float* src, const float* dst) {
    if (have_collinear_points(src, 4) || have_collinear_points(dst, 4)) return false;
    const int tt[12] = {0, 1, 2, 1, 2, 3, 0, 2, 3, 0, 1, 3};
    int negative = 0;
    for (int i = 0; i < 4; i++) {
        const int* t = tt + i * 3;
        const double A[9] = {src[t[0] * 2], src[t[0] * 2 + 1], 1., src[t[1] * 2], src[t[1] * 2 + 1], 1.,
                             src[t[2] * 2], src[t[2] * 2 + 1], 1.};
        const double B[9] = {dst[t[0] * 2], dst[t[0] * 2 + 1], 1., dst[t[1] * 2], dst[t[1] * 2 + 1], 1.,
                             dst[t[2] * 2], dst[t[2] * 2 + 1], 1.};
        negative += matx_det3(A) * matx_det3(B) < 0;
    }
    return negative == 0 || negative == 4;
}

// normalisation constants of HomographyEstimatorCallback::runKernel; returns false when degenerate
struct HNorm {
    double cMx, cMy, cmx, cmy, sMx, sMy, smx, smy;
};
// accumulate LtL row contribution of one correspondence (upper triangle, row-major 9x9)
SM_HD void homography_accumulate(const HNorm& h, float Mx_, float My_, float mx_, float my_, double* LtL) {
    const double x = (mx_ - h.cmx) * h.smx, y = (my_ - h.cmy) * h.smy;
    const double X = (Mx_ - h.cMx) * h.sMx, Y = (My_ - h.cMy) * h.sMy;
    const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
    const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
    for (int j = 0; j < 9; j++)
        for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
}
// finish: symmetric completion, eigen decomposition, de-normalisation, scale so that H[8] = 1
SM_HD void homography_denormalise(const HNorm& h, const double* H0v, double* model);
constexpr int HOMOGRAPHY_FINISH_WS = 9 + 81 + 9;  // doubles: W, V, pivot tables
constexpr int HOMOGRAPHY_KERNEL_WS = 81 + HOMOGRAPHY_FINISH_WS;
SM_HD void homography_finish_ws(const HNorm& h, double* LtL, double* model, double* ws) {
    double *W = ws, *V = ws + 9;
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
    jacobi_eigen_ws<9>(LtL, W, V, reinterpret_cast<int*>(ws + 90));
    homography_denormalise(h, V + 72, model);
}
// H = invHnorm * H0 * Hnorm2 scaled so that H[8] = 1, H0 = the eigenvector of the smallest eigenvalue (row-major 3x3)
SM_HD void homography_denormalise(const HNorm& h, const double* H0v, double* model) {
    const double invHnorm[9] = {1. / h.smx, 0, h.cmx, 0, 1. / h.smy, h.cmy, 0, 0, 1};
    const double Hnorm2[9] = {h.sMx, 0, -h.cMx * h.sMx, 0, h.sMy, -h.cMy * h.sMy, 0, 0, 1};
    double Htemp[9], H0[9];
    mul33(invHnorm, H0v, Htemp);
    mul33(Htemp, Hnorm2, H0);
    const double s = 1. / H0[8];
    for (int k = 0; k < 9; k++) model[k] = H0[k] * s;
}
SM_HD_NOINLINE void homography_finish(const HNorm& h, double* LtL, double* model) {
    double ws[HOMOGRAPHY_FINISH_WS];
    homography_finish_ws(h, LtL, model, ws);
}
// whole kernel for a small set (the 4-point minimal sample); M = source points, m = destination
SM_HD bool homography_kernel_ws(const float* M, const float* m, int count, double* model, double* ws) {
    HNorm h;
    h.cMx = h.cMy = h.cmx = h.cmy = h.sMx = h.sMy = h.smx = h.smy = 0;
    for (int i = 0; i < count; i++) {
        h.cmx += m[i * 2];
        h.cmy += m[i * 2 + 1];
        h.cMx += M[i * 2];
        h.cMy += M[i * 2 + 1];
    }
    h.cmx /= count;
    h.cmy /= count;
    h.cMx /= count;
    h.cMy /= count;
    for (int i = 0; i < count; i++) {
        h.smx += fabs(m[i * 2] - h.cmx);
        h.smy += fabs(m[i * 2 + 1] - h.cmy);
        h.sMx += fabs(M[i * 2] - h.cMx);
        h.sMy += fabs(M[i * 2 + 1] - h.cMy);
    }
    if (fabs(h.smx) < DBL_EPSILON || fabs(h.smy) < DBL_EPSILON || fabs(h.sMx) < DBL_EPSILON || fabs(h.sMy) < DBL_EPSILON)
        return false;
    h.smx = count / h.smx;
    h.smy = count / h.smy;
    h.sMx = count / h.sMx;
    h.sMy = count / h.sMy;
    double* LtL = ws;
    for (int i = 0; i < 81; i++) LtL[i] = 0;
    for (int i = 0; i < count; i++) homography_accumulate(h, M[i * 2], M[i * 2 + 1], m[i * 2], m[i * 2 + 1], LtL);
    homography_finish_ws(h, LtL, model, ws + 81);
    return true;
}
SM_HD_NOINLINE bool homography_kernel(const float* M, const float* m, int count, double* model) {
    double ws[HOMOGRAPHY_KERNEL_WS];
    return homography_kernel_ws(M, m, count, model, ws);
}
// squared reprojection error in float (HomographyEstimatorCallback::computeError)
SM_HD float homography_error(const float* Hf, float Mx, float My, float mx, float my) {
    const float ww = 1.f / (Hf[6] * Mx + Hf[7] * My + 1.f);
    const float dx = (Hf[0] * Mx + Hf[1] * My + Hf[2]) * ww - mx;
    const float dy = (Hf[3] * Mx + Hf[4] * My + Hf[5]) * ww - my;
    return dx * dx + dy * dy;
}

// RANSACUpdateNumIters
SM_HD int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = p > 0. ? p : 0.;
    p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.;
    ep = ep < 1. ? ep : 1.;
    double num = (1. - p) > DBL_MIN ? (1. - p) : DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    if (denom >= 0 || -num >= maxIters * (-denom)) return maxIters;
    return (int)rint(num / denom);
}

// decomposeEssentialMat
SM_HD void decompose_essential(const double* E, double* R1, double* R2, double* t) {
    double D[3], U[9], Vt[9];
    svd_square<3>(E, D, U, Vt);
    if (det3(U) < 0)
        for (int i = 0; i < 9; i++) U[i] *= -1.;
    if (det3(Vt) < 0)
        for (int i = 0; i < 9; i++) Vt[i] *= -1.;
    const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    const double Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double UW[9];
    mul33(U, Wm, UW);
    mul33(UW, Vt, R1);
    mul33(U, Wt, UW);
    mul33(UW, Vt, R2);
    t[0] = U[2];
    t[1] = U[5];
    t[2] = U[8];
}

// one point of cv::triangulatePoints (4x4 DLT, last right singular vector)
SM_HD void triangulate_point(const double* P1, const double* P2, double x1, double y1, double x2, double y2,
                             double* X4) {
    double A[16], w[4], vt[16];
    for (int k = 0; k < 4; k++) {
        A[0 * 4 + k] = x1 * P1[2 * 4 + k] - P1[0 * 4 + k];
        A[1 * 4 + k] = y1 * P1[2 * 4 + k] - P1[1 * 4 + k];
        A[2 * 4 + k] = x2 * P2[2 * 4 + k] - P2[0 * 4 + k];
        A[3 * 4 + k] = y2 * P2[2 * 4 + k] - P2[1 * 4 + k];
    }
    svd_square<4>(A, w, nullptr, vt);
    for (int k = 0; k < 4; k++) X4[k] = vt[12 + k];
}

// recoverPose cheirality test of one triangulated point against P = [R|t] (distance threshold 50)
SM_HD bool cheirality_ok(const double* P, const double* Q4, double dist) {
    double X = Q4[0], Y = Q4[1], Z = Q4[2], Wq = Q4[3];
    bool m = (Z * Wq) > 0;
    X = Wq != 0 ? X / Wq : 0;
    Y = Wq != 0 ? Y / Wq : 0;
    Z = Wq != 0 ? Z / Wq : 0;
    Wq = Wq != 0 ? Wq / Wq : 0;
    m = (Z < dist) && m;
    double z2 = 0;
    z2 += P[8] * X;
    z2 += P[9] * Y;
    z2 += P[10] * Z;
    z2 += P[11] * Wq;
    m = (z2 > 0) && m;
    m = (z2 < dist) && m;
    return m;
}

}  // namespace sm
