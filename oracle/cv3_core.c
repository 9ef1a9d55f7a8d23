/* ORACLE -- see cv3_core.h.  Restates modules/core/src/{rand.cpp, lapack.cpp, mathfuncs.cpp,
 * matmul.cpp (small GEMM)} of OpenCV 3.4.3 for doubles.  Compile with -ffp-contract=off. */
#include "cv3_core.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ RNG (rand.cpp) */
void cv3_rng_init(cv3_rng* r, uint64_t seed) { r->state = seed ? seed : 0xffffffffULL; }
unsigned cv3_rng_next(cv3_rng* r) {
    r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
    return (unsigned)r->state;
}
int cv3_rng_uniform_int(cv3_rng* r, int a, int b) { return a == b ? a : (int)(cv3_rng_next(r) % (unsigned)(b - a) + a); }

int cv3_round(double v) { return (int)lrint(v); }

/* std::hypot in OpenCV; restated with IEEE-exact operations only so that the oracle and the GPU
 * solvers agree bit for bit (glibc's and the device library's hypot differ in the last bit) */
static double cv3_hypot(double x, double y) {
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

double cv3_det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

void cv3_mul33(const double* a, const double* b, double* d) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
    memcpy(d, t, sizeof(t));
}

/* ------------------------------------------------------------------ JacobiSVD (lapack.cpp) */
void cv3_jacobi_svd(double* At, int astep, double* _W, double* Vt, int vstep, int m, int n, int n1) {
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    double* W = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int i, j, k, iter, max_iter = m > 30 ? m : 30;
    double c, s, sd;

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }
    for (iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = cv3_hypot(p, beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = 1;
                if (Vt) {
                    double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        double t0 = c * Vi[k] + s * Vj[k];
                        double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            double tw = W[i];
            W[i] = W[j];
            W[j] = tw;
            if (Vt) {
                for (k = 0; k < m; k++) {
                    double t = At[i * astep + k];
                    At[i * astep + k] = At[j * astep + k];
                    At[j * astep + k] = t;
                }
                for (k = 0; k < n; k++) {
                    double t = Vt[i * vstep + k];
                    Vt[i * vstep + k] = Vt[j * vstep + k];
                    Vt[j * vstep + k] = t;
                }
            }
        }
    }
    for (i = 0; i < n; i++) _W[i] = W[i];
    if (!Vt) {
        free(W);
        return;
    }
    cv3_rng rng;
    cv3_rng_init(&rng, 0x12345678);
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            /* zero singular value: random vector, project out the previous left vectors, normalise */
            const double val0 = 1. / m;
            for (k = 0; k < m; k++) {
                double val = (cv3_rng_next(&rng) & 256) != 0 ? val0 : -val0;
                At[i * astep + k] = val;
            }
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) {
                        double t = At[i * astep + k] - sd * At[j * astep + k];
                        At[i * astep + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++) At[i * astep + k] *= asum;
                }
            }
            sd = 0;
            for (k = 0; k < m; k++) {
                double t = At[i * astep + k];
                sd += t * t;
            }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
    free(W);
}

/* cv::SVD::compute -> _SVDcompute: u is m x ucols (ucols = full ? m : min), vt is vrows x n */
void cv3_svd_compute(const double* src, int m0, int n0, double* w, double* u, double* vt, int full_uv) {
    int m = m0, n = n0, at = 0;
    if (m < n) {
        int t = m;
        m = n;
        n = t;
        at = 1;
    }
    const int urows = full_uv ? m : n;
    double* temp_a = (double*)calloc((size_t)urows * m, sizeof(double)); /* temp_u aliases temp_a; rows >= n are zero */
    double* temp_v = (double*)calloc((size_t)n * n, sizeof(double));
    double* temp_w = (double*)calloc((size_t)n, sizeof(double));
    if (!at) {
        for (int i = 0; i < m0; i++)
            for (int j = 0; j < n0; j++) temp_a[j * m + i] = src[i * n0 + j]; /* transpose(src, temp_a) */
    } else {
        memcpy(temp_a, src, sizeof(double) * (size_t)m0 * n0);
    }
    cv3_jacobi_svd(temp_a, m, temp_w, temp_v, n, m, n, urows);
    for (int i = 0; i < n; i++) w[i] = temp_w[i];
    if (!at) {
        if (u) /* transpose(temp_u (urows x m), u) -> m x urows */
            for (int i = 0; i < urows; i++)
                for (int j = 0; j < m; j++) u[j * urows + i] = temp_a[i * m + j];
        if (vt) memcpy(vt, temp_v, sizeof(double) * (size_t)n * n);
    } else {
        if (u) /* transpose(temp_v (n x n)) -> u is m0 x m0 (m0 == n) */
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) u[j * n + i] = temp_v[i * n + j];
        if (vt) memcpy(vt, temp_a, sizeof(double) * (size_t)urows * m); /* temp_u: urows x n0 */
    }
    free(temp_a);
    free(temp_v);
    free(temp_w);
}

/* ------------------------------------------------------------------ Jacobi eigen (lapack.cpp JacobiImpl_) */
void cv3_jacobi_eigen(double* A, int n, double* W, double* V) {
    const double eps = DBL_EPSILON;
    int i, j, k, m;
    const int astep = n, vstep = n;
    for (i = 0; i < n; i++) {
        for (j = 0; j < n; j++) V[i * vstep + j] = 0;
        V[i * vstep + i] = 1;
    }
    int iters, maxIters = n * n * 30;
    int* indR = (int*)malloc(sizeof(int) * 2 * (size_t)n);
    int* indC = indR + n;
    double mv = 0;
    for (k = 0; k < n; k++) {
        W[k] = A[(astep + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = fabs(A[astep * k + m]), i = k + 2; i < n; i++) {
                double val = fabs(A[astep * k + i]);
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(A[k]), i = 1; i < k; i++) {
                double val = fabs(A[astep * i + k]);
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    if (n > 1)
        for (iters = 0; iters < maxIters; iters++) {
            for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
                double val = fabs(A[astep * i + indR[i]]);
                if (mv < val) mv = val, k = i;
            }
            int l = indR[k];
            for (i = 1; i < n; i++) {
                double val = fabs(A[astep * indC[i] + i]);
                if (mv < val) mv = val, k = indC[i], l = i;
            }
            double p = A[astep * k + l];
            if (fabs(p) <= eps) break;
            double y = (W[l] - W[k]) * 0.5;
            double t = fabs(y) + cv3_hypot(p, y);
            double s = cv3_hypot(p, t);
            double c = t / s;
            s = p / s;
            t = (p / t) * p;
            if (y < 0) s = -s, t = -t;
            A[astep * k + l] = 0;
            W[k] -= t;
            W[l] += t;
            double a0, b0;
#define CV3_ROTATE(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
            for (i = 0; i < k; i++) CV3_ROTATE(A[astep * i + k], A[astep * i + l]);
            for (i = k + 1; i < l; i++) CV3_ROTATE(A[astep * k + i], A[astep * i + l]);
            for (i = l + 1; i < n; i++) CV3_ROTATE(A[astep * k + i], A[astep * l + i]);
            for (i = 0; i < n; i++) CV3_ROTATE(V[vstep * k + i], V[vstep * l + i]);
#undef CV3_ROTATE
            for (j = 0; j < 2; j++) {
                int idx = j == 0 ? k : l;
                if (idx < n - 1) {
                    for (m = idx + 1, mv = fabs(A[astep * idx + m]), i = idx + 2; i < n; i++) {
                        double val = fabs(A[astep * idx + i]);
                        if (mv < val) mv = val, m = i;
                    }
                    indR[idx] = m;
                }
                if (idx > 0) {
                    for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; i++) {
                        double val = fabs(A[astep * i + idx]);
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
    free(indR);
}

/* ------------------------------------------------------------------ LU (lapack.cpp LUImpl) */
int cv3_lu(double* A, int astep, int m, double* b, int bstep, int n) {
    const double eps = DBL_EPSILON * 100;
    int i, j, k, p = 1;
    for (i = 0; i < m; i++) {
        k = i;
        for (j = i + 1; j < m; j++)
            if (fabs(A[j * astep + i]) > fabs(A[k * astep + i])) k = j;
        if (fabs(A[k * astep + i]) < eps) return 0;
        if (k != i) {
            for (j = i; j < m; j++) {
                double t = A[i * astep + j];
                A[i * astep + j] = A[k * astep + j];
                A[k * astep + j] = t;
            }
            if (b)
                for (j = 0; j < n; j++) {
                    double t = b[i * bstep + j];
                    b[i * bstep + j] = b[k * bstep + j];
                    b[k * bstep + j] = t;
                }
            p = -p;
        }
        double d = -1 / A[i * astep + i];
        for (j = i + 1; j < m; j++) {
            double alpha = A[j * astep + i] * d;
            for (k = i + 1; k < m; k++) A[j * astep + k] += alpha * A[i * astep + k];
            if (b)
                for (k = 0; k < n; k++) b[j * bstep + k] += alpha * b[i * bstep + k];
        }
    }
    if (b) {
        for (i = m - 1; i >= 0; i--)
            for (j = 0; j < n; j++) {
                double s = b[i * bstep + j];
                for (k = i + 1; k < m; k++) s -= A[i * astep + k] * b[k * bstep + j];
                b[i * bstep + j] = s / A[i * astep + i];
            }
    }
    return p;
}

int cv3_invert_lu(const double* src, int n, double* dst) {
    double* a = (double*)malloc(sizeof(double) * (size_t)n * n);
    memcpy(a, src, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) dst[i * n + j] = i == j ? 1.0 : 0.0;
    int r = cv3_lu(a, n, n, dst, n, n);
    free(a);
    if (!r) memset(dst, 0, sizeof(double) * (size_t)n * n);
    return r != 0;
}

/* SVBkSbImpl_ specialised to what solve/invert(DECOMP_EIG) pass: u = v = eigenvector rows (uT=vT=true) */
static void svbksb_eig(int n, const double* w, const double* v, const double* b, int nb, double* x) {
    const double eps = DBL_EPSILON * 2;
    double threshold = 0;
    int i, j, k;
    for (i = 0; i < n; i++)
        for (j = 0; j < nb; j++) x[i * nb + j] = 0;
    for (i = 0; i < n; i++) threshold += w[i];
    threshold *= eps;
    for (i = 0; i < n; i++) {
        const double* ui = v + i * n; /* row i of the eigenvector matrix */
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        if (nb == 1) {
            double s = 0;
            if (b)
                for (j = 0; j < n; j++) s += ui[j] * b[j];
            else
                s = ui[0];
            s *= wi;
            for (j = 0; j < n; j++) x[j] = x[j] + s * ui[j];
        } else {
            /* b == NULL: right-hand side is the identity (invert) -> buffer[j] = u[j] * wi, x += v (x) buffer */
            for (j = 0; j < nb; j++) {
                double s = b ? 0 : ui[j];
                if (b)
                    for (k = 0; k < n; k++) s += ui[k] * b[k * nb + j];
                s *= wi;
                for (k = 0; k < n; k++) x[k * nb + j] = x[k * nb + j] + s * ui[k];
            }
        }
    }
}

void cv3_solve_eig(const double* A, int n, const double* b, double* x) {
    double* a = (double*)malloc(sizeof(double) * (size_t)(2 * n * n + n));
    double* v = a + n * n;
    double* w = v + n * n;
    memcpy(a, A, sizeof(double) * (size_t)n * n);
    cv3_jacobi_eigen(a, n, w, v);
    svbksb_eig(n, w, v, b, 1, x);
    free(a);
}

void cv3_invert_eig(const double* A, int n, double* dst) {
    double* a = (double*)malloc(sizeof(double) * (size_t)(2 * n * n + n));
    double* v = a + n * n;
    double* w = v + n * n;
    memcpy(a, A, sizeof(double) * (size_t)n * n);
    cv3_jacobi_eigen(a, n, w, v);
    svbksb_eig(n, w, v, NULL, n, dst);
    free(a);
}

/* ------------------------------------------------------------------ solvePoly (mathfuncs.cpp), Durand-Kerner */
void cv3_solve_poly(const double* coeffs, int n0, double* rre, double* rim, int maxIters) {
    int n = n0, iter, i, j;
    double cre[32], cim[32];
    for (i = 0; i <= n0; i++) {
        cre[i] = coeffs[i];
        cim[i] = 0;
    }
    for (; n > 1; n--)
        if (fabs(cre[n]) + fabs(cim[n]) > DBL_EPSILON) break;
    double pre = 1, pim = 0;
    const double rr = 1, ri = 1;
    for (i = 0; i < n; i++) {
        rre[i] = pre;
        rim[i] = pim;
        double tre = pre * rr - pim * ri, tim = pre * ri + pim * rr;
        pre = tre;
        pim = tim;
    }
    maxIters = maxIters <= 0 ? 1000 : maxIters;
    for (iter = 0; iter < maxIters; iter++) {
        double maxDiff = 0;
        for (i = 0; i < n; i++) {
            pre = rre[i];
            pim = rim[i];
            double nre = cre[n], nim = cim[n], dre = cre[n], dim = cim[n];
            for (j = 0; j < n; j++) {
                /* num = num*p + coeffs[n-j-1] */
                double tre = nre * pre - nim * pim, tim = nre * pim + nim * pre;
                nre = tre + cre[n - j - 1];
                nim = tim + cim[n - j - 1];
                if (j != i) {
                    double qre = pre - rre[j], qim = pim - rim[j];
                    tre = dre * qre - dim * qim;
                    tim = dre * qim + dim * qre;
                    dre = tre;
                    dim = tim;
                }
            }
            /* num /= denom */
            double t = 1. / (dre * dre + dim * dim);
            double qre = (nre * dre + nim * dim) * t, qim = (-nre * dim + nim * dre) * t;
            nre = qre;
            nim = qim;
            rre[i] = pre - nre;
            rim[i] = pim - nim;
            double an = sqrt(nre * nre + nim * nim);
            maxDiff = maxDiff > an ? maxDiff : an;
        }
        if (maxDiff <= 0) break;
    }
    for (i = 0; i < n; i++)
        if (fabs(rim[i]) < 1e-100) rim[i] = 0;
    for (; n < n0; n++) {
        rre[n] = rre[n - 1]; /* roots[n+1] = roots[n] on the 1-based buffer of the original */
        rim[n] = rim[n - 1];
    }
}
