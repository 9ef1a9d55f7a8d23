/* ORACLE -- see cv3_calib3d.h / cv3_core.h.  Restates OpenCV 3.4.3 modules/calib3d/src/
 * {ptsetreg.cpp, five-point.cpp, fundam.cpp, triangulate.cpp, levmarq.cpp}.
 * Compile with -ffp-contract=off (OpenCV's generic x86-64 build has no FMA contraction). */
#include "cv3_calib3d.h"
#include "cv3_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================
 * ptsetreg.cpp
 * ====================================================================================== */
int cv3_ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = p > 0. ? p : 0.;
    p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.;
    ep = ep < 1. ? ep : 1.;
    double num = (1. - p) > DBL_MIN ? (1. - p) : DBL_MIN;
    double denom = 1. - pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cv3_round(num / denom);
}

static int ransac_get_subset(const cv3_ransac_cb* cb, const char* m1, const char* m2, int count, char* ms1, char* ms2,
                             cv3_rng* rng, int maxAttempts) {
    int idx[16];
    int i = 0, j, iters = 0;
    const int modelPoints = cb->model_points;
    for (; iters < maxAttempts; iters++) {
        for (i = 0; i < modelPoints && iters < maxAttempts;) {
            int idx_i = 0;
            for (;;) {
                idx_i = idx[i] = cv3_rng_uniform_int(rng, 0, count);
                for (j = 0; j < i; j++)
                    if (idx_i == idx[j]) break;
                if (j == i) break;
            }
            memcpy(ms1 + (size_t)i * cb->esz1, m1 + (size_t)idx_i * cb->esz1, cb->esz1);
            memcpy(ms2 + (size_t)i * cb->esz2, m2 + (size_t)idx_i * cb->esz2, cb->esz2);
            i++;
        }
        if (i == modelPoints && cb->check_subset && !cb->check_subset(cb->ctx, ms1, ms2, i)) continue;
        break;
    }
    return i == modelPoints && iters < maxAttempts;
}

static int ransac_find_inliers(const cv3_ransac_cb* cb, const void* m1, const void* m2, int n, const double* model,
                               float* err, unsigned char* mask, double thresh) {
    cb->compute_error(cb->ctx, m1, m2, n, model, err);
    const float t = (float)(thresh * thresh);
    int nz = 0;
    for (int i = 0; i < n; i++) {
        int f = err[i] <= t;
        mask[i] = (unsigned char)f;
        nz += f;
    }
    return nz;
}

/* RANSACPointSetRegistrator::run; returns 1 on success.  stats (optional): [0] iterations run,
 * [1] winning iteration, [2] winning model index */
int cv3_ransac_run(const cv3_ransac_cb* cb, const void* m1, const void* m2, int count, double threshold,
                      double confidence, int maxIters, double* model_out, unsigned char* mask_out, int* stats) {
    const int modelPoints = cb->model_points;
    int iter, niters = maxIters > 1 ? maxIters : 1;
    int maxGoodCount = 0;
    cv3_rng rng;
    cv3_rng_init(&rng, (uint64_t)-1);
    if (stats) stats[0] = stats[1] = stats[2] = -1;
    if (count < modelPoints) return 0;
    double models[10 * 16];
    double* bestModel = (double*)malloc(sizeof(double) * (size_t)cb->model_size);
    if (count == modelPoints) {
        int nm = cb->run_kernel(cb->ctx, m1, m2, count, models);
        if (nm <= 0) {
            free(bestModel);
            return 0;
        }
        memcpy(model_out, models, sizeof(double) * (size_t)cb->model_size);
        memset(mask_out, 1, (size_t)count);
        free(bestModel);
        return 1;
    }
    float* err = (float*)malloc(sizeof(float) * (size_t)count);
    unsigned char* mask = (unsigned char*)malloc((size_t)count);
    unsigned char* bestMask = (unsigned char*)malloc((size_t)count);
    char* ms1 = (char*)malloc(cb->esz1 * 16);
    char* ms2 = (char*)malloc(cb->esz2 * 16);
    int result = 0, early = 0;
    for (iter = 0; iter < niters; iter++) {
        int i, nmodels;
        int found = ransac_get_subset(cb, (const char*)m1, (const char*)m2, count, ms1, ms2, &rng, 10000);
        if (!found) {
            if (iter == 0) early = 1;
            break;
        }
        nmodels = cb->run_kernel(cb->ctx, ms1, ms2, modelPoints, models);
        if (nmodels <= 0) continue;
        for (i = 0; i < nmodels; i++) {
            const double* model_i = models + (size_t)i * cb->model_size;
            int goodCount = ransac_find_inliers(cb, m1, m2, count, model_i, err, mask, threshold);
            if (goodCount > (maxGoodCount > modelPoints - 1 ? maxGoodCount : modelPoints - 1)) {
                unsigned char* t = mask;
                mask = bestMask;
                bestMask = t;
                memcpy(bestModel, model_i, sizeof(double) * (size_t)cb->model_size);
                maxGoodCount = goodCount;
                niters = cv3_ransac_update_num_iters(confidence, (double)(count - goodCount) / count, modelPoints, niters);
                if (stats) {
                    stats[1] = iter;
                    stats[2] = i;
                }
            }
        }
    }
    if (stats) stats[0] = iter;
    if (!early && maxGoodCount > 0) {
        memcpy(mask_out, bestMask, (size_t)count);
        memcpy(model_out, bestModel, sizeof(double) * (size_t)cb->model_size);
        result = 1;
    }
    free(err);
    free(mask);
    free(bestMask);
    free(ms1);
    free(ms2);
    free(bestModel);
    return result;
}

/* ======================================================================================
 * five-point.cpp
 * ====================================================================================== */
/* monomial bookkeeping for the 10 cubic constraints on E = x X + y Y + z Z + W */
static const int kLinExp[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
static const int kQuadExp[10][3] = {{2, 0, 0}, {0, 2, 0}, {0, 0, 2}, {1, 1, 0}, {1, 0, 1},
                                    {0, 1, 1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
/* column order of the 10x20 coefficient matrix: the ten leading monomials are eliminated, the
 * last ten are [x z^2, x z, x, y z^2, y z, y, z^3, z^2, z, 1] (Nister's ordering) */
static const int kCubExp[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1},
                                   {0, 2, 0}, {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2},
                                   {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
static int g_qidx[4][4], g_cidx[10][4], g_tables_ready = 0;

static void build_tables(void) {
    if (g_tables_ready) return;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            int e[3] = {kLinExp[i][0] + kLinExp[j][0], kLinExp[i][1] + kLinExp[j][1], kLinExp[i][2] + kLinExp[j][2]};
            for (int q = 0; q < 10; q++)
                if (kQuadExp[q][0] == e[0] && kQuadExp[q][1] == e[1] && kQuadExp[q][2] == e[2]) g_qidx[i][j] = q;
        }
    for (int q = 0; q < 10; q++)
        for (int j = 0; j < 4; j++) {
            int e[3] = {kQuadExp[q][0] + kLinExp[j][0], kQuadExp[q][1] + kLinExp[j][1], kQuadExp[q][2] + kLinExp[j][2]};
            for (int c = 0; c < 20; c++)
                if (kCubExp[c][0] == e[0] && kCubExp[c][1] == e[1] && kCubExp[c][2] == e[2]) g_cidx[q][j] = c;
        }
    g_tables_ready = 1;
}

/* out(quad) += a(lin) * b(lin) */
static void lin_mul_acc(const double* a, const double* b, double* out) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[g_qidx[i][j]] = out[g_qidx[i][j]] + a[i] * b[j];
}
/* out(cubic) += sign * q(quad) * l(lin) */
static void quad_mul_acc(const double* q, const double* l, double sign, double* out) {
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 4; j++) {
            const double p = q[i] * l[j];
            out[g_cidx[i][j]] = sign > 0 ? out[g_cidx[i][j]] + p : out[g_cidx[i][j]] - p;
        }
}

/* getCoeffMat: e = 4 null-space vectors (each 9), A = 10 x 20 row-major */
static void five_point_coeff_mat(const double* e, double* A) {
    build_tables();
    double L[3][3][4]; /* E_ij as a linear polynomial [x, y, z, 1] */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            for (int v = 0; v < 4; v++) L[i][j][v] = e[v * 9 + i * 3 + j];
    memset(A, 0, sizeof(double) * 200);
    /* row 0: det(E) */
    {
        double m0[10] = {0}, m1[10] = {0}, m2[10] = {0}, t[10];
        /* m0 = E11 E22 - E12 E21 */
        memset(t, 0, sizeof(t));
        lin_mul_acc(L[1][1], L[2][2], m0);
        lin_mul_acc(L[1][2], L[2][1], t);
        for (int k = 0; k < 10; k++) m0[k] = m0[k] - t[k];
        /* m1 = E10 E22 - E12 E20 */
        memset(t, 0, sizeof(t));
        lin_mul_acc(L[1][0], L[2][2], m1);
        lin_mul_acc(L[1][2], L[2][0], t);
        for (int k = 0; k < 10; k++) m1[k] = m1[k] - t[k];
        /* m2 = E10 E21 - E11 E20 */
        memset(t, 0, sizeof(t));
        lin_mul_acc(L[1][0], L[2][1], m2);
        lin_mul_acc(L[1][1], L[2][0], t);
        for (int k = 0; k < 10; k++) m2[k] = m2[k] - t[k];
        quad_mul_acc(m0, L[0][0], 1.0, A);
        quad_mul_acc(m1, L[0][1], -1.0, A);
        quad_mul_acc(m2, L[0][2], 1.0, A);
    }
    /* rows 1..9: 2 E E^T E - trace(E E^T) E */
    double EEt[3][3][10];
    memset(EEt, 0, sizeof(EEt));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) lin_mul_acc(L[i][k], L[j][k], EEt[i][j]);
    double tr[10];
    for (int k = 0; k < 10; k++) tr[k] = (EEt[0][0][k] + EEt[1][1][k]) + EEt[2][2][k];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double* row = A + (size_t)(1 + i * 3 + j) * 20;
            double acc[20];
            memset(acc, 0, sizeof(acc));
            for (int k = 0; k < 3; k++) quad_mul_acc(EEt[i][k], L[k][j], 1.0, acc);
            for (int c = 0; c < 20; c++) acc[c] = 2.0 * acc[c];
            quad_mul_acc(tr, L[i][j], -1.0, acc);
            memcpy(row, acc, sizeof(acc));
        }
}

/* ascending-power polynomial product, out[na+nb-1] */
static void poly_mul(const double* a, int na, const double* b, int nb, double* out) {
    for (int i = 0; i < na + nb - 1; i++) out[i] = 0;
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++) out[i + j] = out[i + j] + a[i] * b[j];
}

static double norm_l2_9(const double* v) {
    double s = 0;
    int i = 0;
    for (; i <= 9 - 4; i += 4) s += v[i] * v[i] + v[i + 1] * v[i + 1] + v[i + 2] * v[i + 2] + v[i + 3] * v[i + 3];
    for (; i < 9; i++) s += v[i] * v[i];
    return sqrt(s);
}

int cv3_five_point(const double* q1, const double* q2, double* E_out) {
    double Q[5 * 9];
    for (int i = 0; i < 5; i++) {
        const double x1 = q1[i * 2], y1 = q1[i * 2 + 1], x2 = q2[i * 2], y2 = q2[i * 2 + 1];
        double* r = Q + i * 9;
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
    double W[5], Vt[81];
    cv3_svd_compute(Q, 5, 9, W, NULL, Vt, 1);
    const double* EE = Vt + 5 * 9; /* 4 null-space vectors, 9 each */
    double A[200];
    five_point_coeff_mat(EE, A);
    /* A = A(:,0:10)^-1 * A(:,10:20) */
    double A1[100], A1inv[100], A2[100], Ar[100];
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            A1[i * 10 + j] = A[i * 20 + j];
            A2[i * 10 + j] = A[i * 20 + 10 + j];
        }
    if (!cv3_invert_lu(A1, 10, A1inv)) return 0;
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            double s = 0;
            for (int k = 0; k < 10; k++) s += A1inv[i * 10 + k] * A2[k * 10 + j];
            Ar[i * 10 + j] = s;
        }
    double b[3 * 13];
    for (int i = 0; i < 3; i++) {
        const double* r1 = Ar + (i * 2 + 4) * 10;
        const double* r2 = Ar + (i * 2 + 5) * 10;
        double row1[13] = {0}, row2[13] = {0};
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
    /* determinant of the 3x3 polynomial matrix (entries in ascending powers of z) */
    double p[3][3][5];
    for (int j = 0; j < 3; j++) {
        const double* br = b + j * 13;
        for (int k = 0; k < 4; k++) {
            p[j][0][k] = br[3 - k];
            p[j][1][k] = br[7 - k];
        }
        p[j][0][4] = p[j][1][4] = 0;
        for (int k = 0; k < 5; k++) p[j][2][k] = br[12 - k];
    }
    double c[11], t1[8], t2[8], m[8], pr[11];
    for (int k = 0; k < 11; k++) c[k] = 0;
    /* + p00 (p11 p22 - p12 p21) */
    poly_mul(p[1][1], 4, p[2][2], 5, t1);
    poly_mul(p[1][2], 5, p[2][1], 4, t2);
    for (int k = 0; k < 8; k++) m[k] = t1[k] - t2[k];
    poly_mul(p[0][0], 4, m, 8, pr);
    for (int k = 0; k < 11; k++) c[k] = c[k] + pr[k];
    /* - p01 (p10 p22 - p12 p20) */
    poly_mul(p[1][0], 4, p[2][2], 5, t1);
    poly_mul(p[1][2], 5, p[2][0], 4, t2);
    for (int k = 0; k < 8; k++) m[k] = t1[k] - t2[k];
    poly_mul(p[0][1], 4, m, 8, pr);
    for (int k = 0; k < 11; k++) c[k] = c[k] - pr[k];
    /* + p02 (p10 p21 - p11 p20) */
    poly_mul(p[1][0], 4, p[2][1], 4, t1);
    poly_mul(p[1][1], 4, p[2][0], 4, t2);
    for (int k = 0; k < 7; k++) m[k] = t1[k] - t2[k];
    poly_mul(p[0][2], 5, m, 7, pr);
    for (int k = 0; k < 11; k++) c[k] = c[k] + pr[k];

    double rre[10], rim[10];
    cv3_solve_poly(c, 10, rre, rim, 300);
    int count = 0;
    for (int i = 0; i < 10; i++) {
        if (fabs(rim[i]) > 1e-10) continue;
        const double z1 = rre[i];
        const double z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
        for (int j = 0; j < 3; j++) {
            const double* br = b + j * 13;
            bz[j * 3 + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        double w3[3], vt3[9];
        cv3_svd_compute(bz, 3, 3, w3, NULL, vt3, 0); /* SVD::solveZ: last row of vt */
        const double* xy1 = vt3 + 6;
        if (fabs(xy1[2]) < 1e-10) continue;
        const double xs = xy1[0] / xy1[2], ys = xy1[1] / xy1[2], zs = z1;
        double* Ev = E_out + count * 9;
        for (int k = 0; k < 9; k++) {
            const double t = EE[k] * xs + EE[9 + k] * ys;
            const double u = t + EE[18 + k] * zs;
            Ev[k] = u + EE[27 + k];
        }
        const double inv = 1. / norm_l2_9(Ev);
        for (int k = 0; k < 9; k++) Ev[k] = Ev[k] * inv;
        count++;
    }
    return count;
}

static int em_run_kernel(const void* ctx, const void* ms1, const void* ms2, int count, double* models) {
    (void)ctx;
    (void)count;
    return cv3_five_point((const double*)ms1, (const double*)ms2, models);
}

static void em_compute_error(const void* ctx, const void* m1, const void* m2, int n, const double* E, float* err) {
    (void)ctx;
    const double* x1p = (const double*)m1;
    const double* x2p = (const double*)m2;
    for (int i = 0; i < n; i++) {
        const double x1[3] = {x1p[i * 2], x1p[i * 2 + 1], 1.};
        const double x2[3] = {x2p[i * 2], x2p[i * 2 + 1], 1.};
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
        err[i] = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
    }
}

static void normalise_points(const double* pts, int n, double f, double cx, double cy, double* out) {
    /* MatExpr (col - c) / f evaluates as col*(1/f) + (-c*(1/f)) through convertTo */
    const double a = 1. / f, bx = -cx * a, by = -cy * a;
    for (int i = 0; i < n; i++) {
        out[i * 2] = pts[i * 2] * a + bx;
        out[i * 2 + 1] = pts[i * 2 + 1] * a + by;
    }
}

int cv3_find_essential_mat_ex(const double* pts1, const double* pts2, int n, double focal, double ppx, double ppy,
                              double prob, double threshold, int max_iters, double* E, unsigned char* mask,
                              int* iters_run, int* best_iter, int* best_model) {
    double* p1 = (double*)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    double* p2 = (double*)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    normalise_points(pts1, n, focal, ppx, ppy, p1);
    normalise_points(pts2, n, focal, ppx, ppy, p2);
    threshold /= (focal + focal) / 2;
    cv3_ransac_cb cb;
    memset(&cb, 0, sizeof(cb));
    cb.model_points = 5;
    cb.model_size = 9;
    cb.run_kernel = em_run_kernel;
    cb.compute_error = em_compute_error;
    cb.check_subset = NULL;
    cb.esz1 = cb.esz2 = 2 * sizeof(double);
    int stats[3];
    int r = cv3_ransac_run(&cb, p1, p2, n, threshold, prob, max_iters, E, mask, stats);
    if (iters_run) *iters_run = stats[0];
    if (best_iter) *best_iter = stats[1];
    if (best_model) *best_model = stats[2];
    free(p1);
    free(p2);
    return r;
}

int cv3_find_essential_mat(const double* pts1, const double* pts2, int n, double focal, double ppx, double ppy,
                           double prob, double threshold, int max_iters, double* E, unsigned char* mask) {
    return cv3_find_essential_mat_ex(pts1, pts2, n, focal, ppx, ppy, prob, threshold, max_iters, E, mask, NULL, NULL,
                                     NULL);
}

void cv3_decompose_essential_mat(const double* E, double* R1, double* R2, double* t) {
    double D[3], U[9], Vt[9];
    cv3_svd_compute(E, 3, 3, D, U, Vt, 0);
    if (cv3_det3(U) < 0)
        for (int i = 0; i < 9; i++) U[i] *= -1.;
    if (cv3_det3(Vt) < 0)
        for (int i = 0; i < 9; i++) Vt[i] *= -1.;
    const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    const double Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double UW[9];
    cv3_mul33(U, Wm, UW);
    cv3_mul33(UW, Vt, R1);
    cv3_mul33(U, Wt, UW);
    cv3_mul33(UW, Vt, R2);
    t[0] = U[2];
    t[1] = U[5];
    t[2] = U[8];
}

void cv3_triangulate_points(const double* P1, const double* P2, const double* x1, const double* x2, int n,
                            double* X4) {
    const double* P[2] = {P1, P2};
    const double* xs[2] = {x1, x2};
    for (int i = 0; i < n; i++) {
        double A[16], w[4], vt[16];
        for (int j = 0; j < 2; j++) {
            const double x = xs[j][i], y = xs[j][n + i];
            for (int k = 0; k < 4; k++) {
                A[(j * 2 + 0) * 4 + k] = x * P[j][2 * 4 + k] - P[j][0 * 4 + k];
                A[(j * 2 + 1) * 4 + k] = y * P[j][2 * 4 + k] - P[j][1 * 4 + k];
            }
        }
        cv3_svd_compute(A, 4, 4, w, NULL, vt, 0);
        for (int k = 0; k < 4; k++) X4[k * n + i] = vt[12 + k];
    }
}

static int cheirality_count(const double* P, const double* x1, const double* x2, int n, double dist,
                            unsigned char* mask) {
    static const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double* Q = (double*)malloc(sizeof(double) * 4 * (size_t)n);
    cv3_triangulate_points(P0, P, x1, x2, n, Q);
    int good = 0;
    for (int i = 0; i < n; i++) {
        double X = Q[i], Y = Q[n + i], Z = Q[2 * n + i], Wq = Q[3 * n + i];
        int m = (Z * Wq) > 0;
        X = Wq != 0 ? X / Wq : 0;
        Y = Wq != 0 ? Y / Wq : 0;
        Z = Wq != 0 ? Z / Wq : 0;
        Wq = Wq != 0 ? Wq / Wq : 0;
        m = (Z < dist) & m;
        double z2 = 0;
        z2 += P[8] * X;
        z2 += P[9] * Y;
        z2 += P[10] * Z;
        z2 += P[11] * Wq;
        m = (z2 > 0) & m;
        m = (z2 < dist) & m;
        mask[i] = m ? 255 : 0;
        good += m;
    }
    free(Q);
    return good;
}

int cv3_recover_pose(const double* E, const double* pts1, const double* pts2, int n, double focal, double ppx,
                     double ppy, double* R, double* t, unsigned char* mask_out) {
    double* p1 = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* p2 = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* x1 = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* x2 = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    normalise_points(pts1, n, focal, ppx, ppy, p1);
    normalise_points(pts2, n, focal, ppx, ppy, p2);
    for (int i = 0; i < n; i++) { /* points = points.t() -> 2 x N */
        x1[i] = p1[i * 2];
        x1[n + i] = p1[i * 2 + 1];
        x2[i] = p2[i * 2];
        x2[n + i] = p2[i * 2 + 1];
    }
    double R1[9], R2[9], tt[3];
    cv3_decompose_essential_mat(E, R1, R2, tt);
    double P[4][12];
    const double* Rs[4] = {R1, R2, R1, R2};
    const double sg[4] = {1, 1, -1, -1};
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) P[c][r * 4 + k] = Rs[c][r * 3 + k];
            P[c][r * 4 + 3] = sg[c] > 0 ? tt[r] : -tt[r];
        }
    unsigned char* masks[4];
    int good[4];
    for (int c = 0; c < 4; c++) {
        masks[c] = (unsigned char*)malloc((size_t)(n > 0 ? n : 1));
        good[c] = cheirality_count(P[c], x1, x2, n, 50.0, masks[c]);
    }
    int sel;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3])
        sel = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3])
        sel = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3])
        sel = 2;
    else
        sel = 3;
    memcpy(R, Rs[sel], sizeof(double) * 9);
    for (int k = 0; k < 3; k++) t[k] = sg[sel] > 0 ? tt[k] : -tt[k];
    if (mask_out) memcpy(mask_out, masks[sel], (size_t)n);
    const int ret = good[sel];
    for (int c = 0; c < 4; c++) free(masks[c]);
    free(p1);
    free(p2);
    free(x1);
    free(x2);
    return ret;
}

/* ======================================================================================
 * fundam.cpp: homography
 * ====================================================================================== */
static int have_collinear_points(const float* pts, int count) {
    int j, k, i = count - 1;
    for (j = 0; j < i; j++) {
        const double dx1 = pts[j * 2] - pts[i * 2];
        const double dy1 = pts[j * 2 + 1] - pts[i * 2 + 1];
        for (k = 0; k < j; k++) {
            const double dx2 = pts[k * 2] - pts[i * 2];
            const double dy2 = pts[k * 2 + 1] - pts[i * 2 + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 1;
        }
    }
    return 0;
}

static double matx_det3(const double* a) {
    return a[0] * (a[4] * a[8] - a[7] * a[5]) - a[1] * (a[3] * a[8] - a[6] * a[5]) + a[2] * (a[3] * a[7] - a[6] * a[4]);
}

static int h_check_subset(const void* ctx, const void* ms1, const void* ms2, int count) {
    (void)ctx;
    const float* src = (const float*)ms1;
    const float* dst = (const float*)ms2;
    if (have_collinear_points(src, count) || have_collinear_points(dst, count)) return 0;
    if (count == 4) {
        static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; i++) {
            const int* t = tt[i];
            const double A[9] = {src[t[0] * 2], src[t[0] * 2 + 1], 1., src[t[1] * 2], src[t[1] * 2 + 1], 1.,
                                 src[t[2] * 2], src[t[2] * 2 + 1], 1.};
            const double B[9] = {dst[t[0] * 2], dst[t[0] * 2 + 1], 1., dst[t[1] * 2], dst[t[1] * 2 + 1], 1.,
                                 dst[t[2] * 2], dst[t[2] * 2 + 1], 1.};
            negative += matx_det3(A) * matx_det3(B) < 0;
        }
        if (negative != 0 && negative != 4) return 0;
    }
    return 1;
}

static int h_run_kernel(const void* ctx, const void* ms1, const void* ms2, int count, double* model) {
    (void)ctx;
    const float* M = (const float*)ms1;
    const float* m = (const float*)ms2;
    double LtL[9][9], W[9], V[9][9];
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    int i;
    for (i = 0; i < count; i++) {
        cmx += m[i * 2];
        cmy += m[i * 2 + 1];
        cMx += M[i * 2];
        cMy += M[i * 2 + 1];
    }
    cmx /= count;
    cmy /= count;
    cMx /= count;
    cMy /= count;
    for (i = 0; i < count; i++) {
        smx += fabs(m[i * 2] - cmx);
        smy += fabs(m[i * 2 + 1] - cmy);
        sMx += fabs(M[i * 2] - cMx);
        sMy += fabs(M[i * 2 + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON)
        return 0;
    smx = count / smx;
    smy = count / smy;
    sMx = count / sMx;
    sMy = count / sMy;
    const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    memset(LtL, 0, sizeof(LtL));
    for (i = 0; i < count; i++) {
        const double x = (m[i * 2] - cmx) * smx, y = (m[i * 2 + 1] - cmy) * smy;
        const double X = (M[i * 2] - cMx) * sMx, Y = (M[i * 2 + 1] - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[j][k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++) /* completeSymm (upper -> lower) */
        for (int k = 0; k < j; k++) LtL[j][k] = LtL[k][j];
    cv3_jacobi_eigen(&LtL[0][0], 9, W, &V[0][0]);
    double Htemp[9], H0[9];
    cv3_mul33(invHnorm, V[8], Htemp);
    cv3_mul33(Htemp, Hnorm2, H0);
    const double s = 1. / H0[8];
    for (int k = 0; k < 9; k++) model[k] = H0[k] * s;
    return 1;
}

static void h_compute_error(const void* ctx, const void* m1, const void* m2, int n, const double* H, float* err) {
    (void)ctx;
    const float* M = (const float*)m1;
    const float* m = (const float*)m2;
    const float Hf[8] = {(float)H[0], (float)H[1], (float)H[2], (float)H[3], (float)H[4], (float)H[5], (float)H[6], (float)H[7]};
    for (int i = 0; i < n; i++) {
        const float ww = 1.f / (Hf[6] * M[i * 2] + Hf[7] * M[i * 2 + 1] + 1.f);
        const float dx = (Hf[0] * M[i * 2] + Hf[1] * M[i * 2 + 1] + Hf[2]) * ww - m[i * 2];
        const float dy = (Hf[3] * M[i * 2] + Hf[4] * M[i * 2 + 1] + Hf[5]) * ww - m[i * 2 + 1];
        err[i] = dx * dx + dy * dy;
    }
}

/* HomographyRefineCallback::compute */
static void h_refine_compute(const float* M, const float* m, int count, const double* h, double* err, double* J) {
    for (int i = 0; i < count; i++) {
        const double Mx = M[i * 2], My = M[i * 2 + 1];
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        err[i * 2] = xi - m[i * 2];
        err[i * 2 + 1] = yi - m[i * 2 + 1];
        if (J) {
            double* Jp = J + (size_t)i * 16;
            Jp[0] = Mx * ww;
            Jp[1] = My * ww;
            Jp[2] = ww;
            Jp[3] = Jp[4] = Jp[5] = 0.;
            Jp[6] = -Mx * ww * xi;
            Jp[7] = -My * ww * xi;
            Jp[8] = Jp[9] = Jp[10] = 0.;
            Jp[11] = Mx * ww;
            Jp[12] = My * ww;
            Jp[13] = ww;
            Jp[14] = -Mx * ww * yi;
            Jp[15] = -My * ww * yi;
        }
    }
}

static double norm_l2sqr(const double* a, int n) {
    double s = 0;
    int i = 0;
    for (; i <= n - 4; i += 4) {
        const double v0 = a[i], v1 = a[i + 1], v2 = a[i + 2], v3 = a[i + 3];
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    for (; i < n; i++) s += a[i] * a[i];
    return s;
}
static double norm_inf(const double* a, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s = s > fabs(a[i]) ? s : fabs(a[i]);
    return s;
}
static double dot_n(const double* a, const double* b, int n) {
    double r = 0;
    int i = 0;
    for (; i <= n - 4; i += 4) r += a[i] * b[i] + a[i + 1] * b[i + 1] + a[i + 2] * b[i + 2] + a[i + 3] * b[i + 3];
    for (; i < n; i++) r += a[i] * b[i];
    return r;
}
/* mulTransposed(J, A, true): A = J^T J (lx x lx), sequential over the rows of J */
static void jtj(const double* J, int rows, int lx, double* A) {
    for (int i = 0; i < lx; i++)
        for (int j = i; j < lx; j++) {
            double s = 0;
            for (int k = 0; k < rows; k++) s += J[(size_t)k * lx + i] * J[(size_t)k * lx + j];
            A[i * lx + j] = s;
            A[j * lx + i] = s;
        }
}
static void jtr(const double* J, const double* r, int rows, int lx, double* v) {
    for (int i = 0; i < lx; i++) {
        double s = 0;
        for (int k = 0; k < rows; k++) s += J[(size_t)k * lx + i] * r[k];
        v[i] = s;
    }
}

/* LMSolverImpl::run specialised to the homography refinement (levmarq.cpp), maxIters = 10 */
static int h_refine_lm(const float* M, const float* m, int count, double* h8, int maxIters) {
    const int lx = 8, rows = count * 2;
    const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    double x[8], xd[8], d[8], v[8], A[64], Ap[64], D[8], temp_d[8];
    double* r = (double*)malloc(sizeof(double) * (size_t)rows);
    double* rd = (double*)malloc(sizeof(double) * (size_t)rows);
    double* J = (double*)malloc(sizeof(double) * (size_t)rows * lx);
    memcpy(x, h8, sizeof(x));
    h_refine_compute(M, m, count, x, r, J);
    double S = norm_l2sqr(r, rows);
    jtj(J, rows, lx, A);
    jtr(J, r, rows, lx, v);
    for (int i = 0; i < lx; i++) D[i] = A[i * lx + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int i, iter = 0;
    for (;;) {
        memcpy(Ap, A, sizeof(A));
        for (i = 0; i < lx; i++) Ap[i * lx + i] += lambda * D[i];
        cv3_solve_eig(Ap, lx, v, d);
        for (i = 0; i < lx; i++) xd[i] = x[i] - d[i];
        h_refine_compute(M, m, count, xd, rd, NULL);
        const double Sd = norm_l2sqr(rd, rows);
        for (i = 0; i < lx; i++) { /* gemm(A, d, -1, v, 2, temp_d) */
            double s = 0;
            for (int k = 0; k < lx; k++) s += A[i * lx + k] * d[k];
            temp_d[i] = s * -1. + v[i] * 2.;
        }
        const double dS = dot_n(d, temp_d, lx);
        const double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            const double t = dot_n(d, v, lx);
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = nu > 2. ? nu : 2.;
            nu = nu < 10. ? nu : 10.;
            if (lambda == 0) {
                cv3_invert_eig(A, lx, Ap);
                double maxval = DBL_EPSILON;
                for (i = 0; i < lx; i++) maxval = maxval > fabs(Ap[i * lx + i]) ? maxval : fabs(Ap[i * lx + i]);
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            double tx[8];
            memcpy(tx, x, sizeof(tx));
            memcpy(x, xd, sizeof(x));
            memcpy(xd, tx, sizeof(tx));
            h_refine_compute(M, m, count, x, r, J);
            jtj(J, rows, lx, A);
            jtr(J, r, rows, lx, v);
        }
        iter++;
        const int proceed = iter < maxIters && norm_inf(d, lx) >= epsx && norm_inf(r, rows) >= epsf;
        if (!proceed) break;
    }
    memcpy(h8, x, sizeof(x));
    free(r);
    free(rd);
    free(J);
    return iter;
}

/* cv::findHomography(src, dst, method = 0): the normalised DLT over ALL points, then (n > 4) the Levenberg-Marquardt
 * refinement -- what cvFindHomography does for cvFindExtrinsicCameraParams2's planar initialisation (calibration.cpp).
 * The points are converted to CV_32F first, as findHomography does.  Returns 0 (H zeroed, as the C wrapper leaves it) when
 * the kernel rejects the point set. */
int cv3_find_homography_lsq(const double* pts1, const double* pts2, int n, double* H) {
    float* src = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    float* dst = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    for (int i = 0; i < 2 * n; i++) {
        src[i] = (float)pts1[i];
        dst[i] = (float)pts2[i];
    }
    int ok = n >= 4 && h_run_kernel(NULL, src, dst, n, H) > 0;
    if (ok && n > 4) h_refine_lm(src, dst, n, H, 10);
    if (!ok)
        for (int i = 0; i < 9; i++) H[i] = 0;
    free(src);
    free(dst);
    return ok;
}

int cv3_find_homography(const double* pts1, const double* pts2, int n, double ransac_thr, int max_iters,
                        double confidence, double* H, unsigned char* mask) {
    if (n < 4) return 0;
    float* src = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    float* dst = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    for (int i = 0; i < 2 * n; i++) {
        src[i] = (float)pts1[i];
        dst[i] = (float)pts2[i];
    }
    if (ransac_thr <= 0) ransac_thr = 3;
    cv3_ransac_cb cb;
    memset(&cb, 0, sizeof(cb));
    cb.model_points = 4;
    cb.model_size = 9;
    cb.run_kernel = h_run_kernel;
    cb.compute_error = h_compute_error;
    cb.check_subset = h_check_subset;
    cb.esz1 = cb.esz2 = 2 * sizeof(float);
    unsigned char* tmask = (unsigned char*)malloc((size_t)n);
    int result;
    if (n == 4) {
        memset(tmask, 1, (size_t)n);
        result = h_run_kernel(NULL, src, dst, n, H) > 0;
    } else {
        result = cv3_ransac_run(&cb, src, dst, n, ransac_thr, confidence, max_iters, H, tmask, NULL);
    }
    if (result && n > 4) {
        int np = 0;
        for (int i = 0; i < n; i++)
            if (tmask[i]) {
                src[np * 2] = src[i * 2];
                src[np * 2 + 1] = src[i * 2 + 1];
                dst[np * 2] = dst[i * 2];
                dst[np * 2 + 1] = dst[i * 2 + 1];
                np++;
            }
        if (np > 0) {
            h_run_kernel(NULL, src, dst, np, H);
            h_refine_lm(src, dst, np, H, 10);
        }
    }
    if (result)
        memcpy(mask, tmask, (size_t)n);
    else
        memset(mask, 0, (size_t)n);
    free(src);
    free(dst);
    free(tmask);
    return result;
}
