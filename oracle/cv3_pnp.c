/* ORACLE (test infrastructure only) -- see cv3_core.h for the status of this restatement (PARITY UNPINNED:
 * OpenCV itself is not available here).
 *
 * cv2.solvePnPRansac / cv2.Rodrigues as called by /root/reference/libs/tracker/pnp_tracker.py:98-116,
 * restating OpenCV 3.4.3 modules/calib3d/src/{solvepnp.cpp (solvePnPRansac, PnPRansacCallback, solvePnP),
 * epnp.cpp, calibration.cpp (cvRodrigues2, cvProjectPoints2, cvFindExtrinsicCameraParams2), compat_ptsetreg.cpp
 * (CvLevMarq), undistort.cpp (cvUndistortPoints, no distortion)} and modules/core/src/{lapack.cpp (solve/invert
 * DECOMP_SVD, SVBkSb), matmul.cpp (mulTransposed, small gemm), stat.cpp (norm, mean)} for the argument pattern
 * DF-VO uses: float64 points in (converted to float32 by solvePnPRansac), distCoeffs = None,
 * useExtrinsicGuess = false, flags = SOLVEPNP_ITERATIVE, confidence = 0.99.
 *
 * Two deliberate, documented deviations, both so that the HIP path can be bit-identical to this file:
 *  - sin / cos / acos (cvRodrigues2) are evaluated by the det_* functions below (fdlibm kernels, pure IEEE
 *    double arithmetic, < 1 ulp from libm) instead of the platform libm;
 *  - CvLevMarq::step's lambda = exp(lambdaLg10 * log(10.)) is read from a table of the 33 possible values as
 *    glibc evaluates them (tests/test_oracle_pnp.py re-derives the table with math.exp).
 * (Round 3: cvFindExtrinsicCameraParams2's planar initialisation -- object points coplanar, W[2]/W[1] < 1e-3 -- is
 * restated as well, see cv3_find_extrinsic_guess; rounds 1-2 started the LM from the RANSAC model there.)
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "cv3_internal.h"

/* ====================================================================================== deterministic libm */
static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

static double k_sin(double x, double y, int iy) {
    const double z = x * x, v = z * x, r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
static double k_cos(double x, double y) {
    const double z = x * x, w0 = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + (w0 * w0) * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
/* x = n*pi/2 + (y0 + y1), |y0| <= pi/4; two-term Cody-Waite (adequate for |x| < 1e5) */
static int rem_pio2(double x, double* y0, double* y1) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_1t = 6.07710050650619224932e-11;
    if (fabs(x) <= 0.78539816339744827900) {
        *y0 = x;
        *y1 = 0;
        return 0;
    }
    const double t = x * invpio2;
    const long long n = (long long)(t + (t >= 0 ? 0.5 : -0.5));
    const double fn = (double)n;
    const double r = x - fn * pio2_1, w = fn * pio2_1t;
    *y0 = r - w;
    *y1 = (r - *y0) - w;
    return (int)(n & 3);
}
double cv3_det_sin(double x) {
    double y0, y1;
    switch (rem_pio2(x, &y0, &y1)) {
        case 0: return k_sin(y0, y1, 1);
        case 1: return k_cos(y0, y1);
        case 2: return -k_sin(y0, y1, 1);
        default: return -k_cos(y0, y1);
    }
}
double cv3_det_cos(double x) {
    double y0, y1;
    switch (rem_pio2(x, &y0, &y1)) {
        case 0: return k_cos(y0, y1);
        case 1: return -k_sin(y0, y1, 1);
        case 2: return -k_cos(y0, y1);
        default: return k_sin(y0, y1, 1);
    }
}
double cv3_det_acos(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                 pi = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const double ax = fabs(x);
    double z, p, q, r, s, w;
    if (ax >= 1.0) {
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return (x - x) / (x - x);
    }
    if (ax < 0.5) {
        if (ax <= 6.938893903907228e-18) return pio2_hi + pio2_lo; /* 2^-57 */
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (x < 0) {
        z = (1.0 + x) * 0.5;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = sqrt(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    z = (1.0 - x) * 0.5;
    s = sqrt(z);
    uint64_t bits;
    double df = s;
    memcpy(&bits, &df, 8);
    bits &= 0xffffffff00000000ULL;
    memcpy(&df, &bits, 8);
    const double c = (z - df * df) / (s + df);
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    w = r * s + c;
    return 2.0 * (df + w);
}
/* exp(k * log(10.)) for k = -16 .. 16 as glibc evaluates it */
static const double kLambdaTable[33] = {
    0x1.cd2b297d889a0p-54, 0x1.203af9ee755f8p-50, 0x1.6849b86a12b93p-47, 0x1.c25c268497664p-44, 0x1.19799812dea04p-40,
    0x1.5fd7fe179648cp-37, 0x1.b7cdfd9d7bd9cp-34, 0x1.12e0be826d687p-30, 0x1.5798ee2308c2fp-27, 0x1.ad7f29abcaf44p-24,
    0x1.0c6f7a0b5ed87p-20, 0x1.4f8b588e368e5p-17, 0x1.a36e2eb1c4326p-14, 0x1.0624dd2f1a9f9p-10, 0x1.47ae147ae1478p-7,
    0x1.9999999999998p-4,  0x1.0000000000000p+0,  0x1.4000000000001p+3,  0x1.9000000000003p+6,  0x1.f400000000006p+9,
    0x1.3880000000005p+13, 0x1.86a000000000ep+16, 0x1.e84800000000bp+19, 0x1.312d000000003p+23, 0x1.7d7840000000cp+26,
    0x1.dcd6500000018p+29, 0x1.2a05f20000015p+33, 0x1.74876e800000ap+36, 0x1.d1a94a2000015p+39, 0x1.2309ce5400013p+43,
    0x1.6bcc41e900008p+46, 0x1.c6bf52634002fp+49, 0x1.1c37937e08011p+53};
double cv3_lm_lambda(int lambdaLg10) { return kLambdaTable[lambdaLg10 + 16]; }

/* ====================================================================================== lapack.cpp pieces */
void cv3_svbksb(int m, int n, const double* w, const double* u, int ldu, int uT, const double* v, int ldv, int vT,
                const double* b, int ldb, int nb, double* x, int ldx) {
    const double eps = DBL_EPSILON * 2;
    double threshold = 0;
    const int udelta0 = uT ? ldu : 1, udelta1 = uT ? 1 : ldu;
    const int vdelta0 = vT ? ldv : 1, vdelta1 = vT ? 1 : ldv;
    const int nm = m < n ? m : n;
    int i, j, k;
    if (!b) nb = m;
    double* buffer = (double*)malloc(sizeof(double) * (size_t)(nb > 0 ? nb : 1));
    for (i = 0; i < n; i++)
        for (j = 0; j < nb; j++) x[i * ldx + j] = 0;
    for (i = 0; i < nm; i++) threshold += w[i];
    threshold *= eps;
    for (i = 0; i < nm; i++, u += udelta0, v += vdelta0) {
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        if (nb == 1) {
            double s = 0;
            if (b)
                for (j = 0; j < m; j++) s += u[j * udelta1] * b[j * ldb];
            else
                s = u[0];
            s *= wi;
            for (j = 0; j < n; j++) x[j * ldx] = x[j * ldx] + s * v[j * vdelta1];
        } else {
            if (b) {
                for (j = 0; j < nb; j++) buffer[j] = 0;
                for (k = 0; k < m; k++) { /* MatrAXPY(m, nb, b, ldb, u, udelta1, buffer, 0) */
                    const double s = u[k * udelta1];
                    for (j = 0; j < nb; j++) buffer[j] = buffer[j] + s * b[k * ldb + j];
                }
                for (j = 0; j < nb; j++) buffer[j] *= wi;
            } else {
                for (j = 0; j < nb; j++) buffer[j] = u[j * udelta1] * wi;
            }
            for (k = 0; k < n; k++) { /* MatrAXPY(n, nb, buffer, 0, v, vdelta1, x, ldx) */
                const double s = v[k * vdelta1];
                for (j = 0; j < nb; j++) x[k * ldx + j] = x[k * ldx + j] + s * buffer[j];
            }
        }
    }
    free(buffer);
}

void cv3_solve_svd(const double* A, int m, int n, const double* b, double* x) {
    double* a = (double*)malloc(sizeof(double) * (size_t)(n * m + n * n + n));
    double *v = a + n * m, *w = v + n * n;
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) a[j * m + i] = A[i * n + j]; /* transpose(src, a) */
    cv3_jacobi_svd(a, m, w, v, n, m, n, n);
    cv3_svbksb(m, n, w, a, m, 1, v, n, 1, b, 1, 1, x, 1);
    free(a);
}

void cv3_invert_svd(const double* A, int n, double* dst) {
    double* u = (double*)malloc(sizeof(double) * (size_t)(2 * n * n + n));
    double *vt = u + n * n, *w = vt + n * n;
    cv3_svd_compute(A, n, n, w, u, vt, 0);
    cv3_svbksb(n, n, w, u, n, 0, vt, n, 1, NULL, 0, n, dst, n);
    free(u);
}

/* cvSVD(A, W, U, V, flags) for square n x n A: Ut = U^T (CV_SVD_U_T), Vt = V^T (CV_SVD_V_T); either may be NULL */
static void svd_square_t(const double* A, int n, double* w, double* Ut, double* Vt) {
    double* u = (double*)malloc(sizeof(double) * (size_t)(n * n));
    cv3_svd_compute(A, n, n, w, Ut ? u : NULL, Vt, 0);
    if (Ut)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Ut[j * n + i] = u[i * n + j];
    free(u);
}

/* mulTransposed(src (rows x cols), dst, aTa = true, delta (1 x cols or NULL)): every entry a sequential sum over rows */
static void mul_transposed_ata(const double* src, int rows, int cols, const double* delta, double* dst) {
    for (int i = 0; i < cols; i++)
        for (int j = i; j < cols; j++) {
            double s = 0;
            for (int k = 0; k < rows; k++) {
                const double a = delta ? src[k * cols + i] - delta[i] : src[k * cols + i];
                const double bb = delta ? src[k * cols + j] - delta[j] : src[k * cols + j];
                s += a * bb;
            }
            dst[i * cols + j] = s;
            dst[j * cols + i] = s;
        }
}

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* ====================================================================================== cvRodrigues2 */
/* J (optional): 3 x 9, J[i*9 + k] = dR[k] / dr[i] */
static void rodrigues_v2m(const double* rv, double* R, double* J) {
    double rx = rv[0], ry = rv[1], rz = rv[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < DBL_EPSILON) {
        memcpy(R, I, sizeof(I));
        if (J) {
            memset(J, 0, sizeof(double) * 27);
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    const double c = cv3_det_cos(theta), s = cv3_det_sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    rx *= itheta;
    ry *= itheta;
    rz *= itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        const double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0, 0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                                 0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            const double ri = i == 0 ? rx : i == 1 ? ry : rz;
            const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            const double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}
void cv3_rodrigues_v2m(const double* r, double* R) { rodrigues_v2m(r, R, NULL); }

void cv3_rodrigues_m2v(const double* Rin, double* rv) {
    double W[3], U[9], Vt[9], R[9];
    for (int k = 0; k < 9; k++)
        if (!(Rin[k] > -100 && Rin[k] < 100)) { /* checkRange */
            rv[0] = rv[1] = rv[2] = 0;
            return;
        }
    cv3_svd_compute(Rin, 3, 3, W, U, Vt, 0);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += U[i * 3 + k] * Vt[k * 3 + j];
            R[i * 3 + j] = s;
        }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = cv3_det_acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0)
            rx = ry = rz = 0;
        else {
            t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    rv[0] = rx;
    rv[1] = ry;
    rv[2] = rz;
}

/* ====================================================================================== cvProjectPoints2 (no distortion)
 * M: [n][3] doubles; m_out: [n][2]; dpdr / dpdt (optional): [2n][3] each */
static void project_points(const double* M, int n, const double* rvec, const double* tvec, const double* K, double* m_out,
                           double* dpdr, double* dpdt) {
    double R[9], dRdr[27];
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    rodrigues_v2m(rvec, R, dpdr ? dRdr : NULL);
    for (int i = 0; i < n; i++) {
        const double X = M[i * 3], Y = M[i * 3 + 1], Z = M[i * 3 + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        m_out[i * 2] = x * fx + cx;
        m_out[i * 2 + 1] = y * fy + cy;
        if (dpdt) {
            const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            for (int j = 0; j < 3; j++) {
                dpdt[(2 * i) * 3 + j] = fx * dxdt[j];
                dpdt[(2 * i + 1) * 3 + j] = fy * dydt[j];
            }
        }
        if (dpdr) {
            const double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                                     X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
            const double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                                     X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
            const double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                                     X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
            for (int j = 0; j < 3; j++) {
                const double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                const double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                dpdr[(2 * i) * 3 + j] = fx * dxdr;
                dpdr[(2 * i + 1) * 3 + j] = fy * dydr;
            }
        }
    }
}

/* ====================================================================================== epnp.cpp */
typedef struct {
    double uc, vc, fu, fv;
    int n;
    double *pws, *us, *alphas, *pcs;
    double cws[4][3], ccs[4][3];
} epnp_t;

static void epnp_choose_control_points(epnp_t* e) {
    const int n = e->n;
    e->cws[0][0] = e->cws[0][1] = e->cws[0][2] = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) e->cws[0][j] += e->pws[3 * i + j];
    for (int j = 0; j < 3; j++) e->cws[0][j] /= n;
    double* PW0 = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    double pw0tpw0[9], dc[3], uct[9];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) PW0[3 * i + j] = e->pws[3 * i + j] - e->cws[0][j];
    mul_transposed_ata(PW0, n, 3, NULL, pw0tpw0);
    svd_square_t(pw0tpw0, 3, dc, uct, NULL);
    free(PW0);
    for (int i = 1; i < 4; i++) {
        const double k = sqrt(dc[i - 1] / n);
        for (int j = 0; j < 3; j++) e->cws[i][j] = e->cws[0][j] + k * uct[3 * (i - 1) + j];
    }
}

static void epnp_compute_barycentric_coordinates(epnp_t* e) {
    double cc[9], cc_inv[9];
    for (int i = 0; i < 3; i++)
        for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e->cws[j][i] - e->cws[0][i];
    cv3_invert_svd(cc, 3, cc_inv);
    const double* ci = cc_inv;
    for (int i = 0; i < e->n; i++) {
        const double* pi = e->pws + 3 * i;
        double* a = e->alphas + 4 * i;
        for (int j = 0; j < 3; j++)
            a[1 + j] = ci[3 * j] * (pi[0] - e->cws[0][0]) + ci[3 * j + 1] * (pi[1] - e->cws[0][1]) +
                       ci[3 * j + 2] * (pi[2] - e->cws[0][2]);
        a[0] = 1.0f - a[1] - a[2] - a[3];
    }
}

static void epnp_fill_M(const epnp_t* e, double* M, int row, const double* as, double u, double v) {
    double* M1 = M + row * 12;
    double* M2 = M1 + 12;
    for (int i = 0; i < 4; i++) {
        M1[3 * i] = as[i] * e->fu;
        M1[3 * i + 1] = 0.0;
        M1[3 * i + 2] = as[i] * (e->uc - u);
        M2[3 * i] = 0.0;
        M2[3 * i + 1] = as[i] * e->fv;
        M2[3 * i + 2] = as[i] * (e->vc - v);
    }
}

static void epnp_compute_ccs(epnp_t* e, const double* betas, const double* ut) {
    for (int i = 0; i < 4; i++) e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0.0f;
    for (int i = 0; i < 4; i++) {
        const double* v = ut + 12 * (11 - i);
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) e->ccs[j][k] += betas[i] * v[3 * j + k];
    }
}

static void epnp_compute_pcs(epnp_t* e) {
    for (int i = 0; i < e->n; i++) {
        const double* a = e->alphas + 4 * i;
        double* pc = e->pcs + 3 * i;
        for (int j = 0; j < 3; j++)
            pc[j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] + a[3] * e->ccs[3][j];
    }
}

static void epnp_solve_for_sign(epnp_t* e) {
    if (e->pcs[2] < 0.0) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 3; j++) e->ccs[i][j] = -e->ccs[i][j];
        for (int i = 0; i < e->n; i++) {
            e->pcs[3 * i] = -e->pcs[3 * i];
            e->pcs[3 * i + 1] = -e->pcs[3 * i + 1];
            e->pcs[3 * i + 2] = -e->pcs[3 * i + 2];
        }
    }
}

static void epnp_estimate_R_and_t(epnp_t* e, double R[3][3], double t[3]) {
    const int n = e->n;
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double* pc = e->pcs + 3 * i;
        const double* pw = e->pws + 3 * i;
        for (int j = 0; j < 3; j++) {
            pc0[j] += pc[j];
            pw0[j] += pw[j];
        }
    }
    for (int j = 0; j < 3; j++) {
        pc0[j] /= n;
        pw0[j] /= n;
    }
    double abt[9] = {0}, abt_d[3], abt_u[9], abt_vt[9], abt_v[9];
    for (int i = 0; i < n; i++) {
        const double* pc = e->pcs + 3 * i;
        const double* pw = e->pws + 3 * i;
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    /* cvSVD(&ABt, &ABt_D, &ABt_U, &ABt_V, CV_SVD_MODIFY_A): U as is, V = transpose(vt) */
    cv3_svd_compute(abt, 3, 3, abt_d, abt_u, abt_vt, 0);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) abt_v[j * 3 + i] = abt_vt[i * 3 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i][j] = dot3(abt_u + 3 * i, abt_v + 3 * j);
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                       R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) {
        R[2][0] = -R[2][0];
        R[2][1] = -R[2][1];
        R[2][2] = -R[2][2];
    }
    t[0] = pc0[0] - dot3(R[0], pw0);
    t[1] = pc0[1] - dot3(R[1], pw0);
    t[2] = pc0[2] - dot3(R[2], pw0);
}

static double epnp_reprojection_error(const epnp_t* e, double R[3][3], const double t[3]) {
    double sum2 = 0.0;
    for (int i = 0; i < e->n; i++) {
        const double* pw = e->pws + 3 * i;
        const double Xc = dot3(R[0], pw) + t[0], Yc = dot3(R[1], pw) + t[1];
        const double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
        const double ue = e->uc + e->fu * Xc * inv_Zc, ve = e->vc + e->fv * Yc * inv_Zc;
        const double u = e->us[2 * i], v = e->us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / e->n;
}

static double epnp_compute_R_and_t(epnp_t* e, const double* ut, const double* betas, double R[3][3], double t[3]) {
    epnp_compute_ccs(e, betas, ut);
    epnp_compute_pcs(e);
    epnp_solve_for_sign(e);
    epnp_estimate_R_and_t(e, R, t);
    return epnp_reprojection_error(e, R, t);
}

static void epnp_find_betas_approx_1(const double* l_6x10, const double* rho, double* betas) {
    double l_6x4[24], b4[4];
    for (int i = 0; i < 6; i++) {
        l_6x4[i * 4 + 0] = l_6x10[i * 10 + 0];
        l_6x4[i * 4 + 1] = l_6x10[i * 10 + 1];
        l_6x4[i * 4 + 2] = l_6x10[i * 10 + 3];
        l_6x4[i * 4 + 3] = l_6x10[i * 10 + 6];
    }
    cv3_solve_svd(l_6x4, 6, 4, rho, b4);
    if (b4[0] < 0) {
        betas[0] = sqrt(-b4[0]);
        betas[1] = -b4[1] / betas[0];
        betas[2] = -b4[2] / betas[0];
        betas[3] = -b4[3] / betas[0];
    } else {
        betas[0] = sqrt(b4[0]);
        betas[1] = b4[1] / betas[0];
        betas[2] = b4[2] / betas[0];
        betas[3] = b4[3] / betas[0];
    }
}
static void epnp_find_betas_approx_2(const double* l_6x10, const double* rho, double* betas) {
    double l_6x3[18], b3[3];
    for (int i = 0; i < 6; i++) {
        l_6x3[i * 3 + 0] = l_6x10[i * 10 + 0];
        l_6x3[i * 3 + 1] = l_6x10[i * 10 + 1];
        l_6x3[i * 3 + 2] = l_6x10[i * 10 + 2];
    }
    cv3_solve_svd(l_6x3, 6, 3, rho, b3);
    if (b3[0] < 0) {
        betas[0] = sqrt(-b3[0]);
        betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0;
    } else {
        betas[0] = sqrt(b3[0]);
        betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0;
    }
    if (b3[1] < 0) betas[0] = -betas[0];
    betas[2] = 0.0;
    betas[3] = 0.0;
}
static void epnp_find_betas_approx_3(const double* l_6x10, const double* rho, double* betas) {
    double l_6x5[30], b5[5];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 5; j++) l_6x5[i * 5 + j] = l_6x10[i * 10 + j];
    cv3_solve_svd(l_6x5, 6, 5, rho, b5);
    if (b5[0] < 0) {
        betas[0] = sqrt(-b5[0]);
        betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0;
    } else {
        betas[0] = sqrt(b5[0]);
        betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0;
    }
    if (b5[1] < 0) betas[0] = -betas[0];
    betas[2] = b5[3] / betas[0];
    betas[3] = 0.0;
}

static void epnp_compute_L_6x10(const double* ut, double* l_6x10) {
    const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; j++) {
            dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
            dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
            dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
            b++;
            if (b > 3) {
                a++;
                b = a + 1;
            }
        }
    }
    for (int i = 0; i < 6; i++) {
        double* row = l_6x10 + 10 * i;
        row[0] = dot3(dv[0][i], dv[0][i]);
        row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
        row[2] = dot3(dv[1][i], dv[1][i]);
        row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
        row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
        row[5] = dot3(dv[2][i], dv[2][i]);
        row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
        row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
        row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
        row[9] = dot3(dv[3][i], dv[3][i]);
    }
}

static double dist2(const double* p1, const double* p2) {
    return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
}

static void epnp_compute_A_and_b_gauss_newton(const double* l_6x10, const double* rho, const double betas[4], double* A,
                                              double* b) {
    for (int i = 0; i < 6; i++) {
        const double* rowL = l_6x10 + i * 10;
        double* rowA = A + i * 4;
        rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
        rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
        rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
        rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
        b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] +
                         rowL[3] * betas[0] * betas[2] + rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                         rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] + rowL[8] * betas[2] * betas[3] +
                         rowL[9] * betas[3] * betas[3]);
    }
}

/* Householder QR of the nr x nc system (epnp::qr_solve), including its off-by-one pivot scan; X untouched when
 * a column is all zero (the caller zero-initialises it, where upstream leaves it indeterminate) */
static void epnp_qr_solve(double* pA, int nr, int nc, double* pb, double* pX) {
    double A1[8], A2[8];
    double* ppAkk = pA;
    for (int k = 0; k < nc; k++) {
        double *ppAik1 = ppAkk, eta = fabs(*ppAik1);
        for (int i = k + 1; i < nr; i++) {
            const double elt = fabs(*ppAik1);
            if (eta < elt) eta = elt;
            ppAik1 += nc;
        }
        if (eta == 0) {
            A1[k] = A2[k] = 0.0;
            return;
        } else {
            double *ppAik2 = ppAkk, sum2 = 0.0, inv_eta = 1. / eta;
            for (int i = k; i < nr; i++) {
                *ppAik2 *= inv_eta;
                sum2 += *ppAik2 * *ppAik2;
                ppAik2 += nc;
            }
            double sigma = sqrt(sum2);
            if (*ppAkk < 0) sigma = -sigma;
            *ppAkk += sigma;
            A1[k] = sigma * *ppAkk;
            A2[k] = -eta * sigma;
            for (int j = k + 1; j < nc; j++) {
                double *ppAik = ppAkk, sum = 0;
                for (int i = k; i < nr; i++) {
                    sum += *ppAik * ppAik[j - k];
                    ppAik += nc;
                }
                const double tau = sum / A1[k];
                ppAik = ppAkk;
                for (int i = k; i < nr; i++) {
                    ppAik[j - k] -= tau * *ppAik;
                    ppAik += nc;
                }
            }
        }
        ppAkk += nc + 1;
    }
    double* ppAjj = pA;
    for (int j = 0; j < nc; j++) {
        double *ppAij = ppAjj, tau = 0;
        for (int i = j; i < nr; i++) {
            tau += *ppAij * pb[i];
            ppAij += nc;
        }
        tau /= A1[j];
        ppAij = ppAjj;
        for (int i = j; i < nr; i++) {
            pb[i] -= tau * *ppAij;
            ppAij += nc;
        }
        ppAjj += nc + 1;
    }
    pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; i--) {
        double *ppAij = pA + i * nc + (i + 1), sum = 0;
        for (int j = i + 1; j < nc; j++) {
            sum += *ppAij * pX[j];
            ppAij++;
        }
        pX[i] = (pb[i] - sum) / A2[i];
    }
}

static void epnp_gauss_newton(const double* l_6x10, const double* rho, double betas[4]) {
    double a[24], b[6], x[4] = {0, 0, 0, 0};
    for (int k = 0; k < 5; k++) {
        epnp_compute_A_and_b_gauss_newton(l_6x10, rho, betas, a, b);
        epnp_qr_solve(a, 6, 4, b, x);
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

/* epnp::compute_pose.  pws [n][3], us [n][2] (pixel units: x*fu + uc of the undistorted points) */
void cv3_epnp(const double* K, const double* pws, const double* us, int n, double* R_out, double* t_out) {
    epnp_t e;
    e.fu = K[0];
    e.fv = K[4];
    e.uc = K[2];
    e.vc = K[5];
    e.n = n;
    e.pws = (double*)malloc(sizeof(double) * (size_t)n * 12);
    e.us = e.pws + 3 * n;
    e.alphas = e.us + 2 * n;
    e.pcs = e.alphas + 4 * n;
    memcpy(e.pws, pws, sizeof(double) * 3 * (size_t)n);
    memcpy(e.us, us, sizeof(double) * 2 * (size_t)n);
    epnp_choose_control_points(&e);
    epnp_compute_barycentric_coordinates(&e);
    double* M = (double*)malloc(sizeof(double) * 24 * (size_t)n);
    for (int i = 0; i < n; i++) epnp_fill_M(&e, M, 2 * i, e.alphas + 4 * i, e.us[2 * i], e.us[2 * i + 1]);
    double mtm[144], d[12], ut[144];
    mul_transposed_ata(M, 2 * n, 12, NULL, mtm);
    svd_square_t(mtm, 12, d, ut, NULL);
    free(M);
    double l_6x10[60], rho[6];
    epnp_compute_L_6x10(ut, l_6x10);
    rho[0] = dist2(e.cws[0], e.cws[1]);
    rho[1] = dist2(e.cws[0], e.cws[2]);
    rho[2] = dist2(e.cws[0], e.cws[3]);
    rho[3] = dist2(e.cws[1], e.cws[2]);
    rho[4] = dist2(e.cws[1], e.cws[3]);
    rho[5] = dist2(e.cws[2], e.cws[3]);
    double Betas[4][4], rep_errors[4], Rs[4][3][3], ts[4][3];
    epnp_find_betas_approx_1(l_6x10, rho, Betas[1]);
    epnp_gauss_newton(l_6x10, rho, Betas[1]);
    rep_errors[1] = epnp_compute_R_and_t(&e, ut, Betas[1], Rs[1], ts[1]);
    epnp_find_betas_approx_2(l_6x10, rho, Betas[2]);
    epnp_gauss_newton(l_6x10, rho, Betas[2]);
    rep_errors[2] = epnp_compute_R_and_t(&e, ut, Betas[2], Rs[2], ts[2]);
    epnp_find_betas_approx_3(l_6x10, rho, Betas[3]);
    epnp_gauss_newton(l_6x10, rho, Betas[3]);
    rep_errors[3] = epnp_compute_R_and_t(&e, ut, Betas[3], Rs[3], ts[3]);
    int N = 1;
    if (rep_errors[2] < rep_errors[1]) N = 2;
    if (rep_errors[3] < rep_errors[N]) N = 3;
    for (int i = 0; i < 3; i++) {
        t_out[i] = ts[N][i];
        for (int j = 0; j < 3; j++) R_out[i * 3 + j] = Rs[N][i][j];
    }
    free(e.pws);
}

/* ====================================================================================== solvePnP pieces */
typedef struct {
    double K[9];
} pnp_ctx;

/* solvePnP(EPNP) on float32 points: undistortPoints (float in, float out), epnp, Rodrigues */
void cv3_solve_pnp_epnp_f32(const double* K, const float* obj, const float* img, int n, double* rvec, double* tvec) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5], ifx = 1. / fx, ify = 1. / fy;
    double* pws = (double*)malloc(sizeof(double) * 5 * (size_t)n);
    double* us = pws + 3 * n;
    for (int i = 0; i < n; i++) {
        double x = img[i * 2], y = img[i * 2 + 1];
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        const float ux = (float)x, uy = (float)y; /* undistortedPoints is CV_32FC2 */
        pws[i * 3] = obj[i * 3];
        pws[i * 3 + 1] = obj[i * 3 + 1];
        pws[i * 3 + 2] = obj[i * 3 + 2];
        us[i * 2] = ux * fx + cx;
        us[i * 2 + 1] = uy * fy + cy;
    }
    double R[9];
    cv3_epnp(K, pws, us, n, R, tvec);
    cv3_rodrigues_m2v(R, rvec);
    free(pws);
}

static int pnp_run_kernel(const void* ctx, const void* ms1, const void* ms2, int count, double* model) {
    const pnp_ctx* c = (const pnp_ctx*)ctx;
    double rvec[3], tvec[3];
    cv3_solve_pnp_epnp_f32(c->K, (const float*)ms1, (const float*)ms2, count, rvec, tvec);
    for (int i = 0; i < 3; i++) { /* hconcat(rvec, tvec): 3 x 2 */
        model[i * 2] = rvec[i];
        model[i * 2 + 1] = tvec[i];
    }
    return 1;
}

static void pnp_compute_error(const void* ctx, const void* m1, const void* m2, int n, const double* model, float* err) {
    const pnp_ctx* c = (const pnp_ctx*)ctx;
    const float* op = (const float*)m1;
    const float* ip = (const float*)m2;
    const double rvec[3] = {model[0], model[2], model[4]}, tvec[3] = {model[1], model[3], model[5]};
    double* M = (double*)malloc(sizeof(double) * 5 * (size_t)n);
    double* proj = M + 3 * n;
    for (int i = 0; i < 3 * n; i++) M[i] = op[i];
    project_points(M, n, rvec, tvec, c->K, proj, NULL, NULL);
    for (int i = 0; i < n; i++) {
        const float px = (float)proj[i * 2], py = (float)proj[i * 2 + 1]; /* projpoints is CV_32F */
        const float dx = ip[i * 2] - px, dy = ip[i * 2 + 1] - py;
        float s = 0;
        s += dx * dx;
        s += dy * dy;
        err[i] = s;
    }
    free(M);
}

/* norm(a, NORM_L2) over n doubles: normL2Sqr_ accumulates four squares at a time */
static double norm_l2(const double* a, int n) {
    double s = 0;
    int i = 0;
    for (; i <= n - 4; i += 4) {
        const double v0 = a[i], v1 = a[i + 1], v2 = a[i + 2], v3 = a[i + 3];
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    for (; i < n; i++) {
        const double v = a[i];
        s += v * v;
    }
    return sqrt(s);
}
static double norm_diff_l2(const double* a, const double* b, int n) {
    double s = 0;
    int i = 0;
    for (; i <= n - 4; i += 4) {
        const double v0 = a[i] - b[i], v1 = a[i + 1] - b[i + 1], v2 = a[i + 2] - b[i + 2], v3 = a[i + 3] - b[i + 3];
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    for (; i < n; i++) {
        const double v = a[i] - b[i];
        s += v * v;
    }
    return sqrt(s);
}

/* CvLevMarq::step with all 6 parameters active: param = prevParam - solve(JtJ with diag *= 1 + lambda, JtErr, SVD) */
static void lm_step(const double* JtJ, const double* JtErr, int lambdaLg10, const double* prevParam, double* param) {
    const double lambda = cv3_lm_lambda(lambdaLg10);
    double A[36], d[6];
    memcpy(A, JtJ, sizeof(A));
    for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
    cv3_solve_svd(A, 6, 6, JtErr, d);
    for (int i = 0; i < 6; i++) param[i] = prevParam[i] - d[i];
}

/* cvFindExtrinsicCameraParams2(useExtrinsicGuess = false) on double points M [n][3], m [n][2].
 * Returns 1.  stats (optional): [0] LM iterations */
int cv3_find_extrinsic(const double* M, const double* m, int n, const double* K, double* rvec, double* tvec, int* stats) {
    return cv3_find_extrinsic_guess(M, m, n, K, NULL, rvec, tvec, stats);
}

/* as above.  Round 3: the planar initialisation is restated too (calibration.cpp, the `W[2]/W[1] < 1e-3` branch): object
 * points rotated into their principal plane (R_transform = V^T of the covariance's SVD, made right-handed; T_transform =
 * -R_transform Mc), cvFindHomography (method 0: normalised DLT over all points + LM) between the in-plane coordinates
 * and the normalised image points, H's first two columns normalised, third = their cross product, the rotation
 * orthonormalised by a Rodrigues round trip, t = H T_transform + h3 * 2 / (|h1| + |h2|), R = H R_transform.
 * `planar_guess` is kept for callers that want the old behaviour (start the LM from a given rvec | tvec instead); NULL
 * = OpenCV's initialisation. */
int cv3_find_extrinsic_guess(const double* M, const double* m, int n, const double* K, const double* planar_guess,
                             double* rvec, double* tvec, int* stats) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5], ifx = 1. / fx, ify = 1. / fy;
    double* mn = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    for (int i = 0; i < n; i++) { /* cvUndistortPoints, no distortion, R = P = identity */
        mn[i * 2] = (m[i * 2] - cx) * ifx;
        mn[i * 2 + 1] = (m[i * 2 + 1] - cy) * ify;
    }
    /* Mc = cvAvg(matM): channel sums times 1/n */
    double Mc[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) Mc[j] += M[i * 3 + j];
    const double inv_n = 1. / n;
    for (int j = 0; j < 3; j++) Mc[j] *= inv_n;
    double MM[9], W[3], V[9];
    mul_transposed_ata(M, n, 3, Mc, MM);
    svd_square_t(MM, 3, W, NULL, V);
    double param[6];
    const int planar = W[2] / W[1] < 1e-3 || n < 4;
    if (planar && !planar_guess) {
        /* R_transform = matV (CV_SVD_V_T: rows are the right singular vectors) */
        double Rt[9];
        memcpy(Rt, V, sizeof(Rt));
        if (V[2] * V[2] + V[5] * V[5] < 1e-10) {
            memset(Rt, 0, sizeof(Rt));
            Rt[0] = Rt[4] = Rt[8] = 1.;
        }
        if (cv3_det3(Rt) < 0)
            for (int i = 0; i < 9; i++) Rt[i] *= -1.;
        double Tt[3]; /* cvGEMM(R_transform, Mc, -1, 0, 0, T_transform, CV_GEMM_B_T) */
        for (int i = 0; i < 3; i++) Tt[i] = (Rt[i * 3] * Mc[0] + Rt[i * 3 + 1] * Mc[1] + Rt[i * 3 + 2] * Mc[2]) * -1.;
        double* Mxy = (double*)malloc(sizeof(double) * 2 * (size_t)n);
        for (int i = 0; i < n; i++) {
            const double* src = M + i * 3;
            Mxy[i * 2] = Rt[0] * src[0] + Rt[1] * src[1] + Rt[2] * src[2] + Tt[0];
            Mxy[i * 2 + 1] = Rt[3] * src[0] + Rt[4] * src[1] + Rt[5] * src[2] + Tt[1];
        }
        double h[9];
        cv3_find_homography_lsq(Mxy, mn, n, h);
        free(Mxy);
        int finite = 1;
        for (int i = 0; i < 9; i++) finite = finite && isfinite(h[i]);
        double R[9], tt[3] = {0, 0, 0};
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
            /* _h3 = _h1 x _h2 */
            h[2] = h[3] * h[7] - h[6] * h[4];
            h[5] = h[6] * h[1] - h[0] * h[7];
            h[8] = h[0] * h[4] - h[3] * h[1];
            double r3[3], Hr[9];
            cv3_rodrigues_m2v(h, r3);
            cv3_rodrigues_v2m(r3, Hr);
            for (int i = 0; i < 3; i++) /* cvMatMulAdd(H, T_transform, t, t) */
                tt[i] = (Hr[i * 3] * Tt[0] + Hr[i * 3 + 1] * Tt[1] + Hr[i * 3 + 2] * Tt[2]) + tt[i];
            for (int i = 0; i < 3; i++) /* cvMatMul(H, R_transform, R) */
                for (int j = 0; j < 3; j++) R[i * 3 + j] = Hr[i * 3] * Rt[j] + Hr[i * 3 + 1] * Rt[3 + j] + Hr[i * 3 + 2] * Rt[6 + j];
        } else {
            memset(R, 0, sizeof(R));
            R[0] = R[4] = R[8] = 1.;
        }
        cv3_rodrigues_m2v(R, param);
        for (int i = 0; i < 3; i++) param[3 + i] = tt[i];
    } else if (planar) {
        for (int i = 0; i < 6; i++) param[i] = planar_guess[i];
    } else {
        double* L = (double*)malloc(sizeof(double) * 24 * (size_t)n);
        for (int i = 0; i < n; i++) {
            double* Lr = L + i * 24;
            const double x = -mn[i * 2], y = -mn[i * 2 + 1];
            Lr[0] = Lr[16] = M[i * 3];
            Lr[1] = Lr[17] = M[i * 3 + 1];
            Lr[2] = Lr[18] = M[i * 3 + 2];
            Lr[3] = Lr[19] = 1.;
            Lr[4] = Lr[5] = Lr[6] = Lr[7] = 0.;
            Lr[12] = Lr[13] = Lr[14] = Lr[15] = 0.;
            Lr[8] = x * M[i * 3];
            Lr[9] = x * M[i * 3 + 1];
            Lr[10] = x * M[i * 3 + 2];
            Lr[11] = x;
            Lr[20] = y * M[i * 3];
            Lr[21] = y * M[i * 3 + 1];
            Lr[22] = y * M[i * 3 + 2];
            Lr[23] = y;
        }
        double LL[144], LW[12], LV[144];
        mul_transposed_ata(L, 2 * n, 12, NULL, LL);
        svd_square_t(LL, 12, LW, NULL, LV);
        free(L);
        double RRt[12];
        memcpy(RRt, LV + 11 * 12, sizeof(RRt));
        double RR[9], tt[3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) RR[i * 3 + j] = RRt[i * 4 + j];
            tt[i] = RRt[i * 4 + 3];
        }
        if (cv3_det3(RR) < 0) {
            for (int i = 0; i < 9; i++) RR[i] *= -1;
            for (int i = 0; i < 3; i++) tt[i] *= -1;
        }
        /* cvNorm(&_RR): _RR is a 3-column view of the 3 x 4 row, so the squares are accumulated row by row */
        double sc2 = 0;
        for (int i = 0; i < 3; i++) {
            double r = 0;
            for (int j = 0; j < 3; j++) r += RR[i * 3 + j] * RR[i * 3 + j];
            sc2 += r;
        }
        const double sc = sqrt(sc2);
        double Wr[3], Ut[9], Vt[9], R[9];
        svd_square_t(RR, 3, Wr, Ut, Vt);
        for (int i = 0; i < 3; i++) /* cvGEMM(U_T, V_T, CV_GEMM_A_T): R = U * Vt */
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += Ut[k * 3 + i] * Vt[k * 3 + j];
                R[i * 3 + j] = s;
            }
        const double scale = norm_l2(R, 9) / sc;
        for (int i = 0; i < 3; i++) param[3 + i] = tt[i] * scale;
        cv3_rodrigues_m2v(R, param);
    }
    /* CvLevMarq(6, 2n, TermCriteria(EPS + ITER, 20, FLT_EPSILON), completeSymmFlag = true) */
    double* J = (double*)malloc(sizeof(double) * 12 * (size_t)n);
    double* err = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* dpdr = (double*)malloc(sizeof(double) * 12 * (size_t)n);
    double* dpdt = dpdr + 6 * n;
    double JtJ[36], JtErr[6], prevParam[6];
    double prevErrNorm = DBL_MAX, errNorm = 0;
    int lambdaLg10 = -3, iters = 0;
    const int max_iter = 20;
    const double epsilon = FLT_EPSILON;
    enum { CALC_J, CHECK_ERR } state = CALC_J;
    for (;;) {
        /* caller side of solver.update(): residuals (and Jacobian) at the current parameters */
        if (state == CALC_J) {
            project_points(M, n, param, param + 3, K, err, dpdr, dpdt);
            for (int i = 0; i < 2 * n; i++)
                for (int j = 0; j < 3; j++) {
                    J[i * 6 + j] = dpdr[i * 3 + j];
                    J[i * 6 + 3 + j] = dpdt[i * 3 + j];
                }
        } else {
            project_points(M, n, param, param + 3, K, err, NULL, NULL);
        }
        for (int i = 0; i < 2 * n; i++) err[i] = err[i] - m[i];
        if (state == CALC_J) {
            mul_transposed_ata(J, 2 * n, 6, NULL, JtJ);
            for (int j = 0; j < 6; j++) { /* cvGEMM(J, err, 1, 0, 0, JtErr, CV_GEMM_A_T) */
                double s = 0;
                for (int k = 0; k < 2 * n; k++) s += J[k * 6 + j] * err[k];
                JtErr[j] = s;
            }
            memcpy(prevParam, param, sizeof(prevParam));
            lm_step(JtJ, JtErr, lambdaLg10, prevParam, param);
            if (iters == 0) prevErrNorm = norm_l2(err, 2 * n);
            state = CHECK_ERR;
            continue;
        }
        errNorm = norm_l2(err, 2 * n);
        if (errNorm > prevErrNorm) {
            if (++lambdaLg10 <= 16) {
                lm_step(JtJ, JtErr, lambdaLg10, prevParam, param);
                continue;
            }
        }
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        if (++iters >= max_iter || norm_diff_l2(param, prevParam, 6) / (norm_l2(prevParam, 6) + DBL_EPSILON) < epsilon) break;
        prevErrNorm = errNorm;
        state = CALC_J;
    }
    if (stats) stats[0] = iters;
    for (int i = 0; i < 3; i++) {
        rvec[i] = param[i];
        tvec[i] = param[3 + i];
    }
    free(J);
    free(err);
    free(dpdr);
    free(mn);
    return 1;
}

int cv3_solve_pnp_ransac(const double* obj, const double* img, int n, const double* K, int iterations,
                         double reproj_err, double confidence, double* rvec, double* tvec, int* inliers,
                         int* n_inliers) {
    *n_inliers = 0;
    rvec[0] = rvec[1] = rvec[2] = tvec[0] = tvec[1] = tvec[2] = 0;
    if (n < 4) return 0; /* CV_Assert(npoints >= 4) upstream */
    if (n == 4) return -3; /* SOLVEPNP_P3P kernel: not reachable from DF-VO (it requires more than 4 points) */
    float* opoints = (float*)malloc(sizeof(float) * 5 * (size_t)n);
    float* ipoints = opoints + 3 * n;
    for (int i = 0; i < 3 * n; i++) opoints[i] = (float)obj[i]; /* convertTo(CV_32F) */
    for (int i = 0; i < 2 * n; i++) ipoints[i] = (float)img[i];
    pnp_ctx ctx;
    memcpy(ctx.K, K, sizeof(ctx.K));
    cv3_ransac_cb cb;
    memset(&cb, 0, sizeof(cb));
    cb.model_points = 5;
    cb.model_size = 6;
    cb.run_kernel = pnp_run_kernel;
    cb.compute_error = pnp_compute_error;
    cb.check_subset = NULL;
    cb.esz1 = 3 * sizeof(float);
    cb.esz2 = 2 * sizeof(float);
    cb.ctx = &ctx;
    double model[6];
    unsigned char* mask = (unsigned char*)malloc((size_t)n);
    memset(mask, 0, (size_t)n);
    const int result = cv3_ransac_run(&cb, opoints, ipoints, n, reproj_err, confidence, iterations, model, mask, NULL);
    int rc = 0;
    if (result > 0) {
        double* oi = (double*)malloc(sizeof(double) * 5 * (size_t)n);
        double* ii = oi + 3 * n;
        int np = 0;
        for (int i = 0; i < n; i++)
            if (mask[i]) {
                for (int j = 0; j < 3; j++) oi[np * 3 + j] = opoints[i * 3 + j];
                for (int j = 0; j < 2; j++) ii[np * 2 + j] = ipoints[i * 2 + j];
                np++;
            }
        rc = cv3_find_extrinsic_guess(oi, ii, np, K, NULL, rvec, tvec, NULL); /* (planar inliers: OpenCV's homography initialisation) */
        if (rc == 1)
            for (int i = 0; i < n; i++)
                if (mask[i]) inliers[(*n_inliers)++] = i;
        free(oi);
    }
    free(mask);
    free(opoints);
    return rc;
}
