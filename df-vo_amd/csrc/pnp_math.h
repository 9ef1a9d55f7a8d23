// Per-lane f64 arithmetic of the PnP fallback (solvePnPRansac = 5-point EPnP hypotheses + reprojection
// scoring + DLT/Levenberg-Marquardt refinement on the inliers), host/device like solver_math.h.
//
// Replaces cv2.solvePnPRansac / cv2.Rodrigues at /root/reference/libs/tracker/pnp_tracker.py:98-116
// (OpenCV 3.4.3 solvepnp.cpp, epnp.cpp, calibration.cpp: cvRodrigues2 / cvProjectPoints2 /
// cvFindExtrinsicCameraParams2, compat_ptsetreg.cpp: CvLevMarq).  sin / cos / acos are evaluated with
// fdlibm-style kernels in plain IEEE arithmetic and the LM damping 10^k comes from a table, so that the
// device and the CPU oracle (oracle/cv3_pnp.c, which makes the same two choices) agree bit for bit.
// Dense work arrays are passed in by the caller (`ws`): on the GPU they are per-lane LDS slices.
#pragma once
#include "solver_math.h"

namespace sm {

// ------------------------------------------------------------------------------------------------
// deterministic sin / cos / acos
// ------------------------------------------------------------------------------------------------
SM_HD double pm_ksin(double x, double y, int iy) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, v = z * x, r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
SM_HD double pm_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x, w0 = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + (w0 * w0) * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
SM_HD int pm_rem_pio2(double x, double* y0, double* y1) {
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
SM_HD double det_sin(double x) {
    double y0, y1;
    const int q = pm_rem_pio2(x, &y0, &y1);
    const double s = pm_ksin(y0, y1, 1), c = pm_kcos(y0, y1);
    return q == 0 ? s : q == 1 ? c : q == 2 ? -s : -c;
}
SM_HD double det_cos(double x) {
    double y0, y1;
    const int q = pm_rem_pio2(x, &y0, &y1);
    const double s = pm_ksin(y0, y1, 1), c = pm_kcos(y0, y1);
    return q == 0 ? c : q == 1 ? -s : q == 2 ? -c : s;
}
SM_HD double pm_clear_low_word(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double(__double_as_longlong(v) & 0xffffffff00000000LL);
#else
    uint64_t b;
    __builtin_memcpy(&b, &v, 8);
    b &= 0xffffffff00000000ULL;
    __builtin_memcpy(&v, &b, 8);
    return v;
#endif
}
SM_HD double det_acos(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                 pi = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const double ax = fabs(x);
    if (ax >= 1.0) {
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return (x - x) / (x - x);
    }
    if (ax < 0.5) {
        if (ax <= 6.938893903907228e-18) return pio2_hi + pio2_lo;
        const double z = x * x;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (x < 0) {
        const double z = (1.0 + x) * 0.5;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double s = sqrt(z);
        const double r = p / q;
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    const double z = (1.0 - x) * 0.5;
    const double s = sqrt(z);
    const double df = pm_clear_low_word(s);
    const double c = (z - df * df) / (s + df);
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double r = p / q;
    const double w = r * s + c;
    return 2.0 * (df + w);
}
// exp(k * log(10.)), k = -16 .. 16, as glibc evaluates it (CvLevMarq::step)
SM_HD double lm_lambda(int lambdaLg10) {
    const double tbl[33] = {
        0x1.cd2b297d889a0p-54, 0x1.203af9ee755f8p-50, 0x1.6849b86a12b93p-47, 0x1.c25c268497664p-44, 0x1.19799812dea04p-40,
        0x1.5fd7fe179648cp-37, 0x1.b7cdfd9d7bd9cp-34, 0x1.12e0be826d687p-30, 0x1.5798ee2308c2fp-27, 0x1.ad7f29abcaf44p-24,
        0x1.0c6f7a0b5ed87p-20, 0x1.4f8b588e368e5p-17, 0x1.a36e2eb1c4326p-14, 0x1.0624dd2f1a9f9p-10, 0x1.47ae147ae1478p-7,
        0x1.9999999999998p-4,  0x1.0000000000000p+0,  0x1.4000000000001p+3,  0x1.9000000000003p+6,  0x1.f400000000006p+9,
        0x1.3880000000005p+13, 0x1.86a000000000ep+16, 0x1.e84800000000bp+19, 0x1.312d000000003p+23, 0x1.7d7840000000cp+26,
        0x1.dcd6500000018p+29, 0x1.2a05f20000015p+33, 0x1.74876e800000ap+36, 0x1.d1a94a2000015p+39, 0x1.2309ce5400013p+43,
        0x1.6bcc41e900008p+46, 0x1.c6bf52634002fp+49, 0x1.1c37937e08011p+53};
    return tbl[lambdaLg10 + 16];
}

// ------------------------------------------------------------------------------------------------
// SVD-based solve / invert (cv::solve / cv::invert with DECOMP_SVD) and the cvSVD flavours used here
// ------------------------------------------------------------------------------------------------
// x (n) = sum_i (u_i . b / w_i) v_i over the rows u_i (length m) of `u` and v_i (length n) of `v` (SVBkSb, nb = 1)
SM_HD void svbksb_vec(int m, int n, const double* w, const double* u, int ldu, const double* v, int ldv, const double* b,
                      double* x) {
    const double eps = DBL_EPSILON * 2;
    double threshold = 0;
    const int nm = m < n ? m : n;
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int i = 0; i < nm; i++) threshold += w[i];
    threshold *= eps;
    for (int i = 0; i < nm; i++) {
        const double* ui = u + i * ldu;
        const double* vi = v + i * ldv;
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < m; j++) s += ui[j] * b[j];
        s *= wi;
        for (int j = 0; j < n; j++) x[j] = x[j] + s * vi[j];
    }
}
// cv::solve(A (m x n, m >= n), b, DECOMP_SVD); ws: n*m + n*n + n doubles
SM_HD void solve_svd(const double* A, int m, int n, const double* b, double* x, double* ws) {
    double *a = ws, *v = ws + n * m, *w = v + n * n;
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) a[j * m + i] = A[i * n + j];
    jacobi_svd_impl(a, m, w, v, n, m, n, n);
    svbksb_vec(m, n, w, a, m, v, n, b, x);
}
// cvSVD(A (N x N), W, U^T, V^T): ut = rows of left singular vectors, vt = rows of right ones
template <int N>
SM_HD void svd_square_t(const double* A, double* w, double* ut, double* vt) {
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) ut[j * N + i] = A[i * N + j];
    jacobi_svd_impl(ut, N, w, vt, N, N, N, N);
}
// cv::invert(A (3 x 3), DECOMP_SVD); ws: 21 doubles
SM_HD void invert_svd3(const double* A, double* dst, double* ws) {
    double *ut = ws, *vt = ws + 9, *w = ws + 18;
    svd_square_t<3>(A, w, ut, vt);
    const double eps = DBL_EPSILON * 2;
    double threshold = 0;
    for (int i = 0; i < 9; i++) dst[i] = 0;
    for (int i = 0; i < 3; i++) threshold += w[i];
    threshold *= eps;
    for (int i = 0; i < 3; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double buffer[3];
        for (int j = 0; j < 3; j++) buffer[j] = ut[i * 3 + j] * wi;  // column i of u
        for (int k = 0; k < 3; k++) {
            const double s = vt[i * 3 + k];
            for (int j = 0; j < 3; j++) dst[k * 3 + j] = dst[k * 3 + j] + s * buffer[j];
        }
    }
}
// mulTransposed(src (rows x cols), aTa): upper triangle entry (i, j) as one sequential sum over the rows
SM_HD void mul_transposed_ata(const double* src, int rows, int cols, double* dst) {
    for (int i = 0; i < cols; i++)
        for (int j = i; j < cols; j++) {
            double s = 0;
            for (int k = 0; k < rows; k++) s += src[k * cols + i] * src[k * cols + j];
            dst[i * cols + j] = s;
            dst[j * cols + i] = s;
        }
}
SM_HD double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// ------------------------------------------------------------------------------------------------
// cvRodrigues2
// ------------------------------------------------------------------------------------------------
// J (optional, 27 doubles): J[i*9 + k] = dR[k] / dr[i]
SM_HD void rodrigues_v2m(const double* rv, double* R, double* J) {
    double rx = rv[0], ry = rv[1], rz = rv[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; k++) R[k] = I[k];
        if (J) {
            for (int k = 0; k < 27; k++) J[k] = 0;
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    const double c = det_cos(theta), s = det_sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
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
// ws: 21 doubles
SM_HD void rodrigues_m2v(const double* Rin, double* rv, double* ws) {
    for (int k = 0; k < 9; k++)
        if (!(Rin[k] > -100 && Rin[k] < 100)) {
            rv[0] = rv[1] = rv[2] = 0;
            return;
        }
    double *ut = ws, *vt = ws + 9, *W = ws + 18;
    svd_square_t<3>(Rin, W, ut, vt);
    double R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += ut[k * 3 + i] * vt[k * 3 + j];  // U(i,k) = ut(k,i)
            R[i * 3 + j] = s;
        }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = det_acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0) {
            rx = ry = rz = 0;
        } else {
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

// ------------------------------------------------------------------------------------------------
// cvProjectPoints2 without distortion: one point.  K4 = fx, fy, cx, cy.  jr / jt (optional): the two
// Jacobian rows [du/dr | dv/dr] and [du/dt | dv/dt] as 2 x 3 each (row-major), dRdr from rodrigues_v2m
// ------------------------------------------------------------------------------------------------
SM_HD void project_point(const double* R, const double* t, const double* K4, double X, double Y, double Z, double* u,
                         double* v, const double* dRdr, double* jr, double* jt) {
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    *u = x * K4[0] + K4[2];
    *v = y * K4[1] + K4[3];
    if (jt) {
        const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
        for (int j = 0; j < 3; j++) {
            jt[j] = K4[0] * dxdt[j];
            jt[3 + j] = K4[1] * dydt[j];
        }
    }
    if (jr) {
        const double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                                 X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
        const double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                                 X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
        const double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                                 X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
        for (int j = 0; j < 3; j++) {
            const double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
            const double dydr = z * (dy0dr[j] - y * dz0dr[j]);
            jr[j] = K4[0] * dxdr;
            jr[3 + j] = K4[1] * dydr;
        }
    }
}
// PnPRansacCallback::computeError of one correspondence (float points, float error)
SM_HD float pnp_error(const double* R, const double* t, const double* K4, const float* obj, const float* img) {
    double u, v;
    project_point(R, t, K4, obj[0], obj[1], obj[2], &u, &v, nullptr, nullptr, nullptr);
    const float px = (float)u, py = (float)v;
    const float dx = img[0] - px, dy = img[1] - py;
    float s = 0;
    s += dx * dx;
    s += dy * dy;
    return s;
}

// ------------------------------------------------------------------------------------------------
// EPnP on 5 correspondences (epnp.cpp), the RANSAC kernel of solvePnPRansac
// ------------------------------------------------------------------------------------------------
constexpr int EPNP_N = 5;
// ws layout (EPNP_WS doubles): [0,120) M (10 x 12), later scratch of the small solves; [120,264) MtM -> U^T rows;
// [264,408) V^T; [408,420) D; [420,480) L_6x10; [480,540) pws us alphas pcs; [540,564) cws ccs
constexpr int EPNP_WS = 564;

SM_HD void epnp_qr_solve(double* pA, int nr, int nc, double* pb, double* pX) {
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

SM_HD void epnp_gauss_newton(const double* l_6x10, const double* rho, double* betas, double* scratch /*34*/) {
    double *a = scratch, *b = scratch + 24, *x = scratch + 30;
    for (int i = 0; i < 4; i++) x[i] = 0;
    for (int k = 0; k < 5; k++) {
        for (int i = 0; i < 6; i++) {
            const double* rowL = l_6x10 + i * 10;
            double* rowA = a + i * 4;
            rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
            rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
            rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
            rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
            b[i] = rho[i] -
                   (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] +
                    rowL[3] * betas[0] * betas[2] + rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                    rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] + rowL[8] * betas[2] * betas[3] +
                    rowL[9] * betas[3] * betas[3]);
        }
        epnp_qr_solve(a, 6, 4, b, x);
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

// R (9) and t (3) of one beta hypothesis; returns the mean reprojection error
SM_HD double epnp_compute_R_and_t(const double* K4, const double* ut, const double* betas, const double* pws,
                                  const double* us, const double* alphas, double* pcs, double* ccs, double* R, double* t,
                                  double* scratch /*30*/) {
    const int n = EPNP_N;
    for (int i = 0; i < 12; i++) ccs[i] = 0.0f;
    for (int i = 0; i < 4; i++) {
        const double* v = ut + 12 * (11 - i);
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) ccs[j * 3 + k] += betas[i] * v[3 * j + k];
    }
    for (int i = 0; i < n; i++) {
        const double* a = alphas + 4 * i;
        double* pc = pcs + 3 * i;
        for (int j = 0; j < 3; j++) pc[j] = a[0] * ccs[j] + a[1] * ccs[3 + j] + a[2] * ccs[6 + j] + a[3] * ccs[9 + j];
    }
    if (pcs[2] < 0.0) {
        for (int i = 0; i < 12; i++) ccs[i] = -ccs[i];
        for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];
    }
    // estimate_R_and_t
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) {
            pc0[j] += pcs[3 * i + j];
            pw0[j] += pws[3 * i + j];
        }
    for (int j = 0; j < 3; j++) {
        pc0[j] /= n;
        pw0[j] /= n;
    }
    double *abt = scratch, *abt_ut = scratch + 9, *abt_vt = scratch + 18, *abt_d = scratch + 27;
    for (int i = 0; i < 9; i++) abt[i] = 0;
    for (int i = 0; i < n; i++) {
        const double* pc = pcs + 3 * i;
        const double* pw = pws + 3 * i;
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    svd_square_t<3>(abt, abt_d, abt_ut, abt_vt);
    // R[i][j] = dot(row i of U, row j of V) = sum_k U(i,k) V(j,k) = sum_k ut(k,i) vt(k,j)
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = abt_ut[i] * abt_vt[j] + abt_ut[3 + i] * abt_vt[3 + j] + abt_ut[6 + i] * abt_vt[6 + j];
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] -
                       R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) {
        R[6] = -R[6];
        R[7] = -R[7];
        R[8] = -R[8];
    }
    t[0] = pc0[0] - dot3(R, pw0);
    t[1] = pc0[1] - dot3(R + 3, pw0);
    t[2] = pc0[2] - dot3(R + 6, pw0);
    // reprojection_error
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) {
        const double* pw = pws + 3 * i;
        const double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1];
        const double inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
        const double ue = K4[2] + K4[0] * Xc * inv_Zc, ve = K4[3] + K4[1] * Yc * inv_Zc;
        const double u = us[2 * i], v = us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

// solvePnP(EPNP) on 5 float correspondences -> rvec, tvec (PnPRansacCallback::runKernel)
SM_HD void epnp_kernel(const double* K4, const float* obj /*15*/, const float* img /*10*/, double* rvec, double* tvec,
                       double* ws) {
    const int n = EPNP_N;
    double *M = ws, *mtm_ut = ws + 120, *vt = ws + 264, *d = ws + 408, *l_6x10 = ws + 420;
    double *pws = ws + 480, *us = ws + 495, *alphas = ws + 505, *pcs = ws + 525, *cws = ws + 540, *ccs = ws + 552;
    const double fu = K4[0], fv = K4[1], uc = K4[2], vc = K4[3], ifx = 1. / fu, ify = 1. / fv;
    for (int i = 0; i < n; i++) {
        double x = img[i * 2], y = img[i * 2 + 1];
        x = (x - uc) * ifx;
        y = (y - vc) * ify;
        const float ux = (float)x, uy = (float)y;  // undistortPoints output is float32
        pws[i * 3] = obj[i * 3];
        pws[i * 3 + 1] = obj[i * 3 + 1];
        pws[i * 3 + 2] = obj[i * 3 + 2];
        us[i * 2] = ux * fu + uc;
        us[i * 2 + 1] = uy * fv + vc;
    }
    // choose_control_points
    cws[0] = cws[1] = cws[2] = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) cws[j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[j] /= n;
    {
        double *PW0 = M, *pw0tpw0 = M + 15, *dc = M + 24, *uct = M + 27, *vtmp = M + 36;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) PW0[3 * i + j] = pws[3 * i + j] - cws[j];
        mul_transposed_ata(PW0, n, 3, pw0tpw0);
        svd_square_t<3>(pw0tpw0, dc, uct, vtmp);
        for (int i = 1; i < 4; i++) {
            const double k = sqrt(dc[i - 1] / n);
            for (int j = 0; j < 3; j++) cws[i * 3 + j] = cws[j] + k * uct[3 * (i - 1) + j];
        }
    }
    // compute_barycentric_coordinates
    {
        double *cc = M, *cc_inv = M + 9, *s3 = M + 18;
        for (int i = 0; i < 3; i++)
            for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j * 3 + i] - cws[i];
        invert_svd3(cc, cc_inv, s3);
        const double* ci = cc_inv;
        for (int i = 0; i < n; i++) {
            const double* pi = pws + 3 * i;
            double* a = alphas + 4 * i;
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0]) + ci[3 * j + 1] * (pi[1] - cws[1]) + ci[3 * j + 2] * (pi[2] - cws[2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }
    // M, MtM, SVD
    for (int i = 0; i < n; i++) {
        const double* as = alphas + 4 * i;
        const double u = us[2 * i], v = us[2 * i + 1];
        double* M1 = M + (2 * i) * 12;
        double* M2 = M1 + 12;
        for (int k = 0; k < 4; k++) {
            M1[3 * k] = as[k] * fu;
            M1[3 * k + 1] = 0.0;
            M1[3 * k + 2] = as[k] * (uc - u);
            M2[3 * k] = 0.0;
            M2[3 * k + 1] = as[k] * fv;
            M2[3 * k + 2] = as[k] * (vc - v);
        }
    }
    {
        double* mtm = vt;  // staged in the V^T area, transposed into the U^T area by svd_square_t
        mul_transposed_ata(M, 2 * n, 12, mtm);
        double* ut = mtm_ut;
        for (int i = 0; i < 12; i++)
            for (int j = 0; j < 12; j++) ut[j * 12 + i] = mtm[i * 12 + j];
        jacobi_svd_impl(ut, 12, d, vt, 12, 12, 12, 12);
    }
    const double* ut = mtm_ut;
    // compute_L_6x10
    {
        const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
        double* dv = M;  // [4][6][3]
        for (int i = 0; i < 4; i++) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                dv[(i * 6 + j) * 3 + 0] = v[i][3 * a] - v[i][3 * b];
                dv[(i * 6 + j) * 3 + 1] = v[i][3 * a + 1] - v[i][3 * b + 1];
                dv[(i * 6 + j) * 3 + 2] = v[i][3 * a + 2] - v[i][3 * b + 2];
                b++;
                if (b > 3) {
                    a++;
                    b = a + 1;
                }
            }
        }
        for (int i = 0; i < 6; i++) {
            double* row = l_6x10 + 10 * i;
            const double *d0 = dv + (0 * 6 + i) * 3, *d1 = dv + (1 * 6 + i) * 3, *d2 = dv + (2 * 6 + i) * 3,
                         *d3 = dv + (3 * 6 + i) * 3;
            row[0] = dot3(d0, d0);
            row[1] = 2.0f * dot3(d0, d1);
            row[2] = dot3(d1, d1);
            row[3] = 2.0f * dot3(d0, d2);
            row[4] = 2.0f * dot3(d1, d2);
            row[5] = dot3(d2, d2);
            row[6] = 2.0f * dot3(d0, d3);
            row[7] = 2.0f * dot3(d1, d3);
            row[8] = 2.0f * dot3(d2, d3);
            row[9] = dot3(d3, d3);
        }
    }
    double rho[6];
    {
        const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
        for (int k = 0; k < 6; k++) {
            const double *p1 = cws + 3 * pa[k], *p2 = cws + 3 * pb[k];
            rho[k] = (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
        }
    }
    double Betas[3][4], rep_errors[3], Rs[3][9], ts[3][3];
    double* sc = M;  // scratch for the small solves: [0,30) L sub-matrix, [30,35) solution, [40,100) SVD work
    // approx 1: columns 0 1 3 6
    {
        double *l = sc, *b4 = sc + 30;
        for (int i = 0; i < 6; i++) {
            l[i * 4 + 0] = l_6x10[i * 10 + 0];
            l[i * 4 + 1] = l_6x10[i * 10 + 1];
            l[i * 4 + 2] = l_6x10[i * 10 + 3];
            l[i * 4 + 3] = l_6x10[i * 10 + 6];
        }
        solve_svd(l, 6, 4, rho, b4, sc + 40);
        double* betas = Betas[0];
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
        epnp_gauss_newton(l_6x10, rho, betas, sc);
        rep_errors[0] = epnp_compute_R_and_t(K4, ut, betas, pws, us, alphas, pcs, ccs, Rs[0], ts[0], sc);
    }
    // approx 2: columns 0 1 2
    {
        double *l = sc, *b3 = sc + 30;
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 3; j++) l[i * 3 + j] = l_6x10[i * 10 + j];
        solve_svd(l, 6, 3, rho, b3, sc + 40);
        double* betas = Betas[1];
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
        epnp_gauss_newton(l_6x10, rho, betas, sc);
        rep_errors[1] = epnp_compute_R_and_t(K4, ut, betas, pws, us, alphas, pcs, ccs, Rs[1], ts[1], sc);
    }
    // approx 3: columns 0 1 2 3 4
    {
        double *l = sc, *b5 = sc + 30;
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 5; j++) l[i * 5 + j] = l_6x10[i * 10 + j];
        solve_svd(l, 6, 5, rho, b5, sc + 40);
        double* betas = Betas[2];
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
        epnp_gauss_newton(l_6x10, rho, betas, sc);
        rep_errors[2] = epnp_compute_R_and_t(K4, ut, betas, pws, us, alphas, pcs, ccs, Rs[2], ts[2], sc);
    }
    int N = 0;
    if (rep_errors[1] < rep_errors[0]) N = 1;
    if (rep_errors[2] < rep_errors[N]) N = 2;
    for (int i = 0; i < 3; i++) tvec[i] = ts[N][i];
    rodrigues_m2v(Rs[N], rvec, sc);
}

// ------------------------------------------------------------------------------------------------
// cvFindExtrinsicCameraParams2 pieces that run on one lane
// ------------------------------------------------------------------------------------------------
// DLT finish: LL (12 x 12, symmetric) -> initial [rvec | tvec]; ws: 2*144 + 12 + 30 doubles
constexpr int PNP_DLT_WS = 2 * 144 + 12 + 30;
SM_HD void pnp_dlt_finish(const double* LL, double* param, double* ws) {
    double *ut = ws, *LV = ws + 144, *LW = ws + 288, *s3 = ws + 300;
    svd_square_t<12>(LL, LW, ut, LV);
    const double* RRt = LV + 11 * 12;
    double RR[9], tt[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) RR[i * 3 + j] = RRt[i * 4 + j];
        tt[i] = RRt[i * 4 + 3];
    }
    if (det3(RR) < 0) {
        for (int i = 0; i < 9; i++) RR[i] *= -1;
        for (int i = 0; i < 3; i++) tt[i] *= -1;
    }
    double sc2 = 0;
    for (int i = 0; i < 3; i++) {
        double r = 0;
        for (int j = 0; j < 3; j++) r += RR[i * 3 + j] * RR[i * 3 + j];
        sc2 += r;
    }
    const double sc = sqrt(sc2);
    double *Wr = s3, *Ut = s3 + 3, *Vt = s3 + 12;
    svd_square_t<3>(RR, Wr, Ut, Vt);
    double R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += Ut[k * 3 + i] * Vt[k * 3 + j];
            R[i * 3 + j] = s;
        }
    double nr = 0;  // cvNorm of a continuous 3 x 3: four squares per step
    nr += R[0] * R[0] + R[1] * R[1] + R[2] * R[2] + R[3] * R[3];
    nr += R[4] * R[4] + R[5] * R[5] + R[6] * R[6] + R[7] * R[7];
    nr += R[8] * R[8];
    const double scale = sqrt(nr) / sc;
    for (int i = 0; i < 3; i++) param[3 + i] = tt[i] * scale;
    rodrigues_m2v(R, param, s3);
}
// CvLevMarq::step (6 parameters): param = prev - solve(JtJ with diag *= 1 + lambda, JtErr); ws: 36 + 78 doubles
constexpr int PNP_LM_WS = 36 + 78;
SM_HD void pnp_lm_step(const double* JtJ, const double* JtErr, int lambdaLg10, const double* prev, double* param,
                       double* ws) {
    const double lambda = lm_lambda(lambdaLg10);
    double* A = ws;
    for (int i = 0; i < 36; i++) A[i] = JtJ[i];
    for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
    double d[6];
    solve_svd(A, 6, 6, JtErr, d, ws + 36);
    for (int i = 0; i < 6; i++) param[i] = prev[i] - d[i];
}
// ||a - b|| / (||b|| + eps) over 6 values, with cv::norm's four-at-a-time accumulation
SM_HD double pnp_rel_change6(const double* a, const double* b) {
    double s = 0, q = 0;
    {
        const double v0 = a[0] - b[0], v1 = a[1] - b[1], v2 = a[2] - b[2], v3 = a[3] - b[3];
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
        const double v4 = a[4] - b[4], v5 = a[5] - b[5];
        s += v4 * v4;
        s += v5 * v5;
    }
    {
        q += b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
        q += b[4] * b[4];
        q += b[5] * b[5];
    }
    return sqrt(s) / (sqrt(q) + DBL_EPSILON);
}

}  // namespace sm
