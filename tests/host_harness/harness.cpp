// TEST-ONLY host build of df-vo_amd/csrc/solver_math.h (the per-lane device functions of the HIP
// solvers) so that their arithmetic can be checked bit-for-bit against the C oracle on a machine
// without a GPU.  Never linked into libdfvo_hip.so; the product has no CPU path.
#include <cmath>

#include "../../df-vo_amd/csrc/solver_math.h"

extern "C" {

int hh_five_point(const double* q1, const double* q2, double* E_out) {
    double EE[36], b[39], c[11], rre[10], rim[10];
    if (!sm::five_point_stage1(q1, q2, EE, b, c)) return 0;
    sm::solve_poly10(c, rre, rim);
    return sm::five_point_stage3(EE, b, rre, rim, E_out);
}
float hh_essential_error(const double* E, double a, double b, double c, double d) {
    return sm::essential_error(E, a, b, c, d);
}
int hh_homography_kernel(const float* M, const float* m, int count, double* H) {
    return sm::homography_kernel(M, m, count, H) ? 1 : 0;
}
int hh_homography_check_subset(const float* M, const float* m) { return sm::homography_check_subset(M, m) ? 1 : 0; }
float hh_homography_error(const double* H, float Mx, float My, float mx, float my) {
    float Hf[8];
    for (int i = 0; i < 8; i++) Hf[i] = (float)H[i];
    return sm::homography_error(Hf, Mx, My, mx, my);
}
void hh_decompose_essential(const double* E, double* R1, double* R2, double* t) { sm::decompose_essential(E, R1, R2, t); }
void hh_triangulate_point(const double* P1, const double* P2, double x1, double y1, double x2, double y2, double* X) {
    sm::triangulate_point(P1, P2, x1, y1, x2, y2, X);
}
int hh_ransac_update_num_iters(double p, double ep, int mp, int mi) { return sm::ransac_update_num_iters(p, ep, mp, mi); }
void hh_solve_eig8(const double* A, const double* b, double* x) { sm::solve_eig<8>(A, b, x); }
void hh_invert_eig8(const double* A, double* d) { sm::invert_eig<8>(A, d); }
}

// ---- PnP fallback lane functions
#include "../../df-vo_amd/csrc/pnp_math.h"
extern "C" {
double hh_det_sin(double x) { return sm::det_sin(x); }
double hh_det_cos(double x) { return sm::det_cos(x); }
double hh_det_acos(double x) { return sm::det_acos(x); }
double hh_lm_lambda(int k) { return sm::lm_lambda(k); }
void hh_rodrigues_v2m(const double* r, double* R, double* J) { sm::rodrigues_v2m(r, R, J); }
void hh_rodrigues_m2v(const double* R, double* r) {
    double ws[21];
    sm::rodrigues_m2v(R, r, ws);
}
void hh_epnp_kernel(const double* K4, const float* obj, const float* img, double* rvec, double* tvec) {
    double ws[sm::EPNP_WS];
    sm::epnp_kernel(K4, obj, img, rvec, tvec, ws);
}
float hh_pnp_error(const double* rvec, const double* tvec, const double* K4, const float* obj, const float* img) {
    double R[9];
    sm::rodrigues_v2m(rvec, R, nullptr);
    return sm::pnp_error(R, tvec, K4, obj, img);
}
// the flow of the refinement kernel (k_pnp_refine) written sequentially on the host with the same lane functions:
// centroid / covariance planarity test, DLT (or, for coplanar points, homography) initialisation, CvLevMarq loop.
// cv::findHomography(method 0) for the planar branch: the lane functions the device block is built from
// (sm::homography_kernel, the eigen solvers) driven by the sequential LMSolver loop
static void hh_homography_lsq(const float* M, const float* m, int n, double* H) {
    if (!sm::homography_kernel(M, m, n, H)) {
        for (int i = 0; i < 9; i++) H[i] = 0;
        return;
    }
    if (n <= 4) return;
    const int lx = 8, rows = 2 * n;
    double x[8], xd[8], d[8], v[8], A[64], Ap[64], D[8];
    double* r = new double[rows];
    double* rd = new double[rows];
    double* J = new double[(size_t)rows * 8];
    auto compute = [&](const double* h, double* err, double* Jo) {
        for (int i = 0; i < n; i++) {
            const double Mx = M[i * 2], My = M[i * 2 + 1];
            double ww = h[6] * Mx + h[7] * My + 1.;
            ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
            const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww, yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
            err[i * 2] = xi - m[i * 2];
            err[i * 2 + 1] = yi - m[i * 2 + 1];
            if (Jo) {
                double* Jp = Jo + (size_t)i * 16;
                Jp[0] = Mx * ww; Jp[1] = My * ww; Jp[2] = ww; Jp[3] = Jp[4] = Jp[5] = 0.; Jp[6] = -Mx * ww * xi; Jp[7] = -My * ww * xi;
                Jp[8] = Jp[9] = Jp[10] = 0.; Jp[11] = Mx * ww; Jp[12] = My * ww; Jp[13] = ww; Jp[14] = -Mx * ww * yi; Jp[15] = -My * ww * yi;
            }
        }
    };
    auto nsq = [](const double* a, int cnt) {
        double s = 0;
        int i = 0;
        for (; i <= cnt - 4; i += 4) s += a[i] * a[i] + a[i + 1] * a[i + 1] + a[i + 2] * a[i + 2] + a[i + 3] * a[i + 3];
        for (; i < cnt; i++) s += a[i] * a[i];
        return s;
    };
    auto dot8 = [](const double* a, const double* b) {
        double q = 0;
        q += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
        q += a[4] * b[4] + a[5] * b[5] + a[6] * b[6] + a[7] * b[7];
        return q;
    };
    auto normal = [&]() {
        for (int i = 0; i < lx; i++)
            for (int j = i; j < lx; j++) {
                double q = 0;
                for (int k = 0; k < rows; k++) q += J[(size_t)k * lx + i] * J[(size_t)k * lx + j];
                A[i * lx + j] = A[j * lx + i] = q;
            }
        for (int i = 0; i < lx; i++) {
            double q = 0;
            for (int k = 0; k < rows; k++) q += J[(size_t)k * lx + i] * r[k];
            v[i] = q;
        }
    };
    for (int i = 0; i < 8; i++) x[i] = H[i];
    compute(x, r, J);
    double S = nsq(r, rows);
    normal();
    for (int i = 0; i < lx; i++) D[i] = A[i * lx + i];
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        for (int i = 0; i < 64; i++) Ap[i] = A[i];
        for (int i = 0; i < lx; i++) Ap[i * lx + i] += lambda * D[i];
        sm::solve_eig<8>(Ap, v, d);
        for (int i = 0; i < lx; i++) xd[i] = x[i] - d[i];
        compute(xd, rd, nullptr);
        const double Sd = nsq(rd, rows);
        double temp_d[8];
        for (int i = 0; i < lx; i++) {
            double q = 0;
            for (int k = 0; k < lx; k++) q += A[i * lx + k] * d[k];
            temp_d[i] = q * -1. + v[i] * 2.;
        }
        const double dS = dot8(d, temp_d);
        const double Rr = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (Rr > 0.75) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (Rr < 0.25) {
            const double tq = dot8(d, v);
            double nu = (Sd - S) / (fabs(tq) > DBL_EPSILON ? tq : 1) + 2;
            nu = nu > 2. ? nu : 2.;
            nu = nu < 10. ? nu : 10.;
            if (lambda == 0) {
                sm::invert_eig<8>(A, Ap);
                double maxval = DBL_EPSILON;
                for (int i = 0; i < lx; i++) maxval = maxval > fabs(Ap[i * lx + i]) ? maxval : fabs(Ap[i * lx + i]);
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            for (int i = 0; i < 8; i++) {
                const double tx = x[i];
                x[i] = xd[i];
                xd[i] = tx;
            }
            compute(x, r, J);
            normal();
        }
        iter++;
        double dinf = 0, rinf = 0;
        for (int i = 0; i < lx; i++) dinf = dinf > fabs(d[i]) ? dinf : fabs(d[i]);
        for (int i = 0; i < rows; i++) rinf = rinf > fabs(r[i]) ? rinf : fabs(r[i]);
        if (!(iter < 10 && dinf >= FLT_EPSILON && rinf >= FLT_EPSILON)) break;
    }
    for (int i = 0; i < 8; i++) H[i] = x[i];
    delete[] r;
    delete[] rd;
    delete[] J;
}

int hh_find_extrinsic(const double* M, const double* m, int n, const double* K4, double* rvec, double* tvec) {
    const double ifx = 1. / K4[0], ify = 1. / K4[1];
    double Mc[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) Mc[j] += M[i * 3 + j];
    const double inv_n = 1. / n;
    for (int j = 0; j < 3; j++) Mc[j] *= inv_n;
    double MM[9], W[3], ut[9], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = i; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += (M[k * 3 + i] - Mc[i]) * (M[k * 3 + j] - Mc[j]);
            MM[i * 3 + j] = MM[j * 3 + i] = s;
        }
    sm::svd_square_t<3>(MM, W, ut, V);
    const bool planar = W[2] / W[1] < 1e-3 || n < 4;
    double param[6], ws[sm::PNP_DLT_WS];
    if (planar) {  // cvFindExtrinsicCameraParams2's planar initialisation, as k_pnp_refine does it
        double Rt[9];
        for (int i = 0; i < 9; i++) Rt[i] = V[i];
        if (V[2] * V[2] + V[5] * V[5] < 1e-10) {
            for (int i = 0; i < 9; i++) Rt[i] = 0.;
            Rt[0] = Rt[4] = Rt[8] = 1.;
        }
        if (sm::det3(Rt) < 0)
            for (int i = 0; i < 9; i++) Rt[i] *= -1.;
        double Tt[3];
        for (int i = 0; i < 3; i++) Tt[i] = (Rt[i * 3] * Mc[0] + Rt[i * 3 + 1] * Mc[1] + Rt[i * 3 + 2] * Mc[2]) * -1.;
        float* Mf = new float[2 * n];
        float* mf = new float[2 * n];
        for (int i = 0; i < n; i++) {
            const double X = M[i * 3], Y = M[i * 3 + 1], Z = M[i * 3 + 2];
            Mf[i * 2] = (float)(Rt[0] * X + Rt[1] * Y + Rt[2] * Z + Tt[0]);
            Mf[i * 2 + 1] = (float)(Rt[3] * X + Rt[4] * Y + Rt[5] * Z + Tt[1]);
            mf[i * 2] = (float)((m[i * 2] - K4[2]) * ifx);
            mf[i * 2 + 1] = (float)((m[i * 2 + 1] - K4[3]) * ify);
        }
        double h[9];
        hh_homography_lsq(Mf, mf, n, h);
        delete[] Mf;
        delete[] mf;
        bool finite = true;
        for (int i = 0; i < 9; i++) finite = finite && std::isfinite(h[i]);
        double Rm[9], tt[3] = {0., 0., 0.};
        if (finite) {
            const double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
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
            sm::rodrigues_m2v(h, r3, ws);
            sm::rodrigues_v2m(r3, Hr, nullptr);
            for (int i = 0; i < 3; i++) tt[i] = (Hr[i * 3] * Tt[0] + Hr[i * 3 + 1] * Tt[1] + Hr[i * 3 + 2] * Tt[2]) + tt[i];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) Rm[i * 3 + j] = Hr[i * 3] * Rt[j] + Hr[i * 3 + 1] * Rt[3 + j] + Hr[i * 3 + 2] * Rt[6 + j];
        } else {
            for (int i = 0; i < 9; i++) Rm[i] = 0.;
            Rm[0] = Rm[4] = Rm[8] = 1.;
        }
        sm::rodrigues_m2v(Rm, param, ws);
        for (int i = 0; i < 3; i++) param[3 + i] = tt[i];
    }
    double LL[144];
    if (!planar) {
    for (int a = 0; a < 12; a++)
        for (int b = a; b < 12; b++) {
            double s = 0;
            for (int i = 0; i < n; i++) {
                double L0[12], L1[12];
                const double x = -((m[i * 2] - K4[2]) * ifx), y = -((m[i * 2 + 1] - K4[3]) * ify);
                const double X = M[i * 3], Y = M[i * 3 + 1], Z = M[i * 3 + 2];
                const double r0[12] = {X, Y, Z, 1., 0., 0., 0., 0., x * X, x * Y, x * Z, x};
                const double r1[12] = {0., 0., 0., 0., X, Y, Z, 1., y * X, y * Y, y * Z, y};
                for (int k = 0; k < 12; k++) L0[k] = r0[k], L1[k] = r1[k];
                s += L0[a] * L0[b];
                s += L1[a] * L1[b];
            }
            LL[a * 12 + b] = LL[b * 12 + a] = s;
        }
    sm::pnp_dlt_finish(LL, param, ws);
    }
    double JtJ[36], JtErr[6], prev[6], lmws[sm::PNP_LM_WS];
    double prevErrNorm = DBL_MAX, errNorm = 0;
    int lambdaLg10 = -3, iters = 0;
    bool calc_j = true;
    double* J = new double[12 * n];
    double* err = new double[2 * n];
    auto norm_l2 = [](const double* a, int cnt) {
        double s = 0;
        int i = 0;
        for (; i <= cnt - 4; i += 4) s += a[i] * a[i] + a[i + 1] * a[i + 1] + a[i + 2] * a[i + 2] + a[i + 3] * a[i + 3];
        for (; i < cnt; i++) s += a[i] * a[i];
        return sqrt(s);
    };
    for (;;) {
        double R[9], dRdr[27];
        sm::rodrigues_v2m(param, R, calc_j ? dRdr : nullptr);
        for (int i = 0; i < n; i++) {
            double u, v, jr[6], jt[6];
            sm::project_point(R, param + 3, K4, M[i * 3], M[i * 3 + 1], M[i * 3 + 2], &u, &v, dRdr, calc_j ? jr : nullptr,
                              calc_j ? jt : nullptr);
            err[2 * i] = u - m[2 * i];
            err[2 * i + 1] = v - m[2 * i + 1];
            if (calc_j)
                for (int j = 0; j < 3; j++) {
                    J[(2 * i) * 6 + j] = jr[j];
                    J[(2 * i) * 6 + 3 + j] = jt[j];
                    J[(2 * i + 1) * 6 + j] = jr[3 + j];
                    J[(2 * i + 1) * 6 + 3 + j] = jt[3 + j];
                }
        }
        if (calc_j) {
            sm::mul_transposed_ata(J, 2 * n, 6, JtJ);
            for (int j = 0; j < 6; j++) {
                double s = 0;
                for (int k = 0; k < 2 * n; k++) s += J[k * 6 + j] * err[k];
                JtErr[j] = s;
            }
            for (int i = 0; i < 6; i++) prev[i] = param[i];
            sm::pnp_lm_step(JtJ, JtErr, lambdaLg10, prev, param, lmws);
            if (iters == 0) prevErrNorm = norm_l2(err, 2 * n);
            calc_j = false;
            continue;
        }
        errNorm = norm_l2(err, 2 * n);
        if (errNorm > prevErrNorm) {
            if (++lambdaLg10 <= 16) {
                sm::pnp_lm_step(JtJ, JtErr, lambdaLg10, prev, param, lmws);
                continue;
            }
        }
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        if (++iters >= 20 || sm::pnp_rel_change6(param, prev) < FLT_EPSILON) break;
        prevErrNorm = errNorm;
        calc_j = true;
    }
    for (int i = 0; i < 3; i++) rvec[i] = param[i], tvec[i] = param[3 + i];
    delete[] J;
    delete[] err;
    return 1;
}
}

// ---- numpy legacy RandomState + argpartition emulation
#include "../../df-vo_amd/csrc/kp_select.h"
#include "../../df-vo_amd/csrc/np_legacy.h"
extern "C" {
// state: key[624] + pos (as uint32[625]); functions update it in place
void hh_mt_shuffle(uint32_t* state, int n, int* perm) {
    sm::Mt19937 s;
    for (int i = 0; i < 624; i++) s.key[i] = state[i];
    s.pos = (int)state[624];
    sm::mt_shuffle_arange(s, n, perm);
    for (int i = 0; i < 624; i++) state[i] = s.key[i];
    state[624] = (uint32_t)s.pos;
}
void hh_mt_sample(uint32_t* state, int n_population, int n_samples, int* out) {
    sm::Mt19937 s;
    for (int i = 0; i < 624; i++) s.key[i] = state[i];
    s.pos = (int)state[624];
    int scratch[4096];
    sm::mt_sample_without_replacement(s, n_population, n_samples, out, scratch);
    for (int i = 0; i < 624; i++) state[i] = s.key[i];
    state[624] = (uint32_t)s.pos;
}
// keys-carried-along variant (what k_kp_cell runs out of LDS); padded so that the 4-wide scans may over-read
void hh_argpartition_cp(const float* v, int num, int kth, int* tosort) {
    float* key = new float[num + 8];
    for (int i = 0; i < num + 8; i++) key[i] = 0.f;
    for (int i = 0; i < num; i++) {
        tosort[i] = i;
        key[4 + i] = v[i];
    }
    if (num > 0) sm::kp_introselect_cp<int>(key + 4, tosort, num, kth, 0);
    delete[] key;
}
void hh_argpartition(const float* v, int num, int kth, int* tosort) {
    for (int i = 0; i < num; i++) tosort[i] = i;
    if (num > 0) sm::kp_introselect<int>(v, tosort, num, kth, 0);
}
void hh_cell_bounds(int h, int w, int nr, int nc, int row, int col, int* out) {
    sm::kp_cell_bounds(h, w, nr, nc, row, col, out, out + 1, out + 2, out + 3);
}
double hh_np_pairwise_sum(const double* a, int n) { return sm::np_pairwise_sum(a, n); }
// k_rigid_flow_diff per pixel: mats = Kinv[9] | T[16] | K[9]; out [2][H][W] rigid flow
void hh_rigid_flow(const float* mats, const float* depth, int H, int W, float* out) {
    for (int i = 0; i < H * W; i++)
        sm::rigid_flow_px(mats, mats + 9, mats + 25, (float)(i % W), (float)(i / W), depth[i], out + i, out + (size_t)H * W + i);
}
}

// ---- lock-step host emulation of df-vo_amd/csrc/solver_poly_lanes.h (one root per lane, sixteen lanes per row):
// every statement of the device schedule is executed for all lanes before the next one, the DPP row broadcast is a copy
// from lane S's value.  What it proves on the host: the lane-parallel Gauss-Seidel order yields the sequential bits.
extern "C" void hh_solve_poly10_lockstep(const double* c, double* rre, double* rim) {
    constexpr int L = 16;
    struct Z {
        double cr[11], xr[10], xi[10], pre, pim, nre, nim, dre, dim, md;
    } z[L];
    for (int l = 0; l < L; l++) {
        for (int i = 0; i <= 10; i++) z[l].cr[i] = c[i];
        double pre = 1, pim = 0;
        for (int i = 0; i < 10; i++) {
            z[l].xr[i] = pre;
            z[l].xi[i] = pim;
            const double tre = pre * 1.0 - pim * 1.0, tim = pre * 1.0 + pim * 1.0;
            pre = tre;
            pim = tim;
        }
        const int own = l < 10 ? l : 0;
        z[l].pre = z[l].xr[own];
        z[l].pim = z[l].xi[own];
    }
    bool active = true;
    for (int iter = 0; iter < 300; iter++) {
        if (!active) break;
        for (int l = 0; l < L; l++) {
            Z& q = z[l];
            q.nre = q.cr[10];
            q.nim = 0;
            for (int j = 0; j < 10; j++) {
                const double tre = q.nre * q.pre - q.nim * q.pim, tim = q.nre * q.pim + q.nim * q.pre;
                q.nre = tre + q.cr[10 - j - 1];
                q.nim = tim + 0.0;
            }
            q.dre = q.cr[10];
            q.dim = 0;
            q.md = 0;
        }
        for (int S = 0; S < 10; S++) {
            {
                Z& q = z[S];  // the lane with li == S
                double dre = q.dre, dim = q.dim;
                for (int j = S + 1; j < 10; j++) {
                    const double qre = q.pre - q.xr[j], qim = q.pim - q.xi[j];
                    const double tre = dre * qre - dim * qim, tim = dre * qim + dim * qre;
                    dre = tre;
                    dim = tim;
                }
                const double t = 1. / (dre * dre + dim * dim);
                const double qre = (q.nre * dre + q.nim * dim) * t, qim = (-q.nre * dim + q.nim * dre) * t;
                q.pre = q.pre - qre;
                q.pim = q.pim - qim;
                const double an = qre * qre + qim * qim;
                q.md = q.md > an ? q.md : an;
            }
            const double bre = z[S].pre, bim = z[S].pim, bmd = z[S].md;
            for (int l = 0; l < L; l++) {
                z[l].xr[S] = bre;
                z[l].xi[S] = bim;
                z[l].md = bmd;
            }
            for (int l = S + 1; l < L; l++) {
                Z& q = z[l];
                const double qre = q.pre - q.xr[S], qim = q.pim - q.xi[S];
                const double tre = q.dre * qre - q.dim * qim, tim = q.dre * qim + q.dim * qre;
                q.dre = tre;
                q.dim = tim;
            }
        }
        if (z[0].md <= 0) active = false;
    }
    for (int l = 0; l < 10; l++) {
        rre[l] = z[l].pre;
        rim[l] = fabs(z[l].pim) < 1e-100 ? 0 : z[l].pim;
    }
}
extern "C" int hh_five_point_poly(const double* q1, const double* q2, double* c) {
    double EE[36], b[39];
    return sm::five_point_stage1(q1, q2, EE, b, c) ? 1 : 0;
}
extern "C" void hh_solve_poly10(const double* c, double* rre, double* rim) { sm::solve_poly10(c, rre, rim); }

// ---- lock-step host emulation of jacobi_eigen_coop (df-vo_amd/csrc/h_refine_dev.h, round-5 form): sixteen candidate lanes with
// the pivot-table entry and the eigenvalue they own in "registers" (per-lane variables), the pivot maximum as four pairwise
// exchanges (lane ^ 1, lane ^ 2, mirror inside eight, mirror inside sixteen) in which every lane combines its OLD value with its
// partner's OLD value, the winner read from lane 0, the rotation one element pair per lane, the rescans by the owner lanes.
// What it proves on the host: that schedule visits the same pivots and produces the same bits as the sequential
// sm::jacobi_eigen_ws, ties between equal magnitudes and NaN entries included.
template <int N>
static int jacobi_lockstep(double* A, double* W, double* V) {
    constexpr int L = 16;
    static_assert(2 * N - 2 <= L, "candidates fit one row");
    const double eps = DBL_EPSILON;
    bool own_row[L], own_col[L];
    int oc[L], myind[L] = {};
    double myW[L] = {};
    auto rescan = [&](int lane) {
        const bool r = own_row[lane];
        const int base = r ? N * lane + lane + 1 : oc[lane], stride = r ? 1 : N, cnt = r ? N - 1 - lane : oc[lane], m0 = r ? lane + 1 : 0;
        int m = m0;
        double mv = fabs(A[base]);
        for (int j = 1; j < cnt; j++) {
            const double val = fabs(A[base + j * stride]);
            if (mv < val) mv = val, m = m0 + j;
        }
        return m;
    };
    for (int i = 0; i < N * N; i++) V[i] = (i / N == i % N) ? 1. : 0.;
    for (int lane = 0; lane < L; lane++) {
        own_row[lane] = lane < N - 1;
        own_col[lane] = lane >= N - 1 && lane < 2 * N - 2;
        oc[lane] = lane - (N - 2);
        if (lane < N) myW[lane] = A[(N + 1) * lane];
    }
    for (int lane = 0; lane < L; lane++)
        if (own_row[lane] || own_col[lane]) myind[lane] = rescan(lane);
    const int maxIters = N * N * 30;
    int iters = 0;
    for (; iters < maxIters; iters++) {
        double key[L], p[L];
        int code[L];
        for (int lane = 0; lane < L; lane++) {
            key[lane] = -1.;
            p[lane] = 0.;
            code[lane] = 0x7fffffff;
            if (own_row[lane] || own_col[lane]) {
                const int ck = own_row[lane] ? lane : myind[lane], cl = own_row[lane] ? myind[lane] : oc[lane];
                p[lane] = A[N * ck + cl];
                key[lane] = fabs(p[lane]);
                if (key[lane] != key[lane]) key[lane] = lane == 0 ? INFINITY : -1.;
                code[lane] = lane << 16 | ck << 8 | cl;
            }
        }
        for (int step = 0; step < 4; step++) {
            double ok[L], op[L];
            int ocd[L];
            for (int lane = 0; lane < L; lane++) {
                const int q = step == 0 ? lane ^ 1 : step == 1 ? lane ^ 2 : step == 2 ? (lane & 8) | (7 - (lane & 7)) : 15 - lane;
                ok[lane] = key[q];
                op[lane] = p[q];
                ocd[lane] = code[q];
            }
            for (int lane = 0; lane < L; lane++)
                if (ok[lane] > key[lane] || (ok[lane] == key[lane] && ocd[lane] < code[lane])) {
                    key[lane] = ok[lane];
                    p[lane] = op[lane];
                    code[lane] = ocd[lane];
                }
        }
        const int cd = code[0];
        const double pv = p[0];
        const int k = (cd >> 8) & 0xff, l = cd & 0xff;
        if (fabs(pv) <= eps) break;
        const double y = (myW[l] - myW[k]) * 0.5;
        double t = fabs(y) + sm::hypot_p(pv, y);
        double sn = sm::hypot_p(pv, t);
        const double c = t / sn;
        sn = pv / sn;
        t = (pv / t) * pv;
        if (y < 0) sn = -sn, t = -t;
        myW[k] -= t;
        myW[l] += t;
        A[N * k + l] = 0;
        for (int i = 0; i < N; i++) {
            if (i != k && i != l) {
                const int ak = i < k ? N * i + k : N * k + i, al = i < l ? N * i + l : N * l + i;
                const double a0 = A[ak], b0 = A[al];
                A[ak] = a0 * c - b0 * sn;
                A[al] = a0 * sn + b0 * c;
            }
            const double a0 = V[N * k + i], b0 = V[N * l + i];
            V[N * k + i] = a0 * c - b0 * sn;
            V[N * l + i] = a0 * sn + b0 * c;
        }
        for (int lane = 0; lane < L; lane++)
            if ((own_row[lane] && (lane == k || lane == l)) || (own_col[lane] && (oc[lane] == k || oc[lane] == l))) myind[lane] = rescan(lane);
    }
    for (int i = 0; i < N; i++) W[i] = myW[i];
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
    return iters;
}
extern "C" int hh_jacobi_eigen_lockstep(int n, const double* A_in, double* A_out, double* W, double* V) {
    for (int i = 0; i < n * n; i++) A_out[i] = A_in[i];
    return n == 9 ? jacobi_lockstep<9>(A_out, W, V) : n == 8 ? jacobi_lockstep<8>(A_out, W, V) : -1;
}
extern "C" int hh_jacobi_eigen_sequential(int n, const double* A_in, double* A_out, double* W, double* V) {
    for (int i = 0; i < n * n; i++) A_out[i] = A_in[i];
    if (n == 9)
        sm::jacobi_eigen<9>(A_out, W, V);
    else if (n == 8)
        sm::jacobi_eigen<8>(A_out, W, V);
    else
        return -1;
    return 0;
}
