/* ORACLE (test infrastructure only; never linked into or called by the product path).
 *
 * Plain-C restatement of the OpenCV 3.4.3 routines that DF-VO's trackers call
 * (opencv-python==3.4.3.18 is an un-vendored third-party dependency of the reference:
 * /root/reference/envs/requirement.yml:296).  OpenCV's sources are NOT available in the build
 * container, so these functions restate the PUBLISHED algorithms of the named upstream files from
 * knowledge of them.  PARITY UNPINNED: no OpenCV golden vectors exist here; "bit-exact" claims made
 * against this oracle mean "bit-exact vs this OpenCV-3.4.3-following restatement".
 *
 * cv3_core: modules/core/src/{rand.cpp (RNG), lapack.cpp (JacobiSVD, Jacobi eigen, LU, SVBkSb,
 *           solve/invert DECOMP_EIG), mathfuncs.cpp (solvePoly)}.
 */
#ifndef CV3_CORE_H
#define CV3_CORE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::RNG (rand.cpp): multiply-with-carry, state = (uint32)state*4164903690 + (state >> 32) */
typedef struct { uint64_t state; } cv3_rng;
void cv3_rng_init(cv3_rng* r, uint64_t seed);
unsigned cv3_rng_next(cv3_rng* r);
int cv3_rng_uniform_int(cv3_rng* r, int a, int b);

/* lapack.cpp JacobiSVDImpl_<double>: At is n rows x m (row stride astep), one-sided Hestenes
 * rotations on the rows; W[n]; Vt n x n (may be NULL); n1 = number of (normalised) rows of At
 * wanted on output (rows >= n are completed with the deterministic pseudo-random basis). */
void cv3_jacobi_svd(double* At, int astep, double* W, double* Vt, int vstep, int m, int n, int n1);
/* cv::SVD::compute(src (m x n), w, u, vt, flags): u (m x urows?) see .c; full_uv like SVD::FULL_UV.
 * Outputs: w[min(m,n)], u (m x (full?m:min)), vt ((full?n:min) x n); u/vt may be NULL. */
void cv3_svd_compute(const double* src, int m, int n, double* w, double* u, double* vt, int full_uv);
/* lapack.cpp JacobiImpl_<double>: symmetric eigen decomposition, eigenvalues sorted descending,
 * eigenvectors in the ROWS of V.  A is destroyed. */
void cv3_jacobi_eigen(double* A, int n, double* W, double* V);
/* lapack.cpp LUImpl<double>: solves A X = B in place (B is m x n), returns 0 if singular */
int cv3_lu(double* A, int astep, int m, double* b, int bstep, int n);
/* cv::invert(DECOMP_LU) for n x n, returns 0 if singular */
int cv3_invert_lu(const double* src, int n, double* dst);
/* cv::solve(A (n x n symmetric), b (n), x, DECOMP_EIG) */
void cv3_solve_eig(const double* A, int n, const double* b, double* x);
/* cv::invert(A, DECOMP_EIG) */
void cv3_invert_eig(const double* A, int n, double* dst);
/* cv::solvePoly: coeffs[0..n] (coeffs[i] multiplies x^i), roots re/im [n]; maxIters default 300 */
void cv3_solve_poly(const double* coeffs, int n, double* roots_re, double* roots_im, int maxIters);
/* cv::determinant of a 3x3 */
double cv3_det3(const double* m);
/* 3x3 * 3x3 (gemm small-matrix path: sequential k sums) */
void cv3_mul33(const double* a, const double* b, double* d);
/* round half to even like cvRound */
int cv3_round(double v);

#ifdef __cplusplus
}
#endif
#endif
