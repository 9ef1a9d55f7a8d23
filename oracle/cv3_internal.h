/* ORACLE (test infrastructure only) -- declarations shared by the oracle's own translation units. */
#ifndef CV3_INTERNAL_H
#define CV3_INTERNAL_H
#include <stddef.h>

#include "cv3_calib3d.h"

/* PointSetRegistrator::Callback (ptsetreg.cpp) */
typedef struct {
    int model_points;
    int model_size; /* doubles per model */
    /* runKernel: subset points (model_points of them) -> up to max models, returns count */
    int (*run_kernel)(const void* ctx, const void* ms1, const void* ms2, int count, double* models);
    /* computeError for all n points */
    void (*compute_error)(const void* ctx, const void* m1, const void* m2, int n, const double* model, float* err);
    /* checkSubset (may be NULL = always true) */
    int (*check_subset)(const void* ctx, const void* ms1, const void* ms2, int count);
    size_t esz1, esz2; /* bytes per point */
    const void* ctx;
} cv3_ransac_cb;

/* RANSACPointSetRegistrator::run; returns 1 on success.  stats (optional): [0] iterations run,
 * [1] winning iteration, [2] winning model index */
int cv3_ransac_run(const cv3_ransac_cb* cb, const void* m1, const void* m2, int count, double threshold,
                   double confidence, int maxIters, double* model_out, unsigned char* mask_out, int* stats);

/* lapack.cpp SVBkSbImpl_ (double): x (n x nb) = v * diag(1/w) * u^T * b; b == NULL -> identity (nb = m) */
void cv3_svbksb(int m, int n, const double* w, const double* u, int ldu, int uT, const double* v, int ldv, int vT,
                const double* b, int ldb, int nb, double* x, int ldx);
/* cv::solve(A (m x n, m >= n), b (m x 1), DECOMP_SVD) */
void cv3_solve_svd(const double* A, int m, int n, const double* b, double* x);
/* cv::invert(A (n x n), DECOMP_SVD) */
void cv3_invert_svd(const double* A, int n, double* dst);

#endif
