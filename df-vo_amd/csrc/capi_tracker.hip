// extern "C" surface, part 2: pose solvers with host arrays (see include/dfvo_hip.h).
#include <vector>

#include "solver.h"

using namespace dfvo;

struct dfvo_tracker {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    RansacWorkspace ws;
    double* d_small = nullptr;  // 64 doubles
    double *d_x1 = nullptr, *d_x2 = nullptr, *d_X4 = nullptr;
    int tri_cap = 0;
};

extern "C" {

int dfvo_tracker_create(void* stream, dfvo_tracker** out) {
    DFVO_ARG_CHECK(out, "dfvo_tracker_create: null out");
    dfvo_tracker* t = new dfvo_tracker();
    if (stream) {
        t->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) {
            delete t;
            dfvo::set_last_error("hipStreamCreate failed (no GPU?)");
            return DFVO_ERR_HIP;
        }
        t->own_stream = true;
    }
    if (hipMalloc((void**)&t->d_small, 64 * sizeof(double)) != hipSuccess) {
        delete t;
        dfvo::set_last_error("hipMalloc failed");
        return DFVO_ERR_HIP;
    }
    *out = t;
    return DFVO_OK;
}

void dfvo_tracker_destroy(dfvo_tracker* t) {
    if (!t) return;
    t->ws.release();
    if (t->d_small) (void)hipFree(t->d_small);
    if (t->d_x1) (void)hipFree(t->d_x1);
    if (t->d_x2) (void)hipFree(t->d_x2);
    if (t->d_X4) (void)hipFree(t->d_X4);
    if (t->own_stream && t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
}

static int stage_points(dfvo_tracker* t, const double* h1, const double* h2, int n, int iters) {
    int rc = t->ws.ensure(n > 8 ? n : 8, iters);
    if (rc != DFVO_OK) return rc;
    if (n > 0) {
        DFVO_HIP_CHECK(hipMemcpyAsync(t->ws.pts_a, h1, sizeof(double) * 2 * n, hipMemcpyHostToDevice, t->stream));
        DFVO_HIP_CHECK(hipMemcpyAsync(t->ws.pts_b, h2, sizeof(double) * 2 * n, hipMemcpyHostToDevice, t->stream));
    }
    return DFVO_OK;
}

static int fetch_ransac(dfvo_tracker* t, int n, double* h_model, uint8_t* h_mask, int* h_info) {
    RansacState st;
    DFVO_HIP_CHECK(hipMemcpyAsync(&st, t->ws.state, sizeof(st), hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipMemcpyAsync(h_model, t->ws.out, 9 * sizeof(double), hipMemcpyDeviceToHost, t->stream));
    if (n > 0) DFVO_HIP_CHECK(hipMemcpyAsync(h_mask, t->ws.mask, n, hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    if (h_info) {
        h_info[0] = st.found;
        h_info[1] = st.iter;
        h_info[2] = st.best_iter;
        h_info[3] = st.best_model;
        h_info[4] = st.max_good;
    }
    return DFVO_OK;
}

int dfvo_find_essential_mat(dfvo_tracker* t, const double* h_pts1, const double* h_pts2, int n, double focal,
                            double ppx, double ppy, double prob, double threshold, int max_iters, double* h_E,
                            uint8_t* h_mask, int* h_info) {
    DFVO_ARG_CHECK(t && h_pts1 && h_pts2 && h_E && h_mask && n >= 0 && max_iters >= 1, "dfvo_find_essential_mat: bad argument");
    int rc = stage_points(t, h_pts1, h_pts2, n, max_iters);
    if (rc != DFVO_OK) return rc;
    rc = enqueue_find_essential(t->ws, t->ws.pts_a, t->ws.pts_b, n, focal, ppx, ppy, prob, threshold, max_iters, t->stream);
    if (rc != DFVO_OK) return rc;
    return fetch_ransac(t, n, h_E, h_mask, h_info);
}

int dfvo_find_homography(dfvo_tracker* t, const double* h_pts1, const double* h_pts2, int n, double thr, int max_iters,
                         double confidence, double* h_H, uint8_t* h_mask, int* h_info) {
    DFVO_ARG_CHECK(t && h_pts1 && h_pts2 && h_H && h_mask && n >= 0 && max_iters >= 1, "dfvo_find_homography: bad argument");
    int rc = stage_points(t, h_pts1, h_pts2, n, max_iters);
    if (rc != DFVO_OK) return rc;
    rc = enqueue_find_homography(t->ws, t->ws.pts_a, t->ws.pts_b, n, thr, max_iters, confidence, t->stream);
    if (rc != DFVO_OK) return rc;
    return fetch_ransac(t, n, h_H, h_mask, h_info);
}

int dfvo_recover_pose(dfvo_tracker* t, const double* h_E, const double* h_pts1, const double* h_pts2, int n, double focal,
                      double ppx, double ppy, double* h_R, double* h_t, uint8_t* h_mask, int* h_good) {
    DFVO_ARG_CHECK(t && h_E && h_pts1 && h_pts2 && h_R && h_t && n >= 0, "dfvo_recover_pose: bad argument");
    int rc = stage_points(t, h_pts1, h_pts2, n, 16);
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_small, h_E, 9 * sizeof(double), hipMemcpyHostToDevice, t->stream));
    rc = enqueue_recover_pose(t->ws, t->d_small, t->ws.pts_a, t->ws.pts_b, n, focal, ppx, ppy, t->stream);
    if (rc != DFVO_OK) return rc;
    double res[13];
    DFVO_HIP_CHECK(hipMemcpyAsync(res, t->ws.out + 16, sizeof(res), hipMemcpyDeviceToHost, t->stream));
    if (h_mask && n > 0) DFVO_HIP_CHECK(hipMemcpyAsync(h_mask, t->ws.mask, n, hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    for (int i = 0; i < 9; i++) h_R[i] = res[i];
    for (int i = 0; i < 3; i++) h_t[i] = res[9 + i];
    if (h_good) *h_good = (int)res[12];
    return DFVO_OK;
}

int dfvo_triangulate_points(dfvo_tracker* t, const double* h_P1, const double* h_P2, const double* h_x1,
                            const double* h_x2, int n, double* h_X4) {
    DFVO_ARG_CHECK(t && h_P1 && h_P2 && h_x1 && h_x2 && h_X4 && n >= 0, "dfvo_triangulate_points: bad argument");
    if (n == 0) return DFVO_OK;
    if (n > t->tri_cap) {
        if (t->d_x1) (void)hipFree(t->d_x1);
        if (t->d_x2) (void)hipFree(t->d_x2);
        if (t->d_X4) (void)hipFree(t->d_X4);
        t->tri_cap = n;
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_x1, sizeof(double) * 2 * n));
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_x2, sizeof(double) * 2 * n));
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_X4, sizeof(double) * 4 * n));
    }
    double P[24];
    for (int i = 0; i < 12; i++) {
        P[i] = h_P1[i];
        P[12 + i] = h_P2[i];
    }
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_small, P, sizeof(P), hipMemcpyHostToDevice, t->stream));
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_x1, h_x1, sizeof(double) * 2 * n, hipMemcpyHostToDevice, t->stream));
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_x2, h_x2, sizeof(double) * 2 * n, hipMemcpyHostToDevice, t->stream));
    int rc = enqueue_triangulate(t->d_small, t->d_x1, t->d_x2, n, t->d_X4, t->stream);
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(hipMemcpyAsync(h_X4, t->d_X4, sizeof(double) * 4 * n, hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    return DFVO_OK;
}

}  // extern "C"
