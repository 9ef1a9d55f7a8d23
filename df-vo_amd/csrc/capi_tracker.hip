// extern "C" surface, part 2: pose solvers with host arrays (see include/dfvo_hip.h).
#include <vector>

#include "solver.h"
#include "tracker.h"
#include "capi_types.h"
#include "../../include/dfvo_hip.h"

using namespace dfvo;

// shared with session.hip (declared in capi_types.h)
int dfvo_pose_config_from(const dfvo_pose2d2d_cfg* cfg, dfvo::PoseConfig* pcp) {
    PoseConfig& pc = *pcp;
    pc.fx = cfg->fx;
    pc.cx = cfg->cx;
    pc.cy = cfg->cy;
    pc.reproj_thre = cfg->reproj_thre;
    pc.repeat = cfg->repeat;
    pc.max_iters = cfg->max_iters;
    for (int i = 0; i < 9; i++) {
        pc.KinvT[i] = cfg->KinvT[i];
        pc.Kinv[i] = cfg->Kinv[i];
    }
    DFVO_ARG_CHECK(cfg->validity_method == DFVO_VALIDITY_GRIC || cfg->validity_method == DFVO_VALIDITY_FLOW ||
                       cfg->validity_method == DFVO_VALIDITY_HOMO_RATIO,
                   "dfvo_compute_pose_2d2d: unknown validity_method");
    pc.validity = cfg->validity_method;
    pc.validity_thre = cfg->validity_thre;
    return DFVO_OK;
}

// results of the pose chain enqueued on t->stream: waits, fills out / h_inliers
int dfvo_pose_fetch(dfvo_tracker* t, int n, const dfvo_pose2d2d_cfg* cfg, dfvo_pose2d2d_out* out, uint8_t* h_inliers) {
    PoseState ps;
    DFVO_HIP_CHECK(hipMemcpyAsync(&ps, t->tb.pose, sizeof(ps), hipMemcpyDeviceToHost, t->stream));
    if (n > 0) DFVO_HIP_CHECK(hipMemcpyAsync(h_inliers, t->tb.best_inliers, n, hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    for (int i = 0; i < 9; i++) out->R[i] = ps.R[i];
    for (int i = 0; i < 3; i++) out->t[i] = ps.t[i];
    out->n = ps.n;
    out->best_inlier_cnt = ps.best_cnt;
    out->num_valid = ps.num_valid;
    out->major_valid = (cfg->validity_method == DFVO_VALIDITY_GRIC ? n > 10 : n >= 5) ? ps.major_valid : 0;
    out->cheirality = ps.cheirality;
    out->h_found = ps.h_found;
    out->h_gric = ps.h_gric;
    for (int i = 0; i < 8; i++) {
        out->rep_inliers[i] = ps.rep_cnt[i];
        out->rep_valid[i] = ps.rep_valid[i];
        out->rep_gric[i] = ps.rep_gric[i];
    }
    return DFVO_OK;
}

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
    if (hipMalloc((void**)&t->d_small, 64 * sizeof(double)) != hipSuccess || t->tb.init() != DFVO_OK) {
        delete t;
        dfvo::set_last_error("hipMalloc failed");
        return DFVO_ERR_HIP;
    }
    enqueue_mt_seed(t->tb, 5489u, t->stream);
    if (t->tb.enable_stage_timing() != DFVO_OK) {  // the mirrors feed the reference's Timer sub-keys from these
        dfvo_tracker_destroy(t);
        return DFVO_ERR_HIP;
    }
    *out = t;
    return DFVO_OK;
}

int dfvo_tracker_stage_ms(dfvo_tracker* t, double* h_ms8) {
    DFVO_ARG_CHECK(t && h_ms8, "dfvo_tracker_stage_ms: bad argument");
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    // (key, first mark, last mark) in the order of the header: find H, GRIC-H, find-Ess, GRIC-E, find-Ess (full),
    // recover pose, triangulation, scale ransac
    static const int seg[8][2] = {{0, 1}, {0, 2}, {3, 4}, {4, 5}, {3, 6}, {6, 7}, {8, 9}, {9, 10}};
    for (int k = 0; k < 8; k++) {
        const int a = seg[k][0], b = seg[k][1];
        float ms = 0.f;
        const bool have = (t->tb.seg_mask >> a & 1u) && (t->tb.seg_mask >> b & 1u) && t->tb.ev_seg[a] && t->tb.ev_seg[b];
        h_ms8[k] = have && hipEventElapsedTime(&ms, t->tb.ev_seg[a], t->tb.ev_seg[b]) == hipSuccess ? (double)ms : -1.0;
    }
    // DFVO_STAGE_OFFSETS=1 (diagnostics): every recorded mark relative to mark 0, across the streams the marks were recorded on
    // -- shows which of the homography half and the five-point batch the bookkeeping kernel waited for
    static const bool offsets = getenv("DFVO_STAGE_OFFSETS") != nullptr;
    if (offsets && (t->tb.seg_mask & 1u) && t->tb.ev_seg[0]) {
        fprintf(stderr, "dfvo stage offsets [ms from the start of the homography half]:");
        for (int k = 1; k < TrackerBuffers::N_SEG; k++) {
            float ms = 0.f;
            if ((t->tb.seg_mask >> k & 1u) && t->tb.ev_seg[k] && hipEventElapsedTime(&ms, t->tb.ev_seg[0], t->tb.ev_seg[k]) == hipSuccess)
                fprintf(stderr, " %d:%.3f", k, ms);
        }
        fprintf(stderr, "\n");
    }
    return DFVO_OK;
}

void dfvo_tracker_destroy(dfvo_tracker* t) {
    if (!t) return;
    t->tb.release();
    t->pnp.release();
    t->rigid.release();
    t->bestn.release();
    if (t->d_flow) (void)hipFree(t->d_flow);
    if (t->d_diff) (void)hipFree(t->d_diff);
    if (t->d_depth) (void)hipFree(t->d_depth);
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


int dfvo_tracker_seed(dfvo_tracker* t, uint32_t seed) {
    DFVO_ARG_CHECK(t, "null tracker");
    int rc = enqueue_mt_seed(t->tb, seed, t->stream);
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));  // the shuffles read the key on another stream (see dfvo_pipeline_seed)
    return DFVO_OK;
}
int dfvo_tracker_set_rng_state(dfvo_tracker* t, const uint32_t* h) {
    DFVO_ARG_CHECK(t && h, "dfvo_tracker_set_rng_state: null argument");
    DFVO_HIP_CHECK(hipMemcpyAsync(t->tb.mt_state, h, 625 * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    return DFVO_OK;
}
int dfvo_tracker_get_rng_state(dfvo_tracker* t, uint32_t* h) {
    DFVO_ARG_CHECK(t && h, "dfvo_tracker_get_rng_state: null argument");
    DFVO_HIP_CHECK(hipMemcpyAsync(h, t->tb.mt_state, 625 * sizeof(uint32_t), hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    return DFVO_OK;
}

int dfvo_kp_local_bestn(dfvo_tracker* t, const float* h_flow, const float* h_diff, int H, int W, int num_row,
                        int num_col, int num_bestN, float thre, double* h_kp1, double* h_kp2, int* n_out,
                        int* good_kp_found) {
    return dfvo_kp_local_bestn_ex(t, h_flow, h_diff, H, W, num_row, num_col, num_bestN, thre, DFVO_KP_SCORE_FLOW, h_kp1, h_kp2,
                                  n_out, good_kp_found);
}

int dfvo_kp_local_bestn_ex(dfvo_tracker* t, const float* h_flow, const float* h_diff, int H, int W, int num_row,
                           int num_col, int num_bestN, float thre, int score_method, double* h_kp1, double* h_kp2,
                           int* n_out, int* good_kp_found) {
    DFVO_ARG_CHECK(t && h_flow && h_diff && h_kp1 && h_kp2 && n_out && good_kp_found && H > 0 && W > 0,
                   "dfvo_kp_local_bestn: bad argument");
    DFVO_ARG_CHECK(score_method == DFVO_KP_SCORE_FLOW || score_method == DFVO_KP_SCORE_FLOW_RATIO,
                   "dfvo_kp_local_bestn_ex: score_method must be DFVO_KP_SCORE_FLOW or DFVO_KP_SCORE_FLOW_RATIO");
    const size_t px = (size_t)H * W;
    if (px > t->flow_cap) {
        if (t->d_flow) (void)hipFree(t->d_flow);
        if (t->d_diff) (void)hipFree(t->d_diff);
        t->flow_cap = px;
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_flow, sizeof(float) * 2 * px));
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_diff, sizeof(float) * px));
    }
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_flow, h_flow, sizeof(float) * 2 * px, hipMemcpyHostToDevice, t->stream));
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_diff, h_diff, sizeof(float) * px, hipMemcpyHostToDevice, t->stream));
    int rc = enqueue_local_bestn(t->tb, t->d_flow, t->d_diff, H, W, num_row, num_col, num_bestN, thre, t->stream, score_method);
    if (rc != DFVO_OK) return rc;
    int info[3];
    DFVO_HIP_CHECK(hipMemcpyAsync(info, t->tb.kp_info, sizeof(info), hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    *n_out = info[0];
    *good_kp_found = info[1];
    if (info[1] && info[0] > 0) {
        DFVO_HIP_CHECK(hipMemcpy(h_kp1, t->tb.kp_ref, sizeof(double) * 2 * info[0], hipMemcpyDeviceToHost));
        DFVO_HIP_CHECK(hipMemcpy(h_kp2, t->tb.kp_cur, sizeof(double) * 2 * info[0], hipMemcpyDeviceToHost));
    }
    return DFVO_OK;
}

int dfvo_kp_bestn(dfvo_tracker* t, const float* h_flow, const float* h_diff, int H, int W, int num_bestN, double* h_kp1,
                  double* h_kp2, int* n_out) {
    DFVO_ARG_CHECK(t && h_flow && h_diff && h_kp1 && h_kp2 && n_out && H > 0 && W > 0 && num_bestN >= 1,
                   "dfvo_kp_bestn: bad argument");
    const size_t px = (size_t)H * W;
    if (px > t->flow_cap) {
        if (t->d_flow) (void)hipFree(t->d_flow);
        if (t->d_diff) (void)hipFree(t->d_diff);
        t->flow_cap = px;
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_flow, sizeof(float) * 2 * px));
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_diff, sizeof(float) * px));
    }
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_flow, h_flow, sizeof(float) * 2 * px, hipMemcpyHostToDevice, t->stream));
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_diff, h_diff, sizeof(float) * px, hipMemcpyHostToDevice, t->stream));
    int rc = enqueue_bestn_flow_kp(t->bestn, t->d_flow, t->d_diff, H, W, num_bestN, t->stream);
    if (rc != DFVO_OK) return rc;
    int n = 0;
    DFVO_HIP_CHECK(hipMemcpyAsync(&n, t->bestn.count + 1, sizeof(int), hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    *n_out = n;
    if (n > 0) {
        DFVO_HIP_CHECK(hipMemcpy(h_kp1, t->bestn.kp, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
        DFVO_HIP_CHECK(hipMemcpy(h_kp2, t->bestn.kp + 2 * (size_t)num_bestN, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
    }
    return DFVO_OK;
}

int dfvo_kp_sampled(dfvo_tracker* t, const float* h_flow, int H, int W, int y0, int y1, int x0, int x1, const int* h_idx,
                    int n, double* h_kp1, double* h_kp2) {
    DFVO_ARG_CHECK(t && h_flow && h_idx && h_kp1 && h_kp2 && H > 0 && W > 0 && n >= 0, "dfvo_kp_sampled: bad argument");
    const long long cells = (long long)(y1 - y0) * (x1 - x0);
    for (int i = 0; i < n; ++i) DFVO_ARG_CHECK(h_idx[i] >= 0 && h_idx[i] < cells, "dfvo_kp_sampled: index outside the cropped grid");
    const size_t px = (size_t)H * W;
    if (px > t->flow_cap) {
        if (t->d_flow) (void)hipFree(t->d_flow);
        if (t->d_diff) (void)hipFree(t->d_diff);
        t->flow_cap = px;
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_flow, sizeof(float) * 2 * px));
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_diff, sizeof(float) * px));
    }
    if (n == 0) return DFVO_OK;
    int* d_idx = nullptr;
    double* d_kp = nullptr;
    DFVO_HIP_CHECK(hipMalloc((void**)&d_idx, sizeof(int) * n));
    if (hipMalloc((void**)&d_kp, sizeof(double) * 4 * n) != hipSuccess) {
        (void)hipFree(d_idx);
        set_last_error("dfvo_kp_sampled: hipMalloc failed");
        return DFVO_ERR_HIP;
    }
    hipError_t e = hipMemcpyAsync(t->d_flow, h_flow, sizeof(float) * 2 * px, hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_idx, h_idx, sizeof(int) * n, hipMemcpyHostToDevice, t->stream);
    int rc = e == hipSuccess ? enqueue_kp_sampled(t->d_flow, H, W, y0, y1, x0, x1, d_idx, n, d_kp, d_kp + 2 * n, t->stream)
                             : DFVO_ERR_HIP;
    if (rc == DFVO_OK) {
        e = hipMemcpyAsync(h_kp1, d_kp, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, t->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h_kp2, d_kp + 2 * n, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, t->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
        if (e != hipSuccess) {
            set_last_error(std::string("dfvo_kp_sampled: ") + hipGetErrorString(e));
            rc = DFVO_ERR_HIP;
        }
    }
    (void)hipStreamSynchronize(t->stream);
    (void)hipFree(d_idx);
    (void)hipFree(d_kp);
    return rc;
}

int dfvo_kp_rigid_flow(dfvo_tracker* t, const float* h_flow, const float* h_flow_diff, const float* h_raw_depth, int H,
                       int W, const dfvo_rigid_kp_cfg* cfg, const float* h_rigid_diff_override, double* h_kp1_best,
                       double* h_kp2_best, double* h_kp1_uniform, double* h_kp2_uniform, int* n_out,
                       float* h_rigid_flow_diff) {
    DFVO_ARG_CHECK(t && h_flow && h_flow_diff && h_raw_depth && cfg && h_kp1_best && h_kp2_best && h_kp1_uniform &&
                       h_kp2_uniform && n_out && H > 0 && W > 0,
                   "dfvo_kp_rigid_flow: bad argument");
    DFVO_ARG_CHECK(cfg->score_method == 0 || cfg->score_method == 1, "dfvo_kp_rigid_flow: score_method");
    const size_t px = (size_t)H * W;
    if (px > t->flow_cap) {
        if (t->d_flow) (void)hipFree(t->d_flow);
        if (t->d_diff) (void)hipFree(t->d_diff);
        t->flow_cap = px;
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_flow, sizeof(float) * 2 * px));
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_diff, sizeof(float) * px));
    }
    RigidKpConfig rc;
    rc.num_row = cfg->num_row;
    rc.num_col = cfg->num_col;
    rc.num_bestN = cfg->num_bestN;
    rc.rigid_thre = (float)cfg->rigid_flow_thre;
    rc.opt_thre = (float)cfg->optical_flow_thre;
    rc.score_rigid = cfg->score_method;
    for (int i = 0; i < 9; i++) {
        rc.K[i] = (float)cfg->K[i];
        rc.Kinv[i] = (float)cfg->Kinv[i];
    }
    for (int i = 0; i < 16; i++) rc.T[i] = (float)cfg->T_ref_to_cur[i];
    DFVO_ARG_CHECK(rc.num_row > 0 && rc.num_col > 0, "dfvo_kp_rigid_flow: grid");
    const int cells = rc.num_row * rc.num_col;
    int rcode = t->rigid.ensure(H, W, cells, rc.num_bestN / cells > 0 ? rc.num_bestN / cells : 1,
                                (H / rc.num_row + 2) * (W / rc.num_col + 2));
    if (rcode != DFVO_OK) return rcode;
    hipStream_t s = t->stream;
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_flow, h_flow, sizeof(float) * 2 * px, hipMemcpyHostToDevice, s));
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_diff, h_flow_diff, sizeof(float) * px, hipMemcpyHostToDevice, s));
    DFVO_HIP_CHECK(hipMemcpyAsync(t->rigid.depth32, h_raw_depth, sizeof(float) * px, hipMemcpyHostToDevice, s));
    const float* ovr = nullptr;
    if (h_rigid_diff_override) {
        DFVO_HIP_CHECK(hipMemcpyAsync(t->rigid.rdiff, h_rigid_diff_override, sizeof(float) * px, hipMemcpyHostToDevice, s));
        ovr = t->rigid.rdiff;
    }
    rcode = enqueue_rigid_flow_kp(t->rigid, t->d_flow, t->d_diff, t->rigid.depth32, H, W, rc, ovr, s);
    if (rcode != DFVO_OK) return rcode;
    int info[8];
    DFVO_HIP_CHECK(hipMemcpyAsync(info, t->rigid.info, sizeof(info), hipMemcpyDeviceToHost, s));
    DFVO_HIP_CHECK(hipStreamSynchronize(s));
    const int n = info[0];
    DFVO_ARG_CHECK(n == info[4], "dfvo_kp_rigid_flow: internal count mismatch");
    *n_out = n;
    if (n > 0) {
        const size_t sc = (size_t)t->rigid.sel_cap * 2, bytes = sizeof(double) * 2 * n;
        DFVO_HIP_CHECK(hipMemcpy(h_kp1_best, t->rigid.kp, bytes, hipMemcpyDeviceToHost));
        DFVO_HIP_CHECK(hipMemcpy(h_kp2_best, t->rigid.kp + sc, bytes, hipMemcpyDeviceToHost));
        DFVO_HIP_CHECK(hipMemcpy(h_kp1_uniform, t->rigid.kp + 2 * sc, bytes, hipMemcpyDeviceToHost));
        DFVO_HIP_CHECK(hipMemcpy(h_kp2_uniform, t->rigid.kp + 3 * sc, bytes, hipMemcpyDeviceToHost));
    }
    if (h_rigid_flow_diff)
        DFVO_HIP_CHECK(hipMemcpy(h_rigid_flow_diff, t->rigid.rdiff, sizeof(float) * px, hipMemcpyDeviceToHost));
    return DFVO_OK;
}

static int stage_kp(dfvo_tracker* t, const double* h_a, const double* h_b, int n) {
    int rc = t->tb.ensure_kp(n > 16 ? n : 16, 1, 1);
    if (rc != DFVO_OK) return rc;
    if (n > 0) {
        DFVO_HIP_CHECK(hipMemcpyAsync(t->tb.kp_ref, h_a, sizeof(double) * 2 * n, hipMemcpyHostToDevice, t->stream));
        DFVO_HIP_CHECK(hipMemcpyAsync(t->tb.kp_cur, h_b, sizeof(double) * 2 * n, hipMemcpyHostToDevice, t->stream));
    }
    int info[3] = {n, 1, 0};
    DFVO_HIP_CHECK(hipMemcpyAsync(t->tb.kp_info, info, sizeof(info), hipMemcpyHostToDevice, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));  // `info` is a stack buffer
    return DFVO_OK;
}

int dfvo_compute_pose_2d2d(dfvo_tracker* t, const double* h_kp_ref, const double* h_kp_cur, int n,
                           const dfvo_pose2d2d_cfg* cfg, dfvo_pose2d2d_out* out, uint8_t* h_inliers) {
    DFVO_ARG_CHECK(t && h_kp_ref && h_kp_cur && cfg && out && h_inliers && n >= 0, "dfvo_compute_pose_2d2d: bad argument");
    int rc = stage_kp(t, h_kp_ref, h_kp_cur, n);
    if (rc != DFVO_OK) return rc;
    PoseConfig pc;
    rc = dfvo_pose_config_from(cfg, &pc);
    if (rc != DFVO_OK) return rc;
    rc = enqueue_compute_pose_2d2d(t->tb, n, pc, t->stream);
    if (rc != DFVO_OK) return rc;
    return dfvo_pose_fetch(t, n, cfg, out, h_inliers);
}

int dfvo_set_sklearn_compat(const char* version) { return dfvo::set_sklearn_compat(version); }

// h_depth: the H x W map, or (per_kp) its value at every keypoint's truncated kp2 pixel; h_rng625 (optional): RandomState to run
// under, replaced by the advanced state
static int find_scale_impl(dfvo_tracker* t, const double* h_kp1, const double* h_kp2, int n, const double* h_T21,
                           const double* h_depth, bool per_kp, int H, int W, const dfvo_scale_cfg* cfg, uint32_t* h_rng625,
                           double* scale, int* h_info) {
    DFVO_ARG_CHECK(t && h_kp1 && h_kp2 && h_T21 && h_depth && cfg && scale && n >= 0 && H > 0 && W > 0,
                   "dfvo_find_scale_from_depth: bad argument");
    DFVO_ARG_CHECK(cfg->min_samples >= 1 && cfg->min_samples <= 8, "dfvo_find_scale_from_depth: min_samples in [1,8]");
    int rc = stage_kp(t, h_kp1, h_kp2, n);
    if (rc != DFVO_OK) return rc;
    if (h_rng625)
        DFVO_HIP_CHECK(hipMemcpyAsync(t->tb.mt_state, h_rng625, 625 * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
    const size_t px = per_kp ? (size_t)(n > 0 ? n : 1) : (size_t)H * W;
    if (px > t->depth_cap) {
        if (t->d_depth) (void)hipFree(t->d_depth);
        t->depth_cap = px;
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_depth, sizeof(double) * px));
    }
    if (!(per_kp && n == 0))  // (a zero-length per-keypoint array has no element to read)
        DFVO_HIP_CHECK(hipMemcpyAsync(t->d_depth, h_depth, sizeof(double) * px, hipMemcpyHostToDevice, t->stream));
    DFVO_HIP_CHECK(hipMemcpyAsync(t->d_small, h_T21, 16 * sizeof(double), hipMemcpyHostToDevice, t->stream));
    ScaleConfig sc;
    sc.cx = cfg->cx;
    sc.cy = cfg->cy;
    sc.fx = cfg->fx;
    sc.fy = cfg->fy;
    sc.min_samples = cfg->min_samples;
    sc.max_trials = cfg->max_trials;
    sc.stop_prob = cfg->stop_prob;
    sc.thre = cfg->thre;
    DFVO_ARG_CHECK(cfg->method == DFVO_SCALE_DEPTH_RATIO || cfg->method == DFVO_SCALE_ABS_DIFF,
                   "dfvo_find_scale_from_depth: unknown method");
    sc.method = cfg->method;
    rc = enqueue_find_scale(t->tb, n, t->d_small, t->d_depth, H, W, sc, t->stream, nullptr, false, per_kp);
    if (rc != DFVO_OK) return rc;
    ScaleResult sr;
    DFVO_HIP_CHECK(hipMemcpyAsync(&sr, t->tb.scale_out, sizeof(sr), hipMemcpyDeviceToHost, t->stream));
    if (h_rng625)
        DFVO_HIP_CHECK(hipMemcpyAsync(h_rng625, t->tb.mt_state, 625 * sizeof(uint32_t), hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    *scale = sr.scale;
    if (h_info) {
        h_info[0] = sr.n_valid;
        h_info[1] = sr.n_trials;
        h_info[2] = sr.n_inliers;
        h_info[3] = sr.status;
    }
    return DFVO_OK;
}

int dfvo_find_scale_from_depth(dfvo_tracker* t, const double* h_kp1, const double* h_kp2, int n, const double* h_T21,
                               const double* h_depth, int H, int W, const dfvo_scale_cfg* cfg, double* scale,
                               int* h_info) {
    return find_scale_impl(t, h_kp1, h_kp2, n, h_T21, h_depth, false, H, W, cfg, nullptr, scale, h_info);
}

int dfvo_find_scale_from_depth_at_kp(dfvo_tracker* t, const double* h_kp1, const double* h_kp2, int n, const double* h_T21,
                                     const double* h_depth_at_kp2, int H, int W, const dfvo_scale_cfg* cfg,
                                     uint32_t* h_rng625, double* scale, int* h_info) {
    return find_scale_impl(t, h_kp1, h_kp2, n, h_T21, h_depth_at_kp2, true, H, W, cfg, h_rng625, scale, h_info);
}

int dfvo_ransac_regressor(dfvo_tracker* t, const double* h_x, const double* h_y, int n, const dfvo_scale_cfg* cfg,
                          double* coef, int* h_info) {
    DFVO_ARG_CHECK(t && h_x && cfg && coef && n >= 1, "dfvo_ransac_regressor: bad argument");
    DFVO_ARG_CHECK(cfg->min_samples >= 1 && cfg->min_samples <= 8, "dfvo_ransac_regressor: min_samples in [1,8]");
    int rc = t->tb.ensure_kp(n > 16 ? n : 16, 1, 1);
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(hipMemcpyAsync(t->tb.ratios, h_x, sizeof(double) * n, hipMemcpyHostToDevice, t->stream));
    if (h_y)
        DFVO_HIP_CHECK(hipMemcpyAsync(t->tb.ratios + t->tb.kp_cap, h_y, sizeof(double) * n, hipMemcpyHostToDevice, t->stream));
    ScaleConfig sc;
    sc.cx = sc.cy = sc.fx = sc.fy = 0;
    sc.min_samples = cfg->min_samples;
    sc.max_trials = cfg->max_trials;
    sc.stop_prob = cfg->stop_prob;
    sc.thre = cfg->thre;
    sc.method = h_y ? 1 : 0;
    rc = enqueue_ransac_regressor(t->tb, n, h_y == nullptr, sc, t->stream);
    if (rc != DFVO_OK) return rc;
    ScaleResult sr;
    DFVO_HIP_CHECK(hipMemcpyAsync(&sr, t->tb.scale_out, sizeof(sr), hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    *coef = sr.scale;
    if (h_info) {
        h_info[0] = sr.n_valid;
        h_info[1] = sr.n_trials;
        h_info[2] = sr.n_inliers;
        h_info[3] = sr.status;
    }
    return DFVO_OK;
}

static int pose_3d2d_impl(dfvo_tracker* t, const double* h_kp1, const double* h_kp2, int n, const double* h_depth, bool per_kp,
                          int H, int W, const dfvo_pose3d2d_cfg* cfg, uint32_t* h_rng625, dfvo_pose3d2d_out* out, uint8_t* h_keep) {
    DFVO_ARG_CHECK(t && h_kp1 && h_kp2 && h_depth && cfg && out && n >= 0 && H > 0 && W > 0,
                   "dfvo_compute_pose_3d2d: bad argument");
    DFVO_ARG_CHECK(cfg->repeat >= 1 && cfg->repeat <= MAX_REP && cfg->iters >= 1, "dfvo_compute_pose_3d2d: repeat/iters");
    int rc = stage_kp(t, h_kp1, h_kp2, n);  // kp1 -> tb.kp_ref, kp2 -> tb.kp_cur
    if (rc != DFVO_OK) return rc;
    if (h_rng625)
        DFVO_HIP_CHECK(hipMemcpyAsync(t->tb.mt_state, h_rng625, 625 * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
    const size_t px = per_kp ? (size_t)(n > 0 ? n : 1) : (size_t)H * W;
    if (px > t->depth_cap) {
        if (t->d_depth) (void)hipFree(t->d_depth);
        t->depth_cap = px;
        DFVO_HIP_CHECK(hipMalloc((void**)&t->d_depth, sizeof(double) * px));
    }
    if (!(per_kp && n == 0))  // (a zero-length per-keypoint array has no element to read)
        DFVO_HIP_CHECK(hipMemcpyAsync(t->d_depth, h_depth, sizeof(double) * px, hipMemcpyHostToDevice, t->stream));
    PnpConfig pc;
    pc.fx = cfg->fx;
    pc.fy = cfg->fy;
    pc.cx = cfg->cx;
    pc.cy = cfg->cy;
    for (int i = 0; i < 9; i++) pc.inv_K[i] = cfg->Kinv[i];
    pc.min_depth = cfg->min_depth;
    pc.max_depth = cfg->max_depth;
    pc.repeat = cfg->repeat;
    pc.iters = cfg->iters;
    pc.reproj_thre = cfg->reproj_thre;
    rc = enqueue_compute_pose_3d2d(t->pnp, t->tb.mt_state, t->tb.kp_ref, t->tb.kp_cur, nullptr, n, t->d_depth, H, W, pc,
                                   t->stream, per_kp);
    if (rc != DFVO_OK) return rc;
    PnpResult res;
    DFVO_HIP_CHECK(hipMemcpyAsync(&res, t->pnp.result, sizeof(res), hipMemcpyDeviceToHost, t->stream));
    if (h_keep && n > 0) DFVO_HIP_CHECK(hipMemcpyAsync(h_keep, t->pnp.keep, (size_t)n, hipMemcpyDeviceToHost, t->stream));
    if (h_rng625)
        DFVO_HIP_CHECK(hipMemcpyAsync(h_rng625, t->tb.mt_state, 625 * sizeof(uint32_t), hipMemcpyDeviceToHost, t->stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(t->stream));
    out->found = res.found;
    out->best_inliers = res.best_inliers;
    out->n_filtered = res.n_filtered;
    out->status = res.status;
    for (int i = 0; i < 3; i++) {
        out->rvec[i] = res.rvec[i];
        out->tvec[i] = res.tvec[i];
    }
    for (int i = 0; i < 9; i++) out->R[i] = res.R[i];
    return DFVO_OK;
}

int dfvo_compute_pose_3d2d(dfvo_tracker* t, const double* h_kp1, const double* h_kp2, int n, const double* h_depth,
                           int H, int W, const dfvo_pose3d2d_cfg* cfg, dfvo_pose3d2d_out* out, uint8_t* h_keep) {
    return pose_3d2d_impl(t, h_kp1, h_kp2, n, h_depth, false, H, W, cfg, nullptr, out, h_keep);
}

int dfvo_compute_pose_3d2d_at_kp(dfvo_tracker* t, const double* h_kp1, const double* h_kp2, int n, const double* h_depth_at_kp1,
                                 int H, int W, const dfvo_pose3d2d_cfg* cfg, uint32_t* h_rng625, dfvo_pose3d2d_out* out,
                                 uint8_t* h_keep) {
    return pose_3d2d_impl(t, h_kp1, h_kp2, n, h_depth_at_kp1, true, H, W, cfg, h_rng625, out, h_keep);
}


// rows / first / poses on the DEVICE (e.g. the output of the RCCL all-gather); *h_bad_row = first status-2 row or -1
int dfvo_compose_trajectory_device(const double* d_rows, int n, const double* d_first, double* d_poses, int* h_bad_row,
                                   void* stream) {
    DFVO_ARG_CHECK(n >= 0 && d_poses && h_bad_row, "dfvo_compose_trajectory_device: bad argument");
    hipStream_t s = (hipStream_t)stream;
    int* d_bad = nullptr;
    DFVO_HIP_CHECK(hipMalloc((void**)&d_bad, sizeof(int)));
    int rc = enqueue_compose_trajectory(d_rows, n, d_first, d_poses, d_bad, s);
    hipError_t e = hipStreamSynchronize(s);
    if (rc == DFVO_OK && e == hipSuccess) e = hipMemcpy(h_bad_row, d_bad, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(e);
    return DFVO_OK;
}

// host arrays in and out
int dfvo_compose_trajectory(const double* h_rows, int n, const double* h_first, double* h_poses, int* h_bad_row) {
    DFVO_ARG_CHECK(n >= 0 && h_poses && h_bad_row && (n == 0 || h_rows), "dfvo_compose_trajectory: bad argument");
    double *d_rows = nullptr, *d_first = nullptr, *d_poses = nullptr;
    DFVO_HIP_CHECK(hipMalloc((void**)&d_rows, sizeof(double) * 17 * (size_t)(n > 0 ? n : 1)));
    DFVO_HIP_CHECK(hipMalloc((void**)&d_poses, sizeof(double) * 16 * (size_t)(n + 1)));
    if (h_first) DFVO_HIP_CHECK(hipMalloc((void**)&d_first, sizeof(double) * 16));
    hipError_t e = n ? hipMemcpy(d_rows, h_rows, sizeof(double) * 17 * (size_t)n, hipMemcpyHostToDevice) : hipSuccess;
    if (e == hipSuccess && h_first) e = hipMemcpy(d_first, h_first, sizeof(double) * 16, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? dfvo_compose_trajectory_device(d_rows, n, d_first, d_poses, h_bad_row, nullptr) : DFVO_ERR_HIP;
    if (rc == DFVO_OK) e = hipMemcpy(h_poses, d_poses, sizeof(double) * 16 * (size_t)(n + 1), hipMemcpyDeviceToHost);
    (void)hipFree(d_rows);
    (void)hipFree(d_poses);
    if (d_first) (void)hipFree(d_first);
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(e);
    return DFVO_OK;
}
}  // extern "C"
