// Device buffers and small state blocks of the tracker pipeline (solver_pipeline.hip).
#pragma once
#include <stddef.h>

#include "solver.h"

namespace dfvo {

struct PoseState {
    int n;
    int best_cnt;
    int num_valid;
    int have_best;
    int major_valid;
    int cheirality;
    int valid_case;
    int h_found;
    double h_gric;
    double best_E[9];
    double R[9];
    double t[3];
    int rep_cnt[8];
    int rep_valid[8];
    double rep_gric[8];
};

struct ScaleResult {
    double scale;
    int n_valid;
    int n_trials;
    int n_inliers;
    int status;  // 1 ok, 0 too few valid points (scale -1), -1 no consensus set (sklearn raises)
    int best_is_a;
};

struct PoseConfig {
    double fx, cx, cy;
    double reproj_thre;
    int repeat;
    int max_iters;
    double KinvT[9], Kinv[9];
    int validity = 0;           // e_tracker.validity.method: 0 GRIC, 1 flow, 2 homo_ratio (E_tracker.py:182-194, 243-250)
    double validity_thre = 0;   // flow: mean keypoint displacement [px] above which the pair is tracked at all;
                                // homo_ratio: a repeat is valid while H inliers / (H + E inliers) stays below it
};

struct ScaleConfig {
    double cx, cy, fx, fy;
    int min_samples, max_trials;
    double stop_prob, thre;
    int method = 0;  // scale_recovery.ransac.method: 0 depth_ratio, 1 abs_diff (E_tracker.py:626-635)
};

constexpr int MAX_REP = 8;
constexpr int NUM_REP_STREAMS = 4;  // side streams created per tracker; two are used (see TrackerBuffers::init)
static inline int rep_stream_count() { return NUM_REP_STREAMS; }

// PnpTracker.compute_pose_3d2d (solver_pnp.hip)
struct PnpConfig {
    double fx, fy, cx, cy;
    double inv_K[9];  // Intrinsics.inv_mat, row-major
    double min_depth, max_depth;
    int repeat;  // number of shuffled solvePnPRansac runs (cfg.pnp_tracker.ransac.repeat, or 3)
    int iters;   // iterationsCount
    double reproj_thre;
};
struct PnpRepOut {
    int flag;       // solvePnPRansac returned true
    int n_inliers;  // inlier.shape[0]
    int status;     // 1 ok, 0 no model, -2 planar initialisation branch (not implemented)
    int lm_iters;
    double rvec[3], tvec[3];
};
struct PnpResult {
    int found;         // len(best_rt) != 0
    int best_inliers;
    int n_filtered;    // keypoints that survived the in-image / depth-range masks
    int status;        // 0, or -2 when a repeat met the planar branch
    double rvec[3], tvec[3];
    double R[9];       // cv2.Rodrigues(rvec)
};
struct PnpBuffers {
    int cap = 0, iters_cap = 0;
    int* info = nullptr;
    double *fk1 = nullptr, *fk2 = nullptr, *xyz = nullptr;
    int* perm = nullptr;
    float *obj = nullptr, *img = nullptr;
    RansacState* state = nullptr;
    int* idx = nullptr;
    double* models = nullptr;
    int *nmodels = nullptr, *counts = nullptr;
    uint8_t *mask = nullptr, *keep = nullptr;  // keep[i]: input keypoint i survived the filters
    float* pts5 = nullptr;
    PnpRepOut* rep_out = nullptr;
    PnpResult* result = nullptr;
    int ensure(int n, int iters);
    void release();
};

struct TrackerBuffers {
    RansacWorkspace ws_h, ws_e;          // ws_e: stand-alone findEssentialMat / recoverPose calls
    RansacWorkspace ws_rep[MAX_REP];     // one workspace per repeated findEssentialMat (run concurrently)
    hipStream_t s_rep[MAX_REP] = {};
    int n_rep_owned = 0;  // s_rep[0 .. n_rep_owned) are distinct streams this object destroys
    hipEvent_t ev_rep[MAX_REP] = {};
    hipEvent_t ev_fork = nullptr, ev_start = nullptr, ev_h = nullptr;
    // DFVO_TRACK_TRACE: device-side timestamps of the RNG-ordered chain (start of the shuffles, end of the five-point batch,
    // end of recoverPose, end of the scale stage); null unless tracing
    hipEvent_t ev_t[4] = {nullptr, nullptr, nullptr, nullptr};
    // Stage timestamps for the reference's Timer sub-keys (E_tracker.py:197-296,597-638), created on demand by
    // enable_stage_timing() (the drop-in mirrors' tracker; the fused pipeline leaves them null).  Marks: 0 start of the
    // homography part | 1 findHomography done | 2 GRIC-H done | 3 first shuffle | 4 five-point batch done | 5 GRIC-E done |
    // 6 validity bookkeeping done | 7 recoverPose done | 8 scale stage start | 9 triangulation + depth ratios done |
    // 10 scale RANSAC done.  seg_mask: the marks recorded since the last homography part / scale stage began.
    static constexpr int N_SEG = 11;
    hipEvent_t ev_seg[N_SEG] = {};
    unsigned seg_mask = 0;
    int enable_stage_timing();
    int mark(int i, hipStream_t s);
    bool shared = false;  // streams / events / RandomState borrowed from another TrackerBuffers (see share_from)
    uint32_t* mt_state = nullptr;  // numpy RandomState: key[624], pos
    int* kp_info = nullptr;        // [n, good_kp_found, regions]
    int* kp_total = nullptr;
    int e_pre_iters = 0;  // > 0: enqueue_pose_h_part drew the five-point sampler's first chunk ahead, for this iteration budget
    PoseState* pose = nullptr;
    double* small = nullptr;
    double h_small[18] = {};       // host copy of the 18 intrinsics doubles held in `small` (uploaded only when they change)
    bool small_valid = false;
    ScaleResult* scale_out = nullptr;
    int* winner = nullptr;
    size_t winner_cap = 0;
    unsigned short* lidx = nullptr;
    size_t lidx_cap = 0;
    float* ratio_map = nullptr;  // flow_diff / |flow| per pixel (local_bestN score_method 'flow_ratio')
    size_t ratio_cap = 0;
    // keypoint-sized buffers
    double *kp_ref = nullptr, *kp_cur = nullptr, *pa = nullptr, *pb = nullptr, *res = nullptr, *z2 = nullptr,
           *ratios = nullptr;
    int *perm = nullptr, *cell_count = nullptr, *cell_sel = nullptr, *pix = nullptr, *scratch = nullptr;
    uint8_t *best_inliers = nullptr, *inl_a = nullptr, *inl_b = nullptr;
    int kp_cap = 0, sel_cap = 0;
    // rep0 / rep1: side streams chosen by the caller (the fused pipeline hands out streams by dispatch pipe); null = create
    int init(hipStream_t rep0 = nullptr, hipStream_t rep1 = nullptr);
    // replaces the side streams of an idle, non-shared buffer set by two streams the caller chose by dispatch pipe (the frame
    // session, session.hip); this object owns and destroys them from then on
    int rebind_streams(hipStream_t rep0, hipStream_t rep1);
    // second and further buffer sets of the fused pipeline: own keypoint / RANSAC workspaces, but the numpy
    // RandomState and the (serialised anyway) RNG-side streams and events of `first`
    int init_shared(const TrackerBuffers& first);
    int ensure_kp(int cap, int cells, int n_best);
    void release_kp();
    void release();
};

// score_method: 0 'flow' (the consistency map itself), 1 'flow_ratio' (map / |flow|), kp_selection.py:137-141,151-156
int enqueue_local_bestn(TrackerBuffers& tb, const float* d_flow, const float* d_diff, int H, int W, int num_row,
                        int num_col, int num_bestN, float thre, hipStream_t s, int score_method = 0);
// bestN_flow_kp (kp_selection.py:33-71): whole-image argpartition
struct BestNBuffers {
    float* key_base = nullptr;   // keys carried along with the index array, with slack on either side
    int *tosort = nullptr, *map = nullptr, *Lpos = nullptr, *Rpos = nullptr, *count = nullptr;  // count[1] = result size
    double* kp = nullptr;        // [kp1 | kp2], N x 2 doubles each
    size_t cap = 0;
    int kp_cap = 0;
    void release();
};
int enqueue_bestn_flow_kp(BestNBuffers& bb, const float* d_flow, const float* d_diff, int H, int W, int N, hipStream_t s);

// rigid-flow keypoints (E_tracker.py:645-705 kp_selection_good_depth)
struct RigidKpConfig {
    int num_row, num_col, num_bestN;
    float rigid_thre, opt_thre;
    int score_rigid;              // 1: score = rigid-flow distance, 0: forward-backward distance
    float K[9], Kinv[9], T[16];   // float32 intrinsics, their inverse, the ref -> cur motion (row-major)
};
struct RigidKpBuffers {
    float *depth32 = nullptr, *rdiff = nullptr, *mats = nullptr;
    int *cell_count = nullptr, *cell_sel = nullptr, *cell_sel_uni = nullptr, *info = nullptr, *zero = nullptr;
    unsigned short* lidx = nullptr;
    double* kp = nullptr;         // [4][sel_cap][2]: kp1 best, kp2 best, kp1 uniform, kp2 uniform
    size_t px_cap = 0, lidx_cap = 0;
    int sel_cap = 0;
    int ensure(int H, int W, int cells, int n_best, int cap);
    void release();
};
int enqueue_rigid_flow_kp(RigidKpBuffers& rb, const float* d_flow, const float* d_odiff, const float* d_depth32, int H,
                          int W, const RigidKpConfig& cfg, const float* d_rdiff_override, hipStream_t s);
int enqueue_kp_sampled(const float* d_flow, int H, int W, int y0, int y1, int x0, int x1, const int* d_idx, int n,
                       double* d_kp1, double* d_kp2, hipStream_t s);
int enqueue_mt_seed(TrackerBuffers& tb, uint32_t seed, hipStream_t s);
// d_T21 (optional): 16 doubles that receive the inverse of the accepted pose (input of the scale stage)
int enqueue_compute_pose_2d2d(TrackerBuffers& tb, int n_host, const PoseConfig& cfg, hipStream_t s,
                              double* d_T21 = nullptr);
// update_global_pose over a gathered sequence in one launch (rows [n][17] -> poses [n+1][16]); d_bad[0] = first status-2 row or -1
int enqueue_compose_trajectory(const double* d_rows, int n, const double* d_first, double* d_poses, int* d_bad, hipStream_t s);
int enqueue_pose_h_part(TrackerBuffers& tb, int n_bound, const PoseConfig& cfg, hipStream_t sh);
int enqueue_pose_e_part(TrackerBuffers& tb, int n_host, const PoseConfig& cfg, hipStream_t s, double* d_T21);
int enqueue_mt_shuffle(uint32_t* mt_state, const int* d_n, int n_host, int repeat, int perm_stride, int* perm,
                       hipStream_t s);
// depth_per_kp: d_depth holds, per keypoint, the depth map's value at that keypoint's kp1 pixel (truncated, negative indices
// wrapped as numpy does) -- [n_host] doubles instead of the H x W map
int enqueue_compute_pose_3d2d(PnpBuffers& pb, uint32_t* mt_state, const double* d_kp1, const double* d_kp2,
                              const int* d_n, int n_host, const double* d_depth, int H, int W, const PnpConfig& cfg,
                              hipStream_t s, bool depth_per_kp = false);
// d_gate (optional): device PoseState whose zero translation suppresses the whole stage (no RandomState draws)
// depth_per_kp: d_depth holds, per keypoint, the depth map's value at that keypoint's (truncated) kp2 pixel -- [n_host]
// doubles instead of the H x W map (only those pixels are ever read)
int enqueue_find_scale(TrackerBuffers& tb, int n_host, const double* d_T21, const double* d_depth, int H, int W,
                       const ScaleConfig& cfg, hipStream_t s, const PoseState* d_gate = nullptr, bool prepared = false,
                       bool depth_per_kp = false);
int enqueue_scale_prepare(TrackerBuffers& tb, int H, int W);
int enqueue_ransac_regressor(TrackerBuffers& tb, int n, bool y_is_ones, const ScaleConfig& cfg, hipStream_t s);
int set_sklearn_compat(const char* version);  // "0.20" (the reference's pin, default) | "0.22" and later

}  // namespace dfvo
