// Device-side RANSAC state + workspaces shared by the solver kernels and the tracker pipeline.
#pragma once
#include "dfvo_common.h"

namespace dfvo {

constexpr int MAX_E_BATCH = 8;  // problems per batched five-point RANSAC launch sequence
constexpr int E_WS = 106;  // per-hypothesis scratch doubles: EE 36 | b 39 | c 11 | roots re/im 20

struct RansacState {
    uint64_t rng_state;   // cv::RNG state carried across chunks
    int niters;           // current (shrinking) iteration budget
    int iter;             // iterations replayed so far
    int max_good;         // best inlier count so far
    int best_iter;        // hypothesis (iteration) of the best model
    int best_model;       // model index inside that hypothesis
    int done;             // replay passed niters: later chunks exit at once
    int subset_fail_at;   // iteration at which getSubset failed (-1: never)
    int found;            // 1 when a model was accepted
};

struct RansacWorkspace {
    RansacState* state = nullptr;
    double *pts_a = nullptr, *pts_b = nullptr;    // staged input points [n][2]
    double *norm_a = nullptr, *norm_b = nullptr;  // K-normalised points (E, recoverPose)
    float *f_a = nullptr, *f_b = nullptr;         // float points (homography)
    int* idx = nullptr;                           // subset indices [iters][5]
    double* ws = nullptr;                         // five-point scratch [iters][E_WS]
    int* ok = nullptr;
    double* models = nullptr;                     // [iters][10][9]
    int* nmodels = nullptr;
    int* counts = nullptr;                        // [iters][10]
    uint8_t* mask = nullptr;                      // [n]
    double* out = nullptr;                        // 64 doubles of small results
    double* lm = nullptr;                         // LM / recoverPose scratch
    int* cidx = nullptr;                          // compacted inlier indices
    int cap_n = 0, cap_iters = 0;
    int ensure(int n, int max_iters);
    void release();
};

int enqueue_find_essential(RansacWorkspace& w, const double* d_pts1, const double* d_pts2, int n, double focal,
                           double ppx, double ppy, double prob, double threshold, int max_iters, hipStream_t s);
int enqueue_find_essential_batch(RansacWorkspace* w, const double* const* d_pts1, const double* const* d_pts2, int nrep,
                                 int n, double focal, double ppx, double ppy, double prob, double threshold,
                                 int max_iters, hipStream_t s, const unsigned long long* rng_pre = nullptr);
// the first chunk's five-point subsets ahead of the call (a function of the point count alone): into w0.idx, the sampler's state
// behind them into *d_rng_pre; hand d_rng_pre to enqueue_find_essential_batch
int enqueue_e_subsets_prefetch(RansacWorkspace& w0, const int* d_n, int n_bound, int max_iters, unsigned long long* d_rng_pre,
                               hipStream_t s);
int enqueue_find_homography(RansacWorkspace& w, const double* d_pts1, const double* d_pts2, int n, double thr,
                            int max_iters, double confidence, hipStream_t s, const int* d_n = nullptr);
struct PoseState;
// optional tail of recoverPose in the fused pipeline (null pointers: plain recoverPose)
struct PoseFinish {
    PoseState* ps = nullptr;
    double* T21 = nullptr;
};
int enqueue_recover_pose(RansacWorkspace& w, const double* d_E, const double* d_pts1, const double* d_pts2, int n,
                         double focal, double ppx, double ppy, hipStream_t s, PoseFinish fin = PoseFinish());
int enqueue_triangulate(const double* d_P, const double* d_x1, const double* d_x2, int n, double* d_X4,
                        hipStream_t s);

}  // namespace dfvo
