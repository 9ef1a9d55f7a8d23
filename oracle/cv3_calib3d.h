/* ORACLE (test infrastructure only) -- see cv3_core.h for the status of this restatement.
 *
 * cv3_calib3d: the subset of OpenCV 3.4.3 modules/calib3d/src/{ptsetreg.cpp, five-point.cpp,
 * fundam.cpp, triangulate.cpp, levmarq.cpp, solvepnp.cpp, epnp.cpp, calibration.cpp} behind the cv2
 * calls of the reference trackers:
 *   cv2.findEssentialMat   libs/tracker/E_tracker.py:59-67,231-239
 *   cv2.recoverPose        libs/tracker/E_tracker.py:73-75,251-253,292-295
 *   cv2.findHomography     libs/tracker/E_tracker.py:188-194,199-205
 *   cv2.triangulatePoints  libs/geometry/ops_3d.py:63
 *   cv2.solvePnPRansac     libs/tracker/pnp_tracker.py:98-105
 *   cv2.Rodrigues          libs/tracker/pnp_tracker.py:116
 * All matrices row-major double unless stated.  max_iters: 1000 (E) / 2000 (H) in OpenCV 3.4.3
 * (not a Python parameter for findEssentialMat there; exposed here for BASELINE config 5).
 */
#ifndef CV3_CALIB3D_H
#define CV3_CALIB3D_H
#include "cv3_core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* RANSACUpdateNumIters (ptsetreg.cpp) */
int cv3_ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters);

/* five-point.cpp EMEstimatorCallback::runKernel on 5 normalised correspondences (q1[5][2], q2[5][2]);
 * writes up to 10 essential matrices (9 doubles each), returns their number */
int cv3_five_point(const double* q1, const double* q2, double* E_out);
/* cv::findEssentialMat(points1, points2, focal, pp, RANSAC, prob, threshold, mask).
 * pts: [n][2].  Returns 1 if a model was found (E[9], mask[n] written) else 0 (mask untouched). */
int cv3_find_essential_mat(const double* pts1, const double* pts2, int n, double focal, double ppx, double ppy,
                           double prob, double threshold, int max_iters, double* E, unsigned char* mask);
/* instrumentation for parity tests: also returns the number of RANSAC iterations run and the
 * (iteration, model index) of the winning hypothesis */
int cv3_find_essential_mat_ex(const double* pts1, const double* pts2, int n, double focal, double ppx, double ppy,
                              double prob, double threshold, int max_iters, double* E, unsigned char* mask,
                              int* iters_run, int* best_iter, int* best_model);

void cv3_decompose_essential_mat(const double* E, double* R1, double* R2, double* t);
/* cv::triangulatePoints: P1,P2 3x4; x1,x2 [2][n] (row 0 = x, row 1 = y); X4 [4][n] */
void cv3_triangulate_points(const double* P1, const double* P2, const double* x1, const double* x2, int n,
                            double* X4);
/* cv::recoverPose(E, points1, points2, focal, pp) (distance threshold 50); returns the cheirality count,
 * R[9], t[3]; mask (may be NULL) receives the per-point cheirality flags (0/255) */
int cv3_recover_pose(const double* E, const double* pts1, const double* pts2, int n, double focal, double ppx,
                     double ppy, double* R, double* t, unsigned char* mask);

/* cv::findHomography(points1, points2, RANSAC, thr, mask, maxIters, confidence); H[9]; returns 1/0 */
int cv3_find_homography_lsq(const double* pts1, const double* pts2, int n, double* H);
int cv3_find_homography(const double* pts1, const double* pts2, int n, double ransac_thr, int max_iters,
                        double confidence, double* H, unsigned char* mask);

/* cv::Rodrigues: rotation vector -> matrix, and matrix -> vector */
void cv3_rodrigues_v2m(const double* r, double* R);
void cv3_rodrigues_m2v(const double* R, double* r);
/* cv::solvePnPRansac(objectPoints [n][3], imagePoints [n][2], K [9], no distortion, useExtrinsicGuess=false,
 * iterationsCount, reprojectionError, confidence 0.99, flags=SOLVEPNP_ITERATIVE).
 * Returns 1/0; rvec[3], tvec[3]; inliers[] receives indices, *n_inliers their count. */
int cv3_solve_pnp_ransac(const double* obj, const double* img, int n, const double* K, int iterations,
                         double reproj_err, double confidence, double* rvec, double* tvec, int* inliers,
                         int* n_inliers);
/* pieces of the above, exported for unit tests: epnp::compute_pose on double points (us in pixels),
 * cvFindExtrinsicCameraParams2 (DLT initialisation + CvLevMarq; -2 = planar case, not implemented),
 * and the deterministic sin / cos / acos / lambda table the restatement uses in place of libm */
void cv3_solve_pnp_epnp_f32(const double* K, const float* obj, const float* img, int n, double* rvec, double* tvec);
void cv3_epnp(const double* K, const double* pws, const double* us, int n, double* R_out, double* t_out);
int cv3_find_extrinsic(const double* M, const double* m, int n, const double* K, double* rvec, double* tvec, int* stats);
int cv3_find_extrinsic_guess(const double* M, const double* m, int n, const double* K, const double* planar_guess,
                             double* rvec, double* tvec, int* stats);
double cv3_det_sin(double x);
double cv3_det_cos(double x);
double cv3_det_acos(double x);
double cv3_lm_lambda(int lambdaLg10);

#ifdef __cplusplus
}
#endif
#endif
