/* libdfvo_hip.so -- C ABI of the MI355X (gfx950) DF-VO tracking hot path.
 *
 * The reference (Huangying-Zhan/DF-VO) has no FFI: its boundary is the Python class surface that
 * libs/dfvo.py consumes.  This header is the C ABI placed underneath that
 * surface; each entry point names the reference call site it replaces (paths relative to
 * /root/reference).  The Python mirror classes in df-vo_amd/libs bind these symbols with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative DFVO_ERR_* code otherwise;
 *     dfvo_last_error() returns a thread-local human readable message for the last failure.
 *   - "d_" arguments are DEVICE pointers (HBM), "h_" arguments are HOST pointers.  Nothing is
 *     retained after the call returns unless stated.
 *   - `stream` is a hipStream_t passed as void* (NULL = the object's own stream).
 *   - images are uint8 HWC RGB; activations are NHWC float32; dense maps are row-major.
 *   - handles are opaque, not re-entrant per handle, independent across handles.
 */
#ifndef DFVO_HIP_H
#define DFVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFVO_OK 0
#define DFVO_ERR_HIP (-1)
#define DFVO_ERR_ARG (-2)
#define DFVO_ERR_STATE (-3)
#define DFVO_ERR_RANGE (-4) /* f16x3 / f16 packing: an activation left f16's range in this call (see dfvo_f16s_overflow_count) */

const char* dfvo_last_error(void);
/* number of visible HIP devices (0 when there is none); never fails */
int dfvo_device_count(void);
int dfvo_set_device(int ordinal);
int dfvo_sync_device(void);

/* ---- raw device memory helpers (plumbing for hosts without torch) ---- */
int dfvo_malloc(void** d_ptr, size_t bytes);
int dfvo_free(void* d_ptr);
int dfvo_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int dfvo_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);

/* =====================================================================================
 * Unit-testable operators (each is one kernel launch; used by the nets below)
 * ===================================================================================== */

/* torch.nn.Conv2d + bias + activation (+ residual, + reflection pad, + x2 nearest upsample of
 * source 0, + channel concat of two sources) as an fp32 MFMA implicit GEMM.
 *   lite_flow_net.py:39-75,121-129,171-179,211-240; depth_decoder.py:50-65; layers.py:106-136
 * Weights are given in torch OIHW layout on the HOST (cin = c0 + c1) and packed internally. */
typedef struct dfvo_conv_desc {
    int N, H, W;              /* input batch / spatial size the taps index (after upsample) */
    int kh, kw, stride, pad_h, pad_w;
    int pad_mode;             /* 0 zeros, 1 reflect */
    int c0, cs0, co0, up0;    /* source 0: logical channels, floats per pixel, channel offset, x2 nearest */
    int c1, cs1, co1;         /* source 1 (c1 = 0: absent) */
    int cout;
    int act;                  /* 0 none, 1 leaky(act_param), 2 relu, 3 elu(act_param), 4 sigmoid */
    float act_param;
    int res_cs, res_co;       /* residual view (used when d_res != NULL) */
    int dst_cs, dst_co;
} dfvo_conv_desc;
int dfvo_conv2d(const dfvo_conv_desc* desc, const float* d_src0, const float* d_src1, const float* h_weight_oihw,
                const float* h_bias, const float* d_res, float* d_dst, void* stream);

/* Per-launch timing of the conv kernel family with HIP events recorded on the launch stream (the
 * roofline leg of bench.py).  Between begin and end every conv launch is bracketed by two events;
 * end() returns, per tile configuration (0: 128x128, 1: 32x128, 2: 256x64, 3: 64x64, 4: 256x32,
 * 5: 64x32, 6: 256x16, 7: 64x16, 8: 64x128, 9: 128x64, 10: 128x32, 11: 128x16, 12-15: the LDS-window 3x3 kernel
 * with (8x16)x128, (8x16)x64, (8x16)x32, (4x16)x128 tiles, 16/17/18: the 7x7 / 5x5 / 3x3 one- and two-channel
 * heads (direct kernel; wider 5x5 / 7x7 layers use the window kernel under the same ids); arrays of 19), the
 * summed duration [ms], useful FLOPs and launch count.
 * Do not use while a hipGraph capture is active (disable graphs on the nets first). */
/* arithmetic of the convolution layers packed AFTER this call (nets are packed at *_finalize):
 *   "fp32"   exact fp32 MFMA (v_mfma_f32_16x16x4_f32), the library default
 *   "f16x3"  fp32-class: operands split into two f16 planes (22 mantissa bits), three exact products per term on
 *            v_mfma_f32_32x32x16_f16, fp32 accumulate (df-vo_amd/csrc/conv_win_f16s.h); what bench.py's headline and
 *            the drop-in classes (DeepModel.initialize_models) select
 *   "f16"    NOT fp32-class: operands rounded to f16 (the hi plane alone), one product per term, fp32 accumulate, fp32
 *            activations between layers -- BASELINE config 5's "fp16 flow" (hard part 4: reported
 *            separately, with its flow / keypoint / pose deltas); the one- and two-channel heads stay exact fp32
 * Also read once from the environment variable DFVO_CONV_PRECISION. */
int dfvo_set_conv_precision(const char* name);
/* the current setting ("fp32" | "f16x3" | "f16"): a caller that sets it for its own nets restores it afterwards */
const char* dfvo_get_conv_precision(void);
/* Which scikit-learn the scale-recovery RANSAC (sklearn.linear_model.RANSACRegressor, E_tracker.py:618-636) reproduces
 * where the versions differ: r2_score of a one-sample consensus set is nan from 0.22 on (every later trial with the same
 * inlier count then wins), 1.0 / 0.0 in 0.20.3 -- the version the reference pins (envs/requirement.yml:233) and the
 * default here.  version: "<major>.<minor>[.patch]" of the scikit-learn to reproduce (process-wide, takes effect with the
 * next scale recovery). */
int dfvo_set_sklearn_compat(const char* version);
/* f16x3 range report: the hi plane of the split is f16, so an activation with |x| > 65504 does not fit -- it becomes
 * +-inf and propagates as inf / NaN into the layer's output (never a silently clamped product).  Every splitting kernel
 * counts the threads that saw such an activation, the packer the weights (those are clamped, and counted);
 * *h_count = events since the last reset (0 = the f16x3 result is the 22-bit split of the true operands everywhere).
 * reset != 0 clears the counter. */
int dfvo_f16s_overflow_count(unsigned long long* h_count, int reset);
int dfvo_conv_profile_begin(void);
int dfvo_conv_profile_end(double* h_ms24, double* h_flops24, int* h_launches24);
/* the same, plus the ALGORITHMIC HBM bytes of each configuration's launches (fp32 input map + output map + weights, each
 * once: SURVEY.md section 8d's price of the streaming layers) -- the numerator of bench.py's roofline.hbm entry */
int dfvo_conv_profile_end_bytes(double* h_ms24, double* h_flops24, int* h_launches24, double* h_bytes24);

/* correlation.py:38-106,281-340 (_FunctionCorrelation.forward) followed by leaky_relu(slope)
 * (lite_flow_net.py:145,148).  NHWC inputs [N,H,W,C]; output [N,ceil(H/s),ceil(W/s),52], 49 used.
 * slope = 1 gives the bare correlation. */
int dfvo_correlation(const float* d_first, const float* d_second, int N, int H, int W, int C, int stride,
                     float slope, float* d_out, void* stream);

/* Backward() bilinear warp, lite_flow_net.py:10-28.  h_lin_x/h_lin_y: torch.linspace(-1,1,W/H)
 * tables (HOST; NULL = library formula). */
int dfvo_backward_warp(const float* d_src, const float* d_flow, float flow_mult, int N, int H, int W, int C,
                       const float* h_lin_x, const float* h_lin_y, float* d_dst, void* stream);

/* depthwise ConvTranspose2d(k=4, s=2, p=1, bias=False), lite_flow_net.py:109,117. h_weight [C,1,4,4] */
int dfvo_deconv_dw4x4s2(const float* d_src, int N, int H, int W, int C, int cs, const float* h_weight,
                        float* d_dst, void* stream);

/* F.interpolate(mode='bilinear') on dense NHWC, C % 4 == 0 */
/* uint8 [H,W,3] -> uint8 [out_h,out_w,3], bit-exact with Pillow's Image.resize((out_w,out_h), Image.LANCZOS): 22-bit
 * fixed-point coefficient tables, horizontal pass then vertical pass, each rounded to uint8 (replaces the host-side
 * PIL call at deep_models.py:195-199).  Builds the tables per call; the nets keep theirs resident. */
int dfvo_resize_lanczos_u8(const uint8_t* d_src, int H, int W, uint8_t* d_dst, int out_h, int out_w, void* stream);
/* the resize's fixed-point tables for one axis (host only, no device needed): h_bounds [out_size][2] = (first input
 * sample, count), h_coeffs [out_size][*ksize] 22-bit coefficients (coeff_cap ints available) -- Pillow's
 * precompute_coeffs + normalize_coeffs_8bpc */
int dfvo_lanczos_coeffs(int in_size, int out_size, int* h_bounds, int* h_coeffs, int coeff_cap, int* ksize);
/* uint8 [H,W,C] (C <= 4) -> uint8 [out_h,out_w,C] = cv2.resize(img, (out_w, out_h)) with the default INTER_LINEAR, the
 * resize of the loaded frame to the configured size in read_image (libs/general/utils.py:51): OpenCV 3.4.3's 11-bit
 * fixed-point bilinear arithmetic (exact 2 x 2 decimation: its (a + b + c + d + 2) >> 2 area path).  Asynchronous on
 * `stream`. */
int dfvo_resize_linear_u8(const uint8_t* d_src, int H, int W, int C, uint8_t* d_dst, int out_h, int out_w, void* stream);
/* Everything of read_image (libs/general/utils.py:44-51) behind the decoder, in one launch on the decoded frame as cv2.imread
 * left it: cv2.cvtColor(img, COLOR_BGR2RGB) (bgr != 0: channel order reversed while reading), the crop img[y0:y1, x0:x1]
 * (pass 0, img_h, 0, img_w for none; the caller evaluates int(img_h * crop[0][0]) ... as utils.py:47-49 does) and
 * cv2.resize(img, (out_w, out_h)) with dfvo_resize_linear_u8's arithmetic.  d_decoded uint8 [img_h, img_w, 3], d_dst uint8
 * [out_h, out_w, 3].  PNG / JPEG entropy decoding stays on the host (bit-serial). */
int dfvo_read_image_tail_u8(const uint8_t* d_decoded, int img_h, int img_w, int bgr, int y0, int y1, int x0, int x1, uint8_t* d_dst,
                            int out_h, int out_w, void* stream);
int dfvo_resize_bilinear(const float* d_src, int N, int H, int W, int C, float* d_dst, int Ho, int Wo,
                         int align_corners, void* stream);

/* =====================================================================================
 * LiteFlowNet forward/backward flow + consistency   (replaces LiteFlow.inference_flow,
 * lite_flow.py:89-148, behind DeepModel.forward_flow, deep_models.py:144-182)
 * ===================================================================================== */
typedef struct dfvo_flownet dfvo_flownet;
int dfvo_flownet_create(int img_h, int img_w, void* stream, dfvo_flownet** out);
void dfvo_flownet_destroy(dfvo_flownet* net);
/* state_dict entry by its key (lite_flow.py:45-46), float32 HOST data; also accepts the optional
 * "aux.linspace_x.<level>" / "aux.linspace_y.<level>" tables (level 2..6). */
int dfvo_flownet_set_param(dfvo_flownet* net, const char* name, const float* h_data, int ndim, const int* shape);
int dfvo_flownet_finalize(dfvo_flownet* net);
/* flow-net input size for an image size (host only, no device needed): DeepFlow.get_target_size (deep_flow.py:89-105) as
 * the reference really evaluates it -- its rebinding of h / w makes the result (floor, floor) multiples of 32 (KITTI
 * 376 x 1241 -> 352 x 1216) unless float64 rounding tips it to (ceil, ceil) (192 x 640 -> 224 x 672) */
int dfvo_flow_target_size(int img_h, int img_w, int* net_h, int* net_w);
int dfvo_flownet_net_size(const dfvo_flownet* net, int* net_h, int* net_w);
int dfvo_flownet_set_graph(dfvo_flownet* net, int enable);
/* device in, device out: ref/cur uint8 [H,W,3]; fwd,bwd float [2,H,W]; diff float [H,W] */
int dfvo_flownet_forward(dfvo_flownet* net, const uint8_t* d_ref, const uint8_t* d_cur, float* d_fwd, float* d_bwd,
                         float* d_diff);
/* host in, host out (synchronous) */
int dfvo_flownet_forward_host(dfvo_flownet* net, const uint8_t* h_ref, const uint8_t* h_cur, float* h_fwd,
                              float* h_bwd, float* h_diff);
/* conv + correlation FLOPs (2 x MAC, unpadded) enqueued by the last forward */
double dfvo_flownet_last_flops(const dfvo_flownet* net);
/* debugging / parity: copy the final flow of pyramid level 2..6 ([2,h,w,2] float, HOST) */
int dfvo_flownet_get_level_flow(dfvo_flownet* net, int level, float* h_out, int* h, int* w);
int dfvo_flownet_sync(dfvo_flownet* net);

/* =====================================================================================
 * monodepth2 depth  (replaces Monodepth2DepthNet.inference_depth, monodepth2.py:91-139,
 * behind DeepModel.forward_depth, deep_models.py:184-206)
 * ===================================================================================== */
typedef struct dfvo_depthnet dfvo_depthnet;
int dfvo_depthnet_create(int feed_h, int feed_w, float min_depth, float max_depth, float baseline_mult,
                         void* stream, dfvo_depthnet** out);
void dfvo_depthnet_destroy(dfvo_depthnet* net);
/* keys of encoder.pth ("encoder.conv1.weight", ...) and depth.pth ("decoder.0.conv.conv.weight", ...) */
int dfvo_depthnet_set_param(dfvo_depthnet* net, const char* name, const float* h_data, int ndim, const int* shape);
int dfvo_depthnet_finalize(dfvo_depthnet* net);
int dfvo_depthnet_set_graph(dfvo_depthnet* net, int enable);
/* img uint8 [feed_h, feed_w, 3] -> depth float [feed_h, feed_w] (metres, x baseline multiplier) */
int dfvo_depthnet_forward(dfvo_depthnet* net, const uint8_t* d_img, float* d_depth);
int dfvo_depthnet_forward_host(dfvo_depthnet* net, const uint8_t* h_img, float* h_depth);
/* DeepModel.forward_depth in one call (deep_models.py:184-206): host uint8 image [img_h,img_w,3] of any size ->
 * Pillow-exact LANCZOS resize to the feed size on the device -> net -> host depth [feed_h,feed_w] */
int dfvo_depthnet_forward_image_host(dfvo_depthnet* net, const uint8_t* h_img, int img_h, int img_w, float* h_depth);
double dfvo_depthnet_last_flops(const dfvo_depthnet* net);
int dfvo_depthnet_sync(dfvo_depthnet* net);
/* dfvo.py:314-319 + utils.py:89-114: nearest resize to (H,W), crop rows/cols, range mask.
 * d_depth float [h,w] -> d_raw float [H,W], d_proc double [H,W] */
int dfvo_depth_postprocess(const float* d_depth, int h, int w, int H, int W, int y0, int y1, int x0, int x1,
                           float min_depth, float max_depth, float* d_raw, double* d_proc, void* stream);

/* =====================================================================================
 * Pose solvers (wavefront-parallel RANSAC, f64).  Host arrays in / out, synchronous: these are the
 * calls the Python mirror classes make where the reference calls OpenCV.
 * ===================================================================================== */
typedef struct dfvo_tracker dfvo_tracker;
int dfvo_tracker_create(void* stream, dfvo_tracker** out);
void dfvo_tracker_destroy(dfvo_tracker* trk);
/* Device time [ms] of the stages of the LAST dfvo_compute_pose_2d2d / dfvo_find_scale_from_depth call on this tracker, under
 * the reference's Timer sub-keys (E_tracker.py:197-296,597-638; libs/general/timer.py), from HIP events between the
 * kernels of the chain.  h_ms8 = {find H, GRIC-H (incl. find H, nested as in the reference), find-Ess (shuffles + the
 * `repeat` five-point RANSACs, one batched launch sequence here), GRIC-E, find-Ess (full) (shuffles .. best-model
 * bookkeeping), recover pose, triangulation (+ depth ratios), scale ransac}; -1 for a stage the last call did not run. */
int dfvo_tracker_stage_ms(dfvo_tracker* trk, double* h_ms8);

/* cv2.findEssentialMat(points1, points2, focal, pp, RANSAC, prob, threshold)
 * (E_tracker.py:59-67,231-239).  h_pts [n][2] doubles.  h_E[9], h_mask[n] (0/1).
 * h_info[5] = {found, iterations replayed, winning iteration, winning model, inlier count}.
 * max_iters: 1000 reproduces OpenCV 3.4.3 (not a Python parameter there). */
int dfvo_find_essential_mat(dfvo_tracker* trk, const double* h_pts1, const double* h_pts2, int n, double focal,
                            double ppx, double ppy, double prob, double threshold, int max_iters, double* h_E,
                            uint8_t* h_mask, int* h_info);
/* cv2.findHomography(src, dst, RANSAC, ransacReprojThreshold, maxIters=2000, confidence)
 * (E_tracker.py:188-194,199-205) including the least-squares refit and the LM refinement. */
int dfvo_find_homography(dfvo_tracker* trk, const double* h_pts1, const double* h_pts2, int n, double thr,
                         int max_iters, double confidence, double* h_H, uint8_t* h_mask, int* h_info);
/* cv2.recoverPose(E, points1, points2, focal, pp) (E_tracker.py:73-75,251-253,292-295).
 * h_R[9], h_t[3], h_mask[n] (0/255), *h_good = cheirality count. */
int dfvo_recover_pose(dfvo_tracker* trk, const double* h_E, const double* h_pts1, const double* h_pts2, int n,
                      double focal, double ppx, double ppy, double* h_R, double* h_t, uint8_t* h_mask, int* h_good);
/* cv2.triangulatePoints(P1, P2, x1, x2) (ops_3d.py:63): P 3x4, x [2][n], X4 [4][n] */
int dfvo_triangulate_points(dfvo_tracker* trk, const double* h_P1, const double* h_P2, const double* h_x1,
                            const double* h_x2, int n, double* h_X4);

/* ---- the global numpy RandomState threaded through the device (np.random.shuffle at
 * E_tracker.py:225-228 / pnp_tracker.py:91-92; sklearn RANSACRegressor's random_state=None at
 * E_tracker.py:618-636).  State layout = np.random.get_state(): key[624] then pos, as uint32[625]. */
int dfvo_tracker_seed(dfvo_tracker* trk, uint32_t seed);                 /* np.random.seed(seed) */
int dfvo_tracker_set_rng_state(dfvo_tracker* trk, const uint32_t* h_state625);
int dfvo_tracker_get_rng_state(dfvo_tracker* trk, uint32_t* h_state625);

/* local_bestN keypoint selection (kp_selection.py:74-200 behind KeypointSampler.kp_selection,
 * keypoint_sampler.py:76-143), score_method "flow", depth consistency off.
 * h_flow float [2,H,W], h_diff float [H,W]; h_kp1/h_kp2 double [num_bestN][2] (x,y), *n_out rows valid;
 * *good_kp_found 0 when either "insufficient keypoint" rule fires.  The order of the keypoints is
 * numpy's scalar-introselect argpartition order, cell by cell. */
int dfvo_kp_local_bestn(dfvo_tracker* trk, const float* h_flow, const float* h_diff, int H, int W, int num_row,
                        int num_col, int num_bestN, float thre, double* h_kp1, double* h_kp2, int* n_out,
                        int* good_kp_found);
/* same with cfg.kp_selection.local_bestN.score_method (kp_selection.py:137-141,151-156): 'flow' thresholds and ranks the
 * consistency map itself (what dfvo_kp_local_bestn does), 'flow_ratio' the map divided by the flow magnitude per pixel
 * (float32, numpy's evaluation order); the "enough good pixels" rule stays on the map itself (kp_selection.py:121) */
#define DFVO_KP_SCORE_FLOW 0
#define DFVO_KP_SCORE_FLOW_RATIO 1
int dfvo_kp_local_bestn_ex(dfvo_tracker* trk, const float* h_flow, const float* h_diff, int H, int W, int num_row,
                           int num_col, int num_bestN, float thre, int score_method, double* h_kp1, double* h_kp2,
                           int* n_out, int* good_kp_found);
/* bestN_flow_kp (kp_selection.py:33-71, ablation_correspondences_best_n.yml): the num_bestN pixels of the whole image
 * with the least forward-backward inconsistency, in np.argpartition(flow_diff[flow_diff >= 0], num_bestN)[:num_bestN]
 * order (numpy's scalar introselect).  *n_out = num_bestN, or 0 when the image has no more than num_bestN candidates
 * (numpy raises there). */
int dfvo_kp_bestn(dfvo_tracker* trk, const float* h_flow, const float* h_diff, int H, int W, int num_bestN,
                  double* h_kp1, double* h_kp2, int* n_out);
/* sampled_kp (kp_selection.py:327-378, the "uniform" correspondences of ablation_correspondences_uniform.yml): for each
 * index k of h_idx[n] (KeypointSampler.generate_kp_samples' linspace over the cropped grid [y0:y1, x0:x1], row-major)
 * kp1 = (x, y) of that pixel and kp2 = kp1 + flow there; h_kp1 / h_kp2 [n,2] */
int dfvo_kp_sampled(dfvo_tracker* trk, const float* h_flow, int H, int W, int y0, int y1, int x0, int x1,
                    const int* h_idx, int n, double* h_kp1, double* h_kp2);

/* EssTracker.kp_selection_good_depth (E_tracker.py:645-705): RigidFlow layer of the reference
 * depth under the ref -> cur motion (geometry/{rigid_flow,reprojection,backprojection,transformation3d,projection}.py,
 * float32), its per-pixel distance to the optical flow, then opt_rigid_flow_kp (kp_selection.py:203-324): per grid
 * cell the pixels with rigid distance < rigid_flow_thre and forward-backward distance < optical_flow_thre; "best" =
 * argpartition by the score, "uniform" = every step-th candidate.  h_raw_depth float [H,W] (the unprocessed CNN depth of
 * the reference frame), h_flow float [2,H,W], h_flow_diff float [H,W].  Outputs [num_bestN,2] doubles each, *n_out rows
 * valid (same count for both sets); h_rigid_flow_diff (optional, float [H,W]) receives the distance map ("rigid_flow_mask");
 * h_rigid_diff_override (optional) is used instead of the computed map. */
typedef struct dfvo_rigid_kp_cfg {
    int num_row, num_col, num_bestN;       /* kp_selection.rigid_flow_kp.{num_row,num_col,num_bestN} */
    double rigid_flow_thre, optical_flow_thre;
    int score_method;                      /* 0 "opt_flow", 1 "rigid_flow" */
    double K[9], Kinv[9];                  /* intrinsics and np.linalg.inv of them (cast to float32 like .float()) */
    double T_ref_to_cur[16];               /* ref_data['rigid_flow_pose'].pose */
} dfvo_rigid_kp_cfg;
int dfvo_kp_rigid_flow(dfvo_tracker* trk, const float* h_flow, const float* h_flow_diff, const float* h_raw_depth, int H,
                       int W, const dfvo_rigid_kp_cfg* cfg, const float* h_rigid_diff_override, double* h_kp1_best,
                       double* h_kp2_best, double* h_kp1_uniform, double* h_kp2_uniform, int* n_out,
                       float* h_rigid_flow_diff);

/* EssTracker.compute_pose_2d2d (E_tracker.py:154-307).  validity_method DFVO_VALIDITY_GRIC (default configuration):
 * homography + GRIC-H, `repeat` x (shuffle, findEssentialMat, GRIC-E), best-of by inlier count, recoverPose,
 * cheirality > 10 %.  DFVO_VALIDITY_FLOW (ablation_model_sel_flow.yml): the pair is tracked only when the mean keypoint
 * displacement exceeds validity_thre (else identity and no RandomState draw); a repeat is valid when the cheirality
 * count of recoverPose(E_rep) exceeds 10 % of the keypoints, and it can only become the best model above 5 %
 * (out: h_gric = the mean displacement, rep_gric[] = the per-repeat cheirality counts).
 * DFVO_VALIDITY_HOMO_RATIO (E_tracker.py:186-194,243-250): findHomography with ransacReprojThreshold 0.2; a repeat is
 * valid while H_inliers.sum() / (H_inliers.sum() + inliers.sum()) < validity_thre (out: h_gric = the homography's
 * inlier count, rep_gric[] = the ratios; fewer than 5 keypoints, where the reference raises: identity).
 * Consumes the tracker's RandomState for the shuffles. */
#define DFVO_VALIDITY_GRIC 0
#define DFVO_VALIDITY_FLOW 1
#define DFVO_VALIDITY_HOMO_RATIO 2
typedef struct dfvo_pose2d2d_cfg {
    double fx, cx, cy;       /* focal = fx, principal point (E_tracker.py:231-239) */
    double reproj_thre;      /* e_tracker.ransac.reproj_thre */
    int repeat;              /* e_tracker.ransac.repeat (or 3) */
    int max_iters;           /* 1000 = OpenCV 3.4.3 */
    double KinvT[9];         /* np.linalg.inv(K.T) */
    double Kinv[9];          /* np.linalg.inv(K)   */
    int validity_method;     /* e_tracker.validity.method: DFVO_VALIDITY_GRIC / _FLOW / _HOMO_RATIO */
    double validity_thre;    /* e_tracker.validity.thre (flow, homo_ratio) */
} dfvo_pose2d2d_cfg;
typedef struct dfvo_pose2d2d_out {
    double R[9], t[3];       /* pose cur -> ref (identity / zero when rejected) */
    int n, best_inlier_cnt, num_valid, major_valid, cheirality, h_found;
    double h_gric;
    int rep_inliers[8], rep_valid[8];
    double rep_gric[8];
} dfvo_pose2d2d_out;
int dfvo_compute_pose_2d2d(dfvo_tracker* trk, const double* h_kp_ref, const double* h_kp_cur, int n,
                           const dfvo_pose2d2d_cfg* cfg, dfvo_pose2d2d_out* out, uint8_t* h_inliers);

/* EssTracker.find_scale_from_depth (E_tracker.py:571-643): triangulate (ops_3d.py:44-67), scatter to a
 * sparse depth map (ops_3d.py:15-41), depth ratios on valid pixels, sklearn RANSACRegressor on them.
 * h_T21 4x4; h_depth double [H,W].  *scale = -1 when fewer than 11 valid ratios.
 * method DFVO_SCALE_DEPTH_RATIO: .fit(depth_ratio, ones); DFVO_SCALE_ABS_DIFF: .fit(depth_tri, depth_pred)
 * (scale_recovery.ransac.method, E_tracker.py:626-635).
 * h_info[4] = {n_valid, n_trials, n_inliers, status}.  Consumes the tracker's RandomState. */
#define DFVO_SCALE_DEPTH_RATIO 0
#define DFVO_SCALE_ABS_DIFF 1
typedef struct dfvo_scale_cfg {
    double cx, cy, fx, fy;
    int min_samples, max_trials;
    double stop_prob, thre;
    int method;              /* DFVO_SCALE_DEPTH_RATIO / DFVO_SCALE_ABS_DIFF */
} dfvo_scale_cfg;
int dfvo_find_scale_from_depth(dfvo_tracker* trk, const double* h_kp1, const double* h_kp2, int n,
                               const double* h_T21, const double* h_depth, int H, int W, const dfvo_scale_cfg* cfg,
                               double* scale, int* h_info);
/* The same call without the H x W upload (3.7 MB of float64 at KITTI size, from pageable memory, per pair): the reference reads
 * depth2 only under the sparse triangulated map, i.e. at the pixels (int(kp2.x), int(kp2.y)) (E_tracker.py:604-612,
 * ops_3d.py:29-40).  h_depth_at_kp2 [n] = depth2[int(kp2[i].y), int(kp2[i].x)] (truncation toward zero; any value where that
 * pixel is outside the H x W map or the coordinate is not finite -- such keypoints are dropped before the value is read).
 * h_rng625 (optional, in / out): numpy RandomState words to run under, replaced by the advanced state -- saves the two
 * blocking dfvo_tracker_{set,get}_rng_state round trips of the mirror.  Same result as dfvo_find_scale_from_depth, bit for bit. */
int dfvo_find_scale_from_depth_at_kp(dfvo_tracker* trk, const double* h_kp1, const double* h_kp2, int n, const double* h_T21,
                                     const double* h_depth_at_kp2, int H, int W, const dfvo_scale_cfg* cfg,
                                     uint32_t* h_rng625, double* scale, int* h_info);
/* The regression stage on its own -- sklearn.linear_model.RANSACRegressor(LinearRegression(fit_intercept=False),
 * min_samples, max_trials, stop_probability, residual_threshold).fit(x[:, None], y) as E_tracker.py:618-636 builds it --
 * on raw host arrays: h_x [n], h_y [n] or NULL for y = 1 (the depth-ratio case).  Consumes the tracker's RandomState as
 * sklearn consumes np.random.  coef: estimator_.coef_[0, 0]; h_info as above (status -1: "RANSAC could not find a valid
 * consensus set", where sklearn raises ValueError).  cfg->cx .. fy and method are not read. */
int dfvo_ransac_regressor(dfvo_tracker* trk, const double* h_x, const double* h_y, int n, const dfvo_scale_cfg* cfg,
                          double* coef, int* h_info);

/* PnpTracker.compute_pose_3d2d (pnp_tracker.py:45-125): masks (kp2 inside the image, depth_1[int(kp1)] != 0 and
 * in (min_depth, max_depth)), unprojection_kp (ops_3d.py:70-94), `repeat` x [np.random.shuffle +
 * cv2.solvePnPRansac(iterationsCount = iters, reprojectionError = reproj_thre)], best by inlier count,
 * cv2.Rodrigues.  h_kp1 / h_kp2 [n][2] doubles, h_depth double [H,W] (depth of view 1).
 * out->R / tvec: the solvePnP pose (view-1 points -> view-2 camera); the caller inverts it like the reference
 * does (pose.pose = pose.inv_pose).  h_keep[n] (optional): 1 where the keypoint survived the masks.
 * Consumes the tracker's RandomState (one shuffle per repeat, drawn even when too few points remain).
 * Coplanar object points take cvFindExtrinsicCameraParams2's planar (homography) initialisation, as OpenCV does. */
typedef struct dfvo_pose3d2d_cfg {
    double fx, fy, cx, cy;
    double Kinv[9];                 /* Intrinsics.inv_mat, row-major */
    double min_depth, max_depth;    /* cfg.depth.{min,max}_depth */
    int repeat;                     /* cfg.pnp_tracker.ransac.repeat if is_iterative else 3 */
    int iters;                      /* cfg.pnp_tracker.ransac.iter */
    double reproj_thre;             /* cfg.pnp_tracker.ransac.reproj_thre */
} dfvo_pose3d2d_cfg;
typedef struct dfvo_pose3d2d_out {
    int found, best_inliers, n_filtered, status;
    double rvec[3], tvec[3], R[9];
} dfvo_pose3d2d_out;
int dfvo_compute_pose_3d2d(dfvo_tracker* trk, const double* h_kp1, const double* h_kp2, int n, const double* h_depth,
                           int H, int W, const dfvo_pose3d2d_cfg* cfg, dfvo_pose3d2d_out* out, uint8_t* h_keep);
/* The same call without the H x W upload: the reference reads depth_1 only at kp1's pixels (pnp_tracker.py:71-77:
 * depth_1[kp1[:, 1].astype(int), kp1[:, 0].astype(int)], truncation toward zero, negative indices wrapping as numpy's do).
 * h_depth_at_kp1 [n] = that value per keypoint (any value where the wrapped index is still outside the map: the keypoint is
 * dropped before the value is read).  h_rng625 (optional, in / out) as in dfvo_find_scale_from_depth_at_kp.  Same result as
 * dfvo_compute_pose_3d2d, bit for bit. */
int dfvo_compute_pose_3d2d_at_kp(dfvo_tracker* trk, const double* h_kp1, const double* h_kp2, int n, const double* h_depth_at_kp1,
                                 int H, int W, const dfvo_pose3d2d_cfg* cfg, uint32_t* h_rng625, dfvo_pose3d2d_out* out,
                                 uint8_t* h_keep);

/* =====================================================================================
 * Fused per-pair pipeline: images in HBM -> relative pose, everything between on the device
 * (libs/dfvo.py:299-345 deep_model_inference + :121-262 tracking, hybrid E-tracker path).
 * DFVO_PIPELINE_SLOTS slots buffer the net outputs so that the nets of the following pairs run under the solvers of pair k.
 * ===================================================================================== */
typedef struct dfvo_pipeline dfvo_pipeline;
typedef struct dfvo_pipeline_cfg {
    int img_h, img_w, feed_h, feed_w;
    float net_min_depth, net_max_depth, baseline_mult;   /* monodepth2.py:74-89 */
    double min_depth, max_depth;                         /* cfg.depth.{min,max}_depth (utils.py:89-114) */
    double depth_crop[4];                                /* y0 y1 x0 x1 fractions (cfg.crop.depth_crop) */
    double fx, fy, cx, cy;
    double KinvT[9], Kinv[9];
    int kp_num_row, kp_num_col, kp_num_bestN;
    double kp_thre;
    double e_reproj_thre;
    int e_repeat, e_max_iters;
    int scale_min_samples, scale_max_trials;
    double scale_stop_prob, scale_thre;
    uint32_t seed;                                       /* np.random.seed(cfg.seed), run.py:81-84 */
    int pnp_repeat, pnp_iters;                           /* cfg.pnp_tracker.ransac.{repeat,iter} */
    double pnp_reproj_thre;                              /* cfg.pnp_tracker.ransac.reproj_thre */
} dfvo_pipeline_cfg;
#define DFVO_TRACK_E 0                 /* pose from the essential matrix + depth scale */
#define DFVO_TRACK_CONSTANT_MOTION 1   /* not enough keypoints: caller reuses the previous motion (dfvo.py:157-161) */
#define DFVO_TRACK_NEEDS_PNP 2         /* E rejected or scale == -1 and no reference depth is known yet (first pair) */
#define DFVO_TRACK_PNP 3               /* E rejected or scale == -1: pose from the PnP fallback (dfvo.py:225-250) */
typedef struct dfvo_track_out {
    double R[9], t[3];        /* DFVO_TRACK_E: E-tracker pose cur -> ref (t has unit norm or is zero);
                                 DFVO_TRACK_PNP: the solvePnP pose (ref points -> cur camera; identity when no repeat
                                 found a model), which the caller inverts as pnp_tracker.py:118 does */
    double scale;
    int status, n_kp, good_kp_found, best_inlier_cnt, num_valid, cheirality;
    int scale_n_valid, scale_n_trials, scale_n_inliers;
    int pnp_found, pnp_inliers, pnp_n_filtered;
} dfvo_track_out;
/* Stream placement (dfvo_pipeline_create and dfvo_session_create; df-vo_amd/csrc/stream_pool.hip).  Both objects drive several
 * HIP streams whose hardware queues should sit on different dispatch pipes of the command processor (two busy streams on one
 * pipe slow each other's launches 2.5 x).  Which pipe a stream lands on follows the PROCESS's stream creation order, so the
 * objects create twelve candidates and classify them by a ~10 ms timing probe.  The probe is a measurement:
 *   - a classification is accepted when it has the shape of a process whose queues are all its own (four groups of three)
 *     or when two passes agree stream for stream; a positive ("these two share a pipe") has to show twice in a row;
 *   - when no pass is accepted, or a measurement fails, ONE line goes to stderr and every role keeps a stream in the
 *     runtime's creation order (slower by up to a third, never wrong);
 *   - environment: DFVO_STREAM_POOL=creation skips the probe; DFVO_STREAM_PROBE_VERBOSE=1 prints every pass;
 *     DFVO_STREAM_POOL_FORCE_FAIL=1 (test hook) makes every measurement fail. */
int dfvo_pipeline_create(const dfvo_pipeline_cfg* cfg, dfvo_pipeline** out);
void dfvo_pipeline_destroy(dfvo_pipeline* p);
int dfvo_pipeline_set_flow_param(dfvo_pipeline* p, const char* name, const float* h_data, int ndim, const int* shape);
int dfvo_pipeline_set_depth_param(dfvo_pipeline* p, const char* name, const float* h_data, int ndim, const int* shape);
int dfvo_pipeline_finalize(dfvo_pipeline* p);
int dfvo_pipeline_seed(dfvo_pipeline* p, uint32_t seed);
int dfvo_pipeline_set_graph(dfvo_pipeline* p, int enable);   /* hipGraph replay of the nets (default on) */
/* number of output slots: the host may run the nets this many pairs ahead of the solver stage (the nets of pairs
 * k+1 .. k+3 queue up behind each other on their streams while dfvo_pipeline_track(k) blocks the host) */
#define DFVO_PIPELINE_SLOTS 4
/* Ordering contract for every device frame handed to the three calls below: they are ASYNCHRONOUS and read the frame on
 * the pipeline's own non-blocking HIP streams.  (1) The frame must be complete when the call is made -- work still queued
 * on another stream (e.g. a resize on the caller's stream) is not waited for.  (2) The frame must stay unchanged, and
 * allocated, until dfvo_pipeline_track() of that slot has returned (enqueue_nets) / until dfvo_pipeline_sync() or the
 * first dfvo_pipeline_track() after the call has returned (set_ref_depth, set_ref_image). */
/* enqueue both nets for one pair into `slot` (0 .. DFVO_PIPELINE_SLOTS-1); returns at once.  d_* uint8 device images:
 * ref/cur [img_h,img_w,3], cur_feed [feed_h,feed_w,3] (the PIL-LANCZOS resized current frame) or NULL: the current frame
 * is then resized on the device (dfvo_resize_lanczos_u8's arithmetic) ahead of the depth net.
 * d_ref == NULL (sequences): the reference frame is the current frame of the previous dfvo_pipeline_enqueue_nets call.
 * LiteFlowNet's image pyramid and Features (lite_flow_net.py:78-86, 307-309) are functions of one frame each; the
 * reference's model call runs them on both frames of every pair, i.e. twice per frame of a sequence.  Here the pyramids
 * the previous pass computed for that frame are carried over (device copies, ordered by an event between the flow-net
 * instances) and only the new frame runs through Features: same values, 15 GFLOP less per pair at KITTI size. */
int dfvo_pipeline_enqueue_nets(dfvo_pipeline* p, int slot, const uint8_t* d_ref, const uint8_t* d_cur,
                               const uint8_t* d_cur_feed);
/* depth of the very first reference frame (dfvo.py computes the depth of every frame as it becomes `cur`; the first
 * frame never is): exactly one of d_feed (uint8 [feed_h,feed_w,3], runs the depth net) or d_depth_override (processed
 * depth, double [H,W]).  Later reference depths roll over from the current frame of each tracked pair. */
int dfvo_pipeline_set_ref_depth(dfvo_pipeline* p, const uint8_t* d_feed, const double* d_depth_override);
/* same from the full-size first frame [img_h,img_w,3]: device LANCZOS resize, then the depth net */
int dfvo_pipeline_set_ref_image(dfvo_pipeline* p, const uint8_t* d_img);
/* Optional: enqueue the RNG-independent half of the solver stage of `slot` (keypoint selection, homography RANSAC +
 * refinement, GRIC-H) right after dfvo_pipeline_enqueue_nets(slot, ...).  It waits for the slot's flow outputs on the
 * device and runs on its own stream, i.e. under the nets / solver stage of earlier pairs; dfvo_pipeline_track(slot)
 * then only adds the numpy-RandomState consumers (shuffles, five-point RANSAC, scale, PnP), in pair order.  The
 * overrides, when given, must be the ones later passed to dfvo_pipeline_track.  Returns at once. */
int dfvo_pipeline_prefetch_track(dfvo_pipeline* p, int slot, const float* d_flow_override, const float* d_diff_override);
/* keypoint selection + E-tracker + scale recovery (+ the PnP fallback when the E-tracker result is rejected) on the
 * outputs in `slot` (waits for its nets).  Optional device overrides replace the forward flow [2,H,W] / consistency
 * map [H,W] / processed depth [H,W] double that feed the solver stage (used by bench.py). Synchronous. */
int dfvo_pipeline_track(dfvo_pipeline* p, int slot, const float* d_flow_override, const float* d_diff_override,
                        const double* d_depth_override, dfvo_track_out* out);
/* The same in two halves, so that the host can enqueue the nets of the pairs ahead while the chain of this pair runs:
 * _begin waits for the slot's keypoint stage and enqueues the RandomState-ordered chain + the copy of its results into
 * pinned host memory; _end waits for them, runs the PnP fallback where the reference takes it and rolls the reference
 * depth over.  Pairs must be begun and ended in order, and pair k + 1 must not be begun before pair k was ended (the PnP
 * fallback of pair k consumes the RandomState ahead of pair k + 1's shuffles).  dfvo_pipeline_track = _begin + _end.
 * Lifetime of the overrides with the split calls: _end enqueues the roll-over copy of the reference depth (from
 * d_depth_override when one was given to _begin) and returns WITHOUT waiting for it; a caller-owned d_depth_override must
 * therefore stay unchanged and allocated until the next dfvo_pipeline_track_end / dfvo_pipeline_sync has returned (the
 * flow / consistency overrides: until _end of their own pair has returned).  The pipeline's own buffers are ordered on
 * the device (the depth stream's next write of the slot waits for the copy). */
int dfvo_pipeline_track_begin(dfvo_pipeline* p, int slot, const float* d_flow_override, const float* d_diff_override,
                              const double* d_depth_override);
int dfvo_pipeline_track_end(dfvo_pipeline* p, int slot, dfvo_track_out* out);
int dfvo_pipeline_get_flow(dfvo_pipeline* p, int slot, float* h_fwd, float* h_bwd, float* h_diff, float* h_raw_depth,
                           double* h_depth);
/* inspection (tests, debugging): keypoints selected for `slot` (kp_best of ref / cur, double [n][2], x,y), the E-tracker's
 * best inlier mask (uint8 [n]; ref_data['inliers'], dfvo.py:179) -- up to `cap` entries, *n_out = the full count -- and the
 * numpy RandomState the pipeline carries from pair to pair (uint32 key[624] + pos, np.random.get_state() layout) */
int dfvo_pipeline_get_keypoints(dfvo_pipeline* p, int slot, int cap, double* h_kp_ref, double* h_kp_cur,
                                uint8_t* h_inliers, int* n_out);
int dfvo_pipeline_get_rng_state(dfvo_pipeline* p, uint32_t* h_state625);
int dfvo_pipeline_set_rng_state(dfvo_pipeline* p, const uint32_t* h_state625);
int dfvo_pipeline_sync(dfvo_pipeline* p);
double dfvo_pipeline_net_flops(const dfvo_pipeline* p);

/* ---- frame session: the reference's own synchronous call order, fast (df-vo_amd/csrc/session.hip) ----
 * /root/reference/libs/dfvo.py calls, per frame k: DeepModel.forward_depth([img_k]) (dfvo.py:312), forward_flow(cur, ref)
 * (:324), KeypointSampler.kp_selection (:147), EssTracker.compute_pose_2d2d (:168), scale_recovery (:198) -- blocking
 * calls, host numpy arrays in and out.  Everything they need is known at the first of them; dfvo_session_push_frame
 * (called by the DeepModel mirror from forward_depth) uploads the frame ONCE and enqueues, each on its own stream: the
 * depth net of frame k, the flow net of (k - 1, k) (frame k - 1's pyramids carried over), both outputs into pinned host
 * buffers, and -- when the KeypointSampler / EssTracker mirrors registered their configurations (kp, pose non-NULL) --
 * local_bestN and the RandomState-independent half of compute_pose_2d2d (findHomography + refinement + GRIC-H) behind the
 * flow net.  The later calls wait for an event and hand out the result; each validates that it is asked for exactly what
 * was enqueued (generation, configuration, keypoint arrays byte for byte, RandomState word for word) and otherwise runs the
 * plain entry point.
 * Results are identical to the plain entry points' (tests/test_dropin_gpu.py).  Host pointers returned here are pinned
 * buffers owned by the session, valid until two further frames have been pushed.  Not re-entrant. */
typedef struct dfvo_session dfvo_session;
typedef struct dfvo_session_kp_cfg {
    int num_row, num_col, num_bestN;  /* kp_selection.local_bestN.{num_row, num_col, num_bestN} */
    float thre;                       /* .thre */
    int score_method;                 /* DFVO_KP_SCORE_FLOW / DFVO_KP_SCORE_FLOW_RATIO */
} dfvo_session_kp_cfg;
int dfvo_session_create(dfvo_flownet* flow, dfvo_depthnet* depth, dfvo_tracker* trk, int img_h, int img_w, dfvo_session** out);
void dfvo_session_destroy(dfvo_session* s);
int dfvo_session_reset(dfvo_session* s);             /* forget the held frame (a new sequence) */
int dfvo_session_invalidate_carry(dfvo_session* s);  /* someone else ran the flow net: recompute both pyramids next time */
int dfvo_session_quiesce(dfvo_session* s);           /* a plain solver call is about to use the tracker: wait for the speculative stage */
/* h_img uint8 [img_h, img_w, 3] (copied into the session's pinned ring before the call returns); *generation = index of this
 * frame since create / reset.  flags: DFVO_PUSH_NO_FLOW = depth net only -- no flow pass of (generation - 1, generation), no
 * speculative stage (a caller that never asked for the flow of the last pushes; the next pair runs both frames through
 * Features). */
#define DFVO_PUSH_NO_FLOW 1
int dfvo_session_push_frame(dfvo_session* s, const uint8_t* h_img, const dfvo_session_kp_cfg* kp, const dfvo_pose2d2d_cfg* pose,
                            int flags, long long* generation);
/* float [feed_h, feed_w].  DFVO_ERR_RANGE (pointer still set) when the depth net of this frame drove an activation out of
 * f16's range under an f16x3 / f16 packing: the map holds inf / NaN. */
int dfvo_session_depth(dfvo_session* s, long long generation, const float** h_depth);
/* flow of (generation - 1, generation): fwd / bwd float [2, img_h, img_w], diff float [img_h, img_w]; DFVO_ERR_RANGE as above */
int dfvo_session_flow(dfvo_session* s, long long generation, const float** h_fwd, const float** h_bwd, const float** h_diff);
/* the session's pinned copy of frame `generation` (the newest or the one before), uint8 [img_h, img_w, 3]: what was uploaded --
 * the mirror compares the frames handed to forward_flow with it byte for byte */
int dfvo_session_frame(dfvo_session* s, long long generation, const uint8_t** h_frame);
/* Buffer lifetime: the pointers dfvo_session_depth / _flow return belong to ring slot generation % 3 and are overwritten by
 * the push of generation + 3.  A caller that still references them then (or at dfvo_session_destroy) takes them over:
 * h_old4 = {depth, fwd, bwd, diff} of that slot, the session allocates itself fresh ones; release each with dfvo_host_free. */
int dfvo_session_detach_slot(dfvo_session* s, long long generation, void** h_old4);
int dfvo_host_free(void* h_pinned);
int dfvo_session_keypoints(dfvo_session* s, long long generation, const dfvo_session_kp_cfg* kp, const double** h_kp_ref,
                           const double** h_kp_cur, int* n, int* good_kp_found);
/* The RandomState-consuming half of compute_pose_2d2d (shuffles, five-point RANSACs, GRIC-E, recoverPose) enqueued AHEAD of
 * the call, as soon as the keypoints of `generation` exist, under the RandomState the caller has now (h_rng625: numpy's 624
 * key words + position).  The DeepModel mirror calls it from forward_flow.  *enqueued = 1 when it was. */
int dfvo_session_pose_ahead(dfvo_session* s, long long generation, const uint32_t* h_rng625, const dfvo_pose2d2d_cfg* cfg,
                            int* enqueued);
/* dfvo_compute_pose_2d2d through the session, under the RandomState h_rng625 (np.random's state at the call; the tracker's
 * device copy is set from it unless the result enqueued ahead is handed out).  *used_resident = 2: the call had been enqueued
 * ahead under exactly this RandomState, these keypoints and this configuration; 1: only the RandomState-consuming half had
 * to run; 0: the plain entry point ran.  Read the advanced state back with dfvo_tracker_get_rng_state. */
int dfvo_session_pose_2d2d(dfvo_session* s, const double* h_kp_ref, const double* h_kp_cur, int n, const dfvo_pose2d2d_cfg* cfg,
                           dfvo_pose2d2d_out* out, uint8_t* h_inliers, const uint32_t* h_rng625, int* used_resident);

/* DFVO.update_global_pose (dfvo.py:109-119: t_w += R_w t, then R_w = R_w R) over a whole gathered sequence in ONE launch,
 * constant-motion rows included (dfvo.py:157-161: a row with status 1 reuses the previous pair's relative motion).
 * rows [n][17] = relative pose cur -> ref (4x4 row major) | status (the layout of dist.allgather_poses); first [16] = pose
 * of frame 0 (NULL: identity); poses [n+1][16] out.  *h_bad_row = index of the first status-2 row (a pair that needed the
 * PnP fallback but had no reference depth), -1 when there is none.  The _device variant takes device pointers (the
 * gathered rows are already in HBM after the RCCL all-gather) and a HIP stream. */
int dfvo_compose_trajectory(const double* h_rows, int n, const double* h_first, double* h_poses, int* h_bad_row);
int dfvo_compose_trajectory_device(const double* d_rows, int n, const double* d_first, double* d_poses, int* h_bad_row,
                                   void* stream);

/* ---- the data-parallel exchange step (SURVEY.md section 8e / 8b "dfvo_allgather_poses") ----
 * Every rank tracks its chunks of frame pairs with no activation exchange; ONE RCCL ncclAllGather per job then hands every
 * rank all relative poses (rows of DFVO_POSE_ROW = 17 doubles: 4x4 row-major cur -> ref pose | status), after which each
 * rank (or rank 0) composes the trajectories with dfvo_compose_trajectory.  Replaces the sequential hand-over of
 * update_global_pose between frames (dfvo.py:109-119,157-161) for chunks tracked on different GPUs.
 * dfvo_comm wraps one RCCL communicator (bound with dlopen("librccl.so.1") at first use; its absence is an error, there is
 * no fallback).  Bootstrap as with NCCL: rank 0 calls dfvo_comm_unique_id and ships the DFVO_COMM_ID_BYTES bytes to every
 * rank by any out-of-band means (file, socket, MPI, torch's store); every rank calls dfvo_comm_create (collective; select
 * the GPU with dfvo_set_device first).
 * dfvo_allgather_poses: h_rows [n_local][17] of this rank, counts[world] = rows of every rank (deterministic: the chunking
 * is a function of the job, so no count exchange precedes the collective), h_out [sum counts][17] in rank order.  Ranks
 * pad to max(counts) rows for the fixed-size collective; the padding is trimmed here.  Synchronous.
 * _device: the bare collective on caller-owned device buffers (d_send [rows_per_rank][17], d_recv [world][rows_per_rank]
 * [17]) on `stream` (NULL: the communicator's own), asynchronous. */
#define DFVO_COMM_ID_BYTES 128
#define DFVO_POSE_ROW 17
typedef struct dfvo_comm dfvo_comm;
int dfvo_comm_unique_id(uint8_t* h_id128);
int dfvo_comm_create(const uint8_t* h_id128, int world, int rank, dfvo_comm** out);
int dfvo_comm_destroy(dfvo_comm* c);
int dfvo_allgather_poses(dfvo_comm* c, const double* h_rows, int n_local, const int* counts, double* h_out);
int dfvo_allgather_poses_device(dfvo_comm* c, const double* d_send, int rows_per_rank, double* d_recv, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DFVO_HIP_H */
