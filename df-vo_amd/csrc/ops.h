// Launchers for the streaming (HBM-bound) operators of the two nets.  NHWC float32 everywhere;
// (ptr, cs, co) = buffer, floats per pixel, channel offset.
#pragma once
#include "dfvo_common.h"

namespace dfvo {

// uint8 HWC RGB -> float NHWC4 (4th channel 0), value = float(u8/255.0), bilinear resize with
// align_corners=True to (th, tw).  deep_models.py:160-163 + lite_flow.py:70-76.
int launch_img_u8_to_flow_input(const uint8_t* img, int H, int W, float* dst, int th, int tw, hipStream_t s);
// uint8 HWC RGB (already at feed size) -> float NHWC4, value = (float(u8)/255 - 0.45)/0.225
// (ToTensor deep_models.py:199 + resnet_encoder.py:89).
int launch_img_u8_to_depth_input(const uint8_t* img, int H, int W, float* dst, hipStream_t s);
// generic NHWC bilinear resize (C multiple of 4, dense), torch F.interpolate semantics.
int launch_resize_bilinear(const float* src, int N, int H, int W, int C, float* dst, int Ho, int Wo,
                           int align_corners, hipStream_t s);
// Backward() warp, lite_flow_net.py:10-28: dst[n,y,x,:C] = bilinear(src[swap? N-1-n : n], (x,y)+flow*mult),
// zeros padding, align_corners=True grid.  If append_flow, also dst[..., C..C+1] = flow, C+2..C+3 = 0.
// lin_x/lin_y are torch.linspace(-1,1,W/H) tables (device).
int launch_warp(const float* src, int scs, int sco, int swap, const float* flow, int fcs, int fco, float mult,
                int N, int H, int W, int C, const float* lin_x, const float* lin_y, float* dst, int dcs, int dco,
                int append_flow, hipStream_t s, int step = 1);  // step > 1: only pixels (y % step == 0, x % step == 0)
// per-sample mean of a 2-channel flow over H*W -> mean[N][2]. lite_flow_net.py:255
size_t flow_mean_scratch_floats(int N);  // zero-filled, 8-byte aligned scratch of launch_flow_mean (partials + tickets)
int launch_flow_mean(const float* flow, int fcs, int fco, int N, int HW, float* scratch, float* mean,
                     hipStream_t s);
// Regularization input head, lite_flow_net.py:244-255: dst[n,y,x] = [sqrt(sum_c (I1 - warp(I2))^2 + 1e-6),
// fx - mean_x, fy - mean_y, 0].  I1 = img[n], I2 = img[N-1-n] (NHWC4, 3 used).
int launch_reg_prep(const float* img, const float* flow, int fcs, int fco, float mult, const float* mean, int N,
                    int H, int W, const float* lin_x, const float* lin_y, float* dst, hipStream_t s);
// depthwise ConvTranspose2d k4 s2 p1 no bias (lite_flow_net.py:109,117). w[C][4][4].
int launch_deconv_dw(const float* src, int scs, int sco, int N, int H, int W, int C, const float* w, float* dst,
                     int dcs, int dco, hipStream_t s);
// correlation volume (49 displacements) + leaky-ReLU 0.1; correlation.py:38-106,294 and
// lite_flow_net.py:145,148.  first = f1[n], second = f2[swap2 ? N-1-n : n]; output [N,Ho,Wo,52] (49 used).
int launch_correlation(const float* f1, int cs1, int co1, const float* f2, int cs2, int co2, int swap2, int N,
                       int H, int W, int C, int stride, float* dst, int dcs, float slope, hipStream_t s);
// Regularization output head (f-lconv), lite_flow_net.py:256-264.
int launch_reg_head(const float* dist, int dist_cs, int k, const float* flow, int fcs, int fco, const float* wx,
                    float bx, const float* wy, float by, int N, int H, int W, float* dst, int dcs, int dco,
                    hipStream_t s);
// MaxPool2d(3, stride 2, pad 1) NHWC (C multiple of 4, dense).
int launch_maxpool3x3s2(const float* src, int N, int H, int W, int C, float* dst, hipStream_t s);
// flow post-processing, lite_flow_net.py:322-324 + deep_flow.py:107-129 + deep_flow.py:171-196 +
// layers.py:213-229: net flow [2,h,w,(cs)] * scale, bilinear (align_corners) to (H,W), x*rw, y*rh;
// outputs fwd[2,H,W], bwd[2,H,W] planar and diff[H,W] = || fwd - grid_sample(-bwd, pix+fwd) ||.
int launch_flow_post(const float* netflow, int fcs, int fco, int h, int w, float scale, int H, int W, float* fwd,
                     float* bwd, float* diff, hipStream_t s);
// up to 12 device-to-device copies (float counts multiples of 4, 16-byte aligned) in ONE launch
struct CopySegs {
    const float* src[12];
    float* dst[12];
    unsigned n4[12];      // float4 elements per segment
    unsigned start[13];   // first workgroup of each segment (prefix sums), start[n] = grid size
    int n;
};
int launch_copy_segments(const float* const* src, float* const* dst, const size_t* floats, int n, hipStream_t s);
// monodepth2 tail: sigmoid disp -> depth = mult/(min_disp + (max_disp-min_disp)*disp); monodepth2.py:108-139
int launch_disp_to_depth(const float* disp, int dcs, int dco, int n, float min_disp, float disp_range,
                         float mult, float* depth, hipStream_t s);
// dfvo.py:314-319: nearest resize (cv2.INTER_NEAREST) to (H,W) -> raw f32 and preprocessed f64 map.
int launch_depth_post(const float* depth, int h, int w, int H, int W, int y0, int y1, int x0, int x1,
                      float min_depth, float max_depth, float* raw, double* proc, hipStream_t s);

}  // namespace dfvo
