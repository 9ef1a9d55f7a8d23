// 8-bit LANCZOS resampling, bit-exact with Pillow's Image.resize(size, Image.LANCZOS): the host-side resize the
// reference applies to the depth-net input (/root/reference/libs/deep_models/deep_models.py:195-199).  Pillow is a
// third-party dependency of the reference; the arithmetic follows its src/libImaging/Resample.c:
//   precompute_coeffs        per output sample: window [xmin, xmin + xmax) of width 2 * 3 * max(scale, 1) around the
//                            sample centre, weights sinc(x) sinc(x / 3) normalised to sum 1 (double)
//   normalize_coeffs_8bpc    weights -> int, 22 fractional bits, round half away from zero
//   ImagingResampleHorizontal_8bpc / Vertical_8bpc
//                            int32 accumulation from 1 << 21, result clip8(acc >> 22); horizontal pass first, its
//                            uint8 result is the input of the vertical pass
// The coefficient tables are tiny (out_size x ~13 ints) and built once on the host; the two passes are integer
// kernels, one thread per output pixel (3 channels).  The oracle (oracle/pil_resample.py) is pinned to Pillow itself.
#include "resize_lanczos.h"

#include <algorithm>
#include <cmath>
#include <vector>

#include "dfvo_common.h"

namespace dfvo {

namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;

double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return std::sin(x) / x;
}
double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}

// bounds [out][2], coefficients [out][ksize]
void precompute_coeffs(int in_size, int out_size, std::vector<int>* bounds, std::vector<int>* kk, int* ksize_out) {
    const float in0 = 0.f, in1 = (float)in_size;
    const double scale = (double)(in1 - in0) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 3.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    bounds->assign((size_t)out_size * 2, 0);
    kk->assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        const double ss = 1.0 / filterscale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = lanczos_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            const double v = k[x] * (1 << PRECISION_BITS);
            (*kk)[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        (*bounds)[xx * 2] = xmin;
        (*bounds)[xx * 2 + 1] = xmax;
    }
    *ksize_out = ksize;
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// axis_stride / line_stride in pixels: the horizontal pass walks along a row (axis 1, line 1 row), the vertical pass
// along a column (axis = one row of `w_line` pixels)
template <bool LINES_ON_X>  // which index runs along threadIdx.x (the contiguous one in memory: columns)
__global__ void k_lanczos_pass(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ bounds,
                               const int* __restrict__ kk, int ksize, int n_out, int n_lines, int src_axis_stride,
                               int src_line_stride, int dst_axis_stride, int dst_line_stride) {
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y * blockDim.y + threadIdx.y;
    const int o = LINES_ON_X ? iy : ix;  // output index along the resampled axis
    const int line = LINES_ON_X ? ix : iy;
    if (o >= n_out || line >= n_lines) return;
    const int xmin = bounds[o * 2], xmax = bounds[o * 2 + 1];
    const int* k = kk + (size_t)o * ksize;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    const uint8_t* p = src + ((size_t)line * src_line_stride + (size_t)xmin * src_axis_stride) * 3;
    for (int x = 0; x < xmax; ++x) {
        const int c = k[x];
        s0 += (int)p[0] * c;
        s1 += (int)p[1] * c;
        s2 += (int)p[2] * c;
        p += (size_t)src_axis_stride * 3;
    }
    uint8_t* q = dst + ((size_t)line * dst_line_stride + (size_t)o * dst_axis_stride) * 3;
    q[0] = clip8(s0);
    q[1] = clip8(s1);
    q[2] = clip8(s2);
}

int upload(const std::vector<int>& h, int** d) {
    DFVO_HIP_CHECK(hipMalloc((void**)d, h.size() * sizeof(int)));
    DFVO_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    return DFVO_OK;
}
}  // namespace

// host-only view of the tables (no device involved): lets the CPU test suite compare them with Pillow's
int lanczos_coeffs_host(int in_size, int out_size, int* bounds, int* coeffs, int coeff_cap, int* ksize) {
    DFVO_ARG_CHECK(in_size > 0 && out_size > 0 && bounds && coeffs && ksize, "lanczos_coeffs: bad argument");
    std::vector<int> b, k;
    precompute_coeffs(in_size, out_size, &b, &k, ksize);
    DFVO_ARG_CHECK((size_t)coeff_cap >= k.size(), "lanczos_coeffs: coefficient buffer too small");
    std::copy(b.begin(), b.end(), bounds);
    std::copy(k.begin(), k.end(), coeffs);
    return DFVO_OK;
}

int LanczosResizer::init(int H_, int W_, int oh_, int ow_) {
    DFVO_ARG_CHECK(H_ > 0 && W_ > 0 && oh_ > 0 && ow_ > 0, "LanczosResizer: bad size");
    release();
    H = H_;
    W = W_;
    oh = oh_;
    ow = ow_;
    std::vector<int> b, k;
    precompute_coeffs(W, ow, &b, &k, &ksx);
    int rc = upload(b, &bx);
    if (rc == DFVO_OK) rc = upload(k, &kx);
    precompute_coeffs(H, oh, &b, &k, &ksy);
    if (rc == DFVO_OK) rc = upload(b, &by);
    if (rc == DFVO_OK) rc = upload(k, &ky);
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(hipMalloc((void**)&tmp, (size_t)H * ow * 3));
    return DFVO_OK;
}

void LanczosResizer::release() {
    for (int** p : {&bx, &kx, &by, &ky})
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
    if (tmp) (void)hipFree(tmp);
    tmp = nullptr;
}

int LanczosResizer::enqueue(const uint8_t* d_src, uint8_t* d_dst, hipStream_t s) const {
    DFVO_ARG_CHECK(tmp && d_src && d_dst, "LanczosResizer::enqueue before init / null buffer");
    const dim3 blk(64, 4);
    // Pillow skips a pass whose size does not change; a pass with equal sizes would not be the identity (it filters)
    const uint8_t* hsrc = d_src;
    if (ow != W) {
        uint8_t* hdst = oh != H ? tmp : d_dst;
        hipLaunchKernelGGL(k_lanczos_pass<false>, dim3(cdiv(ow, 64), cdiv(H, 4)), blk, 0, s, d_src, hdst, bx, kx, ksx, ow, H, 1, W, 1, ow);
        hsrc = hdst;
    }
    if (oh != H) {
        hipLaunchKernelGGL(k_lanczos_pass<true>, dim3(cdiv(ow, 64), cdiv(oh, 4)), blk, 0, s, hsrc, d_dst, by, ky, ksy, oh, ow, ow, 1, ow, 1);
    } else if (ow == W) {
        DFVO_HIP_CHECK(hipMemcpyAsync(d_dst, d_src, (size_t)H * W * 3, hipMemcpyDeviceToDevice, s));
    }
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// cv2.resize(img, (w, h)) of the loaded uint8 frame (/root/reference/libs/general/utils.py:51; default INTER_LINEAR).
// OpenCV 3.4.3 imgproc/src/resize.cpp (third party, restated): scale = 1 / (dst / src); coefficient per output sample
// f = float((d + 0.5) * scale - 0.5), s = floor(f), f -= s, 11-bit fixed point saturate_cast<short>((1 - f, f) * 2048)
// (half to even); columns clamp to (0, f = 0) / (W - 1, f = 0), rows clip to [0, H - 1] keeping their weights;
// horizontal S[s] a0 + S[s + 1] a1 in int, vertical (((b0 (r0 >> 4)) >> 16) + ((b1 (r1 >> 4)) >> 16) + 2) >> 2.
// Exact 2 x 2 decimation takes INTER_AREA's fast path (a + b + c + d + 2) >> 2.  One thread per output pixel: the
// coefficients cost a handful of instructions, so no tables are kept; HBM-bound (reads <= 4 source pixels per output,
// neighbouring threads share them through L1/L2).
__device__ __forceinline__ void linear_coef_u8(int d, double scale, int* s_out, int* c0, int* c1, float* f_out) {
    float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
    const int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    *s_out = s;
    *f_out = f;
    *c0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
    *c1 = (int)rintf(__fmul_rn(f, 2048.f));
}

// pitch: pixels per source row (a crop window of a wider frame reads with the frame's pitch); rev: output channel c is source
// channel C - 1 - c (cv2.cvtColor(img, COLOR_BGR2RGB) of read_image, utils.py:45, folded into the read)
__global__ __launch_bounds__(256) void k_resize_linear_u8(const uint8_t* __restrict__ src, int H, int W, int C, int pitch, int rev,
                                                           uint8_t* __restrict__ dst, int oh, int ow, double scale_x,
                                                           double scale_y, int area2) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y * blockDim.y + threadIdx.y;
    if (dx >= ow || dy >= oh) return;
    uint8_t* o = dst + ((size_t)dy * ow + dx) * C;
    if (area2) {
        const uint8_t* p = src + ((size_t)(2 * dy) * pitch + 2 * dx) * C;
        const size_t row = (size_t)pitch * C;
        for (int c = 0; c < C; ++c) {
            const int sc = rev ? C - 1 - c : c;
            o[c] = (uint8_t)((p[sc] + p[C + sc] + p[row + sc] + p[row + C + sc] + 2) >> 2);
        }
        return;
    }
    int sx, a0, a1, sy, b0, b1;
    float f;
    linear_coef_u8(dx, scale_x, &sx, &a0, &a1, &f);
    if (sx < 0) sx = 0, a0 = 2048, a1 = 0;
    if (sx >= W - 1) sx = W - 1, a0 = 2048, a1 = 0;
    linear_coef_u8(dy, scale_y, &sy, &b0, &b1, &f);
    const int y0 = sy < 0 ? 0 : (sy < H ? sy : H - 1), y1 = sy + 1 < 0 ? 0 : (sy + 1 < H ? sy + 1 : H - 1);
    const int sx1 = sx + 1 < W ? sx + 1 : sx;
    const uint8_t *r0 = src + (size_t)y0 * pitch * C, *r1 = src + (size_t)y1 * pitch * C;
    for (int c = 0; c < C; ++c) {
        const int sc = rev ? C - 1 - c : c;
        const int h0 = r0[sx * C + sc] * a0 + r0[sx1 * C + sc] * a1;
        const int h1 = r1[sx * C + sc] * a0 + r1[sx1 * C + sc] * a1;
        o[c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}

int enqueue_resize_linear_u8(const uint8_t* d_src, int H, int W, int C, uint8_t* d_dst, int oh, int ow, hipStream_t s, int pitch,
                             int rev) {
    if (pitch <= 0) pitch = W;
    DFVO_ARG_CHECK(d_src && d_dst && H > 0 && W > 0 && oh > 0 && ow > 0 && C >= 1 && C <= 4 && pitch >= W, "resize_linear_u8: bad argument");
    volatile double inv_x = (double)ow / (double)W, inv_y = (double)oh / (double)H;  // cv::resize: inv_scale = dsize / ssize
    const double scale_x = 1.0 / inv_x, scale_y = 1.0 / inv_y;                       // hal::resize: scale = 1 / inv_scale
    const int area2 = (std::fabs(scale_x - 2.0) < 2.220446049250313e-16 && std::fabs(scale_y - 2.0) < 2.220446049250313e-16) ? 1 : 0;
    hipLaunchKernelGGL(k_resize_linear_u8, dim3(cdiv(ow, 64), cdiv(oh, 4)), dim3(64, 4), 0, s, d_src, H, W, C, pitch, rev ? 1 : 0, d_dst,
                       oh, ow, scale_x, scale_y, area2);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

}  // namespace dfvo
