// extern "C" surface declared in include/dfvo_hip.h (part 1: device helpers, operators, nets).
#include "../../include/dfvo_hip.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "nets.h"
#include "ops.h"
#include "resize_lanczos.h"
#include "capi_types.h"

namespace dfvo {
static thread_local std::string g_err;
void set_last_error(const std::string& s) { g_err = s; }
const char* last_error() { return g_err.c_str(); }

int ensure_dyn_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> configured;
    int dev = 0;
    DFVO_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = configured[{dev, kernel}];
    if (bytes > have) {
        DFVO_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return DFVO_OK;
}
}  // namespace dfvo

using namespace dfvo;


#define API_TRY(expr)                   \
    do {                                \
        int _rc = (expr);               \
        if (_rc != DFVO_OK) return _rc; \
    } while (0)

extern "C" {

const char* dfvo_last_error(void) { return dfvo::last_error(); }

int dfvo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int dfvo_set_device(int ordinal) {
    DFVO_HIP_CHECK(hipSetDevice(ordinal));
    return DFVO_OK;
}
int dfvo_sync_device(void) {
    DFVO_HIP_CHECK(hipDeviceSynchronize());
    return DFVO_OK;
}
int dfvo_malloc(void** d_ptr, size_t bytes) {
    DFVO_HIP_CHECK(hipMalloc(d_ptr, bytes));
    return DFVO_OK;
}
int dfvo_free(void* d_ptr) {
    DFVO_HIP_CHECK(hipFree(d_ptr));
    return DFVO_OK;
}
int dfvo_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
    DFVO_HIP_CHECK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return DFVO_OK;
}
int dfvo_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
    DFVO_HIP_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return DFVO_OK;
}

// ------------------------------------------------------------------------------------------------
int dfvo_conv2d(const dfvo_conv_desc* d, const float* d_src0, const float* d_src1, const float* h_w,
                const float* h_bias, const float* d_res, float* d_dst, void* stream) {
    DFVO_ARG_CHECK(d && d_src0 && h_w && d_dst, "dfvo_conv2d: null argument");
    DFVO_ARG_CHECK(d->c1 == 0 || d_src1, "dfvo_conv2d: c1 > 0 needs d_src1");
    hipStream_t s = (hipStream_t)stream;
    const int Ho = (d->H + 2 * d->pad_h - d->kh) / d->stride + 1;
    const int Wo = (d->W + 2 * d->pad_w - d->kw) / d->stride + 1;
    const long long M = (long long)d->N * Ho * Wo;
    const int cout_pad = conv_cout_pad(d->cout, M);
    const int ksteps = conv_ksteps(d->kh, d->kw, d->c0, d->c1);
    std::vector<float> pw((size_t)(ksteps * 4 + 8) * cout_pad * 4), pb(cout_pad);  // slack: see make_conv
    conv_pack_weights(h_w, h_bias, d->cout, d->c0, d->c1, d->kh, d->kw, cout_pad, nullptr, nullptr, pw.data(),
                      pb.data());
    float *dw = nullptr, *db = nullptr;
    DFVO_HIP_CHECK(hipMalloc((void**)&dw, pw.size() * sizeof(float)));
    DFVO_HIP_CHECK(hipMalloc((void**)&db, pb.size() * sizeof(float)));
    DFVO_HIP_CHECK(hipMemcpy(dw, pw.data(), pw.size() * sizeof(float), hipMemcpyHostToDevice));
    DFVO_HIP_CHECK(hipMemcpy(db, pb.data(), pb.size() * sizeof(float), hipMemcpyHostToDevice));
    ConvLayer L;
    L.wp = dw;
    if (make_f16s_weights(h_w, d->cout, d->c0, d->c1, d->kh, d->kw, nullptr, &L) != DFVO_OK) {
        (void)hipFree(dw);
        (void)hipFree(db);
        return DFVO_ERR_HIP;
    }
    if (make_f16g_weights(h_w, d->cout, d->c0, d->c1, d->kh, d->kw, nullptr, &L) != DFVO_OK) {
        (void)hipFree(dw);
        (void)hipFree(db);
        if (L.wf) (void)hipFree(L.wf);
        return DFVO_ERR_HIP;
    }
    {
        const int hrc = make_head_weights(h_w, d->cout, d->c0, d->c1, d->kh, d->kw, nullptr, &L.wh);
        if (hrc != DFVO_OK) {
            (void)hipFree(dw);
            (void)hipFree(db);
            return hrc;
        }
    }
    L.bias = db;
    L.cout = d->cout;
    L.cout_pad = cout_pad;
    L.c0 = d->c0;
    L.c1 = d->c1;
    L.kh = d->kh;
    L.kw = d->kw;
    L.ksteps = ksteps;
    L.stride = d->stride;
    L.pad_h = d->pad_h;
    L.pad_w = d->pad_w;
    L.pad_mode = d->pad_mode;
    L.act = d->act;
    L.act_param = d->act_param;
    // split-K workspace + tile tickets (as the nets own one each): process-wide, calls are serialised by the sync below
    static DevBuf conv2d_ws;
    static std::mutex conv2d_mu;
    std::lock_guard<std::mutex> conv2d_lock(conv2d_mu);
    if (!conv2d_ws.p && conv2d_ws.alloc((size_t)4 << 20) != DFVO_OK) conv2d_ws.p = nullptr;
    int rc = run_conv(L, d->N, d->H, d->W, View{d_src0, d->cs0, d->co0}, d->up0, View{d_src1, d->cs1, d->co1}, d_res,
                      d->res_cs, d->res_co, d_dst, d->dst_cs, d->dst_co, 0, s, nullptr, conv2d_ws.p ? &conv2d_ws : nullptr);
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    (void)hipFree(db);
    if (L.wh) (void)hipFree(L.wh);
    if (L.wf) (void)hipFree(L.wf);
    if (L.wg) (void)hipFree(L.wg);
    if (L.wg32) (void)hipFree(L.wg32);
    if (L.gtab) (void)hipFree(L.gtab);
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(e);
    return DFVO_OK;
}

int dfvo_set_conv_precision(const char* name) { return conv_set_precision(name); }
const char* dfvo_get_conv_precision(void) { return conv_get_precision(); }
int dfvo_f16s_overflow_count(unsigned long long* h_count, int reset) { return conv_f16s_overflow_count(h_count, reset); }

int dfvo_conv_profile_begin(void) {
    conv_profile_begin();
    return DFVO_OK;
}
int dfvo_conv_profile_end(double* h_ms19, double* h_flops19, int* h_launches19) {
    DFVO_ARG_CHECK(h_ms19 && h_flops19 && h_launches19, "dfvo_conv_profile_end: null argument");
    return conv_profile_end(h_ms19, h_flops19, h_launches19);
}
int dfvo_conv_profile_end_bytes(double* h_ms24, double* h_flops24, int* h_launches24, double* h_bytes24) {
    DFVO_ARG_CHECK(h_ms24 && h_flops24 && h_launches24 && h_bytes24, "dfvo_conv_profile_end_bytes: null argument");
    return conv_profile_end(h_ms24, h_flops24, h_launches24, h_bytes24);
}

int dfvo_correlation(const float* d_first, const float* d_second, int N, int H, int W, int C, int stride,
                     float slope, float* d_out, void* stream) {
    DFVO_ARG_CHECK(d_first && d_second && d_out, "dfvo_correlation: null argument");
    DFVO_ARG_CHECK(stride == 1 || stride == 2, "dfvo_correlation: stride must be 1 or 2");
    return launch_correlation(d_first, C, 0, d_second, C, 0, 0, N, H, W, C, stride, d_out, 52, slope,
                              (hipStream_t)stream);
}

static std::vector<float> lin_default(int n) {
    std::vector<float> v(n);
    const float step = n > 1 ? 2.f / (float)(n - 1) : 0.f;
    const int half = n / 2;
    for (int i = 0; i < n; ++i) v[i] = i < half ? -1.f + step * (float)i : 1.f - step * (float)(n - i - 1);
    return v;
}

int dfvo_backward_warp(const float* d_src, const float* d_flow, float mult, int N, int H, int W, int C,
                       const float* h_lin_x, const float* h_lin_y, float* d_dst, void* stream) {
    DFVO_ARG_CHECK(d_src && d_flow && d_dst, "dfvo_backward_warp: null argument");
    hipStream_t s = (hipStream_t)stream;
    std::vector<float> lx = h_lin_x ? std::vector<float>(h_lin_x, h_lin_x + W) : lin_default(W);
    std::vector<float> ly = h_lin_y ? std::vector<float>(h_lin_y, h_lin_y + H) : lin_default(H);
    float *dx = nullptr, *dy = nullptr;
    DFVO_HIP_CHECK(hipMalloc((void**)&dx, W * sizeof(float)));
    DFVO_HIP_CHECK(hipMalloc((void**)&dy, H * sizeof(float)));
    DFVO_HIP_CHECK(hipMemcpy(dx, lx.data(), W * sizeof(float), hipMemcpyHostToDevice));
    DFVO_HIP_CHECK(hipMemcpy(dy, ly.data(), H * sizeof(float), hipMemcpyHostToDevice));
    int rc = launch_warp(d_src, C, 0, 0, d_flow, 2, 0, mult, N, H, W, C, dx, dy, d_dst, C, 0, 0, s);
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(dx);
    (void)hipFree(dy);
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(e);
    return DFVO_OK;
}

int dfvo_deconv_dw4x4s2(const float* d_src, int N, int H, int W, int C, int cs, const float* h_weight, float* d_dst,
                        void* stream) {
    DFVO_ARG_CHECK(d_src && h_weight && d_dst && cs >= C, "dfvo_deconv_dw4x4s2: bad argument");
    hipStream_t s = (hipStream_t)stream;
    float* dw = nullptr;
    DFVO_HIP_CHECK(hipMalloc((void**)&dw, (size_t)C * 16 * sizeof(float)));
    DFVO_HIP_CHECK(hipMemcpy(dw, h_weight, (size_t)C * 16 * sizeof(float), hipMemcpyHostToDevice));
    int rc = launch_deconv_dw(d_src, cs, 0, N, H, W, C, dw, d_dst, cs, 0, s);
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(e);
    return DFVO_OK;
}

int dfvo_resize_bilinear(const float* d_src, int N, int H, int W, int C, float* d_dst, int Ho, int Wo,
                         int align_corners, void* stream) {
    DFVO_ARG_CHECK(d_src && d_dst, "dfvo_resize_bilinear: null argument");
    return launch_resize_bilinear(d_src, N, H, W, C, d_dst, Ho, Wo, align_corners, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
static int store_param(ParamStore* ps, const char* name, const float* h, int ndim, const int* shape) {
    DFVO_ARG_CHECK(name && h && ndim >= 0 && ndim <= 8 && (ndim == 0 || shape), "set_param: bad argument");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        DFVO_ARG_CHECK(shape[i] > 0, "set_param: non-positive dimension");
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(h, h + n);
    ps->t[name] = std::move(t);
    return DFVO_OK;
}

int dfvo_flownet_create(int img_h, int img_w, void* stream, dfvo_flownet** out) {
    DFVO_ARG_CHECK(out, "dfvo_flownet_create: null out");
    dfvo_flownet* n = new dfvo_flownet();
    int rc = n->net.init(img_h, img_w, (hipStream_t)stream);
    if (rc != DFVO_OK) {
        delete n;
        return rc;
    }
    *out = n;
    return DFVO_OK;
}
void dfvo_flownet_destroy(dfvo_flownet* n) {
    if (!n) return;
    n->net.destroy();
    delete n;
}
int dfvo_flownet_set_param(dfvo_flownet* n, const char* name, const float* h, int ndim, const int* shape) {
    DFVO_ARG_CHECK(n, "null net");
    DFVO_ARG_CHECK(!n->net.finalized, "set_param after finalize");
    return store_param(&n->net.params, name, h, ndim, shape);
}
int dfvo_flownet_finalize(dfvo_flownet* n) {
    DFVO_ARG_CHECK(n, "null net");
    return n->net.finalize();
}
int dfvo_flow_target_size(int img_h, int img_w, int* net_h, int* net_w) {
    DFVO_ARG_CHECK(net_h && net_w && img_h >= 32 && img_w >= 32, "dfvo_flow_target_size: bad argument");
    flow_target_size(img_h, img_w, net_h, net_w);
    return DFVO_OK;
}
int dfvo_flownet_net_size(const dfvo_flownet* n, int* h, int* w) {
    DFVO_ARG_CHECK(n && h && w, "null argument");
    *h = n->net.H;
    *w = n->net.W;
    return DFVO_OK;
}
int dfvo_flownet_set_graph(dfvo_flownet* n, int enable) {
    DFVO_ARG_CHECK(n, "null net");
    n->net.use_graph = enable != 0;
    return DFVO_OK;
}
int dfvo_flownet_forward(dfvo_flownet* n, const uint8_t* d_ref, const uint8_t* d_cur, float* d_fwd, float* d_bwd,
                         float* d_diff) {
    DFVO_ARG_CHECK(n && d_ref && d_cur && d_fwd && d_bwd && d_diff, "dfvo_flownet_forward: null argument");
    return n->net.forward(d_ref, d_cur, d_fwd, d_bwd, d_diff);
}
int dfvo_flownet_forward_host(dfvo_flownet* n, const uint8_t* h_ref, const uint8_t* h_cur, float* h_fwd,
                              float* h_bwd, float* h_diff) {
    DFVO_ARG_CHECK(n && h_ref && h_cur && h_fwd && h_bwd && h_diff, "dfvo_flownet_forward_host: null argument");
    FlowNet& f = n->net;
    DFVO_ARG_CHECK(f.finalized, "forward before finalize");
    const size_t ib = (size_t)f.imgH * f.imgW * 3, px = (size_t)f.imgH * f.imgW;
    DFVO_HIP_CHECK(hipMemcpyAsync(f.u8_ref.p, h_ref, ib, hipMemcpyHostToDevice, f.stream));
    DFVO_HIP_CHECK(hipMemcpyAsync(f.u8_cur.p, h_cur, ib, hipMemcpyHostToDevice, f.stream));
    API_TRY(f.forward((const uint8_t*)f.u8_ref.p, (const uint8_t*)f.u8_cur.p, f.out_fwd.p, f.out_bwd.p, f.out_diff.p));
    DFVO_HIP_CHECK(hipMemcpyAsync(h_fwd, f.out_fwd.p, 2 * px * sizeof(float), hipMemcpyDeviceToHost, f.stream));
    DFVO_HIP_CHECK(hipMemcpyAsync(h_bwd, f.out_bwd.p, 2 * px * sizeof(float), hipMemcpyDeviceToHost, f.stream));
    DFVO_HIP_CHECK(hipMemcpyAsync(h_diff, f.out_diff.p, px * sizeof(float), hipMemcpyDeviceToHost, f.stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(f.stream));
    return DFVO_OK;
}
double dfvo_flownet_last_flops(const dfvo_flownet* n) { return n ? n->net.flops_last : 0.0; }
int dfvo_flownet_get_level_flow(dfvo_flownet* n, int level, float* h_out, int* h, int* w) {
    DFVO_ARG_CHECK(n && h_out && level >= 2 && level <= 6, "dfvo_flownet_get_level_flow: bad argument");
    FlowNet& f = n->net;
    DFVO_ARG_CHECK(f.finalized, "get_level_flow before finalize");
    const int hh = f.lh[level], ww = f.lw[level];
    std::vector<float> tmp((size_t)2 * hh * ww * 4);
    DFVO_HIP_CHECK(hipStreamSynchronize(f.stream));
    DFVO_HIP_CHECK(hipMemcpy(tmp.data(), f.lv[level].flow.p, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)2 * hh * ww; ++i) {
        h_out[i * 2] = tmp[i * 4];
        h_out[i * 2 + 1] = tmp[i * 4 + 1];
    }
    if (h) *h = hh;
    if (w) *w = ww;
    return DFVO_OK;
}
int dfvo_flownet_sync(dfvo_flownet* n) {
    DFVO_ARG_CHECK(n, "null net");
    DFVO_HIP_CHECK(hipStreamSynchronize(n->net.stream));
    return DFVO_OK;
}

// ------------------------------------------------------------------------------------------------
int dfvo_depthnet_create(int feed_h, int feed_w, float min_depth, float max_depth, float baseline_mult, void* stream,
                         dfvo_depthnet** out) {
    DFVO_ARG_CHECK(out, "dfvo_depthnet_create: null out");
    dfvo_depthnet* n = new dfvo_depthnet();
    n->net.min_depth = min_depth;
    n->net.max_depth = max_depth;
    n->net.baseline_mult = baseline_mult;
    int rc = n->net.init(feed_h, feed_w, (hipStream_t)stream);
    if (rc != DFVO_OK) {
        delete n;
        return rc;
    }
    *out = n;
    return DFVO_OK;
}
void dfvo_depthnet_destroy(dfvo_depthnet* n) {
    if (!n) return;
    n->net.destroy();
    n->resize.release();
    if (n->img_full) (void)hipFree(n->img_full);
    delete n;
}
int dfvo_depthnet_set_param(dfvo_depthnet* n, const char* name, const float* h, int ndim, const int* shape) {
    DFVO_ARG_CHECK(n, "null net");
    DFVO_ARG_CHECK(!n->net.finalized, "set_param after finalize");
    return store_param(&n->net.params, name, h, ndim, shape);
}
int dfvo_depthnet_finalize(dfvo_depthnet* n) {
    DFVO_ARG_CHECK(n, "null net");
    return n->net.finalize();
}
int dfvo_depthnet_set_graph(dfvo_depthnet* n, int enable) {
    DFVO_ARG_CHECK(n, "null net");
    n->net.use_graph = enable != 0;
    return DFVO_OK;
}
int dfvo_depthnet_forward(dfvo_depthnet* n, const uint8_t* d_img, float* d_depth) {
    DFVO_ARG_CHECK(n && d_img && d_depth, "dfvo_depthnet_forward: null argument");
    return n->net.forward(d_img, d_depth);
}
int dfvo_depthnet_forward_host(dfvo_depthnet* n, const uint8_t* h_img, float* h_depth) {
    DFVO_ARG_CHECK(n && h_img && h_depth, "dfvo_depthnet_forward_host: null argument");
    DepthNet& d = n->net;
    DFVO_ARG_CHECK(d.finalized, "forward before finalize");
    const size_t px = (size_t)d.H * d.W;
    DFVO_HIP_CHECK(hipMemcpyAsync(d.u8_in.p, h_img, px * 3, hipMemcpyHostToDevice, d.stream));
    API_TRY(d.forward((const uint8_t*)d.u8_in.p, d.depth.p));
    DFVO_HIP_CHECK(hipMemcpyAsync(h_depth, d.depth.p, px * sizeof(float), hipMemcpyDeviceToHost, d.stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(d.stream));
    return DFVO_OK;
}
int dfvo_depthnet_forward_image_host(dfvo_depthnet* n, const uint8_t* h_img, int img_h, int img_w, float* h_depth) {
    DFVO_ARG_CHECK(n && h_img && h_depth && img_h > 0 && img_w > 0, "dfvo_depthnet_forward_image_host: bad argument");
    DepthNet& d = n->net;
    DFVO_ARG_CHECK(d.finalized, "forward before finalize");
    if (n->resize.H != img_h || n->resize.W != img_w || n->resize.oh != d.H || n->resize.ow != d.W)
        API_TRY(n->resize.init(img_h, img_w, d.H, d.W));
    const size_t bytes = (size_t)img_h * img_w * 3;
    if (bytes > n->img_full_bytes) {
        if (n->img_full) (void)hipFree(n->img_full);
        n->img_full = nullptr;
        n->img_full_bytes = 0;
        DFVO_HIP_CHECK(hipMalloc((void**)&n->img_full, bytes));
        n->img_full_bytes = bytes;
    }
    const size_t px = (size_t)d.H * d.W;
    DFVO_HIP_CHECK(hipMemcpyAsync(n->img_full, h_img, bytes, hipMemcpyHostToDevice, d.stream));
    API_TRY(n->resize.enqueue(n->img_full, (uint8_t*)d.u8_in.p, d.stream));
    API_TRY(d.forward((const uint8_t*)d.u8_in.p, d.depth.p));
    DFVO_HIP_CHECK(hipMemcpyAsync(h_depth, d.depth.p, px * sizeof(float), hipMemcpyDeviceToHost, d.stream));
    DFVO_HIP_CHECK(hipStreamSynchronize(d.stream));
    return DFVO_OK;
}
int dfvo_lanczos_coeffs(int in_size, int out_size, int* h_bounds, int* h_coeffs, int coeff_cap, int* ksize) {
    return lanczos_coeffs_host(in_size, out_size, h_bounds, h_coeffs, coeff_cap, ksize);
}
int dfvo_resize_lanczos_u8(const uint8_t* d_src, int H, int W, uint8_t* d_dst, int out_h, int out_w, void* stream) {
    DFVO_ARG_CHECK(d_src && d_dst, "dfvo_resize_lanczos_u8: null argument");
    LanczosResizer r;
    int rc = r.init(H, W, out_h, out_w);
    if (rc == DFVO_OK) rc = r.enqueue(d_src, d_dst, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);  // the tables die with r
    r.release();
    if (rc != DFVO_OK) return rc;
    DFVO_HIP_CHECK(e);
    return DFVO_OK;
}
int dfvo_resize_linear_u8(const uint8_t* d_src, int H, int W, int C, uint8_t* d_dst, int out_h, int out_w, void* stream) {
    return enqueue_resize_linear_u8(d_src, H, W, C, d_dst, out_h, out_w, (hipStream_t)stream);
}
int dfvo_read_image_tail_u8(const uint8_t* d_decoded, int img_h, int img_w, int bgr, int y0, int y1, int x0, int x1, uint8_t* d_dst,
                            int out_h, int out_w, void* stream) {
    DFVO_ARG_CHECK(d_decoded && d_dst && img_h > 0 && img_w > 0 && 0 <= y0 && y0 < y1 && y1 <= img_h && 0 <= x0 && x0 < x1 && x1 <= img_w,
                   "dfvo_read_image_tail_u8: crop outside the frame");
    return enqueue_resize_linear_u8(d_decoded + ((size_t)y0 * img_w + x0) * 3, y1 - y0, x1 - x0, 3, d_dst, out_h, out_w,
                                    (hipStream_t)stream, img_w, bgr);
}
double dfvo_depthnet_last_flops(const dfvo_depthnet* n) { return n ? n->net.flops_last : 0.0; }
int dfvo_depthnet_sync(dfvo_depthnet* n) {
    DFVO_ARG_CHECK(n, "null net");
    DFVO_HIP_CHECK(hipStreamSynchronize(n->net.stream));
    return DFVO_OK;
}
int dfvo_depth_postprocess(const float* d_depth, int h, int w, int H, int W, int y0, int y1, int x0, int x1,
                           float min_depth, float max_depth, float* d_raw, double* d_proc, void* stream) {
    DFVO_ARG_CHECK(d_depth && d_raw && d_proc, "dfvo_depth_postprocess: null argument");
    return launch_depth_post(d_depth, h, w, H, W, y0, y1, x0, x1, min_depth, max_depth, d_raw, d_proc,
                             (hipStream_t)stream);
}

}  // extern "C"
