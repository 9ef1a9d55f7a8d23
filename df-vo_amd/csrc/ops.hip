// Streaming operators of LiteFlowNet / monodepth2 (everything that is not a dense contraction).
// All of them are HBM-bound: one pass over the input, one over the output, coalesced along the
// NHWC channel/pixel axis, 16-byte accesses where the layout allows.
#include "ops.h"

namespace dfvo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static inline unsigned grid1d(long long n, int block) { return (unsigned)((n + block - 1) / block); }

// ---------------------------------------------------------------------------------------------
// image preparation
// ---------------------------------------------------------------------------------------------
__global__ void k_img_u8_to_flow_input(const uint8_t* __restrict__ img, int H, int W, float* __restrict__ dst,
                                       int th, int tw, float sh, float sw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= th * tw) return;
    const int oy = idx / tw, ox = idx - oy * tw;
    const float fy = sh * oy, fx = sw * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly1 = fminf(fmaxf(fy - y0, 0.f), 1.f), lx1 = fminf(fmaxf(fx - x0, 0.f), 1.f);
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v00 = (float)((double)img[(y0 * W + x0) * 3 + c] / 255.0);
        const float v01 = (float)((double)img[(y0 * W + x1) * 3 + c] / 255.0);
        const float v10 = (float)((double)img[(y1 * W + x0) * 3 + c] / 255.0);
        const float v11 = (float)((double)img[(y1 * W + x1) * 3 + c] / 255.0);
        o[c] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    }
    *reinterpret_cast<f32x4*>(dst + (size_t)idx * 4) = o;
}

int launch_img_u8_to_flow_input(const uint8_t* img, int H, int W, float* dst, int th, int tw, hipStream_t s) {
    const float sh = th > 1 ? (float)(H - 1) / (float)(th - 1) : 0.f;
    const float sw = tw > 1 ? (float)(W - 1) / (float)(tw - 1) : 0.f;
    hipLaunchKernelGGL(k_img_u8_to_flow_input, dim3(grid1d((long long)th * tw, 256)), dim3(256), 0, s, img, H, W,
                       dst, th, tw, sh, sw);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

__global__ void k_img_u8_to_depth_input(const uint8_t* __restrict__ img, int n, float* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (float)img[idx * 3 + c] / 255.f;  // ToTensor: byte -> float, div(255)
        o[c] = (v - 0.45f) / 0.225f;
    }
    *reinterpret_cast<f32x4*>(dst + (size_t)idx * 4) = o;
}

int launch_img_u8_to_depth_input(const uint8_t* img, int H, int W, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(k_img_u8_to_depth_input, dim3(grid1d((long long)H * W, 256)), dim3(256), 0, s, img, H * W,
                       dst);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ---------------------------------------------------------------------------------------------
// bilinear resize, NHWC dense, C % 4 == 0
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, bool ac, int& i0, int& i1, float& l0,
                                          float& l1) {
    float real;
    if (ac) {
        real = scale * dst;
    } else {
        real = (float)((double)scale * (dst + 0.5) - 0.5);
        if (real < 0.f) real = 0.f;
    }
    i0 = (int)real;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(real - i0, 0.f), 1.f);
    l0 = 1.f - l1;
}

__global__ void k_resize_bilinear(const float* __restrict__ src, int N, int H, int W, int C4, float* __restrict__ dst,
                                  int Ho, int Wo, float sh, float sw, int ac) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * Ho * Wo * C4;
    if (idx >= total) return;
    const int c = (int)(idx % C4);
    long long pix = idx / C4;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(sh, oy, H, ac, y0, y1, ly0, ly1);
    src_index(sw, ox, W, ac, x0, x1, lx0, lx1);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    const f32x4 v00 = s4[((size_t)(n * H + y0) * W + x0) * C4 + c];
    const f32x4 v01 = s4[((size_t)(n * H + y0) * W + x1) * C4 + c];
    const f32x4 v10 = s4[((size_t)(n * H + y1) * W + x0) * C4 + c];
    const f32x4 v11 = s4[((size_t)(n * H + y1) * W + x1) * C4 + c];
    reinterpret_cast<f32x4*>(dst)[idx] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
}

int launch_resize_bilinear(const float* src, int N, int H, int W, int C, float* dst, int Ho, int Wo,
                           int align_corners, hipStream_t s) {
    DFVO_ARG_CHECK(C % 4 == 0, "resize_bilinear: C % 4");
    float sh, sw;
    if (align_corners) {
        sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
        sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    } else {
        sh = (float)H / (float)Ho;
        sw = (float)W / (float)Wo;
    }
    const long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_resize_bilinear, dim3(grid1d(total, 256)), dim3(256), 0, s, src, N, H, W, C / 4, dst, Ho,
                       Wo, sh, sw, align_corners);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ---------------------------------------------------------------------------------------------
// grid_sample helpers (bilinear, zeros padding, align_corners=True)
// ---------------------------------------------------------------------------------------------
struct Bilin {
    int x0, y0;
    float nw, ne, sw, se;
    bool vx0, vx1, vy0, vy1;
};

__device__ __forceinline__ Bilin bilin_setup(float gx, float gy, int W, int H) {
    // GridSamplerKernel.cpp: unnormalize(in) = (in + 1) * ((size - 1) / 2)
    const float ix = (gx + 1.f) * ((float)(W - 1) / 2.f);
    const float iy = (gy + 1.f) * ((float)(H - 1) / 2.f);
    const float xw = floorf(ix), yn = floorf(iy);
    const float w = ix - xw, e = 1.f - w, n = iy - yn, s = 1.f - n;
    Bilin b;
    b.x0 = (int)xw;
    b.y0 = (int)yn;
    b.nw = s * e;
    b.ne = s * w;
    b.sw = n * e;
    b.se = n * w;
    b.vx0 = b.x0 >= 0 && b.x0 < W;
    b.vx1 = b.x0 + 1 >= 0 && b.x0 + 1 < W;
    b.vy0 = b.y0 >= 0 && b.y0 < H;
    b.vy1 = b.y0 + 1 >= 0 && b.y0 + 1 < H;
    // guard against NaN/huge coordinates turning into wild integer indices
    if (!(ix > -2.f && ix < (float)W + 1.f)) b.vx0 = b.vx1 = false;
    if (!(iy > -2.f && iy < (float)H + 1.f)) b.vy0 = b.vy1 = false;
    return b;
}

__global__ void k_warp(const float* __restrict__ src, int scs, int sco, int swap, const float* __restrict__ flow,
                       int fcs, int fco, float mult, int N, int H, int W, int C4, const float* __restrict__ lin_x,
                       const float* __restrict__ lin_y, float* __restrict__ dst, int dcs, int dco, int append_flow, int step) {
    // step > 1: only the pixels (y % step == 0, x % step == 0) are produced -- all a stride-`step` correlation reads of its
    // second operand (levels 3 and 2: a quarter of the map)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_pix = C4 + (append_flow ? 1 : 0);
    const int Hs = (H + step - 1) / step, Ws = (W + step - 1) / step;
    const long long total = (long long)N * Hs * Ws * per_pix;
    if (idx >= total) return;
    const int c = (int)(idx % per_pix);
    const long long spix = idx / per_pix;
    const int x = (int)(spix % Ws) * step;
    const long long row = spix / Ws;
    const int y = (int)(row % Hs) * step;
    const int n = (int)(row / Hs);
    const long long pix = ((long long)n * H + y) * W + x;
    const float fx = flow[pix * fcs + fco], fy = flow[pix * fcs + fco + 1];
    if (c == C4) {
        *reinterpret_cast<f32x4*>(dst + pix * dcs + dco + C4 * 4) = f32x4{fx, fy, 0.f, 0.f};
        return;
    }
    const float gx = lin_x[x] + (fx * mult) / ((float)(W - 1) / 2.f);
    const float gy = lin_y[y] + (fy * mult) / ((float)(H - 1) / 2.f);
    const Bilin b = bilin_setup(gx, gy, W, H);
    const int ns = swap ? (N - 1 - n) : n;
    const float* base = src + (size_t)ns * H * W * scs + sco + c * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 vnw = (b.vy0 && b.vx0) ? *reinterpret_cast<const f32x4*>(base + ((size_t)b.y0 * W + b.x0) * scs) : z;
    const f32x4 vne =
        (b.vy0 && b.vx1) ? *reinterpret_cast<const f32x4*>(base + ((size_t)b.y0 * W + b.x0 + 1) * scs) : z;
    const f32x4 vsw =
        (b.vy1 && b.vx0) ? *reinterpret_cast<const f32x4*>(base + ((size_t)(b.y0 + 1) * W + b.x0) * scs) : z;
    const f32x4 vse =
        (b.vy1 && b.vx1) ? *reinterpret_cast<const f32x4*>(base + ((size_t)(b.y0 + 1) * W + b.x0 + 1) * scs) : z;
    *reinterpret_cast<f32x4*>(dst + pix * dcs + dco + c * 4) = vnw * b.nw + vne * b.ne + vsw * b.sw + vse * b.se;
}

int launch_warp(const float* src, int scs, int sco, int swap, const float* flow, int fcs, int fco, float mult,
                int N, int H, int W, int C, const float* lin_x, const float* lin_y, float* dst, int dcs, int dco,
                int append_flow, hipStream_t s, int step) {
    DFVO_ARG_CHECK(C % 4 == 0 && scs % 4 == 0 && sco % 4 == 0 && dcs % 4 == 0 && dco % 4 == 0 && step >= 1, "warp: alignment");
    const long long total = (long long)N * cdiv(H, step) * cdiv(W, step) * (C / 4 + (append_flow ? 1 : 0));
    hipLaunchKernelGGL(k_warp, dim3(grid1d(total, 256)), dim3(256), 0, s, src, scs, sco, swap, flow, fcs, fco, mult,
                       N, H, W, C / 4, lin_x, lin_y, dst, dcs, dco, append_flow, step);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ---------------------------------------------------------------------------------------------
// flow mean (per sample, 2 channels)
// ---------------------------------------------------------------------------------------------
// Round 6: the single 1024-thread workgroup per sample of rounds 1-5 read a level-2 map (3.4 MB) alone -- 35 us of a chain in
// which every microsecond is latency.  Now FM_G workgroups per sample write double partial sums; the workgroup that takes the
// last ticket adds the FM_G partials IN INDEX ORDER and rounds once: deterministic, and (the sums being sums of floats in
// double, far from 53 bits) the same float as before.
constexpr int FM_G = 64;
__global__ __launch_bounds__(256) void k_flow_mean(const float* __restrict__ flow, int fcs, int fco, int HW,
                                                    double* __restrict__ partial, unsigned* __restrict__ ticket,
                                                    float* __restrict__ mean) {
    __shared__ double sx[256], sy[256];
    __shared__ bool last;
    const int n = blockIdx.y, g = blockIdx.x;
    const float* f = flow + (size_t)n * HW * fcs + fco;
    const int per = (HW + FM_G - 1) / FM_G, lo = g * per, hi = min(HW, lo + per);
    double ax = 0., ay = 0.;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(f + (size_t)i * fcs);
        ax += v[0];
        ay += v[1];
    }
    sx[threadIdx.x] = ax;
    sy[threadIdx.x] = ay;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sx[threadIdx.x] += sx[threadIdx.x + o];
            sy[threadIdx.x] += sy[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* pp = partial + ((size_t)n * FM_G + g) * 2;
        __hip_atomic_store(pp, sx[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 1, sy[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t = __hip_atomic_fetch_add(ticket + n, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = t == FM_G - 1;
    }
    __syncthreads();
    if (!last || threadIdx.x != 0) return;
    double tx = 0., ty = 0.;
    for (int j = 0; j < FM_G; ++j) {
        tx += __hip_atomic_load(partial + ((size_t)n * FM_G + j) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ty += __hip_atomic_load(partial + ((size_t)n * FM_G + j) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    mean[n * 2] = (float)(tx / (double)HW);
    mean[n * 2 + 1] = (float)(ty / (double)HW);
    __hip_atomic_store(ticket + n, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
}

size_t flow_mean_scratch_floats(int N) { return (size_t)N * FM_G * 2 * 2 + (size_t)N + 4; }  // doubles, then the tickets

// scratch: flow_mean_scratch_floats(N) floats, 8-byte aligned, ZERO when first used (the kernel leaves the tickets at zero)
int launch_flow_mean(const float* flow, int fcs, int fco, int N, int HW, float* scratch, float* mean,
                     hipStream_t s) {
    DFVO_ARG_CHECK(scratch && (fcs % 2) == 0 && (fco % 2) == 0 && ((uintptr_t)flow & 7) == 0 && ((uintptr_t)scratch & 7) == 0,
                   "flow_mean: scratch / 8-byte aligned flow view required");
    double* partial = reinterpret_cast<double*>(scratch);
    unsigned* ticket = reinterpret_cast<unsigned*>(scratch + (size_t)N * FM_G * 2 * 2);
    hipLaunchKernelGGL(k_flow_mean, dim3(FM_G, N), dim3(256), 0, s, flow, fcs, fco, HW, partial, ticket, mean);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ---------------------------------------------------------------------------------------------
// Regularization input: brightness error + mean-subtracted flow
// ---------------------------------------------------------------------------------------------
__global__ void k_reg_prep(const float* __restrict__ img, const float* __restrict__ flow, int fcs, int fco,
                           float mult, const float* __restrict__ mean, int N, int H, int W,
                           const float* __restrict__ lin_x, const float* __restrict__ lin_y,
                           float* __restrict__ dst) {
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (long long)N * H * W) return;
    const int x = (int)(pix % W);
    const long long row = pix / W;
    const int y = (int)(row % H);
    const int n = (int)(row / H);
    const float fx = flow[pix * fcs + fco], fy = flow[pix * fcs + fco + 1];
    const float gx = lin_x[x] + (fx * mult) / ((float)(W - 1) / 2.f);
    const float gy = lin_y[y] + (fy * mult) / ((float)(H - 1) / 2.f);
    const Bilin b = bilin_setup(gx, gy, W, H);
    const float* base = img + (size_t)(N - 1 - n) * H * W * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 vnw = (b.vy0 && b.vx0) ? *reinterpret_cast<const f32x4*>(base + ((size_t)b.y0 * W + b.x0) * 4) : z;
    const f32x4 vne = (b.vy0 && b.vx1) ? *reinterpret_cast<const f32x4*>(base + ((size_t)b.y0 * W + b.x0 + 1) * 4) : z;
    const f32x4 vsw = (b.vy1 && b.vx0) ? *reinterpret_cast<const f32x4*>(base + ((size_t)(b.y0 + 1) * W + b.x0) * 4) : z;
    const f32x4 vse =
        (b.vy1 && b.vx1) ? *reinterpret_cast<const f32x4*>(base + ((size_t)(b.y0 + 1) * W + b.x0 + 1) * 4) : z;
    const f32x4 wv = vnw * b.nw + vne * b.ne + vsw * b.sw + vse * b.se;
    const f32x4 a = *reinterpret_cast<const f32x4*>(img + pix * 4);
    const float d0 = a[0] - wv[0], d1 = a[1] - wv[1], d2 = a[2] - wv[2];
    const float ssum = (d0 * d0 + d1 * d1) + d2 * d2;
    const float diff = sqrtf(ssum + 1e-6f);
    *reinterpret_cast<f32x4*>(dst + pix * 4) = f32x4{diff, fx - mean[n * 2], fy - mean[n * 2 + 1], 0.f};
}

int launch_reg_prep(const float* img, const float* flow, int fcs, int fco, float mult, const float* mean, int N,
                    int H, int W, const float* lin_x, const float* lin_y, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(k_reg_prep, dim3(grid1d((long long)N * H * W, 256)), dim3(256), 0, s, img, flow, fcs, fco,
                       mult, mean, N, H, W, lin_x, lin_y, dst);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ---------------------------------------------------------------------------------------------
// depthwise transposed conv 4x4 stride 2 pad 1
// ---------------------------------------------------------------------------------------------
__global__ void k_deconv_dw(const float* __restrict__ src, int scs, int sco, int N, int H, int W, int C,
                            const float* __restrict__ w, float* __restrict__ dst, int dcs, int dco) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)N * Ho * Wo * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    const int ox = (int)(pix % Wo);
    const long long row = pix / Wo;
    const int oy = (int)(row % Ho);
    const int n = (int)(row / Ho);
    float acc = 0.f;
    // oy = iy*2 - 1 + ky  ->  ky has the parity of oy+1; iterate input rows in increasing order
#pragma unroll
    for (int a = 1; a >= 0; --a) {
        const int ky = ((oy + 1) & 1) + 2 * a;
        const int iy = (oy + 1 - ky) >> 1;
        if ((oy + 1 - ky) < 0 || iy >= H) continue;
#pragma unroll
        for (int bq = 1; bq >= 0; --bq) {
            const int kx = ((ox + 1) & 1) + 2 * bq;
            const int ix = (ox + 1 - kx) >> 1;
            if ((ox + 1 - kx) < 0 || ix >= W) continue;
            acc += src[((size_t)(n * H + iy) * W + ix) * scs + sco + c] * w[c * 16 + ky * 4 + kx];
        }
    }
    dst[pix * dcs + dco + c] = acc;
}

// the same, four channels per thread (16-byte loads / stores; views must be 16-byte aligned): a thread's four outputs are
// four independent sums in the scalar kernel's tap order, so the values are bit-identical.  Channels c >= C of the last
// quad are written as zeros (the consumers read the padded channel groups against zero weights: they must be finite).
__global__ __launch_bounds__(256) void k_deconv_dw4(const float* __restrict__ src, int scs, int sco, int N, int H, int W, int C,
                                                    const float* __restrict__ w, float* __restrict__ dst, int dcs, int dco) {
    const int C4 = (C + 3) >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)N * Ho * Wo * C4;
    if (idx >= total) return;
    const int q = (int)(idx % C4);
    const long long pix = idx / C4;
    const int ox = (int)(pix % Wo);
    const long long row = pix / Wo;
    const int oy = (int)(row % Ho);
    const int n = (int)(row / Ho);
    const int c0 = q * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 1; a >= 0; --a) {
        const int ky = ((oy + 1) & 1) + 2 * a;
        const int iy = (oy + 1 - ky) >> 1;
        if ((oy + 1 - ky) < 0 || iy >= H) continue;
#pragma unroll
        for (int bq = 1; bq >= 0; --bq) {
            const int kx = ((ox + 1) & 1) + 2 * bq;
            const int ix = (ox + 1 - kx) >> 1;
            if ((ox + 1 - kx) < 0 || ix >= W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((size_t)(n * H + iy) * W + ix) * scs + sco + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < C) acc[e] += v[e] * w[(c0 + e) * 16 + ky * 4 + kx];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c0 + e >= C) acc[e] = 0.f;
    *reinterpret_cast<f32x4*>(dst + pix * dcs + dco + c0) = acc;
}

// Round 6: the same sums, one thread per INPUT pixel and channel quad = a 2 x 2 block of outputs.  The four outputs share
// the 3 x 3 input neighbourhood (9 vector loads instead of 16) and the sixteen taps of the quad's channels, which the
// workgroup first transposes into LDS ([tap][channel]: one 16-byte read per tap instead of four scattered 4-byte loads per
// tap and output).  Per output the taps are added in k_deconv_dw4's order (ky high to low, kx high to low, one fma each):
// bit-identical.  Level-2 correlation upsampling (49 channels, 176 x 608 x 2 outputs): 83 -> ~30 us.
__global__ __launch_bounds__(256) void k_deconv_dw4_blk(const float* __restrict__ src, int scs, int sco, int N, int H, int W, int C,
                                                        const float* __restrict__ w, float* __restrict__ dst, int dcs, int dco) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [16][C4 * 4]
    const int C4 = (C + 3) >> 2, CP = C4 * 4;
    for (int i = threadIdx.x; i < 16 * CP; i += 256) {
        const int tap = i / CP, c = i - tap * CP;
        wl[i] = c < C ? w[c * 16 + tap] : 0.f;
    }
    __syncthreads();
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * H * W * C4;
    if (idx >= total) return;
    const int q = (int)(idx % C4);
    const long long ipix = idx / C4;
    const int j = (int)(ipix % W);
    const long long row = ipix / W;
    const int i = (int)(row % H);
    const int n = (int)(row / H);
    const int c0 = q * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 in[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int iy = i + dy - 1, ix = j + dx - 1;
            in[dy][dx] = (iy >= 0 && iy < H && ix >= 0 && ix < W)
                             ? *reinterpret_cast<const f32x4*>(src + ((size_t)(n * H + iy) * W + ix) * scs + sco + c0) : z;
        }
    const int Wo = 2 * W;
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            // output (2i + py, 2j + px): ky = (1 - py) + 2a reads input row i + py - a (a = 1 first), likewise in x
            f32x4 acc = z;
#pragma unroll
            for (int a = 1; a >= 0; --a) {
                const int ky = (1 - py) + 2 * a, dy = py - a + 1;  // row i + py - a  ->  in[dy]
                if (i + py - a < 0 || i + py - a >= H) continue;
#pragma unroll
                for (int b = 1; b >= 0; --b) {
                    const int kx = (1 - px) + 2 * b, dx = px - b + 1;
                    if (j + px - b < 0 || j + px - b >= W) continue;
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + (ky * 4 + kx) * CP + c0);
                    const f32x4 v = in[dy][dx];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += v[e] * wv[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e >= C) acc[e] = 0.f;
            const size_t opix = ((size_t)n * 2 * H + 2 * i + py) * Wo + 2 * j + px;
            *reinterpret_cast<f32x4*>(dst + opix * dcs + dco + c0) = acc;
        }
}

int launch_deconv_dw(const float* src, int scs, int sco, int N, int H, int W, int C, const float* w, float* dst,
                     int dcs, int dco, hipStream_t s) {
    const int C4 = (C + 3) >> 2;
    if (((scs | sco | dcs | dco) & 3) == 0 && C4 * 4 <= scs - sco && C4 * 4 <= dcs - dco) {  // 16-byte views with room for the quad
        static const bool blk = !(getenv("DFVO_DECONV_BLK") && atoi(getenv("DFVO_DECONV_BLK")) == 0);
        if (blk && C4 * 4 * 16 * sizeof(float) <= 32 * 1024) {
            const long long totalb = (long long)N * H * W * C4;
            hipLaunchKernelGGL(k_deconv_dw4_blk, dim3(grid1d(totalb, 256)), dim3(256), (size_t)C4 * 4 * 16 * sizeof(float), s, src, scs, sco,
                               N, H, W, C, w, dst, dcs, dco);
            DFVO_HIP_CHECK(hipGetLastError());
            return DFVO_OK;
        }
        const long long total4 = (long long)N * 2 * H * 2 * W * C4;
        hipLaunchKernelGGL(k_deconv_dw4, dim3(grid1d(total4, 256)), dim3(256), 0, s, src, scs, sco, N, H, W, C, w, dst, dcs, dco);
        DFVO_HIP_CHECK(hipGetLastError());
        return DFVO_OK;
    }
    const long long total = (long long)N * 2 * H * 2 * W * C;
    hipLaunchKernelGGL(k_deconv_dw, dim3(grid1d(total, 256)), dim3(256), 0, s, src, scs, sco, N, H, W, C, w, dst,
                       dcs, dco);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ---------------------------------------------------------------------------------------------
// correlation volume.  One block = TH x TW output pixels.  The first-image tile and the
// second-image halo tile ((TH+6) x (TW+6)) are staged in LDS with a pixel stride of C+1 floats so
// that lanes reading the same channel of different pixels fall on different banks.  Each thread
// owns (pixel, displacement) items and reproduces the reference kernel's summation order: 32
// partial sums over channels c = j, j+32, ... (fmaf chains), added in order j = 0..31, divided by C.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_correlation(const float* __restrict__ f1, int cs1, int co1,
                                                      const float* __restrict__ f2, int cs2, int co2, int swap2,
                                                      int N, int H, int W, int C, int stride, int Ho, int Wo, int TH,
                                                      int TW, float* __restrict__ dst, int dcs, float slope) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = C + 1;
    const int HW2 = TW + 6, HH2 = TH + 6;
    float* s1 = smem;                 // [TH*TW][P]
    float* s2 = smem + TH * TW * P;   // [HH2*HW2][P]
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int n2 = swap2 ? (N - 1 - n) : n;
    const int C4 = C >> 2;
    // stage first-image tile
    for (int it = threadIdx.x; it < TH * TW * C4; it += 256) {
        const int c = it % C4, pp = it / C4;
        const int py = pp / TW, px = pp - py * TW;
        const int oy = oy0 + py, ox = ox0 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (oy < Ho && ox < Wo)
            v = *reinterpret_cast<const f32x4*>(f1 + ((size_t)(n * H + oy * stride) * W + ox * stride) * cs1 + co1 +
                                                c * 4);
        float* d = s1 + pp * P + c * 4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    // stage second-image halo tile (zero outside the image)
    for (int it = threadIdx.x; it < HH2 * HW2 * C4; it += 256) {
        const int c = it % C4, pp = it / C4;
        const int py = pp / HW2, px = pp - py * HW2;
        const int sy = oy0 + py - 3, sx = ox0 + px - 3;  // coordinates on the stride-subsampled grid
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (sy >= 0 && sy < Ho && sx >= 0 && sx < Wo)
            v = *reinterpret_cast<const f32x4*>(f2 + ((size_t)(n2 * H + sy * stride) * W + sx * stride) * cs2 + co2 +
                                                c * 4);
        float* d = s2 + pp * P + c * 4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    const float invC = (float)C;
    for (int it = threadIdx.x; it < TH * TW * 49; it += 256) {
        const int tc = it % 49, pp = it / 49;
        const int py = pp / TW, px = pp - py * TW;
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy >= Ho || ox >= Wo) continue;
        const int dy = tc / 7, dx = tc - dy * 7;  // (tc/7 - 3, tc%7 - 3) + 3 halo offset
        const float* a = s1 + pp * P;
        const float* b = s2 + ((py + dy) * HW2 + (px + dx)) * P;
        float total = 0.f;
        for (int j = 0; j < 32; ++j) {
            float part = 0.f;
            for (int c = j; c < C; c += 32) part = fmaf(a[c], b[c], part);
            total += part;
        }
        float v = total / invC;
        v = v > 0.f ? v : v * slope;
        dst[((size_t)(n * Ho + oy) * Wo + ox) * dcs + tc] = v;
    }
}

// Register-tiled form for C % 32 == 0 (every level of the flow net).  Same staging, pixel stride C + 4 floats (16-byte
// aligned pixel vectors; consecutive pixels 1 bank-quad apart: conflict-free ds_read_b128 for lanes along x).  A thread
// owns TWO vertically adjacent output pixels x the seven vertical displacements of one horizontal displacement: the eight
// second-image vectors it reads serve 14 (pixel, displacement) items -- 10 b128 reads per 56 FMAs instead of 112 b32
// reads.  Summation order of the reference kernel kept exactly: per item 32 partial sums over c = j, j + 32, ...
// (fmaf chains from 0), added in order j = 0 .. 31 -- four consecutive j per 16-byte read.
template <int K32>  // C / 32
__global__ __launch_bounds__(256) void k_correlation_rt(const float* __restrict__ f1, int cs1, int co1,
                                                         const float* __restrict__ f2, int cs2, int co2, int swap2,
                                                         int N, int H, int W, int stride, int Ho, int Wo, int TH, int TW,
                                                         float* __restrict__ dst, int dcs, float slope) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int C = 32 * K32, P = C + 4, C4 = C / 4;
    const int HW2 = TW + 6, HH2 = TH + 6;
    float* s1 = smem;                 // [TH*TW][P]
    float* s2 = smem + TH * TW * P;   // [HH2*HW2][P]
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int n2 = swap2 ? (N - 1 - n) : n;
    // staging in batches of four items: addresses, then the four loads back to back, then the LDS stores (round 6: one load
    // per iteration left every round trip exposed -- twelve in a row for the halo tile of a 64-channel level)
    constexpr int LB = 4;
    for (int it0 = threadIdx.x; it0 < TH * TW * C4; it0 += 256 * LB) {
        f32x4 v[LB];
        int dsto[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int it = it0 + 256 * u;
            const int c = it % C4, pp = it / C4;
            const int py = pp / TW, px = pp - py * TW;
            const int oy = oy0 + py, ox = ox0 + px;
            dsto[u] = it < TH * TW * C4 ? pp * P + c * 4 : -1;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (dsto[u] >= 0 && oy < Ho && ox < Wo)
                v[u] = *reinterpret_cast<const f32x4*>(f1 + ((size_t)(n * H + oy * stride) * W + ox * stride) * cs1 + co1 + c * 4);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u)
            if (dsto[u] >= 0) *reinterpret_cast<f32x4*>(s1 + dsto[u]) = v[u];
    }
    for (int it0 = threadIdx.x; it0 < HH2 * HW2 * C4; it0 += 256 * LB) {
        f32x4 v[LB];
        int dsto[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int it = it0 + 256 * u;
            const int c = it % C4, pp = it / C4;
            const int py = pp / HW2, px = pp - py * HW2;
            const int sy = oy0 + py - 3, sx = ox0 + px - 3;  // coordinates on the stride-subsampled grid
            dsto[u] = it < HH2 * HW2 * C4 ? pp * P + c * 4 : -1;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (dsto[u] >= 0 && sy >= 0 && sy < Ho && sx >= 0 && sx < Wo)
                v[u] = *reinterpret_cast<const f32x4*>(f2 + ((size_t)(n2 * H + sy * stride) * W + sx * stride) * cs2 + co2 + c * 4);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u)
            if (dsto[u] >= 0) *reinterpret_cast<f32x4*>(s2 + dsto[u]) = v[u];
    }
    __syncthreads();
    const int units = TW * 7 * (TH / 2);
    const int u = threadIdx.x;
    if (u >= units) return;
    const int px = u % TW, r = u / TW, dx = r % 7, yp = r / 7;
    const int py = 2 * yp;
    const float* a0 = s1 + (py * TW + px) * P;
    const float* a1 = a0 + TW * P;
    const float* b = s2 + (py * HW2 + px + dx) * P;  // rows py .. py + 7 of the halo tile, column px + dx
    float tot0[7], tot1[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) tot0[d] = tot1[d] = 0.f;
#pragma unroll 1
    for (int g = 0; g < 8; ++g) {  // partial sums j = 4g .. 4g + 3
        f32x4 p0[7], p1[7];
#pragma unroll
        for (int d = 0; d < 7; ++d) p0[d] = p1[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K32; ++k) {
            const int c = 4 * g + 32 * k;
            const f32x4 A0 = *reinterpret_cast<const f32x4*>(a0 + c);
            const f32x4 A1 = *reinterpret_cast<const f32x4*>(a1 + c);
            f32x4 B[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) B[i] = *reinterpret_cast<const f32x4*>(b + i * HW2 * P + c);
#pragma unroll
            for (int d = 0; d < 7; ++d)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p0[d][e] = fmaf(A0[e], B[d][e], p0[d][e]);
                    p1[d][e] = fmaf(A1[e], B[d + 1][e], p1[d][e]);
                }
        }
#pragma unroll
        for (int d = 0; d < 7; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                tot0[d] += p0[d][e];
                tot1[d] += p1[d][e];
            }
    }
    const float invC = (float)C;
    const int ox = ox0 + px;
    if (ox >= Wo) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int oy = oy0 + py + q;
        if (oy >= Ho) continue;
        float* o = dst + ((size_t)(n * Ho + oy) * Wo + ox) * dcs + dx;
#pragma unroll
        for (int d = 0; d < 7; ++d) {
            float v = (q ? tot1[d] : tot0[d]) / invC;
            v = v > 0.f ? v : v * slope;
            o[d * 7] = v;  // displacement index tc = dy * 7 + dx
        }
    }
}

template <int K32>
static int launch_correlation_rt(const float* f1, int cs1, int co1, const float* f2, int cs2, int co2, int swap2, int N,
                                 int H, int W, int stride, float* dst, int dcs, float slope, hipStream_t s) {
    constexpr int C = 32 * K32, P = C + 4;
    const int Ho = cdiv(H, stride), Wo = cdiv(W, stride);
    int TH = 8, TW = 8;
    if ((size_t)(TH * TW + (TH + 6) * (TW + 6)) * P * sizeof(float) > 150 * 1024) TH = 4;
    const size_t lds = (size_t)(TH * TW + (TH + 6) * (TW + 6)) * P * sizeof(float);
    DFVO_ARG_CHECK(lds <= 160 * 1024, "correlation: channel count too large for the LDS tile");
    if (int rc_lds = ensure_dyn_lds((const void*)k_correlation_rt<K32>, lds)) return rc_lds;
    dim3 grid(cdiv(Wo, TW), cdiv(Ho, TH), N);
    hipLaunchKernelGGL(k_correlation_rt<K32>, grid, dim3(256), lds, s, f1, cs1, co1, f2, cs2, co2, swap2, N, H, W, stride, Ho,
                       Wo, TH, TW, dst, dcs, slope);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

int launch_correlation(const float* f1, int cs1, int co1, const float* f2, int cs2, int co2, int swap2, int N,
                       int H, int W, int C, int stride, float* dst, int dcs, float slope, hipStream_t s) {
    DFVO_ARG_CHECK(C % 4 == 0 && cs1 % 4 == 0 && cs2 % 4 == 0 && co1 % 4 == 0 && co2 % 4 == 0,
                   "correlation: alignment");
    static const bool rt = !(getenv("DFVO_CORR_RT") && atoi(getenv("DFVO_CORR_RT")) == 0);
    if (rt && C % 32 == 0 && C >= 32 && C <= 192) {
        switch (C / 32) {
            case 1: return launch_correlation_rt<1>(f1, cs1, co1, f2, cs2, co2, swap2, N, H, W, stride, dst, dcs, slope, s);
            case 2: return launch_correlation_rt<2>(f1, cs1, co1, f2, cs2, co2, swap2, N, H, W, stride, dst, dcs, slope, s);
            case 3: return launch_correlation_rt<3>(f1, cs1, co1, f2, cs2, co2, swap2, N, H, W, stride, dst, dcs, slope, s);
            case 4: return launch_correlation_rt<4>(f1, cs1, co1, f2, cs2, co2, swap2, N, H, W, stride, dst, dcs, slope, s);
            case 6: return launch_correlation_rt<6>(f1, cs1, co1, f2, cs2, co2, swap2, N, H, W, stride, dst, dcs, slope, s);
            default: break;
        }
    }
    const int Ho = cdiv(H, stride), Wo = cdiv(W, stride);
    int TH = 8, TW = 8;
    if (C > 64) { TH = 4; TW = 8; }
    if (C > 96) { TH = 4; TW = 4; }
    const size_t lds = (size_t)(TH * TW + (TH + 6) * (TW + 6)) * (C + 1) * sizeof(float);
    DFVO_ARG_CHECK(lds <= 160 * 1024, "correlation: channel count too large for the LDS tile");
    if (int rc_lds = ensure_dyn_lds((const void*)k_correlation, lds)) return rc_lds;
    dim3 grid(cdiv(Wo, TW), cdiv(Ho, TH), N);
    hipLaunchKernelGGL(k_correlation, grid, dim3(256), lds, s, f1, cs1, co1, f2, cs2, co2, swap2, N, H, W, C, stride,
                       Ho, Wo, TH, TW, dst, dcs, slope);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ---------------------------------------------------------------------------------------------
// Regularization output head (feature-driven local convolution)
// ---------------------------------------------------------------------------------------------
__global__ void k_reg_head(const float* __restrict__ dist, int dist_cs, int k, const float* __restrict__ flow,
                           int fcs, int fco, const float* __restrict__ wx, float bx, const float* __restrict__ wy,
                           float by, int N, int H, int W, float* __restrict__ dst, int dcs, int dco) {
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (long long)N * H * W) return;
    const int x = (int)(pix % W);
    const long long row = pix / W;
    const int y = (int)(row % H);
    const int n = (int)(row / H);
    const int kk = k * k, r = (k - 1) / 2;
    const float* d = dist + pix * dist_cs;
    float m = -INFINITY;
    for (int c = 0; c < kk; ++c) {
        const float v = -(d[c] * d[c]);
        m = fmaxf(m, v);
    }
    double se = 0., ax = 0., ay = 0.;  // (see k_reg_head_v)
    for (int c = 0; c < kk; ++c) {
        const float v = -(d[c] * d[c]);
        const float e = expf(v - m);
        se += e;
        const int ky = c / k, kx = c - ky * k;
        const int yy = y + ky - r, xx = x + kx - r;
        float ux = 0.f, uy = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const float* f = flow + ((size_t)(n * H + yy) * W + xx) * fcs + fco;
            ux = f[0];
            uy = f[1];
        }
        ax += (double)wx[c] * ((double)e * ux);
        ay += (double)wy[c] * ((double)e * uy);
    }
    dst[pix * dcs + dco] = (float)((ax + bx) / se);
    dst[pix * dcs + dco + 1] = (float)((ay + by) / se);
}

// The same head with the pixel's distance vector read ONCE, as 16-byte loads into registers (a thread's k*k values are
// contiguous and consecutive pixels are adjacent: the first form read them twice, four bytes at a time at a 208-byte lane
// stride), the two flow components of a neighbour as one 8-byte load.  Same operations in the same order.
template <int K>
__global__ __launch_bounds__(256) void k_reg_head_v(const float* __restrict__ dist, int dist_cs, const float* __restrict__ flow,
                                                     int fcs, int fco, const float* __restrict__ wx, float bx,
                                                     const float* __restrict__ wy, float by, int N, int H, int W,
                                                     float* __restrict__ dst, int dcs, int dco) {
    constexpr int KK = K * K, KK4 = (KK + 3) / 4, R = (K - 1) / 2;
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (long long)N * H * W) return;
    const int x = (int)(pix % W);
    const long long row = pix / W;
    const int y = (int)(row % H);
    const int n = (int)(row / H);
    float v[KK4 * 4];
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dist + pix * dist_cs);
#pragma unroll
    for (int i = 0; i < KK4; ++i) {
        const f32x4 q = d4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * i + e] = -(q[e] * q[e]);
    }
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < KK; ++c) m = fmaxf(m, v[c]);
    // The weighted mean itself -- up to 49 terms of the size of the flow, the LAST arithmetic of a level -- is accumulated in
    // double and rounded once: a float chain leaves 1-2 ulp of the flow here (3e-7 px on the level-2 map, x 20 on the output:
    // measured against the float64 anchor, tools/flow_error_by_level.py), which is what the keypoint ranking then sees.
    double se = 0., ax = 0., ay = 0.;
#pragma unroll
    for (int c = 0; c < KK; ++c) {
        const float e = expf(v[c] - m);
        se += e;
        const int ky = c / K, kx = c - ky * K;
        const int yy = y + ky - R, xx = x + kx - R;
        float ux = 0.f, uy = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const f32x2 f = *reinterpret_cast<const f32x2*>(flow + ((size_t)(n * H + yy) * W + xx) * fcs + fco);
            ux = f[0];
            uy = f[1];
        }
        ax += (double)wx[c] * ((double)e * ux);
        ay += (double)wy[c] * ((double)e * uy);
    }
    *reinterpret_cast<f32x2*>(dst + pix * dcs + dco) = f32x2{(float)((ax + bx) / se), (float)((ay + by) / se)};
}

// (Round 6: the distance vectors of a workgroup's 256 consecutive pixels staged through LDS -- one contiguous, fully coalesced
// run instead of 16-byte loads at a 208-byte lane stride -- measured SLOWER: 49.1 vs 45.9 us on the level-2 map, 11.4 vs 10.1
// on level 3, profiles/r6i_mirrors_kernel_stats_lds{0,1}.csv.  The kernel is not bound by its load pattern; removed.)
int launch_reg_head(const float* dist, int dist_cs, int k, const float* flow, int fcs, int fco, const float* wx,
                    float bx, const float* wy, float by, int N, int H, int W, float* dst, int dcs, int dco,
                    hipStream_t s) {
    static const bool vec = !(getenv("DFVO_REG_HEAD_V") && atoi(getenv("DFVO_REG_HEAD_V")) == 0);
    const dim3 grid(grid1d((long long)N * H * W, 256));
    const bool aligned = dist_cs % 4 == 0 && ((uintptr_t)dist & 15) == 0 && dist_cs >= ((k * k + 3) & ~3) && fcs % 2 == 0 &&
                         fco % 2 == 0 && ((uintptr_t)flow & 7) == 0 && dcs % 2 == 0 && dco % 2 == 0 && ((uintptr_t)dst & 7) == 0;
    if (vec && aligned && (k == 3 || k == 5 || k == 7)) {
        if (k == 3)
            hipLaunchKernelGGL(k_reg_head_v<3>, grid, dim3(256), 0, s, dist, dist_cs, flow, fcs, fco, wx, bx, wy, by, N, H, W, dst, dcs, dco);
        else if (k == 5)
            hipLaunchKernelGGL(k_reg_head_v<5>, grid, dim3(256), 0, s, dist, dist_cs, flow, fcs, fco, wx, bx, wy, by, N, H, W, dst, dcs, dco);
        else
            hipLaunchKernelGGL(k_reg_head_v<7>, grid, dim3(256), 0, s, dist, dist_cs, flow, fcs, fco, wx, bx, wy, by, N, H, W, dst, dcs, dco);
        DFVO_HIP_CHECK(hipGetLastError());
        return DFVO_OK;
    }
    hipLaunchKernelGGL(k_reg_head, grid, dim3(256), 0, s, dist, dist_cs, k, flow,
                       fcs, fco, wx, bx, wy, by, N, H, W, dst, dcs, dco);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ---------------------------------------------------------------------------------------------
// max pool 3x3 stride 2 pad 1
// ---------------------------------------------------------------------------------------------
__global__ void k_maxpool3x3s2(const float* __restrict__ src, int N, int H, int W, int C4, int Ho, int Wo,
                               float* __restrict__ dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * Ho * Wo * C4;
    if (idx >= total) return;
    const int c = (int)(idx % C4);
    long long pix = idx / C4;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            if (ix < 0 || ix >= W) continue;
            const f32x4 v = reinterpret_cast<const f32x4*>(src)[((size_t)(n * H + iy) * W + ix) * C4 + c];
            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
    }
    reinterpret_cast<f32x4*>(dst)[idx] = m;
}

int launch_maxpool3x3s2(const float* src, int N, int H, int W, int C, float* dst, hipStream_t s) {
    DFVO_ARG_CHECK(C % 4 == 0, "maxpool: C % 4");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_maxpool3x3s2, dim3(grid1d(total, 256)), dim3(256), 0, s, src, N, H, W, C / 4, Ho, Wo, dst);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ---------------------------------------------------------------------------------------------
// flow post-processing: scale, resize to the image size, forward-backward consistency
// ---------------------------------------------------------------------------------------------
__global__ void k_flow_resize(const float* __restrict__ netflow, int fcs, int fco, int h, int w, float scale, int H,
                              int W, float sh, float sw, float rh, float rw, float* __restrict__ fwd,
                              float* __restrict__ bwd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * H * W) return;
    const int n = idx / (H * W);
    const int pix = idx - n * H * W;
    const int oy = pix / W, ox = pix - oy * W;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(sh, oy, h, true, y0, y1, ly0, ly1);
    src_index(sw, ox, w, true, x0, x1, lx0, lx1);
    const float* f = netflow + (size_t)n * h * w * fcs + fco;
    float out[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float v00 = f[((size_t)y0 * w + x0) * fcs + c] * scale;
        const float v01 = f[((size_t)y0 * w + x1) * fcs + c] * scale;
        const float v10 = f[((size_t)y1 * w + x0) * fcs + c] * scale;
        const float v11 = f[((size_t)y1 * w + x1) * fcs + c] * scale;
        out[c] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    }
    float* o = n == 0 ? fwd : bwd;
    o[pix] = out[0] * rw;
    o[H * W + pix] = out[1] * rh;
}

__global__ void k_flow_consistency(const float* __restrict__ fwd, const float* __restrict__ bwd, int H, int W,
                                   float* __restrict__ diff) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= H * W) return;
    const int y = pix / W, x = pix - y * W;
    const float fx = fwd[pix], fy = fwd[H * W + pix];
    // layers.py:213-229: (pix + flow) / (size-1), then (p - 0.5) * 2
    float px = (float)x + fx, py = (float)y + fy;
    px = px / (float)(W - 1);
    py = py / (float)(H - 1);
    px = (px - 0.5f) * 2.f;
    py = (py - 0.5f) * 2.f;
    const Bilin b = bilin_setup(px, py, W, H);
    float wv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float* s = bwd + (size_t)c * H * W;
        const float vnw = (b.vy0 && b.vx0) ? -s[b.y0 * W + b.x0] : 0.f;
        const float vne = (b.vy0 && b.vx1) ? -s[b.y0 * W + b.x0 + 1] : 0.f;
        const float vsw = (b.vy1 && b.vx0) ? -s[(b.y0 + 1) * W + b.x0] : 0.f;
        const float vse = (b.vy1 && b.vx1) ? -s[(b.y0 + 1) * W + b.x0 + 1] : 0.f;
        wv[c] = vnw * b.nw + vne * b.ne + vsw * b.sw + vse * b.se;
    }
    const float dx = fx - wv[0], dy = fy - wv[1];
    diff[pix] = sqrtf(dx * dx + dy * dy);
}

int launch_flow_post(const float* netflow, int fcs, int fco, int h, int w, float scale, int H, int W, float* fwd,
                     float* bwd, float* diff, hipStream_t s) {
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float rh = (float)((double)H / (double)h), rw = (float)((double)W / (double)w);
    hipLaunchKernelGGL(k_flow_resize, dim3(grid1d(2LL * H * W, 256)), dim3(256), 0, s, netflow, fcs, fco, h, w, scale,
                       H, W, sh, sw, rh, rw, fwd, bwd);
    DFVO_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_flow_consistency, dim3(grid1d((long long)H * W, 256)), dim3(256), 0, s, fwd, bwd, H, W, diff);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// several contiguous device-to-device copies in one launch (the flow net's pyramid carry-over: eleven copy nodes in a graph
// cost eleven dispatches of ~5 us each on the net's stream; this is one)
constexpr int COPY_SEG_F4_PER_BLOCK = 256 * 4;
__global__ __launch_bounds__(256) void k_copy_segments(const CopySegs S) {
    int seg = 0;
#pragma unroll 1
    while (seg + 1 < S.n && blockIdx.x >= S.start[seg + 1]) ++seg;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(S.src[seg]);
    float4* __restrict__ dst = reinterpret_cast<float4*>(S.dst[seg]);
    const unsigned base = (blockIdx.x - S.start[seg]) * COPY_SEG_F4_PER_BLOCK + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned i = base + u * 256;
        if (i < S.n4[seg]) dst[i] = src[i];
    }
}

int launch_copy_segments(const float* const* src, float* const* dst, const size_t* floats, int n, hipStream_t s) {
    DFVO_ARG_CHECK(n >= 1 && n <= 12, "launch_copy_segments: 1 .. 12 segments");
    CopySegs S = {};
    S.n = n;
    unsigned blocks = 0;
    for (int i = 0; i < n; ++i) {
        DFVO_ARG_CHECK(floats[i] % 4 == 0 && ((uintptr_t)src[i] & 15) == 0 && ((uintptr_t)dst[i] & 15) == 0,
                       "launch_copy_segments: segments must be 16-byte aligned multiples of 4 floats");
        S.src[i] = src[i];
        S.dst[i] = dst[i];
        S.n4[i] = (unsigned)(floats[i] / 4);
        S.start[i] = blocks;
        blocks += (S.n4[i] + COPY_SEG_F4_PER_BLOCK - 1) / COPY_SEG_F4_PER_BLOCK;
    }
    S.start[n] = blocks;
    if (blocks == 0) return DFVO_OK;
    hipLaunchKernelGGL(k_copy_segments, dim3(blocks), dim3(256), 0, s, S);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ---------------------------------------------------------------------------------------------
// depth tail
// ---------------------------------------------------------------------------------------------
__global__ void k_disp_to_depth(const float* __restrict__ disp, int dcs, int dco, int n, float min_disp,
                                float disp_range, float mult, float* __restrict__ depth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float scaled = min_disp + disp_range * disp[(size_t)i * dcs + dco];
    depth[i] = (1.f / scaled) * mult;
}

int launch_disp_to_depth(const float* disp, int dcs, int dco, int n, float min_disp, float disp_range, float mult,
                         float* depth, hipStream_t s) {
    hipLaunchKernelGGL(k_disp_to_depth, dim3(grid1d(n, 256)), dim3(256), 0, s, disp, dcs, dco, n, min_disp,
                       disp_range, mult, depth);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

__global__ void k_depth_post(const float* __restrict__ depth, int h, int w, int H, int W, double ify, double ifx,
                             int y0, int y1, int x0, int x1, float min_depth, float max_depth,
                             float* __restrict__ raw, double* __restrict__ proc) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= H * W) return;
    const int y = pix / W, x = pix - y * W;
    // cv::resize INTER_NEAREST: sx = min(cvFloor(x * ifx), w - 1)
    int sy = (int)floor(y * ify), sx = (int)floor(x * ifx);
    sy = sy < h - 1 ? sy : h - 1;
    sx = sx < w - 1 ? sx : w - 1;
    const float d = depth[sy * w + sx];
    raw[pix] = d;
    const bool in_crop = (y >= y0 && y < y1 && x >= x0 && x < x1);
    const bool in_range = (d < max_depth) && (d > min_depth);
    proc[pix] = (in_crop && in_range) ? (double)d : 0.0;
}

int launch_depth_post(const float* depth, int h, int w, int H, int W, int y0, int y1, int x0, int x1,
                      float min_depth, float max_depth, float* raw, double* proc, hipStream_t s) {
    const double ify = 1.0 / ((double)H / (double)h), ifx = 1.0 / ((double)W / (double)w);
    hipLaunchKernelGGL(k_depth_post, dim3(grid1d((long long)H * W, 256)), dim3(256), 0, s, depth, h, w, H, W, ify,
                       ifx, y0, y1, x0, x1, min_depth, max_depth, raw, proc);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

}  // namespace dfvo
