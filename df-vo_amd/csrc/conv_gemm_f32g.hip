// Exact-fp32 twin of conv_gemm_f16s_kernel (conv_gemm_f16s.h): the generic implicit-GEMM convolution for everything the
// LDS-window kernels do not take in DFVO_CONV_PRECISION=fp32 mode -- the small maps (pyramid levels 5 / 6, the depth net's
// inner layers), where a launch is a latency chain, and the streaming layers (1x1, k x 1, stride 2, 7x7).
// Round 3 built this skeleton for the f16x3 mode (3.1 -> 1.6 ms per pair on those 76 launches); the exact-fp32 mode -- the
// reference's own precision -- kept running them on conv_igemm_f32_kernel with its cross-XCD split-K hand-over
// (3.5 ms per pair, profiles/r4d_fp32_by_config.json).  Same decomposition here:
//   * a wave owns a 32-pixel x (32 TC)-cout block and a slice of K; the KSP waves of a workgroup that share a block add
//     their partial blocks through LDS in slice order (deterministic, no second launch, no cross-XCD hand-off);
//   * no LDS staging: a lane (pixel = lane & 31, kb = lane >> 5) needs k-groups 2 kb and 2 kb + 1 of its pixel -- two
//     16-byte loads straight from the NHWC activation tensor; the weights of a step come as two 16-byte loads per cout
//     block from the packed [step][cout / 32][kb][32 couts][8 k] layout; both PF steps ahead in a register ring;
//   * the contraction is v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: exact products, 157 TFLOP/s class): MFMA r of a
//     step consumes component r of the lane's eight activation and eight weight values, so the eight MFMAs cover the 16
//     k-values of the step (the k permutation is the same on both sides).
// Reference layers: /root/reference/libs/deep_models/flow/lite_flow_net/lite_flow_net.py:39-75,98-129,164-240,
// depth/monodepth2/resnet_encoder.py:87-98, depth_decoder.py:50-65.
#include "dfvo_common.h"

#include <cstring>
#include <vector>

namespace dfvo {

#include "conv_epi.h"

typedef float f32x16g __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const u32x4g cu32x4g;

template <int WP, int KSP, int TC, int PF = 3>
__global__ __launch_bounds__(64 * WP * KSP) void conv_gemm_f32g_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float f32g_red[];  // [WP][KSP][TC][16][64] partial blocks (KSP > 1)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wp = wave / KSP, wk = wave % KSP;
    const int lp = lane & 31, kb = lane >> 5;
    const int M = p.N * p.Ho * p.Wo;
    // one-dimensional grid, XCD-aware: each XCD walks a contiguous run of logical ids, the cout tile fastest inside a run
    // (the workgroups that read the same pixel block are neighbours on one XCD's L2)
    const int nb = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int ny = p.wf16g_cout_pad / (32 * TC);
    const int ytile = bid % ny;
    bid /= ny;
    const int m = (bid * WP + wp) * 32 + lp;
    const bool vm = m < M;
    int iy0, ix0, nimg;
    {
        const int mm = vm ? m : 0;
        nimg = mm / (p.Ho * p.Wo);
        const int rem = mm - nimg * (p.Ho * p.Wo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        iy0 = oy * p.stride - p.pad_h;
        ix0 = ox * p.stride - p.pad_w;
    }
    const int n0 = ytile * (32 * TC);
    const int S = p.f16g_steps;
    const int per = (S + KSP - 1) / KSP;
    const int s0 = wk * per < S ? wk * per : S;
    const int s1 = s0 + per < S ? s0 + per : S;

    // everything the gather needs from the parameter block lives in SGPRs for the whole kernel
    auto pin = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto pin64 = [](unsigned long long v) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    const cu32x4g* const tab = (const cu32x4g*)pin64((unsigned long long)p.f16g_tab);
    const unsigned long long src0 = pin64((unsigned long long)p.src0), src1 = pin64((unsigned long long)(p.G1 ? p.src1 : p.src0));
    const int cs0 = pin(p.cs0), cs1 = pin(p.cs1), co0 = pin(p.co0), co1 = pin(p.co1), up0 = pin(p.up0);
    const int H = pin(p.H), W = pin(p.W), refl = pin(p.pad_mode == PAD_REFLECT ? 1 : 0);
    const int H0 = H >> up0, W0 = W >> up0;
    // weights: [step][cout_pad / 32][kb][32 couts][8 k] floats
    const float* const wbase = p.wf32g + ((size_t)n0 * 16 + (kb * 32 + lp) * 8);
    const size_t w_step_stride = (size_t)pin(p.wf16g_cout_pad) * 16;  // floats per step

    f32x4 ra[PF][2];
    unsigned rav = 0;  // bit (2 stage + j): the load holds real data
    f32x4 rw[PF][TC][2];
    int nl = s0;
    u32x4g tq = tab[nl < S ? nl : S - 1];
    auto load_step = [&](int st) {
        const unsigned e0 = __builtin_amdgcn_readfirstlane(tq[0]), e1 = __builtin_amdgcn_readfirstlane(tq[1]),
                       e2 = __builtin_amdgcn_readfirstlane(tq[2]), e3 = __builtin_amdgcn_readfirstlane(tq[3]);
        const int sN = nl + 1 < S ? nl + 1 : S - 1;
        tq = tab[sN];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned e = kb ? (j ? e3 : e2) : (j ? e1 : e0);
            const int ky = e & 31, kx = (e >> 5) & 31;
            const bool s1v = ((e >> 11) & 1u) != 0;
            int iy = iy0 + ky, ix = ix0 + kx;
            bool v = vm && ((e >> 10) & 1u);
            if (refl) {
                iy = reflect_idx(iy, H);
                ix = reflect_idx(ix, W);
            }
            v = v && iy >= 0 && iy < H && ix >= 0 && ix < W;
            iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
            ix = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            const int pix0 = (nimg * H0 + (iy >> up0)) * W0 + (ix >> up0);
            const int pix1 = (nimg * H + iy) * W + ix;
            const int off = (s1v ? pix1 * cs1 + co1 : pix0 * cs0 + co0) + (int)(e >> 16);
            const unsigned long long base = s1v ? src1 : src0;
            ra[st][j] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(base + (long long)off * 4);
            rav = (rav & ~(1u << (2 * st + j))) | ((v ? 1u : 0u) << (2 * st + j));
        }
        const float* g = wbase + (size_t)nl * w_step_stride;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            rw[st][i][0] = *reinterpret_cast<const f32x4*>(g + (size_t)i * 512);
            rw[st][i][1] = *reinterpret_cast<const f32x4*>(g + (size_t)i * 512 + 4);
        }
        ++nl;
    };

    f32x16g am[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) am[i][e] = 0.f;

#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (s0 + d < s1) load_step(d);
    for (int s = s0; s < s1; s += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (s + u < s1) {
                const f32x4 x0 = ((rav >> (2 * u)) & 1u) ? ra[u][0] : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 x1 = ((rav >> (2 * u + 1)) & 1u) ? ra[u][1] : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 w0[TC], w1[TC];
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    w0[i] = rw[u][i][0];
                    w1[i] = rw[u][i][1];
                }
                if (s + u + PF < s1) load_step(u);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < TC; ++i) am[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[i][r], x0[r], am[i], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < TC; ++i) am[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[i][r], x1[r], am[i], 0, 0, 0);
            }
        }
    }

    if (KSP == 1) {
        ConvEpi<4 * TC> epi;
        conv_epi_init(p, epi, [&](int q) { return n0 + (q >> 2) * 32 + 8 * (q & 3) + 4 * kb; });
        conv_epi_row(p, epi, (size_t)(vm ? m : 0), vm, [&](int q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = am[q >> 2][4 * (q & 3) + e];
            return v;
        });
        return;
    }
    float* const mine = f32g_red + (size_t)((wp * KSP + wk) * TC) * 16 * 64;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[(i * 16 + e) * 64 + lane] = am[i][e];
    __syncthreads();
    const bool vec_ok = conv_vec_ok(p);
    for (int q = wk; q < 4 * TC; q += KSP) {
        const int i = q >> 2, g = q & 3;
        const int col0 = n0 + i * 32 + 8 * g + 4 * kb;
        const bool fastq = vec_ok && col0 + 3 < p.cout;
        f32x4 b = {0.f, 0.f, 0.f, 0.f}, r = {0.f, 0.f, 0.f, 0.f};
        if (fastq) {  // (the quad's bias / residual requested before the LDS sum, consumed after it)
            b = *reinterpret_cast<const f32x4*>(p.bias + col0);
            if (p.res && vm) r = *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.res_cs + p.res_co + col0);
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < KSP; ++z) {  // slice order: the sum does not depend on which wave finishes the quad
            const float* src = f32g_red + (size_t)((wp * KSP + z) * TC + i) * 16 * 64 + (4 * g) * 64 + lane;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = z == 0 ? src[e * 64] : v[e] + src[e * 64];
        }
        if (!vm) continue;
        if (fastq) {
            f32x4 x = v + b + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = apply_act(x[e], p.act, p.act_param);
            *reinterpret_cast<f32x4*>(p.dst + (size_t)m * p.dst_cs + p.dst_co + col0) = x;
        } else {
            conv_epilogue_quad(p, (size_t)m, col0, v, vec_ok);
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
// fp32 weights in k-group order: [step][cout_pad / 32][k-block (2)][32 couts][8 k]; the 16 k of a step are k-groups
// 4 step .. 4 step + 3 (g = tap (G0 + G1) + channel group, 4 channels each): the order of conv_build_f16g_table, which this
// kernel shares with the f16x3 one.  Returns the number of floats (out may be null)
size_t conv_pack_weights_f32g(const float* w, int cout, int c0, int c1, int kh, int kw, const float* fold_scale, float* out) {
    const int G0 = cdiv(c0, 4), G1 = cdiv(c1, 4), G = G0 + G1, taps = kh * kw;
    const int steps = cdiv(taps * G, 4);
    const int cp = round_up(cout, 32);
    const size_t total = (size_t)steps * cp * 16;
    if (!out) return total;
    memset(out, 0, total * sizeof(float));
    const int cin = c0 + c1;
    for (int tap = 0; tap < taps; ++tap)
        for (int cg = 0; cg < G; ++cg) {
            const int g = tap * G + cg, step = g >> 2, gl = g & 3;
            for (int q = 0; q < 4; ++q) {
                int ci;
                if (cg < G0) {
                    ci = cg * 4 + q;
                    if (ci >= c0) continue;
                } else {
                    ci = (cg - G0) * 4 + q;
                    if (ci >= c1) continue;
                    ci += c0;
                }
                const int k = gl * 4 + q;
                for (int co = 0; co < cout; ++co) {
                    float v = w[((size_t)co * cin + ci) * taps + tap];
                    if (fold_scale) v *= fold_scale[co];
                    out[((size_t)step * cp + (co & ~31)) * 16 + ((k >> 3) * 32 + (co & 31)) * 8 + (k & 7)] = v;
                }
            }
        }
    return total;
}

bool conv_f32g_ok(const ConvParams& p) {
    return p.wf32g && p.f16g_tab && p.kh <= 31 && p.kw <= 31;
}

template <int WP, int KSP, int TC>
static int launch_f32g_cfg(const ConvParams& p, hipStream_t stream, int* grid_x) {
    const long long M = (long long)p.N * p.Ho * p.Wo;
    dim3 grid((unsigned)(((M + 32 * WP - 1) / (32 * WP)) * (p.wf16g_cout_pad / (32 * TC))), 1, 1);
    const size_t lds = KSP > 1 ? (size_t)WP * KSP * TC * 16 * 64 * sizeof(float) : 0;
    if (lds > 48 * 1024)
        if (int rc_lds = ensure_dyn_lds((const void*)conv_gemm_f32g_kernel<WP, KSP, TC>, lds)) return rc_lds;
    hipLaunchKernelGGL((conv_gemm_f32g_kernel<WP, KSP, TC>), grid, dim3(64 * WP * KSP), lds, stream, p);
    DFVO_HIP_CHECK(hipGetLastError());
    if (grid_x) *grid_x = (int)grid.x;
    return DFVO_OK;
}

// Shape choice: the rules of launch_f16g (conv_gemm_f16s.h) -- two cout blocks per wave while that leaves enough tiles, K
// sliced over the waves of a workgroup until the chip holds ~2 waves per SIMD.  *ksp_out: the K slicing chosen (profile row)
int launch_f32g(const ConvParams& p, hipStream_t stream, int* ksp_out, int* grid_x) {
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const long long mblocks = (M + 31) / 32;
    const int nblk = p.wf16g_cout_pad / 32;
    const long long target = 2048;
    bool tc2 = (nblk % 2) == 0;
    if (tc2 && mblocks * (nblk / 2) * 8 * 2 < target && p.f16g_steps >= 64) tc2 = false;
    const long long tiles = mblocks * (tc2 ? nblk / 2 : nblk);
    int ksp = 1;
    while (ksp < (tc2 ? 8 : 16) && tiles * ksp * 2 <= target && p.f16g_steps >= 4 * ksp * 2) ksp *= 2;
    if (ksp_out) *ksp_out = ksp;
    if (tc2) {
        switch (ksp) {
            case 1: return launch_f32g_cfg<4, 1, 2>(p, stream, grid_x);
            case 2: return launch_f32g_cfg<2, 2, 2>(p, stream, grid_x);
            case 4: return launch_f32g_cfg<1, 4, 2>(p, stream, grid_x);
            default: return launch_f32g_cfg<1, 8, 2>(p, stream, grid_x);
        }
    }
    switch (ksp) {
        case 1: return launch_f32g_cfg<4, 1, 1>(p, stream, grid_x);
        case 2: return launch_f32g_cfg<2, 2, 1>(p, stream, grid_x);
        case 4: return launch_f32g_cfg<1, 4, 1>(p, stream, grid_x);
        case 8: return launch_f32g_cfg<1, 8, 1>(p, stream, grid_x);
        default: return launch_f32g_cfg<1, 16, 1>(p, stream, grid_x);
    }
}

}  // namespace dfvo
