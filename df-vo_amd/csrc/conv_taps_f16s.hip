// Multi-tap streaming layers of the f16x3 mode on the large maps: LiteFlowNet's first 7x7 layer on the 3-channel frame, the
// separable 7x1 / 1x7 distance layers of levels 2-4, the stride-2 3x3 layers of Features, the 5x5 distance layer
// (/root/reference/libs/deep_models/flow/lite_flow_net/lite_flow_net.py:39-75,226-240).  Small cin, tens of taps, hundreds
// of thousands of pixels.
//
// On the generic register-ring kernel (conv_gemm_f16s.h) these layers are bound by the VALU, not by memory or the matrix
// pipe: that kernel gathers a pixel's k-groups straight from the activation tensor and splits them into f16 hi / lo planes
// in registers -- once per TAP that touches the element, i.e. 49 / 7 / 9 times per input element here (~400 VALU
// instructions per 39 MFMAs on the 7x7 layer; profiles/r4f_conv_layers_per_launch.csv: 64 us for 62 MB, 0.9-1.7 TB/s).
// Here the input window of a tile is loaded and split ONCE into LDS planes (the LDS-window kernels' idea, for any tap
// pattern), and the B fragments of every step are two 8-byte LDS reads per plane.  The whole (small) K of the layer is
// resident: one load phase, one barrier, then the layer's K steps.
// Arithmetic: the generic kernel's, unchanged -- same packed weights and k-group table (g = tap G + channel group, four
// k-groups per MFMA step), same operand positions in every MFMA, same products (hi x lo, lo x hi -> cross sums, hi x hi ->
// main sums), same step order: outputs are bit-identical to conv_gemm_f16s_kernel<4, 1, TC>.
#include "dfvo_common.h"


namespace dfvo {

#include "conv_epi.h"
#include "conv_f16_split.h"

typedef unsigned int u32x4t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(4))) const u32x4t cu32x4t;

// TH output rows x 32 columns per workgroup, one wave per row; a wave contracts all TC cout blocks (32 couts each) of the layer.
// (Measured alternatives, profiles/r4l_taps_ab.txt: two rows per wave with one cout block each -- half the weight
// fetches per MFMA -- and a six-deep weight ring were both SLOWER per layer, 46 -> 58-62 us on the 7x7 layer.)
// NP: products per term (3 = f16x3; 1 = "f16" mode: hi planes only)
template <int TH, int TC, int NP = 3, int PF = 3>
__global__ __launch_bounds__(64 * TH) void conv_taps_f16s_kernel(const ConvParams p, int G, int pxd, int wcols, int wrows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char taps_lds[];  // [wrows * wcols pixels][pxd bytes] | 16 zero bytes | [S][4] offsets
    constexpr int NT = 64 * TH;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lp = lane & 31, kb = lane >> 5;
    const int tiles_x = (p.Wo + 31) / 32, tiles_y = (p.Ho + TH - 1) / TH;
    const int nb = gridDim.x;
    int bid = blockIdx.x;
    {  // XCD-aware order: each XCD walks a contiguous run of tiles
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int n = bid / (tiles_y * tiles_x);
    const int trem = bid - n * (tiles_y * tiles_x);
    const int oy0 = (trem / tiles_x) * TH, ox0 = (trem % tiles_x) * 32;
    const int iy0 = oy0 * p.stride - p.pad_h, ix0 = ox0 * p.stride - p.pad_w;

    // ---- the window, split once: pixel = [hi plane: G x 4 halves][lo plane: G x 4 halves] (+ 8 bytes of padding)
    float amax = 0.f;
    const int npix = wrows * wcols;
    const int zoff = npix * pxd;  // 16 zero bytes: the k-groups beyond the layer's K (last step's padding)
    if (t < 4) reinterpret_cast<unsigned*>(taps_lds + zoff)[t] = 0u;
    {   // items (pixel, channel group) t, t + NT, ...: the indices advance by constant steps with carries -- no division in
        // the loop (the divisors are run-time values: two divisions per item cost more than the split itself)
        const int dpx = NT / G, dcg = NT - dpx * G;              // item step NT = dpx pixels + dcg groups
        const int dwy = dpx / wcols, dwx = dpx - dwy * wcols;    // pixel step dpx = dwy rows + dwx columns
        int px = t / G, cg = t - px * G;
        int wy = px / wcols, wx = px - wy * wcols;
        // Items in batches of LB: the batch's addresses first, then its loads back to back, then the splits.  (Round 6: one load
        // per loop iteration left every load's HBM latency exposed -- ten dependent round trips on the 7 x 1 distance layer of
        // level 2, whose launch took 64 us for 72 MB: 1.1 TB/s with the matrix pipe 85 % idle.)
        constexpr int LB = 4;
        const int nitems = npix * G;
        for (int it = t; it < nitems; it += NT * LB) {
            f32x4 xb[LB];
            int pxb[LB], cgb[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const bool live = it + u * NT < nitems;
                const int iy = iy0 + wy, ix = ix0 + wx;
                const bool v = live && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                pxb[u] = live ? px : -1;
                cgb[u] = cg;
                xb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (v) xb[u] = *reinterpret_cast<const f32x4*>(p.src0 + ((size_t)(n * p.H + iy) * p.W + ix) * p.cs0 + p.co0 + cg * 4);
                cg += dcg;
                int cpx = dpx;
                if (cg >= G) {
                    cg -= G;
                    cpx += 1;
                }
                px += cpx;
                wx += dwx + (cpx - dpx);
                wy += dwy;
                if (wx >= wcols) {
                    wx -= wcols;
                    wy += 1;
                }
                if (wx >= wcols) {  // (dwx + 1 can exceed wcols once more only when dwx = wcols - 1 and both carries hit)
                    wx -= wcols;
                    wy += 1;
                }
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                if (pxb[u] < 0) continue;
                h16x4 hi, lo;
                unsigned char* d = taps_lds + (size_t)pxb[u] * pxd + cgb[u] * 8;
                if constexpr (NP == 3) {
                    split_f16_planes(xb[u], &hi, &lo, amax);
                    *reinterpret_cast<h16x4*>(d) = hi;
                    *reinterpret_cast<h16x4*>(d + G * 8) = lo;
                } else {
                    split_f16_hi(xb[u], &hi, amax);
                    *reinterpret_cast<h16x4*>(d) = hi;
                }
            }
        }
    }
    // the byte offset of every k-group of the layer relative to a lane's pixel (-1: a group beyond the layer's K), once per
    // workgroup into LDS: the K loop then costs one 8-byte LDS read per step instead of ~70 scalar + vector instructions
    // of table decoding (counters, profiles/r4n_pmc_taps_layers_*.txt: the loop was instruction-issue bound, ~120
    // instructions per 3-MFMA step)
    const int S = p.f16g_steps;
    const int rowb = wcols * pxd, lo_off = G * 8;
    int* const otab = reinterpret_cast<int*>(taps_lds + zoff + 16);  // [S][4]
    for (int i = t; i < S * 4; i += NT) {
        const unsigned e = p.f16g_tab[i];
        const int ky = e & 31, kx = (e >> 5) & 31, cg = (int)(e >> 16) >> 2;
        otab[i] = ((e >> 10) & 1u) ? ky * rowb + kx * pxd + cg * 8 : -1;
    }
    __syncthreads();

    // ---- K steps: the generic kernel's loop with the B fragments from LDS
    const unsigned short* const wbase = p.wf16g + (size_t)(kb * 32 + lp) * 8;
    const size_t w_step_stride = (size_t)p.wf16g_cout_pad * 32;  // halves per step
    const int lane_base = (wave * p.stride * wcols + lp * p.stride) * pxd;  // this lane's output pixel in the window

    h16x8 rw[PF][TC][2];
    u32x2t rx[PF][2][2];  // [stage][k-group j of the lane][plane]
    int nl = 0;
    u32x2t oo = *reinterpret_cast<const u32x2t*>(otab + 2 * kb);  // this lane's two k-group offsets of the next step to load
    auto load_step = [&](int st) {
        const int o0 = (int)oo[0], o1 = (int)oo[1];
        oo = *reinterpret_cast<const u32x2t*>(otab + (nl + 1 < S ? nl + 1 : S - 1) * 4 + 2 * kb);  // one step ahead of its use
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = j ? o1 : o0;
            const int a_hi = o < 0 ? zoff : lane_base + o;
            const int a_lo = o < 0 ? zoff + 8 : lane_base + o + lo_off;
            rx[st][j][0] = *reinterpret_cast<const u32x2t*>(taps_lds + a_hi);
            if constexpr (NP == 3) rx[st][j][1] = *reinterpret_cast<const u32x2t*>(taps_lds + a_lo);
        }
        const unsigned short* g = wbase + (size_t)nl * w_step_stride;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            rw[st][i][0] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 1024);
            if constexpr (NP == 3) rw[st][i][1] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 1024 + 512);
        }
        ++nl;
    };
    f32x16 am[TC], ax[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) am[i][e] = ax[i][e] = 0.f;
#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (d < S) load_step(d);
    for (int s = 0; s < S; s += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (s + u < S) {
                const u32x4t hq = {rx[u][0][0][0], rx[u][0][0][1], rx[u][1][0][0], rx[u][1][0][1]};
                u32x4t lq = {0u, 0u, 0u, 0u};
                if constexpr (NP == 3) lq = u32x4t{rx[u][0][1][0], rx[u][0][1][1], rx[u][1][1][0], rx[u][1][1][1]};
                const h16x8 xh = __builtin_bit_cast(h16x8, hq), xl = __builtin_bit_cast(h16x8, lq);
                h16x8 wh[TC], wl[TC];
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    wh[i] = rw[u][i][0];
                    if constexpr (NP == 3) wl[i] = rw[u][i][1];
                }
                if (s + u + PF < S) load_step(u);
                if constexpr (NP == 3) {
#pragma unroll
                    for (int i = 0; i < TC; ++i) ax[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], xl, ax[i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TC; ++i) ax[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[i], xh, ax[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < TC; ++i) am[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], xh, am[i], 0, 0, 0);
            }
        }
    }
    if (amax > F16S_MAX) atomicAdd(p.f16s_clamp_ctr, 1u);

    // ---- epilogue: every quad's bias first, then only stores (conv_gemm_f16s.h, RAG)
    const int oy = oy0 + wave, ox = ox0 + lp;
    if (oy >= p.Ho || ox >= p.Wo) return;
    float* const drow = p.dst + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.dst_cs + p.dst_co;
    const float slope = p.act == ACT_LEAKY ? p.act_param : 1.f;
    f32x4 bq[4 * TC];
#pragma unroll
    for (int q = 0; q < 4 * TC; ++q) {
        const int col0 = (q >> 2) * 32 + 8 * (q & 3) + 4 * kb;
        bq[q] = *reinterpret_cast<const f32x4*>(p.bias + (col0 + 3 < p.cout_pad ? col0 : p.cout_pad - 4));
    }
#pragma unroll
    for (int q = 0; q < 4 * TC; ++q) {
        const int col0 = (q >> 2) * 32 + 8 * (q & 3) + 4 * kb;
        f32x4 x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = am[q >> 2][4 * (q & 3) + e] + F16S_LO_UNSCALE * ax[q >> 2][4 * (q & 3) + e] + bq[q][e];
            x[e] = v > 0.f ? v : (p.act == ACT_RELU ? 0.f : v * slope);
        }
        if (col0 + 3 < p.cout) {
            *reinterpret_cast<f32x4*>(drow + col0) = x;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (col0 + e < p.cout)
                    drow[col0 + e] = x[e];
                else if (col0 + e < p.dst_zero_to)
                    drow[col0 + e] = 0.f;
            }
        }
    }
}

namespace {
struct TapsGeom {
    int G, pxd, th, wrows, wcols;
    size_t lds;
};
// four output rows per workgroup (one per wave) while the window fits, else two
bool taps_geom(const ConvParams& p, TapsGeom* g) {
    g->G = p.G0;
    g->pxd = p.G0 * 16 + 8;  // both planes + 8 bytes: consecutive pixels start on different LDS banks
    for (int th = 4; th >= 2; th -= 2) {
        g->th = th;
        g->wrows = (th - 1) * p.stride + p.kh;
        g->wcols = 31 * p.stride + p.kw;
        g->lds = (size_t)g->wrows * g->wcols * g->pxd + 16 + (size_t)p.f16g_steps * 16;
        if (g->lds <= 48 * 1024) return true;
    }
    return false;
}
}  // namespace

bool conv_taps_f16s_ok(const ConvParams& p) {
    if (!p.wf16g || !p.f16g_tab || p.G1 != 0 || p.up0 != 0 || p.pad_mode != PAD_ZERO) return false;
    // stride 1 only: the stride-2 layers were measured slower here than on the generic kernel (3x3 / s2 32 -> 32: 55-61 vs 39 us, before and after the offset table --
    // a window of 2 x the pixels for a quarter of the taps per element)
    if (p.kh * p.kw < 3 || p.stride != 1 || p.kh > 31 || p.kw > 31) return false;
    if (p.wf16g_cout_pad != 32 && p.wf16g_cout_pad != 64) return false;
    if ((p.cs0 | p.co0) & 3) return false;
    // the epilogue here: 16-byte stores, bias + none / leaky / relu, no residual
    if (p.res || ((p.dst_cs | p.dst_co) & 3) || p.cout_pad < 4 || (p.act != ACT_NONE && p.act != ACT_LEAKY && p.act != ACT_RELU)) return false;
    TapsGeom g;
    return taps_geom(p, &g);
}

int launch_taps_f16s(const ConvParams& p, hipStream_t stream, int* grid_x) {
    DFVO_ARG_CHECK(p.f16s_clamp_ctr, "conv_taps_f16s: clamp counter not set");
    TapsGeom g;
    DFVO_ARG_CHECK(taps_geom(p, &g), "conv_taps_f16s: window does not fit");
    const int tiles = p.N * ((p.Ho + g.th - 1) / g.th) * ((p.Wo + 31) / 32);
    const bool tc2 = p.wf16g_cout_pad == 64;
#define DFVO_TAPS_LAUNCH(TH_, TC_)                                                                                          \
    do {                                                                                                                   \
        if (p.f16_terms == 1)                                                                                              \
            hipLaunchKernelGGL((conv_taps_f16s_kernel<TH_, TC_, 1>), dim3(tiles), dim3(64 * TH_), g.lds, stream, p, g.G,    \
                               g.pxd, g.wcols, g.wrows);                                                                   \
        else                                                                                                               \
            hipLaunchKernelGGL((conv_taps_f16s_kernel<TH_, TC_, 3>), dim3(tiles), dim3(64 * TH_), g.lds, stream, p, g.G,    \
                               g.pxd, g.wcols, g.wrows);                                                                   \
    } while (0)
    if (g.th == 4) {
        if (tc2)
            DFVO_TAPS_LAUNCH(4, 2);
        else
            DFVO_TAPS_LAUNCH(4, 1);
    } else {
        if (tc2)
            DFVO_TAPS_LAUNCH(2, 2);
        else
            DFVO_TAPS_LAUNCH(2, 1);
    }
#undef DFVO_TAPS_LAUNCH
    DFVO_HIP_CHECK(hipGetLastError());
    if (grid_x) *grid_x = tiles;
    return DFVO_OK;
}

}  // namespace dfvo
