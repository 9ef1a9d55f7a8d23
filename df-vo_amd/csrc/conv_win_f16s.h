// 3x3 / stride-1 convolution with fp32-class accuracy on the f16 matrix cores (opt-in: DFVO_CONV_PRECISION=f16x3).
// Included by conv_igemm_f32.hip (same translation unit: shares the epilogue and the per-launch profiling hooks).
//
// Every fp32 operand x is split into two f16 planes  x = hi + 2^-11 lo,  hi = f16(x),  lo = f16((x - hi) * 2^11):
// 22 mantissa bits, the scaled low plane never leaves f16's normal range (an unscaled residue would be subnormal for
// |x| < 0.125 and lose its bits).  A product keeps three terms,
//        a b  ~=  ah bh  +  2^-11 (ah bl + al bh)                       (dropped: 2^-22 al bl, an fp32-rounding-sized term)
// each an EXACT fp32 number (11 x 11 bit mantissas), accumulated in fp32 on v_mfma_f32_32x32x16_f16 in two accumulator
// sets ("main" and "cross") that the epilogue combines as main + 2^-11 cross.  Against the exact fp32-MFMA path this
// trades 16 K-passes of 64 FLOP/clk/SIMD for 3 passes of 1024 FLOP/clk/SIMD: 5.3x less matrix-pipe time.
//
// Tiling (wave64, 4 or 8 waves per workgroup): a workgroup owns TH rows x 32 columns of output pixels x BN output channels.
// The (TH+2) x 34 input window of one 16-channel chunk is split into planes ONCE, when it is written to LDS
// (padding / reflection / x2-upsample / two-source concat resolved at that load), pixel stride 80 bytes = 5 sixteen-byte
// slots: the 16 lanes of every ds_read_b128 service group hit 16 distinct slots.  One MFMA contracts a whole chunk for a
// 32-channel x 32-pixel block: the weight fragment (A operand: 32 couts x 16 channels) comes straight from L2 in its
// packed plane layout, fetched one tap ahead into registers; the pixel fragment (B operand: 16 channels x 32 pixels of
// one window row) is one ds_read_b128 per plane.  A wave holds TC x TR blocks (main + cross: 32 accumulator registers
// each), the only barrier is the one per chunk (9 taps x 3 TC TR MFMAs of 32 cycles between barriers).
// The accumulator layout (col = lane & 31 = pixel, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = cout) gives a lane
// four consecutive couts of one pixel per register quad: the fp32 kernels' 16-byte vector epilogue is reused as is.
#pragma once
// (included inside namespace dfvo)

#include "conv_f16_split.h"

// |x| > 65504 does not fit the hi plane.  It is neither clamped (a silently wrong product) nor ignored: the conversion
// yields +-inf, which propagates as inf / NaN into the layer's output, and every kernel that splits activations keeps the
// running max |x| of what it split (one v_max3_f32 per two elements -- cheaper than the clamp it replaces) and bumps
// g_f16s_clamped once per thread that saw such a value; dfvo_f16s_overflow_count() reads it (the -m gpu tests assert
// zero after every f16x3 test).
__device__ unsigned int g_f16s_clamped = 0;
__device__ __forceinline__ void f16s_report_clamp(float amax) {
    if (amax > F16S_MAX) atomicAdd(&g_f16s_clamped, 1u);
}
// the counter's device address, for the f16x3 kernels that live in other translation units (device symbols do not cross TUs)
static unsigned* f16s_clamp_counter() {
    static unsigned* ptr = nullptr;
    if (!ptr && hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(g_f16s_clamped)) != hipSuccess) ptr = nullptr;
    return ptr;
}
template <int WC, int WR, int TC, int TR, int NP = 3>  // NP: products per term (3 = f16x3, 1 = "f16" mode: hi planes only)
__global__ __launch_bounds__(64 * WC * WR, 2) void conv_win_f16s_kernel(const ConvParams p) {
    constexpr int NT = 64 * WC * WR;  // 4 or 8 waves per workgroup
    constexpr int TH = WR * TR, TW = 32, WH = TH + 2, WW = TW + 2, PS = 20;  // pixel stride in dwords (80 bytes)
    constexpr int BN = WC * TC * 32;
    constexpr int WIN = WH * WW * PS;  // dwords per window buffer
    constexpr int W_ITEMS = WH * WW * 4;
    constexpr int W_CNT = (W_ITEMS + NT - 1) / NT;
    static_assert(WC * WR == 4 || WC * WR == 8, "4 or 8 waves per block");
    __shared__ __attribute__((aligned(16))) float lds[2 * WIN];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wc = wave / WR, wr = wave % WR;
    const int lp = lane & 31, kb = lane >> 5;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int nb = gridDim.x;
    int bid = blockIdx.x;
    {  // XCD-aware order: each XCD walks a contiguous run of tiles (neighbours share halo rows in its L2)
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int n = bid / (tiles_y * tiles_x);
    const int trem = bid - n * (tiles_y * tiles_x);
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
    const int n0 = blockIdx.y * BN;
    const int nchunk0 = (p.G0 + 3) >> 2, nchunk1 = (p.G1 + 3) >> 2, nchunks = nchunk0 + nchunk1;

    // window items of this thread: (pixel, 4-channel group within the chunk).  Their addresses are recomputed at every chunk
    // (a few dozen integer operations against 27 TC TR MFMAs) instead of living in 4 W_CNT registers for the whole kernel:
    // the accumulators and the weight ring need the register file
    f32x4 rw[W_CNT];
    float amax = 0.f;  // max |x| over everything this thread split (saturation report)
    unsigned rwv = 0;  // bit r: item r holds real data (inside the image, channel group exists)
    // One item per tap: item r of the next chunk is loaded at tap r and written to LDS (split into planes) at tap
    // 9 - W_CNT + r, so that the ~30 VALU instructions of a split sit in the shadow of that tap's MFMAs instead of all
    // W_CNT splits queueing up at one tap, and every load has at least three taps to arrive.
    auto load_window_item = [&](int c, int r) {
        const bool s1 = c >= nchunk0;
        const int cg0 = s1 ? (c - nchunk0) * 4 : c * 4;
        const int Gs = s1 ? p.G1 : p.G0;
        const float* base = s1 ? p.src1 : p.src0;
        const int sh = s1 ? 0 : p.up0;
        const int cs = s1 ? p.cs1 : p.cs0, co = s1 ? p.co1 : p.co0;
        const int id = t + NT * r;
        const int px = id >> 2, q = id & 3;
        const int wy = px / WW, wx = px - wy * WW;
        int iy = ty0 - 1 + wy, ix = tx0 - 1 + wx;
        bool v = id < W_ITEMS && (cg0 + q) < Gs;
        if (p.pad_mode == PAD_REFLECT) {
            iy = reflect_idx(iy, p.H);
            ix = reflect_idx(ix, p.W);
        }
        v = v && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        iy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
        ix = ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix);
        const int off = (((n * (p.H >> sh) + (iy >> sh)) * (p.W >> sh) + (ix >> sh)) * cs) + co + (v ? (cg0 + q) * 4 : 0);
        // masked lanes re-read channel group 0 of a valid pixel
        rw[r] = *reinterpret_cast<const f32x4*>(base + off);
        rwv = (rwv & ~(1u << r)) | ((v ? 1u : 0u) << r);
    };
    auto store_window_item = [&](float* W, int r) {
        const int id = t + NT * r;
        if (id < W_ITEMS) {
            h16x4 hi, lo;
            float* dst = W + (id >> 2) * PS + (id & 3) * 2;  // hi plane: dwords [0, 8), lo plane: [8, 16) of the pixel
            if constexpr (NP == 3) {
                split_f16_planes(((rwv >> r) & 1u) ? rw[r] : f32x4{0.f, 0.f, 0.f, 0.f}, &hi, &lo, amax);
                *reinterpret_cast<h16x4*>(dst) = hi;
                *reinterpret_cast<h16x4*>(dst + 8) = lo;
            } else {
                split_f16_hi(((rwv >> r) & 1u) ? rw[r] : f32x4{0.f, 0.f, 0.f, 0.f}, &hi, amax);
                *reinterpret_cast<h16x4*>(dst) = hi;
            }
        }
    };
    auto load_window = [&](int c) {
#pragma unroll
        for (int r = 0; r < W_CNT; ++r) load_window_item(c, r);
    };
    auto store_window = [&](float* W) {
#pragma unroll
        for (int r = 0; r < W_CNT; ++r) store_window_item(W, r);
    };
    // weight fragments: row (cout) = lane & 31 of block tc, k-block = lane >> 5.  Packed per (tap, chunk, 32-cout block) as
    // [plane][k-block][cout][8 halves]: the 64 lanes of one load instruction read 1 KB of consecutive bytes, lane l its
    // 16 bytes at 16 l (a [cout][plane][k] order made every lane quad touch four different 64-byte segments)
    const unsigned short* wbase = p.wf16 + ((size_t)(n0 + wc * TC * 32) * 32 + (kb * 32 + lp) * 8);
    const size_t w_chunk_stride = (size_t)p.wf16_cout_pad * 32;  // halves per (tap, chunk)
    // [register stage][cout block][plane].  Memory operations retire in order (one vmcnt): waiting for a weight fragment also
    // waits for every older load, so the window loads of the next chunk -- HBM latency -- get exactly as many taps of slack
    // as the fragments are fetched ahead.  One cout block per wave (TC == 1) leaves room for a ring of three = two taps ahead
    // (9 taps = 3 turns of the ring: the stage of a tap is a compile-time constant and nothing is copied at the chunk end);
    // TC == 2 keeps two stages, one tap ahead (a third would spill: 128 accumulator + 48 ring + 16 pixel-fragment +
    // window staging registers)
    // (Round 6: a ring of NINE for the one shape left on this skeleton -- a whole chunk of fragments in registers, tap t of chunk
    // c + 1 requested behind the MFMAs of tap t of chunk c; 183 VGPRs, no scratch -- measured SLOWER on every small-map layer it
    // serves: level-4 128 -> 128 29.4 -> 33.6 us, 96+98 -> 128 42.0 -> 48.1, 128 -> 64 25.4 -> 29.1 (profiles/r6q_heads_ring.txt
    // against r6l_small_map_tiles.txt).  Those launches are not waiting for their weight fragments; reverted.)
    constexpr int R = TC == 1 ? 3 : 2;
    h16x8 wa[R][TC][2];
    auto load_w = [&](int stage, int tap, int c) {
        const unsigned short* g = wbase + ((size_t)tap * nchunks + c) * w_chunk_stride;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            wa[stage][i][0] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 32 * 32);
            if constexpr (NP == 3) wa[stage][i][1] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 32 * 32 + 512);
        }
    };

    f32x16 am[TC][TR], ax[NP == 3 ? TC : 1][NP == 3 ? TR : 1];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TR; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                am[i][j][e] = 0.f;
                if constexpr (NP == 3) ax[i][j][e] = 0.f;
            }

    load_window(0);
    load_w(0, 0, 0);
    if (R == 3) load_w(1, 1, 0);
    store_window(lds);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const float* Wc = lds + (c & 1) * WIN;
        float* Wn = lds + ((c + 1) & 1) * WIN;
        const bool next_chunk = c + 1 < nchunks;
        const int c_next = c + 1;
        constexpr bool SPREAD = W_CNT <= 6;  // (always, with the tiles instantiated below)
        if (next_chunk && !SPREAD) load_window(c_next);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int cur = R == 3 ? tap % 3 : (tap & 1);  // compile-time register stage (the loop is fully unrolled)
            if (R == 3) {
                if (tap < 7)
                    load_w((tap + 2) % 3, tap + 2, c);
                else if (next_chunk)
                    load_w((tap + 2) % 3, tap - 7, c_next);
            } else if (tap < 8)
                load_w(cur ^ 1, tap + 1, c);
            else if (next_chunk)
                load_w(1, 0, c_next);  // tap 8 runs from stage 0: the next chunk's first fragments land in stage 1 ...
            if (SPREAD && next_chunk && tap < W_CNT) load_window_item(c_next, tap);
            // keep the fetch HERE: left alone, the scheduler sinks these loads below the tap's last MFMA (it then needs one
            // register set instead of two) and the L2 round trip is exposed at every tap -- measured 30 % matrix-pipe busy
            __builtin_amdgcn_sched_barrier(0);
            h16x8 xb[TR][2];
#pragma unroll
            for (int j = 0; j < TR; ++j) {
                const float* px = Wc + ((wr * TR + j + ky) * WW + (lp + kx)) * PS + kb * 4;
                xb[j][0] = *reinterpret_cast<const h16x8*>(px);
                if constexpr (NP == 3) xb[j][1] = *reinterpret_cast<const h16x8*>(px + 8);
            }
            // the three product terms block by block per term: consecutive MFMAs never target the same accumulator
            if constexpr (NP == 3) {
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TR; ++j)
                        ax[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i][0], xb[j][1], ax[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TR; ++j)
                        ax[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i][1], xb[j][0], ax[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TR; ++j)
                    am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i][0], xb[j][0], am[i][j], 0, 0, 0);
            if (SPREAD) {
                if (next_chunk && tap >= 9 - W_CNT) store_window_item(Wn, tap - (9 - W_CNT));
            } else if (tap == 4 && next_chunk) {
                store_window(Wn);
            }
        }
        if (R == 2)
#pragma unroll
            for (int i = 0; i < TC; ++i) {  // ... and move to stage 0, where tap 0 expects them (2 TC register-quad copies per chunk)
                wa[0][i][0] = wa[1][i][0];
                if constexpr (NP == 3) wa[0][i][1] = wa[1][i][1];
            }
        __syncthreads();
    }

    f16s_report_clamp(amax);
    // epilogue: register quad g of block (i, j) = couts 8 g + 4 kb .. + 3 of the pixel at x = tx0 + lp, row ty0 + wr TR + j
    // (bias loaded once for the wave's couts, a row's stores back to back: conv_epi_row)
    const int ox = tx0 + lp;
    ConvEpi<4 * TC> epi;
    conv_epi_init(p, epi, [&](int q) { return n0 + (wc * TC + (q >> 2)) * 32 + 8 * (q & 3) + 4 * kb; });
#pragma unroll
    for (int j = 0; j < TR; ++j) {
        const int oy = ty0 + wr * TR + j;
        const bool valid = oy < p.Ho && ox < p.Wo;
        const size_t m = valid ? ((size_t)n * p.Ho + oy) * p.Wo + ox : 0;
        conv_epi_row(p, epi, m, valid, [&](int q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = am[q >> 2][j][4 * (q & 3) + e];
                if constexpr (NP == 3) v[e] += F16S_LO_UNSCALE * ax[q >> 2][j][4 * (q & 3) + e];
            }
            return v;
        });
    }
}

static bool conv_f16s_ok(const ConvParams& p) {
    if (p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1) return false;
    if (p.wf16_cout_pad % 32 != 0 || p.Wo < 24 || p.Ho < 4) return false;
    // small maps stay on the fp32 split-K kernels: with fewer than ~200 of the smallest (32 couts x 4 x 32 px) tiles the
    // grid cannot fill the chip and the un-split K loop makes the launch longer than the fp32 one (measured: pyramid
    // levels 5 / 6 and the depth net's inner layers 2-4x slower, level 4 1.5-2x faster)
    const long long e = (long long)p.N * ((p.Ho + 3) / 4) * ((p.Wo + 31) / 32) * (p.wf16_cout_pad / 32);
    // (round 4: raising the bound to 450 / 900 tiles sends pyramid level 4 / 3 to the generic kernel instead -- the window
    // family's average TFLOP/s rises, 0.265 -> 0.30 / 0.31 of its roofline, and the pair rate FALLS, 287 -> 280 / 272: kept at 200)
    return e >= 200;
}

template <int WC, int WR, int TC, int TR>
static long long f16s_blocks(const ConvParams& p) {
    constexpr int TH = WR * TR, BN = WC * TC * 32;
    if (p.wf16_cout_pad % BN != 0) return 0;
    return (long long)p.N * ((p.Ho + TH - 1) / TH) * ((p.Wo + 31) / 32) * (p.wf16_cout_pad / BN);
}

template <int WC, int WR, int TC, int TR>
static int launch_f16s_cfg(const ConvParams& p, hipStream_t stream, int cfg_id) {
    constexpr int TH = WR * TR, BN = WC * TC * 32;
    const int tiles = p.N * ((p.Ho + TH - 1) / TH) * ((p.Wo + 31) / 32);
    dim3 grid((unsigned)tiles, (unsigned)(p.wf16_cout_pad / BN), 1);
    ConvProfEntry pe;
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventCreate(&pe.e0));
        DFVO_HIP_CHECK(hipEventCreate(&pe.e1));
        pe.cfg = cfg_id;
        DFVO_HIP_CHECK(hipEventRecord(pe.e0, stream));
    }
    if (p.f16_terms == 1)
        hipLaunchKernelGGL((conv_win_f16s_kernel<WC, WR, TC, TR, 1>), grid, dim3(64 * WC * WR), 0, stream, p);
    else
        hipLaunchKernelGGL((conv_win_f16s_kernel<WC, WR, TC, TR, 3>), grid, dim3(64 * WC * WR), 0, stream, p);
    DFVO_HIP_CHECK(hipGetLastError());
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventRecord(pe.e1, stream));
        pe.flops = p.useful_flops;
        const int sh[12] = {p.N, p.H, p.W, p.Ho, p.Wo, (p.G0 + p.G1) * 4, p.cout, p.kh, p.stride, (int)grid.x, (int)grid.y, 1};
        for (int i = 0; i < 12; ++i) pe.shape[i] = sh[i];
        g_prof->push_back(pe);
    }
    return DFVO_OK;
}

// The first skeleton (two waves per SIMD) takes what the one-wave-per-SIMD skeleton of conv_win_f16s2.h leaves: layers with
// fewer than ~200 of its tiles (pyramid level 4, the depth net's outer layers), where one workgroup per CU cannot fill the
// chip.  One shape: 32 couts x (4 x 32) pixels, four waves.  (Rounds 2-3 also instantiated 128- / 64-cout and 8-wave
// tiles of this skeleton; since the second skeleton became the default none of them was reachable.)
static int launch_f16s(const ConvParams& p, hipStream_t stream, int cfg_id) {
    return launch_f16s_cfg<1, 4, 1, 1>(p, stream, cfg_id);
}

// host side: f32 -> (hi, lo) exactly as split_f16_planes does on the device
static unsigned long long g_f16s_clamped_host = 0;  // weights beyond f16's range at pack time (same report as the device counter)
static inline void f16s_split_host(float x, unsigned short* hi, unsigned short* lo) {
    float v = x < -F16S_MAX ? -F16S_MAX : (x > F16S_MAX ? F16S_MAX : x);
    if (v != x && x == x) ++g_f16s_clamped_host;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)((v - (float)h) * F16S_LO_SCALE);
    memcpy(hi, &h, 2);
    memcpy(lo, &l, 2);
}

unsigned* conv_f16s_overflow_counter() { return f16s_clamp_counter(); }

// number of threads (activations) + weights (pack time) that hit the +-65504 saturation of the hi plane since the last reset
int conv_f16s_overflow_count(unsigned long long* n, int reset) {
    unsigned int dev = 0;
    DFVO_HIP_CHECK(hipMemcpyFromSymbol(&dev, HIP_SYMBOL(g_f16s_clamped), sizeof(dev)));
    if (n) *n = (unsigned long long)dev + g_f16s_clamped_host;
    if (reset) {
        const unsigned int zero = 0;
        DFVO_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_f16s_clamped), &zero, sizeof(zero)));
        g_f16s_clamped_host = 0;
    }
    return DFVO_OK;
}

size_t conv_pack_weights_f16s(const float* w, int cout, int c0, int c1, const float* fold_scale, unsigned short* out) {
    const int nch0 = (c0 + 15) / 16, nch1 = (c1 + 15) / 16, nch = nch0 + nch1;
    const int cp = round_up(cout, 32);
    const size_t total = (size_t)9 * nch * cp * 32;
    if (!out) return total;
    memset(out, 0, total * sizeof(unsigned short));
    const int cin = c0 + c1;
    for (int tap = 0; tap < 9; ++tap)
        for (int c = 0; c < nch; ++c)
            for (int co = 0; co < cout; ++co)
                for (int k = 0; k < 16; ++k) {
                    const bool s1 = c >= nch0;
                    const int ch = s1 ? (c - nch0) * 16 + k : c * 16 + k;
                    if (ch >= (s1 ? c1 : c0)) continue;
                    const int ci = s1 ? c0 + ch : ch;
                    float v = w[((size_t)co * cin + ci) * 9 + tap];
                    if (fold_scale) v *= fold_scale[co];
                    // [tap][chunk][cout / 32][plane][k / 8][cout % 32][k % 8]
                    unsigned short* o = out + (((size_t)tap * nch + c) * cp + (co & ~31)) * 32 + ((k >> 3) * 32 + (co & 31)) * 8 + (k & 7);
                    f16s_split_host(v, o, o + 512);
                }
    return total;
}

