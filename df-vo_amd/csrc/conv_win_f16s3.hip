// conv_win_f16s kernel, third skeleton: the one-wave-per-SIMD, software-pipelined tile loop of conv_win_f16s2.h, with a
// workgroup walking a RUN of consecutive tiles instead of one.
//
// What the second skeleton left on the table: launch time over the number of 16-channel chunks is a
// straight line whose intercept -- 24-44 us of a ~200 us launch, i.e. ~8 us per tile at 4.5 tile rounds -- is work the
// matrix pipe does not see: the workgroup's start, the per-item window offsets, the first window's trip global -> split ->
// LDS, the first weight fragments, and the epilogue.  At one workgroup per CU (512 registers per lane) nothing else can
// run on the CU in the meantime, so that time is lost CU time whatever the other streams of the pipeline do.
// Here the LAST chunk of a tile prefetches the FIRST chunk of the run's next tile (window items into the idle LDS buffer,
// weight fragments of taps 0 / 1 into the ring) where the second skeleton prefetched the last chunk again into nowhere:
// after the tile's epilogue stores the next tile's MFMAs start at once.  Per run the prologue is paid once.
// The run length is the launcher's (p.tile_run): long runs amortise more, short runs hand CUs back to the pipeline's other
// streams more often (a CU is released only when a workgroup ends).  tile_run == 1 is the second skeleton.
// Same arithmetic and operand order per output pixel as conv_win_f16s_kernel / conv_win_f16s2_kernel (K order: chunk, tap;
// products hi x lo, lo x hi into the cross sums, hi x hi into the main sums): results are bit-identical for every run
// length.  Reference layers: /root/reference/libs/deep_models/flow/lite_flow_net/lite_flow_net.py:171-179,211-224.
#include "dfvo_common.h"

#include <type_traits>
#include <utility>

namespace dfvo {

#include "conv_epi.h"
#include "conv_f16_split.h"

template <class F, int... T>
__device__ __forceinline__ void f16s3_static_for_impl(F&& f, std::integer_sequence<int, T...>) {
    (f(std::integral_constant<int, T>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void f16s3_static_for(F&& f) {
    f16s3_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int WC, int WR, int TC, int TR>
__global__ __launch_bounds__(64 * WC * WR, 1) void conv_win_f16s3_kernel(const ConvParams p) {
    constexpr int NT = 64 * WC * WR;
    constexpr int TH = WR * TR, TW = 32, WH = TH + 2, WW = TW + 2, PS = 20;  // pixel stride in dwords (80 bytes)
    constexpr int BN = WC * TC * 32;
    constexpr int WIN = (WH * WW + 1) * PS;  // + one pixel slot that absorbs the stores of the items beyond the window
    constexpr int W_ITEMS = WH * WW * 4;
    constexpr int W_CNT = (W_ITEMS + NT - 1) / NT;
    static_assert(W_CNT <= 9, "one window item per tap");
    static_assert(WC * WR == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) float lds[2 * WIN];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wc = wave / WR, wr = wave % WR;
    const int lp = lane & 31, kb = lane >> 5;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int tiles = p.N * tiles_y * tiles_x;
    const int nb = gridDim.x;
    int run = blockIdx.x;
    {  // XCD-aware order: each XCD walks a contiguous range of runs (neighbouring tiles share halo rows in its L2)
        const int q = nb >> 3, r = nb & 7, xcd = run & 7, k = run >> 3;
        run = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    int tile = __builtin_amdgcn_readfirstlane(run * p.tile_run);
    const int tile_end = __builtin_amdgcn_readfirstlane(tile + p.tile_run < tiles ? tile + p.tile_run : tiles);
    const int n0 = blockIdx.y * BN;
    const int nchunk0 = (p.G0 + 3) >> 2, nchunk1 = (p.G1 + 3) >> 2, nchunks = nchunk0 + nchunk1;

    // window items of this thread: (pixel, 4-channel group q = t & 3 of the chunk); their pixel offsets inside both sources
    // are computed once per TILE (at the top of the previous tile's last chunk for all but the run's first tile)
    f32x4 rw[W_CNT];
    float amax = 0.f;
    unsigned rwv = 0;
    int w_off0[W_CNT], w_off1[W_CNT];
    unsigned w_ok = 0;
    const int wq = t & 3;  // (NT is a multiple of 4: every item of a thread has the same channel group)
    int tn = 0, ty0 = 0, tx0 = 0;  // the tile whose accumulators are live (epilogue coordinates)
    auto tile_offsets = [&](int tl, int& n_o, int& ty_o, int& tx_o) {
        const int n = tl / (tiles_y * tiles_x);
        const int trem = tl - n * (tiles_y * tiles_x);
        const int ty = (trem / tiles_x) * TH, tx = (trem % tiles_x) * TW;
        n_o = n;
        ty_o = ty;
        tx_o = tx;
        w_ok = 0;
#pragma unroll
        for (int r = 0; r < W_CNT; ++r) {
            const int id = t + NT * r;
            const int px = id >> 2;
            const int wy = px / WW, wx = px - wy * WW;
            int iy = ty - 1 + wy, ix = tx - 1 + wx;
            bool v = id < W_ITEMS;
            if (p.pad_mode == PAD_REFLECT) {
                iy = reflect_idx(iy, p.H);
                ix = reflect_idx(ix, p.W);
            }
            v = v && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            iy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
            ix = ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix);
            const int sh = p.up0;
            w_off0[r] = (((n * (p.H >> sh) + (iy >> sh)) * (p.W >> sh) + (ix >> sh)) * p.cs0) + p.co0;
            w_off1[r] = (((n * p.H + iy) * p.W + ix) * p.cs1) + p.co1;
            w_ok |= (v ? 1u : 0u) << r;
        }
    };
    const int G0 = __builtin_amdgcn_readfirstlane(p.G0), G1 = __builtin_amdgcn_readfirstlane(p.G1);
    bool ch_s1 = false, ch_v = true;
    int ch_cg4 = 0;
    const float* ch_base = p.src0;
    auto set_chunk = [&](int c) {
        ch_s1 = c >= nchunk0;
        const int cg = (ch_s1 ? (c - nchunk0) * 4 : c * 4) + wq;
        ch_v = cg < (ch_s1 ? G1 : G0);
        ch_cg4 = ch_v ? cg * 4 : 0;  // masked lanes re-read channel group 0
        ch_base = ch_s1 ? p.src1 : p.src0;
    };
    auto load_window_item = [&](int r) {
        const bool v = ((w_ok >> r) & 1u) && ch_v;
        const int off = (ch_s1 ? w_off1[r] : w_off0[r]) + (((w_ok >> r) & 1u) ? ch_cg4 : 0);
        rw[r] = *reinterpret_cast<const f32x4*>(ch_base + off);
        rwv = (rwv & ~(1u << r)) | ((v ? 1u : 0u) << r);
    };
    auto store_window_item = [&](float* W, int r) {
        const int id = t + NT * r;
        const int px = (id >> 2) < WH * WW ? (id >> 2) : WH * WW;  // (no branch: out-of-window items land in the spare slot)
        h16x4 hi, lo;
        split_f16_planes(((rwv >> r) & 1u) ? rw[r] : f32x4{0.f, 0.f, 0.f, 0.f}, &hi, &lo, amax);
        float* dst = W + px * PS + wq * 2;
        *reinterpret_cast<h16x4*>(dst) = hi;
        *reinterpret_cast<h16x4*>(dst + 8) = lo;
    };
    const unsigned short* wbase = p.wf16 + ((size_t)(n0 + wc * TC * 32) * 32 + (kb * 32 + lp) * 8);
    const size_t w_chunk_stride = (size_t)p.wf16_cout_pad * 32;  // halves per (tap, chunk)
    h16x8 wa[3][TC][2];
    auto load_w = [&](int stage, int tap, int c) {
        const unsigned short* g = wbase + ((size_t)tap * nchunks + c) * w_chunk_stride;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            wa[stage][i][0] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 32 * 32);
            wa[stage][i][1] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 32 * 32 + 512);
        }
    };
    auto load_w_piece = [&](int stage, int tap, int c, int piece) {  // piece = cout tile * 2 + plane
        const unsigned short* g = wbase + ((size_t)tap * nchunks + c) * w_chunk_stride;
        wa[stage][piece >> 1][piece & 1] = *reinterpret_cast<const h16x8*>(g + (size_t)(piece >> 1) * 32 * 32 + (piece & 1) * 512);
    };
    h16x8 xb[2][TR][2];
    auto read_x_piece = [&](const float* Wc, int set, int tap, int piece) {  // piece = row tile * 2 + plane
        const int ky = tap / 3, kx = tap - ky * 3, j = piece >> 1;
        const float* px = Wc + ((wr * TR + j + ky) * WW + (lp + kx)) * PS + kb * 4;
        xb[set][j][piece & 1] = *reinterpret_cast<const h16x8*>(px + (piece & 1) * 8);
    };
    auto read_x = [&](const float* Wc, int set, int tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int j = 0; j < TR; ++j) {
            const float* px = Wc + ((wr * TR + j + ky) * WW + (lp + kx)) * PS + kb * 4;
            xb[set][j][0] = *reinterpret_cast<const h16x8*>(px);
            xb[set][j][1] = *reinterpret_cast<const h16x8*>(px + 8);
        }
    };

    f32x16 am[TC][TR], ax[TC][TR];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TR; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) am[i][j][e] = ax[i][j][e] = 0.f;
    };
    zero_acc();

    // ---- prologue of the run: first tile's offsets, first window, first two weight fragments
    tile_offsets(tile, tn, ty0, tx0);
    set_chunk(0);
#pragma unroll
    for (int r = 0; r < W_CNT; ++r) load_window_item(r);
    load_w(0, 0, 0);
    load_w(1, 1, 0);
#pragma unroll
    for (int r = 0; r < W_CNT; ++r) store_window_item(lds, r);
    __syncthreads();
    int buf = 0;  // LDS buffer holding the current chunk's window

    for (;;) {  // tiles of the run
        const bool more = tile + 1 < tile_end;  // wave-uniform (SGPR)
        int nn = tn, nty = ty0, ntx = tx0;
        for (int c = 0; c < nchunks; ++c) {
            const float* Wc = lds + buf * WIN;
            float* Wn = lds + (buf ^ 1) * WIN;
            const bool last = c + 1 == nchunks;
            // what this chunk prefetches: the tile's next chunk; in the last chunk the NEXT TILE's first chunk (whose
            // window offsets replace this tile's, no longer needed: this tile's last window is already in LDS); the run's
            // very last chunk prefetches itself again into the idle buffer (one 9-tap code path, as in the second skeleton)
            const int c_next = last ? (more ? 0 : c) : c + 1;
            if (last && more) tile_offsets(tile + 1, nn, nty, ntx);
            set_chunk(c_next);
            read_x(Wc, 0, 0);
            f16s3_static_for<9>([&](auto tap_c) {
                constexpr int tap = decltype(tap_c)::value;
                constexpr int cur = tap % 3, xs = tap & 1;
                constexpr int ST0 = 9 - W_CNT;
                constexpr int NM = 3 * TC * TR, NR = tap < 8 ? 2 * TR : 0;
                // MFMA k of the tap: product set g = k / (TC * TR) (0: hi x lo, 1: lo x hi -> cross sums, 2: hi x hi -> main sums)
                auto mfma = [&](int k) {
                    const int g = k / (TC * TR), i = (k % (TC * TR)) / TR, j = k % TR;
                    if (g == 0)
                        ax[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i][0], xb[xs][j][1], ax[i][j], 0, 0, 0);
                    else if (g == 1)
                        ax[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i][1], xb[xs][j][0], ax[i][j], 0, 0, 0);
                    else
                        am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i][0], xb[xs][j][0], am[i][j], 0, 0, 0);
                };
                // One request per MFMA, each fenced into that MFMA's shadow: first the next tap's pixel fragments from LDS,
                // then the weight fragments of tap + 2 and one window item of the next chunk from global memory.
                constexpr int NV = 2 * TC + (tap < W_CNT ? 1 : 0);
                constexpr int NF = NR + NV < NM ? NR + NV : NM;  // fenced slots
                f16s3_static_for<NF>([&](auto k_c) {
                    constexpr int k = decltype(k_c)::value;
                    mfma(k);
                    constexpr int per = NF > 0 ? (NR + NV + NF - 1) / NF : 1;  // requests per slot (1 unless the tile has few MFMAs)
#pragma unroll
                    for (int u = 0; u < per; ++u) {
                        const int q = k * per + u;
                        if (q < NR) {
                            read_x_piece(Wc, xs ^ 1, tap + 1, q);
                        } else if (q < NR + NV) {
                            const int v = q - NR;
                            if (v < 2 * TC) {
                                if (tap < 7)
                                    load_w_piece((tap + 2) % 3, tap + 2, c, v);
                                else
                                    load_w_piece((tap + 2) % 3, tap - 7, c_next, v);
                            } else if (tap < W_CNT) {
                                load_window_item(tap);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                if (tap >= ST0) store_window_item(Wn, tap - ST0);
#pragma unroll
                for (int k = NF; k < NM; ++k) mfma(k);
#pragma unroll
                for (int k = NF; k < NM; ++k) {  // the window item's split (about 30 VALU instructions), its two LDS writes last
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    if (tap >= ST0) __builtin_amdgcn_sched_group_barrier(0x2, NM - NF >= 16 ? 2 : 4, 0);
                }
                if (tap >= ST0) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            __syncthreads();
            buf ^= 1;
        }

        // ---- epilogue of the tile (the next tile's first window and weight fragments are already in LDS / registers).
        // Vector path only (the launcher sends layers with ragged couts, unaligned views, residuals or sigmoid to the
        // second skeleton): the bias quads of the wave's couts are loaded together, then every row's stores go out back to
        // back -- no load is waited for behind a store (stores count on vmcnt on this architecture).
        {
            const int ox = tx0 + lp;
            f32x4 bq[4 * TC];
#pragma unroll
            for (int q = 0; q < 4 * TC; ++q)
                bq[q] = *reinterpret_cast<const f32x4*>(p.bias + n0 + (wc * TC + (q >> 2)) * 32 + 8 * (q & 3) + 4 * kb);
            auto rows = [&](auto elu_c) {
                constexpr bool ELU = decltype(elu_c)::value;
                const float slope = p.act == ACT_LEAKY ? p.act_param : 1.f;
                const bool relu = p.act == ACT_RELU;
#pragma unroll
                for (int j = 0; j < TR; ++j) {
                    const int oy = ty0 + wr * TR + j;
                    const bool valid = oy < p.Ho && ox < p.Wo;
                    const size_t m = valid ? ((size_t)tn * p.Ho + oy) * p.Wo + ox : 0;
                    float* d = p.dst + m * p.dst_cs + p.dst_co + n0 + wc * TC * 32 + 4 * kb;
#pragma unroll
                    for (int q = 0; q < 4 * TC; ++q) {
                        f32x4 x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = am[q >> 2][j][4 * (q & 3) + e] + F16S_LO_UNSCALE * ax[q >> 2][j][4 * (q & 3) + e] + bq[q][e];
                            x[e] = ELU ? (v > 0.f ? v : p.act_param * expm1f(v)) : (v > 0.f ? v : (relu ? 0.f : v * slope));
                        }
                        if (valid) *reinterpret_cast<f32x4*>(d + (q >> 2) * 32 + 8 * (q & 3)) = x;
                    }
                }
            };
            if (p.act == ACT_ELU)  // wave-uniform
                rows(std::true_type{});
            else
                rows(std::false_type{});
        }
        if (!more) break;
        ++tile;
        tn = nn;
        ty0 = nty;
        tx0 = ntx;
        zero_acc();
    }
    if (amax > F16S_MAX) atomicAdd(p.f16s_clamp_ctr, 1u);
}

template <int WC, int WR, int TC, int TR>
static int f16s3_launch(const ConvParams& p, hipStream_t stream, int* grid_xy) {
    constexpr int TH = WR * TR, BN = WC * TC * 32;
    const int tiles = p.N * ((p.Ho + TH - 1) / TH) * ((p.Wo + 31) / 32);
    const int runs = (tiles + p.tile_run - 1) / p.tile_run;
    dim3 grid((unsigned)runs, (unsigned)(p.wf16_cout_pad / BN), 1);
    hipLaunchKernelGGL((conv_win_f16s3_kernel<WC, WR, TC, TR>), grid, dim3(64 * WC * WR), 0, stream, p);
    DFVO_HIP_CHECK(hipGetLastError());
    if (grid_xy) {
        grid_xy[0] = (int)grid.x;
        grid_xy[1] = (int)grid.y;
    }
    return DFVO_OK;
}

// the epilogue here is the 16-byte vector path only
bool f16s3_eligible(const ConvParams& p) {
    return p.res == nullptr && p.act != ACT_SIGMOID && ((p.dst_cs | p.dst_co) & 3) == 0 && p.cout == p.wf16_cout_pad &&
           (p.cout & 31) == 0 && p.dst_zero_to <= p.cout;
}

// shape: 0 = <2,2,2,3> (6 rows x 128 couts), 1 = <2,2,2,2> (4 x 128), 2 = <1,4,2,3> (12 x 64), 3 = <1,4,2,2> (8 x 64),
// 4 = <1,4,1,3> (12 x 32), 5 = <1,4,1,2> (8 x 32).  p.tile_run >= 1 and p.f16s_clamp_ctr must be set.
int launch_f16s3_shape(const ConvParams& p, int shape, hipStream_t stream, int* grid_xy) {
    DFVO_ARG_CHECK(p.tile_run >= 1 && p.f16s_clamp_ctr, "conv_win_f16s3: tile_run / clamp counter not set");
    DFVO_ARG_CHECK(f16s3_eligible(p), "conv_win_f16s3: the layer needs the generic epilogue (second skeleton)");
    switch (shape) {
        case 0: return f16s3_launch<2, 2, 2, 3>(p, stream, grid_xy);
        case 1: return f16s3_launch<2, 2, 2, 2>(p, stream, grid_xy);
        case 2: return f16s3_launch<1, 4, 2, 3>(p, stream, grid_xy);
        case 3: return f16s3_launch<1, 4, 2, 2>(p, stream, grid_xy);
        case 4: return f16s3_launch<1, 4, 1, 3>(p, stream, grid_xy);
        case 5: return f16s3_launch<1, 4, 1, 2>(p, stream, grid_xy);
    }
    DFVO_ARG_CHECK(false, "conv_win_f16s3: unknown shape");
}

}  // namespace dfvo
