// conv_win_f16s_kernel, second skeleton: ONE wave per SIMD with a large register tile, software-pipelined.
//
// What the first skeleton (conv_win_f16s.h) could not do inside 256 registers at two waves per SIMD: hold the pixel
// fragments of the NEXT tap while the current tap's MFMAs run.  Its waves issue their eight ds_read_b128 just in time, wait,
// then issue 12 MFMAs (384 cycles); the matrix pipe of a SIMD idles whenever both of its waves sit in that phase together
// (counters, profiles/r2c_pmc_f16s_L2_128x128.txt: 41.6 % busy; a memory-free loop of that shape: 46 % of peak).
// Here a wave owns (32 TC) couts x TR rows x 32 pixels with TC x TR up to 2 x 4: 24 MFMAs (768 cycles) per tap against 4
// weight-fragment loads and 8 pixel-fragment reads, the 2 x TC x TR x 16 accumulator registers in AGPRs, and
//   * the pixel fragments of tap t + 1 are read from LDS while the MFMAs of tap t issue (two register sets),
//   * the weight fragments of tap t + 2 are requested from L2 at tap t (ring of three, as before),
//   * the window items of the next chunk are loaded and split one per tap (as before),
// so the only exposed latencies are the first pixel-fragment read after the per-chunk barrier and the barrier itself
// (once per 9 taps = 216 MFMAs).  One 4-wave workgroup per CU (512 registers per lane: __launch_bounds__(256, 1)).
// Same arithmetic, window layout, weight packing and epilogue as conv_win_f16s_kernel.
#pragma once
// (included inside namespace dfvo, after conv_win_f16s.h)

template <class F, int... T>
__device__ __forceinline__ void f16s2_static_for_impl(F&& f, std::integer_sequence<int, T...>) {
    (f(std::integral_constant<int, T>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void f16s2_static_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
    f16s2_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// NP: products per term -- 3 = f16x3 (hi x lo, lo x hi into the cross sums, hi x hi into the main sums), 1 = the "f16"
// mode (hi x hi only: the lo planes are neither written to LDS nor loaded, the cross set does not exist)
// (Round 6, VERDICT r5 task 4's experiment: a TIMING-ONLY instantiation that took the window item as already split by its producer
// -- 16 bytes = [4 hi halves | 4 lo halves], stored to LDS as they come, no v_cvt / v_sub / v_mul / v_max3 between the taps' MFMA
// groups -- ran the level-2 128 -> 128 layer in 189-193 us instead of 202-211, 64+66 -> 128 in 204 instead of 218, the 1920 x 1280
// layer in 972 instead of 1048 (-7 %): above the kill criterion of 175 us, and the price is a second activation format through
// every producer and consumer of both nets.  Not built; commit "Batched window / halo loads ..." holds the instantiation,
// profiles/r6k_window_presplit_timing.txt the numbers.)
// LDS floats of one tile shape (two window buffers)
template <int WR, int TR>
constexpr int f16s2_lds_floats() {
    return 2 * (((WR * TR + 2) * 34 + 1) * 20);
}

// ONE tile: (32 WC TC) couts from n0 x (WR TR) rows from ty0 x 32 columns from tx0 of sample n.  lds: f16s2_lds_floats<WR, TR>()
// floats.  (Round 6: the kernel body as a device function, so that a launch can hold tiles of two heights -- see
// conv_win_f16s2_mix_kernel.)
template <int WC, int WR, int TC, int TR, int NP>
__device__ __forceinline__ void conv_win_f16s2_tile(const ConvParams& p, float* __restrict__ lds, const int n, const int ty0,
                                                    const int tx0, const int n0) {
    constexpr int NT = 64 * WC * WR;
    constexpr int NPL = NP == 3 ? 2 : 1;  // operand planes in use
    constexpr int TH = WR * TR, TW = 32, WH = TH + 2, WW = TW + 2, PS = 20;  // pixel stride in dwords (80 bytes)
    constexpr int WIN = (WH * WW + 1) * PS;  // + one pixel slot that absorbs the stores of the items beyond the window
    constexpr int W_ITEMS = WH * WW * 4;
    constexpr int W_CNT = (W_ITEMS + NT - 1) / NT;
    static_assert(W_CNT <= 9, "one window item per tap");
    static_assert(WC * WR == 4, "4 waves per block");
    static_assert(2 * WIN == f16s2_lds_floats<WR, TR>(), "LDS size helper out of step");

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wc = wave / WR, wr = wave % WR;
    const int lp = lane & 31, kb = lane >> 5;
    const int nchunk0 = (p.G0 + 3) >> 2, nchunk1 = (p.G1 + 3) >> 2, nchunks = nchunk0 + nchunk1;

    // window items of this thread: (pixel, 4-channel group q = t & 3 of the chunk).  Their pixel offsets inside both sources
    // are computed ONCE (the register file has room at one wave per SIMD): what is left per chunk is one add and one
    // compare per item -- in the first skeleton the per-item address arithmetic and the kernarg re-loads it drags along
    // (SGPR pressure) sat between the taps' MFMA groups, where nothing covers them.
    f32x4 rw[W_CNT];
    float amax = 0.f;
    unsigned rwv = 0;
    int w_off0[W_CNT], w_off1[W_CNT];
    unsigned w_ok = 0;
    const int wq = t & 3;  // (NT is a multiple of 4: every item of a thread has the same channel group)
#pragma unroll
    for (int r = 0; r < W_CNT; ++r) {
        const int id = t + NT * r;
        const int px = id >> 2;
        const int wy = px / WW, wx = px - wy * WW;
        int iy = ty0 - 1 + wy, ix = tx0 - 1 + wx;
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
    const int G0 = __builtin_amdgcn_readfirstlane(p.G0), G1 = __builtin_amdgcn_readfirstlane(p.G1);
    // per-chunk uniforms of the window prefetch (which source, which 4-channel group, whether it exists): set once per chunk
    // by set_chunk(), so that the taps' requests are plain loads -- no uniform branch inside the 9-tap scheduling region
    bool ch_s1 = false, ch_v = true;
    int ch_cg4 = 0;
    const float* ch_base = p.src0;
    auto set_chunk = [&](int c_in) {
        const int c = c_in;
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
        float* dst = W + px * PS + wq * 2;
        if constexpr (NP == 3) {
            split_f16_planes(((rwv >> r) & 1u) ? rw[r] : f32x4{0.f, 0.f, 0.f, 0.f}, &hi, &lo, amax);
            *reinterpret_cast<h16x4*>(dst) = hi;
            *reinterpret_cast<h16x4*>(dst + 8) = lo;
        } else {
            split_f16_hi(((rwv >> r) & 1u) ? rw[r] : f32x4{0.f, 0.f, 0.f, 0.f}, &hi, amax);
            *reinterpret_cast<h16x4*>(dst) = hi;
        }
    };
    const unsigned short* wbase = p.wf16 + ((size_t)(n0 + wc * TC * 32) * 32 + (kb * 32 + lp) * 8);
    const size_t w_chunk_stride = (size_t)p.wf16_cout_pad * 32;  // halves per (tap, chunk)
    h16x8 wa[3][TC][2];
    auto load_w = [&](int stage, int tap, int c) {
        const unsigned short* g = wbase + ((size_t)tap * nchunks + c) * w_chunk_stride;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            wa[stage][i][0] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 32 * 32);
            if constexpr (NP == 3) wa[stage][i][1] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 32 * 32 + 512);
        }
    };
    auto load_w_piece = [&](int stage, int tap, int c, int piece) {  // piece = cout tile * NPL + plane
        const unsigned short* g = wbase + ((size_t)tap * nchunks + c) * w_chunk_stride;
        wa[stage][piece / NPL][piece % NPL] = *reinterpret_cast<const h16x8*>(g + (size_t)(piece / NPL) * 32 * 32 + (piece % NPL) * 512);
    };
    h16x8 xb[2][TR][2];
    auto read_x_piece = [&](const float* Wc, int set, int tap, int piece) {  // piece = row tile * NPL + plane
        const int ky = tap / 3, kx = tap - ky * 3, j = piece / NPL;
        const float* px = Wc + ((wr * TR + j + ky) * WW + (lp + kx)) * PS + kb * 4;
        xb[set][j][piece % NPL] = *reinterpret_cast<const h16x8*>(px + (piece % NPL) * 8);
    };
    auto read_x = [&](const float* Wc, int set, int tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int j = 0; j < TR; ++j) {
            const float* px = Wc + ((wr * TR + j + ky) * WW + (lp + kx)) * PS + kb * 4;
            xb[set][j][0] = *reinterpret_cast<const h16x8*>(px);
            if constexpr (NP == 3) xb[set][j][1] = *reinterpret_cast<const h16x8*>(px + 8);
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

    set_chunk(0);
#pragma unroll
    for (int r = 0; r < W_CNT; ++r) load_window_item(r);
    load_w(0, 0, 0);
    load_w(1, 1, 0);
#pragma unroll
    for (int r = 0; r < W_CNT; ++r) store_window_item(lds, r);
    __syncthreads();
    // one chunk = 9 taps, branch-free, so that the whole chunk is one scheduling region and the interleave below is what
    // the hardware sees.  The last chunk prefetches itself again (window into the idle buffer, first two weight
    // fragments) instead of running a second, prefetch-free copy of the loop body: a second instantiation gets its own
    // register assignment and pays ~280 v_accvgpr moves to get there.
    for (int c = 0; c < nchunks; ++c) {
        const float* Wc = lds + (c & 1) * WIN;
        float* Wn = lds + ((c + 1) & 1) * WIN;
        const int c_next = c + 1 < nchunks ? c + 1 : c;
        set_chunk(c_next);
        read_x(Wc, 0, 0);
        f16s2_static_for<9>([&](auto tap_c) {
            constexpr int tap = decltype(tap_c)::value;
            constexpr int cur = tap % 3, xs = tap & 1;
            constexpr int ST0 = 9 - W_CNT;
            constexpr int NM = NP * TC * TR, NR = tap < 8 ? NPL * TR : 0;
            // MFMA k of the tap: product set g = k / (TC * TR) (0: hi x lo, 1: lo x hi -> cross sums, 2: hi x hi -> main sums)
            auto mfma = [&](int k) {
                const int g = k / (TC * TR), i = (k % (TC * TR)) / TR, j = k % TR;
                if constexpr (NP == 3) {
                    if (g == 0)
                        ax[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i][0], xb[xs][j][1], ax[i][j], 0, 0, 0);
                    else if (g == 1)
                        ax[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i][1], xb[xs][j][0], ax[i][j], 0, 0, 0);
                    else
                        am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i][0], xb[xs][j][0], am[i][j], 0, 0, 0);
                } else {
                    am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i][0], xb[xs][j][0], am[i][j], 0, 0, 0);
                }
            };
            // One request per MFMA, each fenced into that MFMA's shadow (the compiler hoists free-standing loads to the
            // top of the region whatever the group hints say): first the next tap's pixel fragments from LDS, then the
            // weight fragments of tap + 2 and one window item of the next chunk from global memory -- they are
            // consumed two taps / one chunk later -- so that nothing but the split of the window item is left between
            // the taps' MFMA groups.
            constexpr int NV = NPL * TC + (tap < W_CNT ? 1 : 0);
            constexpr int NF = NR + NV < NM ? NR + NV : NM;  // fenced slots
            f16s2_static_for<NF>([&](auto k_c) {
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
                        if (v < NPL * TC) {
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
    }
    f16s_report_clamp(amax);

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

// XCD-aware order of `count` tiles: the hardware hands workgroup b to XCD b & 7; each XCD walks a contiguous run of tiles
// (neighbours share halo rows in its L2)
__device__ __forceinline__ int f16s2_xcd_order(int b, int count) {
    const int q = count >> 3, r = count & 7, xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int WC, int WR, int TC, int TR, int NP = 3>
__global__ __launch_bounds__(64 * WC * WR, 1) void conv_win_f16s2_kernel(const ConvParams p) {
    constexpr int TH = WR * TR, TW = 32, BN = WC * TC * 32;
    __shared__ __attribute__((aligned(16))) float lds[f16s2_lds_floats<WR, TR>()];
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int bid = f16s2_xcd_order(blockIdx.x, gridDim.x);
    const int n = bid / (tiles_y * tiles_x);
    const int trem = bid - n * (tiles_y * tiles_x);
    conv_win_f16s2_tile<WC, WR, TC, TR, NP>(p, lds, n, (trem / tiles_x) * TH, (trem % tiles_x) * TW, blockIdx.y * BN);
}

// Tiles of TWO heights in one launch (round 6).  One workgroup per CU, 256 CUs: a launch of T equal tiles takes ceil(T / 256)
// rounds, and at KITTI size the last round is mostly empty -- level-2 64- / 32-cout layers: 836 tiles = 3.27 rounds run as 4,
// 128-cout: 1140 tiles = 4.45 rounds run as 5.  Here the first rows_a rows of every sample are cut into TRA-row tiles (workgroups
// [0, na), dispatched first), the rest into TRB-row tiles that fill the CUs as the tall ones finish; launch_f16s2 picks the
// split by simulating the dispatch.  na_pad = na rounded up to 8 (the pad workgroups exit), so that the XCD of a workgroup is
// b & 7 in both regions.  Every output pixel is computed by the same instruction sequence whatever tile it falls into:
// bit-identical to the single-height launches.
template <int WC, int WR, int TC, int TRA, int TRB, int NP = 3>
__global__ __launch_bounds__(64 * WC * WR, 1) void conv_win_f16s2_mix_kernel(const ConvParams p, int rows_a, int na, int na_pad,
                                                                             int nb) {
    constexpr int THA = WR * TRA, THB = WR * TRB, TW = 32, BN = WC * TC * 32;
    constexpr int LDSF = f16s2_lds_floats<WR, TRA>() > f16s2_lds_floats<WR, TRB>() ? f16s2_lds_floats<WR, TRA>() : f16s2_lds_floats<WR, TRB>();
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    const int tiles_x = (p.Wo + TW - 1) / TW;
    const int b = blockIdx.x;
    if (b < na_pad) {
        const int tiles_y = rows_a / THA;
        const int bid = f16s2_xcd_order(b, na_pad);
        if (bid >= na) return;
        const int n = bid / (tiles_y * tiles_x);
        const int trem = bid - n * (tiles_y * tiles_x);
        conv_win_f16s2_tile<WC, WR, TC, TRA, NP>(p, lds, n, (trem / tiles_x) * THA, (trem % tiles_x) * TW, blockIdx.y * BN);
    } else {
        const int tiles_y = (p.Ho - rows_a + THB - 1) / THB;
        const int bid = f16s2_xcd_order(b - na_pad, nb);
        const int n = bid / (tiles_y * tiles_x);
        const int trem = bid - n * (tiles_y * tiles_x);
        conv_win_f16s2_tile<WC, WR, TC, TRB, NP>(p, lds, n, rows_a + (trem / tiles_x) * THB, (trem % tiles_x) * TW, blockIdx.y * BN);
    }
}

template <int WC, int WR, int TC, int TR>
static long long f16s2_blocks(const ConvParams& p) {
    constexpr int TH = WR * TR, BN = WC * TC * 32;
    if (p.wf16_cout_pad % BN != 0) return 0;
    return (long long)p.N * ((p.Ho + TH - 1) / TH) * ((p.Wo + 31) / 32) * (p.wf16_cout_pad / BN);
}

template <int WC, int WR, int TC, int TR>
static int launch_f16s2_cfg(const ConvParams& p, hipStream_t stream, int cfg_id) {
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
        hipLaunchKernelGGL((conv_win_f16s2_kernel<WC, WR, TC, TR, 1>), grid, dim3(64 * WC * WR), 0, stream, p);
    else
        hipLaunchKernelGGL((conv_win_f16s2_kernel<WC, WR, TC, TR, 3>), grid, dim3(64 * WC * WR), 0, stream, p);
    DFVO_HIP_CHECK(hipGetLastError());
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventRecord(pe.e1, stream));
        pe.flops = p.useful_flops;
        const int sh[12] = {p.N, p.H, p.W, p.Ho, p.Wo, (p.G0 + p.G1) * 4, p.cout, p.kh, p.stride, (int)grid.x, (int)grid.y, 2};
        for (int i = 0; i < 12; ++i) pe.shape[i] = sh[i];
        g_prof->push_back(pe);
    }
    return DFVO_OK;
}

template <int WC, int WR, int TC, int TRA, int TRB>
static int launch_f16s2_mix(const ConvParams& p, hipStream_t stream, int cfg_id, int rows_a) {
    constexpr int THA = WR * TRA, THB = WR * TRB, BN = WC * TC * 32;
    const int tiles_x = (p.Wo + 31) / 32;
    const int na = p.N * (rows_a / THA) * tiles_x, na_pad = (na + 7) & ~7;
    const int nb = p.N * ((p.Ho - rows_a + THB - 1) / THB) * tiles_x;
    dim3 grid((unsigned)(na_pad + nb), (unsigned)(p.wf16_cout_pad / BN), 1);
    ConvProfEntry pe;
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventCreate(&pe.e0));
        DFVO_HIP_CHECK(hipEventCreate(&pe.e1));
        pe.cfg = cfg_id;
        DFVO_HIP_CHECK(hipEventRecord(pe.e0, stream));
    }
    if (p.f16_terms == 1)
        hipLaunchKernelGGL((conv_win_f16s2_mix_kernel<WC, WR, TC, TRA, TRB, 1>), grid, dim3(64 * WC * WR), 0, stream, p, rows_a, na, na_pad, nb);
    else
        hipLaunchKernelGGL((conv_win_f16s2_mix_kernel<WC, WR, TC, TRA, TRB, 3>), grid, dim3(64 * WC * WR), 0, stream, p, rows_a, na, na_pad, nb);
    DFVO_HIP_CHECK(hipGetLastError());
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventRecord(pe.e1, stream));
        pe.flops = p.useful_flops;
        const int sh[12] = {p.N, p.H, p.W, p.Ho, p.Wo, (p.G0 + p.G1) * 4, p.cout, p.kh, p.stride, (int)grid.x, (int)grid.y, 2};
        for (int i = 0; i < 12; ++i) pe.shape[i] = sh[i];
        g_prof->push_back(pe);
    }
    return DFVO_OK;
}

// Greedy dispatch of `ca` tiles of cost `wa` followed by `cb` tiles of cost `wb` onto 256 CUs (one workgroup per CU): the time
// the last CU finishes.  What the hardware does with a launch of this skeleton, to the accuracy of "a tile costs its rows".
static double f16s2_makespan(long long ca, double wa, long long cb, double wb) {
    const int ncu = 256;
    std::vector<double> h(ncu, 0.0);  // min-heap on finish times
    auto cmp = [](double x, double y) { return x > y; };
    for (long long i = 0; i < ca + cb; ++i) {
        std::pop_heap(h.begin(), h.end(), cmp);
        h.back() += i < ca ? wa : wb;
        std::push_heap(h.begin(), h.end(), cmp);
    }
    return *std::max_element(h.begin(), h.end());
}

// rows_a of the best launch for tiles of th3 (TR = 3) and th2 (TR = 2) rows: 0 = all th2, >= Ho = all th3, else the mixed launch
// (taken only where the simulation promises more than 3 % over the better single height).  Cached per shape: the eager
// (graph-less) path asks at every launch.
static int f16s2_pick_rows_a(const ConvParams& p, int th3, int th2, int ny, bool allow_mix) {
    if (ny != 1) allow_mix = false;  // (several cout blocks: blockIdx.y advances last, the two heights would alternate per block)
    struct Key { int ho, per_row, th3, mix; bool operator<(const Key& o) const { return std::tie(ho, per_row, th3, mix) < std::tie(o.ho, o.per_row, o.th3, o.mix); } };
    static std::map<Key, int> cache;
    static std::mutex mu;
    const int per_row = p.N * ((p.Wo + 31) / 32) * ny;
    const Key key{p.Ho, per_row, th3, allow_mix ? 1 : 0};
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    const double w3 = th3, w2 = th2 * 1.1;  // (small tiles: fewer MFMAs per load)
    const double pure3 = f16s2_makespan((long long)((p.Ho + th3 - 1) / th3) * per_row, w3, 0, w2);
    const double pure2 = f16s2_makespan(0, w3, (long long)((p.Ho + th2 - 1) / th2) * per_row, w2);
    int best_rows = pure3 <= pure2 ? p.Ho : 0;
    double best = pure3 <= pure2 ? pure3 : pure2;
    if (allow_mix) {
        const double need = best * 0.97;
        for (int a = 1; a * th3 < p.Ho; ++a) {
            const int rest = p.Ho - a * th3;
            const double m = f16s2_makespan((long long)a * per_row, w3, (long long)((rest + th2 - 1) / th2) * per_row, w2);
            if (m < need && m < best) {
                best = m;
                best_rows = a * th3;
            }
        }
    }
    std::lock_guard<std::mutex> g(mu);
    cache[key] = best_rows;
    return best_rows;
}

// Tile choice for the one-wave-per-SIMD skeleton (one workgroup per CU): the rows per tile that minimise
// rounds x rows for the layer's grid among the instantiated shapes.  Returns F16S2_NOT_APPLICABLE (positive: every
// DFVO_ERR_* code is negative) when the layer should stay on the first
// skeleton (small grids: fewer than ~200 workgroups cannot fill the chip at one workgroup per CU).
constexpr int F16S2_NOT_APPLICABLE = 1;
// (Round 4 also built a third skeleton -- the same tile loop walking a run of tiles per workgroup, persistent at the limit.
// One launch at a time it was 6-9 % faster, in the pipeline 5 % SLOWER (a CU is handed back only when a workgroup ends and
// the solver chain's kernels wait for CUs); it stayed opt-in for a round and was removed in round 5.  Records:
// profiles/r4c_tile_run_*.txt, r4g_tile_run_policy_ab.txt, r4h_persistent_reserve_ab.txt)
static int launch_f16s2(const ConvParams& p, hipStream_t stream, int cfg_id) {
    // (per layer the two skeletons are within 2 % of each other; inside the pipeline this one gives +3 % pairs/s: half the
    // resident net waves next to the solver's kernels -- round 3)
    // DFVO_WIN_MIX=0: single-height launches only (the round 3-5 rule: the height that minimises rounds x rows)
    static const bool mix_env = !(getenv("DFVO_WIN_MIX") && atoi(getenv("DFVO_WIN_MIX")) == 0);
    static const int force_tr = getenv("DFVO_WIN_FORCE_TR") ? atoi(getenv("DFVO_WIN_FORCE_TR")) : 0;  // (calibration hook: 2 / 3)
    if (force_tr == 2 || force_tr == 3) {
        if (p.wf16_cout_pad % 128 == 0) return force_tr == 3 ? launch_f16s2_cfg<2, 2, 2, 3>(p, stream, cfg_id) : launch_f16s2_cfg<2, 2, 2, 2>(p, stream, cfg_id);
        if (p.wf16_cout_pad % 64 == 0) return force_tr == 3 ? launch_f16s2_cfg<1, 4, 2, 3>(p, stream, cfg_id) : launch_f16s2_cfg<1, 4, 2, 2>(p, stream, cfg_id);
        return force_tr == 3 ? launch_f16s2_cfg<1, 4, 1, 3>(p, stream, cfg_id) : launch_f16s2_cfg<1, 4, 1, 2>(p, stream, cfg_id);
    }
    // (the f16 mode's two-row shapes run two workgroups per CU: like the 32-cout layers below they keep single heights)
    const bool mix_ok = mix_env && p.f16_terms != 1;
    if (p.wf16_cout_pad % 128 == 0) {
        // (an 8-row tile -- 256 accumulator registers -- does not fit: the allocator spills inside the tap loop)
        if (f16s2_blocks<2, 2, 2, 2>(p) < 200) return F16S2_NOT_APPLICABLE;
        const int ra = f16s2_pick_rows_a(p, 6, 4, p.wf16_cout_pad / 128, mix_ok);
        if (ra >= p.Ho) return launch_f16s2_cfg<2, 2, 2, 3>(p, stream, cfg_id);
        if (ra <= 0) return launch_f16s2_cfg<2, 2, 2, 2>(p, stream, cfg_id);
        return launch_f16s2_mix<2, 2, 2, 3, 2>(p, stream, cfg_id, ra);
    }
    if (p.wf16_cout_pad % 64 == 0) {
        if (f16s2_blocks<1, 4, 2, 2>(p) < 200) return F16S2_NOT_APPLICABLE;
        const int ra = f16s2_pick_rows_a(p, 12, 8, p.wf16_cout_pad / 64, mix_ok);
        if (ra >= p.Ho) return launch_f16s2_cfg<1, 4, 2, 3>(p, stream, cfg_id);
        if (ra <= 0) return launch_f16s2_cfg<1, 4, 2, 2>(p, stream, cfg_id);
        return launch_f16s2_mix<1, 4, 2, 3, 2>(p, stream, cfg_id, ra);
    }
    // 32-cout layers: single heights only.  Their 8-row shape needs 198 registers and runs TWO workgroups per CU; inside a mixed
    // launch it would carry the 12-row shape's 262 and run one (measured: 64 -> 32 44 -> 52 us, 32 -> 32 31 -> 34,
    // profiles/r6u_window_mixed_tiles.txt).
    const long long b3 = f16s2_blocks<1, 4, 1, 3>(p), b2 = f16s2_blocks<1, 4, 1, 2>(p);
    if (b2 < 200) return F16S2_NOT_APPLICABLE;
    auto cost = [&](long long blocks, int rows) { return blocks <= 0 ? (1LL << 60) : ((blocks + 255) / 256) * rows; };
    if (cost(b3, 12) <= cost(b2, 8)) return launch_f16s2_cfg<1, 4, 1, 3>(p, stream, cfg_id);
    return launch_f16s2_cfg<1, 4, 1, 2>(p, stream, cfg_id);
}
