// conv_win_f16a_kernel: the f16x3 window kernel with ONE accumulator set, two workgroups per CU (round 5).
//
// Why.  Both earlier skeletons keep two accumulator sets per output block -- "main" (hi x hi) and "cross" (hi x lo + lo x hi,
// scaled by 2^11 because the activations' lo plane is stored scaled) -- 32 registers per 32 x 32 block.  The 128-cout x
// 6-row tile therefore needs 192 accumulator registers, which forces one wave per SIMD (conv_win_f16s2.h): a workgroup's
// prologue (first window from HBM), its epilogue (192 accumulator reads, 24 quad stores per wave), the per-chunk barrier
// and every s_waitcnt leave the matrix pipe of that SIMD idle (counters, profiles/r4q_pmc_window_layers.txt: pipe busy
// 61 % of a wave's life, 19 % parked at waitcnt / barrier, the rest prologue / epilogue; tail quantisation at one
// workgroup per CU another 11-18 %).
// Here the three products of a term land in the SAME accumulator:
//        w x  ~=  wh xh  +  (2^-11 wh) xl'  +  wl xh            xl' = 2^11 (x - xh)   (the activation planes, unchanged)
// which needs a THIRD weight plane wh2 = 2^-11 wh and the UNSCALED residue wl = w - wh.  Both would fall into f16's
// subnormal range for ordinary weights (|w| < 2^-3), so the weights of every output channel are pre-scaled by a power
// of two S_co with max |w S_co| in [2^14, 2^15): wh2 and wl keep their bits unless a weight is more than 2^17 times smaller
// than the channel's largest (then its product is below the fp32 rounding of the sum anyway), and the epilogue multiplies
// the accumulator by the exact 1 / S_co before the bias.  Weights are static: all of this is packing (conv_pack_weights_f16a).
// 96 accumulator registers for the 2 x 3 block tile, __launch_bounds__(256, 2): two 4-wave workgroups per CU = two waves per
// SIMD, one workgroup's prologue / epilogue / barrier under the other's MFMAs, a CU handed back after every ~half-size
// time slice (the property the pipeline measurements say matters: the persistent variants that held CUs longer lost).
// Results differ from the two-set kernels in the last bits (one rounding chain instead of two): gated by the float64
// anchor test (tests/test_nets_gpu.py::test_flownet_distance_to_the_exact_function) and the operator tests, not by CRC.
//
// Loop structure: the first skeleton's (conv_win_f16s.h) -- pixel fragments read just in time, window items of the next
// chunk loaded one per tap and split + stored one per tap, one barrier per 16-channel chunk -- with a single, plane-wise
// reloaded set of weight fragments (see wa below).
#pragma once
// (included inside namespace dfvo, after conv_win_f16s2.h)

template <int WC, int WR, int TC, int TR>
__global__ __launch_bounds__(256, 2) void conv_win_f16a_kernel(const ConvParams p) {
    constexpr int NT = 256;
    constexpr int TH = WR * TR, TW = 32, WH = TH + 2, WW = TW + 2, PS = 20;  // pixel stride in dwords (80 bytes)
    constexpr int BN = WC * TC * 32;
    constexpr int WIN = WH * WW * PS;
    constexpr int W_ITEMS = WH * WW * 4;
    constexpr int W_CNT = (W_ITEMS + NT - 1) / NT;
    static_assert(WC * WR == 4, "4 waves per block");
    static_assert(W_CNT <= 9, "one window item per tap");
    __shared__ __attribute__((aligned(16))) float lds[2 * WIN];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
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

    f32x4 rw[W_CNT];
    float amax = 0.f;
    unsigned rwv = 0;
    auto load_window_item = [&](int c, int r) {
        const bool s1 = c >= nchunk0;
        const int cg0 = s1 ? (c - nchunk0) * 4 : c * 4;
        const int Gs = s1 ? p.G1 : p.G0;
        const float* base = s1 ? p.src1 : p.src0;
        const int sh = s1 ? 0 : p.up0;
        const int cs = s1 ? p.cs1 : p.cs0, co = s1 ? p.co1 : p.co0;
        // (opaque to the optimiser: otherwise the loop-invariant part of every item's address arithmetic -- a dozen values per
        // item -- is hoisted out of the chunk loop and kept in registers this kernel does not have)
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int id = tt + NT * r;
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
        rw[r] = *reinterpret_cast<const f32x4*>(base + off);  // masked lanes re-read channel group 0 of a valid pixel
        rwv = (rwv & ~(1u << r)) | ((v ? 1u : 0u) << r);
    };
    auto store_window_item = [&](float* W, int r) {
        const int id = t + NT * r;
        if (id < W_ITEMS) {
            h16x4 hi, lo;
            split_f16_planes(((rwv >> r) & 1u) ? rw[r] : f32x4{0.f, 0.f, 0.f, 0.f}, &hi, &lo, amax);
            float* dst = W + (id >> 2) * PS + (id & 3) * 2;  // hi plane: dwords [0, 8), lo plane: [8, 16) of the pixel
            *reinterpret_cast<h16x4*>(dst) = hi;
            *reinterpret_cast<h16x4*>(dst + 8) = lo;
        }
    };
    // weight fragments, packed per (tap, chunk, 32-cout block) as [plane wh | wh2 | wl][k-block][cout][8 halves]: 3 x 1 KB of
    // consecutive bytes, lane l its 16 bytes at 16 l of each plane
    const unsigned short* wbase = p.wf16a + ((size_t)(n0 + wc * TC * 32) * 48 + (kb * 32 + lp) * 8);
    const size_t w_chunk_stride = (size_t)p.wf16_cout_pad * 48;  // halves per (tap, chunk)
    // ONE register set of weight fragments, reloaded plane by plane: the plane a product group has just consumed is
    // requested for the NEXT tap right behind that group's last MFMA, i.e. every plane is fetched 2/3 of a tap (2 TC TR
    // MFMAs of this wave, plus whatever the SIMD's other wave issues in between) ahead of its use -- 24 registers instead
    // of the 48 of a two-stage ring, which is what lets the 2 x 3 tile fit 256 registers without scratch.
    h16x8 wa[TC][3];
    auto load_w_plane = [&](int tap, int c, int pl) {
        const unsigned short* g = wbase + ((size_t)tap * nchunks + c) * w_chunk_stride + pl * 512;
#pragma unroll
        for (int i = 0; i < TC; ++i) wa[i][pl] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 32 * 48);
    };

    f32x16 acc[TC][TR];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TR; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
    for (int r = 0; r < W_CNT; ++r) load_window_item(0, r);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) load_w_plane(0, 0, pl);
#pragma unroll
    for (int r = 0; r < W_CNT; ++r) store_window_item(lds, r);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const float* Wc = lds + (c & 1) * WIN;
        float* Wn = lds + ((c + 1) & 1) * WIN;
        const bool next_chunk = c + 1 < nchunks;
        const int c_next = next_chunk ? c + 1 : c;  // (the last chunk re-requests its own first tap: branch-free, never consumed)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int ntap = tap < 8 ? tap + 1 : 0;
            const int nc = tap < 8 ? c : c_next;
            if (next_chunk && tap < W_CNT) load_window_item(c_next, tap);
            h16x8 xb[TR][2];
#pragma unroll
            for (int j = 0; j < TR; ++j) {
                const float* px = Wc + ((wr * TR + j + ky) * WW + (lp + kx)) * PS + kb * 4;
                xb[j][0] = *reinterpret_cast<const h16x8*>(px);
                xb[j][1] = *reinterpret_cast<const h16x8*>(px + 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            // term-major: the three products of a block are TC TR MFMAs apart (no back-to-back dependent accumulators)
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[i][0], xb[j][0], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w_plane(ntap, nc, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[i][1], xb[j][1], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w_plane(ntap, nc, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[i][2], xb[j][0], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w_plane(ntap, nc, 2);
            if (next_chunk && tap >= 9 - W_CNT) store_window_item(Wn, tap - (9 - W_CNT));
        }
        __syncthreads();
    }
    f16s_report_clamp(amax);

    // epilogue: the accumulator carries S_co x the sum; 1 / S_co is a power of two (exact).  Applied in place first, so that
    // the scale quads are dead before the bias quads of the shared epilogue are loaded.
    {
        f32x4 si[4 * TC];
#pragma unroll
        for (int q = 0; q < 4 * TC; ++q)  // (the table is padded to wf16_cout_pad: always a valid address)
            si[q] = *reinterpret_cast<const f32x4*>(p.wf16a_inv + n0 + (wc * TC + (q >> 2)) * 32 + 8 * (q & 3) + 4 * kb);
#pragma unroll
        for (int q = 0; q < 4 * TC; ++q)
#pragma unroll
            for (int j = 0; j < TR; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[q >> 2][j][4 * (q & 3) + e] *= si[q][e];
    }
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
            for (int e = 0; e < 4; ++e) v[e] = acc[q >> 2][j][4 * (q & 3) + e];
            return v;
        });
    }
}

// host side: per output channel the power of two S with max |w S| in [2^14, 2^15) (1 for an all-zero channel), then
// wh = f16(w S), wh2 = f16(2^-11 wh), wl = f16(w S - wh)  -- layout [tap][chunk][cout_pad / 32][3][k / 8][32 couts][8] halves;
// inv[co] = 1 / S (zero-padded to cout_pad with 1).  Returns the number of halves (out may be null).
size_t conv_pack_weights_f16a(const float* w, int cout, int c0, int c1, const float* fold_scale, unsigned short* out, float* inv) {
    const int nch0 = (c0 + 15) / 16, nch1 = (c1 + 15) / 16, nch = nch0 + nch1;
    const int cp = round_up(cout, 32);
    const size_t total = (size_t)9 * nch * cp * 48;
    if (!out) return total;
    memset(out, 0, total * sizeof(unsigned short));
    const int cin = c0 + c1;
    std::vector<float> S(cp, 1.f);
    for (int co = 0; co < cp; ++co) inv[co] = 1.f;
    for (int co = 0; co < cout; ++co) {
        float m = 0.f;
        for (size_t i = 0; i < (size_t)cin * 9; ++i) {
            float v = fabsf(w[(size_t)co * cin * 9 + i]);
            if (fold_scale) v = fabsf(w[(size_t)co * cin * 9 + i] * fold_scale[co]);
            if (v == v && v <= 3.0e38f && v > m) m = v;
        }
        if (m > 0.f) {
            int e;
            frexpf(m, &e);  // m = f 2^e, f in [0.5, 1): m in [2^(e-1), 2^e)
            int sh = 15 - e;
            sh = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
            S[co] = ldexpf(1.f, sh);
            inv[co] = ldexpf(1.f, -sh);
        }
    }
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
                    v *= S[co];  // exact (power of two) unless the weight is subnormal-small
                    unsigned short* o = out + (((size_t)tap * nch + c) * cp + (co & ~31)) * 48 + ((k >> 3) * 32 + (co & 31)) * 8 + (k & 7);
                    const _Float16 h = (_Float16)v;
                    const _Float16 h2 = (_Float16)((float)h * F16S_LO_UNSCALE);
                    const _Float16 l = (_Float16)(v - (float)h);
                    memcpy(o, &h, 2);
                    memcpy(o + 512, &h2, 2);
                    memcpy(o + 1024, &l, 2);
                }
    return total;
}

template <int WC, int WR, int TC, int TR>
static long long f16a_blocks(const ConvParams& p) {
    constexpr int TH = WR * TR, BN = WC * TC * 32;
    if (p.wf16_cout_pad % BN != 0) return 0;
    return (long long)p.N * ((p.Ho + TH - 1) / TH) * ((p.Wo + 31) / 32) * (p.wf16_cout_pad / BN);
}

template <int WC, int WR, int TC, int TR>
static int launch_f16a_cfg(const ConvParams& p, hipStream_t stream, int cfg_id) {
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
    hipLaunchKernelGGL((conv_win_f16a_kernel<WC, WR, TC, TR>), grid, dim3(256), 0, stream, p);
    DFVO_HIP_CHECK(hipGetLastError());
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventRecord(pe.e1, stream));
        pe.flops = p.useful_flops;
        const int sh[12] = {p.N, p.H, p.W, p.Ho, p.Wo, (p.G0 + p.G1) * 4, p.cout, p.kh, p.stride, (int)grid.x, (int)grid.y, 4};
        for (int i = 0; i < 12; ++i) pe.shape[i] = sh[i];
        g_prof->push_back(pe);
    }
    return DFVO_OK;
}

// Tile choice at two workgroups per CU (512 slots): rows per tile that minimise slot-rounds x rows.  Returns
// F16S2_NOT_APPLICABLE for the grids the first skeleton keeps (fewer than ~200 tiles of the smallest shape).
static int launch_f16a(const ConvParams& p, hipStream_t stream, int cfg_id) {
    if (!p.wf16a || !p.wf16a_inv) return F16S2_NOT_APPLICABLE;
    const int slots = 512;
    auto cost = [&](long long blocks, int rows) { return blocks <= 0 ? (1LL << 60) : ((blocks + slots - 1) / slots) * rows; };
    if (p.wf16_cout_pad % 128 == 0) {
        const long long b3 = f16a_blocks<2, 2, 2, 3>(p), b2 = f16a_blocks<2, 2, 2, 2>(p);
        if (b2 < 200) return F16S2_NOT_APPLICABLE;
        if (cost(b3, 6) <= cost(b2, 4) * 11 / 10) return launch_f16a_cfg<2, 2, 2, 3>(p, stream, cfg_id);
        return launch_f16a_cfg<2, 2, 2, 2>(p, stream, cfg_id);
    }
    if (p.wf16_cout_pad % 64 == 0) {
        const long long b3 = f16a_blocks<1, 4, 2, 3>(p), b2 = f16a_blocks<1, 4, 2, 2>(p);
        if (b2 < 200) return F16S2_NOT_APPLICABLE;
        if (cost(b3, 12) <= cost(b2, 8)) return launch_f16a_cfg<1, 4, 2, 3>(p, stream, cfg_id);
        return launch_f16a_cfg<1, 4, 2, 2>(p, stream, cfg_id);
    }
    const long long b3 = f16a_blocks<1, 4, 1, 3>(p), b2 = f16a_blocks<1, 4, 1, 2>(p);
    if (b2 < 200) return F16S2_NOT_APPLICABLE;
    if (cost(b3, 12) <= cost(b2, 8)) return launch_f16a_cfg<1, 4, 1, 3>(p, stream, cfg_id);
    return launch_f16a_cfg<1, 4, 1, 2>(p, stream, cfg_id);
}
