// Generic implicit-GEMM convolution with fp32-class accuracy on the f16 matrix cores (DFVO_CONV_PRECISION=f16x3): every
// layer the LDS-window kernel (conv_win_f16s.h) does not take -- the small maps (pyramid levels 5 / 6, the depth net's inner
// layers: M of a few hundred to a few thousand pixels, where a launch is a latency chain, not a throughput problem), and the
// layers that are not 3x3 / stride 1 (1x1, k x 1 / 1 x k, stride 2, the 7x7 / cin 3 first layers: HBM-streaming layers on
// the large maps).  Same arithmetic as conv_win_f16s.h: x = hi + 2^-11 lo in two f16 planes, three exact products per
// term on v_mfma_f32_32x32x16_f16, fp32 accumulate in a "main" and a "cross" set, combined as main + 2^-11 cross.
//
// Decomposition (wave64): a wave owns a 32-pixel x (32 TC)-cout block of the output and a SLICE of the K range; the KSP
// waves of a workgroup that share a block add their partial blocks through LDS in slice order (deterministic: no atomics,
// no second launch, no cross-XCD hand-off -- the split-K of the fp32 kernels costs 20-40 us per small layer in exactly
// those).  WP pixel blocks per workgroup, so a workgroup is WP x KSP waves.
// K is walked in steps of four k-groups (a k-group = 4 consecutive channels of one tap of one source: the unit of the
// NHWC gather, as in conv_igemm_f32_kernel, same order g = tap (G0 + G1) + channel group).  There is no LDS staging: a
// lane of the B fragment (pixel = lane & 31, k-block = lane >> 5) needs exactly k-groups 2 kb and 2 kb + 1 of its pixel
// -- two 16-byte loads straight from the activation tensor -- and the A fragment (32 couts x 16 k, both planes) is one
// 1 KB-contiguous load per plane from the packed weights; both are requested PF steps ahead into a register ring, so a
// wave has PF x (2 + 2 TC) loads in flight and never waits for a barrier inside its K loop.  The (tap, source, channel
// offset) of the four k-groups of a step come from a table built once per layer on the host and read through the scalar
// cache (one s_load_dwordx4 per step).
#pragma once
// (included inside namespace dfvo, after conv_win_f16s.h)

// k-group table entry: ky [0,5) | kx [5,10) | valid 10 | source 11 | channel offset inside the source << 16
static inline uint32_t f16g_entry(int ky, int kx, int valid, int src, int choff) {
    return (uint32_t)ky | ((uint32_t)kx << 5) | ((uint32_t)valid << 10) | ((uint32_t)src << 11) | ((uint32_t)choff << 16);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const u32x4 cu32x4;

// RAG (KSP = 1 shapes only): the layer's cout is not a multiple of the wave's cout tile (the 49- / 25-channel distance
// layers).  The shared epilogue sends such a wave through conv_epilogue_quad quad by quad -- a bias load, a wait, a store,
// and since stores count on vmcnt every wait drains the stores before it: 8-16 serialised round trips per wave, which is
// most of those layers' launch time.  The RAG instantiation loads every quad's bias first and then only stores.
// NP: products per term (3 = f16x3; 1 = "f16" mode: hi planes only, no cross set)
template <int WP, int KSP, int TC, int PF = 3, bool RAG = false, int NP = 3>  // PF: register ring, loads of step s + PF are issued when step s has been consumed
__global__ __launch_bounds__(64 * WP * KSP) void conv_gemm_f16s_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float f16g_red[];  // [WP][KSP][TC][16][64] partial blocks (KSP > 1)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wp = wave / KSP, wk = wave % KSP;
    const int lp = lane & 31, kb = lane >> 5;
    const int M = p.N * p.Ho * p.Wo;
    // one-dimensional grid, XCD-aware: each XCD walks a contiguous run of logical ids, and the cout tile is the FASTEST
    // index inside a run -- the workgroups that read the same pixel block (one per cout tile) are neighbours on one XCD,
    // so the activations are fetched from HBM once and served to the other cout tiles by that XCD's L2
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
    // K slice of this wave
    const int S = p.f16g_steps;
    const int per = (S + KSP - 1) / KSP;
    const int s0 = wk * per < S ? wk * per : S;
    const int s1 = s0 + per < S ? s0 + per : S;

    // Everything the gather needs from the parameter block lives in SGPRs for the whole kernel.  (Left to itself the
    // compiler turns `s1 ? p.src1 : p.src0` into a per-lane select of ADDRESSES INSIDE THE KERNARG SEGMENT and re-loads the
    // fields with vector loads at every step -- a dependent global load, and a vmcnt(0) that drains the prefetch ring,
    // in front of every activation load.)
    auto pin = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto pin64 = [](unsigned long long v) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    const cu32x4* const tab = (const cu32x4*)pin64((unsigned long long)p.f16g_tab);
    const unsigned long long src0 = pin64((unsigned long long)p.src0), src1 = pin64((unsigned long long)(p.G1 ? p.src1 : p.src0));
    const int cs0 = pin(p.cs0), cs1 = pin(p.cs1), co0 = pin(p.co0), co1 = pin(p.co1), up0 = pin(p.up0);
    const int H = pin(p.H), W = pin(p.W), refl = pin(p.pad_mode == PAD_REFLECT ? 1 : 0);
    const int H0 = H >> up0, W0 = W >> up0;
    const unsigned short* const wbase = p.wf16g + ((size_t)n0 * 32 + (kb * 32 + lp) * 8);
    const size_t w_step_stride = (size_t)pin(p.wf16g_cout_pad) * 32;  // halves per step

    f32x4 ra[PF][2];
    unsigned rav = 0;  // bit (2 stage + j): the load holds real data
    h16x8 rw[PF][TC][2];
    // Address arithmetic of the gather.  Counters on the small-map layers (profiles/r4r_pmc_small_maps.txt) show a wave
    // issuing ~150 vector instructions per 3-MFMA step, 40 % of its life -- most of them decoding the k-group table and
    // clamping coordinates per lane.  Without reflection padding and x2 upsampling a k-group's address is LINEAR in the
    // lane's pixel: base(pixel, source) + delta(group), delta = (ky W + kx) cs + channel offset.  The deltas are computed
    // once per workgroup into an LDS table [S][4] x {delta, flags = ky | kx << 5 | valid << 10 | source << 11}; a step reads
    // its two groups' entries with one 16-byte LDS load, one step ahead of their use.  The padded / upsampled layers (the
    // depth decoder) keep the per-step decode of the scalar table.
    const bool fast_addr = !refl && up0 == 0;  // wave-uniform
    int* const gtab = reinterpret_cast<int*>(f16g_red + (KSP > 1 ? WP * KSP * TC * 16 * 64 : 0));
    if (fast_addr) {
        for (int i = t; i < S * 4; i += 64 * WP * KSP) {
            const unsigned e = p.f16g_tab[i];
            const int ky = e & 31, kx = (e >> 5) & 31;
            const bool s1e = ((e >> 11) & 1u) != 0;
            gtab[2 * i] = (ky * W + kx) * (s1e ? cs1 : cs0) + (int)(e >> 16);
            gtab[2 * i + 1] = (int)(e & 0xfffu);
        }
        __syncthreads();
    }
    const int pbase0 = ((nimg * H + iy0) * W + ix0) * cs0 + co0, pbase1 = ((nimg * H + iy0) * W + ix0) * cs1 + co1;
    // table entries are fetched one load_step ahead (scalar cache latency / LDS latency off the address path); load_step is
    // called for consecutive steps s0, s0 + 1, ...
    int nl = s0;
    u32x4 tq = tab[nl < S ? nl : S - 1];
    u32x4 gq = {0u, 0u, 0u, 0u};
    if (fast_addr) gq = *reinterpret_cast<const u32x4*>(gtab + ((nl < S ? nl : S - 1) * 4 + 2 * kb) * 2);
    auto load_step = [&](int st) {
        const int sN = nl + 1 < S ? nl + 1 : S - 1;
        if (fast_addr) {
            const u32x4 gc = gq;
            gq = *reinterpret_cast<const u32x4*>(gtab + (sN * 4 + 2 * kb) * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int d = (int)gc[2 * j];
                const unsigned f = gc[2 * j + 1];
                const int iy = iy0 + (int)(f & 31u), ix = ix0 + (int)((f >> 5) & 31u);
                const bool s1v = ((f >> 11) & 1u) != 0;
                const bool v = vm && ((f >> 10) & 1u) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                // (masked lanes read the first channels of the source: always inside the tensor)
                const int off = v ? (s1v ? pbase1 : pbase0) + d : (s1v ? co1 : co0);
                const unsigned long long base = s1v ? src1 : src0;
                ra[st][j] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(base + (long long)off * 4);
                rav = (rav & ~(1u << (2 * st + j))) | ((v ? 1u : 0u) << (2 * st + j));
            }
        } else {
        const unsigned e0 = __builtin_amdgcn_readfirstlane(tq[0]), e1 = __builtin_amdgcn_readfirstlane(tq[1]),
                       e2 = __builtin_amdgcn_readfirstlane(tq[2]), e3 = __builtin_amdgcn_readfirstlane(tq[3]);
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
            // (invalid table entries carry channel offset 0 of source 0: the address is always inside the tensor)
            const int pix0 = (nimg * H0 + (iy >> up0)) * W0 + (ix >> up0);
            const int pix1 = (nimg * H + iy) * W + ix;
            const int off = (s1v ? pix1 * cs1 + co1 : pix0 * cs0 + co0) + (int)(e >> 16);
            const unsigned long long base = s1v ? src1 : src0;
            // (address space 1: a pointer rebuilt from an integer would otherwise be FLAT, whose loads count on both
            // wait counters and force a full drain at every use)
            ra[st][j] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(base + (long long)off * 4);
            rav = (rav & ~(1u << (2 * st + j))) | ((v ? 1u : 0u) << (2 * st + j));
        }
        }
        const unsigned short* g = wbase + (size_t)nl * w_step_stride;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            rw[st][i][0] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 1024);
            if constexpr (NP == 3) rw[st][i][1] = *reinterpret_cast<const h16x8*>(g + (size_t)i * 1024 + 512);
        }
        ++nl;
    };

    f32x16 am[TC], ax[TC];  // (ax stays zero in the NP == 1 instantiation: the combination below adds an exact 0)
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) am[i][e] = ax[i][e] = 0.f;
    float amax = 0.f;

#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (s0 + d < s1) load_step(d);
    for (int s = s0; s < s1; s += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (s + u < s1) {
                h16x4 h0, l0 = {}, h1, l1 = {};
                if constexpr (NP == 3) {
                    split_f16_planes(((rav >> (2 * u)) & 1u) ? ra[u][0] : f32x4{0.f, 0.f, 0.f, 0.f}, &h0, &l0, amax);
                    split_f16_planes(((rav >> (2 * u + 1)) & 1u) ? ra[u][1] : f32x4{0.f, 0.f, 0.f, 0.f}, &h1, &l1, amax);
                } else {
                    split_f16_hi(((rav >> (2 * u)) & 1u) ? ra[u][0] : f32x4{0.f, 0.f, 0.f, 0.f}, &h0, amax);
                    split_f16_hi(((rav >> (2 * u + 1)) & 1u) ? ra[u][1] : f32x4{0.f, 0.f, 0.f, 0.f}, &h1, amax);
                }
                const h16x8 xh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                const h16x8 xl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                h16x8 wh[TC], wl[TC];
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    wh[i] = rw[u][i][0];
                    if constexpr (NP == 3) wl[i] = rw[u][i][1];
                }
                if (s + u + PF < s1) load_step(u);
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
    f16s_report_clamp(amax);

    // combine the two accumulator sets; with KSP > 1 the partial blocks of a pixel block meet in LDS and are added in slice
    // order by the wave that finishes the register quad
    if (KSP == 1 && RAG) {  // (host: 16-byte aligned views, no residual, none / leaky / relu)
        if (!vm) return;
        float* const drow = p.dst + (size_t)m * p.dst_cs + p.dst_co;
        const float slope = p.act == ACT_LEAKY ? p.act_param : 1.f;
        f32x4 bq[4 * TC];
#pragma unroll
        for (int q = 0; q < 4 * TC; ++q) {
            const int col0 = n0 + (q >> 2) * 32 + 8 * (q & 3) + 4 * kb;
            bq[q] = *reinterpret_cast<const f32x4*>(p.bias + (col0 + 3 < p.cout_pad ? col0 : p.cout_pad - 4));  // always a valid address
        }
#pragma unroll
        for (int q = 0; q < 4 * TC; ++q) {
            const int col0 = n0 + (q >> 2) * 32 + 8 * (q & 3) + 4 * kb;
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
        return;
    }
    if (KSP == 1) {
        ConvEpi<4 * TC> epi;
        conv_epi_init(p, epi, [&](int q) { return n0 + (q >> 2) * 32 + 8 * (q & 3) + 4 * kb; });
        conv_epi_row(p, epi, (size_t)(vm ? m : 0), vm, [&](int q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = am[q >> 2][4 * (q & 3) + e] + F16S_LO_UNSCALE * ax[q >> 2][4 * (q & 3) + e];
            return v;
        });
        return;
    }
    float* const mine = f16g_red + (size_t)((wp * KSP + wk) * TC) * 16 * 64;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[(i * 16 + e) * 64 + lane] = am[i][e] + F16S_LO_UNSCALE * ax[i][e];
    __syncthreads();
    const bool vec_ok = conv_vec_ok(p);
    for (int q = wk; q < 4 * TC; q += KSP) {
        const int i = q >> 2, g = q & 3;
        const int col0 = n0 + i * 32 + 8 * g + 4 * kb;
        // (the quad's bias / residual requested before the LDS sum, consumed after it)
        const bool fastq = vec_ok && col0 + 3 < p.cout;
        f32x4 b = {0.f, 0.f, 0.f, 0.f}, r = {0.f, 0.f, 0.f, 0.f};
        if (fastq) {
            b = *reinterpret_cast<const f32x4*>(p.bias + col0);
            if (p.res && vm) r = *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.res_cs + p.res_co + col0);
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < KSP; ++z) {
            const float* src = f16g_red + (size_t)((wp * KSP + z) * TC + i) * 16 * 64 + (4 * g) * 64 + lane;
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
// weights as f16 hi / lo planes in k-group order: [step][cout_pad / 32][plane][k-block (2)][32 couts][8 k], where the 16 k
// of a step are k-groups 4 step .. 4 step + 3 (g = tap (G0 + G1) + channel group), 4 channels each.  Returns the number
// of halves (out may be null)
size_t conv_pack_weights_f16g(const float* w, int cout, int c0, int c1, int kh, int kw, const float* fold_scale,
                              unsigned short* out) {
    const int G0 = cdiv(c0, 4), G1 = cdiv(c1, 4), G = G0 + G1, taps = kh * kw;
    const int steps = cdiv(taps * G, 4);
    const int cp = round_up(cout, 32);
    const size_t total = (size_t)steps * cp * 32;
    if (!out) return total;
    memset(out, 0, total * sizeof(unsigned short));
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
                    unsigned short* o = out + ((size_t)step * cp + (co & ~31)) * 32 + ((k >> 3) * 32 + (co & 31)) * 8 + (k & 7);
                    f16s_split_host(v, o, o + 512);
                }
            }
        }
    return total;
}

// the k-group table of a layer: 4 entries per step
void conv_build_f16g_table(int c0, int c1, int kh, int kw, std::vector<uint32_t>* tab) {
    const int G0 = cdiv(c0, 4), G1 = cdiv(c1, 4), G = G0 + G1, taps = kh * kw;
    const int steps = cdiv(taps * G, 4);
    tab->assign((size_t)steps * 4, f16g_entry(0, 0, 0, 0, 0));
    for (int tap = 0; tap < taps; ++tap)
        for (int cg = 0; cg < G; ++cg)
            (*tab)[(size_t)tap * G + cg] = f16g_entry(tap / kw, tap % kw, 1, cg < G0 ? 0 : 1, cg < G0 ? cg * 4 : (cg - G0) * 4);
}

static bool conv_f16g_ok(const ConvParams& p) {
    if (!p.wf16g || !p.f16g_tab) return false;
    if (p.kh > 31 || p.kw > 31) return false;
    return true;
}

template <int WP, int KSP, int TC, int PF = 3, bool RAG = false>
static int launch_f16g_cfg(const ConvParams& p, hipStream_t stream, int cfg_id) {
    const long long M = (long long)p.N * p.Ho * p.Wo;
    dim3 grid((unsigned)(((M + 32 * WP - 1) / (32 * WP)) * (p.wf16g_cout_pad / (32 * TC))), 1, 1);
    // + the delta table, only for the layers whose kernel reads it (fast_addr: no reflection padding, no x2 upsampling).  With
    // the 64 KB reduction buffer of the <1, 16, 1> / <1, 8, 2> shapes this exceeds 64 KB: fine on gfx950 (160 KB per
    // workgroup, the only target of this library; ensure_dyn_lds reports anything the device refuses)
    const bool fast_addr = p.pad_mode != PAD_REFLECT && p.up0 == 0;
    const size_t lds = (KSP > 1 ? (size_t)WP * KSP * TC * 16 * 64 * sizeof(float) : 0) + (fast_addr ? (size_t)p.f16g_steps * 32 : 0);
    if (lds > 48 * 1024)
        if (int rc_lds = ensure_dyn_lds(p.f16_terms == 1 ? (const void*)conv_gemm_f16s_kernel<WP, KSP, TC, PF, RAG, 1>
                                                         : (const void*)conv_gemm_f16s_kernel<WP, KSP, TC, PF, RAG, 3>, lds))
            return rc_lds;
    ConvProfEntry pe;
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventCreate(&pe.e0));
        DFVO_HIP_CHECK(hipEventCreate(&pe.e1));
        pe.cfg = cfg_id;
        DFVO_HIP_CHECK(hipEventRecord(pe.e0, stream));
    }
    if (p.f16_terms == 1)
        hipLaunchKernelGGL((conv_gemm_f16s_kernel<WP, KSP, TC, PF, RAG, 1>), grid, dim3(64 * WP * KSP), lds, stream, p);
    else
        hipLaunchKernelGGL((conv_gemm_f16s_kernel<WP, KSP, TC, PF, RAG, 3>), grid, dim3(64 * WP * KSP), lds, stream, p);
    DFVO_HIP_CHECK(hipGetLastError());
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventRecord(pe.e1, stream));
        pe.flops = p.useful_flops;
        const int sh[12] = {p.N, p.H, p.W, p.Ho, p.Wo, (p.G0 + p.G1) * 4, p.cout, p.kh, p.stride, (int)grid.x, TC, KSP * 100 + 1};
        for (int i = 0; i < 12; ++i) pe.shape[i] = sh[i];
        g_prof->push_back(pe);
    }
    return DFVO_OK;
}

// Shape choice.  TC = 2 (a pixel fragment, whose split costs the VALU work, feeds two cout blocks) whenever the layer has
// an even number of 32-cout blocks.  The K split fills the chip: ~2 waves per SIMD in total, at least 4 steps per wave.
// Large maps (thousands of pixel blocks) run KSP = 1 with four pixel blocks per workgroup.
// multi-tap streaming layers: conv_taps_f16s.hip (its own translation unit; bit-identical to the <4, 1, TC> shape here)
bool conv_taps_f16s_ok(const ConvParams& p);
int launch_taps_f16s(const ConvParams& p, hipStream_t stream, int* grid_x);
static int launch_taps_prof(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    p.f16s_clamp_ctr = f16s_clamp_counter();
    ConvProfEntry pe;
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventCreate(&pe.e0));
        DFVO_HIP_CHECK(hipEventCreate(&pe.e1));
        pe.cfg = 20;
        DFVO_HIP_CHECK(hipEventRecord(pe.e0, stream));
    }
    int gx = 0;
    const int rc = launch_taps_f16s(p, stream, &gx);
    if (rc != DFVO_OK) return rc;
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventRecord(pe.e1, stream));
        pe.flops = p.useful_flops;
        const int sh[12] = {p.N, p.H, p.W, p.Ho, p.Wo, (p.G0 + p.G1) * 4, p.cout, p.kh, p.stride, gx, 0, 7};
        for (int i = 0; i < 12; ++i) pe.shape[i] = sh[i];
        g_prof->push_back(pe);
    }
    return DFVO_OK;
}

static int launch_f16g(const ConvParams& p, hipStream_t stream) {
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const long long mblocks = (M + 31) / 32;
    const int nblk = p.wf16g_cout_pad / 32;
    const long long target = 2048;  // waves on the chip: ~2 per SIMD
    // two cout blocks per wave (a pixel fragment, whose split costs the VALU work, feeds both) only while that still leaves
    // enough (pixel block, cout pair) tiles to fill the chip with at most eight K slices; tiny maps with many couts (the
    // depth net's 6 x 20 / 12 x 40 layers: the launch is one pass over megabytes of weights) take one block per wave and up
    // to 16 slices -- more waves, more loads in flight
    bool tc2 = (nblk % 2) == 0;
    if (tc2 && mblocks * (nblk / 2) * 8 * 2 < target && p.f16g_steps >= 64) tc2 = false;
    const long long tiles = mblocks * (tc2 ? nblk / 2 : nblk);
    int ksp = 1;
    // (16 slices only with one cout block per wave: a 1024-thread workgroup leaves 128 registers per lane)
    while (ksp < (tc2 ? 8 : 16) && tiles * ksp * 2 <= target && p.f16g_steps >= 4 * ksp * 2) ksp *= 2;
    // Measured and dropped (records under profiles/): an eight-stage ring with twice the waves (r3: K-sliced layers 1.24 vs
    // 1.10 ms per pair), K divided over workgroups as well (r3af_nz_ab.txt: launches 7 % shorter one at a time, pair rate
    // -1.3 %), one cout block per wave on the streaming shapes (r3j_stream_tc1_ab.txt: no gain).
    const int cfg = ksp == 1 ? 20 : 21;  // profile rows: 20 streaming (KSP = 1), 21 K-sliced small maps
    // multi-tap layers among the streaming shapes: the tap-window kernel (only where this kernel would not slice K: the two
    // then sum in the same order).  DFVO_TAPS=0 (test hook, tests/test_nets_gpu.py): this file's <4, 1, TC> shape instead
    // -- the form the tap-window kernel is compared with bit for bit
    static const bool taps_on = !(getenv("DFVO_TAPS") && atoi(getenv("DFVO_TAPS")) == 0);
    if (taps_on && ksp == 1 && conv_taps_f16s_ok(p)) return launch_taps_prof(p, stream);
    // ragged cout on a streaming shape: the store-only epilogue (profiles/r4j_rag_ab.txt: the 7 x 1 / 1 x 7 distance layers
    // 80 -> 68 / 104 -> 98 us, +0.8 % pairs/s, bit-identical)
    const bool rag = ksp == 1 && (p.cout % (tc2 ? 64 : 32)) != 0 && !p.res && ((p.dst_cs | p.dst_co) & 3) == 0 &&
                     p.cout_pad >= 4 && (p.act == ACT_NONE || p.act == ACT_LEAKY || p.act == ACT_RELU);
    if (rag) return tc2 ? launch_f16g_cfg<4, 1, 2, 3, true>(p, stream, cfg) : launch_f16g_cfg<4, 1, 1, 3, true>(p, stream, cfg);
    if (tc2) {
        switch (ksp) {
            case 1: return launch_f16g_cfg<4, 1, 2>(p, stream, cfg);
            case 2: return launch_f16g_cfg<2, 2, 2>(p, stream, cfg);
            case 4: return launch_f16g_cfg<1, 4, 2>(p, stream, cfg);
            default: return launch_f16g_cfg<1, 8, 2>(p, stream, cfg);
        }
    }
    switch (ksp) {
        case 1: return launch_f16g_cfg<4, 1, 1>(p, stream, cfg);
        case 2: return launch_f16g_cfg<2, 2, 1>(p, stream, cfg);
        case 4: return launch_f16g_cfg<1, 4, 1>(p, stream, cfg);
        case 8: return launch_f16g_cfg<1, 8, 1>(p, stream, cfg);
        default: return launch_f16g_cfg<1, 16, 1>(p, stream, cfg);
    }
}
