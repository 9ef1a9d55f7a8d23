// Implicit-GEMM convolution on the gfx950 matrix cores, exact fp32 numerics.
//
// Replaces the torch.nn.Conv2d (+bias, +LeakyReLU/ReLU/ELU/sigmoid, +residual, +BatchNorm(eval),
// +ReflectionPad2d, +nearest x2 upsample, +channel concat) call sites of the reference's nets:
//   LiteFlowNet  /root/reference/libs/deep_models/flow/lite_flow_net/lite_flow_net.py:39-75,98-101,
//                121-129,164-179,204-240
//   monodepth2   /root/reference/libs/deep_models/depth/monodepth2/resnet_encoder.py:87-98,
//                depth_decoder.py:29-65, layers.py:106-136,347-350
//
// GEMM view: M = N*Ho*Wo output pixels, N = cout, K = kh*kw*(c0+c1) walked in "k-groups" of 4
// channels (one 16-byte NHWC load) and K-steps of 4 groups.  The contraction runs on
// v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: bit-for-bit an fmaf chain, 157 TF peak).
// A lane (i = lane&15, kq = lane>>4) holds 4 consecutive channels of k-group kq for pixel i in one
// float4; MFMA #r of a K-step consumes component r of the A and of the B fragment, so the four
// MFMAs together cover the 16 k-values of the step (the k permutation is the same on both sides).
//
// Tiles: a 256-thread block (4 wave64) computes BM x BN = (WM*TM*16) x (WN*TN*16); A (pixels x 16 k)
// and B (16 k x couts) tiles are staged global -> registers -> LDS with the next step's loads in
// flight under the current step's MFMAs.  The LDS A image is XOR-swizzled on the k-group so the four
// 16-lane service groups of ds_read_b128 each hit 16 distinct 16-byte slots.
#include "dfvo_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace dfvo {

#include "conv_epi.h"

// Split-K finish inside the contracting kernel.  After a workgroup has written its partial tile to the workspace it draws
// a ticket for that tile; whoever draws the last one (all z-slices are then written) reduces the partials and runs the
// epilogue.  The sum runs over the slices in slice order -- the order of conv_splitk_epilogue -- so the result does not
// depend on which workgroup happens to be last.
// The z-slices of a tile run on different XCDs, whose L2s are not coherent with each other.  A release / acquire fence
// pair would be the textbook handoff, but at device scope it writes back and invalidates the whole L2 -- measured: the
// nets 2x slower, every concurrent kernel pays.  Instead the partials themselves travel with device-scope relaxed atomic
// stores / loads (write-through, L2-bypassing accesses), the writer drains them (vmcnt 0) before its ticket, and the
// ticket is a relaxed device-scope RMW.  The winner leaves the counter at zero for the next launch.
__device__ __forceinline__ void splitk_store(float* dst, f32x4 v) {
    union {
        float f[4];
        unsigned long long u[2];
    } c;
#pragma unroll
    for (int e = 0; e < 4; ++e) c.f[e] = v[e];
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), c.u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + 1, c.u[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 splitk_load(const float* src) {
    union {
        float f[4];
        unsigned long long u[2];
    } c;
    c.u[0] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.u[1] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f32x4{c.f[0], c.f[1], c.f[2], c.f[3]};
}
// Every thread of the workgroup must call this, after its splitk_store()s.
__device__ __forceinline__ bool splitk_last_arriver(const ConvParams& p, unsigned tile) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&p.tile_flags[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev + 1 == gridDim.z ? 1 : 0;
        if (last) __hip_atomic_store(&p.tile_flags[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}

// k-group table entry built once per block in LDS: which tap / source / channel offset a k-group is
struct KGroup {
    uint32_t tap;  // ky | kx << 8 | valid << 16 | src << 17
    uint32_t coff; // float offset of the 4 channels inside the pixel (view offset included)
};
constexpr int MAX_KGROUPS = 2048;

// Branch-free A gather: the address is always clamped into the tensor and the load always issued;
// out-of-image / padded / out-of-range lanes are zeroed by a select when the tile is written to LDS.
// (Per-lane branches around the loads fragment the K-loop into exec-masked blocks and serialise it.)
__device__ __forceinline__ f32x4 conv_load_a(const ConvParams& p, int n, int iy0, int ix0, bool vm, KGroup kg,
                                             bool& valid) {
    const int ky = kg.tap & 0xff, kx = (kg.tap >> 8) & 0xff;
    int iy = iy0 + ky, ix = ix0 + kx;
    bool v = vm && ((kg.tap >> 16) & 1);
    if (p.pad_mode == PAD_REFLECT) {  // wave-uniform
        iy = reflect_idx(iy, p.H);
        ix = reflect_idx(ix, p.W);
    } else {
        v = v && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    }
    iy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
    ix = ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix);
    const bool s1 = ((kg.tap >> 17) & 1) != 0;
    const int sh = s1 ? 0 : p.up0;
    const float* base = s1 ? p.src1 : p.src0;
    const unsigned cs = (unsigned)(s1 ? p.cs1 : p.cs0);
    const unsigned yy = (unsigned)(iy >> sh), xx = (unsigned)(ix >> sh);
    const unsigned HH = (unsigned)(p.H >> sh), WW = (unsigned)(p.W >> sh);
    const unsigned off = (((unsigned)n * HH + yy) * WW + xx) * cs + kg.coff;
    valid = v;
    return *reinterpret_cast<const f32x4*>(base + off);
}

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvParams p) {
    constexpr int BM = WM * TM * 16;
    constexpr int BN = WN * TN * 16;
    constexpr int A_CNT = (BM * 4 + 255) / 256;
    constexpr int B_CNT = (BN * 4 + 255) / 256;
    constexpr int TILE = BM * 16 + BN * 16;  // floats per stage
    static_assert(WM * WN == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
    __shared__ KGroup ktab[MAX_KGROUPS];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;

    // XCD-aware tile order: consecutive block ids land on different XCDs (id % 8), so give each XCD
    // a contiguous run of M-tiles (neighbouring tiles share conv halo rows in that XCD's L2).
    const int nb = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int m0 = bid * BM;
    const int n0 = blockIdx.y * BN;
    const int M = p.N * p.Ho * p.Wo;
    const int G = p.G0 + p.G1;
    const int taps = p.kh * p.kw;
    // split-K: this block contracts K-steps [s_begin, s_end)
    const int per = (p.ksteps + (int)gridDim.z - 1) / (int)gridDim.z;
    const int s_begin = (int)blockIdx.z * per;
    const int s_end = s_begin + per < p.ksteps ? s_begin + per : p.ksteps;

    for (int g = s_begin * 4 + t; g < s_end * 4; g += 256) {
        const int tap = g / G;
        const int cg = g - tap * G;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        KGroup e;
        const uint32_t valid = tap < taps ? 1u : 0u;
        const uint32_t src = cg < p.G0 ? 0u : 1u;
        e.tap = (uint32_t)ky | ((uint32_t)kx << 8) | (valid << 16) | (src << 17);
        e.coff = src == 0 ? (uint32_t)(p.co0 + cg * 4) : (uint32_t)(p.co1 + (cg - p.G0) * 4);
        ktab[g - s_begin * 4] = e;
    }

    // per-thread A staging slots
    int a_n[A_CNT], a_iy0[A_CNT], a_ix0[A_CNT];
    bool a_vm[A_CNT];
#pragma unroll
    for (int r = 0; r < A_CNT; ++r) {
        const int id = t + 256 * r;
        const int m = m0 + (id >> 2);
        const bool vm = (id < BM * 4) && (m < M);
        const int mm = vm ? m : 0;
        const int n = mm / (p.Ho * p.Wo);
        const int rem = mm - n * (p.Ho * p.Wo);
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_n[r] = n;
        a_iy0[r] = oy * p.stride - p.pad_h;
        a_ix0[r] = ox * p.stride - p.pad_w;
        a_vm[r] = vm;
    }
    __syncthreads();

    f32x4 ra[A_CNT], rb[B_CNT];
    bool rv[A_CNT];
    auto load_step = [&](int s) {
        const KGroup kg = ktab[(s - s_begin) * 4 + (t & 3)];
#pragma unroll
        for (int r = 0; r < A_CNT; ++r) ra[r] = conv_load_a(p, a_n[r], a_iy0[r], a_ix0[r], a_vm[r], kg, rv[r]);
#pragma unroll
        for (int r = 0; r < B_CNT; ++r) {
            const int id = t + 256 * r;
            if (id < BN * 4) {
                const int gi = id / BN, j = id - gi * BN;
                rb[r] = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)(4 * s + gi) * p.cout_pad + n0 + j) * 4);
            }
        }
    };
    auto store_step = [&](float* As, float* Bs) {
#pragma unroll
        for (int r = 0; r < A_CNT; ++r) {
            const int id = t + 256 * r;
            if (id < BM * 4) {
                const int m = id >> 2, gi = id & 3;
                const int gs = gi ^ (((m >> 3) & 1) << 1);
                *reinterpret_cast<f32x4*>(As + m * 16 + gs * 4) = rv[r] ? ra[r] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int r = 0; r < B_CNT; ++r) {
            const int id = t + 256 * r;
            if (id < BN * 4) *reinterpret_cast<f32x4*>(Bs + id * 4) = rb[r];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int li = lane & 15, kq = lane >> 4;
    if (s_begin < s_end) {
        load_step(s_begin);
        store_step(lds, lds + BM * 16);
    }
    __syncthreads();
    // double-buffered LDS, ONE barrier per K-step: while step s is contracted out of stage (s&1) the
    // tiles of step s+1 travel global -> registers -> stage ((s+1)&1)
    for (int s = s_begin; s < s_end; ++s) {
        const int st = (s - s_begin) & 1;
        const float* As = lds + st * TILE;
        const float* Bs = As + BM * 16;
        const bool more = (s + 1 < s_end);
        if (more) load_step(s + 1);
        f32x4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = wm * TM * 16 + i * 16 + li;
            const int gs = kq ^ (((m >> 3) & 1) << 1);
            fa[i] = *reinterpret_cast<const f32x4*>(As + m * 16 + gs * 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = wn * TN * 16 + j * 16 + li;
            fb[j] = *reinterpret_cast<const f32x4*>(Bs + (kq * BN + c) * 4);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][r], fa[i][r], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (more) {
            float* An = lds + (st ^ 1) * TILE;
            store_step(An, An + BM * 16);
        }
        __syncthreads();
    }

    // epilogue: D = W^T-fragment x pixels, so lane (li, kq) holds couts kq*4 .. kq*4+3 of pixel li
    if (gridDim.z > 1) {  // split-K partial: raw accumulators to the workspace [z][M][cout_pad]
        float* wsz = p.ws + (size_t)blockIdx.z * M * p.cout_pad;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col0 = n0 + wn * TN * 16 + j * 16 + kq * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * TM * 16 + i * 16 + li;
                if (m >= M) continue;
                if (p.tile_flags)
                    splitk_store(wsz + (size_t)m * p.cout_pad + col0, acc[i][j]);
                else
                    *reinterpret_cast<f32x4*>(wsz + (size_t)m * p.cout_pad + col0) = acc[i][j];
            }
        }
        if (!p.tile_flags) return;  // reduced by conv_splitk_epilogue
        if (!splitk_last_arriver(p, blockIdx.y * gridDim.x + blockIdx.x)) return;
        // slice by slice, all fragments of the thread per slice: the TM x TN loads of a slice are in flight together (they
        // bypass the L2: microseconds each), every element still accumulates its slices in ascending order
        const size_t zs = (size_t)M * p.cout_pad;
        for (int z = 0; z < (int)gridDim.z; ++z) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col0 = n0 + wn * TN * 16 + j * 16 + kq * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = m0 + wm * TM * 16 + i * 16 + li;
                    const f32x4 v = splitk_load(p.ws + z * zs + (size_t)(m < M ? m : 0) * p.cout_pad + col0);
                    acc[i][j] = z == 0 ? v : acc[i][j] + v;
                }
            }
        }
    }
    ConvEpi<TN> epi;  // bias of the wave's couts loaded once, a pixel's stores back to back (see conv_epi_row)
    conv_epi_init(p, epi, [&](int j) { return n0 + wn * TN * 16 + j * 16 + kq * 4; });
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * TM * 16 + i * 16 + li;
        conv_epi_row(p, epi, (size_t)(m < M ? m : 0), m < M, [&](int j) { return acc[i][j]; });
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride-1 convolutions with an LDS-resident input window.
// The gather kernel above re-fetches every input pixel once per tap (9x for a 3x3), which overflows the
// per-XCD L2 on the large maps (profiles/r1k_pmc: 3.5x the algorithmic bytes at the fabric) and makes the
// narrow-N layers vL1D-bound.  Here a workgroup owns a TH x 16 output tile: the (TH+2) x 18 input window of a
// 16-channel chunk is loaded ONCE into LDS (padding, reflection, x2-upsample and the two-source concat resolved
// at that load) and the nine taps read their A fragments straight out of it at compile-time offsets, so a
// K-step (= one tap of the chunk: 4 k-groups) stages only its 16 x BN weight tile.  Window pixel stride is 20
// floats: the 16 lanes of a ds_read_b128 service group hit 16 distinct 4-bank groups.
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int KS>
__global__ __launch_bounds__(256) void conv_win_f32_kernel(const ConvParams p) {
    constexpr int TH = WM * TM, TW = 16, WH = TH + KS - 1, WW = TW + KS - 1, PS = 20;
    constexpr int BN = WN * TN * 16;
    constexpr int WIN = WH * WW * PS;  // floats per window buffer
    constexpr int BT = BN * 16;        // floats per weight stage
    constexpr int W_ITEMS = WH * WW * 4;
    constexpr int W_CNT = (W_ITEMS + 255) / 256;
    constexpr int B_CNT = (BN * 4 + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) float lds[2 * WIN + 2 * BT];
    float* const win = lds;
    float* const bst = lds + 2 * WIN;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, kq = lane >> 4;
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
    const int G = p.G0 + p.G1;
    const int nchunk0 = (p.G0 + 3) >> 2, nchunk1 = (p.G1 + 3) >> 2, nchunks = nchunk0 + nchunk1;
    // split-K over the channel chunks: this workgroup contracts chunks [c_begin, c_end)
    const int cper = (nchunks + (int)gridDim.z - 1) / (int)gridDim.z;
    const int c_begin = (int)blockIdx.z * cper;
    const int c_end = c_begin + cper < nchunks ? c_begin + cper : nchunks;

    // window items of this thread: (pixel, channel group within the chunk)
    int w_off0[W_CNT], w_off1[W_CNT], w_lds[W_CNT];
    bool w_ok[W_CNT];
#pragma unroll
    for (int r = 0; r < W_CNT; ++r) {
        const int id = t + 256 * r;
        const int px = id >> 2, q = id & 3;
        const int wy = px / WW, wx = px - wy * WW;
        int iy = ty0 - p.pad_h + wy, ix = tx0 - p.pad_w + wx;
        bool v = id < W_ITEMS;
        if (p.pad_mode == PAD_REFLECT) {
            iy = reflect_idx(iy, p.H);
            ix = reflect_idx(ix, p.W);
        }
        v = v && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        iy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
        ix = ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix);
        const int sh = p.up0;
        w_off0[r] = (((n * (p.H >> sh) + (iy >> sh)) * (p.W >> sh) + (ix >> sh)) * p.cs0) + p.co0 + q * 4;
        w_off1[r] = (((n * p.H + iy) * p.W + ix) * p.cs1) + p.co1 + q * 4;
        w_ok[r] = v;
        w_lds[r] = (px < WH * WW ? px : 0) * PS + q * 4;
    }
    f32x4 rw[W_CNT], rb[B_CNT];
    bool rwv[W_CNT];
    auto load_window = [&](int c) {
        const bool s1 = c >= nchunk0;
        const int cg0 = s1 ? (c - nchunk0) * 4 : c * 4;
        const int Gs = s1 ? p.G1 : p.G0;
        const float* base = s1 ? p.src1 : p.src0;
#pragma unroll
        for (int r = 0; r < W_CNT; ++r) {
            const int q = (t + 256 * r) & 3;
            const bool v = w_ok[r] && (cg0 + q) < Gs;
            const int off = (s1 ? w_off1[r] : w_off0[r]) + (v ? cg0 * 4 : -(q * 4));  // masked lanes re-read channel 0
            rw[r] = *reinterpret_cast<const f32x4*>(base + off);
            rwv[r] = v;
        }
    };
    auto store_window = [&](float* W) {
#pragma unroll
        for (int r = 0; r < W_CNT; ++r)
            if (t + 256 * r < W_ITEMS)
                *reinterpret_cast<f32x4*>(W + w_lds[r]) = rwv[r] ? rw[r] : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto wrow_of = [&](int c) { return c >= nchunk0 ? p.G0 + (c - nchunk0) * 4 : c * 4; };
    auto load_b = [&](int c, int tap) {
        const int g0 = tap * G + wrow_of(c);
#pragma unroll
        for (int r = 0; r < B_CNT; ++r) {
            const int id = t + 256 * r;
            if (id < BN * 4) {
                const int gi = id / BN, j = id - gi * BN;
                rb[r] = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)(g0 + gi) * p.cout_pad + n0 + j) * 4);
            }
        }
    };
    auto store_b = [&](float* Bs) {
#pragma unroll
        for (int r = 0; r < B_CNT; ++r) {
            const int id = t + 256 * r;
            if (id < BN * 4) *reinterpret_cast<f32x4*>(Bs + id * 4) = rb[r];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int TAPS = KS * KS;
    if (c_begin < c_end) {
        load_window(c_begin);
        load_b(c_begin, 0);
        store_window(win + (c_begin & 1) * WIN);
        store_b(bst);
    }
    __syncthreads();
    int stage = 0;
    for (int c = c_begin; c < c_end; ++c) {
        const float* Wc = win + (c & 1) * WIN;
        float* Wn = win + ((c + 1) & 1) * WIN;
        const bool next_chunk = c + 1 < c_end;
        if (next_chunk) load_window(c + 1);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
            const float* Bs = bst + stage * BT;
            const bool more = tap < TAPS - 1 || next_chunk;
            if (more) load_b(tap < TAPS - 1 ? c : c + 1, tap < TAPS - 1 ? tap + 1 : 0);
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * TM + i;
                fa[i] = *reinterpret_cast<const f32x4*>(Wc + ((row + ky) * WW + (li + kx)) * PS + kq * 4);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * TN * 16 + j * 16 + li;
                fb[j] = *reinterpret_cast<const f32x4*>(Bs + (kq * BN + col) * 4);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][r], fa[i][r], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (more) store_b(bst + (stage ^ 1) * BT);
            if (tap == TAPS / 2 && next_chunk) store_window(Wn);
            __syncthreads();
            stage ^= 1;
        }
    }

    // epilogue: lane (li, kq) holds couts kq*4 .. kq*4+3 of the pixel at x = tx0 + li of its tile rows
    const int ox = tx0 + li;
    if (gridDim.z > 1) {  // split-K partial: raw accumulators to the workspace [z][M][cout_pad]
        const size_t Mtot = (size_t)p.N * p.Ho * p.Wo;
        float* wsz = p.ws + (size_t)blockIdx.z * Mtot * p.cout_pad;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col0 = n0 + wn * TN * 16 + j * 16 + kq * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int oy = ty0 + wm * TM + i;
                if (oy >= p.Ho || ox >= p.Wo) continue;
                float* wd = wsz + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.cout_pad + col0;
                if (p.tile_flags)
                    splitk_store(wd, acc[i][j]);
                else
                    *reinterpret_cast<f32x4*>(wd) = acc[i][j];
            }
        }
        if (!p.tile_flags) return;  // reduced by conv_splitk_epilogue
        if (!splitk_last_arriver(p, blockIdx.y * gridDim.x + blockIdx.x)) return;
        const size_t zs = Mtot * p.cout_pad;
        const int oxc = ox < p.Wo ? ox : p.Wo - 1;
        for (int z = 0; z < (int)gridDim.z; ++z) {  // (same order as in conv_igemm_f32_kernel)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col0 = n0 + wn * TN * 16 + j * 16 + kq * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int oy = ty0 + wm * TM + i;
                    const int oyc = oy < p.Ho ? oy : p.Ho - 1;
                    const f32x4 v = splitk_load(p.ws + z * zs + (((size_t)n * p.Ho + oyc) * p.Wo + oxc) * p.cout_pad + col0);
                    acc[i][j] = z == 0 ? v : acc[i][j] + v;
                }
            }
        }
    }
    ConvEpi<TN> epi;  // bias of the wave's couts loaded once, a pixel's stores back to back (see conv_epi_row)
    conv_epi_init(p, epi, [&](int j) { return n0 + wn * TN * 16 + j * 16 + kq * 4; });
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int oy = ty0 + wm * TM + i;
        const bool valid = oy < p.Ho && ox < p.Wo;
        conv_epi_row(p, epi, valid ? ((size_t)n * p.Ho + oy) * p.Wo + ox : 0, valid, [&](int j) { return acc[i][j]; });
    }
}

// ------------------------------------------------------------------------------------------------

// split-K second pass: ordered sum of the partials + bias + residual + activation
__global__ void conv_splitk_epilogue(const ConvParams p, int splits) {
    const int M = p.N * p.Ho * p.Wo;
    const int cols = p.dst_zero_to > p.cout ? p.dst_zero_to : p.cout;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * cols) return;
    const int col = (int)(idx % cols);
    const int m = (int)(idx / cols);
    if (col >= p.cout) {
        p.dst[(size_t)m * p.dst_cs + p.dst_co + col] = 0.f;
        return;
    }
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += p.ws[((size_t)z * M + m) * p.cout_pad + col];
    v += p.bias[col];
    if (p.res) v += p.res[(size_t)m * p.res_cs + p.res_co + col];
    v = apply_act(v, p.act, p.act_param);
    p.dst[(size_t)m * p.dst_cs + p.dst_co + col] = v;
}

int conv_pick_bn(int cout, long long M) {
    (void)M;
    if (cout <= 16) return 16;
    if (cout <= 32) return 32;
    if (cout % 128 == 0) return 128;
    if (cout % 64 == 0) return 64;
    if (cout % 32 == 0) return 32;  // 96 -> 3 x 32
    return 64;                      // 49 -> 64
}

void conv_pack_weights(const float* w, const float* bias, int cout, int c0, int c1, int kh, int kw, int cout_pad,
                       const float* fold_scale, const float* fold_shift, float* out_w, float* out_b) {
    const int G0 = cdiv(c0, 4), G1 = cdiv(c1, 4), G = G0 + G1;
    const int taps = kh * kw;
    const int ksteps = cdiv(taps * G, 4);
    const int cin = c0 + c1;
    std::fill(out_w, out_w + (size_t)ksteps * 4 * cout_pad * 4, 0.f);
    for (int tap = 0; tap < taps; ++tap) {
        const int ky = tap / kw, kx = tap % kw;
        for (int cg = 0; cg < G; ++cg) {
            const int g = tap * G + cg;
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
                for (int o = 0; o < cout; ++o) {
                    float v = w[(((size_t)o * cin + ci) * kh + ky) * kw + kx];
                    if (fold_scale) v *= fold_scale[o];
                    out_w[((size_t)g * cout_pad + o) * 4 + q] = v;
                }
            }
        }
    }
    for (int o = 0; o < cout_pad; ++o) {
        float b = 0.f;
        if (o < cout) {
            b = bias ? bias[o] : 0.f;
            if (fold_scale) b *= fold_scale[o];
            if (fold_shift) b += fold_shift[o];
        }
        out_b[o] = b;
    }
}


size_t conv_head_weight_floats(int cout, int c0, int c1, int k) {
    return (size_t)(cdiv(c0, 8) + cdiv(c1, 8)) * k * 2 * k * cout * 4;
}
void conv_pack_head_weights(const float* w, int cout, int c0, int c1, int k, const float* fold_scale, float* out) {
    const int nchunk0 = cdiv(c0, 8), nchunks = nchunk0 + cdiv(c1, 8), cin = c0 + c1;
    std::fill(out, out + conv_head_weight_floats(cout, c0, c1, k), 0.f);
    for (int c = 0; c < nchunks; ++c)
        for (int kx = 0; kx < k; ++kx)
            for (int cg = 0; cg < 2; ++cg)
                for (int ky = 0; ky < k; ++ky)
                    for (int o = 0; o < cout; ++o)
                        for (int q = 0; q < 4; ++q) {
                            const int cl = (c < nchunk0 ? c : c - nchunk0) * 8 + cg * 4 + q;  // channel inside its source
                            if (cl >= (c < nchunk0 ? c0 : c1)) continue;
                            const int ci = c < nchunk0 ? cl : c0 + cl;
                            float v = w[(((size_t)o * cin + ci) * k + ky) * k + kx];
                            if (fold_scale) v *= fold_scale[o];
                            out[(((((size_t)c * k + kx) * 2 + cg) * k + ky) * cout + o) * 4 + q] = v;
                        }
}

// ------------------------------------------------------------------------------------------------
// Direct (VALU) convolution for the 1- and 2-channel heads: LiteFlowNet's flow outputs (7x7 / 5x5 / 3x3, 32 -> 2) and
// monodepth2's disparity outputs (3x3, C -> 1).  On the MFMA kernels these layers pad cout to a 16-wide tile and
// throw 7/8 (15/16) of the matrix work away (8.5 TFLOP/s "useful" on the 7x7 head).  Here a thread owns PY = 2
// vertically adjacent output pixels of one column: for every (kx, channel group) it reads the PY + KS - 1 window
// rows once from LDS and feeds all KS vertical taps of both pixels from registers (14 FMA quads per 8 LDS reads at
// 7x7), with the weights arriving through the scalar cache (the address is wave-uniform), so neither LDS nor VGPR
// bandwidth is spent on them.  Window: 8-channel chunks, pixel stride 12 floats (16 consecutive lanes of a
// ds_read_b128 hit 16 distinct 4-bank groups).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(4))) const f32x4 cf32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
// PY: output rows per thread (tile = 16 PY rows x 16 columns).  2 on the large maps; 1 (round 6) where the 32-row tiles leave most
// of the chip idle -- level 3: 105 workgroups, level 4: 26, levels 5 / 6: 7 / 2 --: twice the workgroups, half the tap chain per
// thread.  A pixel's sum runs over (chunk, kx, channel group, ky) in the same order either way: bit-identical.
template <int KS, int CO, int PY = 2>
__global__ __launch_bounds__(256) void conv_head_f32_kernel(const ConvParams p) {
    constexpr int TW = 16, TH = 16 * PY, WH = TH + KS - 1, WW = TW + KS - 1, PS = 12;
    constexpr int WIN = WH * WW * PS;
    constexpr int W_ITEMS = WH * WW * 2;  // (pixel, half of the 8-channel chunk)
    constexpr int W_CNT = (W_ITEMS + 255) / 256;
    constexpr int NR = PY + KS - 1;
    __shared__ __attribute__((aligned(16))) float win[WIN];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int nb = gridDim.x;
    int bid = blockIdx.x;
    {  // XCD-aware order, as in the window kernel
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int n = bid / (tiles_y * tiles_x);
    const int trem = bid - n * (tiles_y * tiles_x);
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
    const int nchunk0 = (p.G0 + 1) >> 1, nchunk1 = (p.G1 + 1) >> 1, nchunks = nchunk0 + nchunk1;

    int w_off0[W_CNT], w_off1[W_CNT], w_lds[W_CNT];
    bool w_ok[W_CNT];
#pragma unroll
    for (int r = 0; r < W_CNT; ++r) {
        const int id = t + 256 * r;
        const int px = id >> 1, q = id & 1;
        const int wy = px / WW, wx = px - wy * WW;
        int iy = ty0 - p.pad_h + wy, ix = tx0 - p.pad_w + wx;
        bool v = id < W_ITEMS;
        if (p.pad_mode == PAD_REFLECT) {
            iy = reflect_idx(iy, p.H);
            ix = reflect_idx(ix, p.W);
        }
        v = v && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        iy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
        ix = ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix);
        const int sh = p.up0;
        w_off0[r] = (((n * (p.H >> sh) + (iy >> sh)) * (p.W >> sh) + (ix >> sh)) * p.cs0) + p.co0 + q * 4;
        w_off1[r] = (((n * p.H + iy) * p.W + ix) * p.cs1) + p.co1 + q * 4;
        w_ok[r] = v;
        w_lds[r] = (px < WH * WW ? px : 0) * PS + q * 4;
    }
    const cf32x4* const wq = (const cf32x4*)(unsigned long long)p.wh;  // constant address space: scalar loads

    f32x2 acc[PY][CO];
#pragma unroll
    for (int y = 0; y < PY; ++y)
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[y][o] = f32x2{0.f, 0.f};

    f32x4 rw[W_CNT];
    auto load_window = [&](int c) {
        const bool s1 = c >= nchunk0;
        const int cg0 = s1 ? (c - nchunk0) * 2 : c * 2;
        const int Gs = s1 ? p.G1 : p.G0;
        const float* base = s1 ? p.src1 : p.src0;
#pragma unroll
        for (int r = 0; r < W_CNT; ++r) {
            const int q = (t + 256 * r) & 1;
            const bool v = w_ok[r] && (cg0 + q) < Gs;
            const int off = (s1 ? w_off1[r] : w_off0[r]) + (v ? cg0 * 4 : -(q * 4));  // masked lanes re-read channel 0
            const f32x4 x = *reinterpret_cast<const f32x4*>(base + off);
            rw[r] = v ? x : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // The tap loop is software-pipelined over steps (kx, channel group, half of the ky range): the weights of step
    // s + 1 (scalar loads) and the window rows of its (kx, group) are requested before the FMAs of step s issue, so
    // the scalar-cache / LDS latency of a step hides behind the arithmetic of the previous one.
    constexpr int HALVES = KS == 7 ? 2 : 1;          // 7x7: two steps per (kx, group), 2 x 32 weight SGPRs in flight
    constexpr int KH0 = (KS + HALVES - 1) / HALVES;  // ky range of half 0: [0, KH0), half 1: [KH0, KS)
    constexpr int STEPS = KS * 2 * HALVES;           // (kx, cg, half)
    struct Wts {
        f32x4 w[KH0][CO];
    };
    load_window(0);
    for (int c = 0; c < nchunks; ++c) {
        if (c > 0) __syncthreads();  // every lane is done with the previous chunk's window
#pragma unroll
        for (int r = 0; r < W_CNT; ++r)
            if (t + 256 * r < W_ITEMS) *reinterpret_cast<f32x4*>(win + w_lds[r]) = rw[r];
        __syncthreads();
        if (c + 1 < nchunks) load_window(c + 1);  // in flight during this chunk's taps
        const cf32x4* const wb = wq + (size_t)c * (KS * 2 * KS * CO);  // this chunk's weights, in step order
        auto load_w = [&](int step, Wts& W) {
            const int kc = step / HALVES, kx = kc >> 1, cg = kc & 1, half = step % HALVES;
#pragma unroll
            for (int k = 0; k < KH0; ++k) {
                const int ky = half * KH0 + k;
                if (ky < KS) {
#pragma unroll
                    for (int o = 0; o < CO; ++o) W.w[k][o] = wb[((kx * 2 + cg) * KS + ky) * CO + o];
                }
            }
        };
        auto load_a = [&](int step, f32x4* a) {
            const int kc = step / HALVES, kx = kc >> 1, cg = kc & 1;
#pragma unroll
            for (int r = 0; r < NR; ++r)
                a[r] = *reinterpret_cast<const f32x4*>(win + ((ty * PY + r) * WW + tx + kx) * PS + cg * 4);
        };
        Wts wbuf[2];       // ping-pong: step s reads wbuf[s & 1] while wbuf[(s + 1) & 1] fills
        f32x4 abuf[2][NR];  // window rows of the current / next (kx, group): index (s / HALVES) & 1
        load_w(0, wbuf[0]);
        load_a(0, abuf[0]);
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            const int half = step % HALVES;
            if (step + 1 < STEPS) {
                load_w(step + 1, wbuf[(step + 1) & 1]);
                if (half == HALVES - 1) load_a(step + 1, abuf[((step + 1) / HALVES) & 1]);  // next step: new (kx, group)
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch distance at one step (SGPR budget)
#pragma unroll
            for (int k = 0; k < KH0; ++k) {
                const int ky = half * KH0 + k;
                if (ky < KS) {
#pragma unroll
                    for (int o = 0; o < CO; ++o) {
                        // packed FMAs on the natural register pairs: (x0, x1) * (w0, w1), then (x2, x3) * (w2, w3)
                        const f32x4 w = wbuf[step & 1].w[k][o];
                        const f32x2 wlo = {w[0], w[1]}, whi = {w[2], w[3]};
#pragma unroll
                        for (int y = 0; y < PY; ++y) {
                            const f32x4 x = abuf[(step / HALVES) & 1][y + ky];
                            const f32x2 xlo = {x[0], x[1]}, xhi = {x[2], x[3]};
                            acc[y][o] = __builtin_elementwise_fma(xlo, wlo, acc[y][o]);
                            acc[y][o] = __builtin_elementwise_fma(xhi, whi, acc[y][o]);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const bool vec_ok = conv_vec_ok(p);
    const int ox = tx0 + tx;
#pragma unroll
    for (int y = 0; y < PY; ++y) {
        const int oy = ty0 + ty * PY + y;
        if (oy < p.Ho && ox < p.Wo) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < CO; ++o) v[o] = acc[y][o][0] + acc[y][o][1];
            conv_epilogue_quad(p, ((size_t)n * p.Ho + oy) * p.Wo + ox, 0, v, vec_ok);
        }
    }
}


// ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline leg)
struct ConvProfEntry {
    hipEvent_t e0, e1;
    int cfg;
    double flops;
    int shape[12];  // N H W Ho Wo cin cout k stride gx gy gz
};
static std::vector<ConvProfEntry>* g_prof = nullptr;

void conv_profile_begin() {
    if (!g_prof) g_prof = new std::vector<ConvProfEntry>();
    g_prof->clear();
}
// cfg ids (BM x BN): 0 <2,2,4,4> 128x128  1 <1,4,2,2> 32x128  2 <4,1,4,4> 256x64  3 <2,2,2,2> 64x64  4 <4,1,4,2> 256x32
//   5 <2,2,2,1> 64x32  6 <4,1,4,1> 256x16  7 <4,1,1,1> 64x16  8 <1,4,4,2> 64x128  9 <2,2,4,2> 128x64  10 <4,1,2,2> 128x32
//   11 <4,1,2,1> 128x16; LDS-window 3x3 kernel: 12 (8x16)x128  13 (8x16)x64  14 (8x16)x32  15 (4x16)x128
//   16 7x7 heads  17 5x5 heads  18 3x3 heads (direct kernel for cout <= 2; the window kernel (8x16)x16 above that)
// bytes (optional): ALGORITHMIC HBM bytes of the launches of a configuration -- fp32 input map once + fp32 output map once +
// fp32 weights once (the figure SURVEY.md section 8d prices the streaming layers with); what the kernels really moved comes
// from the PMC passes under profiles/
int conv_profile_end(double* ms, double* flops, int* launches, double* bytes) {
    for (int i = 0; i < CONV_NUM_CFGS; i++) {
        ms[i] = 0;
        flops[i] = 0;
        launches[i] = 0;
        if (bytes) bytes[i] = 0;
    }
    if (!g_prof) return DFVO_OK;
    // DFVO_CONV_PROFILE_CSV=<path>: one line per launch (tuning aid)
    const char* csv_path = getenv("DFVO_CONV_PROFILE_CSV");
    FILE* csv = csv_path ? fopen(csv_path, "a") : nullptr;
    if (csv) fprintf(csv, "cfg,N,H,W,Ho,Wo,cin,cout,k,stride,gx,gy,gz,us,tflops\n");
    for (auto& e : *g_prof) {
        float t = 0.f;
        DFVO_HIP_CHECK(hipEventSynchronize(e.e1));
        DFVO_HIP_CHECK(hipEventElapsedTime(&t, e.e0, e.e1));
        ms[e.cfg] += t;
        if (csv)
            fprintf(csv, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.4f,%.4f\n", e.cfg, e.shape[0], e.shape[1], e.shape[2],
                    e.shape[3], e.shape[4], e.shape[5], e.shape[6], e.shape[7], e.shape[8], e.shape[9], e.shape[10],
                    e.shape[11], t * 1e3, e.flops / (t * 1e-3) / 1e12);
        flops[e.cfg] += e.flops;
        launches[e.cfg] += 1;
        if (bytes) {
            const double opx = (double)e.shape[0] * e.shape[3] * e.shape[4], ipx = (double)e.shape[0] * e.shape[1] * e.shape[2];
            const double macs_per_px = opx > 0 ? e.flops / (2.0 * opx) : 0.0;  // cout x cin x kh x kw
            bytes[e.cfg] += 4.0 * (ipx * e.shape[5] + opx * e.shape[6] + macs_per_px);
        }
        (void)hipEventDestroy(e.e0);
        (void)hipEventDestroy(e.e1);
    }
    if (csv) fclose(csv);
    delete g_prof;
    g_prof = nullptr;
    return DFVO_OK;
}

#include "conv_win_f16s.h"
#include "conv_gemm_f16s.h"
#include "conv_win_f16s2.h"

// split-K when the grid cannot fill the chip: partials to p.ws, ordered reduction in a second launch
static int conv_pick_splits(const ConvParams& p, long long blocks) {
    const long long M = (long long)p.N * p.Ho * p.Wo;
    int splits = 1;
    if (!p.ws) return 1;
    if (p.force_splits > 0)
        splits = p.force_splits;
    else if (blocks < 600 && p.ksteps >= 16)
        splits = (int)((1024 + blocks - 1) / blocks);
    if (splits > 1) {
        if (splits > p.ksteps / 8) splits = p.ksteps / 8;
        if (splits > 32) splits = 32;
        while (splits > 1 && (size_t)splits * M * p.cout_pad > p.ws_floats) --splits;
        if (splits < 1) splits = 1;
        // no empty z-slices
        const int per = (p.ksteps + splits - 1) / splits;
        splits = (p.ksteps + per - 1) / per;
    }
    return splits;
}

template <int WM, int WN, int TM, int TN>
static int launch_cfg(const ConvParams& p, hipStream_t stream, int cfg_id) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)(p.cout_pad / BN), 1);
    DFVO_ARG_CHECK((p.G0 + p.G1) * p.kh * p.kw + 4 <= MAX_KGROUPS, "launch_conv: too many k-groups for the LDS table");
    const int splits = conv_pick_splits(p, (long long)grid.x * grid.y);
    grid.z = (unsigned)splits;
    ConvProfEntry pe;
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventCreate(&pe.e0));
        DFVO_HIP_CHECK(hipEventCreate(&pe.e1));
        pe.cfg = cfg_id;
        const int cin = (p.G0 + p.G1) * 4;  // padded channels; the caller's useful-FLOP count is kept separately
        (void)cin;
        pe.flops = 0;
        DFVO_HIP_CHECK(hipEventRecord(pe.e0, stream));
    }
    ConvParams pk = p;  // in-kernel split-K finish when the tiles have tickets (else: second launch below)
    const bool fused = splits > 1 && pk.tile_flags && (long long)grid.x * grid.y <= pk.tile_flags_n;
    if (!fused) pk.tile_flags = nullptr;
    hipLaunchKernelGGL((conv_igemm_f32_kernel<WM, WN, TM, TN>), grid, dim3(256), 0, stream, pk);
    DFVO_HIP_CHECK(hipGetLastError());
    if (splits > 1 && !fused) {
        const int cols = p.dst_zero_to > p.cout ? p.dst_zero_to : p.cout;
        const long long total = M * cols;
        hipLaunchKernelGGL(conv_splitk_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, splits);
        DFVO_HIP_CHECK(hipGetLastError());
    }
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventRecord(pe.e1, stream));
        pe.flops = p.useful_flops;
        const int sh[12] = {p.N, p.H, p.W, p.Ho, p.Wo, (p.G0 + p.G1) * 4, p.cout, p.kh, p.stride, (int)grid.x, (int)grid.y, (int)grid.z};
        for (int i = 0; i < 12; ++i) pe.shape[i] = sh[i];
        g_prof->push_back(pe);
    }
    return DFVO_OK;
}

// Tile choice: the widest N tile the layer's cout allows (conv_pick_bn); the M tile from a sweep on MI355X
// (tools/sweep_conv.sh): 64-row tiles win or tie on every layer with BN <= 64 and on BN = 128 below ~1200
// workgroups (more resident workgroups hide the global -> LDS staging latency; the per-tap gather is served by
// L2 either way); 128 x 128 keeps the largest maps.  Small grids additionally split K inside launch_cfg.
// DFVO_CONV_FORCE_BM=<rows> overrides the M tile (tuning aid).
template <int WM, int WN, int TM, int TN, int KS = 3>
static int launch_win3(const ConvParams& p, hipStream_t stream, int cfg_id) {
    constexpr int TH = WM * TM, BN = WN * TN * 16;
    const int tiles = p.N * ((p.Ho + TH - 1) / TH) * ((p.Wo + 15) / 16);
    dim3 grid((unsigned)tiles, (unsigned)(p.cout_pad / BN), 1);
    // split the channel chunks when the tile grid cannot fill the chip
    const int nchunks = ((p.G0 + 3) >> 2) + ((p.G1 + 3) >> 2);
    const long long M = (long long)p.N * p.Ho * p.Wo;
    int splits = 1;
    const long long blocks = (long long)grid.x * grid.y;
    if (p.ws && blocks < 600 && nchunks >= 4) {
        splits = (int)((1024 + blocks - 1) / blocks);
        if (splits > nchunks / 2) splits = nchunks / 2;
        while (splits > 1 && (size_t)splits * M * p.cout_pad > p.ws_floats) --splits;
        if (splits < 1) splits = 1;
        const int per = (nchunks + splits - 1) / splits;
        splits = (nchunks + per - 1) / per;
    }
    grid.z = (unsigned)splits;
    ConvProfEntry pe;
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventCreate(&pe.e0));
        DFVO_HIP_CHECK(hipEventCreate(&pe.e1));
        pe.cfg = cfg_id;
        DFVO_HIP_CHECK(hipEventRecord(pe.e0, stream));
    }
    ConvParams pk = p;  // in-kernel split-K finish (exact fp32 kernel) when the tiles have tickets
    const bool fused = splits > 1 && pk.tile_flags && blocks <= pk.tile_flags_n;
    if (!fused) pk.tile_flags = nullptr;
    hipLaunchKernelGGL((conv_win_f32_kernel<WM, WN, TM, TN, KS>), grid, dim3(256), 0, stream, pk);
    DFVO_HIP_CHECK(hipGetLastError());
    if (splits > 1 && !fused) {
        const int cols = p.dst_zero_to > p.cout ? p.dst_zero_to : p.cout;
        const long long total = M * cols;
        hipLaunchKernelGGL(conv_splitk_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, splits);
        DFVO_HIP_CHECK(hipGetLastError());
    }
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventRecord(pe.e1, stream));
        pe.flops = p.useful_flops;
        const int sh[12] = {p.N, p.H, p.W, p.Ho, p.Wo, (p.G0 + p.G1) * 4, p.cout, p.kh, p.stride, (int)grid.x, (int)grid.y, (int)grid.z};
        for (int i = 0; i < 12; ++i) pe.shape[i] = sh[i];
        g_prof->push_back(pe);
    }
    return DFVO_OK;
}

template <int KS, int CO>
static int launch_head(const ConvParams& p, hipStream_t stream, int cfg_id) {
    static const bool py1_ok = !(getenv("DFVO_HEAD_PY1") && atoi(getenv("DFVO_HEAD_PY1")) == 0);
    const int tiles2 = p.N * ((p.Ho + 31) / 32) * ((p.Wo + 15) / 16);
    const bool py1 = py1_ok && tiles2 < 256;  // (one workgroup per CU is the first round; below that the chain per thread is the launch)
    const int tiles = py1 ? p.N * ((p.Ho + 15) / 16) * ((p.Wo + 15) / 16) : tiles2;
    dim3 grid((unsigned)tiles, 1, 1);
    ConvProfEntry pe;
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventCreate(&pe.e0));
        DFVO_HIP_CHECK(hipEventCreate(&pe.e1));
        pe.cfg = cfg_id;
        DFVO_HIP_CHECK(hipEventRecord(pe.e0, stream));
    }
    if (py1)
        hipLaunchKernelGGL((conv_head_f32_kernel<KS, CO, 1>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((conv_head_f32_kernel<KS, CO, 2>), grid, dim3(256), 0, stream, p);
    DFVO_HIP_CHECK(hipGetLastError());
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventRecord(pe.e1, stream));
        pe.flops = p.useful_flops;
        const int sh[12] = {p.N, p.H, p.W, p.Ho, p.Wo, (p.G0 + p.G1) * 4, p.cout, p.kh, p.stride, (int)grid.x, 1, 1};
        for (int i = 0; i < 12; ++i) pe.shape[i] = sh[i];
        g_prof->push_back(pe);
    }
    return DFVO_OK;
}

// the direct kernel serves the square, stride-1, "same"-padded heads with one or two output channels
static bool conv_use_head(const ConvParams& p) {
    if (p.cout > 2 || !p.wh) return false;
    if (p.kh != p.kw || p.stride != 1 || p.pad_h != p.kh / 2 || p.pad_w != p.kw / 2) return false;
    if (p.kh != 3 && p.kh != 5 && p.kh != 7) return false;
    return (long long)p.N * p.Ho * p.Wo >= 1024;
}

// the LDS-window kernel serves 3x3 / stride-1 layers on maps large enough to fill the chip with TH x 16 tiles
static bool conv_use_window(const ConvParams& p, int bn) {
    if (p.kh != p.kw || p.stride != 1 || p.pad_h != p.kh / 2 || p.pad_w != p.kw / 2) return false;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    if (p.kh == 3) return bn >= 32 && M >= 30000;
    // 5x5 / 7x7 flow heads (32 -> 2 channels): the whole tap loop runs out of one window
    return (p.kh == 5 || p.kh == 7) && bn == 16 && (p.G0 + p.G1) >= 4 && M >= 8000;
}

static int conv_pick_bm(const ConvParams& p, int bn) {
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const int force_bm = p.force_bm;  // (0: the rule below)
    const long long ntiles_n = p.cout_pad / bn;
    auto blocks = [&](int bm) { return ((M + bm - 1) / bm) * ntiles_n; };
    if (bn == 128) {
        int bm = blocks(128) >= 1200 ? 128 : 64;
        if (M <= 4096) bm = 32;
        if (force_bm == 128 || force_bm == 64 || force_bm == 32) bm = force_bm;
        return bm;
    }
    int bm = M >= 400000 ? 128 : 64;
    if (force_bm == 256 || force_bm == 128 || force_bm == 64) bm = force_bm;
    return bm;
}

// exact-fp32 mode: which layers leave conv_igemm_f32_kernel for the register-ring kernel (conv_gemm_f32g.hip).  Not the
// 3x3 / stride-1 layers the LDS-window kernel takes, not the one- / two-channel heads.
// Measured (profiles/r4e_fp32_f32g_ab.txt): the small maps only 156.5 -> 160 pairs/s; every non-window layer 158 (the
// streaming layers are no faster than on conv_igemm_f32_kernel).
static bool conv_f32g_takes(const ConvParams& p, int bn) {
    if (conv_use_head(p)) return false;
    if (conv_use_window(p, bn)) return false;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    return M <= 8192;  // pyramid levels 5 / 6 (2 x 11 x 38 .. 2 x 44 x 152) and the depth net's 6 x 20 .. 48 x 160 maps
}
static int launch_f32g_prof(const ConvParams& p, hipStream_t stream) {
    ConvProfEntry pe;
    if (g_prof) {
        DFVO_HIP_CHECK(hipEventCreate(&pe.e0));
        DFVO_HIP_CHECK(hipEventCreate(&pe.e1));
        DFVO_HIP_CHECK(hipEventRecord(pe.e0, stream));
    }
    int ksp = 1, gx = 0;
    const int rc = launch_f32g(p, stream, &ksp, &gx);
    if (rc != DFVO_OK) return rc;
    if (g_prof) {
        pe.cfg = ksp == 1 ? 22 : 23;  // profile rows: 22 streaming (KSP = 1), 23 K-sliced small maps
        DFVO_HIP_CHECK(hipEventRecord(pe.e1, stream));
        pe.flops = p.useful_flops;
        const int sh[12] = {p.N, p.H, p.W, p.Ho, p.Wo, (p.G0 + p.G1) * 4, p.cout, p.kh, p.stride, gx, 0, ksp * 100 + 1};
        for (int i = 0; i < 12; ++i) pe.shape[i] = sh[i];
        g_prof->push_back(pe);
    }
    return DFVO_OK;
}

int launch_conv(const ConvParams& p, hipStream_t stream) {
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const int bn = conv_pick_bn(p.cout, M);
    DFVO_ARG_CHECK(p.cout_pad % bn == 0, "launch_conv: cout_pad not a multiple of the N tile");
    DFVO_ARG_CHECK((p.cs0 % 4) == 0 && (p.co0 % 4) == 0, "launch_conv: src0 stride/offset must be multiples of 4");
    DFVO_ARG_CHECK(p.G1 == 0 || ((p.cs1 % 4) == 0 && (p.co1 % 4) == 0), "launch_conv: src1 stride/offset");
    if (conv_use_head(p)) {
        if (p.kh == 7) return p.cout == 2 ? launch_head<7, 2>(p, stream, 16) : launch_head<7, 1>(p, stream, 16);
        if (p.kh == 5) return p.cout == 2 ? launch_head<5, 2>(p, stream, 17) : launch_head<5, 1>(p, stream, 17);
        return p.cout == 2 ? launch_head<3, 2>(p, stream, 18) : launch_head<3, 1>(p, stream, 18);
    }
    if (p.wf16 && conv_f16s_ok(p)) {  // f16x3: the 3x3 / stride-1 layers whose map fills the chip
        const int rc2 = launch_f16s2(p, stream, 19);  // one-wave-per-SIMD skeleton where the grid is large enough
        return rc2 != F16S2_NOT_APPLICABLE ? rc2 : launch_f16s(p, stream, 19);  // errors (negative) propagate
    }
    if (conv_f16g_ok(p)) return launch_f16g(p, stream);  // f16x3: everything else (small maps, 1x1, k x 1, stride 2, 7x7)
    if (conv_f32g_ok(p) && conv_f32g_takes(p, bn)) return launch_f32g_prof(p, stream);  // exact fp32: the same skeleton on fp32 MFMAs
    if (conv_use_window(p, bn) && p.kh == 7) return launch_win3<4, 1, 2, 1, 7>(p, stream, 16);
    if (conv_use_window(p, bn) && p.kh == 5) return launch_win3<4, 1, 2, 1, 5>(p, stream, 17);
    if (conv_use_window(p, bn)) {
        const long long tiles8 = (long long)p.N * ((p.Ho + 7) / 8) * ((p.Wo + 15) / 16) * (p.cout_pad / bn);
        if (bn == 128) {
            // 128-wide layers on the largest maps run as two 64-wide column blocks: 2x the workgroups at 4 (instead
            // of 3) per CU shortens the under-filled last round of the grid (+5 % measured at 2 x 192 x 624)
            if (tiles8 >= 1200) return launch_win3<2, 2, 4, 2>(p, stream, 13);
            return tiles8 >= 400 ? launch_win3<2, 2, 4, 4>(p, stream, 12) : launch_win3<1, 4, 4, 2>(p, stream, 15);
        }
        if (bn == 64) return launch_win3<2, 2, 4, 2>(p, stream, 13);
        return launch_win3<4, 1, 2, 2>(p, stream, 14);
    }
    const int bm = conv_pick_bm(p, bn);
    if (bn == 128) {
        if (bm == 128) return launch_cfg<2, 2, 4, 4>(p, stream, 0);
        if (bm == 64) return launch_cfg<1, 4, 4, 2>(p, stream, 8);
        return launch_cfg<1, 4, 2, 2>(p, stream, 1);
    }
    if (bn == 64) {
        if (bm == 256) return launch_cfg<4, 1, 4, 4>(p, stream, 2);
        if (bm == 128) return launch_cfg<2, 2, 4, 2>(p, stream, 9);
        return launch_cfg<2, 2, 2, 2>(p, stream, 3);
    }
    if (bn == 32) {
        if (bm == 256) return launch_cfg<4, 1, 4, 2>(p, stream, 4);
        if (bm == 128) return launch_cfg<4, 1, 2, 2>(p, stream, 10);
        return launch_cfg<2, 2, 2, 1>(p, stream, 5);
    }
    if (bm == 256) return launch_cfg<4, 1, 4, 1>(p, stream, 6);
    if (bm == 128) return launch_cfg<4, 1, 2, 1>(p, stream, 11);
    return launch_cfg<4, 1, 1, 1>(p, stream, 7);
}

}  // namespace dfvo
