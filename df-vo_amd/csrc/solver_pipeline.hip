// Device-side tracker stages around the RANSAC solvers: keypoint selection (local_bestN), GRIC model
// selection, the repeated shuffled five-point RANSAC of EssTracker.compute_pose_2d2d, and depth-ratio
// scale recovery.  Reference call sites (paths relative to /root/reference):
//   libs/matching/kp_selection.py:74-200          local_bestN
//   libs/matching/keypoint_sampler.py:76-163      kp1 = pixel grid, kp2 = kp1 + flow
//   libs/tracker/gric.py:14-132                   Sampson / homography residuals + GRIC
//   libs/tracker/E_tracker.py:154-307             compute_pose_2d2d
//   libs/tracker/E_tracker.py:571-643             find_scale_from_depth (+ ops_3d.py:15-67)
//   sklearn RANSACRegressor.fit (third party)     subset draws from the global numpy RandomState
// Sequential semantics that leak into the results (argpartition order, python-loop summation order,
// the global np.random stream, last-writer-wins scatter) are kept by giving each sequential chain to one
// lane and spreading independent chains over lanes / workgroups.  Built with -ffp-contract=off.
#include "kp_select.h"
#include "np_legacy.h"
#include "solver.h"
#include "solver_math.h"
#include <cstring>

#include "tracker.h"

#include <atomic>

namespace dfvo {

__device__ __forceinline__ int wave_sum_i(int v) {
    v += __builtin_amdgcn_ds_swizzle(v, 0x041F);
    v += __builtin_amdgcn_ds_swizzle(v, 0x081F);
    v += __builtin_amdgcn_ds_swizzle(v, 0x101F);
    v += __builtin_amdgcn_ds_swizzle(v, 0x201F);
    v += __builtin_amdgcn_ds_swizzle(v, 0x401F);
    v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ================================================================================================
// keypoint selection
// ================================================================================================

// ------------------------------------------------------------------------------------------------
// numpy's introselect with the long partition passes run by the whole 256-thread workgroup.
// One pass of the unguarded Hoare partition  for(;;){ do ll++ while(v[ll]<p); do hh-- while(p<v[hh]); if(hh<ll)
// break; swap }  is equivalent to: L_k = k-th position (ascending, from low+2) whose value is not < pivot, R_k =
// k-th position (descending, from high-1) whose value is not > pivot; swap (L_k, R_k) for every k with L_k <= R_k
// (K of them, the pairs are disjoint); the scans of the crossing iteration stop at min(L_K, R_{K-1}) and
// max(R_K, L_{K-1}) because the slots exchanged last now hold stoppers.  The stopper lists are built with an
// ordered ballot/scan compaction, K by a count, the swaps one pair per thread.  Pivot choice, bookkeeping and the
// final short ranges stay on thread 0 (sm::kp_introselect_cp_from), so the resulting order is numpy's.
// ------------------------------------------------------------------------------------------------
constexpr int KP_PAR_MIN = 256;  // ranges shorter than this finish sequentially

__device__ __forceinline__ int kp_block_excl_scan2(int packed, int* s_wsum, int* total) {
    // exclusive scan over 256 threads of two 16-bit counters packed in one int; *total = packed grand total
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int inc = packed;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += s_wsum[w];
    *total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    __syncthreads();
    return base + inc - packed;
}

__device__ void kp_introselect_block(float* key, unsigned short* tosort, int num, int kth, unsigned short* Lpos,
                                     unsigned short* Rpos, int* s_ctl /*8 ints*/, int* s_wsum /*4 ints*/) {
    const int t = threadIdx.x;
    if (kth < 3 || kth == num - 1 || num < KP_PAR_MIN) {  // shortcuts of the scalar algorithm / small inputs
        if (t == 0) sm::kp_introselect_cp<unsigned short>(key, tosort, num, kth, 0);
        __syncthreads();
        return;
    }
    int low = 0, high = num - 1, depth_limit = sm::kp_msb((unsigned)num) * 2;
    while (low + 1 < high) {
        if (high - low < KP_PAR_MIN || depth_limit <= 0) break;  // thread 0 finishes (incl. the median-of-medians path)
        if (t == 0) {  // median of three -> pivot at low, its companion at low + 1
            const int mid = low + (high - low) / 2;
#define KPB_SWAP(i, j)                    \
    {                                     \
        unsigned short _t = tosort[i];    \
        tosort[i] = tosort[j];            \
        tosort[j] = _t;                   \
        float _k = key[i];                \
        key[i] = key[j];                  \
        key[j] = _k;                      \
    }
            if (sm::kp_lt(key[high], key[mid])) KPB_SWAP(high, mid);
            if (sm::kp_lt(key[high], key[low])) KPB_SWAP(high, low);
            if (sm::kp_lt(key[low], key[mid])) KPB_SWAP(low, mid);
            KPB_SWAP(mid, low + 1);
        }
        __syncthreads();
        const float pivot = key[low];
        // stopper lists over [low+1 .. high]: left stoppers from low+2 (high is one by construction), right stoppers
        // down from high-1 (low+1 is one by construction); each thread owns a contiguous segment
        const int r0 = low + 1, n_r = high - low;  // positions r0 .. r0 + n_r - 1
        const int seg = (n_r + 255) / 256;
        const int p0 = r0 + t * seg, p1 = p0 + seg < r0 + n_r ? p0 + seg : r0 + n_r;
        int cl = 0, cr = 0;
        for (int p = p0; p < p1; ++p) {
            const float v = key[p];
            cl += (p >= low + 2 && !sm::kp_lt(v, pivot)) ? 1 : 0;
            cr += (p <= high - 1 && !sm::kp_lt(pivot, v)) ? 1 : 0;
        }
        int total;
        const int ex = kp_block_excl_scan2(cl | (cr << 16), s_wsum, &total);
        const int nL = total & 0xffff, nR = total >> 16;
        int il = ex & 0xffff, ir = ex >> 16;
        for (int p = p0; p < p1; ++p) {
            const float v = key[p];
            if (p >= low + 2 && !sm::kp_lt(v, pivot)) Lpos[il++] = (unsigned short)p;
            if (p <= high - 1 && !sm::kp_lt(pivot, v)) Rpos[nR - 1 - (ir++)] = (unsigned short)p;
        }
        __syncthreads();
        // K = number of leading pairs with L_k <= R_k (monotone predicate)
        const int npair = nL < nR ? nL : nR;
        int cnt = 0;
        for (int k = t; k < npair; k += 256) cnt += Lpos[k] <= Rpos[k] ? 1 : 0;
        int tot2;
        (void)kp_block_excl_scan2(cnt, s_wsum, &tot2);
        const int K = tot2 & 0xffff;
        for (int k = t; k < K; k += 256) {
            const int a = Lpos[k], b = Rpos[k];
            if (a != b) KPB_SWAP(a, b);
        }
        __syncthreads();
        if (t == 0) {
            // crossing iteration: the scans run on from (L_{K-1}, R_{K-1}) and stop at the next original stopper or at
            // the nearest slot exchanged earlier (it now holds a stopper), whichever comes first.  When the last
            // exchanged pair was a self-pair (L == R, value == pivot) that slot lies behind both scans and the
            // pair before it takes its place.  K < nL, nR: the lists end with the sentinels high / low+1.
            int ll = Lpos[K], hh = Rpos[K];
            if (K > 0) {
                int rp = Rpos[K - 1], lp = Lpos[K - 1];
                if (rp == lp) {
                    rp = K > 1 ? Rpos[K - 2] : 0x7fffffff;
                    lp = K > 1 ? Lpos[K - 2] : -1;
                }
                ll = ll < rp ? ll : rp;
                hh = hh > lp ? hh : lp;
            }
            KPB_SWAP(low, hh);
            int nlow = low, nhigh = high;
            if (hh >= kth) nhigh = hh - 1;
            if (hh <= kth) nlow = ll;
            s_ctl[0] = nlow;
            s_ctl[1] = nhigh;
        }
        __syncthreads();
        low = s_ctl[0];
        high = s_ctl[1];
        depth_limit--;
        __syncthreads();
    }
    if (t == 0) sm::kp_introselect_cp_from<unsigned short>(key, tosort, kth, 0, low, high, depth_limit);
    __syncthreads();
#undef KPB_SWAP
}

// The same partition for arrays that live in global memory and are too long for 16-bit positions (bestN_flow_kp selects
// over the whole image): int positions, two plain 32-bit scans instead of one packed scan.
__device__ __forceinline__ int kp_block_excl_scan(int v, int* s_wsum, int* total) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += s_wsum[w];
    *total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    __syncthreads();
    return base + inc - v;
}

__device__ void kp_introselect_block_global(float* key, int* tosort, int num, int kth, int* Lpos, int* Rpos, int* s_ctl,
                                            int* s_wsum) {
    const int t = threadIdx.x;
    if (kth < 3 || kth == num - 1 || num < KP_PAR_MIN) {
        if (t == 0) sm::kp_introselect_cp<int>(key, tosort, num, kth, 0);
        __syncthreads();
        return;
    }
    int low = 0, high = num - 1, depth_limit = sm::kp_msb((unsigned)num) * 2;
    while (low + 1 < high) {
        if (high - low < KP_PAR_MIN || depth_limit <= 0) break;
        if (t == 0) {
            const int mid = low + (high - low) / 2;
#define KPG_SWAP(i, j)             \
    {                              \
        int _t = tosort[i];        \
        tosort[i] = tosort[j];     \
        tosort[j] = _t;            \
        float _k = key[i];         \
        key[i] = key[j];           \
        key[j] = _k;               \
    }
            if (sm::kp_lt(key[high], key[mid])) KPG_SWAP(high, mid);
            if (sm::kp_lt(key[high], key[low])) KPG_SWAP(high, low);
            if (sm::kp_lt(key[low], key[mid])) KPG_SWAP(low, mid);
            KPG_SWAP(mid, low + 1);
        }
        __threadfence_block();
        __syncthreads();
        const float pivot = key[low];
        const int r0 = low + 1, n_r = high - low;
        const int seg = (n_r + 255) / 256;
        const int p0 = r0 + t * seg, p1 = p0 + seg < r0 + n_r ? p0 + seg : r0 + n_r;
        int cl = 0, cr = 0;
        for (int p = p0; p < p1; ++p) {
            const float v = key[p];
            cl += (p >= low + 2 && !sm::kp_lt(v, pivot)) ? 1 : 0;
            cr += (p <= high - 1 && !sm::kp_lt(pivot, v)) ? 1 : 0;
        }
        int nL, nR;
        int il = kp_block_excl_scan(cl, s_wsum, &nL);
        int ir = kp_block_excl_scan(cr, s_wsum, &nR);
        for (int p = p0; p < p1; ++p) {
            const float v = key[p];
            if (p >= low + 2 && !sm::kp_lt(v, pivot)) Lpos[il++] = p;
            if (p <= high - 1 && !sm::kp_lt(pivot, v)) Rpos[nR - 1 - (ir++)] = p;
        }
        __threadfence_block();
        __syncthreads();
        const int npair = nL < nR ? nL : nR;
        int cnt = 0;
        for (int k = t; k < npair; k += 256) cnt += Lpos[k] <= Rpos[k] ? 1 : 0;
        int K;
        (void)kp_block_excl_scan(cnt, s_wsum, &K);
        for (int k = t; k < K; k += 256) {
            const int a = Lpos[k], b = Rpos[k];
            if (a != b) KPG_SWAP(a, b);
        }
        __threadfence_block();
        __syncthreads();
        if (t == 0) {  // crossing iteration, as in kp_introselect_block
            int ll = Lpos[K], hh = Rpos[K];
            if (K > 0) {
                int rp = Rpos[K - 1], lp = Lpos[K - 1];
                if (rp == lp) {
                    rp = K > 1 ? Rpos[K - 2] : 0x7fffffff;
                    lp = K > 1 ? Lpos[K - 2] : -1;
                }
                ll = ll < rp ? ll : rp;
                hh = hh > lp ? hh : lp;
            }
            KPG_SWAP(low, hh);
            int nlow = low, nhigh = high;
            if (hh >= kth) nhigh = hh - 1;
            if (hh <= kth) nlow = ll;
            s_ctl[0] = nlow;
            s_ctl[1] = nhigh;
        }
        __threadfence_block();
        __syncthreads();
        low = s_ctl[0];
        high = s_ctl[1];
        depth_limit--;
        __syncthreads();
    }
    if (t == 0) sm::kp_introselect_cp_from<int>(key, tosort, kth, 0, low, high, depth_limit);
    __syncthreads();
#undef KPG_SWAP
}

// bestN_flow_kp (kp_selection.py:33-71, ablation_correspondences_best_n.yml): np.where(flow_diff >= 0) keeps every
// non-NaN pixel in row-major order; np.argpartition(values, N)[:N] then picks N of them in introselect order
__global__ void k_bestn_fill(const float* __restrict__ diff, int n, float* __restrict__ key, int* __restrict__ tosort,
                             int* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int ok = 0;
    if (i < n) {
        const float v = diff[i];
        key[i] = v;
        tosort[i] = i;
        ok = v >= 0.f ? 1 : 0;
    }
    const int c = wave_sum_i(ok);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// one workgroup: (only when some pixels fail `>= 0`) ordered compaction, then the selection and the gather of the
// first N picks; info[0] = number of keypoints (0 when the image has N or fewer candidates: numpy would raise)
__global__ __launch_bounds__(256) void k_bestn_select(const float* __restrict__ diff, const float* __restrict__ flow, int H,
                                                       int W, int N, float* __restrict__ key, int* __restrict__ tosort,
                                                       int* __restrict__ map, int* __restrict__ Lpos, int* __restrict__ Rpos,
                                                       const int* __restrict__ count, double* __restrict__ kp1,
                                                       double* __restrict__ kp2, int* __restrict__ info) {
    __shared__ int s_ctl[8], s_wsum[4], s_base;
    const int t = threadIdx.x;
    const int n = H * W;
    int cnt = *count;
    const bool identity = cnt == n;
    if (!identity) {  // ordered compaction of the pixels that pass `>= 0`
        if (t == 0) s_base = 0;
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += 256) {
            const int e = c0 + t;
            const float v = e < n ? diff[e] : -1.f;
            const int f = (e < n && v >= 0.f) ? 1 : 0;
            int tot;
            const int ex = kp_block_excl_scan(f, s_wsum, &tot);
            if (f) {
                const int pos = s_base + ex;
                key[pos] = v;
                tosort[pos] = pos;
                map[pos] = e;
            }
            __syncthreads();
            if (t == 0) s_base += tot;
            __syncthreads();
        }
        cnt = s_base;
    }
    if (cnt <= N) {  // kth = N out of bounds
        if (t == 0) info[0] = 0;
        return;
    }
    __threadfence_block();
    __syncthreads();
    kp_introselect_block_global(key, tosort, cnt, N, Lpos, Rpos, s_ctl, s_wsum);
    __threadfence_block();
    __syncthreads();
    for (int i = t; i < N; i += 256) {
        const int c = tosort[i];
        const int e = identity ? c : map[c];
        const int y = e / W, x = e - y * W;
        kp1[i * 2] = (double)x;
        kp1[i * 2 + 1] = (double)y;
        kp2[i * 2] = (double)x + (double)flow[e];
        kp2[i * 2 + 1] = (double)y + (double)flow[(size_t)n + e];
    }
    if (t == 0) info[0] = N;
}

void BestNBuffers::release() {
    void* ptrs[] = {key_base, tosort, map, Lpos, Rpos, count, kp};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    key_base = nullptr;
    tosort = map = Lpos = Rpos = count = nullptr;
    kp = nullptr;
    cap = 0;
    kp_cap = 0;
}

int enqueue_bestn_flow_kp(BestNBuffers& bb, const float* d_flow, const float* d_diff, int H, int W, int N, hipStream_t s) {
    DFVO_ARG_CHECK(H > 0 && W > 0 && N >= 1 && (long long)H * W < (1ll << 30), "bestN: bad size");
    const int n = H * W;
    if ((size_t)n > bb.cap) {
        const int keep_kp = bb.kp_cap;
        double* kp_keep = bb.kp;
        bb.kp = nullptr;
        bb.release();
        bb.kp = kp_keep;
        bb.kp_cap = keep_kp;
        bb.cap = (size_t)n;
        DFVO_HIP_CHECK(hipMalloc((void**)&bb.key_base, sizeof(float) * ((size_t)n + 16)));  // slack: the 4-wide scans over-read
        DFVO_HIP_CHECK(hipMemset(bb.key_base, 0, sizeof(float) * ((size_t)n + 16)));
        DFVO_HIP_CHECK(hipMalloc((void**)&bb.tosort, sizeof(int) * (size_t)n));
        DFVO_HIP_CHECK(hipMalloc((void**)&bb.map, sizeof(int) * (size_t)n));
        DFVO_HIP_CHECK(hipMalloc((void**)&bb.Lpos, sizeof(int) * ((size_t)n + 2)));
        DFVO_HIP_CHECK(hipMalloc((void**)&bb.Rpos, sizeof(int) * ((size_t)n + 2)));
        DFVO_HIP_CHECK(hipMalloc((void**)&bb.count, sizeof(int) * 4));
    }
    if (N > bb.kp_cap) {
        if (bb.kp) (void)hipFree(bb.kp);
        bb.kp_cap = N;
        DFVO_HIP_CHECK(hipMalloc((void**)&bb.kp, sizeof(double) * 4 * (size_t)N));
    }
    float* key = bb.key_base + 8;
    DFVO_HIP_CHECK(hipMemsetAsync(bb.count, 0, sizeof(int) * 4, s));
    hipLaunchKernelGGL(k_bestn_fill, dim3(cdiv(n, 256)), dim3(256), 0, s, d_diff, n, key, bb.tosort, bb.count);
    hipLaunchKernelGGL(k_bestn_select, dim3(1), dim3(256), 0, s, d_diff, d_flow, H, W, N, key, bb.tosort, bb.map, bb.Lpos,
                       bb.Rpos, bb.count, bb.kp, bb.kp + 2 * (size_t)N, bb.count + 1);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// one 256-thread block per grid cell: ordered (row-major) compaction of the candidates into LDS, then
// lane 0 runs numpy's introselect on them (keys carried along with the indices, see kp_select.h); writes
// the picked local indices in argpartition order.
// Blocks [cells, cells + KP_CNT_BLOCKS) do not select: they count the pixels of the whole consistency map `mask_map` under the
// threshold (kp_selection.py:158: mask.sum() < N * 0.1 -> "not enough keypoints") into count_partial[], which k_kp_gather adds up
// (round 6: was a memset + k_kp_count in front of this launch, two more dependent launches on the path to the first pose).
constexpr int KP_CNT_BLOCKS = 64;
__global__ __launch_bounds__(256) void k_kp_cell(const float* __restrict__ diff, int H, int W, int num_row, int num_col,
                                                  float thre, int n_best, int cap, int* __restrict__ cell_count,
                                                  int* __restrict__ cell_sel /*[cells][n_best] (y<<16|x)*/,
                                                  unsigned short* __restrict__ lidx_all /*[cells][cap]*/, int par,
                                                  const float* __restrict__ mask_map, int* __restrict__ count_partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if ((int)blockIdx.x >= num_row * num_col) {
        __shared__ int s_cnt[4];
        const int b = blockIdx.x - num_row * num_col, n = H * W;
        int c = 0;
        for (int i = b * 256 + threadIdx.x; i < n; i += KP_CNT_BLOCKS * 256) c += mask_map[i] < thre ? 1 : 0;
        c = wave_sum_i(c);
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) count_partial[b] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        return;
    }
    float* vals = reinterpret_cast<float*>(smem_raw) + 4;  // 4 floats of slack on either side: the 4-wide scans over-read
    unsigned short* tosort = reinterpret_cast<unsigned short*>(vals + cap + 4);
    unsigned short* Lpos = tosort + cap + 2;  // stopper lists of the workgroup-parallel partition (par != 0)
    unsigned short* Rpos = Lpos + cap + 2;
    __shared__ int s_ctl[8], s_wsum[4];
    unsigned short* lidx = lidx_all + (size_t)blockIdx.x * cap;  // candidate -> tile element (global scratch)
    __shared__ int s_base, s_wave[4];
    const int cell = blockIdx.x;
    const int row = cell / num_col, col = cell - row * num_col;
    int y0, y1, x0, x1;
    sm::kp_cell_bounds(H, W, num_row, num_col, row, col, &y0, &y1, &x0, &x1);
    const int th = y1 - y0 > 0 ? y1 - y0 : 0, tw = x1 - x0 > 0 ? x1 - x0 : 0;
    const int total = th * tw;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_base = 0;
    __syncthreads();
    // the cell's values in batches of eight loads per thread (round 6: one load per compaction step left eighteen dependent
    // round trips to memory in front of the selection of a KITTI-sized cell)
    constexpr int KB = 8;
    float vb[KB];
    for (int c0 = 0; c0 < total; c0 += 256) {
        const int e = c0 + t;
        const int slot = (c0 / 256) % KB;
        if (slot == 0) {
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                const int eu = c0 + u * 256 + t;
                vb[u] = 0.f;
                if (eu < total) {
                    const int ly = eu / tw, lx = eu - ly * tw;
                    vb[u] = diff[(size_t)(y0 + ly) * W + x0 + lx];
                }
            }
        }
        bool f = false;
        float v = 0.f;
        if (e < total) {
            v = vb[0];
#pragma unroll
            for (int u = 1; u < KB; ++u) v = slot == u ? vb[u] : v;
            f = v < thre;
        }
        const unsigned long long b = __ballot(f);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        if (f) {
            const int pos = off + before;
            vals[pos] = v;
            tosort[pos] = (unsigned short)pos;
            lidx[pos] = (unsigned short)e;
        }
        __syncthreads();
        if (t == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    const int cnt = s_base;
    const int pick = cnt < n_best ? cnt : n_best;
    if (pick > 0) {
        if (par) {
            kp_introselect_block(vals, tosort, cnt, pick - 1, Lpos, Rpos, s_ctl, s_wsum);
        } else if (t == 0) {
            sm::kp_introselect_cp<unsigned short>(vals, tosort, cnt, pick - 1, 0);
        }
    }
    if (t == 0) cell_count[cell] = pick;
    __syncthreads();
    if (t < pick) {
        const int e = lidx[tosort[t]];
        const int ly = e / tw, lx = e - ly * tw;
        cell_sel[cell * n_best + t] = ((y0 + ly) << 16) | (x0 + lx);
    }
}

// concatenate the cells (row-major cell order), build kp1 (pixel grid) and kp2 = kp1 + flow.  count_partial (optional): the
// KP_CNT_BLOCKS partial counts of k_kp_cell's counting blocks, else *total_good is the count.  (Round 6: the cell counts go to
// LDS in one parallel read and the (cell, k) items are spread over the threads -- the loop over the cells with a dependent
// global read each was 45 us of one workgroup.)
__global__ __launch_bounds__(256) void k_kp_gather(const int* __restrict__ cell_count, const int* __restrict__ cell_sel,
                                                    int cells, int n_best, const float* __restrict__ flow, int H, int W,
                                                    const int* __restrict__ total_good, const int* __restrict__ count_partial,
                                                    int min_total, int min_regions,
                                                    double* __restrict__ kp1, double* __restrict__ kp2,
                                                    int* __restrict__ info /*[n, good_kp_found, regions]*/) {
    __shared__ int s_off[1025], s_cnt[1024], s_good;
    const int t = threadIdx.x;
    for (int c = t; c < cells; c += 256) s_cnt[c] = cell_count[c];
    __syncthreads();
    if (t == 0) {
        int acc = 0, regions = 0;
        for (int c = 0; c < cells; c++) {
            s_off[c] = acc;
            acc += s_cnt[c];
            regions += s_cnt[c] != 0;
        }
        s_off[cells] = acc;
        int good = 0;
        if (count_partial)
            for (int b = 0; b < KP_CNT_BLOCKS; ++b) good += count_partial[b];
        else
            good = *total_good;
        const bool enough = !(good < min_total);            // (mask.sum() < N*0.1) -> fail
        const bool diverse = !(regions < min_regions);      // good_region_cnt < rows*cols*0.1 -> fail
        info[0] = (enough && diverse) ? acc : 0;
        info[1] = (enough && diverse) ? 1 : 0;
        info[2] = regions;
        s_good = (enough && diverse) ? 1 : 0;
    }
    __syncthreads();
    if (!s_good) return;
    for (int i = t; i < cells * n_best; i += 256) {
        const int c = i / n_best, k = i - c * n_best;
        if (k < s_cnt[c]) {
            const int code = cell_sel[i];
            const int y = code >> 16, x = code & 0xffff;
            const int o = s_off[c] + k;
            kp1[o * 2] = (double)x;
            kp1[o * 2 + 1] = (double)y;
            kp2[o * 2] = (double)x + (double)flow[(size_t)y * W + x];
            kp2[o * 2 + 1] = (double)y + (double)flow[(size_t)H * W + (size_t)y * W + x];
        }
    }
}

// ================================================================================================
// rigid-flow keypoints (RigidFlow layer + opt_rigid_flow_kp, the "kp_depth" correspondences of
// scale_recovery.method iterative / the extended-paper configurations)
// ================================================================================================
// RigidFlow(depth, T, K, inv_K, normalized=False) (geometry/rigid_flow.py, backprojection.py:56-62,
// transformation3d.py:29, projection.py:46-52, layers.py PixToFlow:262) and its distance to the optical flow
// (E_tracker.py:685-689), all in float32.  Every matmul row is a short dot product that torch's CPU GEMM evaluates as
// a0*b0 rounded, then fused multiply-adds in ascending k; written out the same way, the result is bit-identical to the
// torch-CPU oracle.  Kinv: 3x3, T: 4x4, K: 3x3 (its 4th column in the reference is zero).
__global__ void k_rigid_flow_diff(const float* __restrict__ depth, const float* __restrict__ flow, int H, int W,
                                  const float* __restrict__ mats /*Kinv[9] | T[16] | K[9]*/, float* __restrict__ rdiff,
                                  float* __restrict__ rflow /*optional [2,H,W]*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const float x = (float)(i % W), y = (float)(i / W);
    float rx, ry;
    sm::rigid_flow_px(mats, mats + 9, mats + 25, x, y, depth[i], &rx, &ry);
    if (rflow) {
        rflow[i] = rx;
        rflow[(size_t)H * W + i] = ry;
    }
    const float dx = rx - flow[i], dy = ry - flow[(size_t)H * W + i];
    rdiff[i] = sqrtf(dx * dx + dy * dy);  // np.linalg.norm(axis=0) on float32: sqrt(x*x + y*y), each step rounded
}

// opt_rigid_flow_kp (kp_selection.py:203-324), one workgroup per grid cell: candidates = pixels of the cell (last row /
// column dropped) with rigid-flow distance < thr_r AND forward-backward distance < thr_o, in row-major order;
// "uniform" picks every step-th candidate, "best" the num_to_pick smallest scores in numpy's argpartition order
__global__ __launch_bounds__(256) void k_kp_cell_rigid(const float* __restrict__ odiff, const float* __restrict__ rdiff,
                                                        int H, int W, int num_row, int num_col, float thr_o, float thr_r,
                                                        int score_rigid, int n_best, int cap, int* __restrict__ cell_count,
                                                        int* __restrict__ cell_sel, int* __restrict__ cell_sel_uni,
                                                        unsigned short* __restrict__ lidx_all, int par) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* vals = reinterpret_cast<float*>(smem_raw) + 4;
    unsigned short* tosort = reinterpret_cast<unsigned short*>(vals + cap + 4);
    unsigned short* Lpos = tosort + cap + 2;
    unsigned short* Rpos = Lpos + cap + 2;
    __shared__ int s_ctl[8], s_wsum[4];
    unsigned short* lidx = lidx_all + (size_t)blockIdx.x * cap;
    __shared__ int s_base, s_wave[4];
    const int cell = blockIdx.x;
    const int row = cell / num_col, col = cell - row * num_col;
    int y0, y1, x0, x1;
    sm::kp_cell_bounds(H, W, num_row, num_col, row, col, &y0, &y1, &x0, &x1);
    const int th = y1 - y0 > 0 ? y1 - y0 : 0, tw = x1 - x0 > 0 ? x1 - x0 : 0;
    const int total = th * tw;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < total; c0 += 256) {
        const int e = c0 + t;
        bool f = false;
        float v = 0.f;
        if (e < total) {
            const int ly = e / tw, lx = e - ly * tw;
            const size_t g = (size_t)(y0 + ly) * W + x0 + lx;
            const float vr = rdiff[g], vo = odiff[g];
            f = (vr < thr_r) && (vo < thr_o);
            v = score_rigid ? vr : vo;
        }
        const unsigned long long b = __ballot(f);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        if (f) {
            const int pos = off + before;
            vals[pos] = v;
            tosort[pos] = (unsigned short)pos;
            lidx[pos] = (unsigned short)e;
        }
        __syncthreads();
        if (t == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    const int cnt = s_base;
    const int pick = cnt < n_best ? cnt : n_best;
    if (pick > 0) {
        const int step = cnt / pick;  // np.arange(0, cnt, step)[:pick]
        if (t < pick) {
            const int e = lidx[t * step];
            const int ly = e / tw, lx = e - ly * tw;
            cell_sel_uni[cell * n_best + t] = ((y0 + ly) << 16) | (x0 + lx);
        }
        if (par) {
            kp_introselect_block(vals, tosort, cnt, pick - 1, Lpos, Rpos, s_ctl, s_wsum);
        } else if (t == 0) {
            sm::kp_introselect_cp<unsigned short>(vals, tosort, cnt, pick - 1, 0);
        }
    }
    if (t == 0) cell_count[cell] = pick;
    __syncthreads();
    if (t < pick) {
        const int e = lidx[tosort[t]];
        const int ly = e / tw, lx = e - ly * tw;
        cell_sel[cell * n_best + t] = ((y0 + ly) << 16) | (x0 + lx);
    }
}

void RigidKpBuffers::release() {
    void* ptrs[] = {depth32, rdiff, mats, cell_count, cell_sel, cell_sel_uni, lidx, kp, info, zero};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    depth32 = rdiff = mats = nullptr;
    cell_count = cell_sel = cell_sel_uni = info = zero = nullptr;
    lidx = nullptr;
    kp = nullptr;
    px_cap = sel_cap = 0;
    lidx_cap = 0;
}

int RigidKpBuffers::ensure(int H, int W, int cells, int n_best, int cap) {
    const size_t px = (size_t)H * W;
    if (px > px_cap) {
        if (depth32) (void)hipFree(depth32);
        if (rdiff) (void)hipFree(rdiff);
        px_cap = px;
        DFVO_HIP_CHECK(hipMalloc((void**)&depth32, sizeof(float) * px));
        DFVO_HIP_CHECK(hipMalloc((void**)&rdiff, sizeof(float) * px));
    }
    if (!mats) {
        DFVO_HIP_CHECK(hipMalloc((void**)&mats, sizeof(float) * 40));
        DFVO_HIP_CHECK(hipMalloc((void**)&cell_count, sizeof(int) * 1024));
        DFVO_HIP_CHECK(hipMalloc((void**)&info, sizeof(int) * 8));
        DFVO_HIP_CHECK(hipMalloc((void**)&zero, sizeof(int) * 2));
        DFVO_HIP_CHECK(hipMemset(zero, 0, sizeof(int) * 2));
    }
    if (cells * n_best > sel_cap) {
        for (void* p : {(void*)cell_sel, (void*)cell_sel_uni, (void*)kp})
            if (p) (void)hipFree(p);
        sel_cap = cells * n_best;
        DFVO_HIP_CHECK(hipMalloc((void**)&cell_sel, sizeof(int) * sel_cap));
        DFVO_HIP_CHECK(hipMalloc((void**)&cell_sel_uni, sizeof(int) * sel_cap));
        DFVO_HIP_CHECK(hipMalloc((void**)&kp, sizeof(double) * 8 * sel_cap));
    }
    if ((size_t)cells * cap > lidx_cap) {
        if (lidx) (void)hipFree(lidx);
        lidx_cap = (size_t)cells * cap;
        DFVO_HIP_CHECK(hipMalloc((void**)&lidx, sizeof(unsigned short) * lidx_cap));
    }
    return DFVO_OK;
}

// rigid flow of the reference depth under `T` (ref -> cur), its distance to the optical flow (kept in rb.rdiff), then
// the two keypoint sets: rb.kp = [kp1_best | kp2_best | kp1_uniform | kp2_uniform], each sel_cap x 2 doubles;
// rb.info[0] = their common count.  d_rdiff_override (optional) replaces the computed distance map.
int enqueue_rigid_flow_kp(RigidKpBuffers& rb, const float* d_flow, const float* d_odiff, const float* d_depth32, int H,
                          int W, const RigidKpConfig& cfg, const float* d_rdiff_override, hipStream_t s) {
    const int cells = cfg.num_row * cfg.num_col;
    DFVO_ARG_CHECK(cells > 0 && cells <= 1024 && H < 65536 && W < 65536, "rigid_flow_kp: grid too large");
    const int n_best = cfg.num_bestN / cells;
    DFVO_ARG_CHECK(n_best >= 1 && n_best <= 256, "rigid_flow_kp: n_best out of range");
    const int cap = (H / cfg.num_row + 2) * (W / cfg.num_col + 2);
    DFVO_ARG_CHECK(cap < 65536, "rigid_flow_kp: cell larger than 65535 pixels");
    size_t lds = (size_t)cap * (4 + 2) + 32;
    DFVO_ARG_CHECK(lds <= 158 * 1024, "rigid_flow_kp: cell does not fit in LDS");
    const int par = lds + (size_t)cap * 4 + 16 <= 150 * 1024 ? 1 : 0;
    if (par) lds += (size_t)cap * 4 + 16;
    int rc = rb.ensure(H, W, cells, n_best, cap);
    if (rc != DFVO_OK) return rc;
    if (int rc_lds = ensure_dyn_lds((const void*)k_kp_cell_rigid, lds)) return rc_lds;
    const float* rdiff = d_rdiff_override;
    if (!rdiff) {
        float m[34];
        for (int i = 0; i < 9; i++) m[i] = cfg.Kinv[i];
        for (int i = 0; i < 16; i++) m[9 + i] = cfg.T[i];
        for (int i = 0; i < 9; i++) m[25 + i] = cfg.K[i];
        DFVO_HIP_CHECK(hipMemcpyAsync(rb.mats, m, sizeof(m), hipMemcpyHostToDevice, s));
        DFVO_HIP_CHECK(hipStreamSynchronize(s));  // `m` is a stack buffer
        hipLaunchKernelGGL(k_rigid_flow_diff, dim3(cdiv(H * W, 256)), dim3(256), 0, s, d_depth32, d_flow, H, W, rb.mats,
                           rb.rdiff, (float*)nullptr);
        rdiff = rb.rdiff;
    }
    hipLaunchKernelGGL(k_kp_cell_rigid, dim3(cells), dim3(256), lds, s, d_odiff, rdiff, H, W, cfg.num_row, cfg.num_col,
                       cfg.opt_thre, cfg.rigid_thre, cfg.score_rigid, n_best, cap, rb.cell_count, rb.cell_sel,
                       rb.cell_sel_uni, rb.lidx, par);
    // both sets in cell order; no "enough keypoints" rules here (the reference asserts a non-empty selection)
    const size_t sc = (size_t)rb.sel_cap * 2;
    hipLaunchKernelGGL(k_kp_gather, dim3(1), dim3(256), 0, s, rb.cell_count, rb.cell_sel, cells, n_best, d_flow, H, W, rb.zero,
                       (const int*)nullptr, 0, 0, rb.kp, rb.kp + sc, rb.info);
    hipLaunchKernelGGL(k_kp_gather, dim3(1), dim3(256), 0, s, rb.cell_count, rb.cell_sel_uni, cells, n_best, d_flow, H, W,
                       rb.zero, (const int*)nullptr, 0, 0, rb.kp + 2 * sc, rb.kp + 3 * sc, rb.info + 4);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// sampled_kp (kp_selection.py:327-378): the k-th pixel (row-major) of the cropped grid [y0:y1, x0:x1] for every k of
// the uniform index list; kp1 = (x, y), kp2 = kp1 + flow (float32 promoted to float64, as numpy does)
__global__ void k_kp_sampled(const float* __restrict__ flow, int H, int W, int y0, int x0, int cw,
                             const int* __restrict__ idx, int n, double* __restrict__ kp1, double* __restrict__ kp2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = idx[i];
    const int yy = y0 + k / cw, xx = x0 + k % cw;
    const double x = (double)xx, y = (double)yy;
    kp1[i * 2] = x;
    kp1[i * 2 + 1] = y;
    kp2[i * 2] = x + (double)flow[(size_t)yy * W + xx];
    kp2[i * 2 + 1] = y + (double)flow[((size_t)H + yy) * W + xx];
}

int enqueue_kp_sampled(const float* d_flow, int H, int W, int y0, int y1, int x0, int x1, const int* d_idx, int n,
                       double* d_kp1, double* d_kp2, hipStream_t s) {
    DFVO_ARG_CHECK(0 <= y0 && y0 < y1 && y1 <= H && 0 <= x0 && x0 < x1 && x1 <= W, "sampled_kp: crop outside the image");
    if (n <= 0) return DFVO_OK;
    hipLaunchKernelGGL(k_kp_sampled, dim3(cdiv(n, 256)), dim3(256), 0, s, d_flow, H, W, y0, x0, x1 - x0, d_idx, n, d_kp1, d_kp2);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// score_method 'flow_ratio' (kp_selection.py:137-141,155): mask and score are flow_diff / |flow| per pixel, float32 as numpy
// evaluates it -- np.linalg.norm over the 2-vector = sqrt(x*x + y*y) with separately rounded products and sum
__global__ void k_flow_ratio(const float* __restrict__ flow, const float* __restrict__ diff, int px, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= px) return;
    const float fx = flow[i], fy = flow[px + i];
    const float xx = __fmul_rn(fx, fx), yy = __fmul_rn(fy, fy);
    out[i] = __fdiv_rn(diff[i], __fsqrt_rn(__fadd_rn(xx, yy)));
}

int enqueue_local_bestn(TrackerBuffers& tb, const float* d_flow, const float* d_diff, int H, int W, int num_row,
                        int num_col, int num_bestN, float thre, hipStream_t s, int score_method) {
    const int cells = num_row * num_col;
    DFVO_ARG_CHECK(cells > 0 && cells <= 1024 && H < 65536 && W < 65536, "local_bestN: grid too large");
    const int n_best = num_bestN / cells;  // math.floor(N / (rows*cols))
    DFVO_ARG_CHECK(n_best >= 1 && n_best <= 256, "local_bestN: n_best out of range");
    const int cap = (H / num_row + 2) * (W / num_col + 2);
    DFVO_ARG_CHECK(cap < 65536, "local_bestN: cell larger than 65535 pixels");
    size_t lds = (size_t)cap * (4 + 2) + 32;
    DFVO_ARG_CHECK(lds <= 158 * 1024, "local_bestN: cell does not fit in LDS");
    // the workgroup-parallel partition needs two more index lists; very large cells keep the single-lane selection
    const int par = lds + (size_t)cap * 4 + 16 <= 150 * 1024 ? 1 : 0;
    if (par) lds += (size_t)cap * 4 + 16;
    int rc = tb.ensure_kp(cells * n_best, cells, n_best);
    if (rc != DFVO_OK) return rc;
    if ((size_t)cells * cap > tb.lidx_cap) {
        if (tb.lidx) (void)hipFree(tb.lidx);
        tb.lidx_cap = (size_t)cells * cap;
        DFVO_HIP_CHECK(hipMalloc((void**)&tb.lidx, sizeof(unsigned short) * tb.lidx_cap));
    }
    if (int rc_lds = ensure_dyn_lds((const void*)k_kp_cell, lds)) return rc_lds;
    const float* d_key = d_diff;  // what the cells threshold and rank: the consistency map, or its ratio to the flow magnitude
    if (score_method == 1) {
        if ((size_t)H * W > tb.ratio_cap) {
            if (tb.ratio_map) (void)hipFree(tb.ratio_map);
            tb.ratio_cap = (size_t)H * W;
            DFVO_HIP_CHECK(hipMalloc((void**)&tb.ratio_map, sizeof(float) * tb.ratio_cap));
        }
        hipLaunchKernelGGL(k_flow_ratio, dim3(cdiv(H * W, 256)), dim3(256), 0, s, d_flow, d_diff, H * W, tb.ratio_map);
        d_key = tb.ratio_map;
    }
    // (the counting blocks read the consistency map itself, whatever the cells rank: kp_selection.py:158)
    hipLaunchKernelGGL(k_kp_cell, dim3(cells + KP_CNT_BLOCKS), dim3(256), lds, s, d_key, H, W, num_row, num_col, thre, n_best, cap,
                       tb.cell_count, tb.cell_sel, tb.lidx, par, d_diff, tb.kp_total + 8);
    // thresholds exactly as the python float comparisons: count < N*0.1 ; regions < rows*cols*0.1
    const int min_total = (int)ceil((double)num_bestN * 0.1);
    const int min_regions = (int)ceil((double)cells * 0.1);
    hipLaunchKernelGGL(k_kp_gather, dim3(1), dim3(256), 0, s, tb.cell_count, tb.cell_sel, cells, n_best, d_flow, H, W,
                       tb.kp_total, tb.kp_total + 8, min_total, min_regions, tb.kp_ref, tb.kp_cur, tb.kp_info);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ================================================================================================
// numpy RNG on the device
// ================================================================================================
__global__ void k_mt_seed(uint32_t* __restrict__ st, uint32_t seed) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int pos = 0; pos < 624; pos++) {
        st[pos] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
    }
    st[624] = 624;
}

// `repeat` consecutive draws of  perm = np.arange(n); np.random.shuffle(perm)  (n = info[0] on the device).
// MT19937 block regeneration (624 words, three dependency-free phases) and tempering run on all 256
// threads; the Fisher-Yates chain itself is sequential (masked rejection + data-dependent swaps) and runs
// on lane 0 out of LDS, handing control back whenever the block of tempered words is exhausted.
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & 0x9908b0dfu);
}

__device__ __forceinline__ uint32_t mt_mask_of(uint32_t v) {
    v |= v >> 1;
    v |= v >> 2;
    v |= v >> 4;
    v |= v >> 8;
    v |= v >> 16;
    return v;
}

// Two phases per group of `group` repeats (as many as fit the LDS):
//  A  the masked-rejection draws  j_i = random_interval(i), i = n-1 .. 1  of every repeat, in stream order.
//     One sequential chain, run on wave 0 with every loop-carried value wave-uniform (v_readlane of a
//     64-word batch + scalar compare/branch), the MT19937 block regeneration + tempering on all threads.
//  B  the swap chains  p[i] <-> p[j_i]  of the repeats, one lane per repeat in lockstep out of LDS
//     (the repeats are independent once their j lists are known).
__global__ __launch_bounds__(256) void k_mt_shuffle_all(uint32_t* __restrict__ st, const int* __restrict__ n_ptr,
                                                         int repeat, int group, int cap, int* __restrict__ perm_all) {
    __shared__ uint32_t key[624], outw[624];
    __shared__ int s_pos, s_rep, s_i;
    extern __shared__ int s_dyn[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int n = *n_ptr;
    if (n <= 1) {  // nothing is drawn
        if (t == 0 && n == 1)
            for (int r = 0; r < repeat; ++r) perm_all[(size_t)r * cap] = 0;
        return;
    }
    int* p = s_dyn;                                                      // [group][n]
    unsigned short* jl = reinterpret_cast<unsigned short*>(s_dyn + (size_t)group * n);  // [group][n]
    for (int i = t; i < 624; i += 256) {
        key[i] = st[i];
        outw[i] = mt_temper(st[i]);
    }
    if (t == 0) {
        s_pos = (int)st[624];
        s_rep = 0;
        s_i = n - 1;
    }
    __syncthreads();
    for (int g0 = 0; g0 < repeat; g0 += group) {
        const int g_end = g0 + group < repeat ? g0 + group : repeat;
        for (int i = t; i < (g_end - g0) * n; i += 256) p[i] = i % n;
        // ---- phase A
        for (;;) {
            if (t < 64) {
                int pos = __builtin_amdgcn_readfirstlane(s_pos);
                int rep = __builtin_amdgcn_readfirstlane(s_rep);
                int i = __builtin_amdgcn_readfirstlane(s_i);
                const unsigned long long lt = (1ull << lane) - 1ull;
                while (pos < 624 && rep < g_end) {
                    // a batch of up to 64 tempered words under one mask regime: word k is accepted iff
                    // (w_k & mask) <= i - c_k, c_k = accepts before k.  Solved as a fixed point of wave ballots:
                    // starting from the optimistic set the iterates alternate between super- and subsets of the
                    // answer and agree with it on a strictly growing prefix.
                    const int cnt = 624 - pos < 64 ? 624 - pos : 64;
                    const uint32_t mask = mt_mask_of((uint32_t)i);
                    const int lim = i - (int)(mask >> 1);  // accepts left before the mask shrinks (or the repeat ends)
                    const bool have = lane < cnt;
                    const uint32_t m = have ? (outw[pos + lane] & mask) : 0xffffffffu;
                    unsigned long long acc = __ballot(have && m <= (uint32_t)i);
                    for (;;) {
                        const int c = __popcll(acc & lt);
                        const unsigned long long nxt = __ballot(have && c < lim && m <= (uint32_t)(i - c));
                        if (nxt == acc) break;
                        acc = nxt;
                    }
                    const int c = __popcll(acc & lt);
                    if ((acc >> lane) & 1ull) jl[(rep - g0) * n + (i - c)] = (unsigned short)m;
                    const int total = __popcll(acc);
                    // the batch ends at the word that exhausted the regime, otherwise all words were consumed
                    const int used = total == lim ? 64 - __builtin_clzll(acc) : cnt;
                    pos += used;
                    i -= total;
                    if (i == 0) {
                        ++rep;
                        i = n - 1;
                    }
                }
                if (lane == 0) {
                    s_pos = pos;
                    s_rep = rep;
                    s_i = i;
                }
            }
            __syncthreads();
            if (s_rep >= g_end) break;
            // block regeneration: key[i] <- key[i+397 mod 624] ^ twist(key[i], key[i+1]) in three dependency-free
            // ranges [0,227) [227,454) [454,623), then word 623
            uint32_t v = 0;
            if (t < 227) v = mt_mix(key[t], key[t + 1], key[t + 397]);
            __syncthreads();
            if (t < 227) key[t] = v;
            __syncthreads();
            if (t < 227) v = mt_mix(key[227 + t], key[228 + t], key[t]);
            __syncthreads();
            if (t < 227) key[227 + t] = v;
            __syncthreads();
            if (t < 169) v = mt_mix(key[454 + t], key[455 + t], key[227 + t]);
            __syncthreads();
            if (t < 169) key[454 + t] = v;
            __syncthreads();
            if (t == 0) {
                key[623] = mt_mix(key[623], key[0], key[396]);
                s_pos = 0;
            }
            __syncthreads();
            for (int i = t; i < 624; i += 256) outw[i] = mt_temper(key[i]);
            __syncthreads();
        }
        // ---- phase B
        if (t < g_end - g0) {
            int* pr = p + (size_t)t * n;
            const unsigned short* jr = jl + (size_t)t * n;
            int i = n - 1;
            for (; i >= 8; i -= 8) {
                int jj[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) jj[u] = jr[i - u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int a = pr[jj[u]], b = pr[i - u];
                    pr[jj[u]] = b;
                    pr[i - u] = a;
                }
            }
            for (; i >= 1; --i) {
                const int j = jr[i];
                const int a = pr[j], b = pr[i];
                pr[j] = b;
                pr[i] = a;
            }
        }
        __syncthreads();
        for (int r = g0; r < g_end; ++r)
            for (int i = t; i < n; i += 256) perm_all[(size_t)r * cap + i] = p[(size_t)(r - g0) * n + i];
        __syncthreads();
    }
    for (int i = t; i < 624; i += 256) st[i] = key[i];
    if (t == 0) st[624] = (uint32_t)s_pos;
}

// blockIdx.y = repeat: perm / pa / pb advance by perm_stride / pts_stride elements per repeat
__global__ void k_permute_points(const int* __restrict__ n_ptr, const int* __restrict__ perm, int perm_stride,
                                 const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ pa,
                                 double* __restrict__ pb, int pts_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_ptr) return;
    perm += (size_t)blockIdx.y * perm_stride;
    pa += (size_t)blockIdx.y * pts_stride;
    pb += (size_t)blockIdx.y * pts_stride;
    const int p = perm[i];
    pa[i * 2] = a[p * 2];
    pa[i * 2 + 1] = a[p * 2 + 1];
    pb[i * 2] = b[p * 2];
    pb[i * 2 + 1] = b[p * 2 + 1];
}

// ================================================================================================
// GRIC
// ================================================================================================
// res = compute_fundamental_residual(F, kp1, kp2), F = KinvT @ E @ Kinv (gric.py:14-37)
struct GricBatch {
    const double* E[MAX_E_BATCH];
};
// blockIdx.y = repeat: kp1 / kp2 advance by pts_stride, res by res_stride elements per repeat
__global__ void k_gric_f_residual(const GricBatch G, const double* __restrict__ KinvT,
                                  const double* __restrict__ Kinv, const int* __restrict__ n_ptr,
                                  const double* __restrict__ kp1, const double* __restrict__ kp2, int pts_stride,
                                  double* __restrict__ res, int res_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_ptr) return;
    const double* E = G.E[blockIdx.y];
    kp1 += (size_t)blockIdx.y * pts_stride;
    kp2 += (size_t)blockIdx.y * pts_stride;
    res += (size_t)blockIdx.y * res_stride;
    double T[9], F[9];
    sm::mul33(KinvT, E, T);
    sm::mul33(T, Kinv, F);
    const double m0[3] = {kp1[i * 2], kp1[i * 2 + 1], 1.0}, m1[3] = {kp2[i * 2], kp2[i * 2 + 1], 1.0};
    double Fm0[3], Ftm1[3];
    for (int r = 0; r < 3; r++) {
        Fm0[r] = F[r * 3] * m0[0] + F[r * 3 + 1] * m0[1] + F[r * 3 + 2] * m0[2];
        Ftm1[r] = F[r] * m1[0] + F[3 + r] * m1[1] + F[6 + r] * m1[2];
    }
    const double m1Fm0 = Fm0[0] * m1[0] + Fm0[1] * m1[1] + Fm0[2] * m1[2];
    res[i] = m1Fm0 * m1Fm0 / ((Fm0[0] * Fm0[0] + Fm0[1] * Fm0[1]) + (Ftm1[0] * Ftm1[0] + Ftm1[1] * Ftm1[1]));
}

// res = compute_homography_residual(H, kp1, kp2) (gric.py:40-92)
__global__ void k_gric_h_residual(const double* __restrict__ Hm, const int* __restrict__ n_ptr,
                                  const double* __restrict__ kp1, const double* __restrict__ kp2,
                                  double* __restrict__ res) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_ptr) return;
    const double* H = Hm;
    const double m0x = kp1[i * 2], m0y = kp1[i * 2 + 1], m1x = kp2[i * 2], m1y = kp2[i * 2 + 1];
    const double G00 = H[0] - m1x * H[6], G01 = H[1] - m1x * H[7], G02 = -m0x * H[6] - m0y * H[7] - H[8];
    const double G10 = H[3] - m1y * H[6], G11 = H[4] - m1y * H[7], G12 = -m0x * H[6] - m0y * H[7] - H[8];
    const double magG0 = sqrt(G00 * G00 + G01 * G01 + G02 * G02);
    const double magG1 = sqrt(G10 * G10 + G11 * G11 + G12 * G12);
    const double magG0G1 = G00 * G10 + G01 * G11;
    const double alpha = acos(magG0G1 / (magG0 * magG1));
    const double alg0 = m0x * H[0] + m0y * H[1] + H[2] - m1x * (m0x * H[6] + m0y * H[7] + H[8]);
    const double alg1 = m0x * H[3] + m0y * H[4] + H[5] - m1y * (m0x * H[6] + m0y * H[7] + H[8]);
    const double D1 = alg0 / magG0, D2 = alg1 / magG1;
    res[i] = (D1 * D1 + D2 * D2 - 2.0 * D1 * D2 * cos(alpha)) / sin(alpha);
}

// calc_GRIC (gric.py:95-132): the python loop's sequential sum, one lane; residuals staged through LDS
// blockIdx.x = problem: res advances by res_stride, out by one element per problem
__global__ __launch_bounds__(256) void k_gric_sum(const double* __restrict__ res, int res_stride,
                                                   const int* __restrict__ n_ptr, double sigma, int Kp, int D,
                                                   double* __restrict__ out) {
    __shared__ double s_res[2048];
    const int n = *n_ptr;
    res += (size_t)blockIdx.x * res_stride;
    out += blockIdx.x;
    const double R = 4, sigmasq1 = 1. / (sigma * sigma);
    const double lam3RD = 2.0 * (R - D);
    double sum = 0;
    for (int c0 = 0; c0 < n; c0 += 2048) {
        const int cnt = n - c0 < 2048 ? n - c0 : 2048;
        for (int i = threadIdx.x; i < cnt; i += 256) s_res[i] = res[c0 + i];
        __syncthreads();
        if (threadIdx.x == 0) {  // the reference's sequential np.sum order; terms batched so that LDS latency overlaps
            int i = 0;
            for (; i + 8 <= cnt; i += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double tmp = s_res[i + u] * sigmasq1;
                    v[u] = tmp <= lam3RD ? tmp : lam3RD;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += v[u];
            }
            for (; i < cnt; i++) {
                const double tmp = s_res[i] * sigmasq1;
                sum += tmp <= lam3RD ? tmp : lam3RD;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sum += n * D * log(R) + Kp * log(R * n);
        *out = sum;
    }
}

// residual + calc_GRIC in one launch (the two kernels above, for the fused pipeline): the residual of point i is
// produced straight into the LDS staging buffer the sequential sum reads.  blockIdx.x = problem.
struct GricFusedBatch {
    const double* M[MAX_E_BATCH];  // E (mode 0) or H (mode 1) per problem
};
__global__ __launch_bounds__(256) void k_gric_fused(const GricFusedBatch G, int mode, const double* __restrict__ KinvT,
                                                     const double* __restrict__ Kinv, const int* __restrict__ n_ptr,
                                                     const double* __restrict__ kp1, const double* __restrict__ kp2,
                                                     int pts_stride, double sigma, int Kp, int D,
                                                     double* __restrict__ out) {
    __shared__ double s_res[2048];
    __shared__ double s_M[9];
    const int n = *n_ptr;
    const double* Min = G.M[blockIdx.x];
    kp1 += (size_t)blockIdx.x * pts_stride;
    kp2 += (size_t)blockIdx.x * pts_stride;
    out += blockIdx.x;
    if (threadIdx.x == 0) {
        if (mode == 0) {  // F = K^-T E K^-1
            double T[9], F[9];
            sm::mul33(KinvT, Min, T);
            sm::mul33(T, Kinv, F);
            for (int k = 0; k < 9; k++) s_M[k] = F[k];
        } else {
            for (int k = 0; k < 9; k++) s_M[k] = Min[k];
        }
    }
    __syncthreads();
    const double R = 4, sigmasq1 = 1. / (sigma * sigma);
    const double lam3RD = 2.0 * (R - D);
    double sum = 0;
    for (int c0 = 0; c0 < n; c0 += 2048) {
        const int cnt = n - c0 < 2048 ? n - c0 : 2048;
        for (int k = threadIdx.x; k < cnt; k += 256) {
            const int i = c0 + k;
            double r;
            if (mode == 0) {
                const double* F = s_M;
                const double m0[3] = {kp1[i * 2], kp1[i * 2 + 1], 1.0}, m1[3] = {kp2[i * 2], kp2[i * 2 + 1], 1.0};
                double Fm0[3], Ftm1[3];
                for (int q = 0; q < 3; q++) {
                    Fm0[q] = F[q * 3] * m0[0] + F[q * 3 + 1] * m0[1] + F[q * 3 + 2] * m0[2];
                    Ftm1[q] = F[q] * m1[0] + F[3 + q] * m1[1] + F[6 + q] * m1[2];
                }
                const double m1Fm0 = Fm0[0] * m1[0] + Fm0[1] * m1[1] + Fm0[2] * m1[2];
                r = m1Fm0 * m1Fm0 / ((Fm0[0] * Fm0[0] + Fm0[1] * Fm0[1]) + (Ftm1[0] * Ftm1[0] + Ftm1[1] * Ftm1[1]));
            } else {
                const double* H = s_M;
                const double m0x = kp1[i * 2], m0y = kp1[i * 2 + 1], m1x = kp2[i * 2], m1y = kp2[i * 2 + 1];
                const double G00 = H[0] - m1x * H[6], G01 = H[1] - m1x * H[7], G02 = -m0x * H[6] - m0y * H[7] - H[8];
                const double G10 = H[3] - m1y * H[6], G11 = H[4] - m1y * H[7], G12 = -m0x * H[6] - m0y * H[7] - H[8];
                const double magG0 = sqrt(G00 * G00 + G01 * G01 + G02 * G02);
                const double magG1 = sqrt(G10 * G10 + G11 * G11 + G12 * G12);
                const double magG0G1 = G00 * G10 + G01 * G11;
                const double alpha = acos(magG0G1 / (magG0 * magG1));
                const double alg0 = m0x * H[0] + m0y * H[1] + H[2] - m1x * (m0x * H[6] + m0y * H[7] + H[8]);
                const double alg1 = m0x * H[3] + m0y * H[4] + H[5] - m1y * (m0x * H[6] + m0y * H[7] + H[8]);
                const double D1 = alg0 / magG0, D2 = alg1 / magG1;
                r = (D1 * D1 + D2 * D2 - 2.0 * D1 * D2 * cos(alpha)) / sin(alpha);
            }
            s_res[k] = r;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int i = 0;
            for (; i + 8 <= cnt; i += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double tmp = s_res[i + u] * sigmasq1;
                    v[u] = tmp <= lam3RD ? tmp : lam3RD;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += v[u];
            }
            for (; i < cnt; i++) {
                const double tmp = s_res[i] * sigmasq1;
                sum += tmp <= lam3RD ? tmp : lam3RD;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sum += n * D * log(R) + Kp * log(R * n);
        *out = sum;
    }
}

// ================================================================================================
// compute_pose_2d2d bookkeeping
// ================================================================================================
__global__ void k_pose_state_init(PoseState* ps, const int* __restrict__ n_ptr, uint8_t* __restrict__ best_inliers,
                                  int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) best_inliers[i] = 1;  // np.ones((N,1)) == 1
    if (i == 0) {
        ps->best_cnt = 0;
        ps->num_valid = 0;
        ps->have_best = 0;
        ps->h_gric = 0;
        ps->n = *n_ptr;
        for (int k = 0; k < 9; k++) ps->best_E[k] = 0;
        for (int k = 0; k < 9; k++) ps->R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        ps->t[0] = ps->t[1] = ps->t[2] = 0;
        ps->cheirality = 0;
        ps->valid_case = 1;
        ps->major_valid = 0;
        ps->h_found = 0;
        for (int k = 0; k < 8; k++) {
            ps->rep_cnt[k] = 0;
            ps->rep_valid[k] = 0;
            ps->rep_gric[k] = 0;
        }
    }
}

__global__ void k_set_h_gric(PoseState* ps, const RansacState* hst, const double* __restrict__ gric) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ps->h_found = hst->found;
    ps->h_gric = hst->found ? *gric : INFINITY;
}

// after one findEssentialMat repeat: valid_case, inlier_check, un-permuted mask (E_tracker.py:258-285)
__global__ __launch_bounds__(256) void k_rep_update(PoseState* ps, const RansacState* est, const double* __restrict__ E,
                                                     const double* __restrict__ e_gric,
                                                     const uint8_t* __restrict__ mask, const int* __restrict__ perm,
                                                     uint8_t* __restrict__ best_inliers, int rep) {
    __shared__ int s_take;
    const int n = ps->n;
    if (threadIdx.x == 0) {
        const int found = est->found;
        const int cnt = found ? est->max_good : 0;
        const bool valid = found && (ps->h_gric > *e_gric);
        ps->rep_cnt[rep] = cnt;
        ps->rep_valid[rep] = valid ? 1 : 0;
        ps->rep_gric[rep] = found ? *e_gric : INFINITY;
        ps->num_valid += valid ? 1 : 0;
        s_take = (found && cnt > ps->best_cnt) ? 1 : 0;
        if (s_take) {
            ps->best_cnt = cnt;
            ps->have_best = 1;
            for (int k = 0; k < 9; k++) ps->best_E[k] = E[k];
        }
    }
    __syncthreads();
    if (!s_take) return;
    for (int c = threadIdx.x; c < n; c += blockDim.x) best_inliers[perm[c]] = mask[c];
}

__global__ void k_pose_finish(PoseState* ps, const double* __restrict__ rp_out, int repeat, int stage) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (stage == 0) {
        // major_valid = num_valid_case > (max_ransac_iter / 2)
        ps->major_valid = ((double)ps->num_valid > (double)repeat / 2.0 && ps->have_best) ? 1 : 0;
        return;
    }
    if (!ps->major_valid) return;
    const int good = (int)rp_out[12];
    ps->cheirality = good;
    if ((double)good > (double)ps->n * 0.1) {
        for (int k = 0; k < 9; k++) ps->R[k] = rp_out[k];
        for (int k = 0; k < 3; k++) ps->t[k] = rp_out[9 + k];
    }
}

// the whole post-RANSAC bookkeeping of compute_pose_2d2d in one launch: H validity (k_set_h_gric), the `repeat`
// sequential k_rep_update steps, and the major_valid decision (k_pose_finish stage 0)
struct RepBatch {
    const RansacState* st[MAX_REP];
    const double* E[MAX_REP];
    const uint8_t* mask[MAX_REP];
};
__global__ __launch_bounds__(256) void k_rep_update_all(PoseState* ps, const RansacState* hst, const double* __restrict__ h_gric,
                                                         const RepBatch B, const double* __restrict__ e_gric,
                                                         const int* __restrict__ perm, int perm_stride,
                                                         uint8_t* __restrict__ best_inliers, int repeat,
                                                         int by_ratio, double ratio_thre) {
    // by_ratio (validity.method 'homo_ratio', E_tracker.py:186-194,243-250): a repeat is valid while
    // H_inliers.sum() / (H_inliers.sum() + inliers.sum()) < thre (0 / 0 = nan compares false, like numpy);
    // h_gric then carries the homography's inlier count and rep_gric[] the ratios
    __shared__ int s_take;
    const int n = ps->n;
    if (threadIdx.x == 0) {
        ps->h_found = hst->found;
        if (by_ratio)
            ps->h_gric = hst->found ? (double)hst->max_good : 0.0;  // no model: OpenCV returns an all-zero mask
        else
            ps->h_gric = hst->found ? *h_gric : INFINITY;
    }
    __syncthreads();
    for (int rep = 0; rep < repeat; ++rep) {
        if (threadIdx.x == 0) {
            const RansacState* est = B.st[rep];
            const int found = est->found;
            const int cnt = found ? est->max_good : 0;
            bool valid;
            double crit;
            if (by_ratio) {
                crit = ps->h_gric / (ps->h_gric + (double)cnt);
                valid = crit < ratio_thre;
            } else {
                crit = found ? e_gric[rep] : INFINITY;
                valid = found && (ps->h_gric > e_gric[rep]);
            }
            ps->rep_cnt[rep] = cnt;
            ps->rep_valid[rep] = valid ? 1 : 0;
            ps->rep_gric[rep] = crit;
            ps->num_valid += valid ? 1 : 0;
            s_take = (found && cnt > ps->best_cnt) ? 1 : 0;
            if (s_take) {
                ps->best_cnt = cnt;
                ps->have_best = 1;
                for (int k = 0; k < 9; k++) ps->best_E[k] = B.E[rep][k];
            }
        }
        __syncthreads();
        if (s_take) {
            const uint8_t* mask = B.mask[rep];
            const int* pm = perm + (size_t)rep * perm_stride;
            for (int c = threadIdx.x; c < n; c += blockDim.x) best_inliers[pm[c]] = mask[c];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        ps->major_valid = ((double)ps->num_valid > (double)repeat / 2.0 && ps->have_best) ? 1 : 0;
}

// ---- e_tracker.validity.method == "flow" (ablation_model_sel_flow.yml) ------------------------------------------
// valid_case = np.mean(np.linalg.norm(kp_ref - kp_cur, axis=1)) > thre (E_tracker.py:182-185).  gate[0] = the keypoint
// count the shuffles see (n when the pair is tracked, 0 otherwise: a closed gate draws nothing from np.random),
// gate[1] = valid_case; the mean goes to *avg_out
__global__ __launch_bounds__(256) void k_flow_gate(const int* __restrict__ kp_info, const double* __restrict__ kp_ref,
                                                    const double* __restrict__ kp_cur, double thre, double* __restrict__ norms,
                                                    double* __restrict__ avg_out, int* __restrict__ gate) {
    const int n = kp_info[0];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double dx = kp_ref[i * 2] - kp_cur[i * 2], dy = kp_ref[i * 2 + 1] - kp_cur[i * 2 + 1];
        norms[i] = sqrt(dx * dx + dy * dy);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double avg = sm::np_pairwise_sum(norms, n) / (double)n;  // n == 0: nan, compares false like numpy's
        *avg_out = avg;
        const int open = avg > thre ? 1 : 0;
        gate[0] = open ? n : 0;
        gate[1] = open;
    }
}

__global__ void k_copy_double(double* __restrict__ dst, const double* __restrict__ src) { *dst = *src; }

// the per-repeat bookkeeping of the "flow" validity: validity of a repeat = its recoverPose cheirality count above
// 10 % of the keypoints, best model = most RANSAC inliers among the repeats whose count is above 5 %
__global__ __launch_bounds__(256) void k_rep_update_flow(PoseState* ps, const int* __restrict__ gate, const double* __restrict__ avg_flow,
                                                          const RepBatch B, const double* __restrict__ cheir,
                                                          const int* __restrict__ perm, int perm_stride,
                                                          uint8_t* __restrict__ best_inliers, int repeat) {
    __shared__ int s_take;
    const int n = ps->n;
    const int open = gate[1];
    if (threadIdx.x == 0) {
        ps->h_found = 0;
        ps->h_gric = *avg_flow;
    }
    __syncthreads();
    for (int rep = 0; rep < repeat; ++rep) {
        if (threadIdx.x == 0) {
            const RansacState* est = B.st[rep];
            const int found = open && est->found;
            const int cnt = found ? est->max_good : 0;
            const double c = found ? cheir[rep] : 0.0;
            const bool valid = found && c > (double)n * 0.1;
            ps->rep_cnt[rep] = cnt;
            ps->rep_valid[rep] = valid ? 1 : 0;
            ps->rep_gric[rep] = c;
            ps->num_valid += valid ? 1 : 0;
            s_take = (found && cnt > ps->best_cnt && c > (double)n * 0.05) ? 1 : 0;
            if (s_take) {
                ps->best_cnt = cnt;
                ps->have_best = 1;
                for (int k = 0; k < 9; k++) ps->best_E[k] = B.E[rep][k];
            }
        }
        __syncthreads();
        if (s_take) {
            const uint8_t* mask = B.mask[rep];
            const int* pm = perm + (size_t)rep * perm_stride;
            for (int c = threadIdx.x; c < n; c += blockDim.x) best_inliers[pm[c]] = mask[c];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        ps->major_valid = ((double)ps->num_valid > (double)repeat / 2.0 && ps->have_best) ? 1 : 0;
}

// ================================================================================================
// scale recovery (find_scale_from_depth)
// ================================================================================================
// per keypoint: normalise, triangulate with [I|0] / T_21, X2 = T_21[:3] @ (X / X[3]); target pixel of kp2
__global__ void k_scale_triangulate(const int* __restrict__ n_ptr, const double* __restrict__ kp1,
                                    const double* __restrict__ kp2, const double* __restrict__ T21, double cx, double cy,
                                    double fx, double fy, int H, int W, double* __restrict__ z2,
                                    int* __restrict__ pix, int* __restrict__ winner) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_ptr) return;
    const double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double P2[12];
    for (int k = 0; k < 12; k++) P2[k] = T21[k];
    const double x1 = (kp1[i * 2] - cx) / fx, y1 = (kp1[i * 2 + 1] - cy) / fy;
    const double x2 = (kp2[i * 2] - cx) / fx, y2 = (kp2[i * 2 + 1] - cy) / fy;
    double X[4];
    sm::triangulate_point(P1, P2, x1, y1, x2, y2, X);
    const double w = X[3];
    const double Xn[4] = {X[0] / w, X[1] / w, X[2] / w, X[3] / w};
    double z = 0;
    z = P2[8] * Xn[0] + P2[9] * Xn[1] + P2[10] * Xn[2] + P2[11] * Xn[3];
    z2[i] = z;
    // kp.astype(np.int): truncation toward zero
    const double kx = kp2[i * 2], ky = kp2[i * 2 + 1];
    const int ix = (int)kx, iy = (int)ky;
    int p = -1;
    if (ix >= 0 && ix < W && iy >= 0 && iy < H && kx == kx && ky == ky) p = iy * W + ix;
    pix[i] = p;
    if (p >= 0) atomicMax(&winner[p], i);  // numpy fancy assignment: the last index wins
}

// ordered list of depth ratios: pixels (row-major) whose winning keypoint has tri > 0 and CNN depth > 0
__global__ __launch_bounds__(256) void k_scale_ratios(const int* __restrict__ n_ptr, const double* __restrict__ z2,
                                                       const int* __restrict__ pix, const int* __restrict__ winner,
                                                       const double* __restrict__ depth, double* __restrict__ ratios,
                                                       int* __restrict__ n_valid, double* __restrict__ tri_list,
                                                       double* __restrict__ pred_list, int depth_per_kp) {
    // single block: n <= a few thousand.  rank = number of valid entries with a smaller pixel index.
    // depth_per_kp: `depth` holds the depth map's value at keypoint i's pixel, [n], instead of the map (the only pixels read)
    extern __shared__ int s_pix[];
    const int n = *n_ptr;
    const int t = threadIdx.x;
    for (int i = t; i < n; i += blockDim.x) {
        const int p = pix[i];
        bool ok = p >= 0 && winner[p] == i;
        if (ok) {
            double tri = z2[i];
            if (tri < 0) tri = 0;  // depth2_tri[depth2_tri < 0] = 0 (NaN stays NaN and fails > 0)
            ok = (tri > 0) && (depth[depth_per_kp ? i : p] > 0);
        }
        s_pix[i] = ok ? p : -1;
    }
    __syncthreads();
    int local = 0;
    for (int i = t; i < n; i += blockDim.x) {
        const int p = s_pix[i];
        if (p < 0) continue;
        int rank = 0;
        for (int j = 0; j < n; j++) rank += (s_pix[j] >= 0 && s_pix[j] < p) ? 1 : 0;
        const double dp = depth[depth_per_kp ? i : p];
        ratios[rank] = z2[i] / dp;
        if (tri_list) {  // ransac.method 'abs_diff': the regression runs on the two depths themselves
            tri_list[rank] = z2[i];
            pred_list[rank] = dp;
        }
        local++;
    }
    const int s = wave_sum_i(local);
    if ((t & 63) == 0 && s) atomicAdd(n_valid, s);
}

// sklearn RANSACRegressor(LinearRegression(fit_intercept=False), min_samples, max_trials, stop_probability,
// residual_threshold).fit(ratio.reshape(-1,1), ones) -> estimator_.coef_[0,0]; one 256-thread block.
// yv != nullptr (ransac.method 'abs_diff', E_tracker.py:631-635): .fit(depth_tri, depth_pred) -- the same loop with a
// general target: least squares through the origin sum(xy)/sum(xx), residual |y - x coef|, and the tie-break score is
// the real r2_score of the inlier set (1 - ss_res / ss_tot; constant target: 1 when ss_res == 0, else 0).
__global__ __launch_bounds__(256) void k_scale_ransac(uint32_t* __restrict__ mt_state, const double* __restrict__ x,
                                                       const double* __restrict__ yv, const int* __restrict__ n_valid, int min_valid, int min_samples,
                                                       int max_trials, double stop_prob, double thr,
                                                       uint8_t* __restrict__ inl_a, uint8_t* __restrict__ inl_b,
                                                       int* __restrict__ scratch, ScaleResult* __restrict__ out,
                                                       const PoseState* __restrict__ gate, int r2_nan_below_two) {
    __shared__ sm::Mt19937 s;
    __shared__ double s_coef;
    __shared__ int s_cnt[4], s_nz[4];
    __shared__ double s_red[3][4];
    __shared__ int s_ctl;  // 0 continue, 1 stop
    __shared__ int s_best_is_a;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = *n_valid;
    // fused pipeline: scale recovery only runs when ||t|| != 0 (dfvo.py:198); a rejected E-tracker pose must not
    // draw from the numpy stream
    const bool gated = gate && gate->t[0] == 0 && gate->t[1] == 0 && gate->t[2] == 0;
    if (gated || !(n > min_valid)) {  // valid_mask2.sum() > 10
        if (t == 0) {
            out->scale = -1.0;
            out->n_valid = n;
            out->n_trials = 0;
            out->n_inliers = 0;
            out->status = 0;
        }
        return;
    }
    for (int i = t; i < 624; i += 256) s.key[i] = mt_state[i];
    if (t == 0) s.pos = (int)mt_state[624];
    __syncthreads();
    int n_inliers_best = 1, n_trials = 0, trials_cap = max_trials;
    double score_best = -INFINITY;
    bool have_best = false;
    int best_is_a = 0;
    for (;;) {
        if (!(n_trials < trials_cap)) break;
        n_trials++;
        if (t == 0) {
            int idx[8];
            sm::mt_sample_without_replacement(s, n, min_samples, idx, scratch);
            // LinearRegression(fit_intercept=False) on (x_subset, ones): least squares through the origin
            double sx = 0, sxx = 0;
            for (int k = 0; k < min_samples; k++) {
                sx += x[idx[k]] * (yv ? yv[idx[k]] : 1.0);
                sxx += x[idx[k]] * x[idx[k]];
            }
            s_coef = sx / sxx;
        }
        __syncthreads();
        const double coef = s_coef;
        uint8_t* cur = best_is_a ? inl_b : inl_a;  // write the candidate mask into the non-best buffer
        int c = 0, nz = 0;
        double sy = 0;
        for (int i = t; i < n; i += 256) {
            const double pred = x[i] * coef;
            const double yi = yv ? yv[i] : 1.0;
            const double r = fabs(yi - pred);
            const int f = r <= thr ? 1 : 0;
            cur[i] = (uint8_t)f;
            c += f;
            nz += (f && (yi - pred) != 0.0) ? 1 : 0;  // r2_score numerator != 0 on the inlier set
            if (f) sy += yi;
        }
        c = wave_sum_i(c);
        nz = wave_sum_i(nz);
        if (yv) sy = wave_sum_d(sy);
        if (lane == 0) {
            s_cnt[wave] = c;
            s_nz[wave] = nz;
            s_red[0][wave] = sy;
        }
        __syncthreads();
        const int n_in = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        const int nzs = s_nz[0] + s_nz[1] + s_nz[2] + s_nz[3];
        const double mean_y = (s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3]) / (double)n_in;
        __syncthreads();
        if (n_in < n_inliers_best) continue;  // n_skips_no_inliers_
        // r2_score with constant y_true: 1.0 when the residual sum is zero, else 0.0
        double score = nzs == 0 ? 1.0 : 0.0;
        if (yv) {
            double res = 0, tot = 0;
            for (int i = t; i < n; i += 256)
                if (cur[i]) {
                    const double d = yv[i] - x[i] * coef, e = yv[i] - mean_y;
                    res += d * d;
                    tot += e * e;
                }
            res = wave_sum_d(res);
            tot = wave_sum_d(tot);
            if (lane == 0) {
                s_red[1][wave] = res;
                s_red[2][wave] = tot;
            }
            __syncthreads();
            res = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
            tot = s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3];
            __syncthreads();
            if (tot != 0.0) score = 1.0 - res / tot;  // else: the constant-target rule above (force_finite)
        }
        // sklearn >= 0.22: r2_score of fewer than two samples is nan, which loses no comparison below.  scikit-learn 0.20.3
        // (the reference's pin, envs/requirement.yml:233) has no such rule: one sample is a constant target, scored 1.0 /
        // 0.0 by the rule above.  dfvo_set_sklearn_compat selects (default: the reference's pin).
        if (r2_nan_below_two && n_in < 2) score = NAN;
        if (n_in == n_inliers_best && score < score_best) continue;
        n_inliers_best = n_in;
        score_best = score;
        have_best = true;
        best_is_a = best_is_a ? 0 : 1;  // the buffer just written becomes the best
        // _dynamic_max_trials
        {
            const double eps = 2.220446049250313e-16;
            const double ratio = (double)n_in / (double)n;
            double nom = 1 - stop_prob;
            nom = nom > eps ? nom : eps;
            double denom = 1 - pow(ratio, (double)min_samples);
            denom = denom > eps ? denom : eps;
            double dyn;
            if (nom == 1)
                dyn = 0;
            else if (denom == 1)
                dyn = INFINITY;
            else
                dyn = fabs(ceil(log(nom) / log(denom)));
            if (dyn < (double)trials_cap) trials_cap = (int)dyn;
        }
    }
    // final fit on the best inliers (in index order)
    if (t == 0) {
        double scale = -1.0;
        int status = 0;
        if (have_best) {
            const uint8_t* best = best_is_a ? inl_a : inl_b;
            double sx = 0, sxx = 0;
            for (int i = 0; i < n; i++)
                if (best[i]) {
                    sx += x[i] * (yv ? yv[i] : 1.0);
                    sxx += x[i] * x[i];
                }
            scale = sx / sxx;
            status = 1;
        } else {
            status = -1;  // sklearn raises ValueError: no valid consensus set
        }
        out->scale = scale;
        out->n_valid = n;
        out->n_trials = n_trials;
        out->n_inliers = have_best ? n_inliers_best : 0;
        out->status = status;
        out->best_is_a = best_is_a;
    }
    __syncthreads();
    for (int i = t; i < 624; i += 256) mt_state[i] = s.key[i];
    if (t == 0) mt_state[624] = (uint32_t)s.pos;
    (void)s_ctl;
    (void)s_best_is_a;
}

// ================================================================================================
// host-side enqueue helpers
// ================================================================================================
int TrackerBuffers::ensure_kp(int cap, int cells, int n_best) {
    if (cap <= kp_cap && cells * n_best <= sel_cap) return DFVO_OK;
    release_kp();
    kp_cap = cap > kp_cap ? cap : kp_cap;
    sel_cap = cells * n_best > sel_cap ? cells * n_best : sel_cap;
    DFVO_HIP_CHECK(hipMalloc((void**)&kp_ref, sizeof(double) * 2 * kp_cap));
    DFVO_HIP_CHECK(hipMalloc((void**)&kp_cur, sizeof(double) * 2 * kp_cap));
    DFVO_HIP_CHECK(hipMalloc((void**)&pa, sizeof(double) * 2 * kp_cap * MAX_REP));
    DFVO_HIP_CHECK(hipMalloc((void**)&pb, sizeof(double) * 2 * kp_cap * MAX_REP));
    DFVO_HIP_CHECK(hipMalloc((void**)&perm, sizeof(int) * (size_t)(kp_cap + 8) * MAX_REP));
    DFVO_HIP_CHECK(hipMalloc((void**)&res, sizeof(double) * kp_cap * (MAX_REP + 1)));
    DFVO_HIP_CHECK(hipMalloc((void**)&best_inliers, kp_cap + 8));
    DFVO_HIP_CHECK(hipMalloc((void**)&cell_count, sizeof(int) * 1024));
    DFVO_HIP_CHECK(hipMalloc((void**)&cell_sel, sizeof(int) * sel_cap));
    DFVO_HIP_CHECK(hipMalloc((void**)&z2, sizeof(double) * kp_cap));
    DFVO_HIP_CHECK(hipMalloc((void**)&pix, sizeof(int) * kp_cap));
    DFVO_HIP_CHECK(hipMalloc((void**)&ratios, sizeof(double) * kp_cap * 3));  // ratio | triangulated | CNN depth lists
    DFVO_HIP_CHECK(hipMalloc((void**)&inl_a, kp_cap + 8));
    DFVO_HIP_CHECK(hipMalloc((void**)&inl_b, kp_cap + 8));
    DFVO_HIP_CHECK(hipMalloc((void**)&scratch, sizeof(int) * (kp_cap + 8)));
    return DFVO_OK;
}

void TrackerBuffers::release_kp() {
    void* ptrs[] = {kp_ref, kp_cur, pa, pb, perm, res, best_inliers, cell_count, cell_sel, z2, pix, ratios, inl_a, inl_b, scratch};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    kp_ref = kp_cur = pa = pb = res = z2 = ratios = nullptr;
    perm = cell_count = cell_sel = pix = scratch = nullptr;
    best_inliers = inl_a = inl_b = nullptr;
    kp_cap = sel_cap = 0;
}

int TrackerBuffers::init(hipStream_t rep0, hipStream_t rep1) {
    DFVO_HIP_CHECK(hipMalloc((void**)&mt_state, sizeof(uint32_t) * 640));
    DFVO_HIP_CHECK(hipMalloc((void**)&kp_info, sizeof(int) * 8));
    DFVO_HIP_CHECK(hipMalloc((void**)&kp_total, sizeof(int) * (8 + 64 + 2)));  // [0..7] counters (5, 6: the flow gate), [8..71] k_kp_cell's partial counts, [72..73] the five-point sampler's state behind its prefetched subsets
    DFVO_HIP_CHECK(hipMalloc((void**)&pose, sizeof(PoseState)));
    DFVO_HIP_CHECK(hipMalloc((void**)&small, sizeof(double) * 128));
    DFVO_HIP_CHECK(hipMalloc((void**)&scale_out, sizeof(ScaleResult)));
    DFVO_HIP_CHECK(hipMemset(kp_info, 0, sizeof(int) * 8));
    // Side streams: [0] runs the five-point batch, [1] the scale stage's fills; the slots past `n_streams` alias them.
    // How many streams are CREATED here matters although only two are used: the hardware queue a stream gets (and with it
    // the compute pipe that dispatches it) follows the creation order, the fused pipeline creates its two prefetch
    // streams after these, and the pair rate depends on which pipes the prefetch chain shares with the flow nets / the
    // RNG-dependent solver chain.  Measured on MI355X, bench.py order (pipeline created before the process touches the
    // GPU through torch), exact fp32: 2 -> 103, 3 -> 108, 4 -> 133, 5 -> 111, 6 -> 112, 7 -> 116, 8 -> 133 frames/s;
    // with a torch copy issued first the fast settings are 5 .. 7 (126).  DFVO_REP_STREAMS overrides (tuning aid).
    // (The fused pipeline no longer depends on this: it measures which streams share a pipe and passes rep0 / rep1 in,
    // stream_pool.hip.)
    const int n_streams = rep0 ? 2 : rep_stream_count();
    n_rep_owned = n_streams;
    for (int r = 0; r < MAX_REP; r++) {
        if (rep0 && r < 2)
            s_rep[r] = r == 0 ? rep0 : (rep1 ? rep1 : rep0);
        else if (r < n_streams)
            DFVO_HIP_CHECK(create_solver_stream(&s_rep[r], 2));
        else
            s_rep[r] = s_rep[r % n_streams];
        DFVO_HIP_CHECK(hipEventCreateWithFlags(&ev_rep[r], hipEventDisableTiming));
    }
    DFVO_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    DFVO_HIP_CHECK(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
    DFVO_HIP_CHECK(hipEventCreateWithFlags(&ev_h, hipEventDisableTiming));
    return DFVO_OK;
}

int TrackerBuffers::rebind_streams(hipStream_t rep0, hipStream_t rep1) {
    DFVO_ARG_CHECK(!shared && rep0 && rep1 && rep0 != rep1, "TrackerBuffers::rebind_streams: bad argument");
    DFVO_HIP_CHECK(hipDeviceSynchronize());
    for (int r = 0; r < n_rep_owned && r < MAX_REP; r++)
        if (s_rep[r] && !(r == 1 && s_rep[1] == s_rep[0])) (void)hipStreamDestroy(s_rep[r]);
    n_rep_owned = 2;
    for (int r = 0; r < MAX_REP; r++) s_rep[r] = (r & 1) ? rep1 : rep0;
    return DFVO_OK;
}

int TrackerBuffers::init_shared(const TrackerBuffers& first) {
    shared = true;
    mt_state = first.mt_state;
    for (int r = 0; r < MAX_REP; r++) {
        s_rep[r] = first.s_rep[r];
        ev_rep[r] = first.ev_rep[r];
    }
    ev_fork = first.ev_fork;
    DFVO_HIP_CHECK(hipMalloc((void**)&kp_info, sizeof(int) * 8));
    DFVO_HIP_CHECK(hipMalloc((void**)&kp_total, sizeof(int) * (8 + 64 + 2)));  // [0..7] counters (5, 6: the flow gate), [8..71] k_kp_cell's partial counts, [72..73] the five-point sampler's state behind its prefetched subsets
    DFVO_HIP_CHECK(hipMalloc((void**)&pose, sizeof(PoseState)));
    DFVO_HIP_CHECK(hipMalloc((void**)&small, sizeof(double) * 128));
    DFVO_HIP_CHECK(hipMalloc((void**)&scale_out, sizeof(ScaleResult)));
    DFVO_HIP_CHECK(hipMemset(kp_info, 0, sizeof(int) * 8));
    DFVO_HIP_CHECK(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
    DFVO_HIP_CHECK(hipEventCreateWithFlags(&ev_h, hipEventDisableTiming));
    return DFVO_OK;
}

void TrackerBuffers::release() {
    release_kp();
    ws_h.release();
    ws_e.release();
    for (int r = 0; r < MAX_REP; r++) {
        ws_rep[r].release();
        if (!shared) {
            if (s_rep[r] && r < n_rep_owned && !(r == 1 && s_rep[1] == s_rep[0])) (void)hipStreamDestroy(s_rep[r]);
            if (ev_rep[r]) (void)hipEventDestroy(ev_rep[r]);
        }
        s_rep[r] = nullptr;
        ev_rep[r] = nullptr;
    }
    if (ev_fork && !shared) (void)hipEventDestroy(ev_fork);
    if (ev_start) (void)hipEventDestroy(ev_start);
    if (ev_h) (void)hipEventDestroy(ev_h);
    ev_fork = ev_start = ev_h = nullptr;
    for (int i = 0; i < 4; i++) {
        if (ev_t[i]) (void)hipEventDestroy(ev_t[i]);
        ev_t[i] = nullptr;
    }
    for (int i = 0; i < N_SEG; i++) {
        if (ev_seg[i]) (void)hipEventDestroy(ev_seg[i]);
        ev_seg[i] = nullptr;
    }
    if (shared) mt_state = nullptr;
    small_valid = false;
    if (ratio_map) (void)hipFree(ratio_map);
    ratio_map = nullptr;
    ratio_cap = 0;
    void* ptrs[] = {mt_state, kp_info, kp_total, pose, small, scale_out, winner, lidx};
    lidx = nullptr;
    lidx_cap = 0;
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    mt_state = nullptr;
    kp_info = kp_total = winner = nullptr;
    pose = nullptr;
    small = nullptr;
    scale_out = nullptr;
    winner_cap = 0;
}

int enqueue_mt_seed(TrackerBuffers& tb, uint32_t seed, hipStream_t s) {
    hipLaunchKernelGGL(k_mt_seed, dim3(1), dim3(1), 0, s, tb.mt_state, seed);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// `repeat` x (perm = np.arange(n); np.random.shuffle(perm)) from the device-resident numpy stream `mt_state`;
// n = *d_n on the device (n_host bounds it), perm[r * perm_stride + i]
int enqueue_mt_shuffle(uint32_t* mt_state, const int* d_n, int n_host, int repeat, int perm_stride, int* perm,
                       hipStream_t s) {
    const size_t per_rep = 6 * (size_t)(n_host > 0 ? n_host : 1);  // int permutation + uint16 draw list
    DFVO_ARG_CHECK(per_rep <= 144 * 1024, "shuffle: too many keypoints for the LDS permutation buffer");
    int group = (int)((144 * 1024) / per_rep);
    if (group > repeat) group = repeat;
    const size_t perm_lds = per_rep * group + 16;
    if (int rc_lds = ensure_dyn_lds((const void*)k_mt_shuffle_all, perm_lds)) return rc_lds;
    hipLaunchKernelGGL(k_mt_shuffle_all, dim3(1), dim3(256), perm_lds, s, mt_state, d_n, repeat, group, perm_stride, perm);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// EssTracker.compute_pose_2d2d with validity.method == "GRIC" on tb.kp_ref / tb.kp_cur (n = kp_info[0] on the
// device, n_host = upper bound known to the host for launch sizing).
// small[] layout: [0..8] KinvT, [9..17] Kinv, [18] H_gric, [19] E_gric
// RNG-independent half of compute_pose_2d2d: state reset, findHomography + refinement, GRIC-H.  Everything reads the
// keypoint count from the device (tb.kp_info[0]; n_bound only sizes the launches), so it can be enqueued before the
// host knows that count -- the fused pipeline runs it right behind the nets of a pair, while the solver stage of the
// previous pair is still busy.  Records tb.ev_start (keypoints ready) and tb.ev_h (this half done) on sh.
int TrackerBuffers::enable_stage_timing() {
    for (int i = 0; i < N_SEG; i++)
        if (!ev_seg[i]) DFVO_HIP_CHECK(hipEventCreate(&ev_seg[i]));
    return DFVO_OK;
}
int TrackerBuffers::mark(int i, hipStream_t s) {
    if (!ev_seg[i]) return DFVO_OK;
    DFVO_HIP_CHECK(hipEventRecord(ev_seg[i], s));
    seg_mask |= 1u << i;
    return DFVO_OK;
}

int enqueue_pose_h_part(TrackerBuffers& tb, int n_bound, const PoseConfig& cfg, hipStream_t sh) {
    DFVO_ARG_CHECK(n_bound >= 0 && n_bound <= tb.kp_cap, "compute_pose_2d2d: keypoint capacity");
    tb.seg_mask &= ~0xffu;
    // the intrinsics are constant for a pipeline / tracker: uploaded (synchronously) only when they differ from what this
    // buffer set already holds, so the per-pair path contains no host-to-device copy at all (a copy queued behind the
    // stream's wait for the nets delayed the whole keypoint stage by milliseconds; a pageable source could be read late)
    double hk[18];
    for (int i = 0; i < 9; i++) {
        hk[i] = cfg.KinvT[i];
        hk[9 + i] = cfg.Kinv[i];
    }
    if (!tb.small_valid || memcmp(tb.h_small, hk, sizeof(hk)) != 0) {
        DFVO_HIP_CHECK(hipDeviceSynchronize());  // nothing in flight may still read the old values
        DFVO_HIP_CHECK(hipMemcpy(tb.small, hk, sizeof(hk), hipMemcpyHostToDevice));
        memcpy(tb.h_small, hk, sizeof(hk));
        tb.small_valid = true;
    }
    hipLaunchKernelGGL(k_pose_state_init, dim3(cdiv(tb.kp_cap, 256)), dim3(256), 0, sh, tb.pose, tb.kp_info,
                       tb.best_inliers, tb.kp_cap);
    // the five-point sampler's first chunk of subsets depends on the keypoint COUNT only: drawn here, beside the homography
    // chain, instead of inside the RandomState-ordered chain (round 6; consumed by enqueue_pose_e_part through tb.e_pre_*)
    tb.e_pre_iters = 0;
    static const bool subsets_ahead = !(getenv("DFVO_E_SUBSETS_AHEAD") && atoi(getenv("DFVO_E_SUBSETS_AHEAD")) == 0);  // (0: A/B hook)
    if (subsets_ahead && cfg.max_iters >= 1) {
        int rc_pre = enqueue_e_subsets_prefetch(tb.ws_rep[0], tb.kp_info, n_bound, cfg.max_iters,
                                                reinterpret_cast<unsigned long long*>(tb.kp_total + 72), sh);
        if (rc_pre != DFVO_OK) return rc_pre;
        tb.e_pre_iters = cfg.max_iters;
    }
    DFVO_HIP_CHECK(hipEventRecord(tb.ev_start, sh));
    if (tb.mark(0, sh) != DFVO_OK) return DFVO_ERR_HIP;
    if (cfg.validity == 1) {  // "flow": no homography; the mean displacement decides whether the pair is tracked
        hipLaunchKernelGGL(k_flow_gate, dim3(1), dim3(256), 0, sh, tb.kp_info, tb.kp_ref, tb.kp_cur, cfg.validity_thre,
                           tb.pa, tb.small + 18, tb.kp_total + 5);
        DFVO_HIP_CHECK(hipEventRecord(tb.ev_h, sh));
        DFVO_HIP_CHECK(hipGetLastError());
        return DFVO_OK;
    }
    // ---- homography + GRIC-H (kp_cur -> kp_ref); with 10 or fewer keypoints the result is never consumed
    // (E_tracker.py:196) and with fewer than 5 the chain marks itself "no model"
    // homo_ratio: the same call with ransacReprojThreshold 0.2, only its inlier count is used (E_tracker.py:188-194)
    int rc = enqueue_find_homography(tb.ws_h, tb.kp_cur, tb.kp_ref, n_bound, cfg.validity == 2 ? 0.2 : 1.0, 2000, 0.99, sh,
                                     tb.kp_info);
    if (rc != DFVO_OK) return rc;
    if (tb.mark(1, sh) != DFVO_OK) return DFVO_ERR_HIP;
    if (cfg.validity != 2) {
        GricFusedBatch GH;
        for (int r = 0; r < MAX_E_BATCH; ++r) GH.M[r] = tb.ws_h.out;
        hipLaunchKernelGGL(k_gric_fused, dim3(1), dim3(256), 0, sh, GH, 1, tb.small, tb.small + 9, tb.kp_info, tb.kp_cur,
                           tb.kp_ref, 0, 0.8, 8, 2, tb.small + 18);
        if (tb.mark(2, sh) != DFVO_OK) return DFVO_ERR_HIP;
    }
    DFVO_HIP_CHECK(hipEventRecord(tb.ev_h, sh));
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// RNG-consuming half: `repeat` x (shuffle, findEssentialMat, GRIC-E) as one batch on tb.s_rep[0] (after tb.ev_start),
// then on s (after tb.ev_h): validity bookkeeping, recoverPose, pose / T21.  n_host = the keypoint count.
int enqueue_pose_e_part(TrackerBuffers& tb, int n_host, const PoseConfig& cfg, hipStream_t s, double* d_T21) {
    DFVO_ARG_CHECK(n_host >= 0 && n_host <= tb.kp_cap, "compute_pose_2d2d: keypoint capacity");
    DFVO_ARG_CHECK(cfg.repeat >= 1 && cfg.repeat <= MAX_REP, "compute_pose_2d2d: repeat out of range");
    DFVO_HIP_CHECK(hipStreamWaitEvent(s, tb.ev_h, 0));
    const bool by_flow = cfg.validity == 1, by_ratio = cfg.validity == 2;
    // GRIC: only when more than 10 keypoints (E_tracker.py:196); flow / homo_ratio: whenever the five-point solver has
    // its 5 points
    if (by_flow || by_ratio ? n_host >= 5 : n_host > 10) {
        const int nb = cdiv(n_host, 256);
        const int cap = tb.kp_cap;
        hipStream_t sr = tb.s_rep[0];
        const unsigned R = (unsigned)cfg.repeat;
        // flow: the shuffles (and with them np.random) only run behind an open gate: their count is gate[0] = n or 0
        const int* d_n = by_flow ? tb.kp_total + 5 : tb.kp_info;
        DFVO_HIP_CHECK(hipStreamWaitEvent(sr, by_flow ? tb.ev_h : tb.ev_start, 0));
        if (tb.ev_t[0]) DFVO_HIP_CHECK(hipEventRecord(tb.ev_t[0], sr));
        if (tb.mark(3, sr) != DFVO_OK) return DFVO_ERR_HIP;
        int rc = enqueue_mt_shuffle(tb.mt_state, d_n, n_host, cfg.repeat, cap + 8, tb.perm, sr);
        if (rc != DFVO_OK) return rc;
        hipLaunchKernelGGL(k_permute_points, dim3(nb, R), dim3(256), 0, sr, d_n, tb.perm, cap + 8, tb.kp_cur,
                           tb.kp_ref, tb.pa, tb.pb, 2 * cap);
        // the `repeat` five-point RANSACs as one batched launch sequence (blockIdx.y = repeat)
        const double *pas[MAX_REP], *pbs[MAX_REP];
        for (int rep = 0; rep < cfg.repeat; ++rep) {
            pas[rep] = tb.pa + (size_t)rep * 2 * cap;
            pbs[rep] = tb.pb + (size_t)rep * 2 * cap;
        }
        // (the first chunk's subsets were drawn by the homography half for this keypoint count, if it ran with this budget)
        const unsigned long long* rng_pre = tb.e_pre_iters == cfg.max_iters ? reinterpret_cast<const unsigned long long*>(tb.kp_total + 72) : nullptr;
        tb.e_pre_iters = 0;
        rc = enqueue_find_essential_batch(tb.ws_rep, pas, pbs, cfg.repeat, n_host, cfg.fx, cfg.cx, cfg.cy, 0.99,
                                          cfg.reproj_thre, cfg.max_iters, sr, rng_pre);
        if (rc != DFVO_OK) return rc;
        if (tb.mark(4, sr) != DFVO_OK) return DFVO_ERR_HIP;
        if (by_flow) {
            // cv2.recoverPose(E_rep, shuffled points): only its count is used (E_tracker.py:243-250); the homography
            // workspace, idle in this mode, is the scratch of the `repeat` calls
            for (int rep = 0; rep < cfg.repeat; ++rep) {
                rc = enqueue_recover_pose(tb.ws_h, tb.ws_rep[rep].out, pas[rep], pbs[rep], n_host, cfg.fx, cfg.cx, cfg.cy, sr);
                if (rc != DFVO_OK) return rc;
                hipLaunchKernelGGL(k_copy_double, dim3(1), dim3(1), 0, sr, tb.small + 19 + rep, tb.ws_h.out + 16 + 12);
            }
        } else if (!by_ratio) {
            GricFusedBatch GE;
            for (int rep = 0; rep < MAX_E_BATCH; ++rep) GE.M[rep] = rep < cfg.repeat ? tb.ws_rep[rep].out : nullptr;
            hipLaunchKernelGGL(k_gric_fused, dim3(R), dim3(256), 0, sr, GE, 0, tb.small, tb.small + 9, tb.kp_info, tb.pa, tb.pb,
                               2 * cap, 0.8, 5, 3, tb.small + 19);
        }
        if (tb.ev_t[1]) DFVO_HIP_CHECK(hipEventRecord(tb.ev_t[1], sr));
        if (!by_flow && !by_ratio && tb.mark(5, sr) != DFVO_OK) return DFVO_ERR_HIP;
        DFVO_HIP_CHECK(hipEventRecord(tb.ev_rep[0], sr));
        DFVO_HIP_CHECK(hipStreamWaitEvent(s, tb.ev_rep[0], 0));
        {
            RepBatch RB;
            for (int rep = 0; rep < MAX_REP; ++rep) {
                RB.st[rep] = rep < cfg.repeat ? tb.ws_rep[rep].state : nullptr;
                RB.E[rep] = rep < cfg.repeat ? tb.ws_rep[rep].out : nullptr;
                RB.mask[rep] = rep < cfg.repeat ? tb.ws_rep[rep].mask : nullptr;
            }
            if (by_flow)
                hipLaunchKernelGGL(k_rep_update_flow, dim3(1), dim3(256), 0, s, tb.pose, tb.kp_total + 5, tb.small + 18, RB,
                                   tb.small + 19, tb.perm, cap + 8, tb.best_inliers, cfg.repeat);
            else
                hipLaunchKernelGGL(k_rep_update_all, dim3(1), dim3(256), 0, s, tb.pose, tb.ws_h.state, tb.small + 18, RB,
                                   tb.small + 19, tb.perm, cap + 8, tb.best_inliers, cfg.repeat, by_ratio ? 1 : 0,
                                   cfg.validity_thre);
        }
        if (tb.mark(6, s) != DFVO_OK) return DFVO_ERR_HIP;
        // recoverPose(best_E, kp_cur, kp_ref): always enqueued, consumed only when major_valid; its last kernel also
        // writes the pose bookkeeping and (fused pipeline) the inverse pose for the scale stage
        PoseFinish fin;
        fin.ps = tb.pose;
        fin.T21 = d_T21;
        rc = enqueue_recover_pose(tb.ws_rep[0], (const double*)((const char*)tb.pose + offsetof(PoseState, best_E)),
                                  tb.kp_cur, tb.kp_ref, n_host, cfg.fx, cfg.cx, cfg.cy, s, fin);
        if (rc != DFVO_OK) return rc;
        if (tb.ev_t[2]) DFVO_HIP_CHECK(hipEventRecord(tb.ev_t[2], s));
        if (tb.mark(7, s) != DFVO_OK) return DFVO_ERR_HIP;
    }
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// EssTracker.compute_pose_2d2d with validity.method == "GRIC" on tb.kp_ref / tb.kp_cur (n = kp_info[0] on the
// device, n_host = the same count known to the host).  small[] layout: [0..8] KinvT, [9..17] Kinv, [18] H_gric,
// [19..] E_gric per repeat
int enqueue_compute_pose_2d2d(TrackerBuffers& tb, int n_host, const PoseConfig& cfg, hipStream_t s, double* d_T21) {
    int rc = enqueue_pose_h_part(tb, n_host, cfg, s);
    if (rc != DFVO_OK) return rc;
    return enqueue_pose_e_part(tb, n_host, cfg, s, d_T21);
}

// find_scale_from_depth on tb.kp_ref (kp1) / tb.kp_cur (kp2); d_T21: 16 doubles; d_depth: H x W doubles
// clears the scatter map and the ratio counter of the scale stage on a side stream (tb.s_rep[1]) and records
// tb.ev_rep[1]; enqueue_find_scale(prepared = true) then only waits for that event, so the two fills leave the
// solver's chain of dependent launches
int enqueue_scale_prepare(TrackerBuffers& tb, int H, int W) {
    if ((size_t)H * W > tb.winner_cap) {
        if (tb.winner) (void)hipFree(tb.winner);
        tb.winner_cap = (size_t)H * W;
        DFVO_HIP_CHECK(hipMalloc((void**)&tb.winner, sizeof(int) * tb.winner_cap));
    }
    hipStream_t side = tb.s_rep[1];
    DFVO_HIP_CHECK(hipMemsetAsync(tb.winner, 0xff, sizeof(int) * (size_t)H * W, side));
    DFVO_HIP_CHECK(hipMemsetAsync(tb.kp_total + 4, 0, sizeof(int), side));
    DFVO_HIP_CHECK(hipEventRecord(tb.ev_rep[1], side));
    return DFVO_OK;
}

// which scikit-learn the depth-ratio RANSAC reproduces where the versions differ (dfvo_set_sklearn_compat)
std::atomic<int> g_sklearn_r2_nan_below_two{0};
int set_sklearn_compat(const char* version) {
    DFVO_ARG_CHECK(version, "dfvo_set_sklearn_compat: null version");
    int major = 0, minor = 0;
    DFVO_ARG_CHECK(sscanf(version, "%d.%d", &major, &minor) == 2, "dfvo_set_sklearn_compat: expected \"<major>.<minor>[...]\"");
    // r2_score's "fewer than two samples -> nan" rule exists from scikit-learn 0.22 on
    g_sklearn_r2_nan_below_two.store((major > 0 || minor >= 22) ? 1 : 0);
    return DFVO_OK;
}

// sklearn.linear_model.RANSACRegressor(LinearRegression(fit_intercept=False), min_samples, max_trials, stop_probability,
// residual_threshold).fit(x[:, None], y) on raw arrays already in tb.ratios (x at [0, n), y at [kp_cap, kp_cap + n), or
// d_y_is_ones: y = 1): the regression stage of find_scale_from_depth on its own (no "more than 10 valid points" gate)
int enqueue_ransac_regressor(TrackerBuffers& tb, int n, bool y_is_ones, const ScaleConfig& cfg, hipStream_t s) {
    DFVO_ARG_CHECK(n >= 1 && n <= tb.kp_cap, "ransac_regressor: capacity");
    DFVO_HIP_CHECK(hipMemcpyAsync(tb.kp_total + 4, &n, sizeof(int), hipMemcpyHostToDevice, s));
    DFVO_HIP_CHECK(hipStreamSynchronize(s));  // (n is a stack value)
    hipLaunchKernelGGL(k_scale_ransac, dim3(1), dim3(256), 0, s, tb.mt_state, tb.ratios,
                       y_is_ones ? (const double*)nullptr : tb.ratios + tb.kp_cap, tb.kp_total + 4, -1, cfg.min_samples,
                       cfg.max_trials, cfg.stop_prob, cfg.thre, tb.inl_a, tb.inl_b, tb.scratch, tb.scale_out,
                       (const PoseState*)nullptr, g_sklearn_r2_nan_below_two.load());
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

int enqueue_find_scale(TrackerBuffers& tb, int n_host, const double* d_T21, const double* d_depth, int H, int W,
                       const ScaleConfig& cfg, hipStream_t s, const PoseState* d_gate, bool prepared, bool depth_per_kp) {
    DFVO_ARG_CHECK(n_host >= 0 && n_host <= tb.kp_cap, "find_scale: keypoint capacity");
    if (prepared) {
        DFVO_ARG_CHECK((size_t)H * W <= tb.winner_cap, "find_scale: enqueue_scale_prepare was not called for this size");
        DFVO_HIP_CHECK(hipStreamWaitEvent(s, tb.ev_rep[1], 0));
    } else {
        if ((size_t)H * W > tb.winner_cap) {
            if (tb.winner) (void)hipFree(tb.winner);
            tb.winner_cap = (size_t)H * W;
            DFVO_HIP_CHECK(hipMalloc((void**)&tb.winner, sizeof(int) * tb.winner_cap));
        }
        DFVO_HIP_CHECK(hipMemsetAsync(tb.winner, 0xff, sizeof(int) * (size_t)H * W, s));
        DFVO_HIP_CHECK(hipMemsetAsync(tb.kp_total + 4, 0, sizeof(int), s));
    }
    const int nb = cdiv(n_host > 0 ? n_host : 1, 256);
    const bool abs_diff = cfg.method == 1;
    tb.seg_mask &= ~0x700u;
    if (tb.mark(8, s) != DFVO_OK) return DFVO_ERR_HIP;
    hipLaunchKernelGGL(k_scale_triangulate, dim3(nb), dim3(256), 0, s, tb.kp_info, tb.kp_ref, tb.kp_cur, d_T21, cfg.cx,
                       cfg.cy, cfg.fx, cfg.fy, H, W, tb.z2, tb.pix, tb.winner);
    hipLaunchKernelGGL(k_scale_ratios, dim3(1), dim3(256), sizeof(int) * (size_t)(n_host > 0 ? n_host : 1), s, tb.kp_info,
                       tb.z2, tb.pix, tb.winner, d_depth, tb.ratios, tb.kp_total + 4, abs_diff ? tb.ratios + tb.kp_cap : nullptr,
                       abs_diff ? tb.ratios + 2 * (size_t)tb.kp_cap : nullptr, depth_per_kp ? 1 : 0);
    if (tb.mark(9, s) != DFVO_OK) return DFVO_ERR_HIP;
    hipLaunchKernelGGL(k_scale_ransac, dim3(1), dim3(256), 0, s, tb.mt_state, abs_diff ? tb.ratios + tb.kp_cap : tb.ratios,
                       abs_diff ? tb.ratios + 2 * (size_t)tb.kp_cap : (const double*)nullptr, tb.kp_total + 4, 10,
                       cfg.min_samples, cfg.max_trials, cfg.stop_prob, cfg.thre, tb.inl_a, tb.inl_b, tb.scratch,
                       tb.scale_out, d_gate, g_sklearn_r2_nan_below_two.load());
    if (tb.mark(10, s) != DFVO_OK) return DFVO_ERR_HIP;
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

// ================================================================================================
// Trajectory composition: DFVO.update_global_pose (dfvo.py:109-119) over a whole gathered
// sequence in ONE launch -- rows [n][17] = relative pose (4x4 row major) | status word; status 1 (constant motion,
// dfvo.py:157-161) reuses the previous pair's relative motion.  The recurrence is sequential by definition
// (t_w += R_w t ; R_w = R_w R, in that order, no re-association): one lane walks it, rows staged through LDS by the rest of
// the wave.  Where this pays: the N-rank run, whose gathered rows already sit in HBM after the RCCL all-gather -- the
// composed poses come back with one copy instead of n x 136-byte rows + a Python loop.
// bad[0] = index of the first row with status 2 (needs PnP but had no reference depth), -1 if none.
// ================================================================================================
__global__ __launch_bounds__(64) void k_compose_trajectory(const double* __restrict__ rows, int n, const double* __restrict__ first,
                                                            double* __restrict__ poses, int* __restrict__ bad) {
    __shared__ double s_rows[64 * 17];
    const int lane = threadIdx.x;
    double g[16], prev[16];
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            g[i] = first ? first[i] : ((i % 5) == 0 ? 1.0 : 0.0);
            prev[i] = (i % 5) == 0 ? 1.0 : 0.0;
            poses[i] = g[i];
        }
        bad[0] = -1;
    }
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int cnt = n - c0 < 64 ? n - c0 : 64;
        for (int i = lane; i < cnt * 17; i += 64) s_rows[i] = rows[(size_t)c0 * 17 + i];
        __syncthreads();
        if (lane == 0) {
            for (int k = 0; k < cnt; ++k) {
                const double* r = s_rows + k * 17;
                const int st = (int)r[16];
                if (st == 2 && bad[0] < 0) bad[0] = c0 + k;
                double rel[16];
#pragma unroll
                for (int i = 0; i < 16; i++) rel[i] = st == 1 ? prev[i] : r[i];
                double nt[3], nR[9];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    double a = 0.0;  // (R_w @ t)[i], terms in column order, then + t_w[i]
#pragma unroll
                    for (int j = 0; j < 3; j++) a += g[i * 4 + j] * rel[j * 4 + 3];
                    nt[i] = a + g[i * 4 + 3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        double b = 0.0;
#pragma unroll
                        for (int j = 0; j < 3; j++) b += g[i * 4 + j] * rel[j * 4 + c];
                        nR[i * 3 + c] = b;
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    g[i * 4 + 3] = nt[i];
#pragma unroll
                    for (int c = 0; c < 3; c++) g[i * 4 + c] = nR[i * 3 + c];
                }
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    prev[i] = rel[i];
                    poses[(size_t)(c0 + k + 1) * 16 + i] = g[i];
                }
            }
        }
        __syncthreads();
    }
}

int enqueue_compose_trajectory(const double* d_rows, int n, const double* d_first, double* d_poses, int* d_bad, hipStream_t s) {
    DFVO_ARG_CHECK(n >= 0 && d_poses && d_bad && (n == 0 || d_rows), "compose_trajectory: bad argument");
    hipLaunchKernelGGL(k_compose_trajectory, dim3(1), dim3(64), 0, s, d_rows, n, d_first, d_poses, d_bad);
    DFVO_HIP_CHECK(hipGetLastError());
    return DFVO_OK;
}

}  // namespace dfvo
