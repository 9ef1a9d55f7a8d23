// numpy legacy RandomState (MT19937) stream, reproduced draw for draw, so that the shuffles of
// /root/reference/libs/tracker/E_tracker.py:225-228 and pnp_tracker.py:91-92 (np.random.shuffle) and
// the subset draws of sklearn's RANSACRegressor (E_tracker.py:618-636: random_state=None -> the
// global RandomState, sample_without_replacement -> RandomState.randint) can run on the device
// without leaving the global np.random stream: the 624-word state is uploaded from
// np.random.get_state(), consumed on the GPU in the reference's order and handed back.
//
// Follows numpy/random/mtrand (legacy): mt19937_gen / genrand tempering, random_interval (masked
// rejection on 32-bit draws), _shuffle_raw (Fisher-Yates from the top, swapping even when j == i),
// _bounded_integers masked rejection for randint; and sklearn/utils/_random.pyx
// sample_without_replacement ("auto": tracking selection below 1 %, reservoir sampling below 99 %).
#pragma once
#include <stdint.h>

#include "solver_math.h"

namespace sm {

struct Mt19937 {
    uint32_t key[624];
    int pos;
};

SM_HD void mt_regen(Mt19937& s) {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    int i;
    uint32_t y;
    for (i = 0; i < 624 - 397; i++) {
        y = (s.key[i] & UPPER) | (s.key[i + 1] & LOWER);
        s.key[i] = s.key[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; i < 623; i++) {
        y = (s.key[i] & UPPER) | (s.key[i + 1] & LOWER);
        s.key[i] = s.key[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (s.key[623] & UPPER) | (s.key[0] & LOWER);
    s.key[623] = s.key[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    s.pos = 0;
}

SM_HD uint32_t mt_next32(Mt19937& s) {
    if (s.pos == 624) mt_regen(s);
    uint32_t y = s.key[s.pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// random_interval(max): uniform integer in [0, max] (max < 2^32 here)
SM_HD uint32_t mt_interval(Mt19937& s, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    while ((v = (mt_next32(s) & mask)) > max) {
    }
    return v;
}

// RandomState.randint(n) / randint(0, n): uniform in [0, n)
SM_HD uint32_t mt_randint(Mt19937& s, uint32_t n) { return mt_interval(s, n - 1); }

// np.random.shuffle(np.arange(n)) -> perm[n]
SM_HD void mt_shuffle_arange(Mt19937& s, int n, int* perm) {
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int i = n - 1; i >= 1; i--) {
        const int j = (int)mt_interval(s, (uint32_t)i);
        const int t = perm[j];
        perm[j] = perm[i];
        perm[i] = t;
    }
}

// sklearn.utils.random.sample_without_replacement(n_population, n_samples, method="auto"), n_samples <= 8:
//   0.01 < ratio < 0.99 -> RandomState.permutation(n_population)[:n_samples]
//   otherwise ratio < 0.2 -> tracking selection, else reservoir sampling.
// `scratch` must hold n_population ints when the permutation branch can be taken (n_population < 100 * n_samples).
SM_HD void mt_sample_without_replacement(Mt19937& s, int n_population, int n_samples, int* out, int* scratch) {
    const double ratio = n_population != 0 ? (double)n_samples / (double)n_population : 1.0;
    if (ratio > 0.01 && ratio < 0.99) {
        mt_shuffle_arange(s, n_population, scratch);
        for (int i = 0; i < n_samples; i++) out[i] = scratch[i];
    } else if (ratio < 0.2) {  // tracking selection
        for (int i = 0; i < n_samples; i++) {
            int j;
            bool dup;
            do {
                j = (int)mt_randint(s, (uint32_t)n_population);
                dup = false;
                for (int k = 0; k < i; k++) dup = dup || (out[k] == j);
            } while (dup);
            out[i] = j;
        }
    } else {  // reservoir sampling
        for (int i = 0; i < n_samples; i++) out[i] = i;
        for (int i = n_samples; i < n_population; i++) {
            const int j = (int)mt_randint(s, (uint32_t)(i + 1));
            if (j < n_samples) out[j] = i;
        }
    }
}

// np.add.reduce over a contiguous float64 array: numpy's pairwise summation (loops.c.src pairwise_sum_DOUBLE): below 8
// elements a plain loop, up to 128 eight interleaved partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and
// the tail added one by one, above that a split at n/2 rounded down to a multiple of 8
SM_HD_NOINLINE double np_pairwise_sum(const double* a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// RigidFlow layer for one pixel (geometry/{backprojection,transformation3d,projection}.py, layers.py PixToFlow) in
// float32: every matmul row is a0*b0 rounded, then fused multiply-adds in ascending k (the order of the torch-CPU GEMM
// that produced tests/golden/rigid_flow_kp.npz).  Ki: inverse intrinsics 3x3, T: 4x4 motion, K: intrinsics 3x3.
SM_HD void rigid_flow_px(const float* Ki, const float* T, const float* K, float x, float y, float d, float* rx, float* ry) {
    float P[4];
    for (int r = 0; r < 3; ++r) {
        float a = Ki[r * 3] * x;
        a = __builtin_fmaf(Ki[r * 3 + 1], y, a);
        a = __builtin_fmaf(Ki[r * 3 + 2], 1.0f, a);
        P[r] = d * a;
    }
    P[3] = 1.0f;
    float Q[4];
    for (int r = 0; r < 4; ++r) {
        float a = T[r * 4] * P[0];
        a = __builtin_fmaf(T[r * 4 + 1], P[1], a);
        a = __builtin_fmaf(T[r * 4 + 2], P[2], a);
        a = __builtin_fmaf(T[r * 4 + 3], P[3], a);
        Q[r] = a;
    }
    float U[3];
    for (int r = 0; r < 3; ++r) {
        float a = K[r * 3] * Q[0];
        a = __builtin_fmaf(K[r * 3 + 1], Q[1], a);
        a = __builtin_fmaf(K[r * 3 + 2], Q[2], a);
        a = __builtin_fmaf(0.0f, Q[3], a);  // the zero 4th column of the 3x4 intrinsics
        U[r] = a;
    }
    const float den = U[2] + 1e-7f;
    *rx = U[0] / den - x;
    *ry = U[1] / den - y;
}
}  // namespace sm
