// np.argpartition(score, kth) emulation (numpy's scalar introselect for float32 keys, argsort flavour)
// used by local_bestN (/root/reference/libs/matching/kp_selection.py:166-173).  The ORDER in which
// argpartition returns the k smallest elements is an artefact of the selection algorithm, and that
// order feeds the RANSAC sample sequence downstream (SURVEY.md hard part 3), so the algorithm itself
// is reproduced: numpy/core/src/npysort/selection.c.src `aintroselect_float` (median-of-3 quickselect
// with the dumb-select and find-max shortcuts and the median-of-medians fallback), i.e. what
// numpy 1.16 ran and what current numpy runs when its AVX-512/AVX2 sort dispatch is disabled.
#pragma once
#include "solver_math.h"

namespace sm {

// FLOAT_LT with NaNs sorted to the end
SM_HD bool kp_lt(float a, float b) { return a < b || (b != b && a == a); }

#define KP_SWAP(a, b) \
    {                 \
        IdxT _t = (a); \
        (a) = (b);    \
        (b) = _t;     \
    }

template <typename IdxT>
SM_HD void kp_dumbselect(const float* v, IdxT* tosort, int num, int kth) {
    for (int i = 0; i <= kth; i++) {
        int minidx = i;
        float minval = v[tosort[i]];
        for (int k = i + 1; k < num; k++) {
            if (kp_lt(v[tosort[k]], minval)) {
                minidx = k;
                minval = v[tosort[k]];
            }
        }
        KP_SWAP(tosort[i], tosort[minidx]);
    }
}

template <typename IdxT>
SM_HD int kp_median5(const float* v, IdxT* tosort) {
    if (kp_lt(v[tosort[1]], v[tosort[0]])) KP_SWAP(tosort[1], tosort[0]);
    if (kp_lt(v[tosort[4]], v[tosort[3]])) KP_SWAP(tosort[4], tosort[3]);
    if (kp_lt(v[tosort[3]], v[tosort[0]])) KP_SWAP(tosort[3], tosort[0]);
    if (kp_lt(v[tosort[4]], v[tosort[1]])) KP_SWAP(tosort[4], tosort[1]);
    if (kp_lt(v[tosort[2]], v[tosort[1]])) KP_SWAP(tosort[2], tosort[1]);
    if (kp_lt(v[tosort[3]], v[tosort[2]])) {
        if (kp_lt(v[tosort[3]], v[tosort[1]])) return 1;
        return 3;
    }
    return 2;
}

template <typename IdxT>
SM_HD_NOINLINE void kp_introselect(const float* v, IdxT* tosort, int num, int kth, int depth);

template <typename IdxT>
SM_HD int kp_median_of_median5(const float* v, IdxT* tosort, int num, int depth) {
    const int right = num - 1;
    const int nmed = (right + 1) / 5;
    for (int i = 0, subleft = 0; i < nmed; i++, subleft += 5) {
        const int m = kp_median5(v, tosort + subleft);
        KP_SWAP(tosort[subleft + m], tosort[i]);
    }
    if (nmed > 2 && depth < 4) kp_introselect(v, tosort, nmed, nmed / 2, depth + 1);
    return nmed / 2;
}

SM_HD int kp_msb(unsigned n) {
    int d = 0;
    while (n >>= 1) d++;
    return d;
}

// tosort: permutation of [0, num) (argsort indices), partially ordered on return: the element that
// belongs at position kth is there, smaller ones before it, larger ones after it.
template <typename IdxT>
SM_HD_NOINLINE void kp_introselect(const float* v, IdxT* tosort, int num, int kth, int depth) {
    int low = 0, high = num - 1;
    if (kth - low < 3) {
        kp_dumbselect(v + 0, tosort + low, high - low + 1, kth - low);
        return;
    } else if (kth == num - 1) {
        int maxidx = low;
        float maxval = v[tosort[low]];
        for (int k = low + 1; k < num; k++) {
            if (!kp_lt(v[tosort[k]], maxval)) {
                maxidx = k;
                maxval = v[tosort[k]];
            }
        }
        KP_SWAP(tosort[kth], tosort[maxidx]);
        return;
    }
    int depth_limit = kp_msb((unsigned)num) * 2;
    for (; low + 1 < high;) {
        int ll = low + 1, hh = high;
        if (depth_limit > 0 || hh - ll < 5) {
            const int mid = low + (high - low) / 2;
            if (kp_lt(v[tosort[high]], v[tosort[mid]])) KP_SWAP(tosort[high], tosort[mid]);
            if (kp_lt(v[tosort[high]], v[tosort[low]])) KP_SWAP(tosort[high], tosort[low]);
            if (kp_lt(v[tosort[low]], v[tosort[mid]])) KP_SWAP(tosort[low], tosort[mid]);
            KP_SWAP(tosort[mid], tosort[low + 1]);
        } else {
            const int mid = ll + kp_median_of_median5(v, tosort + ll, hh - ll, depth);
            KP_SWAP(tosort[mid], tosort[low]);
            ll--;
            hh++;
        }
        depth_limit--;
        const float pivot = v[tosort[low]];
        for (;;) {
            do ll++;
            while (kp_lt(v[tosort[ll]], pivot));
            do hh--;
            while (kp_lt(pivot, v[tosort[hh]]));
            if (hh < ll) break;
            KP_SWAP(tosort[hh], tosort[ll]);
        }
        KP_SWAP(tosort[low], tosort[hh]);
        if (hh >= kth) high = hh - 1;
        if (hh <= kth) low = ll;
    }
    if (high == low + 1) {
        if (kp_lt(v[tosort[high]], v[tosort[low]])) KP_SWAP(tosort[high], tosort[low]);
    }
}

// ------------------------------------------------------------------------------------------------
// Same algorithm with the keys carried along: key[i] == v[tosort[i]] is kept true by swapping both
// arrays, so every scan reads consecutive addresses (no dependent tosort -> v lookup) and the two
// partition scans fetch four keys per round trip.  On the GPU both arrays sit in LDS, where the
// dependent lookup costs a full LDS latency per element.  Results are identical to kp_introselect
// (same comparisons on the same values in the same order); `key` is left permuted.
// The 4-wide scans may LOAD up to 3 elements past the scanned range (never use them): the caller keeps
// key[-3 .. num+2] readable.
// ------------------------------------------------------------------------------------------------
#define KP_SWAP2(i, j)               \
    {                                \
        IdxT _t = tosort[i];         \
        tosort[i] = tosort[j];       \
        tosort[j] = _t;              \
        float _k = key[i];           \
        key[i] = key[j];             \
        key[j] = _k;                 \
    }

template <typename IdxT>
SM_HD void kp_dumbselect_cp(float* key, IdxT* tosort, int num, int kth) {
    for (int i = 0; i <= kth; i++) {
        int minidx = i;
        float minval = key[i];
        for (int k = i + 1; k < num; k++) {
            if (kp_lt(key[k], minval)) {
                minidx = k;
                minval = key[k];
            }
        }
        KP_SWAP2(i, minidx);
    }
}

template <typename IdxT>
SM_HD int kp_median5_cp(float* key, IdxT* tosort) {
    if (kp_lt(key[1], key[0])) KP_SWAP2(1, 0);
    if (kp_lt(key[4], key[3])) KP_SWAP2(4, 3);
    if (kp_lt(key[3], key[0])) KP_SWAP2(3, 0);
    if (kp_lt(key[4], key[1])) KP_SWAP2(4, 1);
    if (kp_lt(key[2], key[1])) KP_SWAP2(2, 1);
    if (kp_lt(key[3], key[2])) {
        if (kp_lt(key[3], key[1])) return 1;
        return 3;
    }
    return 2;
}

template <typename IdxT>
SM_HD_NOINLINE void kp_introselect_cp(float* key, IdxT* tosort, int num, int kth, int depth);

template <typename IdxT>
SM_HD int kp_median_of_median5_cp(float* key, IdxT* tosort, int num, int depth) {
    const int right = num - 1;
    const int nmed = (right + 1) / 5;
    for (int i = 0, subleft = 0; i < nmed; i++, subleft += 5) {
        const int m = kp_median5_cp(key + subleft, tosort + subleft);
        KP_SWAP2(subleft + m, i);
    }
    if (nmed > 2 && depth < 4) kp_introselect_cp(key, tosort, nmed, nmed / 2, depth + 1);
    return nmed / 2;
}

// main loop of the selection from an intermediate state (low, high, depth_limit): lets a caller run the first,
// long partition passes some other way (k_kp_cell does them with the whole workgroup) and finish here
template <typename IdxT>
SM_HD_NOINLINE void kp_introselect_cp_from(float* key, IdxT* tosort, int kth, int depth, int low, int high,
                                           int depth_limit);

template <typename IdxT>
SM_HD_NOINLINE void kp_introselect_cp(float* key, IdxT* tosort, int num, int kth, int depth) {
    int low = 0, high = num - 1;
    if (kth - low < 3) {
        kp_dumbselect_cp(key + low, tosort + low, high - low + 1, kth - low);
        return;
    } else if (kth == num - 1) {
        int maxidx = low;
        float maxval = key[low];
        for (int k = low + 1; k < num; k++) {
            if (!kp_lt(key[k], maxval)) {
                maxidx = k;
                maxval = key[k];
            }
        }
        KP_SWAP2(kth, maxidx);
        return;
    }
    kp_introselect_cp_from(key, tosort, kth, depth, low, high, kp_msb((unsigned)num) * 2);
}

template <typename IdxT>
SM_HD_NOINLINE void kp_introselect_cp_from(float* key, IdxT* tosort, int kth, int depth, int low, int high,
                                           int depth_limit) {
    for (; low + 1 < high;) {
        int ll = low + 1, hh = high;
        if (depth_limit > 0 || hh - ll < 5) {
            const int mid = low + (high - low) / 2;
            if (kp_lt(key[high], key[mid])) KP_SWAP2(high, mid);
            if (kp_lt(key[high], key[low])) KP_SWAP2(high, low);
            if (kp_lt(key[low], key[mid])) KP_SWAP2(low, mid);
            KP_SWAP2(mid, low + 1);
        } else {
            const int mid = ll + kp_median_of_median5_cp(key + ll, tosort + ll, hh - ll, depth);
            KP_SWAP2(mid, low);
            ll--;
            hh++;
        }
        depth_limit--;
        const float pivot = key[low];
        for (;;) {
            // do ll++ while (key[ll] < pivot): four candidates per fetch
            for (;;) {
                const float k0 = key[ll + 1], k1 = key[ll + 2], k2 = key[ll + 3], k3 = key[ll + 4];
                if (!kp_lt(k0, pivot)) { ll += 1; break; }
                if (!kp_lt(k1, pivot)) { ll += 2; break; }
                if (!kp_lt(k2, pivot)) { ll += 3; break; }
                ll += 4;
                if (!kp_lt(k3, pivot)) break;
            }
            // do hh-- while (pivot < key[hh])
            for (;;) {
                const float k0 = key[hh - 1], k1 = key[hh - 2], k2 = key[hh - 3], k3 = key[hh - 4];
                if (!kp_lt(pivot, k0)) { hh -= 1; break; }
                if (!kp_lt(pivot, k1)) { hh -= 2; break; }
                if (!kp_lt(pivot, k2)) { hh -= 3; break; }
                hh -= 4;
                if (!kp_lt(pivot, k3)) break;
            }
            if (hh < ll) break;
            KP_SWAP2(hh, ll);
        }
        KP_SWAP2(low, hh);
        if (hh >= kth) high = hh - 1;
        if (hh <= kth) low = ll;
    }
    if (high == low + 1) {
        if (kp_lt(key[high], key[low])) KP_SWAP2(high, low);
    }
}

// local_bestN cell bounds (kp_selection.py:129-131): python float arithmetic, int() truncation
SM_HD void kp_cell_bounds(int h, int w, int num_row, int num_col, int row, int col, int* y0, int* y1, int* x0,
                          int* x1) {
    *y0 = (int)((double)h / (double)num_row * (double)row);
    *x0 = (int)((double)w / (double)num_col * (double)col);
    *y1 = (int)((double)h / (double)num_row * (double)(row + 1)) - 1;
    *x1 = (int)((double)w / (double)num_col * (double)(col + 1)) - 1;
}

}  // namespace sm
