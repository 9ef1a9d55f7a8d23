/* ORACLE (test infrastructure only).  numpy's scalar introselect for np.argpartition on float32 keys
 * (numpy/core/src/npysort/selection.c.src: aintroselect_float, amedian_of_median5_float, amedian5_float,
 * adumb_select_float), i.e. the order numpy 1.16 produced and current numpy produces with its SIMD sort
 * dispatch disabled.  Pinned against real numpy in tests/test_oracle_tracker.py (subprocess with
 * NPY_DISABLE_CPU_FEATURES). */
#include <stdint.h>

static int flt_lt(float a, float b) { return a < b || (b != b && a == a); }
#define IDX(i) tosort[i]
#define SWAPI(a, b)            \
    {                          \
        int64_t _t = tosort[a]; \
        tosort[a] = tosort[b]; \
        tosort[b] = _t;        \
    }

static void dumb_select(const float* v, int64_t* tosort, int64_t num, int64_t kth) {
    for (int64_t i = 0; i <= kth; i++) {
        int64_t minidx = i;
        float minval = v[IDX(i)];
        for (int64_t k = i + 1; k < num; k++)
            if (flt_lt(v[IDX(k)], minval)) {
                minidx = k;
                minval = v[IDX(k)];
            }
        SWAPI(i, minidx);
    }
}

static int64_t median5(const float* v, int64_t* tosort) {
    if (flt_lt(v[IDX(1)], v[IDX(0)])) SWAPI(1, 0);
    if (flt_lt(v[IDX(4)], v[IDX(3)])) SWAPI(4, 3);
    if (flt_lt(v[IDX(3)], v[IDX(0)])) SWAPI(3, 0);
    if (flt_lt(v[IDX(4)], v[IDX(1)])) SWAPI(4, 1);
    if (flt_lt(v[IDX(2)], v[IDX(1)])) SWAPI(2, 1);
    if (flt_lt(v[IDX(3)], v[IDX(2)])) {
        if (flt_lt(v[IDX(3)], v[IDX(1)])) return 1;
        return 3;
    }
    return 2;
}

void np_aintroselect_float(const float* v, int64_t* tosort, int64_t num, int64_t kth);

static int64_t median_of_median5(const float* v, int64_t* tosort, int64_t num) {
    int64_t right = num - 1, nmed = (right + 1) / 5;
    for (int64_t i = 0, subleft = 0; i < nmed; i++, subleft += 5) {
        int64_t m = median5(v, tosort + subleft);
        int64_t t = tosort[subleft + m];
        tosort[subleft + m] = tosort[i];
        tosort[i] = t;
    }
    if (nmed > 2) np_aintroselect_float(v, tosort, nmed, nmed / 2);
    return nmed / 2;
}

static int msb(uint64_t n) {
    int d = 0;
    while (n >>= 1) d++;
    return d;
}

void np_aintroselect_float(const float* v, int64_t* tosort, int64_t num, int64_t kth) {
    int64_t low = 0, high = num - 1;
    if (num <= 0) return;
    if (kth - low < 3) {
        dumb_select(v, tosort + low, high - low + 1, kth - low);
        return;
    } else if (kth == num - 1) {
        int64_t maxidx = low;
        float maxval = v[IDX(low)];
        for (int64_t k = low + 1; k < num; k++)
            if (!flt_lt(v[IDX(k)], maxval)) {
                maxidx = k;
                maxval = v[IDX(k)];
            }
        SWAPI(kth, maxidx);
        return;
    }
    int depth_limit = msb((uint64_t)num) * 2;
    for (; low + 1 < high;) {
        int64_t ll = low + 1, hh = high;
        if (depth_limit > 0 || hh - ll < 5) {
            const int64_t mid = low + (high - low) / 2;
            if (flt_lt(v[IDX(high)], v[IDX(mid)])) SWAPI(high, mid);
            if (flt_lt(v[IDX(high)], v[IDX(low)])) SWAPI(high, low);
            if (flt_lt(v[IDX(low)], v[IDX(mid)])) SWAPI(low, mid);
            SWAPI(mid, low + 1);
        } else {
            int64_t mid = ll + median_of_median5(v, tosort + ll, hh - ll);
            SWAPI(mid, low);
            ll--;
            hh++;
        }
        depth_limit--;
        const float pivot = v[IDX(low)];
        for (;;) {
            do ll++;
            while (flt_lt(v[IDX(ll)], pivot));
            do hh--;
            while (flt_lt(pivot, v[IDX(hh)]));
            if (hh < ll) break;
            SWAPI(hh, ll);
        }
        SWAPI(low, hh);
        if (hh >= kth) high = hh - 1;
        if (hh <= kth) low = ll;
    }
    if (high == low + 1) {
        if (flt_lt(v[IDX(high)], v[IDX(low)])) SWAPI(high, low);
    }
}
