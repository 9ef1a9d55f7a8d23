// cv::solvePoly (Durand-Kerner, degree 10, 300 sweeps) with ONE ROOT PER LANE, sixteen lanes (one DPP row) per
// polynomial -- the five-point solver's polynomial stage behind cv2.findEssentialMat
// (/root/reference/libs/tracker/E_tracker.py:231-239).
//
// The sequential sweep (sm::solve_poly_fixed, solver_math.h) updates root i from
//     num   = Horner(c, p_i)                                              -- depends on p_i only
//     denom = ((((c[n] * (p_i - r_0)) * (p_i - r_1)) ... ) * (p_i - r_9)),  j != i, strictly left to right,
// where r_j is ALREADY UPDATED for j < i and still the old value for j > i (Gauss-Seidel order).  In that left-to-right
// product the updated roots come first.  So with one root per lane:
//     * every lane runs its Horner chain at once;
//     * step s = 0..9: lane s multiplies its remaining factors (p_s - old r_j), j = s+1..9, divides, and owns the new
//       root s; the new root is broadcast over the row (v_mov_b32_dpp row_newbcast:s -- a VALU move, no LDS round
//       trip); every lane i > s multiplies its denominator by (p_i - new r_s) -- exactly the factor the sequential
//       loop multiplies next for that root.
// Each root sees the same operands in the same order, hence the same bits; the instruction stream of a sweep shrinks
// from 10 x (10 Horner + 9 factor + 1 divide) complex steps to 10 + 54 + 10.  The running `maxDiff` of the sweep is
// carried through the same broadcasts so that the early exit (`maxDiff <= 0`, NaN semantics of the ternary included)
// is the sequential one.  Device only: built from DPP moves; tests/host_harness holds a lock-step host emulation of
// this schedule that is compared bit for bit with the oracle's cv3_solve_poly.
#pragma once
#include <hip/hip_runtime.h>

#include "solver_math.h"

namespace sm {

template <int S>
__device__ __forceinline__ double row_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + S, 0xf, 0xf, false);  // row_newbcast:S
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + S, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

struct PolyLaneState {
    double cr[11];
    double xr[10], xi[10];  // every lane's copy of the ten roots
    double pre, pim;        // this lane's own root
    double nre, nim, dre, dim, md;
};

template <int S>
__device__ __forceinline__ void poly_lane_step(PolyLaneState& z, const int li, const bool active) {
    if (li == S && active) {
        double dre = z.dre, dim = z.dim;
#pragma unroll
        for (int j = S + 1; j < 10; j++) {
            const double qre = z.pre - z.xr[j], qim = z.pim - z.xi[j];
            const double tre = dre * qre - dim * qim, tim = dre * qim + dim * qre;
            dre = tre;
            dim = tim;
        }
        const double t = 1. / (dre * dre + dim * dim);
        const double qre = (z.nre * dre + z.nim * dim) * t, qim = (-z.nre * dim + z.nim * dre) * t;
        z.pre = z.pre - qre;
        z.pim = z.pim - qim;
        const double an = qre * qre + qim * qim;
        z.md = z.md > an ? z.md : an;
    }
    z.xr[S] = row_bcast<S>(z.pre);
    z.xi[S] = row_bcast<S>(z.pim);
    z.md = row_bcast<S>(z.md);
    if (li > S) {
        const double qre = z.pre - z.xr[S], qim = z.pim - z.xi[S];
        const double tre = z.dre * qre - z.dim * qim, tim = z.dre * qim + z.dim * qre;
        z.dre = tre;
        z.dim = tim;
    }
}

// All 64 lanes of the wave must call this together (the broadcasts read across the row).  `li` = lane within its
// 16-lane row; `run` (uniform over a row) = this row has a degree-10 polynomial to solve.  On return lanes li < 10 of a
// running row hold root li in (rre, rim), |im| < 1e-100 flushed to 0 as cv::solvePoly does for real coefficients.
__device__ __forceinline__ void solve_poly10_row(const double* c, const int li, const bool run, double& rre, double& rim) {
    PolyLaneState z;
#pragma unroll
    for (int i = 0; i <= 10; i++) z.cr[i] = c[i];
    {
        double pre = 1, pim = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            z.xr[i] = pre;
            z.xi[i] = pim;
            const double tre = pre * 1.0 - pim * 1.0, tim = pre * 1.0 + pim * 1.0;
            pre = tre;
            pim = tim;
        }
    }
    z.pre = z.xr[0];
    z.pim = z.xi[0];
#pragma unroll
    for (int i = 1; i < 10; i++)
        if (li == i) {
            z.pre = z.xr[i];
            z.pim = z.xi[i];
        }
    bool active = run;
    for (int iter = 0; iter < 300; iter++) {
        if (!__any(active)) break;
        z.nre = z.cr[10];
        z.nim = 0;
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const double tre = z.nre * z.pre - z.nim * z.pim, tim = z.nre * z.pim + z.nim * z.pre;
            z.nre = tre + z.cr[10 - j - 1];
            z.nim = tim + 0.0;
        }
        z.dre = z.cr[10];
        z.dim = 0;
        z.md = 0;
        poly_lane_step<0>(z, li, active);
        poly_lane_step<1>(z, li, active);
        poly_lane_step<2>(z, li, active);
        poly_lane_step<3>(z, li, active);
        poly_lane_step<4>(z, li, active);
        poly_lane_step<5>(z, li, active);
        poly_lane_step<6>(z, li, active);
        poly_lane_step<7>(z, li, active);
        poly_lane_step<8>(z, li, active);
        poly_lane_step<9>(z, li, active);
        if (z.md <= 0) active = false;  // the sequential loop's `if (maxDiff <= 0) break`
    }
    rre = z.pre;
    rim = fabs(z.pim) < 1e-100 ? 0 : z.pim;
}

}  // namespace sm
