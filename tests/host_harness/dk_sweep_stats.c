// Durand-Kerner sweep statistics of the five-point solver's degree-10 polynomials (round 6): how many sweeps
// until cv::solvePoly's `maxDiff <= 0` fires, and do the rows that run all 300 sweeps sit in an exact cycle of the root state?
// Test infrastructure (links the CPU oracle); answers whether k_e_poly_stage3's 0.65 ms -- 300 sweeps -- could be cut short
// bit-exactly.  Build and run (CPU only):
//   cd tests/host_harness && gcc -O2 -ffp-contract=off -I../../oracle -Dcv3_solve_poly=my_solve_poly -c ../../oracle/cv3_calib3d.c -o /tmp/calib.o \
//     && gcc -O2 -ffp-contract=off -I../../oracle -c ../../oracle/cv3_core.c -o /tmp/core.o \
//     && gcc -O2 -ffp-contract=off -I../../oracle -Dcv3_solve_poly=my_solve_poly dk_sweep_stats.c /tmp/calib.o /tmp/core.o -lm -o /tmp/dk && /tmp/dk 20000 0.3
// Result (20 000 five-point subsets of a 2000-point two-view scene, 0.3 / 1.0 px noise): EVERY polynomial runs all 300 sweeps --
// the exit test never fires, because a correction can be non-zero and still be absorbed by the root it is subtracted from --;
// 79 % of them are in an exact cycle of the 20-double state (periods 1 ... 299; median first repetition at sweep 52), 21 % are not
// within 300 sweeps.  A launch of 640 rows therefore always contains rows that need all 300 sweeps: no bit-exact shortcut
// shortens the kernel (a cycle shortcut would cut its ENERGY by ~2/3, not its latency).
#include <math.h>
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "cv3_core.h"
#include "cv3_calib3d.h"
static long hist[302];
static long n_polys = 0, n_full = 0, n_cyc = 0, cyc_hist[301], first_cyc_hist[302];
void my_solve_poly(const double* coeffs, int n0, double* rre, double* rim, int maxIters) {
    int n = n0, iter, i, j;
    double cre[32], cim[32];
    for (i = 0; i <= n0; i++) { cre[i] = coeffs[i]; cim[i] = 0; }
    for (; n > 1; n--) if (fabs(cre[n]) + fabs(cim[n]) > DBL_EPSILON) break;
    double pre = 1, pim = 0; const double rr = 1, ri = 1;
    for (i = 0; i < n; i++) { rre[i] = pre; rim[i] = pim; double tre = pre * rr - pim * ri, tim = pre * ri + pim * rr; pre = tre; pim = tim; }
    static double snap[512][20];
    int period = 0, first_cyc = -1;
    for (iter = 0; iter < maxIters; iter++) {
        double maxDiff = 0;
        for (i = 0; i < n; i++) {
            pre = rre[i]; pim = rim[i];
            double nre = cre[n], nim = cim[n], dre = cre[n], dim = cim[n];
            for (j = 0; j < n; j++) {
                double tre = nre * pre - nim * pim, tim = nre * pim + nim * pre;
                nre = tre + cre[n - j - 1]; nim = tim + cim[n - j - 1];
                if (j != i) { double qre = pre - rre[j], qim = pim - rim[j]; tre = dre * qre - dim * qim; tim = dre * qim + dim * qre; dre = tre; dim = tim; }
            }
            double t = 1. / (dre * dre + dim * dim);
            double qre = (nre * dre + nim * dim) * t, qim = (-nre * dim + nim * dre) * t;
            nre = qre; nim = qim;
            rre[i] = pre - nre; rim[i] = pim - nim;
            double an = sqrt(nre * nre + nim * nim);
            maxDiff = maxDiff > an ? maxDiff : an;
        }
        if (maxDiff <= 0) break;
        // cycle detection against the last 64 states (bitwise)
        double cur[20]; memcpy(cur, rre, 80); memcpy(cur + 10, rim, 80);
        if (first_cyc < 0)
            for (int back = 1; back <= 299 && back <= iter; ++back)
                if (!memcmp(cur, snap[(iter - back) & 511], 160)) { period = back; first_cyc = iter; break; }
        memcpy(snap[iter & 511], cur, 160);
    }
    hist[iter]++; n_polys++;
    if (iter >= maxIters) { n_full++; if (period) { n_cyc++; cyc_hist[period]++; first_cyc_hist[first_cyc]++; } }
    for (i = 0; i < n; i++) if (fabs(rim[i]) < 1e-100) rim[i] = 0;
    for (; n < n0; n++) { rre[n] = rre[n - 1]; rim[n] = rim[n - 1]; }
}
static double urand(void) { return rand() / (double)RAND_MAX; }
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 20000;
    const double noise = argc > 2 ? atof(argv[2]) : 0.3;  // pixels
    srand(7);
    // a tunnel-like scene: points at depth 4..40 m, KITTI intrinsics, forward + small lateral motion
    const double f = 718.856, cx = 607.19, cy = 185.2;
    const int NP = 2000;
    static double p1[2000 * 2], p2[2000 * 2];
    const double ang = 0.01, tx = 0.4, ty = 0.02, tz = 1.0;
    for (int k = 0; k < NP; ++k) {
        double X = (urand() - 0.5) * 30, Y = (urand() - 0.5) * 6, Z = 4 + urand() * 36;
        double x1 = f * X / Z + cx, y1 = f * Y / Z + cy;
        double Xr = cos(ang) * X + sin(ang) * Z - tx, Yr = Y - ty, Zr = -sin(ang) * X + cos(ang) * Z - tz;
        double x2 = f * Xr / Zr + cx, y2 = f * Yr / Zr + cy;
        p1[2 * k] = (x1 + (urand() - 0.5) * 2 * noise - cx) / f; p1[2 * k + 1] = (y1 + (urand() - 0.5) * 2 * noise - cy) / f;
        p2[2 * k] = (x2 + (urand() - 0.5) * 2 * noise - cx) / f; p2[2 * k + 1] = (y2 + (urand() - 0.5) * 2 * noise - cy) / f;
    }
    for (int it = 0; it < N; ++it) {
        double q1[10], q2[10], E[90];
        for (int i = 0; i < 5; ++i) { int k = rand() % NP; q1[2 * i] = p1[2 * k]; q1[2 * i + 1] = p1[2 * k + 1]; q2[2 * i] = p2[2 * k]; q2[2 * i + 1] = p2[2 * k + 1]; }
        cv3_five_point(q1, q2, E);
    }
    printf("polynomials %ld | ran all 300 sweeps: %ld (%.2f%%) of which in an exact cycle (period <= 64): %ld\n", n_polys, n_full, 100.0 * n_full / n_polys, n_cyc);
    printf("exit sweep histogram (deciles): ");
    long acc = 0; int q = 1;
    for (int i = 0; i <= 300; ++i) { acc += hist[i]; while (q <= 10 && acc * 10 >= q * n_polys) { printf("p%d0=%d ", q, i); q++; } }
    printf("\nperiods: "); long big = 0; for (int p = 65; p <= 300; ++p) big += cyc_hist[p]; printf("periods > 64: %ld", big);
    printf("\nfirst sweep at which the cycle is detected (deciles over cycling rows): ");
    acc = 0; q = 1; for (int i = 0; i <= 300; ++i) { acc += first_cyc_hist[i]; while (q <= 10 && n_cyc && acc * 10 >= q * n_cyc) { printf("p%d0=%d ", q, i); q++; } }
    printf("\n");
    return 0;
}
