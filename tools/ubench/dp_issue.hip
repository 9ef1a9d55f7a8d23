// Per-wave issue cost of fp64 VALU work: does a wavefront with few active lanes execute faster, and how much does
// instruction-level parallelism help?  One wavefront per block, one block, so the numbers are single-wave latencies
// (what the one-hypothesis-per-lane solver kernels are bound by).
//   hipcc --offload-arch=gfx950 -O3 dp_issue.hip -o dp_issue
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ __launch_bounds__(64) void k(double* out, int iters, int active, long long* cyc) {
    double x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = threadIdx.x * 1e-3 + i;
    const double a = 1.0000001, b = 1e-9;
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < active) {
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) x[i] = __builtin_fma(x[i], a, b);
        }
        t1 = __builtin_readcyclecounter();
        double s = 0;
        for (int i = 0; i < ILP; i++) s += x[i];
        out[threadIdx.x] = s;
        if (threadIdx.x == 0) *cyc = t1 - t0;
    }
}
template <int ILP>
void run(int active) {
    double* out;
    long long* cyc;
    hipMalloc(&out, 64 * sizeof(double));
    hipMalloc(&cyc, sizeof(long long));
    const int iters = 100000;
    k<ILP><<<1, 64>>>(out, 100, active, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k<ILP><<<1, 64>>>(out, iters, active, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("active lanes %2d  ILP %d: %.3f ms -> %.2f ns per fp64 FMA instruction\n", active, ILP, ms,
           ms * 1e6 / ((double)iters * ILP));
    hipFree(out);
    hipFree(cyc);
}
int main() {
    for (int a : {64, 32, 16, 8, 1}) {
        run<1>(a);
        run<4>(a);
        run<8>(a);
    }
    return 0;
}
