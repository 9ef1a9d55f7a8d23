// Sustained v_mfma_f32_16x16x4_f32 rate of the whole chip (no memory traffic): the practical ceiling the conv
// kernels are compared with in DESIGN.md.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
    float* out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4 /*waves*/ * blocks;
    printf("acc=%2d blocks/CU=%d: %.3f ms, %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<16>(1, 20000);
    run<16>(2, 20000);
    run<16>(3, 20000);
    run<4>(2, 40000);
    run<8>(1, 40000);  // 8 accumulators = the (8x16)x64 window tile: dependent distance 8 MFMAs
    run<8>(2, 40000);
    run<8>(3, 40000);
    run<4>(3, 40000);
    run<16>(3, 200000);  // ~1 s: sustained clocks
    return 0;
}
