// Does v_mfma_f32_16x16x4_f32 sustain its 32-cycle issue when the A / B source registers change from instruction to
// instruction (as in the conv kernels: 4 pixel fragments x 2 weight fragments x 4 k-slices per tap) and the
// accumulators live in AGPRs?  Pure register loop, no memory traffic.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    f32x4 acc[4][2];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0, 0, 0, 0};
    f32x4 fa[4], fb[2];
    for (int i = 0; i < 4; i++) fa[i] = *(const f32x4*)(in + (threadIdx.x * 6 + i) * 4);
    for (int j = 0; j < 2; j++) fb[j] = *(const f32x4*)(in + (threadIdx.x * 6 + 4 + j) * 4);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (MODE == 0)  // one pair of source registers for every instruction
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[0][0], fa[0][0], acc[i][j], 0, 0, 0);
                    else  // the conv kernels' pattern
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][r], fa[i][r], acc[i][j], 0, 0, 0);
                }
    }
    f32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 2; j++) s += acc[i][j];
    *(f32x4*)(out + (blockIdx.x * 256 + threadIdx.x) * 4) = s;
}
template <int MODE>
void run(int blocks_per_cu, int iters) {
    float *in, *out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&in, sizeof(float) * 256 * 24);
    hipMemset(in, 0, sizeof(float) * 256 * 24);
    hipMalloc(&out, sizeof(float) * 1024 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(in, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * 32.0 * iters * 4 * blocks;
    printf("%s sources, %d workgroups/CU: %.3f ms, %.1f TFLOP/s\n", MODE ? "varying" : "fixed", blocks_per_cu, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(in);
    hipFree(out);
}
int main() {
    for (int b = 1; b <= 3; b++) {
        run<0>(b, 5000);
        run<1>(b, 5000);
    }
    return 0;
}
