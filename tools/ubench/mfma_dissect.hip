// Dissection of the conv window kernel's tap loop: the bare 32-MFMA burst (v_mfma_f32_16x16x4_f32, 8 accumulators), then
// with the loop's other ingredients added one at a time, at 3 workgroups (12 waves) per CU like the real kernel:
//   1 = + 6 x ds_read_b128 of the operand fragments per tap      2 = 1 + ds_write_b128 of a weight tile
//   3 = 2 + __syncthreads() per tap                             4 = 3 + one global_load_dwordx4 per tap (L2 resident)
// hipcc --offload-arch=gfx950 -O3 mfma_dissect.hip -o mfma_dissect
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ g, float* out, int taps) {
    __shared__ __attribute__((aligned(16))) float lds[9216];
    const int t = threadIdx.x, lane = t & 63, li = lane & 15, kq = lane >> 4, wave = t >> 6;
    for (int i = t; i < 9216; i += 256) lds[i] = 1e-3f * (i & 255);
    __syncthreads();
    f32x4 acc[4][2];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0, 0, 0, 0};
    f32x4 fa[4], fb[2], rb = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) fa[i] = *(const f32x4*)(lds + ((wave * 4 + i) * 18 + li) * 20 + kq * 4);
    for (int j = 0; j < 2; j++) fb[j] = *(const f32x4*)(lds + 7200 + (kq * 64 + j * 16 + li) * 4);
    for (int tap = 0; tap < taps; tap++) {
        if (MODE >= 4) rb = *(const f32x4*)(g + ((tap & 63) * 256 + t) * 4);
        if (MODE >= 1) {
            const int kx = tap % 3, ky = (tap / 3) % 3;
            for (int i = 0; i < 4; i++) fa[i] = *(const f32x4*)(lds + ((wave * 4 + i + ky) * 18 + li + kx) * 20 + kq * 4);
            for (int j = 0; j < 2; j++) fb[j] = *(const f32x4*)(lds + 7200 + (tap & 1) * 1024 + (kq * 64 + j * 16 + li) * 4);
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][r], fa[i][r], acc[i][j], 0, 0, 0);
        if (MODE >= 2) *(f32x4*)(lds + 7200 + ((tap + 1) & 1) * 1024 + t * 4) = MODE >= 4 ? rb : fb[0];
        if (MODE >= 3) __syncthreads();
    }
    f32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 2; j++) s += acc[i][j];
    *(f32x4*)(out + (blockIdx.x * 256 + t) * 4) = s;
}
template <int MODE>
void run(int blocks_per_cu, int taps) {
    float *g, *out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&g, sizeof(float) * 64 * 1024);
    hipMemset(g, 0, sizeof(float) * 64 * 1024);
    hipMalloc(&out, sizeof(float) * 1024 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(g, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(g, out, taps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * 32.0 * taps * 4 * blocks;
    printf("mode %d, %d workgroups/CU: %.3f ms, %.1f TFLOP/s (%.0f cycles per wave-tap at 2.4 GHz)\n", MODE, blocks_per_cu, ms,
           flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / taps);
    hipFree(g);
    hipFree(out);
}
int main() {
    for (int b = 1; b <= 3; b += 2) {
        run<0>(b, 20000);
        run<1>(b, 20000);
        run<2>(b, 20000);
        run<3>(b, 20000);
        run<4>(b, 20000);
    }
    return 0;
}
