// Which HIP streams share a hardware dispatch pipe?  For every ordered pair (a, b) of a pool of streams: stream a is kept busy
// with back-to-back single-workgroup kernels of ~20 us, stream b runs a chain of dependent empty kernels; the chain's
// latency goes up when the two streams' hardware queues are served by the same pipe of the command processor.
//   hipcc --offload-arch=gfx950 -O2 -o queue_probe queue_probe.hip ; GPU_MAX_HW_QUEUES=12 ./queue_probe [pre_streams]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_spin(long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_empty() {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int pre = argc > 1 ? atoi(argv[1]) : 0, P = argc > 2 ? atoi(argv[2]) : 12, CH = 60;
    std::vector<hipStream_t> dummy(pre), s(P);
    for (auto& d : dummy) { CK(hipStreamCreateWithFlags(&d, hipStreamNonBlocking)); hipLaunchKernelGGL(k_empty, 1, 1, 0, d); }
    for (auto& q : s) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& q : s) hipLaunchKernelGGL(k_empty, 1, 1, 0, q);
    CK(hipDeviceSynchronize());
    const long long spin = 2000;  // wall_clock64 ticks at 100 MHz: 20 us
    printf("pre=%d pool=%d: chain latency (us) of %d empty kernels on stream b (column) while stream a (row) is busy\n", pre, P, CH);
    for (int a = -1; a < P; ++a) {
        printf("%3d |", a);
        for (int b = 0; b < P; ++b) {
            if (a == b) { printf("     - "); continue; }
            if (a >= 0) for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(k_spin, 1, 64, 0, s[a], spin);
            CK(hipEventRecord(e0, s[b]));
            for (int k = 0; k < CH; ++k) hipLaunchKernelGGL(k_empty, 1, 64, 0, s[b]);
            CK(hipEventRecord(e1, s[b]));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipDeviceSynchronize());
            printf(" %6.0f", ms * 1e3);
        }
        printf("\n");
    }
    return 0;
}
