// Do two dependent chains of SMALL kernels on two hardware queues overlap?  (round 6: the class surface runs the depth net and the
// flow net on two streams, and the time at which both are done equals the SUM of their solo times.)
// Each chain: CH launches of a kernel with G workgroups x 256 threads that (a) spins ~T us of FMAs and (b) optionally streams
// `kb` KB per workgroup through memory (read + write).  Timed: chain A alone, chain B alone, both at once, and both in ONE stream.
//   hipcc --offload-arch=gfx950 -O2 -o overlap_probe overlap_probe.hip ; GPU_MAX_HW_QUEUES=12 ./overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__global__ void k_work(float* buf, int kb, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
    float4* p = reinterpret_cast<float4*>(buf) + (size_t)blockIdx.x * kb * 64;  // kb KB per workgroup = kb * 64 float4
    float4 acc = {a, 0, 0, 0};
    for (int i = threadIdx.x; i < kb * 64; i += blockDim.x) {
        float4 v = p[i];
        acc.x += v.x; acc.y += v.y;
        v.x += 1.f;
        p[i] = v;
    }
    if (acc.x == 123.456f) buf[0] = acc.y;
}
__global__ void k_empty() {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static float run(hipStream_t sa, hipStream_t sb, bool doA, bool doB, float* bufA, float* bufB, int G, int kb, int iters, int CH,
                 hipEvent_t e0, hipEvent_t ea, hipEvent_t eb, hipStream_t s0) {
    hipDeviceSynchronize();
    hipEventRecord(e0, s0);
    hipStreamWaitEvent(sa, e0, 0);
    hipStreamWaitEvent(sb, e0, 0);
    for (int k = 0; k < CH; ++k) {  // interleaved submission, as a host thread enqueuing two nets would
        if (doA) hipLaunchKernelGGL(k_work, G, 256, 0, sa, bufA, kb, iters);
        if (doB) hipLaunchKernelGGL(k_work, G, 256, 0, sb, bufB, kb, iters);
    }
    hipEventRecord(ea, sa);
    hipEventRecord(eb, sb);
    hipEventSynchronize(ea);
    hipEventSynchronize(eb);
    float ta = 0, tb = 0;
    hipEventElapsedTime(&ta, e0, ea);
    hipEventElapsedTime(&tb, e0, eb);
    return std::max(ta, tb) * 1e3f;
}
int main(int argc, char** argv) {
    const int CH = 40;
    hipStream_t s0, sa, sb;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    float *bufA, *bufB;
    CK(hipMalloc(&bufA, 256u << 20));
    CK(hipMalloc(&bufB, 256u << 20));
    hipEvent_t e0, ea, eb;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    hipLaunchKernelGGL(k_empty, 1, 1, 0, sa); hipLaunchKernelGGL(k_empty, 1, 1, 0, sb);
    CK(hipDeviceSynchronize());
    printf("chains of %d dependent launches; us for: A alone | B alone | A and B on two streams | A and B in ONE stream\n", CH);
    const int Gs[] = {1, 32, 100, 240}, kbs[] = {0, 64, 512}, its[] = {0, 4000};
    for (int G : Gs) for (int kb : kbs) for (int it : its) {
        if ((size_t)G * kb * 1024 > (256u << 20)) continue;
        float a = 0, b = 0, ab = 0, one = 0;
        for (int rep = 0; rep < 3; ++rep) {  // last repetition counts
            a = run(sa, sb, true, false, bufA, bufB, G, kb, it, CH, e0, ea, eb, s0);
            b = run(sa, sb, false, true, bufA, bufB, G, kb, it, CH, e0, ea, eb, s0);
            ab = run(sa, sb, true, true, bufA, bufB, G, kb, it, CH, e0, ea, eb, s0);
            one = run(sa, sa, true, true, bufA, bufB, G, kb, it, CH, e0, ea, eb, s0);
        }
        printf("G=%4d wgs, %4d KB/wg, %5d fma iters: %7.0f | %7.0f | %7.0f | %7.0f   (per launch alone %.1f us; overlap gain %.2f)\n", G, kb, it, a, b,
               ab, one, a / CH, one / ab);
    }
    return 0;
}
