// tools/probe5.hip -- what one dependent kernel boundary costs on this chip as a function of the launch shape:
// back-to-back launches of an empty kernel (same stream), workgroup size / LDS / kernarg size varied.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Big { float f[60]; };
template <int LDS>
__global__ void k_small(int *p) { __shared__ char l[LDS > 0 ? LDS : 4]; if (p == (int *) 1) p[0] = l[threadIdx.x]; }
template <int LDS>
__global__ void k_big(int *p, Big b) { __shared__ char l[LDS > 0 ? LDS : 4]; if (p == (int *) 1) p[0] = l[threadIdx.x] + (int) b.f[3]; }
template <int LDS>
__global__ void k_store(int *p) { __shared__ char l[LDS > 0 ? LDS : 4]; if (p == (int *) 1) p[0] = l[threadIdx.x]; p[blockIdx.x*blockDim.x + threadIdx.x] = 1; }

template <class F> static float tm(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms/reps*1e3f;
}
int main()
{
    int *p; CK(hipMalloc(&p, 1 << 24)); Big b = {};
    const int R = 500;
    printf("empty, 256 WG x  64 thr, no LDS, small args : %.2f us\n", tm([&] { hipLaunchKernelGGL(k_small<0>, dim3(256), dim3(64), 0, 0, p); }, R));
    printf("empty, 256 WG x 256 thr, no LDS, small args : %.2f us\n", tm([&] { hipLaunchKernelGGL(k_small<0>, dim3(256), dim3(256), 0, 0, p); }, R));
    printf("empty, 256 WG x 320 thr, no LDS, small args : %.2f us\n", tm([&] { hipLaunchKernelGGL(k_small<0>, dim3(256), dim3(320), 0, 0, p); }, R));
    printf("empty, 256 WG x 320 thr, 48K LDS, small args: %.2f us\n", tm([&] { hipLaunchKernelGGL(k_small<49152>, dim3(256), dim3(320), 0, 0, p); }, R));
    printf("empty, 256 WG x 320 thr, 48K LDS, 248B args : %.2f us\n", tm([&] { hipLaunchKernelGGL(k_big<49152>, dim3(256), dim3(320), 0, 0, p, b); }, R));
    printf("empty, 1024 WG x 64 thr, no LDS, small args : %.2f us\n", tm([&] { hipLaunchKernelGGL(k_small<0>, dim3(1024), dim3(64), 0, 0, p); }, R));
    printf("empty, 1024 WG x 64 thr, 12K LDS, small args: %.2f us\n", tm([&] { hipLaunchKernelGGL(k_small<12288>, dim3(1024), dim3(64), 0, 0, p); }, R));
    printf("4B store/thread, 256 WG x 320, 48K LDS      : %.2f us\n", tm([&] { hipLaunchKernelGGL(k_store<49152>, dim3(256), dim3(320), 0, 0, p); }, R));
    printf("4B store/thread, 1024 WG x 64               : %.2f us\n", tm([&] { hipLaunchKernelGGL(k_store<0>, dim3(1024), dim3(64), 0, 0, p); }, R));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto ts = [&](auto f, int reps) { hipEvent_t a, b2; CK(hipEventCreate(&a)); CK(hipEventCreate(&b2)); f(); f(); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b2, s)); CK(hipEventSynchronize(b2)); float ms; CK(hipEventElapsedTime(&ms, a, b2)); return ms/reps*1e3f; };
    printf("non-default stream: empty 256 x 320, 48K LDS : %.2f us\n", ts([&] { hipLaunchKernelGGL(k_small<49152>, dim3(256), dim3(320), 0, s, p); }, R));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_small<49152>, dim3(256), dim3(320), 0, s, p);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("graph of 20 such launches, per launch        : %.2f us\n", ts([&] { CK(hipGraphLaunch(ge, s)); }, 50)/20);
    return 0;
}
