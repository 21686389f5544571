// tools/probe4.hip -- what a block end costs a lone wave: Bank<8>::finish + DtmfDet::decide in a loop, ticks per call.
// Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe4.hip -o tools/probe4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../spandsp_amd/csrc/tone_dev.hpp"
using namespace spg;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void kd(float *out, long long *cyc, int iters, ToneLaunch L)
{
    Bank<8> bk;
    f32x2 fac[4];
    DtmfDet<false> det;
    for (int i = 0; i < 4; i++)
        fac[i] = f32x2{L.fac[2*i], L.fac[2*i + 1]};
    float energy = 1000.0f + threadIdx.x;
    uint32_t w0 = 0;
    int32_t w1 = 0;
    uint32_t acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++)
    {
        for (int i = 0; i < 4; i++)
        {
            bk.a[i] = f32x2{100.0f*(i + 1) + it, 200.0f + threadIdx.x + (float) (acc & 7)};
            bk.b[i] = f32x2{50.0f + threadIdx.x*(i + 1), 25.0f*(it & 15)};
        }
        float e[8];
        if (MODE != 2)
            bk.finish(fac, e);
        else
            for (int i = 0; i < 8; i++) e[i] = bk.a[i >> 1].x*(i + 1);
        if (MODE != 1)
            acc += det.decide(L, e, energy, w0, w1, (int) threadIdx.x, 0, true);
        else
            for (int i = 0; i < 8; i++) acc += (uint32_t) e[i];
        energy += 3.0f;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x*64 + threadIdx.x] = (float) acc + (float) w0 + (float) w1;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static void run(const char *name)
{
    float *out; long long *cyc; const int iters = 2000;
    CK(hipMalloc(&out, 1024*64*sizeof(float))); CK(hipMalloc(&cyc, 8));
    ToneLaunch L; memset(&L, 0, sizeof(L));
    for (int i = 0; i < 16; i++) L.fac[i] = 1.0f + 0.001f*i;
    L.threshold = 171029200.0f; L.normal_twist = 6.309f; L.reverse_twist = 2.512f; L.n_ch = 64;
    hipLaunchKernelGGL((kd<MODE>), dim3(1024), dim3(64), 0, 0, out, cyc, iters, L);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((kd<MODE>), dim3(1024), dim3(64), 0, 0, out, cyc, iters, L);
    CK(hipDeviceSynchronize());
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-28s lone wave: %8.1f ticks per block end\n", name, (double) c/iters);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("finish + decide");
    run<1>("finish only");
    run<2>("decide only");
    return 0;
}
