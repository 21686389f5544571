// tools/probe3.hip -- VALU issue / latency matrix for the Goertzel recurrence on gfx950: packed (v_pk_*_f32) against
// plain fp32 ops, 1..8 independent chains per lane, 1..4 waves per SIMD, each chain = (mul, sub, add) dependent per
// step, chains phased as Bank::step does.  Reports wall cycles per step per SIMD (2.4 GHz nominal) and s_memtime
// ticks, so the clock can be read off too.  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe3.hip -o tools/probe3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// PK: packed pairs; CH chains; EXTRA: per step also one sdwa convert feeding the add, and the energy mul + add
template <bool PK, int CH, bool EXTRA>
__global__ __launch_bounds__(64) void k(float *out, long long *cyc, int iters, float seed, int word)
{
    f2 a[CH], b[CH], f[CH];
    for (int i = 0; i < CH; i++) { a[i] = f2{seed*(i + 1), seed + i}; b[i] = f2{seed*0.5f + threadIdx.x, seed*0.25f}; f[i] = f2{1.0f + 0.001f*i, 1.0f - 0.001f*i}; }
    float energy = 0.0f;
    int w = word;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            f2 xx = f2{seed, seed};
            if (EXTRA)
            {
                float x, e2;
                asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(x) : "v"(w));
                asm volatile("v_mul_f32 %0, %1, %1" : "=v"(e2) : "v"(x));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(energy) : "v"(e2));
                xx = f2{x, x};
            }
            f2 t[CH];
            if (PK)
            {
#pragma unroll
                for (int i = 0; i < CH; i++) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t[i]) : "v"(f[i]), "v"(b[i]));
#pragma unroll
                for (int i = 0; i < CH; i++) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t[i]) : "v"(t[i]), "v"(a[i]));
#pragma unroll
                for (int i = 0; i < CH; i++) { a[i] = b[i]; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(b[i]) : "v"(t[i]), "v"(xx)); }
            }
            else
            {
#pragma unroll
                for (int i = 0; i < CH; i++) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t[i].x) : "v"(f[i].x), "v"(b[i].x));
#pragma unroll
                for (int i = 0; i < CH; i++) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t[i].x) : "v"(t[i].x), "v"(a[i].x));
#pragma unroll
                for (int i = 0; i < CH; i++) { a[i].x = b[i].x; asm volatile("v_add_f32 %0, %1, %2" : "=v"(b[i].x) : "v"(t[i].x), "v"(xx.x)); }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = energy;
    for (int i = 0; i < CH; i++) s += a[i].x + a[i].y + b[i].x + b[i].y;
    out[blockIdx.x*64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <bool PK, int CH, bool EXTRA>
static void run(int waves_per_simd)
{
    float *out; long long *cyc; const int blocks = 256*4*waves_per_simd; const int iters = 4000;
    CK(hipMalloc(&out, blocks*64*sizeof(float))); CK(hipMalloc(&cyc, 8));
    hipLaunchKernelGGL((k<PK, CH, EXTRA>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0f, 77);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<PK, CH, EXTRA>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0f, 77); hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double steps = (double) iters*8;
    const int bins = PK ? 2*CH : CH;
    const double cyc_step_simd = ms*1e-3*2.4e9/steps/waves_per_simd;        // wall cycles one step of one wave costs its SIMD
    printf("%s chains=%d bins/lane=%2d %s waves/SIMD=%d: ticks/step %7.2f  wall %8.1f us  %6.1f cyc/step/wave-on-SIMD  = %5.2f cyc per bin-step  (8-bin sample: %6.1f)\n",
           PK ? "pk " : "f32", CH, bins, EXTRA ? "+cvt+energy" : "           ", waves_per_simd, c/steps, ms*1e3, cyc_step_simd, cyc_step_simd/bins, cyc_step_simd/bins*8);
    hipFree(out); hipFree(cyc);
}

#include "../spandsp_amd/csrc/tone_dev.hpp"

// The product's two-sample block (Bank<8>::step2) in a loop: FACS = coefficients in SGPRs, EXTRA = converts + energy dealt
// around the block as tone_fast.hpp does
template <bool FACS, bool EXTRA>
__global__ __launch_bounds__(64) void kb(float *out, long long *cyc, int iters, spg::ToneLaunch L, int word)
{
    spg::Bank<8> bk;
    spg::f32x2 fac[4];
    for (int i = 0; i < 4; i++)
    {
        bk.a[i] = spg::f32x2{0.001f*(i + 1), 0.002f + threadIdx.x};
        bk.b[i] = spg::f32x2{0.5f + threadIdx.x, 0.25f};
        if (FACS)
            fac[i] = spg::f32x2{L.fac[2*i], L.fac[2*i + 1]};
        else
            fac[i] = spg::f32x2{L.fac[2*i] + 0.0f*threadIdx.x, L.fac[2*i + 1] + 0.0f*threadIdx.x};
    }
    float energy = 0.0f;
    int w = word + threadIdx.x;
    spg::f32x2 x = spg::f32x2{1.0f, 2.0f};
    spg::f32x2 sq = x*x;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            spg::f32x2 xn = x;
            if (EXTRA)
            {
                xn.x = spg::s16_lo(w);
                xn.y = spg::s16_hi(w);
                __builtin_amdgcn_sched_barrier(0);
                energy += sq.x;
                __builtin_amdgcn_sched_barrier(0);
            }
            bk.step2<FACS>(fac, x);
            if (EXTRA)
            {
                __builtin_amdgcn_sched_barrier(0);
                energy += sq.y;
                __builtin_amdgcn_sched_barrier(0);
                sq = xn*xn;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" : "+v"(w));
            }
            x = xn;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = energy;
    for (int i = 0; i < 4; i++) s += bk.a[i].x + bk.a[i].y + bk.b[i].x + bk.b[i].y;
    out[blockIdx.x*64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <bool FACS, bool EXTRA>
static void runb(int waves_per_simd)
{
    float *out; long long *cyc; const int blocks = 256*4*waves_per_simd; const int iters = 4000;
    CK(hipMalloc(&out, blocks*64*sizeof(float))); CK(hipMalloc(&cyc, 8));
    spg::ToneLaunch L; memset(&L, 0, sizeof(L));
    for (int i = 0; i < 16; i++) L.fac[i] = 1.0f + 0.001f*i;
    hipLaunchKernelGGL((kb<FACS, EXTRA>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, L, 77);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((kb<FACS, EXTRA>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, L, 77); hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double pairs = (double) iters*8;
    printf("step2 block fac in %s %s waves/SIMD=%d: ticks/sample %7.2f  wall %8.1f us  %6.1f nominal cyc/sample/wave-on-SIMD\n",
           FACS ? "SGPR" : "VGPR", EXTRA ? "+cvt+energy" : "           ", waves_per_simd, c/pairs/2, ms*1e3, ms*1e-3*2.4e9/pairs/2/waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main(int argc, char **argv)
{
    if (argc > 1)
    {
        for (int w = 1; w <= 4; w *= 2)
        {
            runb<false, false>(w); runb<true, false>(w); runb<false, true>(w); runb<true, true>(w);
        }
        return 0;
    }
    for (int w = 1; w <= 4; w *= 2)
    {
        run<true, 1, false>(w); run<true, 2, false>(w); run<true, 4, false>(w); run<true, 8, false>(w);
        run<false, 2, false>(w); run<false, 4, false>(w); run<false, 8, false>(w); run<false, 16, false>(w);
        run<true, 2, true>(w); run<true, 4, true>(w); run<false, 4, true>(w); run<false, 8, true>(w);
    }
    run<true, 4, false>(8); run<false, 8, false>(8); run<true, 4, true>(8); run<false, 8, true>(8);
    return 0;
}
