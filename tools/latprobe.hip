// tools/latprobe.hip -- dependent-issue latency / issue interval of the VALU ops the Goertzel recurrence is made of,
// measured with s_memtime around a long unrolled loop, ONE wave per SIMD (and optionally 2).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int CH, int MODE>
__global__ __launch_bounds__(64) void k(float *out, long long *cyc, int iters, float seed)
{
    f2 a[CH], b[CH], f[CH];
    for (int i = 0; i < CH; i++) { a[i] = f2{seed*(i + 1), seed + i}; b[i] = f2{seed*0.5f + threadIdx.x, seed*0.25f}; f[i] = f2{1.0f + 0.001f*i, 1.0f - 0.001f*i}; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            if (MODE == 0)          // pk: 3 dependent ops per chain per step (mul, sub, add) phased across chains
            {
#pragma unroll
                for (int i = 0; i < CH; i++) { f2 t; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(f[i]), "v"(a[i])); b[i] = t; }
#pragma unroll
                for (int i = 0; i < CH; i++) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(a[i])); }
#pragma unroll
                for (int i = 0; i < CH; i++) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(f[i]), "v"(a[i])); }
            }
            else if (MODE >= 8)     // kernel-like variants, phased (bit 0: SGPR fac, bit 1: neg / op_sel modifiers, bit 2: sdwa cvt)
            {
                f2 xx = f2{seed + u, seed - u};
                if (MODE & 4)
                    asm volatile("v_cvt_f32_i32_sdwa %0, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_cvt_f32_i32_sdwa %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(xx.y), "=v"(xx.x) : "v"(it + u));
#pragma unroll
                for (int i = 0; i < CH; i++)
                {
                    f2 t;
                    if (MODE & 1)
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "s"(f[i]), "v"(a[i]));
                    else
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(f[i]), "v"(a[i]));
                    b[i] = t;
                }
#pragma unroll
                for (int i = 0; i < CH; i++)
                {
                    if (MODE & 2)
                        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a[i]) : "v"(b[i]), "v"(a[i]));
                    else
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(a[i]));
                }
#pragma unroll
                for (int i = 0; i < CH; i++)
                {
                    if (MODE & 2)
                        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(a[i]) : "v"(xx), "v"(a[i]));
                    else
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(xx), "v"(a[i]));
                }
            }
            else if (MODE == 1)     // scalar f32 on .x only
            {
#pragma unroll
                for (int i = 0; i < CH; i++) { float t; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(f[i].x), "v"(a[i].x)); b[i].x = t; }
#pragma unroll
                for (int i = 0; i < CH; i++) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i].x) : "v"(b[i].x), "v"(a[i].x)); }
#pragma unroll
                for (int i = 0; i < CH; i++) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i].x) : "v"(f[i].x), "v"(a[i].x)); }
            }
            else if (MODE == 2)     // pk, fully serial single dependent chain per CH (back-to-back dependent)
            {
#pragma unroll
                for (int i = 0; i < CH; i++) { asm volatile("v_pk_mul_f32 %0, %1, %0\n\tv_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(f[i]), "v"(b[i])); }
            }
            else                    // scalar fully serial
            {
#pragma unroll
                for (int i = 0; i < CH; i++) { asm volatile("v_mul_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(f[i].x), "v"(b[i].x)); }
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < CH; i++) s += a[i].x + a[i].y + b[i].x;
    out[blockIdx.x*64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CH, int MODE>
static void run(const char *name, int waves_per_simd)
{
    float *out; long long *cyc; const int blocks = 256*4*waves_per_simd; const int iters = 2000;
    CK(hipMalloc(&out, blocks*64*sizeof(float))); CK(hipMalloc(&cyc, 8));
    hipLaunchKernelGGL((k<CH, MODE>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<CH, MODE>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0f); hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double ops = (double) iters*8*CH*3;
    printf("%-14s chains=%d waves/SIMD=%d: %6.2f s_memtime ticks/op (x24 = %6.1f core cycles at 2.4 GHz if 100 MHz), wall %8.1f us => %6.2f core-cycles/op/wave\n",
           name, CH, waves_per_simd, c/ops, c/ops*24.0, ms*1e3, ms*1e-3*2.4e9/ops/waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<4, 8>("pk plain x", 1); run<4, 9>("pk sgpr fac", 1); run<4, 10>("pk neg/opsel", 1); run<4, 11>("pk sgpr+neg", 1); run<4, 15>("pk all+cvt", 1);
    run<4, 8>("pk plain x", 2); run<4, 15>("pk all+cvt", 2);
    return 0;
    run<1, 2>("pk serial", 1); run<1, 3>("f32 serial", 1);
    run<1, 0>("pk phased", 1); run<2, 0>("pk phased", 1); run<4, 0>("pk phased", 1); run<8, 0>("pk phased", 1);
    run<1, 1>("f32 phased", 1); run<2, 1>("f32 phased", 1); run<4, 1>("f32 phased", 1); run<8, 1>("f32 phased", 1);
    run<2, 0>("pk phased", 2); run<4, 0>("pk phased", 2); run<8, 1>("f32 phased", 2);
    run<4, 0>("pk phased", 4); run<8, 1>("f32 phased", 4);
    return 0;
}
