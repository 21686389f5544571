// tools/probe6.hip -- issue rate of the integer instructions of the echo canceller kernels on gfx950: cycles per wave64
// instruction with 1, 2 and 4 waves per SIMD, eight independent instructions per loop body (no dependent chains).
// Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe6.hip -o tools/probe6
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
template <int OP>
__global__ __launch_bounds__(64) void k(int *out, int iters, int seed)
{
    int a[8], b[8], c[8];
    for (int i = 0; i < 8; i++) { a[i] = seed*(i + 3) + threadIdx.x; b[i] = seed ^ (i*77); c[i] = i; }
    int fifteen = 15; asm volatile("" : "+v"(fifteen));
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            if (OP == 0) {
#define I0(i) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]));
                BODY8(I0) }
            if (OP == 1) {
#define I1(i) asm volatile("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]));
                BODY8(I1) }
            if (OP == 2) {
#define I2(i) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]));
                BODY8(I2) }
            if (OP == 3) {
#define I3(i) asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(c[i]) : "v"(fifteen), "v"(a[i]));
                BODY8(I3) }
            if (OP == 4) {
#define I4(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(c[i]) : "v"(a[i]), "v"(b[i]), "s"(0x05040302));
                BODY8(I4) }
            if (OP == 5) {
#define I5(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(c[i]) : "v"(a[i]));
                BODY8(I5) }
            if (OP == 6) {
#define I6(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(c[i]) : "v"(a[i]));
                BODY8(I6) }
            if (OP == 7) {       // the exec round trip of one LMS piece around 8 mads
                unsigned long long sv;
                asm volatile("s_and_saveexec_b64 %0, %1" : "=&s"(sv) : "s"(0x5555555555555555ull) : "scc");
                BODY8(I1)
                asm volatile("s_mov_b64 exec, %0" :: "s"(sv));
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 8; i++) s += c[i];
    out[blockIdx.x*64 + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name)
{
    for (int w = 1; w <= 4; w *= 2)
    {
        int *out; const int blocks = 256*4*w; const int iters = 2000;
        CK(hipMalloc(&out, blocks*64*sizeof(int)));
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(64), 0, 0, out, iters, 3);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(64), 0, 0, out, iters, 3); hipEventRecord(e1);
        CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double) iters*32*w;       // instructions per SIMD
        printf("%-28s waves/SIMD %d: %6.2f cycles per instruction per SIMD (2.4 GHz nominal)\n", name, w, ms*1e-3*2.4e9/n);
        hipFree(out);
    }
}

int main()
{
    run<5>("v_add_u32");
    run<0>("v_mad_i32_i24");
    run<1>("v_mad_i32_i16 op_sel");
    run<2>("v_dot2c_i32_i16");
    run<3>("v_lshrrev_b32_sdwa WORD_1");
    run<4>("v_perm_b32");
    run<6>("v_mov_b32_dpp quad_perm");
    run<7>("8 x v_mad_i32_i16 in exec pair");
    return 0;
}
