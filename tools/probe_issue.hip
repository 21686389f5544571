// probe_issue.hip -- issue rate of the integer VALU instructions the echo canceller's common body is made of, on gfx950,
// with 1, 2, 3 and 4 waves on a SIMD (TEST / BUILDER TOOL, never in the product).  hipcc --offload-arch=gfx950 -O3
// Each wave runs ITER iterations of 64 instructions of one kind over 8 independent registers (dependent distance 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// LANES: how many lanes of each wave take part (the others leave at once: EXEC has only the low LANES bits set for the whole
// loop) -- round 5: does a wave with half or a quarter of its lanes alive issue its vector instructions faster?
template <int OP, int LANES = 64>
__global__ __launch_bounds__(256) void probe(int *out, int iters, int seed)
{
    if ((int) (threadIdx.x & 63) >= LANES)
        return;
    int r0 = seed + threadIdx.x, r1 = r0*3, r2 = r0*5, r3 = r0*7, r4 = r0*11, r5 = r0*13, r6 = r0*17, r7 = r0*19;
    int a = seed | 1, b = seed*3 + 1;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p0 = {1.0f + seed, 1.5f}, p1 = {2.0f, 2.5f}, p2 = {3.0f, 3.5f}, p3 = {4.0f, 4.5f};
    const f32x2 pc = {1.0000001f, 0.9999999f};
    for (int i = 0;  i < iters;  i++)
    {
        if constexpr (OP == 0)
        {
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r##k) : "v"(a));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 1)
        {
#define X(k) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(r##k) : "v"(a), "v"(b));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 2)
        {
#define X(k) asm volatile("v_bfe_i32 %0, %0, 15, 16" : "+v"(r##k));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 3)
        {
#define X(k) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r##k) : "v"(a));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 4)
        {
#define X(k) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r##k) : "v"(a));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 5)
        {
#define X(k) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(r##k));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 6)
        {
#define X(k) asm volatile("v_mul_i32_i24_sdwa %0, sext(%1), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(r##k) : "v"(a));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 7)
        {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##k) : "v"(a) : "vcc");
            REP64(X)
#undef X
        }
        else if constexpr (OP == 8)
        {
            // the canceller's mix: 4 mads then a bfe pair, dependent as in the LMS update (mad -> bfe of the same register 6 later)
#define X(k) asm volatile("v_mad_i32_i24 %0, %1, %2, %0\n\tv_bfe_i32 %3, %0, 15, 16" : "+v"(r##k), "+v"(b) : "v"(a), "v"(b));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        }
        else if constexpr (OP == 9)
        {
#define X(k) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\ts_and_b64 s[20:21], vcc, exec" :: "v"(r##k), "v"(a) : "vcc", "s20", "s21", "scc");
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        }
        else if constexpr (OP == 10)
        {
            // dependent chain: every instruction reads the one before
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r0) : "v"(a));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 11)
        {
#define X(k) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 13)
        {
#define X(k) asm volatile("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(r##k) : "v"(a), "v"(b));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 14)
        {
#define X(k) asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(r##k) : "v"(a), "v"(b));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 15)
        {
#define X(k) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(r##k) : "v"(a), "v"(b));
            REP64(X)
#undef X
        }
        else if constexpr (OP == 12)
        {
            // the tone kernels' packed multiply (register pairs: r0:r1 ... as 64-bit operands)
            asm volatile(
#define X(k) "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                REP8(X) REP8(X)
#undef X
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
        }
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + b + (int) (p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y);
}

template <int OP, int LANES = 64>
static void run(const char *name, int *d_out, int per)
{
    const int iters = 2000;
    printf("%-34s", name);
    for (int w = 1;  w <= 4;  w++)
    {
        const int blocks = 256*w;               // 4 waves a block, one per SIMD; w blocks a CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        std::vector<float> t;
        for (int r = 0;  r < 7;  r++)
        {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((probe<OP, LANES>), dim3(blocks), dim3(256), 0, 0, d_out, iters, r);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        const double cyc = t[1]*1e-3*2.4e9;     // (second fastest)
        const double insts = (double) iters*per*w;         // per SIMD
        printf("  w%d %5.2f cyc/inst", w, cyc/insts);
    }
    printf("\n");
}

int main()
{
    int *d_out;
    hipMalloc(&d_out, 256*4*256*sizeof(int));
    run<0>("v_add_u32 (8 independent)", d_out, 64);
    run<1>("v_mad_i32_i24", d_out, 64);
    run<2>("v_bfe_i32", d_out, 64);
    run<3>("v_mov_b32_dpp quad_perm", d_out, 64);
    run<4>("v_add_u32_dpp quad_perm", d_out, 64);
    run<5>("v_ashrrev_i32", d_out, 64);
    run<6>("v_mul_i32_i24_sdwa", d_out, 64);
    run<7>("v_cndmask_b32 (vcc)", d_out, 64);
    run<8>("mad + bfe pairs", d_out, 64);
    run<9>("v_cmp + s_and pairs", d_out, 64);
    run<10>("v_add_u32 dependent chain", d_out, 64);
    run<11>("v_mad_i32_i24 dependent chain", d_out, 64);
    run<12>("v_pk_mul_f32 (4 independent)", d_out, 64);
    run<13>("v_mad_i32_i16 op_sel hi", d_out, 64);
    run<14>("v_mad_i32_i16", d_out, 64);
    run<15>("v_dot2_i32_i16", d_out, 64);
    // round 5: waves with 32 / 16 of their 64 lanes alive
    run<0, 32>("v_add_u32, 32 lanes alive", d_out, 64);
    run<0, 16>("v_add_u32, 16 lanes alive", d_out, 64);
    run<1, 32>("v_mad_i32_i24, 32 lanes alive", d_out, 64);
    run<1, 16>("v_mad_i32_i24, 16 lanes alive", d_out, 64);
    run<12, 32>("v_pk_mul_f32, 32 lanes alive", d_out, 64);
    run<12, 16>("v_pk_mul_f32, 16 lanes alive", d_out, 64);
    run<3, 32>("v_mov_b32_dpp, 32 lanes alive", d_out, 64);
    hipFree(d_out);
    return 0;
}
