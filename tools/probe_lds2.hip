// probe_lds2.hip -- the cost of LDS read instructions by form for a lone wave per SIMD (TEST / BUILDER TOOL, never in the product):
// cycles per instruction and per byte and lane, with lane-linear addresses (no bank conflict possible) and with the receivers'
// per-channel addressing (a quad per channel, channels 264 words apart).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// FORM: 0 ds_read_b32, 1 ds_read_b64, 2 ds_read_b128, 3 ds_read2_b32 (8 bytes apart), 4 ds_read2_b64 (16 bytes apart); ADDR: 0 linear, 1 per channel
template <int FORM, int ADDR>
__global__ __launch_bounds__(256) void probe(uint32_t *out, int iters, int seed)
{
    if (threadIdx.x >= blockDim.x)
        return;
    __shared__ uint32_t s[4*16*264 + 64];
    for (int i = threadIdx.x;  i < 4*16*264 + 64;  i += blockDim.x)
        s[i] = i*2654435761u + seed;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
    const int width = (FORM == 0)  ?  4  :  (FORM == 1  ||  FORM == 3)  ?  8  :  16;
    uint32_t base;
    if (ADDR == 0)
        base = (uint32_t) (uintptr_t) s + wv*16*264*4 + lane*width;
    else
        base = (uint32_t) (uintptr_t) s + (wv*16 + (lane >> 2))*264*4 + (lane & 3)*width;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int it = 0;  it < iters;  it++)
    {
        uint32_t b = base + ((it & 7) << 6);
        asm volatile("" : "+v"(b));
#define ONE(off) \
        if (FORM == 0) { uint32_t v; asm volatile("ds_read_b32 %0, %1 offset:" #off : "=v"(v) : "v"(b)); asm volatile("s_waitcnt lgkmcnt(4)\n\tv_or_b32 %0, %0, %1" : "+v"(a0) : "v"(v)); } \
        else if (FORM == 1) { uint64_t v; asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(v) : "v"(b)); asm volatile("s_waitcnt lgkmcnt(4)\n\tv_or_b32 %0, %0, %1" : "+v"(a0) : "v"((uint32_t) v)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(a1) : "v"((uint32_t) (v >> 32))); } \
        else if (FORM == 3) { uint64_t v; asm volatile("ds_read2_b32 %0, %1 offset0:" #off "/4 offset1:" #off "/4+2" : "=v"(v) : "v"(b)); asm volatile("s_waitcnt lgkmcnt(4)\n\tv_or_b32 %0, %0, %1" : "+v"(a0) : "v"((uint32_t) v)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(a1) : "v"((uint32_t) (v >> 32))); } \
        else if (FORM == 2) { uint4 v; asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(v) : "v"(b)); asm volatile("s_waitcnt lgkmcnt(4)\n\tv_or_b32 %0, %0, %1" : "+v"(a0) : "v"(v.x)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(a1) : "v"(v.y)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(a2) : "v"(v.z)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(a3) : "v"(v.w)); } \
        else { uint4 v; asm volatile("ds_read2_b64 %0, %1 offset0:" #off "/8 offset1:" #off "/8+2" : "=v"(v) : "v"(b)); asm volatile("s_waitcnt lgkmcnt(4)\n\tv_or_b32 %0, %0, %1" : "+v"(a0) : "v"(v.x)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(a1) : "v"(v.y)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(a2) : "v"(v.z)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(a3) : "v"(v.w)); }
        ONE(0) ONE(32) ONE(64) ONE(96) ONE(128) ONE(160) ONE(192) ONE(224)
        ONE(256) ONE(288) ONE(320) ONE(352) ONE(384) ONE(416) ONE(448) ONE(480)
#undef ONE
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    out[blockIdx.x*blockDim.x + threadIdx.x] = a0 | a1 | a2 | a3;
}

template <int FORM, int ADDR>
static void run(const char *name, uint32_t *d_out)
{
    const int iters = 4000;
    const int width = (FORM == 0)  ?  4  :  (FORM == 1  ||  FORM == 3)  ?  8  :  16;
    printf("%-46s", name);
    for (int w = 4;  w >= 1;  w -= 3)
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        std::vector<float> t;
        for (int r = 0;  r < 7;  r++)
        {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((probe<FORM, ADDR>), dim3(256), dim3(64*w), 0, 0, d_out, iters, r);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        const double cyc = t[1]*1e-3*2.4e9/((double) iters*16);
        printf("  %d wave(s) a CU %5.1f cyc/inst %5.2f cyc/(4 bytes a lane)", w, cyc, cyc/(width/4));
    }
    printf("\n");
}

int main()
{
    uint32_t *d_out;
    hipMalloc(&d_out, 256*256*sizeof(uint32_t));
    run<0, 0>("ds_read_b32   lane-linear", d_out);
    run<1, 0>("ds_read_b64   lane-linear", d_out);
    run<2, 0>("ds_read_b128  lane-linear", d_out);
    run<3, 0>("ds_read2_b32  lane-linear", d_out);
    run<4, 0>("ds_read2_b64  lane-linear", d_out);
    run<0, 1>("ds_read_b32   a quad per channel, 264 words", d_out);
    run<1, 1>("ds_read_b64   a quad per channel, 264 words", d_out);
    run<2, 1>("ds_read_b128  a quad per channel, 264 words", d_out);
    run<3, 1>("ds_read2_b32  a quad per channel, 264 words", d_out);
    run<4, 1>("ds_read2_b64  a quad per channel, 264 words", d_out);
    hipFree(d_out);
    return 0;
}
