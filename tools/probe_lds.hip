// probe_lds.hip -- what the operands of a short inner product cost a LONE wave per SIMD when they come from LDS (TEST / BUILDER
// TOOL, never in the product): the V.29 / V.17 receivers' equaliser sum (v29_quad.hpp: 33 terms, x as 8-byte pairs, taps as
// two words three apart) in the forms one could give it.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize
// The arrays have the kernel's strides from channel to channel (264 and 100 words).  Prints cycles per term and wave with a wave
// on every SIMD of a CU (as the kernel runs) and with one wave on a CU alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x2v __attribute__((ext_vector_type(2)));
constexpr int N = 33;

template <int MODE, int SKEW = 0>
__global__ __launch_bounds__(256) void probe(float *out, int iters, int seed)
{
    if (threadIdx.x >= blockDim.x)
        return;
    __shared__ float2 s_x[4*16*132 + 8];
    __shared__ float s_c[4*16*100];
    const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3, role = lane & 3, cw = lane >> 2;
    float2 *X = s_x + (wv*16 + cw)*132 + ((cw & 4)  ?  SKEW  :  0);       // SKEW: every other group of four channels of a wave that many pairs further on
    float *Cc = s_c + (wv*16 + cw)*100;
    for (int i = role;  i < 132;  i += 4)
        X[i] = make_float2(1.0f + 0.001f*i + seed, 0.5f - 0.002f*i);
    for (int i = role;  i < 100;  i += 4)
        Cc[i] = 0.01f*i - 0.3f;
    __syncthreads();
    float tot = 0.0f;
    float cr[2*N];
    if (MODE == 2  ||  MODE == 6)
    {
        for (int i = 0;  i < N;  i++)
        {
            cr[2*i] = Cc[3*i + (role & 1)];
            cr[2*i + 1] = Cc[3*i + (role & 1) + 1];
        }
    }
    int pos = seed & 15;
    for (int it = 0;  it < iters;  it++)
    {
        asm volatile("" : "+v"(pos));
        const float2 *x = &X[pos + ((role & 2)  ?  N  :  0)];
        const float *c = &Cc[role & 1];
        float acc = 0.0f;
        if (MODE == 0)
        {
            // arithmetic only: operands in registers (the same every term)
            float a = tot + 1.0f, b = 0.5f, d = 0.25f, e = 0.125f;
            asm volatile("" : "+v"(a), "+v"(b), "+v"(d), "+v"(e));
#pragma unroll
            for (int i = 0;  i < N;  i++)
            {
                acc += a*d - b*e;
                asm volatile("" : "+v"(a));
            }
        }
        else if (MODE == 1)
        {
            // as the kernel has it
#pragma unroll
            for (int i = 0;  i < N;  i++)
                acc += x[i].x*c[3*i] - x[i].y*c[3*i + 1];
        }
        else if (MODE == 2)
        {
            // taps in registers
#pragma unroll
            for (int i = 0;  i < N;  i++)
                acc += x[i].x*cr[2*i] - x[i].y*cr[2*i + 1];
        }
        else if (MODE == 3)
        {
            // loads only (every operand folded into one value by integer ors: 1 VALU per loaded word)
            uint32_t o = 0;
#pragma unroll
            for (int i = 0;  i < N;  i++)
                o |= __float_as_uint(x[i].x) | __float_as_uint(x[i].y) | __float_as_uint(c[3*i]) | __float_as_uint(c[3*i + 1]);
            acc = __uint_as_float(o);
        }
        else if (MODE == 4)
        {
            // packed multiply
#pragma unroll
            for (int i = 0;  i < N;  i++)
            {
                const f32x2v p = (f32x2v) {x[i].x, x[i].y}*(f32x2v) {c[3*i], c[3*i + 1]};
                acc += p.x - p.y;
            }
        }
        else if (MODE == 5)
        {
            // all operands first (one wait), then the chain
            float2 xs[N];
            float ca[N], cb[N];
#pragma unroll
            for (int i = 0;  i < N;  i++)
            {
                xs[i] = x[i];
                ca[i] = c[3*i];
                cb[i] = c[3*i + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0;  i < N;  i++)
                acc += xs[i].x*ca[i] - xs[i].y*cb[i];
        }
        else if (MODE == 6)
        {
            // taps in registers, packed multiply
#pragma unroll
            for (int i = 0;  i < N;  i++)
            {
                const f32x2v p = (f32x2v) {x[i].x, x[i].y}*(f32x2v) {cr[2*i], cr[2*i + 1]};
                acc += p.x - p.y;
            }
        }
        else if (MODE == 7)
        {
            // taps as aligned pairs {re, im} in an array of their own: one 8-byte read per term (what a layout with room for it gives)
            const float2 *c2 = (const float2 *) (Cc + 2*(role & 1));
#pragma unroll
            for (int i = 0;  i < N;  i++)
            {
                const float2 cv = c2[i];
                acc += x[i].x*cv.x - x[i].y*cv.y;
            }
        }
        else if (MODE == 8)
        {
            // one word of x per lane (even lanes the real part, odd lanes the imaginary one), the other from the neighbour's register
            const float *x1 = (const float *) x + (role & 1);
#pragma unroll
            for (int i = 0;  i < N;  i++)
            {
                const float xv = x1[2*i];
                const float xr = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(xv), 0xA0, 0xF, 0xF, true));
                const float xi = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(xv), 0xF5, 0xF, 0xF, true));
                acc += xr*c[3*i] - xi*c[3*i + 1];
            }
        }
        tot += acc;
        pos = (pos + 1) & 15;
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = tot;
}

template <int MODE, int SKEW = 0>
static void run(const char *name, float *d_out)
{
    const int iters = 2000;
    printf("%-58s", name);
    for (int w = 4;  w >= 1;  w -= 3)
    {
        const int blocks = 256;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        std::vector<float> t;
        for (int r = 0;  r < 7;  r++)
        {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((probe<MODE, SKEW>), dim3(blocks), dim3(64*w), 0, 0, d_out, iters, r);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("  %d wave(s) a CU %6.1f cyc/term", w, t[1]*1e-3*2.4e9/((double) iters*N));
    }
    printf("\n");
}

int main()
{
    float *d_out;
    hipMalloc(&d_out, 2*256*256*sizeof(float));
    run<0>("arithmetic only (mul, mul, sub, add; operands in registers)", d_out);
    run<1>("x and taps from LDS, as v29_quad.hpp", d_out);
    run<2>("x from LDS, taps in registers", d_out);
    run<3>("the LDS reads alone (an or per word)", d_out);
    run<4>("x and taps from LDS, packed multiply", d_out);
    run<5>("x and taps from LDS, all reads, one wait, the chain", d_out);
    run<6>("x from LDS, taps in registers, packed multiply", d_out);
    run<7>("x and taps from LDS, taps as aligned pairs", d_out);
    run<8>("one word of x per lane + DPP, taps from LDS", d_out);
    run<8, 2>("   ... every other four channels four words further on", d_out);
    run<1, 2>("as v29_quad.hpp, every other four channels 4 words on", d_out);
    hipFree(d_out);
    return 0;
}
