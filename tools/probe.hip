// tools/probe.hip -- on-GPU micro-probes used to steer kernel design (not part of the
// product): (1) VALU issue rates of the unfused Goertzel recurrence written as scalar
// fp32 ops vs packed v_pk_* ops, (2) tone_bank_kernel<DtmfDet> launch time over channel
// counts and frame lengths, (3) a plain streaming read for the practical HBM ceiling.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe.hip -o tools/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../spandsp_amd/csrc/tone_dev.hpp"

using namespace spg;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

// 8 independent recurrences, scalar ops, unfused (contract off)
__global__ __launch_bounds__(256) void valu_scalar(float *out, int iters, float seed)
{
    float v2[8];
    float v3[8];
    float fac[8];
    for (int i = 0;  i < 8;  i++)
    {
        v2[i] = seed*(i + 1);
        v3[i] = seed*(i + 2) + threadIdx.x;
        fac[i] = 1.0f + 0.01f*i;
    }
    float x = seed;
    for (int it = 0;  it < iters;  it++)
    {
#pragma unroll
        for (int u = 0;  u < 4;  u++)
        {
#pragma unroll
            for (int i = 0;  i < 8;  i++)
            {
                float v1 = v2[i];
                v2[i] = v3[i];
                asm volatile("" : "+v"(v2[i]));     // keep scalar: defeat SLP packing
                v3[i] = fac[i]*v2[i] - v1 + x;
            }
        }
    }
    float s = 0.0f;
    for (int i = 0;  i < 8;  i++)
        s += v2[i] + v3[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

// same work as 4 packed pairs
__global__ __launch_bounds__(256) void valu_packed(float *out, int iters, float seed)
{
    float2v v2[4];
    float2v v3[4];
    float2v fac[4];
    for (int i = 0;  i < 4;  i++)
    {
        v2[i] = float2v{seed*(i + 1), seed*(i + 5)};
        v3[i] = float2v{seed*(i + 2) + threadIdx.x, seed*(i + 7)};
        fac[i] = float2v{1.0f + 0.01f*i, 1.0f + 0.02f*i};
    }
    float2v x = float2v{seed, seed};
    for (int it = 0;  it < iters;  it++)
    {
#pragma unroll
        for (int u = 0;  u < 4;  u++)
        {
#pragma unroll
            for (int i = 0;  i < 4;  i++)
            {
                float2v v1 = v2[i];
                v2[i] = v3[i];
                v3[i] = fac[i]*v2[i] - v1 + x;
            }
        }
    }
    float2v s = float2v{0.0f, 0.0f};
    for (int i = 0;  i < 4;  i++)
        s += v2[i] + v3[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s.x + s.y;
}

// fused variant for reference (not usable for bit-exact parity)
__global__ __launch_bounds__(256) void valu_fma(float *out, int iters, float seed)
{
    float v2[8];
    float v3[8];
    float fac[8];
    for (int i = 0;  i < 8;  i++)
    {
        v2[i] = seed*(i + 1);
        v3[i] = seed*(i + 2) + threadIdx.x;
        fac[i] = 1.0f + 0.01f*i;
    }
    float x = seed;
    for (int it = 0;  it < iters;  it++)
    {
#pragma unroll
        for (int u = 0;  u < 4;  u++)
        {
#pragma unroll
            for (int i = 0;  i < 8;  i++)
            {
                float v1 = v2[i];
                v2[i] = v3[i];
                asm volatile("" : "+v"(v2[i]));
                v3[i] = __builtin_fmaf(fac[i], v2[i], -v1) + x;
            }
        }
    }
    float s = 0.0f;
    for (int i = 0;  i < 8;  i++)
        s += v2[i] + v3[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void stream_read(const int4 *in, int4 *out, size_t n)
{
    int4 acc = make_int4(0, 0, 0, 0);
    for (size_t i = (size_t) blockIdx.x*blockDim.x + threadIdx.x;  i < n;  i += (size_t) gridDim.x*blockDim.x)
    {
        int4 v = in[i];
        acc.x ^= v.x;
        acc.y ^= v.y;
        acc.z ^= v.z;
        acc.w ^= v.w;
    }
    if (acc.x == 0x12345678)
        out[0] = acc;
}

template <class F>
static float time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0;  i < reps;  i++)
        launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms/reps;
}

static void probe_valu()
{
    float *out;
    const int blocks = 256*8;           // 8 blocks of 4 waves per CU
    CK(hipMalloc(&out, (size_t) blocks*256*sizeof(float)));
    const int iters = 2000;
    const double samples = (double) blocks*256*iters*4;     // lane-samples (8 bins each)
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(valu_scalar, dim3(blocks), dim3(256), 0, 0, out, iters, 0.001f); }, 5);
    printf("valu_scalar : %8.3f ms  %7.2f G lane-samples/s  (%6.2f T lane-ops/s at 24 ops/sample)\n", ms, samples/ms/1e6, samples*24/ms/1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(valu_packed, dim3(blocks), dim3(256), 0, 0, out, iters, 0.001f); }, 5);
    printf("valu_packed : %8.3f ms  %7.2f G lane-samples/s  (%6.2f T pk-instr/s at 12 pk/sample)\n", ms, samples/ms/1e6, samples*12/ms/1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(valu_fma, dim3(blocks), dim3(256), 0, 0, out, iters, 0.001f); }, 5);
    printf("valu_fma    : %8.3f ms  %7.2f G lane-samples/s  (%6.2f T lane-ops/s at 16 ops/sample)\n", ms, samples/ms/1e6, samples*16/ms/1e9);
    // one wave per SIMD only (what a 65 536-channel bank gives): 256 blocks of 4 waves
    ms = time_ms([&] { hipLaunchKernelGGL(valu_packed, dim3(256), dim3(256), 0, 0, out, iters, 0.001f); }, 5);
    printf("valu_packed 1 wave/SIMD : %8.3f ms  %7.2f G lane-samples/s\n", ms, (double) 256*256*iters*4/ms/1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(valu_packed, dim3(512), dim3(256), 0, 0, out, iters, 0.001f); }, 5);
    printf("valu_packed 2 waves/SIMD: %8.3f ms  %7.2f G lane-samples/s\n", ms, (double) 512*256*iters*4/ms/1e6);
    CK(hipFree(out));
}

static void probe_stream()
{
    const size_t bytes = (size_t) 2 << 30;
    int4 *in;
    int4 *out;
    CK(hipMalloc(&in, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(in, 1, bytes));
    float ms = time_ms([&] { hipLaunchKernelGGL(stream_read, dim3(256*8), dim3(256), 0, 0, in, out, bytes/16); }, 5);
    printf("stream_read 2 GiB: %8.3f ms  %7.1f GB/s\n", ms, bytes/ms/1e6);
    // small (L2/MALL-resident) 21 MB read, like one 65 536 x 160 frame
    const size_t small = (size_t) 65536*320;
    ms = time_ms([&] { hipLaunchKernelGGL(stream_read, dim3(256*4), dim3(256), 0, 0, in, out, small/16); }, 20);
    printf("stream_read 21 MB (cache-resident): %8.3f us  %7.1f GB/s\n", ms*1e3, small/ms/1e6);
    CK(hipFree(in));
    CK(hipFree(out));
}

template <class Det, int LPC = 1, int ABL = 0>
static void probe_tone(const char *name, int n_ch, int samples, int n_frames, int block_len, bool divergent)
{
    ToneLaunch L;
    memset(&L, 0, sizeof(L));
    const size_t frame_elems = (size_t) n_ch*samples;
    int16_t *amp;
    CK(hipMalloc(&amp, frame_elems*n_frames*sizeof(int16_t)));
    std::vector<int16_t> h(frame_elems);
    unsigned s = 12345;
    for (size_t i = 0;  i < frame_elems;  i++)
    {
        s = s*1664525u + 1013904223u;
        h[i] = (int16_t) ((int) (s >> 16) % 8000 - 4000);
    }
    for (int f = 0;  f < n_frames;  f++)
        CK(hipMemcpy(amp + f*frame_elems, h.data(), frame_elems*sizeof(int16_t), hipMemcpyHostToDevice));
    const int maxb = (samples + block_len - 1)/block_len;
    CK(hipMalloc(&L.sf, (size_t) Det::NSF*n_ch*sizeof(float)));
    CK(hipMalloc(&L.si, (size_t) 2*n_ch*sizeof(int32_t)));
    CK(hipMalloc(&L.rec, (size_t) maxb*n_ch*sizeof(uint32_t)));
    CK(hipMemset(L.sf, 0, (size_t) Det::NSF*n_ch*sizeof(float)));
    CK(hipMemset(L.si, 0, (size_t) 2*n_ch*sizeof(int32_t)));
    if (divergent)
    {
        std::vector<int32_t> si(2*n_ch, 0);
        for (int c = 0;  c < n_ch;  c++)
            si[c] = (c*37) % block_len;
        CK(hipMemcpy(L.si, si.data(), si.size()*sizeof(int32_t), hipMemcpyHostToDevice));
    }
    L.stride = samples;
    L.samples = samples;
    L.n_ch = n_ch;
    L.layout = 0;
    L.aligned16 = 1;
    L.maxb = maxb;
    L.nbins = Det::NB;
    L.block_len = block_len;
    for (int i = 0;  i < kMaxBins;  i++)
        L.fac[i] = 1.0f + 0.05f*i;
    L.threshold = 171029200.0f;
    L.normal_twist = 6.309f;
    L.reverse_twist = 2.512f;
    const int waves = (n_ch + kWave/LPC - 1)/(kWave/LPC);
    const int blocks = (waves + kWavesPerBlock - 1)/kWavesPerBlock;
    long long *d_ts = nullptr;
    if (ABL & 32)
    {
        CK(hipMalloc(&d_ts, (size_t) blocks*kWavesPerBlock*16*sizeof(long long)));
        CK(hipMemset(d_ts, 0, (size_t) blocks*kWavesPerBlock*16*sizeof(long long)));
        L.probe_ts = d_ts;
    }
    int f = 0;
    float ms = time_ms([&] {
        L.amp = amp + (size_t) (f % n_frames)*frame_elems;
        f++;
        hipLaunchKernelGGL((tone_bank_kernel<Det, LPC, ABL>), dim3(blocks), dim3(kWave*kWavesPerBlock), 0, 0, L);
    }, 50);
    const double smp = (double) n_ch*samples;
    const double rd = (double) n_ch*(samples*2 + 80);
    printf("%-10s abl=%2d lpc=%d ch=%8d samples=%5d %s: %9.2f us/launch  %8.1f Gsamples/s  alg-read %7.1f GB/s (%4.1f%% of 8 TB/s)\n",
           name, ABL, LPC, n_ch, samples, divergent  ?  "divergent"  :  "uniform  ", ms*1e3, smp/ms/1e6, rd/ms/1e6, rd/ms/1e6/80.0);
    if (ABL & 32)
    {
        std::vector<long long> t((size_t) blocks*kWavesPerBlock*16);
        CK(hipMemcpy(t.data(), d_ts, t.size()*sizeof(long long), hipMemcpyDeviceToHost));
        long long tmin = t[0], tmax = 0;
        for (int w = 0;  w < waves;  w++)
        {
            if (t[(size_t) w*16] < tmin) tmin = t[(size_t) w*16];
            if (t[(size_t) w*16 + 15] > tmax) tmax = t[(size_t) w*16 + 15];
        }
        printf("   kernel span %lld ticks (ticks/us = %.1f from event time); mean over waves of (stamp - kernel start), in ticks:\n     ", tmax - tmin, (tmax - tmin)/(ms*1e3));
        const int nst = 3 + (samples + kSeg - 1)/kSeg;
        for (int k = 0;  k < 16;  k++)
        {
            if (k >= nst  &&  k != 15)
                continue;
            double acc = 0;
            for (int w = 0;  w < waves;  w++)
                acc += (double) (t[(size_t) w*16 + k] - tmin);
            printf("s%d=%.0f ", k, acc/waves);
        }
        printf("\n");
        CK(hipFree(d_ts));
    }
    CK(hipFree(amp));
    CK(hipFree(L.sf));
    CK(hipFree(L.si));
    CK(hipFree(L.rec));
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  clock=%d MHz  memclk=%d MHz\n", p.name, p.multiProcessorCount, p.clockRate/1000, p.memoryClockRate/1000);
    const bool tone_only = (argc > 1  &&  strcmp(argv[1], "tone") == 0);
    if (argc > 1  &&  strcmp(argv[1], "abl") == 0)
    {
        probe_tone<DtmfDet<false>, 1, 0>("dtmf", 1048576, 160, 8, 102, false);
        probe_tone<DtmfDet<false>, 1, 1>("dtmf", 1048576, 160, 8, 102, false);
        probe_tone<DtmfDet<false>, 1, 2>("dtmf", 1048576, 160, 8, 102, false);
        probe_tone<DtmfDet<false>, 1, 4>("dtmf", 1048576, 160, 8, 102, false);
        probe_tone<DtmfDet<false>, 1, 8>("dtmf", 1048576, 160, 8, 102, false);
        probe_tone<DtmfDet<false>, 1, 16>("dtmf", 1048576, 160, 8, 102, false);
        probe_tone<DtmfDet<false>, 1, 7>("dtmf", 1048576, 160, 8, 102, false);
        probe_tone<DtmfDet<false>, 1, 23>("dtmf", 1048576, 160, 8, 102, false);
        probe_tone<DtmfDet<false>, 1, 31>("dtmf", 1048576, 160, 8, 102, false);
        probe_tone<DtmfDet<false>, 2, 0>("dtmf", 65536, 160, 64, 102, false);
        probe_tone<DtmfDet<false>, 2, 4>("dtmf", 65536, 160, 64, 102, false);
        probe_tone<DtmfDet<false>, 2, 7>("dtmf", 65536, 160, 64, 102, false);
        probe_tone<DtmfDet<false>, 2, 23>("dtmf", 65536, 160, 64, 102, false);
        probe_tone<DtmfDet<false>, 2, 31>("dtmf", 65536, 160, 64, 102, false);
        return 0;
    }
    if (argc > 1  &&  strcmp(argv[1], "ts") == 0)
    {
        probe_tone<DtmfDet<false>, 1, 32>("dtmf", 65536, 160, 64, 102, false);
        probe_tone<DtmfDet<false>, 2, 32>("dtmf", 65536, 160, 64, 102, false);
        probe_tone<DtmfDet<false>, 1, 32>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 1, 32>("dtmf", 1048576, 160, 8, 102, false);
        return 0;
    }
    if (argc > 1  &&  strcmp(argv[1], "abl2") == 0)
    {
        probe_tone<DtmfDet<false>, 1, 0>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 1, 1>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 1, 2>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 1, 4>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 1, 8>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 1, 16>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 1, 7>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 1, 23>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 2, 0>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 2, 1>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 2, 2>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 2, 4>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 2, 8>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 2, 16>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 2, 7>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 2, 23>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 1, 0>("dtmf", 65536, 816, 16, 102, false);
        probe_tone<DtmfDet<false>, 1, 0>("dtmf", 65536, 800, 16, 80, false);
        return 0;
    }
    if (tone_only)
    {
        probe_tone<DtmfDet<false>, 2>("dtmf", 65536, 160, 64, 102, false);
        probe_tone<DtmfDet<false>, 2>("dtmf", 65536, 800, 16, 102, false);
        probe_tone<DtmfDet<false>, 1>("dtmf", 65536, 800, 16, 102, false);
        return 0;
    }
    probe_valu();
    probe_stream();
    probe_tone<DtmfDet<false>>("dtmf", 65536, 160, 64, 102, false);
    probe_tone<DtmfDet<false>, 2>("dtmf", 65536, 160, 64, 102, false);
    probe_tone<DtmfDet<false>, 2>("dtmf", 131072, 160, 32, 102, false);
    probe_tone<DtmfDet<false>, 2>("dtmf", 262144, 160, 16, 102, false);
    probe_tone<DtmfDet<false>, 2>("dtmf", 1048576, 160, 8, 102, false);
    probe_tone<DtmfDet<false>, 2>("dtmf", 65536, 800, 16, 102, false);
    probe_tone<DtmfDet<false>, 2>("dtmf", 65536, 160, 64, 102, true);
    probe_tone<BellMfDet, 2>("bell", 131072, 160, 32, 120, false);
    probe_tone<DtmfDet<false>>("dtmf", 131072, 160, 32, 102, false);
    probe_tone<DtmfDet<false>>("dtmf", 262144, 160, 16, 102, false);
    probe_tone<DtmfDet<false>>("dtmf", 1048576, 160, 8, 102, false);
    probe_tone<DtmfDet<false>>("dtmf", 65536, 800, 16, 102, false);
    probe_tone<DtmfDet<false>>("dtmf", 65536, 160, 64, 102, true);
    probe_tone<DtmfDet<true>>("dtmf+filt", 65536, 160, 64, 102, false);
    probe_tone<BellMfDet>("bell", 131072, 160, 32, 120, false);
    probe_tone<R2MfDet>("r2", 131072, 160, 32, 133, false);
    probe_tone<MultiDet<8, true>>("supertone8", 131072, 160, 32, 128, false);
    return 0;
}
