// tools/probe7.hip -- feasibility probe for resident tone banks: a kernel that stays on the chip and is handed ticks through
// a doorbell word written by the stream (hipStreamWriteValue32) and answers through a counter the stream waits on
// (hipStreamWaitValue32).  Measures the round trip of an empty tick and of a tick that reads a frame.  Every wait in the
// kernel is bounded by the wall clock, so a mistake ends the kernel instead of hanging the GPU.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe7.hip -o tools/probe7
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Ctrl
{
    uint32_t go;            // tick number, written by the stream
    uint32_t pad0[15];
    uint32_t arrived;       // += 1 per workgroup and tick
    uint32_t pad1[15];
    uint32_t quit;
};

constexpr long long kTimeoutTicks = 100000000ll/5;     // 0.2 s of the 100 MHz wall clock

__global__ __launch_bounds__(320) void resident(Ctrl *c, const int4 *frame, int words_per_wave, int4 *sink)
{
    __shared__ uint32_t s_tick;
    __shared__ int s_quit;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    uint32_t last = 0;
    for (;;)
    {
        if (wv == 4)
        {
            // the polling wave
            const long long t0 = wall_clock64();
            uint32_t g;
            int quit = 0;
            for (;;)
            {
                g = __hip_atomic_load(&c->go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (g != last)
                    break;
                if (__hip_atomic_load(&c->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)  ||  wall_clock64() - t0 > kTimeoutTicks)
                {
                    quit = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0)
            {
                s_tick = g;
                s_quit = quit;
            }
        }
        __syncthreads();
        if (s_quit)
            return;
        last = s_tick;
        if (wv < 4  &&  words_per_wave > 0)
        {
            // stand-in for the work: every consumer wave reads its share of a frame
            const int4 *p = frame + ((size_t) (blockIdx.x*4 + wv)*words_per_wave)*64 + lane;
            int4 acc = {0, 0, 0, 0};
            for (int i = 0;  i < words_per_wave;  i++)
            {
                const int4 v = p[(size_t) i*64];
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
            if (acc.x == 0x12345678)
                sink[threadIdx.x] = acc;
        }
        __syncthreads();
        if (threadIdx.x == 0)
        {
            __threadfence();
            atomicAdd(&c->arrived, 1u);
        }
    }
}

int main(int argc, char **argv)
{
    const int steps = (argc > 1)  ?  atoi(argv[1])  :  2000;
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Ctrl *c = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **) &c, sizeof(Ctrl), hipMallocSignalMemory);
    printf("signal memory: %s\n", hipGetErrorString(e));
    if (e != hipSuccess)
        CK(hipMalloc((void **) &c, sizeof(Ctrl)));
    const int blocks = 256;
    for (int mode = 0;  mode < 2;  mode++)
    {
        const int words = mode  ?  20  :  0;            // 20 x 1 KiB per wave = the 320-byte rows of 64 channels
        int4 *frame = nullptr;
        int4 *sink = nullptr;
        CK(hipMalloc((void **) &frame, (size_t) blocks*4*20*64*sizeof(int4)));
        CK(hipMalloc((void **) &sink, 4096*sizeof(int4)));
        CK(hipMemset(frame, 1, (size_t) blocks*4*20*64*sizeof(int4)));
        CK(hipMemset(c, 0, sizeof(Ctrl)));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(resident, dim3(blocks), dim3(320), 0, st, c, frame, words, sink);
        // the ticks go through a second stream: the first one is busy with the resident kernel
        hipStream_t st2;
        CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        uint32_t t = 0;
        for (int w = 0;  w < 50;  w++)
        {
            t++;
            CK(hipStreamWriteValue32(st2, &c->go, t, 0));
            CK(hipStreamWaitValue32(st2, &c->arrived, (uint32_t) blocks*t, hipStreamWaitValueGte, 0xFFFFFFFFu));
        }
        CK(hipStreamSynchronize(st2));
        CK(hipEventRecord(a, st2));
        for (int i = 0;  i < steps;  i++)
        {
            t++;
            CK(hipStreamWriteValue32(st2, &c->go, t, 0));
            CK(hipStreamWaitValue32(st2, &c->arrived, (uint32_t) blocks*t, hipStreamWaitValueGte, 0xFFFFFFFFu));
        }
        CK(hipEventRecord(b, st2));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("mode %d (%s): %d ticks, %.3f us per tick\n", mode, mode  ?  "each tick reads a 21 MB frame"  :  "empty tick", steps, ms*1e3/steps);
        CK(hipStreamWriteValue32(st2, &c->quit, 1, 0));
        CK(hipStreamSynchronize(st2));
        CK(hipStreamSynchronize(st));
        printf("   resident kernel ended\n");
        CK(hipFree(frame));
        CK(hipFree(sink));
        CK(hipStreamDestroy(st2));
    }
    return 0;
}
