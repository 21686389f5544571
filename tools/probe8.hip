// tools/probe8.hip -- round 5: does a tone bank cut into K sub-banks, each free-running on a stream (= hardware queue) of
// its own, hide the fixed part of a launch (launch boundary, the start burst, the write-back: DESIGN 4.1) under the other
// sub-banks' steady state?  The product kernel (tone_fast.hpp), unchanged; K = 1, 2, 4, 8; launches issued round-robin
// by one host thread, by one host thread per stream, and replayed from one hipGraph per stream.  The figure is wall time
// per tick of the whole bank (every sub-bank advanced by one 160-sample frame), device idle at both ends.
// A digest over the channels' state and records, in channel order, says that the cut changes nothing.
// Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 tools/probe8.hip -o tools/probe8 -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <chrono>
#include <thread>
#include <atomic>

#include "../spandsp_amd/csrc/tone_fast.hpp"

using namespace spg;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef DtmfDet<false> D;

struct Sub
{
    ToneLaunch L;
    hipStream_t st;
    int ch0;
    int f;
};

struct Rig
{
    int n_ch, samples, n_frames;
    int16_t *amp;               // [n_frames][n_ch][samples]
    std::vector<Sub> sub;
};

static inline uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

static int16_t *make_frames(int n_ch, int samples, int n_frames)
{
    int16_t *d;
    const size_t fe = (size_t) n_ch*samples;
    CK(hipMalloc(&d, fe*n_frames*sizeof(int16_t)));
    std::vector<int16_t> h(fe);
    for (int f = 0;  f < n_frames;  f++)
    {
        for (size_t i = 0;  i < fe;  i++)
            h[i] = (int16_t) ((int) (mix((uint32_t) i*2654435761u + f*40503u) >> 16) % 8000 - 4000);
        CK(hipMemcpy(d + f*fe, h.data(), fe*sizeof(int16_t), hipMemcpyHostToDevice));
    }
    return d;
}

static Rig make_rig(int16_t *amp, int n_ch, int samples, int n_frames, int K, bool same_stream)
{
    Rig r;
    r.n_ch = n_ch; r.samples = samples; r.n_frames = n_frames; r.amp = amp;
    const int maxb = (samples + 101)/102;
    hipStream_t shared = 0;
    if (same_stream)
        CK(hipStreamCreateWithFlags(&shared, hipStreamNonBlocking));
    for (int k = 0;  k < K;  k++)
    {
        Sub s;
        memset(&s.L, 0, sizeof(s.L));
        // sub-banks are whole waves; cut at multiples of 256 channels (a workgroup)
        const int wg = n_ch/256;
        const int lo = (int) ((long long) wg*k/K)*256;
        const int hi = (int) ((long long) wg*(k + 1)/K)*256;
        const int n = hi - lo;
        s.ch0 = lo;
        s.f = 0;
        ToneLaunch &L = s.L;
        CK(hipMalloc(&L.sf, (size_t) D::NSF*n*sizeof(float)));
        CK(hipMalloc(&L.si, (size_t) 2*n*sizeof(int32_t)));
        CK(hipMalloc(&L.rec, (size_t) maxb*n*sizeof(uint32_t)));
        L.stride = samples;
        L.samples = samples;
        L.n_ch = n;
        L.layout = 0;
        L.aligned16 = 1;
        L.maxb = maxb;
        L.nbins = D::NB;
        L.block_len = 102;
        for (int i = 0;  i < kMaxBins;  i++)
            L.fac[i] = 1.0f + 0.05f*i;
        L.threshold = 171029200.0f;
        L.normal_twist = 6.309f;
        L.reverse_twist = 2.512f;
        if (same_stream)
            s.st = shared;
        else
            CK(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
        r.sub.push_back(s);
    }
    return r;
}

static void reset_rig(Rig &r)
{
    for (auto &s : r.sub)
    {
        CK(hipMemset(s.L.sf, 0, (size_t) D::NSF*s.L.n_ch*sizeof(float)));
        CK(hipMemset(s.L.si, 0, (size_t) 2*s.L.n_ch*sizeof(int32_t)));
        CK(hipMemset(s.L.rec, 0, (size_t) s.L.maxb*s.L.n_ch*sizeof(uint32_t)));
        s.f = 0;
    }
    CK(hipDeviceSynchronize());
}

static void free_rig(Rig &r, bool same_stream)
{
    for (size_t k = 0;  k < r.sub.size();  k++)
    {
        auto &s = r.sub[k];
        CK(hipFree(s.L.sf));
        CK(hipFree(s.L.si));
        CK(hipFree(s.L.rec));
        if (!same_stream  ||  k == 0)
            CK(hipStreamDestroy(s.st));
    }
}

// state and records of every channel, in channel order whatever the cut
static unsigned long long digest(const Rig &r)
{
    unsigned long long h = 1469598103934665603ull;
    for (auto &s : r.sub)
    {
        const int n = s.L.n_ch;
        const int W = D::NSF + 2 + s.L.maxb;
        std::vector<uint32_t> a((size_t) W*n);
        CK(hipMemcpy(a.data(), s.L.sf, (size_t) D::NSF*n*4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(a.data() + (size_t) D::NSF*n, s.L.si, (size_t) 2*n*4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(a.data() + (size_t) (D::NSF + 2)*n, s.L.rec, (size_t) s.L.maxb*n*4, hipMemcpyDeviceToHost));
        for (int c = 0;  c < n;  c++)
        {
            for (int w = 0;  w < W;  w++)
            {
                h ^= a[(size_t) w*n + c];
                h *= 1099511628211ull;
            }
        }
    }
    return h;
}

template <bool LDR, int WPB>
static inline void launch_one(const Rig &r, Sub &s)
{
    const size_t fe = (size_t) r.n_ch*r.samples;
    s.L.amp = r.amp + (size_t) (s.f % r.n_frames)*fe + (size_t) s.ch0*r.samples;
    s.f++;
    const int waves = (s.L.n_ch + kWave - 1)/kWave;
    const int blocks = (waves + WPB - 1)/WPB;
    launch_tone_fast<D, 1, 2, false, false, WPB, 0, LDR>(s.L, blocks, s.st);
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

enum { EAGER = 0, THREADS = 1, GRAPH = 2 };
static int g_only_mode = -1;

template <bool LDR, int WPB>
static double run(Rig &r, int mode, int ticks)
{
    const int K = (int) r.sub.size();
    if (mode == EAGER)
    {
        CK(hipDeviceSynchronize());
        const double t0 = now_us();
        for (int t = 0;  t < ticks;  t++)
        {
            for (int k = 0;  k < K;  k++)
                launch_one<LDR, WPB>(r, r.sub[k]);
        }
        CK(hipDeviceSynchronize());
        return (now_us() - t0)/ticks;
    }
    if (mode == THREADS)
    {
        CK(hipDeviceSynchronize());
        std::atomic<int> go(0);
        std::vector<std::thread> th;
        for (int k = 0;  k < K;  k++)
        {
            th.emplace_back([&, k] {
                CK(hipSetDevice(0));
                while (!go.load()) { }
                for (int t = 0;  t < ticks;  t++)
                    launch_one<LDR, WPB>(r, r.sub[k]);
                CK(hipStreamSynchronize(r.sub[k].st));
            });
        }
        const double t0 = now_us();
        go.store(1);
        for (auto &t : th)
            t.join();
        return (now_us() - t0)/ticks;
    }
    // one graph of `chunk` launches per stream, replayed
    const int chunk = 100;
    const int reps = std::max(1, ticks/chunk);
    std::vector<hipGraphExec_t> ge(K);
    for (int k = 0;  k < K;  k++)
    {
        hipGraph_t g;
        CK(hipStreamBeginCapture(r.sub[k].st, hipStreamCaptureModeThreadLocal));
        for (int t = 0;  t < chunk;  t++)
            launch_one<LDR, WPB>(r, r.sub[k]);
        CK(hipStreamEndCapture(r.sub[k].st, &g));
        CK(hipGraphInstantiate(&ge[k], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    for (int k = 0;  k < K;  k++)
        CK(hipGraphLaunch(ge[k], r.sub[k].st));
    CK(hipDeviceSynchronize());
    const double t0 = now_us();
    for (int i = 0;  i < reps;  i++)
    {
        for (int k = 0;  k < K;  k++)
            CK(hipGraphLaunch(ge[k], r.sub[k].st));
    }
    CK(hipDeviceSynchronize());
    const double dt = (now_us() - t0)/(reps*chunk);
    for (int k = 0;  k < K;  k++)
        CK(hipGraphExecDestroy(ge[k]));
    return dt;
}

template <bool LDR, int WPB>
static void series(int16_t *amp, int n_ch, int n_frames, int K, bool same_stream, int ticks, int rounds, unsigned long long *want)
{
    Rig r = make_rig(amp, n_ch, 160, n_frames, K, same_stream);
    reset_rig(r);
    for (int t = 0;  t < 5;  t++)
    {
        for (auto &s : r.sub)
            launch_one<LDR, WPB>(r, s);
    }
    CK(hipDeviceSynchronize());
    const unsigned long long dg = digest(r);
    if (*want == 0)
        *want = dg;
    const char *mn[3] = {"eager, one thread", "eager, thread per stream", "graph per stream"};
    for (int mode = 0;  mode < 3;  mode++)
    {
        if (same_stream  &&  mode != EAGER)
            continue;
        if (K == 1  &&  mode == THREADS)
            continue;
        if (g_only_mode >= 0  &&  mode != g_only_mode)
            continue;
        std::vector<double> t;
        run<LDR, WPB>(r, mode, ticks/4);
        for (int i = 0;  i < rounds;  i++)
            t.push_back(run<LDR, WPB>(r, mode, ticks));
        std::sort(t.begin(), t.end());
        const double us = t[t.size()/2];
        printf("ch=%7d K=%d x %6d %-12s wpb%d %-11s %-26s: median %7.2f us/tick  min %7.2f  => %5.3f of 8 TB/s  %s\n",
               n_ch, K, r.sub[0].L.n_ch, same_stream  ?  "(one stream)"  :  "(K streams)", WPB, LDR  ?  "loader"  :  "self-fetch", mn[mode],
               us, t[0], (double) n_ch*400.0/(us*1e-6)/8e12, (dg == *want)  ?  "same digest"  :  "!!! digest differs");
        fflush(stdout);
    }
    free_rig(r, same_stream);
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  GPU_MAX_HW_QUEUES=%s\n", p.name, p.multiProcessorCount, getenv("GPU_MAX_HW_QUEUES")  ?  getenv("GPU_MAX_HW_QUEUES")  :  "(default)");
    const int n_ch = (argc > 1)  ?  atoi(argv[1])  :  65536;
    const int ticks = (argc > 2)  ?  atoi(argv[2])  :  2000;
    const int rounds = (argc > 3)  ?  atoi(argv[3])  :  5;
    const bool trace = (argc > 4  &&  strcmp(argv[4], "trace") == 0);
    const int n_frames = std::max(2, std::min(64, (int) (1342177280ll/((long long) n_ch*320))));
    int16_t *amp = make_frames(n_ch, 160, n_frames);
    unsigned long long want = 0;
    if (trace)
    {
        // a short run for rocprofv3 --kernel-trace: K = 1, 2, 4 eager
        g_only_mode = EAGER;
        series<true, 4>(amp, n_ch, n_frames, 1, false, 200, 1, &want);
        series<true, 4>(amp, n_ch, n_frames, 2, false, 200, 1, &want);
        series<true, 4>(amp, n_ch, n_frames, 4, false, 200, 1, &want);
        return 0;
    }
    series<true, 4>(amp, n_ch, n_frames, 1, false, ticks, rounds, &want);
    series<false, 4>(amp, n_ch, n_frames, 1, false, ticks, rounds, &want);
    for (int K = 2;  K <= 8;  K *= 2)
    {
        series<true, 4>(amp, n_ch, n_frames, K, true, ticks, rounds, &want);        // the cut alone: K launches on one stream
        series<true, 4>(amp, n_ch, n_frames, K, false, ticks, rounds, &want);
        series<false, 4>(amp, n_ch, n_frames, K, false, ticks, rounds, &want);
    }
    // one launch with smaller workgroups
    series<true, 2>(amp, n_ch, n_frames, 1, false, ticks, rounds, &want);
    series<false, 2>(amp, n_ch, n_frames, 1, false, ticks, rounds, &want);
    series<false, 1>(amp, n_ch, n_frames, 1, false, ticks, rounds, &want);
    // half-size workgroups spread a sub-bank over twice the CUs
    series<true, 2>(amp, n_ch, n_frames, 2, false, ticks, rounds, &want);
    series<true, 2>(amp, n_ch, n_frames, 4, false, ticks, rounds, &want);
    series<false, 2>(amp, n_ch, n_frames, 4, false, ticks, rounds, &want);
    series<false, 1>(amp, n_ch, n_frames, 4, false, ticks, rounds, &want);
    CK(hipFree(amp));
    return 0;
}
