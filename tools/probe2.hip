// tools/probe2.hip -- round-2 on-GPU probes for the streaming Goertzel bank kernel (tone_fast.hpp): launch time over
// lane mappings, ring depths, workgroup sizes and cache policy, against the round-1 kernel on the same frames, with an
// A-B check that both leave identical state and records; knock-outs (no recurrence / no DMA) and per-wave timestamps.
// Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe2.hip -o tools/probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#include "../spandsp_amd/csrc/tone_fast.hpp"

using namespace spg;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Rig
{
    ToneLaunch L;
    int16_t *amp;
    size_t frame_elems;
    int n_frames;
    int nsf;
    long long *d_ts;
};

template <class Det>
static Rig make_rig(int n_ch, int samples, int n_frames, int block_len, bool divergent)
{
    Rig r;
    memset(&r, 0, sizeof(r));
    ToneLaunch &L = r.L;
    r.frame_elems = (size_t) n_ch*samples;
    r.n_frames = n_frames;
    r.nsf = Det::NSF;
    CK(hipMalloc(&r.amp, r.frame_elems*n_frames*sizeof(int16_t)));
    std::vector<int16_t> h(r.frame_elems);
    unsigned s = 12345;
    for (int f = 0;  f < n_frames;  f++)
    {
        for (size_t i = 0;  i < r.frame_elems;  i++)
        {
            s = s*1664525u + 1013904223u;
            h[i] = (int16_t) ((int) (s >> 16) % 8000 - 4000);
        }
        CK(hipMemcpy(r.amp + f*r.frame_elems, h.data(), r.frame_elems*sizeof(int16_t), hipMemcpyHostToDevice));
    }
    const int maxb = (samples + block_len - 1)/block_len;
    CK(hipMalloc(&L.sf, (size_t) Det::NSF*n_ch*sizeof(float)));
    CK(hipMalloc(&L.si, (size_t) 2*n_ch*sizeof(int32_t)));
    CK(hipMalloc(&L.rec, (size_t) maxb*n_ch*sizeof(uint32_t)));
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
    CK(hipMalloc(&r.d_ts, (size_t) (n_ch/8 + 64)*16*sizeof(long long)));
    L.probe_ts = r.d_ts;
    (void) divergent;
    return r;
}

static void reset_rig(Rig &r, bool divergent, int block_len)
{
    CK(hipMemset(r.L.sf, 0, (size_t) r.nsf*r.L.n_ch*sizeof(float)));
    std::vector<int32_t> si(2*(size_t) r.L.n_ch, 0);
    if (divergent)
    {
        for (int c = 0;  c < r.L.n_ch;  c++)
            si[c] = (c*37) % block_len;
    }
    CK(hipMemcpy(r.L.si, si.data(), si.size()*sizeof(int32_t), hipMemcpyHostToDevice));
    CK(hipMemset(r.d_ts, 0, (size_t) (r.L.n_ch/8 + 64)*16*sizeof(long long)));
    CK(hipDeviceSynchronize());
}

static void free_rig(Rig &r)
{
    CK(hipFree(r.amp));
    CK(hipFree(r.L.sf));
    CK(hipFree(r.L.si));
    CK(hipFree(r.L.rec));
    CK(hipFree(r.d_ts));
}

static bool g_graph = false;
static hipStream_t g_stream = 0;

template <class F>
static float time_ms(F launch, int reps)
{
    if (g_graph)
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        launch();
        CK(hipStreamSynchronize(g_stream));
        CK(hipStreamBeginCapture(g_stream, hipStreamCaptureModeGlobal));
        for (int i = 0;  i < reps;  i++)
            launch();
        CK(hipStreamEndCapture(g_stream, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, g_stream));
        CK(hipStreamSynchronize(g_stream));
        CK(hipEventRecord(a, g_stream));
        CK(hipGraphLaunch(ge, g_stream));
        CK(hipEventRecord(b, g_stream));
        CK(hipEventSynchronize(b));
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, a, b));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        return ms/reps;
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, g_stream));
    for (int i = 0;  i < reps;  i++)
        launch();
    CK(hipEventRecord(b, g_stream));
    CK(hipEventSynchronize(b));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms/reps;
}

static unsigned long long digest(const Rig &r)
{
    const size_t nf = (size_t) r.nsf*r.L.n_ch;
    std::vector<uint32_t> a(nf + 2*(size_t) r.L.n_ch + (size_t) r.L.maxb*r.L.n_ch);
    CK(hipMemcpy(a.data(), r.L.sf, nf*4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a.data() + nf, r.L.si, 2*(size_t) r.L.n_ch*4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a.data() + nf + 2*(size_t) r.L.n_ch, r.L.rec, (size_t) r.L.maxb*r.L.n_ch*4, hipMemcpyDeviceToHost));
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0;  i < a.size();  i++)
    {
        h ^= a[i];
        h *= 1099511628211ull;
    }
    return h;
}

static void report(const char *name, const Rig &r, float ms, int waves, bool ts, unsigned long long dg)
{
    const double smp = (double) r.L.n_ch*r.L.samples;
    const double rd = (double) r.L.n_ch*(r.L.samples*2 + 80);
    printf("%-34s ch=%8d n=%4d : %8.2f us  %7.1f Gsmp/s  alg-read %6.1f GB/s (%4.1f%%)  digest %016llx\n",
           name, r.L.n_ch, r.L.samples, ms*1e3, smp/ms/1e6, rd/ms/1e6, rd/ms/1e6/80.0, dg);
    if (ts)
    {
        std::vector<long long> t((size_t) waves*16);
        CK(hipMemcpy(t.data(), r.d_ts, t.size()*sizeof(long long), hipMemcpyDeviceToHost));
        long long tmin = 0x7fffffffffffffffll, tmax = 0;
        for (int w = 0;  w < waves;  w++)
        {
            if (t[(size_t) w*16]) tmin = std::min(tmin, t[(size_t) w*16]);
            tmax = std::max(tmax, t[(size_t) w*16 + 15]);
        }
        printf("     last launch: first wave start -> last wave end %lld ticks = %.2f of the event time per launch if 2.4 ticks/ns; per stamp min/mean/max over waves, ticks after the first start:\n", tmax - tmin, (tmax - tmin)/2400.0/(ms*1e3));
        for (int k = 0;  k < 16;  k++)
        {
            double acc = 0;
            long long mx = 0, mn = 0x7fffffffffffffffll;
            int cnt = 0;
            for (int w = 0;  w < waves;  w++)
            {
                const long long v = t[(size_t) w*16 + k];
                if (v == 0)
                    continue;
                const long long d = v - t[(size_t) w*16];        // ticks since this wave's own start (counters differ between XCDs)
                acc += (double) d;
                mx = std::max(mx, d);
                mn = std::min(mn, d);
                cnt++;
            }
            if (cnt)
                printf("       s%-2d %7lld %9.0f %7lld\n", k, mn, acc/cnt, mx);
        }
        // loader waves (if any): stamps at slot waves + 8 + wg
        std::vector<long long> tl((size_t) 1024*16);
        const int nwg = std::min(1024, (waves + 3)/4);
        CK(hipMemcpy(tl.data(), r.d_ts + ((size_t) waves + 8)*16, (size_t) nwg*16*sizeof(long long), hipMemcpyDeviceToHost));
        if (tl[0])
        {
            printf("     loader waves, ticks after their own start (mean over %d workgroups):", nwg);
            for (int k = 0;  k < 16;  k++)
            {
                double acc = 0;
                int cnt = 0;
                for (int w = 0;  w < nwg;  w++)
                {
                    if (tl[(size_t) w*16 + k] == 0)
                        continue;
                    acc += (double) (tl[(size_t) w*16 + k] - tl[(size_t) w*16]);
                    cnt++;
                }
                if (cnt)
                    printf(" l%d=%.0f", k, acc/cnt);
            }
            printf("\n");
        }
    }
}

// One frame through the kernel from a fresh state (for the A-B digest), then the timing loop over the frame ring.
template <class Det, int LPC, int ABL>
static unsigned long long run_old(const char *name, Rig &r, int block_len, bool divergent, int reps)
{
    const int waves = (r.L.n_ch + kWave/LPC - 1)/(kWave/LPC);
    const int blocks = (waves + kWavesPerBlock - 1)/kWavesPerBlock;
    reset_rig(r, divergent, block_len);
    int f = 0;
    auto go = [&] {
        r.L.amp = r.amp + (size_t) (f % r.n_frames)*r.frame_elems;
        f++;
        hipLaunchKernelGGL((tone_bank_kernel<Det, LPC, ABL>), dim3(blocks), dim3(kWave*kWavesPerBlock), 0, g_stream, r.L);
    };
    go(); go(); go();
    CK(hipDeviceSynchronize());
    const unsigned long long dg = digest(r);
    const float ms = time_ms(go, reps);
    CK(hipDeviceSynchronize());
    report(name, r, ms, waves, (ABL & 32) != 0, dg);
    return dg;
}

template <class Det, int LPC, int R, bool NT, int WPB, int ABL, bool LDR = false>
static unsigned long long run_new(const char *name, Rig &r, int block_len, bool divergent, int reps)
{
    const int waves = (r.L.n_ch + kWave/LPC - 1)/(kWave/LPC);
    const int blocks = (waves + WPB - 1)/WPB;
    reset_rig(r, divergent, block_len);
    int f = 0;
    auto go = [&] {
        r.L.amp = r.amp + (size_t) (f % r.n_frames)*r.frame_elems;
        f++;
        launch_tone_fast<Det, LPC, R, false, NT, WPB, ABL, LDR>(r.L, blocks, g_stream);
    };
    go(); go(); go();
    CK(hipDeviceSynchronize());
    const unsigned long long dg = digest(r);
    const float ms = time_ms(go, reps);
    report(name, r, ms, waves, (ABL & 32) != 0, dg);
    return dg;
}

#define OLD(LPC, ABL) run_old<D, LPC, ABL>("r1 lpc" #LPC " abl" #ABL, r, 102, false, reps)
#define NEWL(LPC, R, NT, ABL) run_new<D, LPC, R, NT, 4, ABL, true>("ring+loader lpc" #LPC " R" #R " nt" #NT " abl" #ABL, r, 102, false, reps)
#define NEW(LPC, R, NT, WPB, ABL) run_new<D, LPC, R, NT, WPB, ABL>("ring lpc" #LPC " R" #R " nt" #NT " wpb" #WPB " abl" #ABL, r, 102, false, reps)

static void sweep(int n_ch, int samples, int n_frames, int reps, bool full)
{
    typedef DtmfDet<false> D;
    Rig r = make_rig<D>(n_ch, samples, n_frames, 102, false);
    printf("---- DTMF, %d channels x %d samples, %d distinct frames ----\n", n_ch, samples, n_frames);
    const unsigned long long want = OLD(1, 0);
    unsigned long long got[16];
    int k = 0;
    got[k++] = NEWL(1, 2, false, 0);
    got[k++] = NEWL(1, 3, false, 0);
    got[k++] = NEW(1, 2, false, 4, 0);
    got[k++] = NEW(1, 2, false, 1, 0);
    got[k++] = NEW(1, 3, false, 4, 0);
    got[k++] = NEW(1, 2, true, 4, 0);
    for (int i = 0;  i < k;  i++)
    {
        if (got[i] != want)
            printf("   !!! variant %d differs from the round-1 kernel\n", i);
    }
    if (full)
    {
        NEWL(1, 3, false, 128);
        NEWL(1, 3, false, 256);
        NEWL(1, 3, false, 272);
        NEWL(1, 3, false, 24);
        NEWL(1, 3, false, 32);
    }
    free_rig(r);
}

// What the block ends cost: the generic 8-bin bank (same arithmetic as DTMF up to the decision) with 102-sample blocks and
// with blocks so long that none ends
static void sweep_blocks(int n_ch)
{
    typedef MultiDet<8, false> D;
    const int reps = (n_ch > 200000)  ?  20  :  200;
    for (int bl = 0;  bl < 2;  bl++)
    {
        const int block_len = bl  ?  30000  :  102;
        Rig r = make_rig<D>(n_ch, 160, (n_ch > 200000)  ?  6  :  64, block_len, false);
        r.L.block_len = block_len;
        printf("---- generic 8-bin bank, %d channels x 160 samples, %d-sample blocks ----\n", n_ch, block_len);
        if (n_ch <= 393216)
            run_new<D, 1, 2, false, 4, 0, true>("ring+loader", r, block_len, false, reps);
        run_new<D, 1, 2, false, 4, 0, false>("ring", r, block_len, false, reps);
        free_rig(r);
    }
}

// Where a big bank's time goes: the self-fetching kernel at 1 M channels with parts switched off, and per-wave stamps
static void sweep_big(int n_ch)
{
    typedef DtmfDet<false> D;
    const int reps = 20;
    Rig r = make_rig<D>(n_ch, 160, 6, 102, false);
    printf("---- DTMF, %d channels x 160 samples: ablations of the self-fetching kernel ----\n", n_ch);
    NEW(1, 2, false, 4, 0);
    NEW(1, 2, false, 4, 8);         // no recurrence
    NEW(1, 2, false, 4, 16);        // no DMA (pieces not fetched)
    NEW(1, 2, false, 4, 24);        // neither
    NEW(1, 2, false, 4, 256);       // state only
    NEW(1, 2, false, 4, 1040);      // no DMA, no state traffic: the compute side alone (recurrence, LDS reads, block ends)
    NEW(1, 2, false, 4, 1024);      // no state traffic: frame stream + compute
    NEW(1, 2, false, 4, 1032);      // no state traffic, no recurrence: the frame stream alone
    NEW(1, 2, false, 4, 32);        // stamps
    NEWL(1, 2, false, 0);
    NEWL(1, 2, false, 32);
    // ring depth and waves per workgroup at this size (the loader variants)
    NEWL(1, 3, false, 0);
    NEWL(1, 4, false, 0);
    NEW(1, 3, false, 4, 0);
    NEW(1, 2, false, 8, 0);
    NEW(1, 3, false, 8, 0);
    free_rig(r);
}

// Round 4: the lean block end (ABL bit 11 = the general decision, i.e. the kernel before), the state write-back as
// non-temporal / absent / write-through stores, at the headline size and at 1 M channels, eager and from a graph.
static void sweep_r4(int n_ch, int n_frames, int reps)
{
    typedef DtmfDet<false> D;
    Rig r = make_rig<D>(n_ch, 160, n_frames, 102, false);
    printf("---- r4: DTMF, %d channels x 160 samples ----\n", n_ch);
    if (n_ch <= 393216)
    {
        const unsigned long long want = NEWL(1, 2, false, 2048);
        const unsigned long long lean = NEWL(1, 2, false, 0);
        if (lean != want)
            printf("   !!! the lean block end differs from the general one\n");
        if (NEWL(1, 2, false, 131072) != want)
            printf("   !!! no-touch variant differs\n");
        NEWL(1, 2, false, 16384);
        NEWL(1, 2, false, 32768);
        if (NEWL(1, 2, false, 65536) != want)
            printf("   !!! write-through variant differs\n");
        if (NEWL(1, 2, false, 262144) != want)
            printf("   !!! sc1 variant differs\n");
        if (NEWL(1, 2, false, 524288) != want)
            printf("   !!! nt write-through variant differs\n");
        NEWL(1, 2, false, 128);
        NEWL(1, 2, false, 256);
        NEWL(1, 2, false, 32);
    }
    else
    {
        const unsigned long long want = NEW(1, 2, false, 4, 2048);
        if (NEW(1, 2, false, 4, 0) != want)
            printf("   !!! the lean block end differs from the general one\n");
        NEW(1, 2, false, 4, 32768);
        if (NEW(1, 2, false, 4, 65536) != want)
            printf("   !!! write-through variant differs\n");
    }
    free_rig(r);
}

// A-B with the variants taken in turn, several rounds, median and minimum per variant (boxes and moments differ by
// more than the effects looked for).
template <class Det, int ABL, bool LDR, int R = 2, int WPB = 4>
static float ab_time(Rig &r, int reps, unsigned long long *dg)
{
    const int waves = (r.L.n_ch + kWave - 1)/kWave;
    const int blocks = (waves + WPB - 1)/WPB;
    int f = 0;
    auto go = [&] {
        r.L.amp = r.amp + (size_t) (f % r.n_frames)*r.frame_elems;
        f++;
        launch_tone_fast<Det, 1, R, false, false, WPB, ABL, LDR>(r.L, blocks, g_stream);
    };
    if (dg)
    {
        reset_rig(r, false, 102);
        go(); go(); go();
        CK(hipDeviceSynchronize());
        *dg = digest(r);
    }
    return time_ms(go, reps);
}

template <bool LDR>
static void ab_r4(int n_ch, int n_frames, int reps, int rounds)
{
    typedef DtmfDet<false> D;
    Rig r = make_rig<D>(n_ch, 160, n_frames, 102, false);
    printf("---- r4 A-B: DTMF, %d channels x 160 samples, %s, %d rounds ----\n", n_ch, LDR  ?  "loader wave"  :  "self-fetching", rounds);
    const char *names[] = {"general block end, rolled loop (round 3)", "lean block end, rolled loop", "product (lean + asm pairs + write-through)", "... nt stores",
                           "... plain stores", "... ring of 3 slots", "... no stores (ablation)", "... no DMA issued (ablation)",
                           "... self-fetching, no DMA: no loader, no barriers (ablation)", "... no recurrence (ablation)"};
    constexpr int NV = 10;
    std::vector<float> t[NV];
    unsigned long long dg[NV];
    for (int k = 0;  k < rounds;  k++)
    {
        t[0].push_back(ab_time<D, 2048 + 131072, LDR>(r, reps, (k == 0)  ?  &dg[0]  :  nullptr));
        t[1].push_back(ab_time<D, 131072, LDR>(r, reps, (k == 0)  ?  &dg[1]  :  nullptr));
        t[2].push_back(ab_time<D, 0, LDR>(r, reps, (k == 0)  ?  &dg[2]  :  nullptr));
        t[3].push_back(ab_time<D, 16384, LDR>(r, reps, (k == 0)  ?  &dg[3]  :  nullptr));
        t[4].push_back(ab_time<D, 65536, LDR>(r, reps, (k == 0)  ?  &dg[4]  :  nullptr));
        t[5].push_back(ab_time<D, 0, LDR, 3>(r, reps, (k == 0)  ?  &dg[5]  :  nullptr));
        t[6].push_back(ab_time<D, 32768, LDR>(r, reps, (k == 0)  ?  &dg[6]  :  nullptr));
        t[7].push_back(ab_time<D, 16, LDR>(r, reps, (k == 0)  ?  &dg[7]  :  nullptr));
        t[8].push_back(ab_time<D, 16, false>(r, reps, (k == 0)  ?  &dg[8]  :  nullptr));
        t[9].push_back(ab_time<D, 8, LDR>(r, reps, (k == 0)  ?  &dg[9]  :  nullptr));
    }
    for (int v = 0;  v < NV;  v++)
    {
        std::sort(t[v].begin(), t[v].end());
        printf("%-44s median %7.2f us  min %7.2f us  %s\n", names[v], t[v][t[v].size()/2]*1e3, t[v][0]*1e3,
               (v >= 6)  ?  ""  :  (dg[v] == dg[0])  ?  "same digest"  :  "!!! digest differs");
    }
    free_rig(r);
}

// Is a block end's cost the code being cold?  Two frames' worth of samples in one launch: three block ends per wave, the
// stamps show what the first costs against the later ones.
static void sweep_r4_long(int n_ch)
{
    typedef DtmfDet<false> D;
    Rig r = make_rig<D>(n_ch, 320, 32, 102, false);
    printf("---- r4: DTMF, %d channels x 320 samples (stamps per 32-sample piece) ----\n", n_ch);
    const int reps = 100;
    NEWL(1, 2, false, 0);
    NEWL(1, 2, false, 32);
    free_rig(r);
}

// Round 5: what do the two record rows' stores cost the end of a launch?  (ABL bit 18: not stored.)  Variants taken in turn.
static void ab_rec(int n_ch, int n_frames, int reps, int rounds)
{
    typedef DtmfDet<false> D;
    Rig r = make_rig<D>(n_ch, 160, n_frames, 102, false);
    printf("---- r5 record stores: DTMF, %d channels x 160 samples, %d rounds ----\n", n_ch, rounds);
    std::vector<float> t[3];
    for (int k = 0;  k < rounds;  k++)
    {
        t[0].push_back(ab_time<D, 0, true>(r, reps, nullptr));
        t[1].push_back(ab_time<D, 262144, true>(r, reps, nullptr));
        t[2].push_back(ab_time<D, 32768, true>(r, reps, nullptr));
    }
    const char *names[3] = {"product", "... the two record rows not stored", "... no stores at all"};
    for (int v = 0;  v < 3;  v++)
    {
        std::sort(t[v].begin(), t[v].end());
        printf("%-40s median %7.2f us  min %7.2f us\n", names[v], t[v][t[v].size()/2]*1e3, t[v][0]*1e3);
    }
    free_rig(r);
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  clock=%d MHz\n", p.name, p.multiProcessorCount, p.clockRate/1000);
    const bool quick = (argc > 1  &&  strcmp(argv[1], "quick") == 0);
    if (argc > 1  &&  strcmp(argv[1], "blocks") == 0)
    {
        CK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
        sweep_blocks(65536);
        sweep_blocks(1048576);
        return 0;
    }
    if (argc > 1  &&  strcmp(argv[1], "r4") == 0)
    {
        CK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
        if (argc > 2  &&  strcmp(argv[2], "graph") == 0)
            g_graph = true;
        if (argc > 2  &&  strcmp(argv[2], "ab") == 0)
        {
            ab_r4<true>(65536, 64, 300, 7);
            ab_r4<true>(131072, 32, 200, 5);
            ab_r4<false>(1048576, 6, 30, 5);
            g_graph = true;
            printf("(from a hipGraph)\n");
            ab_r4<true>(65536, 64, 300, 7);
            return 0;
        }
        if (argc > 2  &&  strcmp(argv[2], "long") == 0)
        {
            sweep_r4_long(65536);
            return 0;
        }
        sweep_r4(65536, 64, 300);
        sweep_r4(131072, 32, 200);
        sweep_r4(1048576, 6, 30);
        return 0;
    }
    if (argc > 1  &&  strcmp(argv[1], "rec") == 0)
    {
        CK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
        ab_rec(65536, 64, 300, 9);
        ab_rec(131072, 32, 200, 7);
        return 0;
    }
    if (argc > 1  &&  strcmp(argv[1], "big") == 0)
    {
        CK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
        sweep_big((argc > 2)  ?  atoi(argv[2])  :  1048576);
        return 0;
    }
    CK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    if (argc > 2  &&  strcmp(argv[2], "graph") == 0)
        g_graph = true;
    sweep(65536, 160, 64, 200, true);
    if (quick)
        return 0;
    sweep(1048576, 160, 6, 20, false);
    sweep(131072, 160, 32, 100, false);
    sweep(262144, 160, 16, 50, false);
    sweep(524288, 160, 8, 30, false);
    return 0;
}
