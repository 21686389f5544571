// spangpu_api.hip -- the C ABI of libspangpu.so (include/spangpu.h): bank lifetime,
// frame submission, record decode.  Device code is in tone_dev.hpp (and the modem /
// echo headers); this file owns HBM allocations, the bank stream and launch geometry.
//
// There is deliberately no CPU implementation behind these entry points: without a
// HIP device every call that needs one returns SPANGPU_ERR_NO_DEVICE.

#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/spangpu.h"
#include "../../include/spangpu_refstate.h"
#include "tone_dev.hpp"
#include "tone_fast.hpp"

using namespace spg;

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do                                                                                  \
    {                                                                                   \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(SPANGPU_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    }                                                                                   \
    while (0)

struct spangpu_bank_s
{
    int device;
    int kind;
    int n_ch;
    hipStream_t stream;
    bool own_stream;
    bool timing;
    // Queue mode (spangpu_bank_set_queues(bank, 2)): the streaming kernel's launch is cut in two workgroup ranges, the second
    // on a stream (= hardware queue) of its own, so that one half's launch boundary, start burst and write-back lie under the
    // other half's steady state (profiles/r5_probe_mq.log).  ev_in orders the second stream behind whatever the bank's stream
    // held when the launch was queued; ev_q, recorded behind the second half, is what joined() makes the bank's stream wait for.
    int queues;
    hipStream_t stream2;
    hipEvent_t ev_in;
    hipEvent_t ev_q;
    bool s2_busy;               // the second stream holds launches the bank's stream has not been made to wait for
    bool main_touched;          // the bank's stream was handed out or used since the last split launch: the next one orders
                                // the second stream behind it.  (Between two split launches with nothing else in between
                                // NO event is recorded or waited for: the two queues run free, which is where the gain is --
                                // a wait per tick made 131 072 channels take 38 us a tick instead of 16.5.)
    spangpu_tone_params_t tp;
    // geometry of the detector
    int nb;                 // bins compiled into the kernel used
    int nsf;                // float state words per channel
    int block_len;
    float fac[kMaxBins];
    float threshold;
    float normal_twist;
    float reverse_twist;
    // device state
    float *sf;
    int32_t *si;
    // per-call outputs (capacity = maxb_cap blocks)
    int maxb_cap;
    uint32_t *rec;
    uint32_t *ext_rec;          // caller-owned record buffer for the next launches (spangpu_bank_set_records_buffer)
    size_t ext_rec_bytes;
    uint8_t *ext_digits;        // caller-owned digit bytes of the next launches (spangpu_bank_set_digits_buffer / _ring):
    size_t ext_digits_bytes;    // ... bytes per slice, slices, and which slice the next launch fills
    int ext_digits_slices;
    int ext_digits_next;
    uint32_t *cur_rec;          // where the last launch wrote its records
    int next_fmt;               // sample format of the launch being prepared (0 linear, 1 A-law, 2 u-law)
    const int32_t *next_lens;   // per-channel lengths of the launch being prepared (device), or nullptr
    bool next_ragged;           // ... some of them are neither 0 nor the longest
    int32_t *d_lens;            // [n_ch], device
    int32_t *h_lens;            // [n_ch], pinned
    float *chan_parms;          // DTMF, once a channel was given parameters of its own: [4][n_ch] on the device
    float *h_chan_parms;        // ... and its host mirror
    int n_filter_on;            // channels whose dial tone filter is on
    float *rec_energy;
    int32_t *rec_dur;
    float *trace;
    // host mirrors (pinned)
    uint32_t *h_rec;
    float *h_energy;
    int32_t *h_dur;
    // staging for host-resident frames
    int16_t *d_amp;
    size_t d_amp_cap;       // in samples
    // last call
    int last_maxb;
    int last_samples;
    unsigned launch_serial;     // counts launches (the cadence matcher takes each launch's records once)
    struct Cadence *cad;        // super-tone: the cadence matcher on the device (spangpu_bank_set_cadences)
    hipEvent_t ev0;
    hipEvent_t ev1;
    bool ev_valid;
};

// The bank's stream, with the second queue's last launch joined into it if one is outstanding: every entry point that puts work
// on the stream, waits for it, or hands it out goes through here (the split launch itself excepted).
static hipStream_t joined(spangpu_bank_s *b)
{
    if (b->s2_busy)
    {
        (void) hipEventRecord(b->ev_q, b->stream2);
        (void) hipStreamWaitEvent(b->stream, b->ev_q, 0);
        b->s2_busy = false;
    }
    b->main_touched = true;
    return b->stream;
}
static hipStream_t joined(const spangpu_bank_s *b) { return joined(const_cast<spangpu_bank_s *>(b)); }

// Lanes per channel: 2 while the bank is too small to put >= 4 one-channel-per-lane waves
// on every SIMD (1024 SIMDs x 64 lanes x 4), else 1.  SPANGPU_LPC=1|2 overrides (tuning).
static std::atomic<int> g_forced_lpc{-1};      // (the tuning knobs are process-wide and may be turned from any thread: relaxed atomics, read once per use)

static int forced_lpc(void)
{
    int v = g_forced_lpc.load(std::memory_order_relaxed);
    if (v < 0)
    {
        const char *e = getenv("SPANGPU_LPC");
        v = (e  &&  (e[0] == '1'  ||  e[0] == '2'))  ?  (e[0] - '0')  :  0;
        g_forced_lpc.store(v, std::memory_order_relaxed);
    }
    return v;
}

// the general kernel's mapping
static int pick_lpc(int n_ch)
{
    const int f = forced_lpc();
    if (f)
        return f;
    return (n_ch < 262144)  ?  2  :  1;
}

// Which kernel family serves a launch: the streaming kernels (tone_fast.hpp) whenever the frame is channel-major with
// 16-byte aligned rows, else the general one (tone_dev.hpp: sample-major / unaligned frames, zero-length calls).  Of the
// streaming kernels, banks up to kLoaderMaxChannels take the one with a loader wave per workgroup (the consumer waves
// of a small bank are alone on their SIMDs: nothing covers a wave that stands in the memory pipeline's queue), larger
// banks the one in which every wave fetches for itself (there the other waves of the SIMD cover it, and a loader wave
// would only take a wave slot).  Crossover measured in profiles/r2_probe.log.  spangpu_tune_tone_kernel() forces one
// family (A-B measurements, parity tests of all of them).
static std::atomic<int> g_tone_variant{0};  // 0 auto, 1 general, 2 streaming with loader waves, 3 streaming without
constexpr int kLoaderMaxChannels = 393216;

static bool fast_eligible(const ToneLaunch &L)
{
    return g_tone_variant != 1  &&  L.layout == 0  &&  L.aligned16  &&  L.samples > 0  &&  !L.lens_ragged
           &&  (unsigned long long) L.stride*2ull*256ull < 0xFFFFFFFFull  &&  L.n_ch <= (1 << 29);
}

static bool use_loader(int n_ch)
{
    if (g_tone_variant == 2)
        return true;
    if (g_tone_variant == 3)
        return false;
    return n_ch <= kLoaderMaxChannels;
}

constexpr int kFastWPB = 4;
constexpr int kRingLoader = 2;          // segment slots per consumer wave, kernels with a loader wave
constexpr int kRingSelf = 2;            // ... kernels whose waves fetch for themselves (three slots measured slower: r2_probe.log)

template <class Det> struct is_super_tone { static constexpr bool value = false; };
template <int N> struct is_super_tone<MultiDet<N, true>> { static constexpr bool value = true; };
static thread_local bool g_cadence_fused = false;   // the launch this thread just made matched the cadences in its epilogue

template <class Det, int LPC, bool G711>
static void launch_fast(const ToneLaunch &L, hipStream_t st, bool loader)
{
    const int waves = (L.n_ch + kWave/LPC - 1)/(kWave/LPC);
    const int blocks = (L.wgn > 0)  ?  L.wgn  :  (waves + kFastWPB - 1)/kFastWPB;
    if constexpr (is_super_tone<Det>::value  &&  LPC == 1  &&  !G711)
    {
        if (L.cad.state)
        {
            if (loader)
                launch_tone_fast<Det, 1, kRingLoader, false, false, kFastWPB, kToneCadence, true>(L, blocks, st);
            else
                launch_tone_fast<Det, 1, kRingSelf, false, false, kFastWPB, kToneCadence, false>(L, blocks, st);
            g_cadence_fused = true;
            return;
        }
    }
    if constexpr (Det::kDigits)
    {
        // a launch that also reports one digit byte per block runs the variant with those stores compiled in
        if (L.digits)
        {
            if (loader  &&  LPC == 1)
                launch_tone_fast<Det, 1, kRingLoader, G711, false, kFastWPB, kToneDigits, true>(L, blocks, st);
            else
                launch_tone_fast<Det, LPC, kRingSelf, G711, false, kFastWPB, kToneDigits, false>(L, blocks, st);
            return;
        }
    }
    if (loader  &&  LPC == 1)
        launch_tone_fast<Det, 1, kRingLoader, G711, false, kFastWPB, 0, true>(L, blocks, st);
    else
        launch_tone_fast<Det, LPC, kRingSelf, G711, false, kFastWPB, 0, false>(L, blocks, st);
}

template <class Det, int LPC>
static void launch_general(const ToneLaunch &L, hipStream_t st)
{
    const int waves = (L.n_ch + kWave/LPC - 1)/(kWave/LPC);
    const int blocks = (waves + kWavesPerBlock - 1)/kWavesPerBlock;
    if constexpr (Det::kDigits)
    {
        if (L.digits)
        {
            hipLaunchKernelGGL((tone_bank_kernel<Det, LPC, kToneDigits>), dim3(blocks), dim3(kWave*kWavesPerBlock), 0, st, L);
            return;
        }
    }
    hipLaunchKernelGGL((tone_bank_kernel<Det, LPC>), dim3(blocks), dim3(kWave*kWavesPerBlock), 0, st, L);
}

template <class Det>
static void launch_tone(const ToneLaunch &L, hipStream_t st)
{
    // Lanes per channel: the streaming kernels run one channel per lane unless two are asked for (the split mapping
    // only ever paid on the general kernel, for banks too small to give every SIMD two waves).
    if (fast_eligible(L)  &&  !(L.fmt != 0  &&  forced_lpc() == 2))
    {
        const bool loader = use_loader(L.n_ch);
        if (L.fmt != 0)
            launch_fast<Det, 1, true>(L, st, loader);
        else if (g_forced_lpc == 2)
            launch_fast<Det, 2, false>(L, st, false);
        else
            launch_fast<Det, 1, false>(L, st, loader);
        return;
    }
    if (L.fmt != 0  ||  pick_lpc(L.n_ch) == 2)              // G.711 input: the LPC = 2 kernels hold the decode table
        launch_general<Det, 2>(L, st);
    else
        launch_general<Det, 1>(L, st);
}

// Banks of more than 16 bins per channel: always two lanes per channel (16 bins = 8 packed pairs per lane is what one
// lane's registers hold), linear PCM only; the streaming kernel without loader waves where the frame allows it.
template <class Det>
static void launch_tone_wide(const ToneLaunch &L, hipStream_t st)
{
    const int waves = (L.n_ch + 31)/32;
    const int blocks = (waves + kWavesPerBlock - 1)/kWavesPerBlock;
    if (fast_eligible(L)  &&  L.fmt == 0)
        launch_tone_fast<Det, 2, kRingSelf, false, false, kFastWPB, 0, false>(L, blocks, st);
    else
        hipLaunchKernelGGL((tone_bank_kernel<Det, 2>), dim3(blocks), dim3(kWave*kWavesPerBlock), 0, st, L);
}

// Banks of more than 32 bins per channel (a super-tone descriptor may name 64 pitches, private/super_tone_rx.h:44): four
// lanes per channel, 16 channels per wavefront, the general kernel.
template <class Det>
static void launch_tone_quad(const ToneLaunch &L, hipStream_t st)
{
    const int waves = (L.n_ch + 15)/16;
    const int blocks = (waves + kWavesPerBlock - 1)/kWavesPerBlock;
    hipLaunchKernelGGL((tone_bank_kernel<Det, 4>), dim3(blocks), dim3(kWave*kWavesPerBlock), 0, st, L);
}

// dtmf_rx_parms(), dtmf.c:421-445, on top of the defaults of dtmf_rx_init() (dtmf.c:470-476).  A field takes effect when
// its bit of set_mask is set and the reference's own test passes (twists >= 0 dB, threshold > -99 dBm0); without
// set_mask (a zeroed struct, or one from a caller built before the mask existed) a positive twist and a non-zero
// threshold do.
static void dtmf_levels(const spangpu_tone_params_t &tp, float &threshold, float &normal_twist, float &reverse_twist)
{
    threshold = 171029200.0f;
    normal_twist = 6.309f;
    reverse_twist = 2.512f;
    const bool m = (tp.set_mask != 0);
    if (m  ?  ((tp.set_mask & SPANGPU_TP_TWIST)  &&  tp.twist_db >= 0.0f)  :  (tp.twist_db > 0.0f))
        normal_twist = powf(10.0f, tp.twist_db/10.0f);
    if (m  ?  ((tp.set_mask & SPANGPU_TP_REVERSE_TWIST)  &&  tp.reverse_twist_db >= 0.0f)  :  (tp.reverse_twist_db > 0.0f))
        reverse_twist = powf(10.0f, tp.reverse_twist_db/10.0f);
    if (m  ?  ((tp.set_mask & SPANGPU_TP_THRESHOLD)  &&  tp.threshold_dbm0 > -99.0f)
           :  (tp.threshold_dbm0 > -99.0f  &&  tp.threshold_dbm0 != 0.0f))
        threshold = (float) ((102*102*32768.0f*32768.0f/2.0f)*powf(10.0f, (tp.threshold_dbm0 - 3.14f)/10.0f));
}

// The digits of the last launch as a compact list, for reports that leave the GPU (an RCCL gather to another rank, a
// small D2H copy): out[0] = number of blocks in which the debouncer accepted a digit (SPANGPU_BLK_CHANGE with a non-zero
// code: dtmf.c:318-340, Bell MF :629-655), out[1 + i] = channel | code << 20 | block << 28, in no particular order.
// A 65536-channel DTMF tick has 131072 record words and, on live lines, a few thousand such blocks.
__global__ __launch_bounds__(256) void digit_events_kernel(const uint32_t *rec, int n_ch, int maxb, uint32_t *out, int cap)
{
    const int ch = (int) (blockIdx.x*256 + threadIdx.x);
    if (ch >= n_ch)
        return;
    for (int b = 0;  b < maxb;  b++)
    {
        const uint32_t w = rec[(size_t) b*n_ch + ch];
        const uint32_t flags = (w >> 16) & 0xFF;
        const uint32_t code = (w >> 8) & 0xFF;
        if ((flags & kBlkValid)  &&  (flags & kBlkChange)  &&  code)
        {
            const uint32_t at = atomicAdd(out, 1u);
            if (at < (uint32_t) cap)
                out[1 + at] = (uint32_t) ch | (code << 20) | ((uint32_t) b << 28);
        }
    }
}


// ---- super-tone cadences on the device (cadence_dev.hpp has the walk) ---------------------------------------------------
struct Cadence
{
    int n_tones;
    int n_elems;
    int tone_len[kCadLdsTones + 1];     // elements of each of the first tones (host copy: spangpu_bank_cadence_set_state)
    int32_t *d_first;       // [n_tones + 1]
    int4 *d_elem;           // (pair, least blocks, most blocks, 0)
    int32_t *d_state;       // [kCadWords][n_ch]
    uint32_t *d_ev;         // [slots][n_ch][2]
    int32_t *d_count;       // [n_ch]
    uint32_t *h_ev;         // pinned
    int32_t *h_count;
    uint32_t *d_list;       // [2 + 3*slots_cap*n_ch]: two counters used in turn (a launch clears the next one's), then
                            // (channel, word 0, word 1) per event, a channel's together
    int which;              // the counter of the last launch
    uint32_t *h_list;       // pinned
    int slots_cap;
    int segments;
    unsigned done_serial;   // the launch whose records were matched last
    int last_slots;
    bool fused;             // the last detector launch matched the cadences itself (tone_fast.hpp, kToneCadence)
    bool list_due;          // ... and the compact list of its events has not been made yet
};

static void cadence_args(const spangpu_bank_s *b, const Cadence *c, CadenceArgs &A, int which);

__global__ __launch_bounds__(256) void cadence_init_kernel(int32_t *st, int n_ch, int first, int n)
{
    const int i = (int) (blockIdx.x*256 + threadIdx.x);
    if (i >= n)
        return;
    const int ch = first + i;
    for (int w = 0;  w < kCadWords;  w++)
        st[(size_t) w*n_ch + ch] = (w == 3  ||  w >= 4 + kCadHistory)  ?  0  :  -1;
}

// A launch of its own over the records of the last detector launch (whatever kernel made them).
__global__ __launch_bounds__(256) void cadence_kernel(const uint32_t *__restrict__ rec, int n_ch, int maxb, const CadenceArgs A)
{
    __shared__ uint32_t wg_events;
    __shared__ uint32_t wg_at;
    if (threadIdx.x == 0)
        wg_events = 0;
    __syncthreads();
    // lanes past the end of the bank walk the last channel again and write nothing (they have barriers to keep)
    const bool active = ((int) (blockIdx.x*256 + threadIdx.x) < n_ch);
    const int ch = active  ?  (int) (blockIdx.x*256 + threadIdx.x)  :  n_ch - 1;
    // the records of a 160-sample frame (one or two blocks) are asked for along with the state: one trip to memory in all
    const uint32_t rec0 = (maxb > 0)  ?  rec[ch]  :  0;
    const uint32_t rec1 = (maxb > 1)  ?  rec[(size_t) n_ch + ch]  :  0;
    const int n_ev = cadence_walk(A, ch, n_ch, maxb, rec0, rec1, rec, active);
    // the compact list: the workgroup takes room for all its events with one atomic on the list's counter (thousands of lanes
    // adding to one address cost more than the rest of the kernel), each channel a piece of that
    const uint32_t mine = (n_ev > 0)  ?  atomicAdd(&wg_events, (uint32_t) n_ev)  :  0;
    __syncthreads();
    if (threadIdx.x == 0  &&  wg_events > 0)
        wg_at = atomicAdd(A.list + A.which, wg_events);
    if (blockIdx.x == 0  &&  threadIdx.x == 0)
        A.list[A.which ^ 1] = 0;                // the next launch's counter (it starts when this one is over)
    __syncthreads();
    if (n_ev > 0)
        cadence_list_copy(A, ch, n_ch, n_ev, wg_at + mine);
}

// The compact list from the slot arrays, for launches whose detector kernel matched the cadences itself.
__global__ __launch_bounds__(256) void cadence_list_kernel(int n_ch, const CadenceArgs A)
{
    __shared__ uint32_t wg_events;
    __shared__ uint32_t wg_at;
    if (threadIdx.x == 0)
        wg_events = 0;
    __syncthreads();
    const int ch = (int) (blockIdx.x*256 + threadIdx.x);
    const int n_ev = (ch < n_ch)  ?  A.count[ch]  :  0;
    const uint32_t mine = (n_ev > 0)  ?  atomicAdd(&wg_events, (uint32_t) n_ev)  :  0;
    __syncthreads();
    if (threadIdx.x == 0  &&  wg_events > 0)
        wg_at = atomicAdd(A.list + A.which, wg_events);
    if (blockIdx.x == 0  &&  threadIdx.x == 0)
        A.list[A.which ^ 1] = 0;
    __syncthreads();
    if (n_ev > 0)
        cadence_list_copy(A, ch, n_ch, n_ev, wg_at + mine);
}

static void cadence_free(Cadence *c)
{
    if (c == nullptr)
        return;
    if (c->d_first) (void) hipFree(c->d_first);
    if (c->d_elem) (void) hipFree(c->d_elem);
    if (c->d_state) (void) hipFree(c->d_state);
    if (c->d_ev) (void) hipFree(c->d_ev);
    if (c->d_count) (void) hipFree(c->d_count);
    if (c->h_ev) (void) hipHostFree(c->h_ev);
    if (c->h_count) (void) hipHostFree(c->h_count);
    if (c->d_list) (void) hipFree(c->d_list);
    if (c->h_list) (void) hipHostFree(c->h_list);
    free(c);
}

extern "C" int spangpu_set_error(int code, const char *msg)
{
    return fail(code, "%s", msg);
}

__global__ __launch_bounds__(256) void probe_read_kernel(const uint4 *p, size_t n16, uint32_t *sink)
{
    const size_t stride = (size_t) gridDim.x*blockDim.x;
    uint32_t acc = 0;
    for (size_t i = (size_t) blockIdx.x*blockDim.x + threadIdx.x;  i < n16;  i += stride)
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load((const u32x4 *) (p + i));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

extern "C" {

int spangpu_tune_lanes_per_channel(int lpc)
{
    if (lpc != 0  &&  lpc != 1  &&  lpc != 2)
        return fail(SPANGPU_ERR_BAD_ARG, "lanes per channel must be 0 (auto), 1 or 2");
    g_forced_lpc = lpc;
    return SPANGPU_OK;
}

int spangpu_tune_tone_kernel(int variant)
{
    if (variant < 0  ||  variant > 3)
        return fail(SPANGPU_ERR_BAD_ARG, "variant must be 0 (auto), 1 (general), 2 (streaming, loader waves) or 3 (streaming)");
    g_tone_variant = variant;
    return SPANGPU_OK;
}

int spangpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

const char *spangpu_last_error(void)
{
    return g_err;
}

// Measurement aid: the streaming read ceiling of this device as seen by a plain kernel -- `bytes` of device memory (more
// than the 256 MB last level cache for an HBM figure) read `reps` times with 16-byte loads, timed with events.  The
// benchmarks quote it beside the 8 TB/s datasheet peak ("also report vs the measured stream ceiling").
int spangpu_probe_stream_read(int device, size_t bytes, int reps, double *gb_per_s)
{
    if (gb_per_s == nullptr  ||  bytes < (1u << 20)  ||  reps <= 0)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(device));
    uint4 *buf = nullptr;
    uint32_t *sink = nullptr;
    hipEvent_t e0 = nullptr;
    hipEvent_t e1 = nullptr;
    const size_t n16 = bytes/16;
    if (hipMalloc(&buf, n16*16) != hipSuccess  ||  hipMalloc(&sink, 64) != hipSuccess)
    {
        if (buf) (void) hipFree(buf);
        return fail(SPANGPU_ERR_NO_MEMORY, "no memory for the stream probe");
    }
    int rc = SPANGPU_OK;
    if (hipMemset(buf, 0x5A, n16*16) != hipSuccess  ||  hipEventCreate(&e0) != hipSuccess  ||  hipEventCreate(&e1) != hipSuccess)
        rc = fail(SPANGPU_ERR_HIP, "stream probe setup failed");
    if (rc == SPANGPU_OK)
    {
        hipLaunchKernelGGL(probe_read_kernel, dim3(256*16), dim3(256), 0, 0, buf, n16, sink);      // warm
        (void) hipEventRecord(e0, 0);
        for (int r = 0;  r < reps;  r++)
            hipLaunchKernelGGL(probe_read_kernel, dim3(256*16), dim3(256), 0, 0, buf, n16, sink);
        (void) hipEventRecord(e1, 0);
        float ms = 0.0f;
        if (hipEventSynchronize(e1) != hipSuccess  ||  hipEventElapsedTime(&ms, e0, e1) != hipSuccess  ||  ms <= 0.0f)
            rc = fail(SPANGPU_ERR_HIP, "stream probe failed");
        else
            *gb_per_s = (double) n16*16.0*reps/(ms*1e-3)/1e9;
    }
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    (void) hipFree(buf);
    (void) hipFree(sink);
    return rc;
}

const char *spangpu_version(void)
{
    return "spangpu 0.1 (gfx950)";
}

// make_goertzel_descriptor(), tone_detect.c:60-68: the argument is formed in double
// (M_PI is a double) and narrowed to float at the cosf() call.
float spangpu_goertzel_fac(float freq_hz)
{
    const double two_pi = 2.0f*3.14159265358979323846264338327;
    const float ratio = freq_hz/8000.0f;
    return 2.0f*cosf((float) (two_pi*ratio));
}

static int cadence_catch_up(spangpu_bank_t *b);

static void free_outputs(spangpu_bank_t *b)
{
    if (b->cur_rec == b->rec)
        b->cur_rec = nullptr;           // (a records buffer of the caller's stays what it is)
    if (b->rec) (void) hipFree(b->rec);
    if (b->rec_energy) (void) hipFree(b->rec_energy);
    if (b->rec_dur) (void) hipFree(b->rec_dur);
    if (b->trace) (void) hipFree(b->trace);
    if (b->h_rec) (void) hipHostFree(b->h_rec);
    if (b->h_energy) (void) hipHostFree(b->h_energy);
    if (b->h_dur) (void) hipHostFree(b->h_dur);
    b->rec = nullptr;
    b->rec_energy = nullptr;
    b->rec_dur = nullptr;
    b->trace = nullptr;
    b->h_rec = nullptr;
    b->h_energy = nullptr;
    b->h_dur = nullptr;
    b->maxb_cap = 0;
}

static int ensure_outputs(spangpu_bank_t *b, int maxb)
{
    if (maxb <= b->maxb_cap)
        return SPANGPU_OK;
    // The records of the launch before may still be owed to the cadence matcher (cadence_catch_up()): it reads them
    // from the buffer that is about to be freed, so it runs now, and has finished before the buffer goes.
    if (b->rec)
    {
        const int caught = cadence_catch_up(b);
        if (caught != SPANGPU_OK)
            return caught;
        HIP_TRY(hipStreamSynchronize(joined(b)));
    }
    free_outputs(b);
    const size_t n = (size_t) maxb*b->n_ch;
    HIP_TRY(hipMalloc(&b->rec, n*sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc(&b->h_rec, n*sizeof(uint32_t)));
    const bool want_energy = (b->kind == SPANGPU_DTMF  &&  b->tp.report_mode == SPANGPU_REPORT_REALTIME)
                             ||  b->kind == SPANGPU_SUPER_TONE;
    if (want_energy)
    {
        HIP_TRY(hipMalloc(&b->rec_energy, n*sizeof(float)));
        HIP_TRY(hipHostMalloc(&b->h_energy, n*sizeof(float)));
        HIP_TRY(hipMemsetAsync(b->rec_energy, 0, n*sizeof(float), joined(b)));
    }
    if (b->kind == SPANGPU_DTMF  &&  b->tp.report_mode == SPANGPU_REPORT_REALTIME)
    {
        HIP_TRY(hipMalloc(&b->rec_dur, n*sizeof(int32_t)));
        HIP_TRY(hipHostMalloc(&b->h_dur, n*sizeof(int32_t)));
        HIP_TRY(hipMemsetAsync(b->rec_dur, 0, n*sizeof(int32_t), joined(b)));
    }
    if (b->tp.trace  ||  b->kind == SPANGPU_GOERTZEL)
    {
        HIP_TRY(hipMalloc(&b->trace, n*(b->nb + 1)*sizeof(float)));
        HIP_TRY(hipMemsetAsync(b->trace, 0, n*(b->nb + 1)*sizeof(float), joined(b)));
    }
    b->maxb_cap = maxb;
    return SPANGPU_OK;
}

int spangpu_bank_create(spangpu_bank_t **bank, int device, int kind, int n_channels,
                        const void *params, size_t params_size)
{
    if (bank == nullptr  ||  n_channels <= 0)
        return fail(SPANGPU_ERR_BAD_ARG, "bad bank/n_channels");
    *bank = nullptr;
    if (spangpu_device_count() <= 0)
        return fail(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    if (device < 0  ||  device >= spangpu_device_count())
        return fail(SPANGPU_ERR_BAD_ARG, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));

    spangpu_bank_t *b = (spangpu_bank_t *) calloc(1, sizeof(*b));
    if (b == nullptr)
        return fail(SPANGPU_ERR_NO_MEMORY, "calloc");
    b->device = device;
    b->kind = kind;
    b->n_ch = n_channels;
    if (params)
        memcpy(&b->tp, params, (params_size < sizeof(b->tp))  ?  params_size  :  sizeof(b->tp));

    switch (kind)
    {
    case SPANGPU_DTMF:
    {
        // dtmf.c:104-119
        static const float freqs[8] = {697.0f, 770.0f, 852.0f, 941.0f, 1209.0f, 1336.0f, 1477.0f, 1633.0f};
        b->nb = 8;
        b->nsf = 21;     // v2[8] v3[8] energy z350[2] z440[2] (filter words unused when the notch is off)
        b->block_len = 102;
        for (int i = 0;  i < 8;  i++)
            b->fac[i] = spangpu_goertzel_fac(freqs[i]);
        dtmf_levels(b->tp, b->threshold, b->normal_twist, b->reverse_twist);
        break;
    }
    case SPANGPU_BELL_MF:
    {
        static const int freqs[6] = {700, 900, 1100, 1300, 1500, 1700};         // bell_r2_mf.c:251-254
        b->nb = 6;
        b->nsf = 12;
        b->block_len = 120;
        for (int i = 0;  i < 6;  i++)
            b->fac[i] = spangpu_goertzel_fac((float) freqs[i]);
        break;
    }
    case SPANGPU_R2_MF:
    {
        static const int fwd[6] = {1380, 1500, 1620, 1740, 1860, 1980};         // bell_r2_mf.c:264-267
        static const int back[6] = {1140, 1020, 900, 780, 660, 540};            // bell_r2_mf.c:269-272
        b->nb = 6;
        b->nsf = 12;
        b->block_len = 133;
        for (int i = 0;  i < 6;  i++)
            b->fac[i] = spangpu_goertzel_fac((float) ((b->tp.r2_fwd)  ?  fwd[i]  :  back[i]));
        break;
    }
    case SPANGPU_SUPER_TONE:
    case SPANGPU_GOERTZEL:
    {
        const int m = b->tp.n_bins;
        if (m < ((kind == SPANGPU_SUPER_TONE)  ?  2  :  1)  ||  m > SPANGPU_MAX_BINS)
        {
            free(b);
            return fail(SPANGPU_ERR_BAD_ARG, "n_bins %d out of range", m);
        }
        b->nb = (m <= 4)  ?  4  :  (m <= 8)  ?  8  :  (m <= 12)  ?  12  :  (m <= 16)  ?  16  :  (m <= 24)  ?  24  :  (m <= 32)  ?  32  :  64;
        b->nsf = 2*b->nb + 1;
        b->block_len = (kind == SPANGPU_SUPER_TONE)  ?  128  :  b->tp.block_len;
        if (b->block_len <= 0  ||  b->block_len > 65535)
        {
            free(b);
            return fail(SPANGPU_ERR_BAD_ARG, "block_len %d out of range", b->block_len);
        }
        if (kind == SPANGPU_GOERTZEL  &&  (b->tp.functor < 0  ||  b->tp.functor > SPANGPU_FUNCTOR_ADEMCO
                                           ||  (b->tp.functor == SPANGPU_FUNCTOR_ADEMCO  &&  m != 2)))
        {
            free(b);
            return fail(SPANGPU_ERR_BAD_ARG, "functor %d does not fit a bank of %d bins", b->tp.functor, m);
        }
        for (int i = 0;  i < m;  i++)
            b->fac[i] = b->tp.bin_fac[i];
        break;
    }
    default:
        free(b);
        return fail(SPANGPU_ERR_UNSUPPORTED, "bank kind %d not available from this entry point", kind);
    }

    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e != hipSuccess)
    {
        free(b);
        return fail(SPANGPU_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    b->own_stream = true;
    const size_t nf = (size_t) b->nsf*b->n_ch;
    if (hipMalloc(&b->sf, nf*sizeof(float)) != hipSuccess
        ||  hipMalloc(&b->si, (size_t) 2*b->n_ch*sizeof(int32_t)) != hipSuccess)
    {
        spangpu_bank_destroy(b);
        return fail(SPANGPU_ERR_NO_MEMORY, "hipMalloc of bank state failed");
    }
    (void) hipMemsetAsync(b->sf, 0, nf*sizeof(float), joined(b));
    (void) hipMemsetAsync(b->si, 0, (size_t) 2*b->n_ch*sizeof(int32_t), joined(b));
    (void) hipEventCreate(&b->ev0);
    (void) hipEventCreate(&b->ev1);
    (void) hipStreamSynchronize(joined(b));
    *bank = b;
    return SPANGPU_OK;
}

int spangpu_bank_destroy(spangpu_bank_t *b)
{
    if (b == nullptr)
        return SPANGPU_OK;
    (void) hipSetDevice(b->device);
    if (b->stream)
        (void) hipStreamSynchronize(joined(b));
    if (b->stream2)
    {
        (void) hipStreamSynchronize(b->stream2);
        (void) hipStreamDestroy(b->stream2);
    }
    if (b->ev_in) (void) hipEventDestroy(b->ev_in);
    if (b->ev_q) (void) hipEventDestroy(b->ev_q);
    free_outputs(b);
    cadence_free(b->cad);
    if (b->sf) (void) hipFree(b->sf);
    if (b->si) (void) hipFree(b->si);
    if (b->d_amp) (void) hipFree(b->d_amp);
    if (b->d_lens) (void) hipFree(b->d_lens);
    if (b->h_lens) (void) hipHostFree(b->h_lens);
    if (b->chan_parms) (void) hipFree(b->chan_parms);
    free(b->h_chan_parms);
    if (b->ev0) (void) hipEventDestroy(b->ev0);
    if (b->ev1) (void) hipEventDestroy(b->ev1);
    if (b->own_stream  &&  b->stream)
        (void) hipStreamDestroy(b->stream);
    free(b);
    return SPANGPU_OK;
}

int spangpu_bank_kind(const spangpu_bank_t *b) { return b  ?  b->kind  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_bank_device(const spangpu_bank_t *b) { return b  ?  b->device  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_bank_channels(const spangpu_bank_t *b) { return b  ?  b->n_ch  :  SPANGPU_ERR_BAD_ARG; }

int spangpu_bank_set_stream(spangpu_bank_t *b, void *hip_stream)
{
    if (b == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "null bank");
    (void) hipStreamSynchronize(joined(b));
    if (b->own_stream)
        (void) hipStreamDestroy(b->stream);
    if (hip_stream)
    {
        b->stream = (hipStream_t) hip_stream;
        b->own_stream = false;
    }
    else
    {
        HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
        b->own_stream = true;
    }
    return SPANGPU_OK;
}

// Streams of their own for the banks of one tick, on hardware queues that are different ones.  A HIP stream is bound to one
// of a few hardware queues by the runtime, in an order a caller does not control: two of three freshly made streams can land
// on one queue, and the launches of those two banks then run one after the other (configs[2], round 6: Bell MF and super-tone
// on one queue, 30 us a tick instead of 17 -- profiles/r6_mixed_trace_overlap_collision.txt).  Stream priorities do give
// queues of their own, but the lower queues then wait for the higher ones (measured: 42.7 us a tick where three plain streams
// on three queues take 15.3; only the longest bank's stream at high priority: 25 - 26 us against 20).  So: candidates are made, and a pair is PROBED -- a spin kernel of 200 us on each, started
// together: they end together on two queues and one after the other on one -- until every bank has a stream that runs beside
// all the others' (a few milliseconds, once).  Returns the number of banks whose stream was proven to run beside every other
// bank's (n_banks unless the device ran out of queues: then the rest share).
__global__ void queue_probe_kernel(long long ticks)
{
    const long long t0 = wall_clock64();           // the 100 MHz constant clock
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(8);
}

static bool probe_streams_overlap(hipStream_t a, hipStream_t b)
{
    constexpr long long kTicks = 20000;             // 200 us
    for (int attempt = 0;  attempt < 3;  attempt++)
    {
        (void) hipStreamSynchronize(a);
        (void) hipStreamSynchronize(b);
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(queue_probe_kernel, dim3(1), dim3(64), 0, a, kTicks);
        hipLaunchKernelGGL(queue_probe_kernel, dim3(1), dim3(64), 0, b, kTicks);
        (void) hipStreamSynchronize(a);
        (void) hipStreamSynchronize(b);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < 290.0)
            return true;                            // both spun at once
        if (us > 390.0  &&  us < 600.0)
            return false;                           // one behind the other
        // (anything else: the host was held up; look again)
    }
    return false;
}

int spangpu_banks_own_queues(spangpu_bank_t *const *banks, int n_banks)
{
    if (banks == nullptr  ||  n_banks < 1  ||  n_banks > kMaxMulti)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments (at most %d banks)", kMaxMulti);
    for (int k = 0;  k < n_banks;  k++)
    {
        if (banks[k] == nullptr  ||  banks[k]->device != banks[0]->device)
            return fail(SPANGPU_ERR_BAD_ARG, "null bank, or banks on different devices");
    }
    HIP_TRY(hipSetDevice(banks[0]->device));
    constexpr int kCandidates = 12;
    hipStream_t cand[kCandidates];
    int n_cand = 0;
    hipStream_t chosen[kMaxMulti];
    int n_chosen = 0;
    while (n_chosen < n_banks  &&  n_cand < kCandidates)
    {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
            break;
        cand[n_cand++] = st;
        bool beside_all = true;
        for (int j = 0;  j < n_chosen  &&  beside_all;  j++)
            beside_all = probe_streams_overlap(chosen[j], st);
        if (beside_all)
        {
            chosen[n_chosen++] = st;
            cand[--n_cand] = nullptr;               // (kept: not one of the candidates to give back)
        }
    }
    const int proven = n_chosen;
    // out of queues (or of candidates): the banks left over take candidates as they are
    while (n_chosen < n_banks  &&  n_cand > 0)
        chosen[n_chosen++] = cand[--n_cand];
    while (n_chosen < n_banks)
    {
        hipStream_t st = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        chosen[n_chosen++] = st;
    }
    for (int i = 0;  i < n_cand;  i++)
        (void) hipStreamDestroy(cand[i]);
    for (int k = 0;  k < n_banks;  k++)
    {
        spangpu_bank_t *b = banks[k];
        HIP_TRY(hipStreamSynchronize(joined(b)));
        if (b->own_stream)
            (void) hipStreamDestroy(b->stream);
        b->stream = chosen[k];
        b->own_stream = true;
    }
    return proven;
}

// The HIP stream the bank launches on (its own, unless spangpu_bank_set_stream() gave it another).
void *spangpu_bank_get_stream(spangpu_bank_t *b)
{
    return b  ?  (void *) joined(b)  :  nullptr;
}

// Queue mode.  queues = 2: launches the streaming kernel serves are cut in two workgroup ranges, the second on a stream
// (hardware queue) the bank owns; queues = 1: one launch on the bank's stream (the default); queues = 0: the library's
// choice -- two from 131 072 channels (measured, profiles/r5_probe_mq.log: 131 072 channels 18.3 -> 16.5 us a tick, 262 144
// 35.0 -> 29.5, 1 048 576 118 -> 109; 65 536 channels 11.15 -> 11.7: one).  Results are those of one launch, bit for bit.
// Ordering: the second stream starts a tick behind whatever the bank's stream held when the tick was queued, and every
// spangpu_bank_* call that reads results, edits state or waits joins it back first.  A caller that puts work of its own on
// the bank's stream behind a launch -- an event, a collective reading the records buffer -- calls spangpu_bank_join() first.
int spangpu_bank_set_queues(spangpu_bank_t *b, int queues)
{
    if (b == nullptr  ||  queues < 0  ||  queues > 2)
        return fail(SPANGPU_ERR_BAD_ARG, "queues must be 0 (the library's choice), 1 or 2");
    HIP_TRY(hipSetDevice(b->device));
    // The library's own choice is two queues only for a bank on its OWN stream: there every frame reaches the bank through this
    // library (the host copy, or a device frame that is complete when the call is made), and the second queue's ordering is the
    // library's business.  On a caller's stream (spangpu_bank_set_stream) frames may be produced by the caller's kernels on
    // that stream between two ticks, which the second queue would not wait for -- half the channels would read a frame still
    // being written.  A caller who asks for two queues explicitly takes the contract of include/spangpu.h (frames complete at
    // the call, or spangpu_bank_join() / _get_stream() behind whatever made them) with it.
    if (queues == 0)
        queues = (b->n_ch >= 131072  &&  b->own_stream)  ?  2  :  1;
    (void) joined(b);
    if (queues == 2  &&  b->stream2 == nullptr)
    {
        HIP_TRY(hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&b->ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b->ev_q, hipEventDisableTiming));
    }
    b->queues = queues;
    return queues;
}

int spangpu_bank_join(spangpu_bank_t *b)
{
    if (b == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "null bank");
    (void) joined(b);
    return SPANGPU_OK;
}

int spangpu_bank_set_timing(spangpu_bank_t *b, int on)
{
    if (b == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "null bank");
    b->timing = (on != 0);
    return SPANGPU_OK;
}

static void fill_launch(ToneLaunch &L, spangpu_bank_t *b, const int16_t *d_amp, long long d_stride, int samples, int layout,
                        int maxb, int force_end)
{
    memset(&L, 0, sizeof(L));
    L.amp = d_amp;
    L.stride = d_stride;
    L.samples = samples;
    L.n_ch = b->n_ch;
    L.layout = layout;
    L.fmt = b->next_fmt;
    L.lens = b->next_lens;
    L.lens_ragged = (b->next_lens  &&  b->next_ragged)  ?  1  :  0;
    L.chan_parms = b->chan_parms;
    L.digits = nullptr;
    if (b->ext_digits  &&  (size_t) maxb*b->n_ch <= b->ext_digits_bytes)
    {
        L.digits = b->ext_digits + (size_t) b->ext_digits_next*b->ext_digits_bytes;
        b->ext_digits_next = (b->ext_digits_next + 1 < b->ext_digits_slices)  ?  (b->ext_digits_next + 1)  :  0;
    }
    L.functor = (b->kind == SPANGPU_GOERTZEL)  ?  b->tp.functor  :  0;
    L.functor_threshold = b->tp.functor_threshold;
    {
        const int spc = L.fmt  ?  16  :  8;                 // samples per 16 bytes
        L.aligned16 = (layout == SPANGPU_LAYOUT_CHANNEL_MAJOR
                       &&  (((uintptr_t) d_amp) & 15) == 0
                       &&  (d_stride & (spc - 1)) == 0
                       &&  d_stride >= ((samples + spc - 1) & ~(spc - 1)))  ?  1  :  0;
    }
    L.sf = b->sf;
    L.si = b->si;
    L.rec = b->ext_rec  ?  b->ext_rec  :  b->rec;
    b->cur_rec = L.rec;
    memset(&L.cad, 0, sizeof(L.cad));
    if (b->cad)
    {
        // the streaming kernel matches the cadences in its epilogue when the event buffers hold this launch's slots and
        // every channel takes part (launch_fast() decides; any other kernel leaves them to cadence_kernel)
        b->cad->fused = false;
        b->cad->list_due = false;
        if (kCadSlotsPerBlock*maxb <= b->cad->slots_cap  &&  maxb > 0  &&  b->next_lens == nullptr  &&  !force_end
            &&  b->cad->n_tones <= kCadLdsTones  &&  b->cad->n_elems <= kCadLdsElems)
            cadence_args(b, b->cad, L.cad, b->cad->which ^ 1);
    }
    L.rec_energy = b->rec_energy;
    L.rec_dur = b->rec_dur;
    L.trace = b->trace;
    L.maxb = maxb;
    L.nbins = (b->kind == SPANGPU_SUPER_TONE  ||  b->kind == SPANGPU_GOERTZEL)  ?  b->tp.n_bins  :  b->nb;
    L.block_len = b->block_len;
    L.force_end = force_end;
    L.realtime = (b->kind == SPANGPU_DTMF  &&  b->tp.report_mode == SPANGPU_REPORT_REALTIME)  ?  1  :  0;
    for (int i = 0;  i < kMaxBins;  i++)
        L.fac[i] = b->fac[i];
    L.threshold = b->threshold;
    L.normal_twist = b->normal_twist;
    L.reverse_twist = b->reverse_twist;
}

// A launch is about to overwrite the block records of the one before it.  If cadences are installed and nobody has taken
// that launch's records to the matcher yet (it was not served by the streaming kernel, and the caller did not ask for its
// cadence events), do it now: whether a bank's runs are counted must not depend on which kernel its frames happen to get.
static int cadence_catch_up(spangpu_bank_t *b)
{
    if (b->cad == nullptr  ||  b->launch_serial == 0  ||  b->cad->done_serial == b->launch_serial)
        return SPANGPU_OK;
    const int rc = spangpu_bank_cadence_run(b);
    return (rc < 0)  ?  rc  :  SPANGPU_OK;
}

static int launch_bank(spangpu_bank_t *b, const int16_t *d_amp, long long d_stride, int samples, int layout, int maxb, int force_end)
{
    const int caught = cadence_catch_up(b);
    if (caught != SPANGPU_OK)
        return caught;
    ToneLaunch L;
    fill_launch(L, b, d_amp, d_stride, samples, layout, maxb, force_end);

    // Queue mode: a launch the one-lane streaming kernel serves, of a bank of at least two workgroups per half and without
    // cadences (their epilogue's list is made on the bank's stream), goes out as two launches on two streams; the second stream
    // first waits for what the bank's stream holds now (the caller's frame, an earlier launch of another kind).  Anything
    // else joins the second stream into the bank's stream and runs there as ever.
    const int total_wg = (b->n_ch + kWave*kFastWPB - 1)/(kWave*kFastWPB);
    const bool split = b->queues == 2  &&  b->stream2 != nullptr  &&  !b->timing  &&  b->cad == nullptr  &&  b->nb <= 16  &&  total_wg >= 4
                       &&  fast_eligible(L)  &&  forced_lpc() != 2;
    if (split  &&  b->main_touched)
    {
        // something else went onto the bank's stream since the last split launch (a state edit, a launch of the other kernel
        // family, work of the caller's behind spangpu_bank_get_stream() / _join()): the second half runs behind it
        HIP_TRY(hipEventRecord(b->ev_in, b->stream));
        HIP_TRY(hipStreamWaitEvent(b->stream2, b->ev_in, 0));
        b->main_touched = false;
    }
    for (int half = 0;  half < (split  ?  2  :  1);  half++)
    {
    hipStream_t st = split  ?  b->stream  :  joined(b);
    if (split)
    {
        L.wg0 = half  ?  total_wg/2  :  0;
        L.wgn = half  ?  (total_wg - total_wg/2)  :  total_wg/2;
        st = half  ?  b->stream2  :  b->stream;
    }
    if (b->timing)
        HIP_TRY(hipEventRecord(b->ev0, st));
    g_cadence_fused = false;
    switch (b->kind)
    {
    case SPANGPU_DTMF:
        if (b->chan_parms  ?  (b->n_filter_on > 0)  :  (b->tp.filter_dialtone != 0))
            launch_tone<DtmfDet<true>>(L, st);
        else
            launch_tone<DtmfDet<false>>(L, st);
        break;
    case SPANGPU_BELL_MF:
        launch_tone<BellMfDet>(L, st);
        break;
    case SPANGPU_R2_MF:
        launch_tone<R2MfDet>(L, st);
        break;
    case SPANGPU_SUPER_TONE:
        switch (b->nb)
        {
        case 4:  launch_tone<MultiDet<4, true>>(L, st);  break;
        case 8:  launch_tone<MultiDet<8, true>>(L, st);  break;
        case 12: launch_tone<MultiDet<12, true>>(L, st); break;
        case 16: launch_tone<MultiDet<16, true>>(L, st); break;
        case 24: launch_tone_wide<MultiDet<24, true>>(L, st); break;
        case 32: launch_tone_wide<MultiDet<32, true>>(L, st); break;
        default: launch_tone_quad<MultiDet<64, true>>(L, st); break;
        }
        break;
    case SPANGPU_GOERTZEL:
        switch (b->nb)
        {
        case 4:  launch_tone<MultiDet<4, false>>(L, st);  break;
        case 8:  launch_tone<MultiDet<8, false>>(L, st);  break;
        case 12: launch_tone<MultiDet<12, false>>(L, st); break;
        case 16: launch_tone<MultiDet<16, false>>(L, st); break;
        case 24: launch_tone_wide<MultiDet<24, false>>(L, st); break;
        case 32: launch_tone_wide<MultiDet<32, false>>(L, st); break;
        default: launch_tone_quad<MultiDet<64, false>>(L, st); break;
        }
        break;
    default:
        return fail(SPANGPU_ERR_UNSUPPORTED, "kind %d", b->kind);
    }
    HIP_TRY(hipGetLastError());
    }
    if (split)
        b->s2_busy = true;
    if (b->cad  &&  g_cadence_fused)
    {
        b->cad->fused = true;
        b->cad->list_due = true;
    }
    if (b->timing)
    {
        HIP_TRY(hipEventRecord(b->ev1, joined(b)));
        b->ev_valid = true;
    }
    return SPANGPU_OK;
}

int spangpu_bank_rx(spangpu_bank_t *b, const int16_t *amp, int mem, int layout, int samples, long long stride)
{
    if (b == nullptr  ||  amp == nullptr  ||  samples < 0)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (layout != SPANGPU_LAYOUT_CHANNEL_MAJOR  &&  layout != SPANGPU_LAYOUT_SAMPLE_MAJOR)
        return fail(SPANGPU_ERR_BAD_ARG, "bad layout");
    if (samples == 0)
        return 0;
    HIP_TRY(hipSetDevice(b->device));
    if (stride <= 0)
        stride = (layout == SPANGPU_LAYOUT_CHANNEL_MAJOR)  ?  samples  :  b->n_ch;

    const int maxb = (samples + b->block_len - 1)/b->block_len;
    int rc = ensure_outputs(b, (maxb > 0)  ?  maxb  :  1);
    if (rc != SPANGPU_OK)
        return rc;

    const int16_t *d_amp = amp;
    long long d_stride = stride;
    if (mem == SPANGPU_MEM_HOST)
    {
        if (layout == SPANGPU_LAYOUT_CHANNEL_MAJOR)
        {
            const long long padded = (samples + 7) & ~7LL;
            const size_t need = (size_t) padded*b->n_ch + 8;
            if (need > b->d_amp_cap)
            {
                if (b->d_amp) (void) hipFree(b->d_amp);
                b->d_amp = nullptr;
                b->d_amp_cap = 0;
                HIP_TRY(hipMalloc(&b->d_amp, need*sizeof(int16_t)));
                b->d_amp_cap = need;
            }
            HIP_TRY(hipMemcpy2DAsync(b->d_amp, padded*sizeof(int16_t), amp, stride*sizeof(int16_t),
                                     samples*sizeof(int16_t), b->n_ch, hipMemcpyHostToDevice, joined(b)));
            HIP_TRY(hipStreamSynchronize(joined(b)));       // amp[] is only borrowed for the duration of the call
            d_stride = padded;
        }
        else
        {
            const size_t need = (size_t) samples*b->n_ch + 8;
            if (need > b->d_amp_cap)
            {
                if (b->d_amp) (void) hipFree(b->d_amp);
                b->d_amp = nullptr;
                b->d_amp_cap = 0;
                HIP_TRY(hipMalloc(&b->d_amp, need*sizeof(int16_t)));
                b->d_amp_cap = need;
            }
            HIP_TRY(hipMemcpy2DAsync(b->d_amp, b->n_ch*sizeof(int16_t), amp, stride*sizeof(int16_t),
                                     b->n_ch*sizeof(int16_t), samples, hipMemcpyHostToDevice, joined(b)));
            HIP_TRY(hipStreamSynchronize(joined(b)));
            d_stride = b->n_ch;
        }
        d_amp = b->d_amp;
    }
    else if (mem != SPANGPU_MEM_DEVICE)
    {
        return fail(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    }

    if (b->ext_rec  &&  (size_t) maxb*b->n_ch*sizeof(uint32_t) > b->ext_rec_bytes)
        return fail(SPANGPU_ERR_BAD_ARG, "records buffer too small for %d blocks", maxb);
    rc = launch_bank(b, d_amp, d_stride, samples, layout, maxb, 0);
    if (rc < 0)
        return rc;
    b->last_maxb = maxb;
    b->launch_serial++;
    b->last_samples = samples;
    return 0;
}

// spangpu_bank_rx() for a tick in which not every channel has a frame, or not all frames are of one length: channel c
// takes part with lens[c] samples of its row (0: it sits the call out -- its filters, block phase and debounce state are
// exactly as they were, and it reports no block).  lens[] is host memory whatever `mem` says; frames are channel-major.
int spangpu_bank_rx_var(spangpu_bank_t *b, const int16_t *amp, int mem, const int32_t *lens, int max_samples, long long stride)
{
    if (b == nullptr  ||  amp == nullptr  ||  lens == nullptr  ||  max_samples < 0)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    int longest = 0;
    for (int c = 0;  c < b->n_ch;  c++)
    {
        if (lens[c] < 0  ||  lens[c] > max_samples)
            return fail(SPANGPU_ERR_BAD_ARG, "channel %d: %d samples, outside 0..%d", c, lens[c], max_samples);
        if (lens[c] > longest)
            longest = lens[c];
    }
    if (longest == 0)
        return 0;
    if (stride <= 0)
        stride = max_samples;
    bool ragged = false;
    bool all = true;
    for (int c = 0;  c < b->n_ch;  c++)
    {
        ragged |= (lens[c] != 0  &&  lens[c] != longest);
        all &= (lens[c] == longest);
    }
    if (all)
        return spangpu_bank_rx(b, amp, mem, SPANGPU_LAYOUT_CHANNEL_MAJOR, longest, stride);
    HIP_TRY(hipSetDevice(b->device));
    if (b->d_lens == nullptr)
    {
        HIP_TRY(hipMalloc(&b->d_lens, (size_t) b->n_ch*sizeof(int32_t)));
        HIP_TRY(hipHostMalloc(&b->h_lens, (size_t) b->n_ch*sizeof(int32_t), hipHostMallocDefault));
    }
    HIP_TRY(hipStreamSynchronize(joined(b)));               // the previous call's copy out of h_lens is done
    memcpy(b->h_lens, lens, (size_t) b->n_ch*sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(b->d_lens, b->h_lens, (size_t) b->n_ch*sizeof(int32_t), hipMemcpyHostToDevice, joined(b)));
    b->next_lens = b->d_lens;
    b->next_ragged = ragged;
    const int rc = spangpu_bank_rx(b, amp, mem, SPANGPU_LAYOUT_CHANNEL_MAJOR, longest, stride);
    b->next_lens = nullptr;
    b->next_ragged = false;
    return rc;
}

// The per-channel parameter table of a DTMF bank, made on first use with the bank's values in every channel
static int ensure_chan_parms(spangpu_bank_t *b)
{
    const size_t n = (size_t) b->n_ch;
    if (b->chan_parms)
        return SPANGPU_OK;
    if ((b->h_chan_parms = (float *) malloc(4*n*sizeof(float))) == nullptr)
        return fail(SPANGPU_ERR_NO_MEMORY, "malloc");
    for (size_t c = 0;  c < n;  c++)
    {
        b->h_chan_parms[c] = b->threshold;
        b->h_chan_parms[n + c] = b->normal_twist;
        b->h_chan_parms[2*n + c] = b->reverse_twist;
        b->h_chan_parms[3*n + c] = b->tp.filter_dialtone  ?  1.0f  :  0.0f;
    }
    b->n_filter_on = b->tp.filter_dialtone  ?  b->n_ch  :  0;
    float *d = nullptr;
    if (hipMalloc(&d, 4*n*sizeof(float)) != hipSuccess)
    {
        free(b->h_chan_parms);
        b->h_chan_parms = nullptr;
        return fail(SPANGPU_ERR_NO_MEMORY, "hipMalloc of per-channel parameters failed");
    }
    HIP_TRY(hipMemcpy(d, b->h_chan_parms, 4*n*sizeof(float), hipMemcpyHostToDevice));
    b->chan_parms = d;
    return SPANGPU_OK;
}

// Parameters of ONE channel of a DTMF bank: what dtmf_rx_parms() (dtmf.c:421-445) does to one detector.  The fields of
// `params` that count are filter_dialtone (< 0: leave as it is; else the notch states of the channel restart, as in
// dtmf.c:428-434), twist_db, reverse_twist_db and threshold_dbm0 under set_mask.  From the first such call on the bank
// carries its thresholds per channel (three loads per channel and block end more); channels never named keep the
// bank's values.
int spangpu_bank_set_channel_params(spangpu_bank_t *b, int channel, const spangpu_tone_params_t *params, size_t params_size)
{
    if (b == nullptr  ||  params == nullptr  ||  channel < 0  ||  channel >= b->n_ch)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (b->kind != SPANGPU_DTMF)
        return fail(SPANGPU_ERR_UNSUPPORTED, "per-channel parameters exist for DTMF banks only");
    spangpu_tone_params_t tp;
    memset(&tp, 0, sizeof(tp));
    memcpy(&tp, params, (params_size < sizeof(tp))  ?  params_size  :  sizeof(tp));
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    const size_t n = (size_t) b->n_ch;
    const int rc0 = ensure_chan_parms(b);
    if (rc0 != SPANGPU_OK)
        return rc0;
    float *h = b->h_chan_parms;
    // the channel's present values stand where the call leaves a field alone
    spangpu_tone_params_t q = tp;
    if (q.set_mask == 0)
        q.set_mask = ((q.twist_db > 0.0f)  ?  SPANGPU_TP_TWIST  :  0)  |  ((q.reverse_twist_db > 0.0f)  ?  SPANGPU_TP_REVERSE_TWIST  :  0)
                     |  ((q.threshold_dbm0 > -99.0f  &&  q.threshold_dbm0 != 0.0f)  ?  SPANGPU_TP_THRESHOLD  :  0)  |  0x40000000;
    float thr, nt, rt;
    dtmf_levels(q, thr, nt, rt);
    if ((q.set_mask & SPANGPU_TP_THRESHOLD)  &&  q.threshold_dbm0 > -99.0f)
        h[channel] = thr;
    if ((q.set_mask & SPANGPU_TP_TWIST)  &&  q.twist_db >= 0.0f)
        h[n + channel] = nt;
    if ((q.set_mask & SPANGPU_TP_REVERSE_TWIST)  &&  q.reverse_twist_db >= 0.0f)
        h[2*n + channel] = rt;
    if (tp.filter_dialtone >= 0)
    {
        const float on = tp.filter_dialtone  ?  1.0f  :  0.0f;
        b->n_filter_on += (int) on - (int) h[3*n + channel];
        h[3*n + channel] = on;
        for (int i = 0;  i < 4;  i++)
            HIP_TRY(hipMemsetAsync(b->sf + (size_t) (17 + i)*n + channel, 0, sizeof(float), joined(b)));
    }
    for (int i = 0;  i < 4;  i++)
        HIP_TRY(hipMemcpyAsync(b->chan_parms + (size_t) i*n + channel, &h[(size_t) i*n + channel], sizeof(float), hipMemcpyHostToDevice, joined(b)));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    return SPANGPU_OK;
}

// spangpu_bank_rx() for G.711 input: `codes` holds one A-law or u-law byte per sample, channel-major (the wire format
// of a trunk); the kernel decodes through a 256-entry table in LDS (alaw_to_linear / ulaw_to_linear,
// src/spandsp/g711.h:165-175,239-252), so the PCM read traffic of the detector is halved and no decode pass is needed.
int spangpu_bank_rx_g711(spangpu_bank_t *b, const uint8_t *codes, int mem, int law, int samples, long long stride)
{
    if (b == nullptr  ||  codes == nullptr  ||  samples < 0)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (law != SPANGPU_G711_ALAW  &&  law != SPANGPU_G711_ULAW)
        return fail(SPANGPU_ERR_BAD_ARG, "law must be SPANGPU_G711_ALAW or SPANGPU_G711_ULAW");
    if (samples == 0)
        return 0;
    HIP_TRY(hipSetDevice(b->device));
    if (stride <= 0)
        stride = samples;
    const int maxb = (samples + b->block_len - 1)/b->block_len;
    int rc = ensure_outputs(b, (maxb > 0)  ?  maxb  :  1);
    if (rc != SPANGPU_OK)
        return rc;
    const uint8_t *d_codes = codes;
    long long d_stride = stride;
    if (mem == SPANGPU_MEM_HOST)
    {
        const long long padded = (samples + 15) & ~15LL;
        const size_t need = ((size_t) padded*b->n_ch + 16 + 1)/2;          // in int16 units of the staging buffer
        if (need > b->d_amp_cap)
        {
            if (b->d_amp) (void) hipFree(b->d_amp);
            b->d_amp = nullptr;
            b->d_amp_cap = 0;
            HIP_TRY(hipMalloc(&b->d_amp, need*sizeof(int16_t)));
            b->d_amp_cap = need;
        }
        HIP_TRY(hipMemcpy2DAsync(b->d_amp, padded, codes, stride, samples, b->n_ch, hipMemcpyHostToDevice, joined(b)));
        HIP_TRY(hipStreamSynchronize(joined(b)));           // codes[] is only borrowed for the duration of the call
        d_codes = (const uint8_t *) b->d_amp;
        d_stride = padded;
    }
    else if (mem != SPANGPU_MEM_DEVICE)
    {
        return fail(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    }
    if (b->ext_rec  &&  (size_t) maxb*b->n_ch*sizeof(uint32_t) > b->ext_rec_bytes)
        return fail(SPANGPU_ERR_BAD_ARG, "records buffer too small for %d blocks", maxb);
    b->next_fmt = law;
    rc = launch_bank(b, (const int16_t *) d_codes, d_stride, samples, SPANGPU_LAYOUT_CHANNEL_MAJOR, maxb, 0);
    b->next_fmt = 0;
    if (rc < 0)
        return rc;
    b->last_maxb = maxb;
    b->launch_serial++;
    b->last_samples = samples;
    return 0;
}

// Several banks, one launch (tone_multi_kernel): banks[k] is advanced by `samples` samples of amps[k].  All banks must
// be on the same device and stream and hold device-resident, channel-major frames; kinds that can share a launch are
// DTMF (without the dial-tone filter), Bell MF, R2 MF and super-tone.  Results are read per bank as after
// spangpu_bank_rx().
int spangpu_banks_rx(spangpu_bank_t *const *banks, const int16_t *const *amps, int n_banks, int samples, const long long *strides)
{
    if (banks == nullptr  ||  amps == nullptr  ||  n_banks < 1  ||  n_banks > kMaxMulti  ||  samples < 0)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments (at most %d banks per launch)", kMaxMulti);
    if (samples == 0)
        return 0;
    ToneMultiLaunch M;
    memset(&M, 0, sizeof(M));
    int total_ch = 0;
    for (int k = 0;  k < n_banks;  k++)
    {
        spangpu_bank_t *b = banks[k];
        if (b == nullptr  ||  amps[k] == nullptr)
            return fail(SPANGPU_ERR_BAD_ARG, "null bank or frame");
        if (b->device != banks[0]->device)
            return fail(SPANGPU_ERR_BAD_ARG, "banks of one call must be on one device");
        total_ch += b->n_ch;
    }
    // Banks that were given streams of their own (spangpu_bank_set_stream) are advanced by a launch each, every one on its
    // bank's stream: free-running hardware queues put one bank's launch boundary, start burst and write-back under the other
    // banks' steady state (configs[2], 131 072 channels in three banks: 23.7 us a tick as one launch, 18.4 us as three
    // launches on three streams; three launches on ONE stream 32.3 -- profiles/r5_mixed_streams.log).
    // (Round 6: banks are grouped by stream -- those that share one share a launch on it, a bank alone on its stream gets its
    // own: configs[2] as Bell MF + R2 MF in one launch on one queue beside the super-tone bank with its cadence matcher on
    // another balances the two queues.)
    bool own_streams = false;
    for (int k = 1;  k < n_banks;  k++)
        own_streams = own_streams  ||  (banks[k]->stream != banks[0]->stream);
    if (own_streams)
    {
        bool done[kMaxMulti] = {false, false, false, false};
        for (int k = 0;  k < n_banks;  k++)
        {
            if (done[k])
                continue;
            spangpu_bank_t *grp[kMaxMulti];
            const int16_t *gamps[kMaxMulti];
            long long gstrides[kMaxMulti];
            int n = 0;
            for (int j = k;  j < n_banks;  j++)
            {
                if (!done[j]  &&  banks[j]->stream == banks[k]->stream)
                {
                    grp[n] = banks[j];
                    gamps[n] = amps[j];
                    gstrides[n] = strides  ?  strides[j]  :  0;
                    n++;
                    done[j] = true;
                }
            }
            const int rc = (n == 1)  ?  spangpu_bank_rx(grp[0], gamps[0], SPANGPU_MEM_DEVICE, SPANGPU_LAYOUT_CHANNEL_MAJOR, samples, gstrides[0])
                                     :  spangpu_banks_rx(grp, gamps, n, samples, strides  ?  gstrides  :  nullptr);
            if (rc < 0)
                return rc;
        }
        return 0;
    }
    for (int k = 0;  k < n_banks;  k++)
        (void) joined(banks[k]);
    HIP_TRY(hipSetDevice(banks[0]->device));
    bool all_fast = true;
    for (int k = 0;  k < n_banks;  k++)
    {
        spangpu_bank_t *b = banks[k];
        switch (b->kind)
        {
        case SPANGPU_DTMF:
            // the shared launch runs the unfiltered detector: what counts is whether any channel's filter is on (a bank
            // given per-channel parameters keeps the switch per channel; launch_bank() tests the same)
            if (b->chan_parms  ?  (b->n_filter_on > 0)  :  (b->tp.filter_dialtone != 0))
                return fail(SPANGPU_ERR_UNSUPPORTED, "a DTMF bank with a dial-tone filter switched on cannot share a launch");
            M.kind[k] = TONE_K_DTMF;
            break;
        case SPANGPU_BELL_MF: M.kind[k] = TONE_K_BELL; break;
        case SPANGPU_R2_MF: M.kind[k] = TONE_K_R2; break;
        case SPANGPU_SUPER_TONE:
            if (b->nb > 16)
                return fail(SPANGPU_ERR_UNSUPPORTED, "a super-tone bank of more than 16 bins cannot share a launch");
            M.kind[k] = (b->nb == 4)  ?  TONE_K_ST4  :  (b->nb == 8)  ?  TONE_K_ST8  :  (b->nb == 12)  ?  TONE_K_ST12  :  TONE_K_ST16;
            break;
        default:
            return fail(SPANGPU_ERR_UNSUPPORTED, "bank kind %d cannot share a launch", b->kind);
        }
        // the kernels of a shared launch write records, not digit bytes: a bank whose caller collects digit bytes
        // (spangpu_bank_set_digits_buffer / _ring) would hand on stale bytes and lose a slice of its ring
        if (b->ext_digits)
            return fail(SPANGPU_ERR_UNSUPPORTED, "a bank with a digits buffer cannot share a launch (it would get no digit bytes)");
        const long long stride = (strides  &&  strides[k] > 0)  ?  strides[k]  :  samples;
        const int maxb = (samples + b->block_len - 1)/b->block_len;
        const int rc = ensure_outputs(b, (maxb > 0)  ?  maxb  :  1);
        if (rc != SPANGPU_OK)
            return rc;
        if (b->ext_rec  &&  (size_t) maxb*b->n_ch*sizeof(uint32_t) > b->ext_rec_bytes)
            return fail(SPANGPU_ERR_BAD_ARG, "records buffer of bank %d too small for %d blocks", k, maxb);
        const int caught = cadence_catch_up(b);
        if (caught != SPANGPU_OK)
            return caught;
        fill_launch(M.bank[k], b, amps[k], stride, samples, SPANGPU_LAYOUT_CHANNEL_MAJOR, maxb, 0);
        all_fast = all_fast  &&  fast_eligible(M.bank[k]);
        b->last_maxb = maxb;
    b->launch_serial++;
        b->last_samples = samples;
    }
    // lanes per channel: the streaming kernels run one channel per lane unless two are asked for
    const int lpc = all_fast  ?  ((forced_lpc() == 2)  ?  2  :  1)  :  pick_lpc(total_ch);
    const int cpw = kWave/lpc;
    // Workgroups are dispatched in index order and a launch of this kind fills the chip more than once: the detectors with
    // the most work per channel go first, so that what runs last, when the chip drains, is the cheap kind (the super-tone
    // banks' workgroups at the end of the range left a tail of their own length; measured in tools/bench_paths.py mixed).
    auto cost = [](int kind) { return (kind == TONE_K_ST16)  ?  6  :  (kind == TONE_K_ST12)  ?  5  :  (kind == TONE_K_ST8)  ?  4
                                      :  (kind == TONE_K_DTMF)  ?  3  :  (kind == TONE_K_ST4)  ?  2  :  1; };
    for (int i = 1;  i < n_banks;  i++)
    {
        for (int k = i;  k > 0  &&  cost(M.kind[k]) > cost(M.kind[k - 1]);  k--)
        {
            const int kk = M.kind[k];
            M.kind[k] = M.kind[k - 1];
            M.kind[k - 1] = kk;
            const ToneLaunch tl = M.bank[k];
            M.bank[k] = M.bank[k - 1];
            M.bank[k - 1] = tl;
        }
    }
    int first = 0;
    for (int k = 0;  k < n_banks;  k++)
    {
        M.first[k] = first;
        const int waves = (M.bank[k].n_ch + cpw - 1)/cpw;
        first += (waves + kWavesPerBlock - 1)/kWavesPerBlock;
    }
    for (int k = n_banks;  k <= kMaxMulti;  k++)
        M.first[k] = first;
    M.n = n_banks;
    static_assert(kFastWPB == kWavesPerBlock, "the workgroup ranges above serve both kernel families");
    if (all_fast  &&  lpc == 1  &&  use_loader(total_ch))
        hipLaunchKernelGGL((tone_multi_fast_kernel<1, kRingLoader, true>), dim3(first), dim3(kWave*(kWavesPerBlock + 1)), 0, banks[0]->stream, M);
    else if (all_fast  &&  lpc == 1)
        hipLaunchKernelGGL((tone_multi_fast_kernel<1, kRingSelf, false>), dim3(first), dim3(kWave*kWavesPerBlock), 0, banks[0]->stream, M);
    else if (all_fast)
        hipLaunchKernelGGL((tone_multi_fast_kernel<2, kRingSelf, false>), dim3(first), dim3(kWave*kWavesPerBlock), 0, banks[0]->stream, M);
    else if (lpc == 2)
        hipLaunchKernelGGL(tone_multi_kernel<2>, dim3(first), dim3(kWave*kWavesPerBlock), 0, banks[0]->stream, M);
    else
        hipLaunchKernelGGL(tone_multi_kernel<1>, dim3(first), dim3(kWave*kWavesPerBlock), 0, banks[0]->stream, M);
    HIP_TRY(hipGetLastError());
    return 0;
}

int spangpu_bank_force_block(spangpu_bank_t *b)
{
    if (b == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "null bank");
    HIP_TRY(hipSetDevice(b->device));
    int rc = ensure_outputs(b, 1);
    if (rc != SPANGPU_OK)
        return rc;
    rc = launch_bank(b, nullptr, 0, 0, SPANGPU_LAYOUT_CHANNEL_MAJOR, 1, 1);
    if (rc < 0)
        return rc;
    b->last_maxb = 1;
    b->launch_serial++;
    b->last_samples = 0;
    return SPANGPU_OK;
}

int spangpu_bank_sync(spangpu_bank_t *b)
{
    if (b == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "null bank");
    HIP_TRY(hipStreamSynchronize(joined(b)));
    return SPANGPU_OK;
}

float spangpu_bank_last_kernel_ms(spangpu_bank_t *b)
{
    float ms = -1.0f;
    if (b == nullptr  ||  !b->ev_valid)
        return -1.0f;
    if (hipEventSynchronize(b->ev1) != hipSuccess)
        return -1.0f;
    if (hipEventElapsedTime(&ms, b->ev0, b->ev1) != hipSuccess)
        return -1.0f;
    return ms;
}

int spangpu_bank_blocks(spangpu_bank_t *b, spangpu_block_t *out, int max)
{
    if (b == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "null bank");
    if (b->last_maxb <= 0)
        return 0;
    HIP_TRY(hipSetDevice(b->device));
    const size_t n = (size_t) b->last_maxb*b->n_ch;
    HIP_TRY(hipMemcpyAsync(b->h_rec, b->cur_rec  ?  b->cur_rec  :  b->rec, n*sizeof(uint32_t), hipMemcpyDeviceToHost, joined(b)));
    if (b->rec_energy)
        HIP_TRY(hipMemcpyAsync(b->h_energy, b->rec_energy, n*sizeof(float), hipMemcpyDeviceToHost, joined(b)));
    if (b->rec_dur)
        HIP_TRY(hipMemcpyAsync(b->h_dur, b->rec_dur, n*sizeof(int32_t), hipMemcpyDeviceToHost, joined(b)));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    int count = 0;
    const bool bias = (b->kind == SPANGPU_SUPER_TONE);
    for (int ch = 0;  ch < b->n_ch;  ch++)
    {
        for (int k = 0;  k < b->last_maxb;  k++)
        {
            const size_t idx = (size_t) k*b->n_ch + ch;
            const uint32_t r = b->h_rec[idx];
            const int flags = (int) (r >> 16);
            if (!(flags & SPANGPU_BLK_VALID))
                continue;
            if (out  &&  count < max)
            {
                spangpu_block_t *o = &out[count];
                o->channel = ch;
                o->block = k;
                o->hit = (int) (r & 0xFF) - (bias  ?  1  :  0);
                o->code = (int) ((r >> 8) & 0xFF) - (bias  ?  1  :  0);
                o->flags = flags;
                o->duration = (b->h_dur  &&  (flags & SPANGPU_BLK_REPORT))  ?  b->h_dur[idx]  :  0;
                o->energy = (b->h_energy)  ?  b->h_energy[idx]  :  0.0f;
            }
            count++;
        }
    }
    return count;
}

// Have the following launches write their block records straight into a caller-owned device buffer (for example the
// send buffer of an RCCL gather: no copy between the kernel and the collective).  The buffer must hold
// ceil(samples/block)*n_channels words for the frames to come (2*n_channels covers 160-sample frames of every
// detector); NULL goes back to the bank's own buffer.  spangpu_bank_blocks() reads whichever was written last.
int spangpu_bank_set_records_buffer(spangpu_bank_t *b, void *dev_ptr, size_t bytes)
{
    if (b == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "null bank");
    b->ext_rec = (uint32_t *) dev_ptr;
    b->ext_rec_bytes = dev_ptr  ?  bytes  :  0;
    return SPANGPU_OK;
}

long long spangpu_bank_copy_records(spangpu_bank_t *b, void *dst_device, size_t dst_bytes)
{
    if (b == nullptr  ||  dst_device == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "null argument");
    const size_t bytes = (size_t) b->last_maxb*b->n_ch*sizeof(uint32_t);
    if (bytes > dst_bytes)
        return fail(SPANGPU_ERR_BAD_ARG, "record buffer too small: need %zu bytes", bytes);
    if (bytes == 0)
        return 0;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpyAsync(dst_device, b->cur_rec  ?  b->cur_rec  :  b->rec, bytes, hipMemcpyDeviceToDevice, joined(b)));
    return (long long) bytes;
}

// The launches from now on also write one byte per block and channel into a caller-owned device buffer (e.g. a slice of
// an RCCL send buffer): digits[block][channel] = the digit the block delivered, 0 = none (DTMF: the debouncer accepted a
// digit; Bell MF / R2 MF: the digit of a report), blocks a channel did not complete in the call = 0.  A quarter of the
// record words' bytes, no extra launch, no atomics.  A launch whose blocks do not fit `bytes` leaves the buffer alone.
// _ring: the buffer is n_slices slices of slice_bytes; successive launches fill successive slices, round and round (a
// reporting interval of n steps is then set up with one call instead of one per step).
int spangpu_bank_set_digits_ring(spangpu_bank_t *b, void *dev_ptr, size_t slice_bytes, int n_slices)
{
    if (b == nullptr  ||  (dev_ptr  &&  n_slices <= 0))
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (dev_ptr  &&  b->kind != SPANGPU_DTMF  &&  b->kind != SPANGPU_BELL_MF  &&  b->kind != SPANGPU_R2_MF)
        return fail(SPANGPU_ERR_UNSUPPORTED, "digit bytes exist for DTMF / Bell MF / R2 MF banks");
    b->ext_digits = (uint8_t *) dev_ptr;
    b->ext_digits_bytes = dev_ptr  ?  slice_bytes  :  0;
    b->ext_digits_slices = dev_ptr  ?  n_slices  :  0;
    b->ext_digits_next = 0;
    return SPANGPU_OK;
}

int spangpu_bank_set_digits_buffer(spangpu_bank_t *b, void *dev_ptr, size_t bytes)
{
    return spangpu_bank_set_digits_ring(b, dev_ptr, bytes, 1);
}

int spangpu_bank_digit_events(spangpu_bank_t *b, uint32_t *dst_device, int cap_entries)
{
    if (b == nullptr  ||  dst_device == nullptr  ||  cap_entries < 0)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (b->n_ch > (1 << 20)  ||  b->last_maxb > 16)
        return fail(SPANGPU_ERR_UNSUPPORTED, "digit events pack the channel in 20 bits and the block in 4");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemsetAsync(dst_device, 0, sizeof(uint32_t), joined(b)));
    if (b->last_maxb > 0)
    {
        hipLaunchKernelGGL(digit_events_kernel, dim3((b->n_ch + 255)/256), dim3(256), 0, joined(b),
                           (const uint32_t *) (b->cur_rec  ?  b->cur_rec  :  b->rec), b->n_ch, b->last_maxb, dst_device, cap_entries);
        HIP_TRY(hipGetLastError());
    }
    return SPANGPU_OK;
}

int spangpu_bank_trace(spangpu_bank_t *b, float *energies, size_t max_floats)
{
    if (b == nullptr  ||  energies == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "null argument");
    if (b->trace == nullptr)
        return fail(SPANGPU_ERR_STATE, "bank was created without trace");
    const size_t n = (size_t) b->last_maxb*(b->nb + 1)*b->n_ch;
    if (n > max_floats)
        return fail(SPANGPU_ERR_BAD_ARG, "trace buffer too small: need %zu floats", n);
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    HIP_TRY(hipMemcpy(energies, b->trace, n*sizeof(float), hipMemcpyDeviceToHost));
    return b->last_maxb;
}

int spangpu_bank_bins(const spangpu_bank_t *b)
{
    return b  ?  b->nb  :  SPANGPU_ERR_BAD_ARG;
}

// State import/export: fstate[nsf] in device order (v2[nb], v3[nb], [energy], [z350[2], z440[2]]),
// istate[4] = {current_sample, w0 byte 2, w0 byte 3, w1}.
int spangpu_bank_get_state(spangpu_bank_t *b, int channel, float *fstate, int max_f, int32_t *istate, int max_i)
{
    if (b == nullptr  ||  channel < 0  ||  channel >= b->n_ch  ||  max_f < b->nsf  ||  max_i < 4)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    HIP_TRY(hipMemcpy2D(fstate, sizeof(float), b->sf + channel, (size_t) b->n_ch*sizeof(float),
                        sizeof(float), b->nsf, hipMemcpyDeviceToHost));
    int32_t w[2];
    HIP_TRY(hipMemcpy2D(w, sizeof(int32_t), b->si + channel, (size_t) b->n_ch*sizeof(int32_t),
                        sizeof(int32_t), 2, hipMemcpyDeviceToHost));
    istate[0] = w[0] & 0xFFFF;
    istate[1] = (w[0] >> 16) & 0xFF;
    istate[2] = (w[0] >> 24) & 0xFF;
    istate[3] = w[1];
    return b->nsf;
}

int spangpu_bank_set_state(spangpu_bank_t *b, int channel, const float *fstate, int n_f, const int32_t *istate, int n_i)
{
    if (b == nullptr  ||  channel < 0  ||  channel >= b->n_ch  ||  n_f != b->nsf  ||  n_i != 4)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    HIP_TRY(hipMemcpy2D(b->sf + channel, (size_t) b->n_ch*sizeof(float), fstate, sizeof(float),
                        sizeof(float), b->nsf, hipMemcpyHostToDevice));
    int32_t w[2];
    w[0] = (istate[0] & 0xFFFF) | ((istate[1] & 0xFF) << 16) | ((int32_t) ((uint32_t) (istate[2] & 0xFF) << 24));
    w[1] = istate[3];
    HIP_TRY(hipMemcpy2D(b->si + channel, (size_t) b->n_ch*sizeof(int32_t), w, sizeof(int32_t),
                        sizeof(int32_t), 2, hipMemcpyHostToDevice));
    return SPANGPU_OK;
}

// ---- one DTMF channel in the reference's struct layout (include/spangpu_refstate.h; src/spandsp/private/dtmf.h:54-117) --
int spangpu_dtmf_import_state(spangpu_bank_t *b, int channel, const spangpu_ref_dtmf_rx_t *s)
{
    if (b == nullptr  ||  s == nullptr  ||  channel < 0  ||  channel >= b->n_ch  ||  b->kind != SPANGPU_DTMF)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    float f[21];
    int32_t w[4];
    for (int i = 0;  i < 4;  i++)
    {
        f[i] = s->row_out[i].v2;
        f[4 + i] = s->col_out[i].v2;
        f[8 + i] = s->row_out[i].v3;
        f[12 + i] = s->col_out[i].v3;
    }
    f[16] = s->energy;
    f[17] = s->z350[0];
    f[18] = s->z350[1];
    f[19] = s->z440[0];
    f[20] = s->z440[1];
    w[0] = s->current_sample;
    w[1] = s->last_hit;
    w[2] = s->in_digit;
    w[3] = s->duration;
    int rc = spangpu_bank_set_state(b, channel, f, 21, w, 4);
    if (rc != SPANGPU_OK)
        return rc;
    // the detector's own thresholds: the bank's table takes them as they stand
    const size_t n = (size_t) b->n_ch;
    const bool differs = s->threshold != b->threshold  ||  s->normal_twist != b->normal_twist  ||  s->reverse_twist != b->reverse_twist
                         ||  (s->filter_dialtone  ?  1  :  0) != (b->tp.filter_dialtone  ?  1  :  0);
    if (b->chan_parms == nullptr  &&  !differs)
        return SPANGPU_OK;
    if ((rc = ensure_chan_parms(b)) != SPANGPU_OK)
        return rc;
    float *h = b->h_chan_parms;
    const float on = s->filter_dialtone  ?  1.0f  :  0.0f;
    b->n_filter_on += (int) on - (int) h[3*n + channel];
    h[channel] = s->threshold;
    h[n + channel] = s->normal_twist;
    h[2*n + channel] = s->reverse_twist;
    h[3*n + channel] = on;
    for (int i = 0;  i < 4;  i++)
        HIP_TRY(hipMemcpy(b->chan_parms + (size_t) i*n + channel, &h[(size_t) i*n + channel], sizeof(float), hipMemcpyHostToDevice));
    return SPANGPU_OK;
}

int spangpu_dtmf_export_state(spangpu_bank_t *b, int channel, spangpu_ref_dtmf_rx_t *s)
{
    if (b == nullptr  ||  s == nullptr  ||  channel < 0  ||  channel >= b->n_ch  ||  b->kind != SPANGPU_DTMF)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    float f[2*kMaxBins + 8];
    int32_t w[4];
    const int rc = spangpu_bank_get_state(b, channel, f, 2*kMaxBins + 8, w, 4);
    if (rc < 0)
        return rc;
    const size_t n = (size_t) b->n_ch;
    for (int i = 0;  i < 4;  i++)
    {
        // dtmf_rx() drives its Goertzels through goertzel_samplex(): their own sample counters stay where
        // goertzel_init() left them (tone_detect.c:96-106), the block phase is the detector's current_sample
        s->row_out[i].v2 = f[i];
        s->col_out[i].v2 = f[4 + i];
        s->row_out[i].v3 = f[8 + i];
        s->col_out[i].v3 = f[12 + i];
        s->row_out[i].fac = b->fac[i];
        s->col_out[i].fac = b->fac[4 + i];
        s->row_out[i].samples = s->col_out[i].samples = 102;
        s->row_out[i].current_sample = s->col_out[i].current_sample = 0;
    }
    s->energy = f[16];
    s->z350[0] = f[17];
    s->z350[1] = f[18];
    s->z440[0] = f[19];
    s->z440[1] = f[20];
    s->current_sample = w[0];
    s->last_hit = (uint8_t) w[1];
    s->in_digit = (uint8_t) w[2];
    s->duration = w[3];
    if (b->chan_parms)
    {
        const float *h = b->h_chan_parms;
        s->threshold = h[channel];
        s->normal_twist = h[n + channel];
        s->reverse_twist = h[2*n + channel];
        s->filter_dialtone = (h[3*n + channel] != 0.0f);
    }
    else
    {
        s->threshold = b->threshold;
        s->normal_twist = b->normal_twist;
        s->reverse_twist = b->reverse_twist;
        s->filter_dialtone = (b->tp.filter_dialtone != 0);
    }
    return SPANGPU_OK;
}

// ---- Bell MF and MFC/R2 detectors in the reference's struct layout (src/spandsp/private/bell_r2_mf.h:62-116).  Both
// drive their six Goertzels through goertzel_samplex(), like dtmf_rx(): the block position is the detector's own
// current_sample, the Goertzels' counters stay where goertzel_init() left them.
int spangpu_bell_mf_import_state(spangpu_bank_t *b, int channel, const spangpu_ref_bell_mf_rx_t *s)
{
    if (b == nullptr  ||  s == nullptr  ||  channel < 0  ||  channel >= b->n_ch  ||  b->kind != SPANGPU_BELL_MF)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    float f[12];
    int32_t w[4];
    for (int i = 0;  i < 6;  i++)
    {
        f[i] = s->out[i].v2;
        f[6 + i] = s->out[i].v3;
    }
    // bell_r2_mf.c:629-661: hits[0] is the oldest of the five block results kept
    w[0] = s->current_sample;
    w[1] = s->hits[0];
    w[2] = s->hits[1];
    w[3] = (int32_t) ((uint32_t) s->hits[2] | ((uint32_t) s->hits[3] << 8) | ((uint32_t) s->hits[4] << 16));
    return spangpu_bank_set_state(b, channel, f, 12, w, 4);
}

int spangpu_bell_mf_export_state(spangpu_bank_t *b, int channel, spangpu_ref_bell_mf_rx_t *s)
{
    if (b == nullptr  ||  s == nullptr  ||  channel < 0  ||  channel >= b->n_ch  ||  b->kind != SPANGPU_BELL_MF)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    float f[2*kMaxBins + 8];
    int32_t w[4];
    const int rc = spangpu_bank_get_state(b, channel, f, 2*kMaxBins + 8, w, 4);
    if (rc < 0)
        return rc;
    for (int i = 0;  i < 6;  i++)
    {
        s->out[i].v2 = f[i];
        s->out[i].v3 = f[6 + i];
        s->out[i].fac = b->fac[i];
        s->out[i].samples = 120;
        s->out[i].current_sample = 0;
    }
    s->current_sample = w[0];
    s->hits[0] = (uint8_t) w[1];
    s->hits[1] = (uint8_t) w[2];
    s->hits[2] = (uint8_t) (w[3] & 0xFF);
    s->hits[3] = (uint8_t) ((w[3] >> 8) & 0xFF);
    s->hits[4] = (uint8_t) ((w[3] >> 16) & 0xFF);
    return SPANGPU_OK;
}

int spangpu_r2_mf_import_state(spangpu_bank_t *b, int channel, const spangpu_ref_r2_mf_rx_t *s)
{
    if (b == nullptr  ||  s == nullptr  ||  channel < 0  ||  channel >= b->n_ch  ||  b->kind != SPANGPU_R2_MF)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if ((s->fwd  ?  1  :  0) != (b->tp.r2_fwd  ?  1  :  0))
        return fail(SPANGPU_ERR_BAD_ARG, "the detector listens for the other direction's tones");
    float f[12];
    int32_t w[4];
    for (int i = 0;  i < 6;  i++)
    {
        f[i] = s->out[i].v2;
        f[6 + i] = s->out[i].v3;
    }
    w[0] = s->current_sample;
    w[1] = s->current_digit & 0xFF;
    w[2] = 0;
    w[3] = 0;
    return spangpu_bank_set_state(b, channel, f, 12, w, 4);
}

int spangpu_r2_mf_export_state(spangpu_bank_t *b, int channel, spangpu_ref_r2_mf_rx_t *s)
{
    if (b == nullptr  ||  s == nullptr  ||  channel < 0  ||  channel >= b->n_ch  ||  b->kind != SPANGPU_R2_MF)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if ((s->fwd  ?  1  :  0) != (b->tp.r2_fwd  ?  1  :  0))
        return fail(SPANGPU_ERR_BAD_ARG, "the detector listens for the other direction's tones");
    float f[2*kMaxBins + 8];
    int32_t w[4];
    const int rc = spangpu_bank_get_state(b, channel, f, 2*kMaxBins + 8, w, 4);
    if (rc < 0)
        return rc;
    for (int i = 0;  i < 6;  i++)
    {
        s->out[i].v2 = f[i];
        s->out[i].v3 = f[6 + i];
        s->out[i].fac = b->fac[i];
        s->out[i].samples = 133;
        s->out[i].current_sample = 0;
    }
    s->current_sample = w[0];
    s->current_digit = w[1];
    return SPANGPU_OK;
}

int spangpu_bank_reset_channel(spangpu_bank_t *b, int channel, int fillin_only)
{
    if (b == nullptr  ||  channel < 0  ||  channel >= b->n_ch)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    float f[2*kMaxBins + 8];
    int32_t w[4];
    int rc = spangpu_bank_get_state(b, channel, f, 2*kMaxBins + 8, w, 4);
    if (rc < 0)
        return rc;
    // Goertzel states, block energy and block position always restart (dtmf.c:363-379);
    // a full reset also clears the filters and the hit history (dtmf.c:447-504).
    const int nclear = (fillin_only)  ?  (2*b->nb + ((b->nsf > 2*b->nb)  ?  1  :  0))  :  b->nsf;
    for (int i = 0;  i < nclear;  i++)
        f[i] = 0.0f;
    w[0] = 0;
    if (!fillin_only)
        w[1] = w[2] = w[3] = 0;
    return spangpu_bank_set_state(b, channel, f, b->nsf, w, 4);
}

}   // extern "C"

static void cadence_args(const spangpu_bank_s *b, const Cadence *c, CadenceArgs &A, int which)
{
    A.first = c->d_first;
    A.elem = c->d_elem;
    A.state = c->d_state;
    A.ev = c->d_ev;
    A.count = c->d_count;
    A.list = c->d_list;
    A.list_cap = (uint32_t) ((size_t) c->slots_cap*b->n_ch);
    A.n_tones = c->n_tones;
    A.segments = c->segments;
    A.which = which;
    A.n_elems = c->n_elems;
}

// Event buffers for launches of up to `slots` slots per channel.
static int cadence_event_room(spangpu_bank_t *b, int slots)
{
    Cadence *c = b->cad;
    if (slots <= c->slots_cap)
        return SPANGPU_OK;
    HIP_TRY(hipStreamSynchronize(joined(b)));
    if (c->d_ev) (void) hipFree(c->d_ev);
    if (c->h_ev) (void) hipHostFree(c->h_ev);
    if (c->d_list) (void) hipFree(c->d_list);
    if (c->h_list) (void) hipHostFree(c->h_list);
    c->d_ev = nullptr;
    c->h_ev = nullptr;
    c->d_list = nullptr;
    c->h_list = nullptr;
    c->slots_cap = 0;
    HIP_TRY(hipMalloc(&c->d_ev, (size_t) slots*b->n_ch*2*sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc(&c->h_ev, (size_t) slots*b->n_ch*2*sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->d_list, ((size_t) slots*b->n_ch*3 + 2)*sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc(&c->h_list, ((size_t) slots*b->n_ch*3 + 2)*sizeof(uint32_t)));
    HIP_TRY(hipMemset(c->d_list, 0, 2*sizeof(uint32_t)));
    c->which = 0;
    c->slots_cap = slots;
    return SPANGPU_OK;
}

extern "C" {

// ---- super-tone cadences (src/super_tone_rx.c:164-228, :364-448) on the device ------------------------------------------
int spangpu_bank_set_cadences(spangpu_bank_t *b, const int32_t *tone_elems, int n_tones, const spangpu_cadence_elem_t *elems,
                              int want_segments)
{
    if (b == nullptr  ||  n_tones < 0  ||  (n_tones > 0  &&  (tone_elems == nullptr  ||  elems == nullptr)))
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (b->kind != SPANGPU_SUPER_TONE)
        return fail(SPANGPU_ERR_UNSUPPORTED, "cadences belong to a super-tone bank");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    std::vector<int32_t> first(n_tones + 1, 0);
    for (int t = 0;  t < n_tones;  t++)
    {
        if (tone_elems[t] < 0)
            return fail(SPANGPU_ERR_BAD_ARG, "tone %d has a negative element count", t);
        first[t + 1] = first[t] + tone_elems[t];
    }
    const int n_elems = first[n_tones];
    std::vector<int4> el(n_elems > 0  ?  n_elems  :  1);
    for (int i = 0;  i < n_elems;  i++)
    {
        if (elems[i].f1 < -1  ||  elems[i].f1 >= 255  ||  elems[i].f2 < -1  ||  elems[i].f2 >= 255  ||  elems[i].min_ms < 0
            ||  elems[i].max_ms < 0  ||  elems[i].min_ms > 0x7FFFFFFF/8)
            return fail(SPANGPU_ERR_BAD_ARG, "element %d: bins are -1..254, times are milliseconds >= 0", i);
        // milliseconds are kept in samples, 0 = no upper limit (super_tone_rx.c:157-158); a run of b blocks is 128 b samples
        // long, so the window in blocks is ceil(lo/128) .. floor(hi/128)
        const long long lo = 8LL*elems[i].min_ms;
        const long long hi = (elems[i].max_ms == 0)  ?  0x7FFFFFFFLL  :  8LL*elems[i].max_ms;
        el[i] = make_int4(cad_pair(elems[i].f1, elems[i].f2), (int) ((lo + 127)/128), (int) ((hi > 0x7FFFFFFFLL  ?  0x7FFFFFFFLL  :  hi)/128), 0);
    }
    Cadence *c = b->cad;
    const bool fresh = (c == nullptr);
    if (fresh)
    {
        if ((c = (Cadence *) calloc(1, sizeof(Cadence))) == nullptr)
            return fail(SPANGPU_ERR_NO_MEMORY, "out of memory");
        if (hipMalloc(&c->d_state, (size_t) kCadWords*b->n_ch*sizeof(int32_t)) != hipSuccess
            ||  hipMalloc(&c->d_count, (size_t) b->n_ch*sizeof(int32_t)) != hipSuccess
            ||  hipHostMalloc(&c->h_count, (size_t) b->n_ch*sizeof(int32_t)) != hipSuccess)
        {
            cadence_free(c);
            return fail(SPANGPU_ERR_NO_MEMORY, "out of device memory for the cadence state");
        }
        hipLaunchKernelGGL(cadence_init_kernel, dim3((b->n_ch + 255)/256), dim3(256), 0, joined(b), c->d_state, b->n_ch, 0, b->n_ch);
        c->done_serial = b->launch_serial;      // what was received before now is not matched
        b->cad = c;
        const int rc = cadence_event_room(b, kCadSlotsPerBlock*((b->maxb_cap > 2)  ?  b->maxb_cap  :  2));
        if (rc != SPANGPU_OK)
            return rc;
    }
    // the new tables are made and filled first and swapped in only when they are complete: a failure leaves the bank on
    // its old set (a live bank's next launch reads first[] and elem[] on the device)
    {
        int32_t *n_first = nullptr;
        int4 *n_elem = nullptr;
        if (hipMalloc(&n_first, first.size()*sizeof(int32_t)) != hipSuccess
            ||  hipMalloc(&n_elem, el.size()*sizeof(int4)) != hipSuccess
            ||  hipMemcpy(n_first, first.data(), first.size()*sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess
            ||  hipMemcpy(n_elem, el.data(), el.size()*sizeof(int4), hipMemcpyHostToDevice) != hipSuccess)
        {
            if (n_first) (void) hipFree(n_first);
            if (n_elem) (void) hipFree(n_elem);
            return fail(SPANGPU_ERR_NO_MEMORY, "out of device memory for the cadence tables");
        }
        HIP_TRY(hipStreamSynchronize(joined(b)));        // no launch still reads the old ones
        if (c->d_first) (void) hipFree(c->d_first);
        if (c->d_elem) (void) hipFree(c->d_elem);
        c->d_first = n_first;
        c->d_elem = n_elem;
    }
    if (!fresh)
    {
        // the tone numbers of the old set mean nothing in the new one: nobody is following a tone (the run histories stay)
        // (-2, not -1: "none, and look at every cadence at the next block" -- the channel may be in the middle of a run that
        // one of the new cadences ends with, see cadence_dev.hpp)
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) (c->d_state + (size_t) 2*b->n_ch), (int) 0xFFFFFFFEu, (size_t) b->n_ch, joined(b)));
        HIP_TRY(hipMemsetAsync(c->d_state + (size_t) 3*b->n_ch, 0, (size_t) b->n_ch*sizeof(int32_t), joined(b)));
    }
    c->n_tones = n_tones;
    c->n_elems = n_elems;
    for (int t = 0;  t < n_tones  &&  t < kCadLdsTones;  t++)
        c->tone_len[t] = first[t + 1] - first[t];
    c->segments = want_segments  ?  1  :  0;
    HIP_TRY(hipStreamSynchronize(joined(b)));
    return SPANGPU_OK;
}

int spangpu_bank_cadence_run(spangpu_bank_t *b)
{
    if (b == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "null bank");
    Cadence *c = b->cad;
    if (c == nullptr)
        return fail(SPANGPU_ERR_STATE, "no cadences were given to this bank (spangpu_bank_set_cadences)");
    if (c->done_serial == b->launch_serial)
        return c->last_slots;
    HIP_TRY(hipSetDevice(b->device));
    const int slots = kCadSlotsPerBlock*b->last_maxb;
    if (c->fused)
    {
        // the detector launch did it
        c->fused = false;
        c->done_serial = b->launch_serial;
        c->last_slots = slots;
        return slots;
    }
    const int rc = cadence_event_room(b, slots);
    if (rc != SPANGPU_OK)
        return rc;
    if (b->last_maxb > 0)
    {
        CadenceArgs A;
        c->which ^= 1;
        c->list_due = false;
        cadence_args(b, c, A, c->which);
        hipLaunchKernelGGL(cadence_kernel, dim3((b->n_ch + 255)/256), dim3(256), 0, joined(b),
                           (const uint32_t *) (b->cur_rec  ?  b->cur_rec  :  b->rec), b->n_ch, b->last_maxb, A);
        HIP_TRY(hipGetLastError());
    }
    else
    {
        HIP_TRY(hipMemsetAsync(c->d_count, 0, (size_t) b->n_ch*sizeof(int32_t), joined(b)));
    }
    c->done_serial = b->launch_serial;
    c->last_slots = slots;
    return slots;
}

int spangpu_bank_cadence_events(spangpu_bank_t *b, const uint32_t **events, const int32_t **counts)
{
    const int slots = spangpu_bank_cadence_run(b);
    if (slots < 0)
        return slots;
    Cadence *c = b->cad;
    HIP_TRY(hipMemcpyAsync(c->h_count, c->d_count, (size_t) b->n_ch*sizeof(int32_t), hipMemcpyDeviceToHost, joined(b)));
    if (slots > 0)
        HIP_TRY(hipMemcpyAsync(c->h_ev, c->d_ev, (size_t) slots*b->n_ch*2*sizeof(uint32_t), hipMemcpyDeviceToHost, joined(b)));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    if (events)
        *events = c->h_ev;
    if (counts)
        *counts = c->h_count;
    return slots;
}

int spangpu_bank_cadence_list(spangpu_bank_t *b, const uint32_t **list)
{
    const int slots = spangpu_bank_cadence_run(b);
    if (slots < 0)
        return slots;
    Cadence *c = b->cad;
    if (slots == 0  ||  c->d_list == nullptr  ||  b->last_maxb <= 0)
        return 0;
    if (c->list_due)
    {
        CadenceArgs A;
        c->which ^= 1;
        c->list_due = false;
        cadence_args(b, c, A, c->which);
        hipLaunchKernelGGL(cadence_list_kernel, dim3((b->n_ch + 255)/256), dim3(256), 0, joined(b), b->n_ch, A);
        HIP_TRY(hipGetLastError());
    }
    // as a rule a tick has few reports: one small copy brings the counters and the first of them
    const size_t cap = (size_t) c->slots_cap*b->n_ch;
    const size_t first = (cap < 4096)  ?  cap  :  4096;
    HIP_TRY(hipMemcpyAsync(c->h_list, c->d_list, (2 + 3*first)*sizeof(uint32_t), hipMemcpyDeviceToHost, joined(b)));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    const size_t n = c->h_list[c->which];
    if (n > cap)
        return fail(SPANGPU_ERR_STATE, "cadence event list overran its buffer");
    if (n > first)
    {
        HIP_TRY(hipMemcpyAsync(c->h_list + 2 + 3*first, c->d_list + 2 + 3*first, 3*(n - first)*sizeof(uint32_t), hipMemcpyDeviceToHost, joined(b)));
        HIP_TRY(hipStreamSynchronize(joined(b)));
    }
    if (list)
        *list = c->h_list + 2;
    return (int) n;
}

int spangpu_bank_cadence_device(spangpu_bank_t *b, const uint32_t **events_dev, const int32_t **counts_dev)
{
    if (b == nullptr  ||  b->cad == nullptr)
        return fail(SPANGPU_ERR_STATE, "no cadences were given to this bank");
    if (events_dev)
        *events_dev = b->cad->d_ev;
    if (counts_dev)
        *counts_dev = b->cad->d_count;
    return b->cad->last_slots;
}

int spangpu_bank_cadence_reset(spangpu_bank_t *b, int channel)
{
    if (b == nullptr  ||  b->cad == nullptr)
        return fail(SPANGPU_ERR_STATE, "no cadences were given to this bank");
    if (channel < -1  ||  channel >= b->n_ch)
        return fail(SPANGPU_ERR_BAD_ARG, "bad channel");
    HIP_TRY(hipSetDevice(b->device));
    const int first = (channel < 0)  ?  0  :  channel;
    const int n = (channel < 0)  ?  b->n_ch  :  1;
    hipLaunchKernelGGL(cadence_init_kernel, dim3((n + 255)/256), dim3(256), 0, joined(b), b->cad->d_state, b->n_ch, first, n);
    HIP_TRY(hipGetLastError());
    return SPANGPU_OK;
}

int spangpu_bank_cadence_state_words(void)
{
    return kCadWords;
}

int spangpu_bank_cadence_get_state(spangpu_bank_t *b, int channel, int32_t *words)
{
    if (b == nullptr  ||  b->cad == nullptr)
        return fail(SPANGPU_ERR_STATE, "no cadences were given to this bank");
    if (channel < 0  ||  channel >= b->n_ch  ||  words == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    HIP_TRY(hipMemcpy2D(words, sizeof(int32_t), b->cad->d_state + channel, (size_t) b->n_ch*sizeof(int32_t), sizeof(int32_t), kCadWords,
                        hipMemcpyDeviceToHost));
    if (words[2] < -1)
        words[2] = -1;          // -2 is the engine's own "none, and every cadence is looked at once at the next block" (cadence_dev.hpp)
    return kCadWords;
}

int spangpu_bank_cadence_set_state(spangpu_bank_t *b, int channel, const int32_t *words)
{
    if (b == nullptr  ||  b->cad == nullptr)
        return fail(SPANGPU_ERR_STATE, "no cadences were given to this bank");
    if (channel < 0  ||  channel >= b->n_ch  ||  words == nullptr)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (words[2] < -1  ||  words[2] >= b->cad->n_tones  ||  words[3] < 0  ||  words[3] > 255)
        return fail(SPANGPU_ERR_BAD_ARG, "state words out of range");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(joined(b)));
    int32_t w[kCadWords];
    memcpy(w, words, sizeof(w));
    if (w[2] == -1)
        w[2] = -2;              // a state from outside may sit in the middle of a run a cadence ends with: look at all of them once
    // the elements of the followed cadence that have gone by are counted modulo its length on the device (head of the ring: 0)
    if (w[2] >= 0  &&  w[2] < kCadLdsTones  &&  b->cad->tone_len[w[2]] > 0)
        w[3] = w[3]%b->cad->tone_len[w[2]];
    HIP_TRY(hipMemcpy2D(b->cad->d_state + channel, (size_t) b->n_ch*sizeof(int32_t), w, sizeof(int32_t), sizeof(int32_t), kCadWords,
                        hipMemcpyHostToDevice));
    return SPANGPU_OK;
}

}   // extern "C"

// ---- test hook: the two block-end decisions of a detector side by side -----------------------------------------------
// Det::decide() (the reference's scan, every option) and Det::decide_plain() (the production case of the streaming kernels)
// on caller-made energies and history words -- ties among the energies, zeros, values either side of every threshold: cases a
// synthesised signal reaches rarely or never (tests/test_tone_gpu.py: test_lean_block_end_*).  kind: SPANGPU_DTMF, _BELL_MF, _R2_MF.
// e: [n][8] floats (the MF detectors use the first six), energy / w0 / w1: [n]; out: [6][n] words: record, w0, w1 of decide(), then
// of decide_plain().  Host pointers.  Not part of the drop-in boundary.
template <class Det>
__global__ void debug_decide_kernel(const float *e8, const float *energy, const uint32_t *w0in, const int32_t *w1in, uint32_t *out, int n, const ToneLaunch L)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float e[Det::NB];
#pragma unroll
    for (int k = 0;  k < Det::NB;  k++)
        e[k] = e8[(size_t) i*8 + k];
    Det det;
    det.load_extra(L, 0);
    float en = energy[i];
    uint32_t w0 = w0in[i] & 0xFFFF0000u;
    int32_t w1 = w1in[i];
    out[i] = det.decide(L, e, en, w0, w1, 0, 0, false);
    out[(size_t) n + i] = w0;
    out[(size_t) 2*n + i] = (uint32_t) w1;
    en = energy[i];
    w0 = w0in[i] & 0xFFFF0000u;
    w1 = w1in[i];
    if constexpr (Det::kLean)
        out[(size_t) 3*n + i] = det.decide_plain(L, e, en, w0, w1);
    out[(size_t) 4*n + i] = w0;
    out[(size_t) 5*n + i] = (uint32_t) w1;
}

extern "C" __attribute__((visibility("default")))
int spangpu_debug_decide(int kind, const float *e8, const float *energy, const uint32_t *w0, const int32_t *w1, uint32_t *out, int n)
{
    if (e8 == nullptr  ||  energy == nullptr  ||  w0 == nullptr  ||  w1 == nullptr  ||  out == nullptr  ||  n <= 0)
        return fail(SPANGPU_ERR_BAD_ARG, "bad arguments");
    ToneLaunch L;
    memset(&L, 0, sizeof(L));
    L.n_ch = 1;
    L.threshold = 171029200.0f;         // dtmf.c:104-110
    L.normal_twist = 6.309f;
    L.reverse_twist = 2.512f;
    float *d_e = nullptr, *d_en = nullptr;
    uint32_t *d_w0 = nullptr, *d_out = nullptr;
    int32_t *d_w1 = nullptr;
    HIP_TRY(hipMalloc(&d_e, (size_t) n*8*sizeof(float)));
    HIP_TRY(hipMalloc(&d_en, (size_t) n*sizeof(float)));
    HIP_TRY(hipMalloc(&d_w0, (size_t) n*4));
    HIP_TRY(hipMalloc(&d_w1, (size_t) n*4));
    HIP_TRY(hipMalloc(&d_out, (size_t) n*6*4));
    HIP_TRY(hipMemcpy(d_e, e8, (size_t) n*8*sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_en, energy, (size_t) n*sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_w0, w0, (size_t) n*4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_w1, w1, (size_t) n*4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_out, 0, (size_t) n*6*4));
    const dim3 grid((n + 255)/256), block(256);
    if (kind == SPANGPU_DTMF)
        hipLaunchKernelGGL(debug_decide_kernel<DtmfDet<false>>, grid, block, 0, 0, d_e, d_en, d_w0, d_w1, d_out, n, L);
    else if (kind == SPANGPU_BELL_MF)
        hipLaunchKernelGGL(debug_decide_kernel<BellMfDet>, grid, block, 0, 0, d_e, d_en, d_w0, d_w1, d_out, n, L);
    else if (kind == SPANGPU_R2_MF)
        hipLaunchKernelGGL(debug_decide_kernel<R2MfDet>, grid, block, 0, 0, d_e, d_en, d_w0, d_w1, d_out, n, L);
    else
        return fail(SPANGPU_ERR_UNSUPPORTED, "no lean block end for detector kind %d", kind);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out, (size_t) n*6*4, hipMemcpyDeviceToHost));
    (void) hipFree(d_e); (void) hipFree(d_en); (void) hipFree(d_w0); (void) hipFree(d_w1); (void) hipFree(d_out);
    return SPANGPU_OK;
}
