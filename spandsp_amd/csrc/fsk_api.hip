// fsk_api.hip -- C ABI of the FSK receiver banks (include/spangpu.h, "FSK receiver banks"): batched
// fsk_rx().  Device code: fsk_dev.hpp.  No CPU implementation of the receive path exists behind these
// entry points; the control-plane calls (restart, cutoff, frame parameters, fill-in) edit one channel's
// state words on the host, as the reference's own functions edit one object.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"
#include "fsk_dev.hpp"

using namespace spg;

extern "C" int spangpu_set_error(int code, const char *msg);

#define FSK_TRY(expr)                                                                       \
    do                                                                                      \
    {                                                                                       \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
        {                                                                                   \
            char m_[256];                                                                   \
            snprintf(m_, sizeof(m_), "%s failed: %s", #expr, hipGetErrorString(e_));        \
            return spangpu_set_error(SPANGPU_ERR_HIP, m_);                                  \
        }                                                                                   \
    }                                                                                       \
    while (0)

namespace spg
{

__global__ __launch_bounds__(64) void fsk_bank_kernel(const FskLaunch L)
{
    extern __shared__ int32_t win[];        // [4*span][64]
    __shared__ uint32_t wave[kFskWave];     // a separate object, so that table reads can move across window writes
    const int lane = threadIdx.x;
    const int ch = blockIdx.x*64 + lane;
    const bool live = ch < L.n_ch;
    const size_t n = (size_t) L.n_ch;
    const int span = L.span;

    fsk_fill_wave(wave, L.quarter, lane, 64);
    // (a lane past the end of the bank reads channel 0's words and stops after the barrier)
    int32_t *st = L.st + (live  ?  ch  :  0);
    fsk_load_window(win, st + (size_t) kFskScalars*n, n, span, lane);
    __syncthreads();
    if (!live)
        return;

    FskRegs r;
    fsk_load_regs(r, st, n);
    int16_t *ev = L.events + (size_t) ch*L.ev_cap;
    const int ev_cap = L.ev_cap;
    auto emit = [&](int v) __attribute__((always_inline))
    {
        if (r.n_ev < ev_cap)
            ev[r.n_ev] = (int16_t) v;
        r.n_ev++;
    };
    const int mylen = L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;
    auto frame = [&](auto aligned, auto framed) __attribute__((always_inline))
    {
        FskRow<decltype(aligned)::value> row;
        fsk_row_begin(row, L.pcm + (size_t) ch*L.stride, mylen);
        for (int base = 0;  base < L.samples;  base += 8)
        {
            const int todo = max(0, min(8, mylen - base));      // per lane when the call carries per-channel lengths
            int32_t a[8];
            int32_t c0[8];
            int32_t q0[8];
            int32_t c1[8];
            int32_t q1[8];
            fsk_row_block(row, base, todo, a);
            fsk_block_lookups(r, wave, todo, c0, q0, c1, q1);
            // (a whole block in every lane is the common case: no per-sample guard, whose bodies the compiler moves out of line)
            if (__builtin_expect(__all(todo == 8), 1))
            {
#pragma unroll
                for (int k = 0;  k < 8;  k++)
                    fsk_step<decltype(framed)::value>(r, win, lane, span, a[k], c0[k], q0[k], c1[k], q1[k], emit);
            }
            else
            {
#pragma unroll
                for (int k = 0;  k < 8;  k++)
                {
                    if (k < todo)
                        fsk_step<true>(r, win, lane, span, a[k], c0[k], q0[k], c1[k], q1[k], emit);
                }
            }
        }
    };
    // (unaligned rows are the rare case: one copy, the general one)
    if (!L.vec)
        frame(std::false_type{}, std::true_type{});
    else if (__any(r.framing == 2))
        frame(std::true_type{}, std::true_type{});
    else
        frame(std::true_type{}, std::false_type{});
    fsk_store_regs(r, st, n);
    fsk_store_window(win, st + (size_t) kFskScalars*n, n, span, lane);
    L.ev_count[ch] = r.n_ev;
}

// The same receiver as two waves per 64 channels (fsk_dev.hpp, "A receiver over two waves"): wave 0 is the signal
// side, wave 1 the bit side, one block behind.
__global__ __launch_bounds__(128) void fsk_pair_kernel(const FskLaunch L)
{
    extern __shared__ int32_t win[];        // [4*span][64], then the two message buffers [2][kFskMsgWords][64]
    __shared__ uint32_t wave[kFskWave];
    const int lane = threadIdx.x & 63;
    const int side = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int ch = blockIdx.x*64 + lane;
    const bool live = ch < L.n_ch;
    const size_t n = (size_t) L.n_ch;
    const int span = L.span;
    int32_t *msg = win + 4*span*64;

    fsk_fill_wave(wave, L.quarter, threadIdx.x, 128);
    int32_t *st = L.st + (live  ?  ch  :  0);
    fsk_load_window_half(win, st + (size_t) kFskScalars*n, n, span, lane, side);
    __syncthreads();

    const int mylen = !live  ?  0  :  L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;
    const int n_blk = (L.samples + 7) >> 3;
    const int16_t *pcm_row = L.pcm + (size_t) (live  ?  ch  :  0)*L.stride;
    if (side == 0)
    {
        FskSigSide s;
        fsk_sig_load(s, st, n);
        auto frame = [&](auto aligned) __attribute__((always_inline))
        {
            FskRow<decltype(aligned)::value> row;
            fsk_row_begin(row, pcm_row, mylen);
            for (int blk = 0;  blk <= n_blk;  blk++)
            {
                if (blk < n_blk)
                    fsk_sig_block(s, win, wave, msg + (blk & 1)*kFskMsgWords*64, lane, span, row, blk*8, max(0, min(8, mylen - blk*8)));
                __syncthreads();
            }
        };
        if (L.vec)
            frame(std::true_type{});
        else
            frame(std::false_type{});
        if (live)
            fsk_sig_store(s, st, n);
    }
    else
    {
        FskBitSide t;
        fsk_bit_load(t, st, n);
        int16_t *ev = L.events + (size_t) (live  ?  ch  :  0)*L.ev_cap;
        const int ev_cap = L.ev_cap;
        auto emit = [&](int v) __attribute__((always_inline))
        {
            if (t.n_ev < ev_cap)
                ev[t.n_ev] = (int16_t) v;
            t.n_ev++;
        };
        auto frame = [&](auto aligned, auto framed) __attribute__((always_inline))
        {
            FskRow<decltype(aligned)::value> row;
            fsk_row_begin(row, pcm_row, mylen);
            for (int blk = 0;  blk <= n_blk;  blk++)
            {
                if (blk > 0)
                    fsk_bit_block<decltype(aligned)::value, decltype(framed)::value>(t, win, wave, msg + ((blk - 1) & 1)*kFskMsgWords*64, lane, span, row, (blk - 1)*8,
                                  max(0, min(8, mylen - (blk - 1)*8)), emit);
                __syncthreads();
            }
        };
        if (!L.vec)
            frame(std::false_type{}, std::true_type{});
        else if (__any(live  &&  t.b.framing == 2))
            frame(std::true_type{}, std::true_type{});
        else
            frame(std::true_type{}, std::false_type{});
        if (live)
        {
            fsk_bit_store(t, st, n);
            L.ev_count[ch] = t.n_ev;
        }
    }
    if (live)
        fsk_store_window_half(win, st + (size_t) kFskScalars*n, n, span, lane, side);
}

}   // namespace spg

// 0 = the library's choice (two waves per 64 channels), 1 = the whole receiver in one wave, 2 = two waves
static int g_fsk_waves = 0;

extern "C" int spangpu_tune_fsk_waves(int waves)
{
    if (waves < 0  ||  waves > 2)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "0 (the library's choice), 1 or 2 waves per 64 channels");
    g_fsk_waves = waves;
    return SPANGPU_OK;
}

// (for the other bank families the knob covers; not part of the ABI)
extern "C" __attribute__((visibility("hidden"))) int spangpu_fsk_waves_choice(void)
{
    return g_fsk_waves;
}

struct spangpu_fsk_s
{
    const int32_t *next_lens;   // per-channel lengths of the call being prepared (device), or NULL
    int32_t *d_lens;            // [n_ch], device
    int32_t *h_lens;            // [n_ch], pinned
    int device;
    int n_ch;
    int span;
    int words;                  // per channel
    spangpu_fsk_spec_t spec;
    hipStream_t stream;
    bool own_stream;
    int32_t *st;
    int16_t *quarter;
    int16_t *d_pcm;             // staging for host-resident frames
    size_t pcm_cap;
    int16_t *events;
    int32_t *ev_count;
    int ev_cap;
    int last_cap;
    int16_t *h_events;
    int32_t *h_count;
    size_t h_events_cap;
};

// preset_fsk_specs[], src/fsk.c:60-155: freq_zero, freq_one, tx_level, min_level, baud_rate x 100
static const spangpu_fsk_spec_t k_presets[11] =
{
    {1080 + 100, 1080 - 100, -14, -30, 300*100},        // V21 ch 1
    {1750 + 100, 1750 - 100, -14, -30, 300*100},        // V21 ch 2
    {1700 + 400, 1700 - 400, -14, -30, 1200*100},       // V23 ch 1
    {420 + 30, 420 - 30, -14, -30, 75*100},             // V23 ch 2
    {1170 - 100, 1170 + 100, -14, -30, 300*100},        // Bell103 ch 1
    {2125 - 100, 2125 + 100, -14, -30, 300*100},        // Bell103 ch 2
    {1700 + 500, 1700 - 500, -14, -30, 1200*100},       // Bell202
    {1600 + 200, 1600 - 200, -14, -30, 4545},           // Weitbrecht 45.45
    {1600 + 200, 1600 - 200, -14, -30, 50*100},         // Weitbrecht 50
    {1600 + 200, 1600 - 200, -14, -30, 4760},           // Weitbrecht 47.6
    {1080 + 100, 1080 - 100, -14, -30, 110*100}         // V21 (110bps) ch 1
};

static int32_t power_level_dbm0(float level)
{
    // power_meter_level_dbm0(), power_meter.c:82-92 (DBM0_MAX_POWER = 3.14 + 3.02)
    level -= (3.14f + 3.02f);
    if (level > 0.0)
        level = 0.0;
    return (int32_t) (powf(10.0f, level/10.0f)*(32767.0f*32767.0f));
}

static void cutoff_words(int32_t *w, float cutoff)
{
    // fsk_rx_set_signal_cutoff(), fsk.c:270-276
    w[FS_ON_POWER] = power_level_dbm0(cutoff + 2.5f - 5.3f);
    w[FS_OFF_POWER] = power_level_dbm0(cutoff - 2.5f - 5.3f);
}

static void frame_words(int32_t *w, int data_bits, int parity, int stop_bits)
{
    // fsk_rx_set_frame_parameters(), fsk.c:300-316
    if (w[FS_FRAMING] != SPANGPU_FSK_FRAME_MODE_FRAMED)
        return;
    w[FS_DATA_BITS] = data_bits;
    w[FS_PARITY] = parity;
    w[FS_STOP_BITS] = stop_bits;
    w[FS_TOTAL_BITS] = data_bits + ((parity != 0)  ?  1  :  0);
}

static int span_of(const spangpu_fsk_spec_t *spec)
{
    int span = kFskRateX100/spec->baud_rate;
    return (span > kFskMaxWindow)  ?  kFskMaxWindow  :  span;
}

static void restart_words(int32_t *w, const spangpu_fsk_spec_t *spec, int framing_mode)
{
    // fsk_rx_restart(), fsk.c:660-720.  The window and the running dot products are left as they are,
    // as the reference leaves them.
    w[FS_BAUD_RATE] = spec->baud_rate;
    w[FS_FRAMING] = framing_mode;
    if (framing_mode == SPANGPU_FSK_FRAME_MODE_FRAMED)
        frame_words(w, 8, 0, 1);
    cutoff_words(w, (float) spec->min_level);
    w[FS_RATE0] = (int32_t) ((float) spec->freq_zero*65536.0f*65536.0f/8000);      // dds_phase_rate(), dds_int.c
    w[FS_RATE1] = (int32_t) ((float) spec->freq_one*65536.0f*65536.0f/8000);
    w[FS_ACC0] = 0;
    w[FS_ACC1] = 0;
    w[FS_LAST_SAMPLE] = 0;
    w[FS_SPAN] = span_of(spec);
    int shift = 0;
    for (int chop = w[FS_SPAN];  chop != 0;  chop >>= 1)
        shift++;
    w[FS_SHIFT] = shift;
    w[FS_BAUD_PHASE] = 0;
    w[FS_FRAME_POS] = -2;
    w[FS_FRAME] = 0;
    w[FS_LAST_BIT] = 0;
    w[FS_POWER] = 0;
    w[FS_SIGNAL_PRESENT] = 0;
}

static int read_words(spangpu_fsk_s *f, int ch, int32_t *w)
{
    FSK_TRY(hipSetDevice(f->device));
    FSK_TRY(hipMemcpy2DAsync(w, sizeof(int32_t), f->st + ch, (size_t) f->n_ch*sizeof(int32_t), sizeof(int32_t), f->words,
                             hipMemcpyDeviceToHost, f->stream));
    FSK_TRY(hipStreamSynchronize(f->stream));
    return SPANGPU_OK;
}

static int write_words(spangpu_fsk_s *f, int ch, const int32_t *w)
{
    FSK_TRY(hipSetDevice(f->device));
    FSK_TRY(hipMemcpy2DAsync(f->st + ch, (size_t) f->n_ch*sizeof(int32_t), w, sizeof(int32_t), sizeof(int32_t), f->words,
                             hipMemcpyHostToDevice, f->stream));
    FSK_TRY(hipStreamSynchronize(f->stream));
    return SPANGPU_OK;
}

extern "C" {

int spangpu_fsk_preset(int which, spangpu_fsk_spec_t *spec)
{
    if (which < 0  ||  which > SPANGPU_FSK_V21CH1_110  ||  spec == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "no such FSK preset");
    *spec = k_presets[which];
    return SPANGPU_OK;
}

int spangpu_fsk_create(spangpu_fsk_t **out, int device, int n_channels, const spangpu_fsk_spec_t *spec, int framing_mode)
{
    if (out == NULL  ||  spec == NULL  ||  n_channels <= 0  ||  spec->baud_rate <= 0  ||  framing_mode < 0  ||  framing_mode > 2)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *out = NULL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess  ||  count <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    if (device < 0  ||  device >= count)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "device out of range");
    FSK_TRY(hipSetDevice(device));
    spangpu_fsk_s *f = (spangpu_fsk_s *) calloc(1, sizeof(*f));
    if (f == NULL)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    f->device = device;
    f->n_ch = n_channels;
    f->spec = *spec;
    f->span = span_of(spec);
    f->words = kFskScalars + 4*f->span;
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess)
    {
        free(f);
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipStreamCreate failed");
    }
    f->own_stream = true;
    const size_t words = (size_t) f->words*n_channels;
    if (hipMalloc(&f->st, words*sizeof(int32_t)) != hipSuccess
        ||  hipMalloc(&f->quarter, 257*sizeof(int16_t)) != hipSuccess
        ||  hipMalloc(&f->ev_count, (size_t) n_channels*sizeof(int32_t)) != hipSuccess
        ||  (f->h_count = (int32_t *) malloc((size_t) n_channels*sizeof(int32_t))) == NULL)
    {
        spangpu_fsk_destroy(f);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "allocation of the FSK bank failed");
    }
    // dds_int.c: one quadrant of a sine, 257 entries
    int16_t quarter[257];
    for (int i = 0;  i <= 256;  i++)
        quarter[i] = (int16_t) lrint(32767.0*sin(i*3.14159265358979323846/512.0));
    // fsk_rx_init() = memset + fsk_rx_restart(), fsk.c:723-742
    int32_t *one = (int32_t *) calloc(f->words, sizeof(int32_t));
    int32_t *host = (int32_t *) calloc(words, sizeof(int32_t));
    if (one == NULL  ||  host == NULL)
    {
        free(one);
        free(host);
        spangpu_fsk_destroy(f);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    }
    restart_words(one, spec, framing_mode);
    for (int w = 0;  w < kFskScalars;  w++)
    {
        for (int c = 0;  c < n_channels;  c++)
            host[(size_t) w*n_channels + c] = one[w];
    }
    hipError_t e = hipMemcpy(f->st, host, words*sizeof(int32_t), hipMemcpyHostToDevice);
    free(one);
    free(host);
    if (e == hipSuccess)
        e = hipMemcpy(f->quarter, quarter, sizeof(quarter), hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        spangpu_fsk_destroy(f);
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    }
    *out = f;
    return SPANGPU_OK;
}

void spangpu_fsk_destroy(spangpu_fsk_t *f)
{
    if (f == NULL)
        return;
    (void) hipSetDevice(f->device);
    if (f->stream)
        (void) hipStreamSynchronize(f->stream);
    (void) hipFree(f->st);
    (void) hipFree(f->quarter);
    (void) hipFree(f->d_pcm);
    (void) hipFree(f->d_lens);
    if (f->h_lens) (void) hipHostFree(f->h_lens);
    (void) hipFree(f->events);
    (void) hipFree(f->ev_count);
    free(f->h_events);
    free(f->h_count);
    if (f->own_stream  &&  f->stream)
        (void) hipStreamDestroy(f->stream);
    free(f);
}

int spangpu_fsk_channels(const spangpu_fsk_t *f) { return f  ?  f->n_ch  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_fsk_state_words(const spangpu_fsk_t *f) { return f  ?  f->words  :  SPANGPU_ERR_BAD_ARG; }

int spangpu_fsk_set_stream(spangpu_fsk_t *f, void *stream)
{
    if (f == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    FSK_TRY(hipSetDevice(f->device));
    FSK_TRY(hipStreamSynchronize(f->stream));
    if (f->own_stream)
        (void) hipStreamDestroy(f->stream);
    f->stream = (hipStream_t) stream;
    f->own_stream = false;
    return SPANGPU_OK;
}

int spangpu_fsk_sync(spangpu_fsk_t *f)
{
    if (f == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    FSK_TRY(hipSetDevice(f->device));
    FSK_TRY(hipStreamSynchronize(f->stream));
    return SPANGPU_OK;
}

int spangpu_fsk_rx(spangpu_fsk_t *f, const int16_t *amp, int mem_kind, int samples, long long stride)
{
    if (f == NULL  ||  amp == NULL  ||  samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (mem_kind != SPANGPU_MEM_HOST  &&  mem_kind != SPANGPU_MEM_DEVICE)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    if (stride <= 0)
        stride = samples;
    if (stride < samples)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "stride < samples");
    FSK_TRY(hipSetDevice(f->device));
    // at most one event per sample (a status change and a bit can share one sample: + 2)
    const int cap = samples + 2;
    if (cap > f->ev_cap)
    {
        FSK_TRY(hipStreamSynchronize(f->stream));
        (void) hipFree(f->events);
        f->events = NULL;
        f->ev_cap = 0;
        if (hipMalloc(&f->events, (size_t) f->n_ch*cap*sizeof(int16_t)) != hipSuccess)
            return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "event buffer");
        f->ev_cap = cap;
    }
    FskLaunch L;
    memset(&L, 0, sizeof(L));
    L.st = f->st;
    L.quarter = f->quarter;
    L.events = f->events;
    L.ev_count = f->ev_count;
    L.n_ch = f->n_ch;
    L.samples = samples;
    L.lens = f->next_lens;
    L.span = f->span;
    L.ev_cap = f->ev_cap;
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        const size_t need = (size_t) ((samples + 7) & ~7);
        if (need > f->pcm_cap)
        {
            FSK_TRY(hipStreamSynchronize(f->stream));
            (void) hipFree(f->d_pcm);
            f->d_pcm = NULL;
            f->pcm_cap = 0;
            if (hipMalloc(&f->d_pcm, need*f->n_ch*sizeof(int16_t)) != hipSuccess)
                return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "pcm staging");
            f->pcm_cap = need;
        }
        FSK_TRY(hipMemcpy2DAsync(f->d_pcm, f->pcm_cap*sizeof(int16_t), amp, (size_t) stride*sizeof(int16_t),
                                 (size_t) samples*sizeof(int16_t), f->n_ch, hipMemcpyHostToDevice, f->stream));
        // the caller's buffer is only borrowed for the call
        FSK_TRY(hipStreamSynchronize(f->stream));
        L.pcm = f->d_pcm;
        L.stride = (long long) f->pcm_cap;
    }
    else
    {
        L.pcm = amp;
        L.stride = stride;
    }
    L.vec = ((L.stride & 7) == 0  &&  (reinterpret_cast<uintptr_t>(L.pcm) & 15) == 0)  ?  1  :  0;
    const size_t lds = (size_t) (4*f->span*64)*sizeof(int32_t);
    if (g_fsk_waves != 1)
        hipLaunchKernelGGL(fsk_pair_kernel, dim3((f->n_ch + 63)/64), dim3(128), lds + 2*kFskMsgWords*64*sizeof(int32_t), f->stream, L);
    else
        hipLaunchKernelGGL(fsk_bank_kernel, dim3((f->n_ch + 63)/64), dim3(64), lds, f->stream, L);
    FSK_TRY(hipGetLastError());
    f->last_cap = f->ev_cap;
    return SPANGPU_OK;
}

// spangpu_fsk_rx() for a tick in which not every channel has a frame, or frames differ in length: channel c takes lens[c] samples
// of its row (0: it sits the call out, its state as it was, no events).  lens[] is host memory.
int spangpu_fsk_rx_var(spangpu_fsk_t *f, const int16_t *amp, int mem_kind, const int32_t *lens, int max_samples, long long stride)
{
    if (f == NULL  ||  amp == NULL  ||  lens == NULL  ||  max_samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    int longest = 0;
    bool all = true;
    for (int c = 0;  c < f->n_ch;  c++)
    {
        if (lens[c] < 0  ||  lens[c] > max_samples)
            return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "a channel's length is outside 0..max_samples");
        if (lens[c] > longest)
            longest = lens[c];
    }
    if (longest == 0)
        return SPANGPU_OK;
    for (int c = 0;  c < f->n_ch;  c++)
        all &= (lens[c] == longest);
    if (stride <= 0)
        stride = max_samples;
    if (all)
        return spangpu_fsk_rx(f, amp, mem_kind, longest, stride);
    FSK_TRY(hipSetDevice(f->device));
    if (f->d_lens == NULL)
    {
        FSK_TRY(hipMalloc(&f->d_lens, (size_t) f->n_ch*sizeof(int32_t)));
        FSK_TRY(hipHostMalloc(&f->h_lens, (size_t) f->n_ch*sizeof(int32_t)));
    }
    FSK_TRY(hipStreamSynchronize(f->stream));
    memcpy(f->h_lens, lens, (size_t) f->n_ch*sizeof(int32_t));
    FSK_TRY(hipMemcpyAsync(f->d_lens, f->h_lens, (size_t) f->n_ch*sizeof(int32_t), hipMemcpyHostToDevice, f->stream));
    f->next_lens = f->d_lens;
    const int rc = spangpu_fsk_rx(f, amp, mem_kind, longest, stride);
    f->next_lens = NULL;
    return rc;
}

int spangpu_fsk_events(spangpu_fsk_t *f, const int16_t **events, const int32_t **counts)
{
    if (f == NULL  ||  events == NULL  ||  counts == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (f->last_cap <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no spangpu_fsk_rx() yet");
    FSK_TRY(hipSetDevice(f->device));
    const size_t bytes = (size_t) f->n_ch*f->last_cap*sizeof(int16_t);
    if (bytes > f->h_events_cap)
    {
        free(f->h_events);
        f->h_events_cap = 0;
        if ((f->h_events = (int16_t *) malloc(bytes)) == NULL)
            return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "host event buffer");
        f->h_events_cap = bytes;
    }
    FSK_TRY(hipMemcpyAsync(f->h_events, f->events, bytes, hipMemcpyDeviceToHost, f->stream));
    FSK_TRY(hipMemcpyAsync(f->h_count, f->ev_count, (size_t) f->n_ch*sizeof(int32_t), hipMemcpyDeviceToHost, f->stream));
    FSK_TRY(hipStreamSynchronize(f->stream));
    *events = f->h_events;
    *counts = f->h_count;
    return f->last_cap;
}

int spangpu_fsk_get_state(spangpu_fsk_t *f, int channel, int32_t *words)
{
    if (f == NULL  ||  words == NULL  ||  channel < 0  ||  channel >= f->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    return read_words(f, channel, words);
}

int spangpu_fsk_set_state(spangpu_fsk_t *f, int channel, const int32_t *words)
{
    if (f == NULL  ||  words == NULL  ||  channel < 0  ||  channel >= f->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (words[FS_SPAN] != f->span)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "the correlation span of a channel is fixed by its bank's baud rate");
    return write_words(f, channel, words);
}

static int edit(spangpu_fsk_s *f, int channel, int what, int a, int b, int c, float x)
{
    if (f == NULL  ||  channel < 0  ||  channel >= f->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    int32_t *w = (int32_t *) malloc((size_t) f->words*sizeof(int32_t));
    if (w == NULL)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "malloc");
    int rc = read_words(f, channel, w);
    if (rc == SPANGPU_OK)
    {
        switch (what)
        {
        case 0:
            restart_words(w, &f->spec, a);
            break;
        case 1:
            cutoff_words(w, x);
            break;
        case 2:
            frame_words(w, a, b, c);
            break;
        case 3:
            // fsk_rx_fillin(), fsk.c:625-657: the current window slot is cleared, the oscillators run on, and
            // the slot index does not move
            if (a > 0)
            {
                int32_t *slot = w + kFskScalars + 4*w[FS_BUF_PTR];
                w[FS_DOT0RE] -= slot[0];
                w[FS_DOT0IM] -= slot[1];
                w[FS_DOT1RE] -= slot[2];
                w[FS_DOT1IM] -= slot[3];
                slot[0] = slot[1] = slot[2] = slot[3] = 0;
                w[FS_ACC0] = (int32_t) ((uint32_t) w[FS_ACC0] + (uint32_t) a*(uint32_t) w[FS_RATE0]);
                w[FS_ACC1] = (int32_t) ((uint32_t) w[FS_ACC1] + (uint32_t) a*(uint32_t) w[FS_RATE1]);
            }
            break;
        }
        rc = write_words(f, channel, w);
    }
    free(w);
    return rc;
}

int spangpu_fsk_restart(spangpu_fsk_t *f, int channel, int framing_mode)
{
    if (framing_mode < 0  ||  framing_mode > 2)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad framing mode");
    return edit(f, channel, 0, framing_mode, 0, 0, 0.0f);
}

int spangpu_fsk_set_signal_cutoff(spangpu_fsk_t *f, int channel, float cutoff_dbm0)
{
    return edit(f, channel, 1, 0, 0, 0, cutoff_dbm0);
}

int spangpu_fsk_set_frame_parameters(spangpu_fsk_t *f, int channel, int data_bits, int parity, int stop_bits)
{
    return edit(f, channel, 2, data_bits, parity, stop_bits, 0.0f);
}

int spangpu_fsk_fillin(spangpu_fsk_t *f, int channel, int len)
{
    return edit(f, channel, 3, len, 0, 0, 0.0f);
}

}   // extern "C"
