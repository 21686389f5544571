// sigtone_api.hip -- C ABI of the in-band signalling tone banks (include/spangpu.h, "signalling tone banks"):
// batched sig_tone_rx() and sig_tone_tx().  Device code: sigtone_dev.hpp.  No CPU implementation of either path
// exists behind these entry points.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"
#include "sigtone_dev.hpp"

using namespace spg;

extern "C" int spangpu_set_error(int code, const char *msg);

#define SIG_TRY(expr)                                                                       \
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

static const float kMaxPower = 3.14f + 3.02f;       // DBM0_MAX_POWER

static int32_t power_level_dbm0(float level)
{
    // power_meter_level_dbm0(), power_meter.c:82-92
    level -= kMaxPower;
    if (level > 0.0)
        level = 0.0;
    return (int32_t) (powf(10.0f, level/10.0f)*(32767.0f*32767.0f));
}

// the parts of the three descriptors the host needs (sig_tone.c:137-223)
struct SigDesc
{
    int freq[2];
    int amp[2][2];
    int high_low_timeout;
    int tones;
    float detection_ratio;
    float sharp_threshold;
    float flat_threshold;
};

static const SigDesc kDesc[3] =
{
    {{2280, 0},    {{-10, -20}, {0, 0}}, 400*8, 1, 13.0f, -30.0f, -30.0f},
    {{2600, 0},    {{-8, -8}, {0, 0}},   0,     1, 15.6f, -30.0f, -30.0f},
    {{2400, 2600}, {{-8, -8}, {-8, -8}}, 0,     2, 15.6f, -30.0f, -30.0f}
};

static int check_device(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess  ||  count <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    if (device < 0  ||  device >= count)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "device out of range");
    return SPANGPU_OK;
}

// ---- receiver banks ----------------------------------------------------------------------------------------------

struct spangpu_sigtone_rx_s
{
    int device;
    int n_ch;
    int tone_type;
    hipStream_t stream;
    bool own_stream;
    int32_t *st;
    int16_t *d_pcm;
    size_t pcm_cap;
    int32_t *d_lens;
    int32_t *h_lens;
    const int32_t *next_lens;
    int32_t *events;
    int32_t *ev_count;
    int ev_cap;
    int last_cap;
    int32_t *h_events;
    int32_t *h_count;
    size_t h_events_cap;
    int32_t thresholds[3];
};

extern "C" {

int spangpu_sigtone_rx_create(spangpu_sigtone_rx_t **out, int device, int tone_type, int n_channels)
{
    if (out == NULL  ||  n_channels <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *out = NULL;
    // sig_tone_rx_init() refuses other types (sig_tone.c:679-680)
    if (tone_type < SPANGPU_SIG_TONE_2280HZ  ||  tone_type > SPANGPU_SIG_TONE_2400HZ_2600HZ)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "not a signalling tone type");
    int rc = check_device(device);
    if (rc != SPANGPU_OK)
        return rc;
    SIG_TRY(hipSetDevice(device));
    spangpu_sigtone_rx_s *b = (spangpu_sigtone_rx_s *) calloc(1, sizeof(*b));
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    b->device = device;
    b->n_ch = n_channels;
    b->tone_type = tone_type;
    // sig_tone.c:714-716
    const SigDesc &d = kDesc[tone_type - 1];
    b->thresholds[0] = power_level_dbm0(d.flat_threshold);
    b->thresholds[1] = power_level_dbm0(d.sharp_threshold);
    b->thresholds[2] = (int32_t) (powf(10.0f, d.detection_ratio/10.0f) + 1.0f);
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess)
    {
        free(b);
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipStreamCreate failed");
    }
    b->own_stream = true;
    const size_t words = (size_t) kSigRxWords*n_channels;
    if (hipMalloc(&b->st, words*sizeof(int32_t)) != hipSuccess
        ||  hipMalloc(&b->ev_count, (size_t) n_channels*sizeof(int32_t)) != hipSuccess
        ||  (b->h_count = (int32_t *) malloc((size_t) n_channels*sizeof(int32_t))) == NULL)
    {
        spangpu_sigtone_rx_destroy(b);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "allocation of the signalling tone bank failed");
    }
    // memset(s, 0, sizeof(*s)) and last_sample_tone_present = -1 (sig_tone.c:690-704)
    int32_t *host = (int32_t *) calloc(words, sizeof(int32_t));
    if (host == NULL)
    {
        spangpu_sigtone_rx_destroy(b);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    }
    for (int c = 0;  c < n_channels;  c++)
        host[(size_t) SG_LAST_PRESENT*n_channels + c] = -1;
    const hipError_t e = hipMemcpy(b->st, host, words*sizeof(int32_t), hipMemcpyHostToDevice);
    free(host);
    if (e != hipSuccess)
    {
        spangpu_sigtone_rx_destroy(b);
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    }
    *out = b;
    return SPANGPU_OK;
}

void spangpu_sigtone_rx_destroy(spangpu_sigtone_rx_t *b)
{
    if (b == NULL)
        return;
    (void) hipSetDevice(b->device);
    if (b->stream)
        (void) hipStreamSynchronize(b->stream);
    (void) hipFree(b->st);
    (void) hipFree(b->d_pcm);
    (void) hipFree(b->d_lens);
    if (b->h_lens) (void) hipHostFree(b->h_lens);
    (void) hipFree(b->events);
    (void) hipFree(b->ev_count);
    free(b->h_events);
    free(b->h_count);
    if (b->own_stream  &&  b->stream)
        (void) hipStreamDestroy(b->stream);
    free(b);
}

int spangpu_sigtone_rx_channels(const spangpu_sigtone_rx_t *b) { return b  ?  b->n_ch  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_sigtone_rx_state_words(const spangpu_sigtone_rx_t *b) { return b  ?  kSigRxWords  :  SPANGPU_ERR_BAD_ARG; }

int spangpu_sigtone_rx_set_stream(spangpu_sigtone_rx_t *b, void *stream)
{
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    SIG_TRY(hipSetDevice(b->device));
    SIG_TRY(hipStreamSynchronize(b->stream));
    if (b->own_stream)
        (void) hipStreamDestroy(b->stream);
    b->stream = (hipStream_t) stream;
    b->own_stream = false;
    return SPANGPU_OK;
}

int spangpu_sigtone_rx_sync(spangpu_sigtone_rx_t *b)
{
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    SIG_TRY(hipSetDevice(b->device));
    SIG_TRY(hipStreamSynchronize(b->stream));
    return SPANGPU_OK;
}

// sig_tone_rx_set_mode(s, mode, duration), sig_tone.c:666-669, for one channel or (channel < 0) for all
int spangpu_sigtone_rx_set_mode(spangpu_sigtone_rx_t *b, int channel, int mode)
{
    if (b == NULL  ||  channel >= b->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    SIG_TRY(hipSetDevice(b->device));
    int32_t *row = b->st + (size_t) SG_RX_TONE*b->n_ch;
    if (channel >= 0)
    {
        const int32_t v = mode;
        SIG_TRY(hipMemcpyAsync(row + channel, &v, sizeof(v), hipMemcpyHostToDevice, b->stream));
        SIG_TRY(hipStreamSynchronize(b->stream));
        return SPANGPU_OK;
    }
    int32_t *host = (int32_t *) malloc((size_t) b->n_ch*sizeof(int32_t));
    if (host == NULL)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "malloc");
    for (int c = 0;  c < b->n_ch;  c++)
        host[c] = mode;
    const hipError_t e = hipMemcpyAsync(row, host, (size_t) b->n_ch*sizeof(int32_t), hipMemcpyHostToDevice, b->stream);
    const hipError_t e2 = hipStreamSynchronize(b->stream);
    free(host);
    if (e != hipSuccess  ||  e2 != hipSuccess)
        return spangpu_set_error(SPANGPU_ERR_HIP, "mode upload failed");
    return SPANGPU_OK;
}

int spangpu_sigtone_rx(spangpu_sigtone_rx_t *b, int16_t *amp, int mem_kind, int samples, long long stride)
{
    if (b == NULL  ||  amp == NULL  ||  samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (mem_kind != SPANGPU_MEM_HOST  &&  mem_kind != SPANGPU_MEM_DEVICE)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    if (stride <= 0)
        stride = samples;
    if (stride < samples)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "stride < samples");
    SIG_TRY(hipSetDevice(b->device));
    // a tone is declared after 3 ms of consistent detection and withdrawn after 8 ms (or at once in flat mode, which
    // takes 225 ms to enter): 24 samples between two reports at the very least
    const int cap = 4 + samples/16;
    if (cap > b->ev_cap)
    {
        SIG_TRY(hipStreamSynchronize(b->stream));
        (void) hipFree(b->events);
        b->events = NULL;
        b->ev_cap = 0;
        if (hipMalloc(&b->events, (size_t) b->n_ch*cap*3*sizeof(int32_t)) != hipSuccess)
            return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "event buffer");
        b->ev_cap = cap;
    }
    SigRxLaunch L;
    memset(&L, 0, sizeof(L));
    L.st = b->st;
    L.events = b->events;
    L.ev_count = b->ev_count;
    L.n_ch = b->n_ch;
    L.samples = samples;
    L.lens = b->next_lens;
    L.ev_cap = b->ev_cap;
    L.flat_threshold = b->thresholds[0];
    L.sharp_threshold = b->thresholds[1];
    L.detection_ratio = b->thresholds[2];
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        const size_t need = (size_t) ((samples + 7) & ~7);
        if (need > b->pcm_cap)
        {
            SIG_TRY(hipStreamSynchronize(b->stream));
            (void) hipFree(b->d_pcm);
            b->d_pcm = NULL;
            b->pcm_cap = 0;
            if (hipMalloc(&b->d_pcm, need*b->n_ch*sizeof(int16_t)) != hipSuccess)
                return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "pcm staging");
            b->pcm_cap = need;
        }
        SIG_TRY(hipMemcpy2DAsync(b->d_pcm, b->pcm_cap*sizeof(int16_t), amp, (size_t) stride*sizeof(int16_t),
                                 (size_t) samples*sizeof(int16_t), b->n_ch, hipMemcpyHostToDevice, b->stream));
        L.pcm = b->d_pcm;
        L.stride = (long long) b->pcm_cap;
    }
    else
    {
        L.pcm = amp;
        L.stride = stride;
    }
    L.vec = ((L.stride & 7) == 0  &&  (reinterpret_cast<uintptr_t>(L.pcm) & 15) == 0)  ?  1  :  0;
    const dim3 grid((b->n_ch + 63)/64);
    switch (b->tone_type)
    {
    case SPANGPU_SIG_TONE_2280HZ:   hipLaunchKernelGGL(sigtone_rx_kernel<1>, grid, dim3(64), 0, b->stream, L); break;
    case SPANGPU_SIG_TONE_2600HZ:   hipLaunchKernelGGL(sigtone_rx_kernel<2>, grid, dim3(64), 0, b->stream, L); break;
    default:                        hipLaunchKernelGGL(sigtone_rx_kernel<3>, grid, dim3(64), 0, b->stream, L); break;
    }
    SIG_TRY(hipGetLastError());
    b->last_cap = b->ev_cap;
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        // the frame goes back as the receiver left it; the caller's buffer is only borrowed for the call
        SIG_TRY(hipMemcpy2DAsync(amp, (size_t) stride*sizeof(int16_t), b->d_pcm, b->pcm_cap*sizeof(int16_t),
                                 (size_t) samples*sizeof(int16_t), b->n_ch, hipMemcpyDeviceToHost, b->stream));
        SIG_TRY(hipStreamSynchronize(b->stream));
    }
    return SPANGPU_OK;
}

// spangpu_sigtone_rx() for a tick in which not every channel has a frame, or frames differ in length: channel c takes
// lens[c] samples of its row (0: it sits the call out, its state and its row as they were, no events).  lens[] is host memory.
int spangpu_sigtone_rx_var(spangpu_sigtone_rx_t *b, int16_t *amp, int mem_kind, const int32_t *lens, int max_samples, long long stride)
{
    if (b == NULL  ||  amp == NULL  ||  lens == NULL  ||  max_samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    int longest = 0;
    bool all = true;
    for (int c = 0;  c < b->n_ch;  c++)
    {
        if (lens[c] < 0  ||  lens[c] > max_samples)
            return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "a channel's length is outside 0..max_samples");
        if (lens[c] > longest)
            longest = lens[c];
    }
    if (longest == 0)
        return SPANGPU_OK;
    for (int c = 0;  c < b->n_ch;  c++)
        all &= (lens[c] == longest);
    if (stride <= 0)
        stride = max_samples;
    if (all)
        return spangpu_sigtone_rx(b, amp, mem_kind, longest, stride);
    SIG_TRY(hipSetDevice(b->device));
    if (b->d_lens == NULL)
    {
        SIG_TRY(hipMalloc(&b->d_lens, (size_t) b->n_ch*sizeof(int32_t)));
        SIG_TRY(hipHostMalloc(&b->h_lens, (size_t) b->n_ch*sizeof(int32_t)));
    }
    SIG_TRY(hipStreamSynchronize(b->stream));
    memcpy(b->h_lens, lens, (size_t) b->n_ch*sizeof(int32_t));
    SIG_TRY(hipMemcpyAsync(b->d_lens, b->h_lens, (size_t) b->n_ch*sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    b->next_lens = b->d_lens;
    const int rc = spangpu_sigtone_rx(b, amp, mem_kind, longest, stride);
    b->next_lens = NULL;
    return rc;
}

int spangpu_sigtone_rx_events(spangpu_sigtone_rx_t *b, const int32_t **events, const int32_t **counts)
{
    if (b == NULL  ||  events == NULL  ||  counts == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (b->last_cap <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no spangpu_sigtone_rx() yet");
    SIG_TRY(hipSetDevice(b->device));
    const size_t bytes = (size_t) b->n_ch*b->last_cap*3*sizeof(int32_t);
    if (bytes > b->h_events_cap)
    {
        free(b->h_events);
        b->h_events_cap = 0;
        if ((b->h_events = (int32_t *) malloc(bytes)) == NULL)
            return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "host event buffer");
        b->h_events_cap = bytes;
    }
    SIG_TRY(hipMemcpyAsync(b->h_events, b->events, bytes, hipMemcpyDeviceToHost, b->stream));
    SIG_TRY(hipMemcpyAsync(b->h_count, b->ev_count, (size_t) b->n_ch*sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    SIG_TRY(hipStreamSynchronize(b->stream));
    for (int c = 0;  c < b->n_ch;  c++)
    {
        if (b->h_count[c] > b->last_cap)
            return spangpu_set_error(SPANGPU_ERR_STATE, "a channel reported more often than the event buffer holds");
    }
    *events = b->h_events;
    *counts = b->h_count;
    return b->last_cap;
}

int spangpu_sigtone_rx_get_state(spangpu_sigtone_rx_t *b, int channel, int32_t *words)
{
    if (b == NULL  ||  words == NULL  ||  channel < 0  ||  channel >= b->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    SIG_TRY(hipSetDevice(b->device));
    SIG_TRY(hipMemcpy2DAsync(words, sizeof(int32_t), b->st + channel, (size_t) b->n_ch*sizeof(int32_t), sizeof(int32_t), kSigRxWords,
                             hipMemcpyDeviceToHost, b->stream));
    SIG_TRY(hipStreamSynchronize(b->stream));
    return SPANGPU_OK;
}

// The reverse of spangpu_sigtone_rx_get_state(): a channel's 27 words as a caller holds them (a snapshot taken earlier).
int spangpu_sigtone_rx_set_state(spangpu_sigtone_rx_t *b, int channel, const int32_t *words)
{
    if (b == NULL  ||  words == NULL  ||  channel < 0  ||  channel >= b->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    SIG_TRY(hipSetDevice(b->device));
    SIG_TRY(hipMemcpy2DAsync(b->st + channel, (size_t) b->n_ch*sizeof(int32_t), words, sizeof(int32_t), sizeof(int32_t), kSigRxWords,
                             hipMemcpyHostToDevice, b->stream));
    SIG_TRY(hipStreamSynchronize(b->stream));
    return SPANGPU_OK;
}

int spangpu_sigtone_rx_thresholds(const spangpu_sigtone_rx_t *b, int32_t out[3])
{
    if (b == NULL  ||  out == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    memcpy(out, b->thresholds, sizeof(b->thresholds));
    return SPANGPU_OK;
}

}   // extern "C"

// ---- sender banks --------------------------------------------------------------------------------------------------

struct spangpu_sigtone_tx_s
{
    int device;
    int n_ch;
    int tone_type;
    hipStream_t stream;
    bool own_stream;
    int32_t *st;
    int16_t *quarter;
    int16_t *d_pcm;
    size_t pcm_cap;
    int32_t *start;
    int32_t *request;
    int32_t *d_modes;           // [2][n_ch]: modes, durations
    int32_t *h_modes;           // pinned
    int32_t *h_request;
    int32_t *h_start;
    int samples;                // of the frame in progress (0: none)
    int32_t phase_rate[2];
    int32_t scaling[2][2];
};

static int tx_run(spangpu_sigtone_tx_s *b, int16_t *amp, int mem_kind, long long stride)
{
    SigTxLaunch L;
    memset(&L, 0, sizeof(L));
    L.st = b->st;
    L.quarter = b->quarter;
    L.start = b->start;
    L.request = b->request;
    L.n_ch = b->n_ch;
    L.samples = b->samples;
    L.tones = kDesc[b->tone_type - 1].tones;
    memcpy(L.phase_rate, b->phase_rate, sizeof(L.phase_rate));
    memcpy(L.scaling, b->scaling, sizeof(L.scaling));
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        const size_t need = (size_t) ((b->samples + 7) & ~7);
        if (need > b->pcm_cap)
        {
            SIG_TRY(hipStreamSynchronize(b->stream));
            (void) hipFree(b->d_pcm);
            b->d_pcm = NULL;
            b->pcm_cap = 0;
            if (hipMalloc(&b->d_pcm, need*b->n_ch*sizeof(int16_t)) != hipSuccess)
                return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "pcm staging");
            b->pcm_cap = need;
        }
        SIG_TRY(hipMemcpy2DAsync(b->d_pcm, b->pcm_cap*sizeof(int16_t), amp, (size_t) stride*sizeof(int16_t),
                                 (size_t) b->samples*sizeof(int16_t), b->n_ch, hipMemcpyHostToDevice, b->stream));
        L.pcm = b->d_pcm;
        L.stride = (long long) b->pcm_cap;
    }
    else
    {
        L.pcm = amp;
        L.stride = stride;
    }
    hipLaunchKernelGGL(sigtone_tx_kernel, dim3((b->n_ch + 63)/64), dim3(64), 0, b->stream, L);
    SIG_TRY(hipGetLastError());
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        SIG_TRY(hipMemcpy2DAsync(amp, (size_t) stride*sizeof(int16_t), b->d_pcm, b->pcm_cap*sizeof(int16_t),
                                 (size_t) b->samples*sizeof(int16_t), b->n_ch, hipMemcpyDeviceToHost, b->stream));
    }
    SIG_TRY(hipMemcpyAsync(b->h_request, b->request, (size_t) b->n_ch*sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    SIG_TRY(hipMemcpyAsync(b->h_start, b->start, (size_t) b->n_ch*sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    SIG_TRY(hipStreamSynchronize(b->stream));
    int pending = 0;
    for (int c = 0;  c < b->n_ch;  c++)
        pending += (b->h_request[c] != 0);
    return pending;
}

extern "C" {

int spangpu_sigtone_tx_create(spangpu_sigtone_tx_t **out, int device, int tone_type, int n_channels)
{
    if (out == NULL  ||  n_channels <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *out = NULL;
    if (tone_type < SPANGPU_SIG_TONE_2280HZ  ||  tone_type > SPANGPU_SIG_TONE_2400HZ_2600HZ)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "not a signalling tone type");
    int rc = check_device(device);
    if (rc != SPANGPU_OK)
        return rc;
    SIG_TRY(hipSetDevice(device));
    spangpu_sigtone_tx_s *b = (spangpu_sigtone_tx_s *) calloc(1, sizeof(*b));
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    b->device = device;
    b->n_ch = n_channels;
    b->tone_type = tone_type;
    // sig_tone_tx_init(), sig_tone.c:369-380: dds_phase_rate() and dds_scaling_dbm0() (dds_int.c:316-331)
    const SigDesc &d = kDesc[tone_type - 1];
    for (int i = 0;  i < 2;  i++)
    {
        b->phase_rate[i] = d.freq[i]  ?  (int32_t) ((float) d.freq[i]*65536.0f*65536.0f/8000)  :  0;
        for (int k = 0;  k < 2;  k++)
            b->scaling[i][k] = (int16_t) (powf(10.0f, ((float) d.amp[i][k] - 3.14f)/20.0f)*32767.0f);
    }
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess)
    {
        free(b);
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipStreamCreate failed");
    }
    b->own_stream = true;
    const size_t nb = (size_t) n_channels*sizeof(int32_t);
    if (hipMalloc(&b->st, kSigTxWords*nb) != hipSuccess
        ||  hipMalloc(&b->quarter, 257*sizeof(int16_t)) != hipSuccess
        ||  hipMalloc(&b->start, nb) != hipSuccess
        ||  hipMalloc(&b->request, nb) != hipSuccess
        ||  hipMalloc(&b->d_modes, 2*nb) != hipSuccess
        ||  hipHostMalloc(&b->h_modes, 2*nb) != hipSuccess
        ||  (b->h_request = (int32_t *) malloc(nb)) == NULL
        ||  (b->h_start = (int32_t *) malloc(nb)) == NULL)
    {
        spangpu_sigtone_tx_destroy(b);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "allocation of the signalling tone sender bank failed");
    }
    int16_t quarter[257];
    for (int i = 0;  i <= 256;  i++)
        quarter[i] = (int16_t) lrint(32767.0*sin(i*3.14159265358979323846/512.0));
    if (hipMemcpy(b->quarter, quarter, sizeof(quarter), hipMemcpyHostToDevice) != hipSuccess
        ||  hipMemset(b->st, 0, kSigTxWords*nb) != hipSuccess
        ||  hipMemset(b->start, 0, nb) != hipSuccess
        ||  hipMemset(b->request, 0, nb) != hipSuccess)
    {
        spangpu_sigtone_tx_destroy(b);
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    }
    *out = b;
    return SPANGPU_OK;
}

void spangpu_sigtone_tx_destroy(spangpu_sigtone_tx_t *b)
{
    if (b == NULL)
        return;
    (void) hipSetDevice(b->device);
    if (b->stream)
        (void) hipStreamSynchronize(b->stream);
    (void) hipFree(b->st);
    (void) hipFree(b->quarter);
    (void) hipFree(b->d_pcm);
    (void) hipFree(b->start);
    (void) hipFree(b->request);
    (void) hipFree(b->d_modes);
    if (b->h_modes) (void) hipHostFree(b->h_modes);
    free(b->h_request);
    free(b->h_start);
    if (b->own_stream  &&  b->stream)
        (void) hipStreamDestroy(b->stream);
    free(b);
}

int spangpu_sigtone_tx_channels(const spangpu_sigtone_tx_t *b) { return b  ?  b->n_ch  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_sigtone_tx_state_words(const spangpu_sigtone_tx_t *b) { return b  ?  kSigTxWords  :  SPANGPU_ERR_BAD_ARG; }

int spangpu_sigtone_tx_set_stream(spangpu_sigtone_tx_t *b, void *stream)
{
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    SIG_TRY(hipSetDevice(b->device));
    SIG_TRY(hipStreamSynchronize(b->stream));
    if (b->own_stream)
        (void) hipStreamDestroy(b->stream);
    b->stream = (hipStream_t) stream;
    b->own_stream = false;
    return SPANGPU_OK;
}

// sig_tone_tx_set_mode(s, mode, duration) on every channel whose modes[] entry is not negative (host arrays of
// n_channels entries)
int spangpu_sigtone_tx_set_modes(spangpu_sigtone_tx_t *b, const int32_t *modes, const int32_t *durations)
{
    if (b == NULL  ||  modes == NULL  ||  durations == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    SIG_TRY(hipSetDevice(b->device));
    SIG_TRY(hipStreamSynchronize(b->stream));
    const size_t nb = (size_t) b->n_ch*sizeof(int32_t);
    memcpy(b->h_modes, modes, nb);
    memcpy(b->h_modes + b->n_ch, durations, nb);
    SIG_TRY(hipMemcpyAsync(b->d_modes, b->h_modes, 2*nb, hipMemcpyHostToDevice, b->stream));
    hipLaunchKernelGGL(sigtone_tx_set_mode_kernel, dim3((b->n_ch + 255)/256), dim3(256), 0, b->stream,
                       b->st, b->n_ch, b->d_modes, b->d_modes + b->n_ch, kDesc[b->tone_type - 1].high_low_timeout);
    SIG_TRY(hipGetLastError());
    return SPANGPU_OK;
}

int spangpu_sigtone_tx_set_mode(spangpu_sigtone_tx_t *b, int channel, int mode, int duration)
{
    if (b == NULL  ||  channel >= b->n_ch  ||  mode < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    SIG_TRY(hipSetDevice(b->device));
    SIG_TRY(hipStreamSynchronize(b->stream));
    if (channel >= 0)
    {
        // one channel: its five words through the host (sig_tone_tx_set_mode(), sig_tone.c:326-345), not two arrays of
        // n_channels entries through a kernel
        int32_t w[kSigTxWords];
        const size_t pitch = (size_t) b->n_ch*sizeof(int32_t);
        SIG_TRY(hipMemcpy2D(w, sizeof(int32_t), b->st + channel, pitch, sizeof(int32_t), kSigTxWords, hipMemcpyDeviceToHost));
        const int old_tones = w[SX_TONE] & (SIG_1_PRESENT | SIG_2_PRESENT);
        const int new_tones = mode & (SIG_1_PRESENT | SIG_2_PRESENT);
        if (new_tones  &&  old_tones != new_tones)
            w[SX_HIGH_LOW] = kDesc[b->tone_type - 1].high_low_timeout;
        if ((mode & SIG_1_PRESENT)  &&  !(w[SX_TONE] & SIG_1_PRESENT))
            w[SX_PHASE0] = 0;
        if ((mode & SIG_2_PRESENT)  &&  !(w[SX_TONE] & SIG_2_PRESENT))
            w[SX_PHASE1] = 0;
        w[SX_TONE] = mode;
        w[SX_TIMEOUT] = duration;
        SIG_TRY(hipMemcpy2D(b->st + channel, pitch, w, sizeof(int32_t), sizeof(int32_t), kSigTxWords, hipMemcpyHostToDevice));
        return SPANGPU_OK;
    }
    for (int c = 0;  c < b->n_ch;  c++)
    {
        b->h_modes[c] = mode;
        b->h_modes[b->n_ch + c] = duration;
    }
    SIG_TRY(hipMemcpyAsync(b->d_modes, b->h_modes, 2*(size_t) b->n_ch*sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    hipLaunchKernelGGL(sigtone_tx_set_mode_kernel, dim3((b->n_ch + 255)/256), dim3(256), 0, b->stream,
                       b->st, b->n_ch, b->d_modes, b->d_modes + b->n_ch, kDesc[b->tone_type - 1].high_low_timeout);
    SIG_TRY(hipGetLastError());
    return SPANGPU_OK;
}

// A new frame for every channel.  Returns the number of channels that stopped for their update request (see
// spangpu_sigtone_tx_requests()), 0 when the whole frame is done, or a negative error.
int spangpu_sigtone_tx(spangpu_sigtone_tx_t *b, int16_t *amp, int mem_kind, int samples, long long stride)
{
    if (b == NULL  ||  amp == NULL  ||  samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (mem_kind != SPANGPU_MEM_HOST  &&  mem_kind != SPANGPU_MEM_DEVICE)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    if (stride <= 0)
        stride = samples;
    if (stride < samples)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "stride < samples");
    SIG_TRY(hipSetDevice(b->device));
    SIG_TRY(hipMemsetAsync(b->start, 0, (size_t) b->n_ch*sizeof(int32_t), b->stream));
    b->samples = samples;
    return tx_run(b, amp, mem_kind, stride);
}

// The same frame again, for the channels that had stopped: each goes on from where it stopped, in whatever mode its
// callback has set meanwhile.  Same return value.
int spangpu_sigtone_tx_continue(spangpu_sigtone_tx_t *b, int16_t *amp, int mem_kind, long long stride)
{
    if (b == NULL  ||  amp == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (b->samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no frame in progress");
    if (stride <= 0)
        stride = b->samples;
    SIG_TRY(hipSetDevice(b->device));
    return tx_run(b, amp, mem_kind, stride);
}

// After spangpu_sigtone_tx() / _continue(): request[c] != 0 for the channels whose callback is due, stopped[c] = the
// sample of the frame they stopped at (= samples for the channels that are done).
int spangpu_sigtone_tx_requests(spangpu_sigtone_tx_t *b, const int32_t **request, const int32_t **stopped)
{
    if (b == NULL  ||  request == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (b->samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no frame in progress");
    *request = b->h_request;
    if (stopped)
        *stopped = b->h_start;
    return SPANGPU_OK;
}

int spangpu_sigtone_tx_get_state(spangpu_sigtone_tx_t *b, int channel, int32_t *words)
{
    if (b == NULL  ||  words == NULL  ||  channel < 0  ||  channel >= b->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    SIG_TRY(hipSetDevice(b->device));
    SIG_TRY(hipMemcpy2DAsync(words, sizeof(int32_t), b->st + channel, (size_t) b->n_ch*sizeof(int32_t), sizeof(int32_t), kSigTxWords,
                             hipMemcpyDeviceToHost, b->stream));
    SIG_TRY(hipStreamSynchronize(b->stream));
    return SPANGPU_OK;
}

}   // extern "C"
