// mct_api.hip -- C ABI of the modem connect tone detector banks (include/spangpu.h, "modem connect tone
// banks"): batched modem_connect_tones_rx().  Device code: mct_dev.hpp (+ fsk_dev.hpp for the V.21 preamble
// hunter).  No CPU implementation of the receive path exists behind these entry points.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"
#include "mct_dev.hpp"

using namespace spg;

extern "C" int spangpu_set_error(int code, const char *msg);

#define MCT_TRY(expr)                                                                       \
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

struct spangpu_mct_s
{
    const int32_t *next_lens;   // per-channel lengths of the call being prepared (device), or NULL
    int32_t *d_lens;            // [n_ch], device
    int32_t *h_lens;            // [n_ch], pinned
    int device;
    int n_ch;
    int tone_type;              // after modem_connect_tones_rx_init()'s folding of the ANS variants
    int latch;
    int words;
    hipStream_t stream;
    bool own_stream;
    int32_t *st;
    int16_t *quarter;
    int16_t *d_pcm;
    size_t pcm_cap;
    int32_t *events;
    int32_t *ev_count;
    int ev_cap;
    int last_cap;
    int32_t *h_events;
    int32_t *h_count;
    size_t h_events_cap;
};

static const float kMaxPower = 3.14f + 3.02f;       // DBM0_MAX_POWER

// The level a report carries, from the integer the kernel recorded (libm's log10f, like the reference).
static int level_of(int tone, int32_t from)
{
    if (tone == MCT_NONE)
        return -99;
    if (tone == MCT_FAX_PREAMBLE)
    {
        // lfastrintf(fsk_rx_signal_power()): power_meter_current_dbm0(), power_meter.c:114-121
        const float dbm0 = (from <= 0)  ?  (-96.329f + kMaxPower)
                                        :  10.0f*log10f((float) from/(32767.0f*32767.0f) + 1.0e-10f) + kMaxPower;
        return (int) (long) dbm0;
    }
    // modem_connect_tones.c:561 (and :643,:659,:723,:777)
    const float db = (from == 0)  ?  (-96.329f + kMaxPower)  :  20.0f*log10f(from/32768.0f);
    return (int) (long) (db + kMaxPower + 0.8f);
}

static int32_t power_level_dbm0(float level)
{
    // power_meter_level_dbm0(), power_meter.c:82-92
    level -= kMaxPower;
    if (level > 0.0)
        level = 0.0;
    return (int32_t) (powf(10.0f, level/10.0f)*(32767.0f*32767.0f));
}

extern "C" int spangpu_fsk_waves_choice(void);       // fsk_api.hip: what spangpu_tune_fsk_waves() was given

template <int TYPE>
static void launch(const spangpu_mct_s *m, const MctLaunch &L)
{
    const bool fsk = (TYPE == MCT_FAX_PREAMBLE  ||  TYPE == MCT_FAX_CED_OR_PREAMBLE);
    const size_t lds = fsk  ?  (size_t) (4*kMctV21Span*64)*sizeof(int32_t)  :  0;
    if (TYPE == MCT_FAX_CED_OR_PREAMBLE  &&  spangpu_fsk_waves_choice() != 1)
        hipLaunchKernelGGL(mct_ced_pair_kernel, dim3((m->n_ch + 63)/64), dim3(128), lds + 3*64*sizeof(int32_t), m->stream, L);
    else
        hipLaunchKernelGGL(mct_bank_kernel<TYPE>, dim3((m->n_ch + 63)/64), dim3(64), lds, m->stream, L);
}

extern "C" {

int spangpu_mct_create(spangpu_mct_t **out, int device, int tone_type, int n_channels, int use_callback)
{
    if (out == NULL  ||  n_channels <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *out = NULL;
    // modem_connect_tones_rx_init(), modem_connect_tones.c:812-834: modifiers off, the ANS family is one detector
    int type = tone_type & 0xFFF;
    if (type == MCT_ANS_PR  ||  type == MCT_ANSAM  ||  type == MCT_ANSAM_PR)
        type = MCT_ANS;
    if (type < MCT_FAX_CNG  ||  type > MCT_CALLING_TONE)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "not a modem connect tone type");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess  ||  count <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    if (device < 0  ||  device >= count)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "device out of range");
    MCT_TRY(hipSetDevice(device));
    spangpu_mct_s *m = (spangpu_mct_s *) calloc(1, sizeof(*m));
    if (m == NULL)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    m->device = device;
    m->n_ch = n_channels;
    m->tone_type = type;
    m->latch = use_callback  ?  0  :  1;
    const bool fsk = (type == MCT_FAX_PREAMBLE  ||  type == MCT_FAX_CED_OR_PREAMBLE);
    m->words = kMctWords + (fsk  ?  (kFskScalars + 4*kMctV21Span)  :  0);
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess)
    {
        free(m);
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipStreamCreate failed");
    }
    m->own_stream = true;
    const size_t words = (size_t) m->words*n_channels;
    if (hipMalloc(&m->st, words*sizeof(int32_t)) != hipSuccess
        ||  hipMalloc(&m->quarter, 257*sizeof(int16_t)) != hipSuccess
        ||  hipMalloc(&m->ev_count, (size_t) n_channels*sizeof(int32_t)) != hipSuccess
        ||  (m->h_count = (int32_t *) malloc((size_t) n_channels*sizeof(int32_t))) == NULL)
    {
        spangpu_mct_destroy(m);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "allocation of the connect tone bank failed");
    }
    int16_t quarter[257];
    for (int i = 0;  i <= 256;  i++)
        quarter[i] = (int16_t) lrint(32767.0*sin(i*3.14159265358979323846/512.0));
    int32_t *one = (int32_t *) calloc(m->words, sizeof(int32_t));
    int32_t *host = (int32_t *) calloc(words, sizeof(int32_t));
    if (one == NULL  ||  host == NULL)
    {
        free(one);
        free(host);
        spangpu_mct_destroy(m);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    }
    one[MC_TONE_TYPE] = type;
    if (fsk)
    {
        // fsk_rx_init(&preset_fsk_specs[FSK_V21CH2], FSK_FRAME_MODE_SYNC) + fsk_rx_set_signal_cutoff(-45.5),
        // modem_connect_tones.c:820-821 (fsk.c:660-742,270-276)
        int32_t *w = one + kMctWords;
        w[FS_BAUD_RATE] = 300*100;
        w[FS_FRAMING] = 1;
        w[FS_ON_POWER] = power_level_dbm0(-45.5f + 2.5f - 5.3f);
        w[FS_OFF_POWER] = power_level_dbm0(-45.5f - 2.5f - 5.3f);
        w[FS_RATE0] = (int32_t) ((float) (1750 + 100)*65536.0f*65536.0f/8000);
        w[FS_RATE1] = (int32_t) ((float) (1750 - 100)*65536.0f*65536.0f/8000);
        w[FS_SPAN] = kMctV21Span;
        w[FS_SHIFT] = 5;
        w[FS_FRAME_POS] = -2;
    }
    for (int k = 0;  k < kMctWords + (fsk  ?  kFskScalars  :  0);  k++)
    {
        for (int c = 0;  c < n_channels;  c++)
            host[(size_t) k*n_channels + c] = one[k];
    }
    hipError_t e = hipMemcpy(m->st, host, words*sizeof(int32_t), hipMemcpyHostToDevice);
    free(one);
    free(host);
    if (e == hipSuccess)
        e = hipMemcpy(m->quarter, quarter, sizeof(quarter), hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        spangpu_mct_destroy(m);
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    }
    *out = m;
    return SPANGPU_OK;
}

void spangpu_mct_destroy(spangpu_mct_t *m)
{
    if (m == NULL)
        return;
    (void) hipSetDevice(m->device);
    if (m->stream)
        (void) hipStreamSynchronize(m->stream);
    (void) hipFree(m->st);
    (void) hipFree(m->quarter);
    (void) hipFree(m->d_pcm);
    (void) hipFree(m->d_lens);
    if (m->h_lens) (void) hipHostFree(m->h_lens);
    (void) hipFree(m->events);
    (void) hipFree(m->ev_count);
    free(m->h_events);
    free(m->h_count);
    if (m->own_stream  &&  m->stream)
        (void) hipStreamDestroy(m->stream);
    free(m);
}

int spangpu_mct_channels(const spangpu_mct_t *m) { return m  ?  m->n_ch  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_mct_state_words(const spangpu_mct_t *m) { return m  ?  m->words  :  SPANGPU_ERR_BAD_ARG; }

int spangpu_mct_set_stream(spangpu_mct_t *m, void *stream)
{
    if (m == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    MCT_TRY(hipSetDevice(m->device));
    MCT_TRY(hipStreamSynchronize(m->stream));
    if (m->own_stream)
        (void) hipStreamDestroy(m->stream);
    m->stream = (hipStream_t) stream;
    m->own_stream = false;
    return SPANGPU_OK;
}

int spangpu_mct_sync(spangpu_mct_t *m)
{
    if (m == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    MCT_TRY(hipSetDevice(m->device));
    MCT_TRY(hipStreamSynchronize(m->stream));
    return SPANGPU_OK;
}

int spangpu_mct_rx(spangpu_mct_t *m, const int16_t *amp, int mem_kind, int samples, long long stride)
{
    if (m == NULL  ||  amp == NULL  ||  samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (mem_kind != SPANGPU_MEM_HOST  &&  mem_kind != SPANGPU_MEM_DEVICE)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    if (stride <= 0)
        stride = samples;
    if (stride < samples)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "stride < samples");
    MCT_TRY(hipSetDevice(m->device));
    // a tone needs >= 415 ms to be declared and can only be withdrawn once declared: two reports per ~3300
    // samples at the very most; the preamble hunter needs 40 bits (1067 samples) per declaration
    const int cap = 8 + samples/256;
    if (cap > m->ev_cap)
    {
        MCT_TRY(hipStreamSynchronize(m->stream));
        (void) hipFree(m->events);
        m->events = NULL;
        m->ev_cap = 0;
        if (hipMalloc(&m->events, (size_t) m->n_ch*cap*2*sizeof(int32_t)) != hipSuccess)
            return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "event buffer");
        m->ev_cap = cap;
    }
    MctLaunch L;
    memset(&L, 0, sizeof(L));
    L.st = m->st;
    L.quarter = m->quarter;
    L.events = m->events;
    L.ev_count = m->ev_count;
    L.n_ch = m->n_ch;
    L.samples = samples;
    L.lens = m->next_lens;
    L.ev_cap = m->ev_cap;
    L.latch = m->latch;
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        const size_t need = (size_t) ((samples + 7) & ~7);
        if (need > m->pcm_cap)
        {
            MCT_TRY(hipStreamSynchronize(m->stream));
            (void) hipFree(m->d_pcm);
            m->d_pcm = NULL;
            m->pcm_cap = 0;
            if (hipMalloc(&m->d_pcm, need*m->n_ch*sizeof(int16_t)) != hipSuccess)
                return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "pcm staging");
            m->pcm_cap = need;
        }
        MCT_TRY(hipMemcpy2DAsync(m->d_pcm, m->pcm_cap*sizeof(int16_t), amp, (size_t) stride*sizeof(int16_t),
                                 (size_t) samples*sizeof(int16_t), m->n_ch, hipMemcpyHostToDevice, m->stream));
        MCT_TRY(hipStreamSynchronize(m->stream));       // the caller's buffer is only borrowed for the call
        L.pcm = m->d_pcm;
        L.stride = (long long) m->pcm_cap;
    }
    else
    {
        L.pcm = amp;
        L.stride = stride;
    }
    L.vec = ((L.stride & 7) == 0  &&  (reinterpret_cast<uintptr_t>(L.pcm) & 15) == 0)  ?  1  :  0;
    switch (m->tone_type)
    {
    case MCT_FAX_CNG:               launch<MCT_FAX_CNG>(m, L); break;
    case MCT_ANS:                   launch<MCT_ANS>(m, L); break;
    case MCT_FAX_PREAMBLE:          launch<MCT_FAX_PREAMBLE>(m, L); break;
    case MCT_FAX_CED_OR_PREAMBLE:   launch<MCT_FAX_CED_OR_PREAMBLE>(m, L); break;
    case MCT_BELL_ANS:              launch<MCT_BELL_ANS>(m, L); break;
    default:                        launch<MCT_CALLING_TONE>(m, L); break;
    }
    MCT_TRY(hipGetLastError());
    m->last_cap = m->ev_cap;
    return SPANGPU_OK;
}

// spangpu_mct_rx() for a tick in which not every channel has a frame, or frames differ in length: channel c takes lens[c] samples
// of its row (0: it sits the call out, its state as it was, no events).  lens[] is host memory.
int spangpu_mct_rx_var(spangpu_mct_t *m, const int16_t *amp, int mem_kind, const int32_t *lens, int max_samples, long long stride)
{
    if (m == NULL  ||  amp == NULL  ||  lens == NULL  ||  max_samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    int longest = 0;
    bool all = true;
    for (int c = 0;  c < m->n_ch;  c++)
    {
        if (lens[c] < 0  ||  lens[c] > max_samples)
            return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "a channel's length is outside 0..max_samples");
        if (lens[c] > longest)
            longest = lens[c];
    }
    if (longest == 0)
        return SPANGPU_OK;
    for (int c = 0;  c < m->n_ch;  c++)
        all &= (lens[c] == longest);
    if (stride <= 0)
        stride = max_samples;
    if (all)
        return spangpu_mct_rx(m, amp, mem_kind, longest, stride);
    MCT_TRY(hipSetDevice(m->device));
    if (m->d_lens == NULL)
    {
        MCT_TRY(hipMalloc(&m->d_lens, (size_t) m->n_ch*sizeof(int32_t)));
        MCT_TRY(hipHostMalloc(&m->h_lens, (size_t) m->n_ch*sizeof(int32_t)));
    }
    MCT_TRY(hipStreamSynchronize(m->stream));
    memcpy(m->h_lens, lens, (size_t) m->n_ch*sizeof(int32_t));
    MCT_TRY(hipMemcpyAsync(m->d_lens, m->h_lens, (size_t) m->n_ch*sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
    m->next_lens = m->d_lens;
    const int rc = spangpu_mct_rx(m, amp, mem_kind, longest, stride);
    m->next_lens = NULL;
    return rc;
}

int spangpu_mct_events(spangpu_mct_t *m, const int32_t **events, const int32_t **counts)
{
    if (m == NULL  ||  events == NULL  ||  counts == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (m->last_cap <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no spangpu_mct_rx() yet");
    MCT_TRY(hipSetDevice(m->device));
    const size_t bytes = (size_t) m->n_ch*m->last_cap*2*sizeof(int32_t);
    if (bytes > m->h_events_cap)
    {
        free(m->h_events);
        m->h_events_cap = 0;
        if ((m->h_events = (int32_t *) malloc(bytes)) == NULL)
            return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "host event buffer");
        m->h_events_cap = bytes;
    }
    MCT_TRY(hipMemcpyAsync(m->h_events, m->events, bytes, hipMemcpyDeviceToHost, m->stream));
    MCT_TRY(hipMemcpyAsync(m->h_count, m->ev_count, (size_t) m->n_ch*sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    MCT_TRY(hipStreamSynchronize(m->stream));
    for (int c = 0;  c < m->n_ch;  c++)
    {
        int32_t *e = m->h_events + (size_t) c*m->last_cap*2;
        const int cnt = (m->h_count[c] < m->last_cap)  ?  m->h_count[c]  :  m->last_cap;
        for (int i = 0;  i < cnt;  i++)
            e[2*i + 1] = level_of(e[2*i], e[2*i + 1]);
    }
    *events = m->h_events;
    *counts = m->h_count;
    return m->last_cap;
}

int spangpu_mct_get(spangpu_mct_t *m, int channel)
{
    if (m == NULL  ||  channel < 0  ||  channel >= m->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    // modem_connect_tones_rx_get(), modem_connect_tones.c:793-797: read and clear the latch
    MCT_TRY(hipSetDevice(m->device));
    int32_t hit = 0;
    const int32_t zero = 0;
    int32_t *at = m->st + (size_t) MC_HIT*m->n_ch + channel;
    MCT_TRY(hipMemcpyAsync(&hit, at, sizeof(hit), hipMemcpyDeviceToHost, m->stream));
    MCT_TRY(hipMemcpyAsync(at, &zero, sizeof(zero), hipMemcpyHostToDevice, m->stream));
    MCT_TRY(hipStreamSynchronize(m->stream));
    return hit;
}

int spangpu_mct_get_state(spangpu_mct_t *m, int channel, int32_t *words)
{
    if (m == NULL  ||  words == NULL  ||  channel < 0  ||  channel >= m->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    MCT_TRY(hipSetDevice(m->device));
    MCT_TRY(hipMemcpy2DAsync(words, sizeof(int32_t), m->st + channel, (size_t) m->n_ch*sizeof(int32_t), sizeof(int32_t), m->words,
                             hipMemcpyDeviceToHost, m->stream));
    MCT_TRY(hipStreamSynchronize(m->stream));
    return SPANGPU_OK;
}

// The reverse of spangpu_mct_get_state(): a channel's words as a caller holds them.
int spangpu_mct_set_state(spangpu_mct_t *m, int channel, const int32_t *words)
{
    if (m == NULL  ||  words == NULL  ||  channel < 0  ||  channel >= m->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    MCT_TRY(hipSetDevice(m->device));
    MCT_TRY(hipStreamSynchronize(m->stream));
    MCT_TRY(hipMemcpy2DAsync(m->st + channel, (size_t) m->n_ch*sizeof(int32_t), words, sizeof(int32_t), sizeof(int32_t), m->words,
                             hipMemcpyHostToDevice, m->stream));
    MCT_TRY(hipStreamSynchronize(m->stream));
    return SPANGPU_OK;
}

}   // extern "C"
