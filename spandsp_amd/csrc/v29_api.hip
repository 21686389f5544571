// v29_api.hip -- C ABI of the batched V.29 receiver (include/spangpu.h, "V.29 receiver banks").
// Device code: v29_dev.hpp; constant tables: modem_tables.c.  No CPU implementation exists
// behind these entry points.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"
#include "modem_tables.h"
#include "v29_dev.hpp"

using namespace spg;

extern "C" int spangpu_set_error(int code, const char *msg);

#define V29_TRY(expr)                                                                       \
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

struct spangpu_v29_s
{
    int device;
    int n_ch;
    int bit_rate;
    hipStream_t stream;
    bool own_stream;
    uint32_t *state;            // [kV29Words][n_ch]
    V29Tables *tab;
    int16_t *d_amp;
    size_t amp_cap;
    int8_t *events;
    int32_t *ev_count;
    int ev_cap;
    int8_t *h_events;
    int32_t *h_count;
    int last_cap;
};

// power_meter_level_dbm0(), power_meter.c:82-92
static int32_t level_dbm0(float level)
{
    level -= (3.14f + 3.02f);
    if (level > 0.0)
        level = 0.0;
    const float l = powf(10.0f, level/10.0f)*(32767.0f*32767.0f);
    return (int32_t) l;
}

// v29_rx_init() + v29_rx_restart(.., false), v29rx.c:1019-1131, as one channel's state words
static int initial_words(uint32_t *w, int bit_rate, float cutoff_dbm0)
{
    float f[kV29Floats];
    int32_t i[kV29Ints];
    memset(f, 0, sizeof(f));
    memset(i, 0, sizeof(i));
    switch (bit_rate)
    {
    case 9600: i[VI_TRAINING_CD] = 0; break;
    case 7200: i[VI_TRAINING_CD] = 2; break;
    case 4800: i[VI_TRAINING_CD] = 4; break;
    default: return -1;
    }
    i[VI_BIT_RATE] = bit_rate;
    i[VI_TRAIN_SCRAMBLE] = 0x2A;
    i[VI_STAGE] = V29_SYMBOL_ACQUISITION;
    i[VI_PHASE_RATE] = (int32_t) (1700.0f*65536.0f*65536.0f/8000);
    i[VI_ON_POWER] = (int32_t) (level_dbm0(cutoff_dbm0 + 2.5f)*0.4f);       // v29rx.c:163-169
    i[VI_OFF_POWER] = (int32_t) (level_dbm0(cutoff_dbm0 - 2.5f)*0.4f);
    i[VI_EQ_PUT_STEP] = kRrcSets*10/(3*2) - 1;
    f[VF_EQ_COEFF + 2*16] = 3.0f;                                           // equalizer_reset()
    f[VF_EQ_DELTA] = 0.21f/kEqLen;
    f[VF_AGC] = (1.25f/1.0f)/735.0f;
    f[VF_TRACK_I] = 8000.0f;
    f[VF_TRACK_P] = 8000000.0f;
    memcpy(w, f, sizeof(f));
    memcpy(w + kV29Floats, i, sizeof(i));
    return 0;
}

extern "C" {

int spangpu_v29_create(spangpu_v29_t **out, int device, int n_channels, int bit_rate)
{
    if (out == nullptr  ||  n_channels <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *out = nullptr;
    uint32_t w[kV29Words];
    if (initial_words(w, bit_rate, -28.5f) < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "V.29 bit rate must be 9600, 7200 or 4800");
    if (spangpu_device_count() <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    if (device < 0  ||  device >= spangpu_device_count())
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "device out of range");
    V29_TRY(hipSetDevice(device));
    spangpu_v29_t *m = (spangpu_v29_t *) calloc(1, sizeof(*m));
    if (m == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    m->device = device;
    m->n_ch = n_channels;
    m->bit_rate = bit_rate;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess)
    {
        free(m);
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipStreamCreate failed");
    }
    m->own_stream = true;
    const size_t n = (size_t) n_channels;
    V29Tables *ht = (V29Tables *) calloc(1, sizeof(V29Tables));
    uint32_t *hs = (uint32_t *) malloc(n*kV29Words*sizeof(uint32_t));
    if (ht == nullptr  ||  hs == nullptr
        ||  hipMalloc(&m->state, n*kV29Words*sizeof(uint32_t)) != hipSuccess
        ||  hipMalloc(&m->tab, sizeof(V29Tables)) != hipSuccess
        ||  hipMalloc(&m->ev_count, n*sizeof(int32_t)) != hipSuccess
        ||  hipHostMalloc(&m->h_count, n*sizeof(int32_t)) != hipSuccess)
    {
        free(ht);
        free(hs);
        spangpu_v29_destroy(m);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "allocation of V.29 bank failed");
    }
    // constant tables (modem_tables.c): V.29 rx pulse shaper = 48 x 27, 1700 Hz, 2400 baud, 50 % excess
    // bandwidth (make_modem_filter.c:403-411); Godard = 1700 Hz, 2400 baud, alpha 0.99, triggers 1000 / 30,
    // steps 5 / 1 (src/Makefile.am:559-560)
    spg_make_rx_pulseshaper(kRrcSets, kRrcLen, 1700.0, 2400.0, 0.5, ht->rrc_re, ht->rrc_im);
    spg_make_sine_table(ht->sine);
    spg_make_sqrt_table(ht->sqrt_tab);
    spg_make_godard(1700.0, 2400.0, 0.99, ht->godard);
    ht->coarse_trigger = 1000.0f;
    ht->fine_trigger = 30.0f;
    ht->coarse_step = 5;
    ht->fine_step = 1;
    spg_make_v29_space_map(ht->space_map);
    for (int k = 0;  k < kV29Words;  k++)
    {
        for (size_t c = 0;  c < n;  c++)
            hs[(size_t) k*n + c] = w[k];
    }
    hipError_t rc1 = hipMemcpy(m->tab, ht, sizeof(V29Tables), hipMemcpyHostToDevice);
    hipError_t rc2 = hipMemcpy(m->state, hs, n*kV29Words*sizeof(uint32_t), hipMemcpyHostToDevice);
    free(ht);
    free(hs);
    if (rc1 != hipSuccess  ||  rc2 != hipSuccess)
    {
        spangpu_v29_destroy(m);
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    }
    *out = m;
    return SPANGPU_OK;
}

int spangpu_v29_destroy(spangpu_v29_t *m)
{
    if (m == nullptr)
        return SPANGPU_OK;
    (void) hipSetDevice(m->device);
    if (m->stream)
        (void) hipStreamSynchronize(m->stream);
    if (m->state) (void) hipFree(m->state);
    if (m->tab) (void) hipFree(m->tab);
    if (m->d_amp) (void) hipFree(m->d_amp);
    if (m->events) (void) hipFree(m->events);
    if (m->ev_count) (void) hipFree(m->ev_count);
    if (m->h_events) (void) hipHostFree(m->h_events);
    if (m->h_count) (void) hipHostFree(m->h_count);
    if (m->own_stream  &&  m->stream)
        (void) hipStreamDestroy(m->stream);
    free(m);
    return SPANGPU_OK;
}

int spangpu_v29_channels(const spangpu_v29_t *m) { return m  ?  m->n_ch  :  SPANGPU_ERR_BAD_ARG; }

int spangpu_v29_set_stream(spangpu_v29_t *m, void *hip_stream)
{
    if (m == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    (void) hipStreamSynchronize(m->stream);
    if (m->own_stream)
        (void) hipStreamDestroy(m->stream);
    if (hip_stream)
    {
        m->stream = (hipStream_t) hip_stream;
        m->own_stream = false;
    }
    else
    {
        V29_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
        m->own_stream = true;
    }
    return SPANGPU_OK;
}

int spangpu_v29_sync(spangpu_v29_t *m)
{
    if (m == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    V29_TRY(hipStreamSynchronize(m->stream));
    return SPANGPU_OK;
}

int spangpu_v29_rx(spangpu_v29_t *m, const int16_t *amp, int mem, int samples, long long stride)
{
    if (m == nullptr  ||  amp == nullptr  ||  samples < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (samples == 0)
        return 0;
    if (stride <= 0)
        stride = samples;
    V29_TRY(hipSetDevice(m->device));
    // at most 4 bits per baud, a baud every 8000/2400 samples, plus a handful of status events
    const int cap = ((samples*3*4 + 9)/10 + 8 + 15) & ~15;
    if (cap > m->ev_cap)
    {
        if (m->events) (void) hipFree(m->events);
        if (m->h_events) (void) hipHostFree(m->h_events);
        m->events = nullptr;
        m->h_events = nullptr;
        m->ev_cap = 0;
        V29_TRY(hipMalloc(&m->events, (size_t) m->n_ch*cap));
        V29_TRY(hipHostMalloc(&m->h_events, (size_t) m->n_ch*cap));
        m->ev_cap = cap;
    }
    const int16_t *d_amp = amp;
    long long d_stride = stride;
    if (mem == SPANGPU_MEM_HOST)
    {
        if ((size_t) samples > m->amp_cap)
        {
            if (m->d_amp) (void) hipFree(m->d_amp);
            m->d_amp = nullptr;
            m->amp_cap = 0;
            V29_TRY(hipMalloc(&m->d_amp, (size_t) m->n_ch*samples*sizeof(int16_t)));
            m->amp_cap = samples;
        }
        V29_TRY(hipMemcpy2DAsync(m->d_amp, m->amp_cap*sizeof(int16_t), amp, stride*sizeof(int16_t),
                                 samples*sizeof(int16_t), m->n_ch, hipMemcpyHostToDevice, m->stream));
        d_amp = m->d_amp;
        d_stride = (long long) m->amp_cap;
    }
    else if (mem != SPANGPU_MEM_DEVICE)
    {
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    }
    V29Launch L;
    memset(&L, 0, sizeof(L));
    L.amp = d_amp;
    L.stride = d_stride;
    L.samples = samples;
    L.n_ch = m->n_ch;
    L.state = m->state;
    L.events = m->events;
    L.ev_count = m->ev_count;
    L.ev_cap = m->ev_cap;
    L.tab = m->tab;
    // enough workgroups to put a wave on every SIMD (256 CUs x 4) before filling the waves
    if (m->n_ch >= 64*1024)
        hipLaunchKernelGGL(v29_bank_kernel<64>, dim3((m->n_ch + 63)/64), dim3(64), 0, m->stream, L);
    else if (m->n_ch >= 32*1024)
        hipLaunchKernelGGL(v29_bank_kernel<32>, dim3((m->n_ch + 31)/32), dim3(64), 0, m->stream, L);
    else
        hipLaunchKernelGGL(v29_bank_kernel<16>, dim3((m->n_ch + 15)/16), dim3(64), 0, m->stream, L);
    V29_TRY(hipGetLastError());
    m->last_cap = m->ev_cap;
    if (mem == SPANGPU_MEM_HOST)
        V29_TRY(hipStreamSynchronize(m->stream));
    return 0;
}

// The put_bit stream of the last spangpu_v29_rx() call: for channel c, counts[c] entries at
// events + c*cap, each 0/1 (a descrambled data bit) or a negative SIG_STATUS_* code
// (spandsp/async.h:66-103), in the order v29_rx() would have called put_bit().  Returns cap.
int spangpu_v29_events(spangpu_v29_t *m, const int8_t **events, const int32_t **counts)
{
    if (m == nullptr  ||  events == nullptr  ||  counts == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (m->last_cap <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no spangpu_v29_rx() yet");
    V29_TRY(hipSetDevice(m->device));
    V29_TRY(hipMemcpyAsync(m->h_events, m->events, (size_t) m->n_ch*m->last_cap, hipMemcpyDeviceToHost, m->stream));
    V29_TRY(hipMemcpyAsync(m->h_count, m->ev_count, (size_t) m->n_ch*sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    V29_TRY(hipStreamSynchronize(m->stream));
    *events = m->h_events;
    *counts = m->h_count;
    return m->last_cap;
}

// One channel's state as 238 float words + 43 int words (order: v29_dev.hpp "State word map").
int spangpu_v29_get_state(spangpu_v29_t *m, int channel, float *fwords, int32_t *iwords)
{
    if (m == nullptr  ||  channel < 0  ||  channel >= m->n_ch  ||  fwords == nullptr  ||  iwords == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    V29_TRY(hipSetDevice(m->device));
    V29_TRY(hipStreamSynchronize(m->stream));
    uint32_t w[kV29Words];
    V29_TRY(hipMemcpy2D(w, sizeof(uint32_t), m->state + channel, (size_t) m->n_ch*sizeof(uint32_t),
                        sizeof(uint32_t), kV29Words, hipMemcpyDeviceToHost));
    memcpy(fwords, w, kV29Floats*sizeof(float));
    memcpy(iwords, w + kV29Floats, kV29Ints*sizeof(int32_t));
    return SPANGPU_OK;
}

// v29_rx_restart(s, bit_rate, false) for one channel (v29rx.c:1019-1098)
int spangpu_v29_restart(spangpu_v29_t *m, int channel)
{
    if (m == nullptr  ||  channel < 0  ||  channel >= m->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    V29_TRY(hipSetDevice(m->device));
    V29_TRY(hipStreamSynchronize(m->stream));
    uint32_t w[kV29Words];
    uint32_t old[kV29Words];
    initial_words(w, m->bit_rate, -28.5f);
    V29_TRY(hipMemcpy2D(old, sizeof(uint32_t), m->state + channel, (size_t) m->n_ch*sizeof(uint32_t),
                        sizeof(uint32_t), kV29Words, hipMemcpyDeviceToHost));
    // what a restart keeps: the cutoff powers, the saved equaliser and the saved carrier rate
    w[kV29Floats + VI_ON_POWER] = old[kV29Floats + VI_ON_POWER];
    w[kV29Floats + VI_OFF_POWER] = old[kV29Floats + VI_OFF_POWER];
    w[kV29Floats + VI_PHASE_RATE_SAVE] = old[kV29Floats + VI_PHASE_RATE_SAVE];
    for (int k = 0;  k < 2*kEqLen;  k++)
        w[VF_EQ_SAVE + k] = old[VF_EQ_SAVE + k];
    w[VF_TRAIN_ERR] = old[VF_TRAIN_ERR];
    w[kV29Floats + VI_LAST_ANGLES] = old[kV29Floats + VI_LAST_ANGLES];
    w[kV29Floats + VI_LAST_ANGLES + 1] = old[kV29Floats + VI_LAST_ANGLES + 1];
    w[kV29Floats + VI_EQ_SKIP] = 0;
    V29_TRY(hipMemcpy2D(m->state + channel, (size_t) m->n_ch*sizeof(uint32_t), w, sizeof(uint32_t),
                        sizeof(uint32_t), kV29Words, hipMemcpyHostToDevice));
    return SPANGPU_OK;
}

// The constant tables this library builds (for tests): rrc_re/im [48*27], sine [2048],
// sqrt [193], godard [7].
int spangpu_modem_tables(float *rrc_re, float *rrc_im, float *sine, uint16_t *sqrt_tab, float *godard)
{
    if (rrc_re  &&  rrc_im)
        spg_make_rx_pulseshaper(kRrcSets, kRrcLen, 1700.0, 2400.0, 0.5, rrc_re, rrc_im);
    if (sine)
        spg_make_sine_table(sine);
    if (sqrt_tab)
        spg_make_sqrt_table(sqrt_tab);
    if (godard)
        spg_make_godard(1700.0, 2400.0, 0.99, godard);
    return SPANGPU_OK;
}

}   // extern "C"
