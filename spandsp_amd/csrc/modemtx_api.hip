// modemtx_api.hip -- C ABI of the modem transmitter banks (include/spangpu.h, "modem transmitter banks"): batched
// v29_tx() / v27ter_tx() as device-side signal sources.  Device code: modemtx_dev.hpp.  No CPU implementation exists behind
// these entry points.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"
#include "modem_tables.h"
#include "modemtx_dev.hpp"

using namespace spg;

extern "C" int spangpu_set_error(int code, const char *msg);

#define VT_TRY(expr)                                                                        \
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

struct spangpu_modemtx_s
{
    int device;
    int kind;               // kTxV29, kTxV27ter or kTxV17
    float *constel;         // V.17: the constellations of every rate + the ABCD training points
    int n_ch;
    hipStream_t stream;
    bool own_stream;
    int32_t *st;
    float *sine;
    float *shaper;
    int16_t *d_pcm;
    size_t pcm_cap;
};

static void put_f(int32_t *w, int idx, float v)
{
    memcpy(w + idx, &v, sizeof(float));
}

static float get_f(const int32_t *w, int idx)
{
    float v;
    memcpy(&v, w + idx, sizeof(float));
    return v;
}

static void gain_words(int32_t *w, int kind)
{
    if (kind != kTxV29)
        return;             // V.27ter keeps one gain per rate, V.17 one gain; both are set by xxx_tx_power()
    // set_working_gain(), v29tx.c:286-320
    const float base = get_f(w, VT_BASE_GAIN);
    switch (w[VT_BIT_RATE])
    {
    case 9600: put_f(w, VT_GAIN, 0.387f*base); break;
    case 7200: put_f(w, VT_GAIN, 0.605f*base); break;
    case 4800: put_f(w, VT_GAIN, 0.470f*base); break;
    }
}

static void power_words(int32_t *w, int kind, float power)
{
    if (kind == kTxV29)
    {
        // v29_tx_power(), v29tx.c:322-338; TX_PULSESHAPER_GAIN = 1.0f in the float build
        put_f(w, VT_BASE_GAIN, spg_db_to_amplitude_ratio(power - 3.14f)*32768.0f/1.000000f);
        gain_words(w, kind);
        return;
    }
    if (kind == kTxV17)
    {
        // v17_tx_power(), v17tx.c:371-383
        put_f(w, VT_BASE_GAIN, 0.223f*spg_db_to_amplitude_ratio(power - 3.14f)*32768.0f/1.000000f);
        return;
    }
    // v27ter_tx_power(), v27ter_tx.c:352-364: gain_2400 (word 1) and gain_4800 (word 2), both shaper gains 1.0f
    const float gain = spg_db_to_amplitude_ratio(power - 3.14f)*32768.0f;
    put_f(w, VT_BASE_GAIN, gain/1.000000f);
    put_f(w, VT_GAIN, gain/1.000000f);
}

static int restart_words(int32_t *w, int kind, int bit_rate, int tep, int short_train = 0)
{
    if (kind == kTxV17)
    {
        // v17_tx_restart(), v17tx.c:397-450
        if (bit_rate != 14400  &&  bit_rate != 12000  &&  bit_rate != 9600  &&  bit_rate != 7200  &&  bit_rate != 4800)
            return -1;
        w[VT_BIT_RATE] = bit_rate;
        w[VT_GAIN] = short_train  ?  0  :  1;          // diff
        for (int i = 0;  i < 18;  i++)
            w[VT_RRC_RE + i] = 0;
        w[VT_RRC_STEP] = 0;
        w[VT_TRAIN_SCRAMBLE] = 0;                       // convolution
        w[VT_SCRAMBLE] = 0x2ECDD5;
        w[VT_IN_TRAINING] = 1;
        w[VT_TRAINING_OFFSET] = short_train  ?  1  :  0;
        w[VT_TRAINING_STEP] = tep  ?  0  :  kV17Seg1;
        w[VT_CARRIER_PHASE] = 0;
        w[VT_BAUD_PHASE] = 0;
        w[VT_CONSTELLATION] = 0;
        return 0;
    }
    if (kind == kTxV27ter)
    {
        // v27ter_tx_restart(), v27ter_tx.c:384-409
        if (bit_rate != 4800  &&  bit_rate != 2400)
            return -1;
        w[VT_BIT_RATE] = bit_rate;
        for (int i = 0;  i < 18;  i++)
            w[VT_RRC_RE + i] = 0;
        w[VT_RRC_STEP] = 0;
        w[VT_SCRAMBLE] = 0x3C;
        w[VT_TRAIN_SCRAMBLE] = 0;               // scrambler_pattern_count
        w[VT_IN_TRAINING] = 1;
        w[VT_TRAINING_STEP] = tep  ?  0  :  kV27Seg2;
        w[VT_CARRIER_PHASE] = 0;
        w[VT_BAUD_PHASE] = 0;
        w[VT_CONSTELLATION] = 0;
        return 0;
    }
    // v29_tx_restart(), v29tx.c:365-404
    if (bit_rate != 9600  &&  bit_rate != 7200  &&  bit_rate != 4800)
        return -1;
    w[VT_BIT_RATE] = bit_rate;
    gain_words(w, kind);
    switch (bit_rate)
    {
    case 9600: w[VT_TRAINING_OFFSET] = 0; break;
    case 7200: w[VT_TRAINING_OFFSET] = 2; break;
    case 4800: w[VT_TRAINING_OFFSET] = 4; break;
    default: return -1;
    }
    for (int i = 0;  i < 18;  i++)
        w[VT_RRC_RE + i] = 0;
    w[VT_RRC_STEP] = 0;
    w[VT_SCRAMBLE] = 0;
    w[VT_TRAIN_SCRAMBLE] = 0x2A;
    w[VT_IN_TRAINING] = 1;
    w[VT_TRAINING_STEP] = tep  ?  0  :  kVtSeg1;
    w[VT_CARRIER_PHASE] = 0;
    w[VT_BAUD_PHASE] = 0;
    w[VT_CONSTELLATION] = 0;
    return 0;
}

static int rw_words(spangpu_modemtx_s *t, int ch, int32_t *w, bool write)
{
    VT_TRY(hipSetDevice(t->device));
    if (write)
        VT_TRY(hipMemcpy2DAsync(t->st + ch, (size_t) t->n_ch*sizeof(int32_t), w, sizeof(int32_t), sizeof(int32_t), kV29TxWords,
                                hipMemcpyHostToDevice, t->stream));
    else
        VT_TRY(hipMemcpy2DAsync(w, sizeof(int32_t), t->st + ch, (size_t) t->n_ch*sizeof(int32_t), sizeof(int32_t), kV29TxWords,
                                hipMemcpyDeviceToHost, t->stream));
    VT_TRY(hipStreamSynchronize(t->stream));
    return SPANGPU_OK;
}

extern "C" {

int spangpu_modemtx_create(spangpu_modemtx_t **out, int device, int modem, int n_channels, int bit_rate, int tep, const uint32_t *seeds)
{
    if (out == NULL  ||  n_channels <= 0  ||  (modem != SPANGPU_V29  &&  modem != SPANGPU_V27TER  &&  modem != SPANGPU_V17))
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments (modem SPANGPU_V29, SPANGPU_V27TER or SPANGPU_V17)");
    const int kind = (modem == SPANGPU_V29)  ?  kTxV29  :  ((modem == SPANGPU_V27TER)  ?  kTxV27ter  :  kTxV17);
    int32_t probe[kV29TxWords];
    memset(probe, 0, sizeof(probe));
    if (restart_words(probe, kind, bit_rate, tep) != 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bit rate not valid for this modem (V.29: 9600/7200/4800, V.27ter: 4800/2400, V.17: 14400/12000/9600/7200/4800)");
    *out = NULL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess  ||  count <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    if (device < 0  ||  device >= count)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "device out of range");
    VT_TRY(hipSetDevice(device));
    spangpu_modemtx_s *t = (spangpu_modemtx_s *) calloc(1, sizeof(*t));
    if (t == NULL)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    t->device = device;
    t->kind = kind;
    t->n_ch = n_channels;
    if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess)
    {
        free(t);
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipStreamCreate failed");
    }
    t->own_stream = true;
    const size_t words = (size_t) kV29TxWords*n_channels;
    if (hipMalloc(&t->st, words*sizeof(int32_t)) != hipSuccess
        ||  hipMalloc(&t->sine, 2048*sizeof(float)) != hipSuccess
        ||  hipMalloc(&t->shaper, 225*sizeof(float)) != hipSuccess
        ||  hipMalloc(&t->constel, 496*sizeof(float)) != hipSuccess)
    {
        spangpu_modemtx_destroy(t);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "allocation of the V.29 transmitter bank failed");
    }
    float sine[2048];
    float shaper[225];
    memset(shaper, 0, sizeof(shaper));
    spg_make_sine_table(sine);
    // make_modem_filter -t: V.29 10 phases x 9 taps, excess bandwidth 0.25; V.27ter 4800 bps 5 x 9 and 2400 bps
    // 20 x 9, excess bandwidth 0.5 (make_modem_filter.c:375-412)
    // V.17 shapes with the V.29 parameters (make_modem_filter.c:319-331)
    if (((kind != kTxV27ter)  ?  spg_make_tx_pulseshaper(10, 9, 0.25, shaper)
                           :  (spg_make_tx_pulseshaper(5, 9, 0.5, shaper) | spg_make_tx_pulseshaper(20, 9, 0.5, shaper + 45))) != 0)
    {
        spangpu_modemtx_destroy(t);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "table scratch");
    }
    // v29_tx_init(), v29tx.c:406-434 / v27ter_tx_init(), v27ter_tx.c:411-437
    int32_t one[kV29TxWords];
    memset(one, 0, sizeof(one));
    one[VT_BIT_RATE] = bit_rate;
    one[VT_CARRIER_RATE] = spg_dds_phase_ratef((kind == kTxV29)  ?  1700.0f  :  1800.0f);
    // the V.17 constellations (v17_v32bis_tx_constellation_maps.h) and the ABCD training points (:314-323)
    float constel[496];
    memset(constel, 0, sizeof(constel));
    {
        static const int rates[5] = {14400, 12000, 9600, 7200, 4800};
        static const float abcd[8] = {-6.0f, -2.0f, 2.0f, -6.0f, 6.0f, 2.0f, -2.0f, 6.0f};
        int at = 0;
        for (int r = 0;  r < 5;  r++)
        {
            int8_t pts[128][2];
            const int np = spg_make_v17_constellation(rates[r], pts);
            for (int i = 0;  i < np;  i++)
            {
                constel[2*(at + i)] = (float) pts[i][0];
                constel[2*(at + i) + 1] = (float) pts[i][1];
            }
            at += np;
        }
        memcpy(constel + 2*at, abcd, sizeof(abcd));
    }
    power_words(one, kind, -14.0f);
    restart_words(one, kind, bit_rate, tep);
    int32_t *host = (int32_t *) malloc(words*sizeof(int32_t));
    if (host == NULL)
    {
        spangpu_modemtx_destroy(t);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "malloc");
    }
    for (int k = 0;  k < kV29TxWords;  k++)
    {
        for (int c = 0;  c < n_channels;  c++)
            host[(size_t) k*n_channels + c] = one[k];
    }
    for (int c = 0;  c < n_channels;  c++)
        host[(size_t) VT_PRBS*n_channels + c] = (int32_t) ((seeds  ?  seeds[c]  :  (uint32_t) (c*2654435761u + 1u)) & 0x7FFFu);
    hipError_t e = hipMemcpy(t->st, host, words*sizeof(int32_t), hipMemcpyHostToDevice);
    free(host);
    if (e == hipSuccess)
        e = hipMemcpy(t->sine, sine, sizeof(sine), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(t->shaper, shaper, sizeof(shaper), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(t->constel, constel, sizeof(constel), hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        spangpu_modemtx_destroy(t);
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    }
    *out = t;
    return SPANGPU_OK;
}

void spangpu_modemtx_destroy(spangpu_modemtx_t *t)
{
    if (t == NULL)
        return;
    (void) hipSetDevice(t->device);
    if (t->stream)
        (void) hipStreamSynchronize(t->stream);
    (void) hipFree(t->st);
    (void) hipFree(t->sine);
    (void) hipFree(t->shaper);
    (void) hipFree(t->constel);
    (void) hipFree(t->d_pcm);
    if (t->own_stream  &&  t->stream)
        (void) hipStreamDestroy(t->stream);
    free(t);
}

int spangpu_modemtx_channels(const spangpu_modemtx_t *t) { return t  ?  t->n_ch  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_modemtx_state_words(void) { return kV29TxWords; }

int spangpu_modemtx_set_stream(spangpu_modemtx_t *t, void *stream)
{
    if (t == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    VT_TRY(hipSetDevice(t->device));
    VT_TRY(hipStreamSynchronize(t->stream));
    if (t->own_stream)
        (void) hipStreamDestroy(t->stream);
    t->stream = (hipStream_t) stream;
    t->own_stream = false;
    return SPANGPU_OK;
}

int spangpu_modemtx_sync(spangpu_modemtx_t *t)
{
    if (t == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    VT_TRY(hipSetDevice(t->device));
    VT_TRY(hipStreamSynchronize(t->stream));
    return SPANGPU_OK;
}

int spangpu_modemtx_power(spangpu_modemtx_t *t, int channel, float power_dbm0)
{
    if (t == NULL  ||  channel < 0  ||  channel >= t->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    int32_t w[kV29TxWords];
    int rc = rw_words(t, channel, w, false);
    if (rc != SPANGPU_OK)
        return rc;
    power_words(w, t->kind, power_dbm0);
    return rw_words(t, channel, w, true);
}

// Every channel's level and carrier frequency in one pass over the state words: a population of lines for the receiver
// banks' workloads (carrier 1700 Hz +- 7 Hz, level -30 .. -10 dBm0).  The level is xxx_tx_power()'s; the
// carrier frequency is not something the reference's modulator lets a caller choose (v29tx.c:431 fixes it) -- it stands for the
// frequency shift of the line between the modems.
int spangpu_modemtx_line(spangpu_modemtx_t *t, const float *power_dbm0, const float *carrier_hz)
{
    if (t == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    if (power_dbm0 == NULL  &&  carrier_hz == NULL)
        return SPANGPU_OK;
    for (int c = 0;  carrier_hz  &&  c < t->n_ch;  c++)
    {
        if (!(carrier_hz[c] > 0.0f  &&  carrier_hz[c] < 4000.0f))
            return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "carrier frequency out of range");
    }
    VT_TRY(hipSetDevice(t->device));
    VT_TRY(hipStreamSynchronize(t->stream));
    const size_t n = (size_t) t->n_ch;
    int32_t *host = (int32_t *) malloc((size_t) kV29TxWords*n*sizeof(int32_t));
    if (host == NULL)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "malloc");
    hipError_t e = hipMemcpy(host, t->st, (size_t) kV29TxWords*n*sizeof(int32_t), hipMemcpyDeviceToHost);
    for (int c = 0;  e == hipSuccess  &&  c < t->n_ch;  c++)
    {
        int32_t w[kV29TxWords];
        for (int k = 0;  k < kV29TxWords;  k++)
            w[k] = host[(size_t) k*n + c];
        if (power_dbm0)
            power_words(w, t->kind, power_dbm0[c]);
        if (carrier_hz)
            w[VT_CARRIER_RATE] = spg_dds_phase_ratef(carrier_hz[c]);
        for (int k = 0;  k < kV29TxWords;  k++)
            host[(size_t) k*n + c] = w[k];
    }
    if (e == hipSuccess)
        e = hipMemcpy(t->st, host, (size_t) kV29TxWords*n*sizeof(int32_t), hipMemcpyHostToDevice);
    free(host);
    if (e != hipSuccess)
        return spangpu_set_error(SPANGPU_ERR_HIP, "state transfer failed");
    return SPANGPU_OK;
}

int spangpu_modemtx_restart(spangpu_modemtx_t *t, int channel, int bit_rate, int tep)
{
    return spangpu_modemtx_restart_ex(t, channel, bit_rate, tep, 0);
}

int spangpu_modemtx_restart_ex(spangpu_modemtx_t *t, int channel, int bit_rate, int tep, int short_train)
{
    if (t == NULL  ||  channel < 0  ||  channel >= t->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    int32_t w[kV29TxWords];
    int rc = rw_words(t, channel, w, false);
    if (rc != SPANGPU_OK)
        return rc;
    if (restart_words(w, t->kind, bit_rate, tep, short_train) != 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bit rate not valid for this modem");
    return rw_words(t, channel, w, true);
}

int spangpu_modemtx_get_state(spangpu_modemtx_t *t, int channel, int32_t *words)
{
    if (t == NULL  ||  words == NULL  ||  channel < 0  ||  channel >= t->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    return rw_words(t, channel, words, false);
}

int spangpu_modemtx_tx(spangpu_modemtx_t *t, int mem_kind, int16_t *pcm, long long stride, int samples)
{
    if (t == NULL  ||  pcm == NULL  ||  samples < 0  ||  stride < samples)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (mem_kind != SPANGPU_MEM_HOST  &&  mem_kind != SPANGPU_MEM_DEVICE)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    if (samples == 0)
        return 0;
    VT_TRY(hipSetDevice(t->device));
    V29TxLaunch L;
    memset(&L, 0, sizeof(L));
    L.st = t->st;
    L.sine = t->sine;
    L.shaper = t->shaper;
    L.constel = t->constel;
    L.n_ch = t->n_ch;
    L.samples = samples;
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        const size_t need = (size_t) ((samples + 7) & ~7);
        if (need > t->pcm_cap)
        {
            VT_TRY(hipStreamSynchronize(t->stream));
            (void) hipFree(t->d_pcm);
            t->d_pcm = NULL;
            t->pcm_cap = 0;
            if (hipMalloc(&t->d_pcm, need*t->n_ch*sizeof(int16_t)) != hipSuccess)
                return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "pcm staging");
            t->pcm_cap = need;
        }
        L.pcm = t->d_pcm;
        L.stride = (long long) t->pcm_cap;
    }
    else
    {
        L.pcm = pcm;
        L.stride = stride;
    }
    L.vec = ((L.stride & 7) == 0  &&  (reinterpret_cast<uintptr_t>(L.pcm) & 15) == 0)  ?  1  :  0;
    if (t->kind == kTxV29)
        hipLaunchKernelGGL(modemtx_bank_kernel<kTxV29>, dim3((t->n_ch + 63)/64), dim3(64), 0, t->stream, L);
    else if (t->kind == kTxV17)
        hipLaunchKernelGGL(modemtx_bank_kernel<kTxV17>, dim3((t->n_ch + 63)/64), dim3(64), 0, t->stream, L);
    else
        hipLaunchKernelGGL(modemtx_bank_kernel<kTxV27ter>, dim3((t->n_ch + 63)/64), dim3(64), 0, t->stream, L);
    VT_TRY(hipGetLastError());
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        VT_TRY(hipMemcpy2DAsync(pcm, (size_t) stride*sizeof(int16_t), t->d_pcm, t->pcm_cap*sizeof(int16_t),
                                (size_t) samples*sizeof(int16_t), t->n_ch, hipMemcpyDeviceToHost, t->stream));
        VT_TRY(hipStreamSynchronize(t->stream));
    }
    return samples;
}

// The pulse shaper tables this library builds (for tests).  which: 0 V.29 [10][9], 1 V.27ter 4800 bps [5][9],
// 2 V.27ter 2400 bps [20][9].  Returns the number of floats.
int spangpu_modemtx_table(int which, float *out, int max)
{
    static const int sets[3] = {10, 5, 20};
    static const double excess[3] = {0.25, 0.5, 0.5};
    if (out == NULL  ||  which < 0  ||  which > 2  ||  max < sets[which]*9)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad table request");
    if (spg_make_tx_pulseshaper(sets[which], 9, excess[which], out) != 0)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "table scratch");
    return sets[which]*9;
}

}   // extern "C"
