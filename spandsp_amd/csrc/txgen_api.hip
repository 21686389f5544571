// txgen_api.hip -- C ABI of the signal-source banks (include/spangpu.h, "signal source banks"):
// batched tone_gen() / dtmf_tx() / bell_mf_tx() / r2_mf_tx().  Device code: txgen_dev.hpp.
// No CPU implementation exists behind these entry points.

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"
#include "modem_tables.h"
#include "txgen_dev.hpp"

using namespace spg;

extern "C" int spangpu_set_error(int code, const char *msg);

#define TX_TRY(expr)                                                                        \
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

struct spangpu_txbank_s
{
    int device;
    int kind;
    int n_ch;
    hipStream_t stream;
    bool own_stream;
    int32_t *st;            // [kTxWords][n_ch]
    float *sine;            // [2048]
    int16_t *d_pcm;         // staging for host-resident output
    size_t pcm_cap;         // samples per channel
    int32_t *d_lens;        // [n_ch]
    uint8_t *d_digits;      // staging for put
    size_t digits_cap;
    int32_t *d_put_lens;    // [n_ch]
    int32_t *d_put_res;     // [n_ch]
    TxDigitTable dig;
};

// The frequency plans of the reference's digit senders.
static const int k_dtmf_row[4] = {697, 770, 852, 941};                  // dtmf.c:114-121
static const int k_dtmf_col[4] = {1209, 1336, 1477, 1633};
static const char k_dtmf_keys[] = "123A456B789C*0#D";                   // dtmf.c:123
static const int k_bell_freq[6] = {700, 900, 1100, 1300, 1500, 1700};   // bell_r2_mf.c:104-121: all pairs i < j, by j then i
static const char k_bell_keys[] = "1234567890CA*B#";                    // bell_r2_mf.c:124
static const int k_r2_fwd_freq[6] = {1380, 1500, 1620, 1740, 1860, 1980};   // bell_r2_mf.c:131-149
static const int k_r2_back_freq[6] = {1140, 1020, 900, 780, 660, 540};      // bell_r2_mf.c:151-169
static const char k_r2_keys[] = "1234567890BCDEF";                      // bell_r2_mf.c:172

static void pair_of(int k, int &lo, int &hi)
{
    // k-th two-of-six combination in the order the reference lists them: (0,1) (0,2) (1,2) (0,3) (1,3) (2,3) ...
    int at = 0;
    for (hi = 1;  hi < 6;  hi++)
    {
        for (lo = 0;  lo < hi;  lo++)
        {
            if (at++ == k)
                return;
        }
    }
    lo = hi = 0;
}

static void build_digit_table(TxDigitTable *t, int kind)
{
    int32_t w[13];

    memset(t, 0, sizeof(*t));
    if (kind == TXK_DTMF)
    {
        // dtmf_tx_initialise(), dtmf.c:522-547: -10 dBm0, 50 ms on, 55 ms off (levels and timing are then
        // overridden per channel, dtmf.c:577-580)
        t->n = 16;
        memcpy(t->keys, k_dtmf_keys, 17);
        for (int k = 0;  k < 16;  k++)
        {
            spg_make_tone_descriptor(w, k_dtmf_row[k >> 2], -10, k_dtmf_col[k & 3], -10, 50, 55, 0, 0, 0);
            t->rate[k][0] = w[0];
            t->rate[k][1] = w[1];
            memcpy(&t->gain[k][0], &w[4], 8);
            t->on[k] = w[8];
            t->off[k] = w[9];
        }
    }
    else if (kind == TXK_BELL_MF)
    {
        // bell_mf_gen_init(), bell_r2_mf.c:278-304: -7 dBm0 each, 68/68 ms, KP ('*') 100 ms on
        t->n = 15;
        memcpy(t->keys, k_bell_keys, 16);
        for (int k = 0;  k < 15;  k++)
        {
            int lo, hi;
            pair_of(k, lo, hi);
            spg_make_tone_descriptor(w, k_bell_freq[lo], -7, k_bell_freq[hi], -7, (k_bell_keys[k] == '*')  ?  100  :  68, 68, 0, 0, 0);
            t->rate[k][0] = w[0];
            t->rate[k][1] = w[1];
            memcpy(&t->gain[k][0], &w[4], 8);
            t->on[k] = w[8];
            t->off[k] = w[9];
        }
    }
}

static int launch_cfg(int n, int *blocks)
{
    *blocks = (n + 255)/256;
    return 256;
}

extern "C" {

int spangpu_txbank_create(spangpu_txbank_t **out, int device, int kind, int n_channels)
{
    if (out == NULL  ||  n_channels <= 0  ||  kind < SPANGPU_TX_TONE_GEN  ||  kind > SPANGPU_TX_R2_MF_BACK)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *out = NULL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess  ||  count <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    if (device < 0  ||  device >= count)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "device out of range");
    TX_TRY(hipSetDevice(device));
    spangpu_txbank_s *b = (spangpu_txbank_s *) calloc(1, sizeof(*b));
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    b->device = device;
    b->kind = kind;
    b->n_ch = n_channels;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess)
    {
        free(b);
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipStreamCreate failed");
    }
    b->own_stream = true;
    const size_t words = (size_t) kTxWords*n_channels;
    if (hipMalloc(&b->st, words*sizeof(int32_t)) != hipSuccess
        ||  hipMalloc(&b->sine, 2048*sizeof(float)) != hipSuccess
        ||  hipMalloc(&b->d_lens, (size_t) n_channels*sizeof(int32_t)) != hipSuccess
        ||  hipMalloc(&b->d_put_lens, (size_t) n_channels*sizeof(int32_t)) != hipSuccess
        ||  hipMalloc(&b->d_put_res, (size_t) n_channels*sizeof(int32_t)) != hipSuccess)
    {
        spangpu_txbank_destroy(b);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "allocation of the sender bank failed");
    }
    float sine[2048];
    spg_make_sine_table(sine);
    build_digit_table(&b->dig, kind);
    // Initial state: xxx_tx_init() leaves every generator idle (dtmf.c:640-660, bell_r2_mf.c:358-381,
    // bell_r2_mf.c:430-487) -- all words zero except section = -1 and the DTMF defaults.
    int32_t *host = (int32_t *) calloc(words, sizeof(int32_t));
    if (host == NULL)
    {
        spangpu_txbank_destroy(b);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    }
    const float lvl = spg_dds_scaling_dbm0f(-10.0f);
    int32_t lvl_bits;
    memcpy(&lvl_bits, &lvl, 4);
    for (int c = 0;  c < n_channels;  c++)
    {
        host[(size_t) TX_SECTION*n_channels + c] = -1;
        if (kind == TXK_DTMF)
        {
            host[(size_t) TX_LOW*n_channels + c] = lvl_bits;
            host[(size_t) TX_HIGH*n_channels + c] = lvl_bits;
            host[(size_t) TX_ON*n_channels + c] = 50*8000/1000;
            host[(size_t) TX_OFF*n_channels + c] = 55*8000/1000;
        }
    }
    hipError_t e = hipMemcpy(b->st, host, words*sizeof(int32_t), hipMemcpyHostToDevice);
    free(host);
    if (e == hipSuccess)
        e = hipMemcpy(b->sine, sine, sizeof(sine), hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        spangpu_txbank_destroy(b);
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    }
    *out = b;
    return SPANGPU_OK;
}

void spangpu_txbank_destroy(spangpu_txbank_t *b)
{
    if (b == NULL)
        return;
    (void) hipSetDevice(b->device);
    if (b->stream)
        (void) hipStreamSynchronize(b->stream);
    (void) hipFree(b->st);
    (void) hipFree(b->sine);
    (void) hipFree(b->d_pcm);
    (void) hipFree(b->d_lens);
    (void) hipFree(b->d_digits);
    (void) hipFree(b->d_put_lens);
    (void) hipFree(b->d_put_res);
    if (b->own_stream  &&  b->stream)
        (void) hipStreamDestroy(b->stream);
    free(b);
}

int spangpu_txbank_channels(const spangpu_txbank_t *b) { return b  ?  b->n_ch  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_txbank_state_words(void) { return kTxWords; }

int spangpu_txbank_set_stream(spangpu_txbank_t *b, void *stream)
{
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    TX_TRY(hipSetDevice(b->device));
    TX_TRY(hipStreamSynchronize(b->stream));
    if (b->own_stream)
        (void) hipStreamDestroy(b->stream);
    b->stream = (hipStream_t) stream;
    b->own_stream = false;
    return SPANGPU_OK;
}

int spangpu_txbank_sync(spangpu_txbank_t *b)
{
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    TX_TRY(hipSetDevice(b->device));
    TX_TRY(hipStreamSynchronize(b->stream));
    return SPANGPU_OK;
}

static int check_range(const spangpu_txbank_s *b, int first, int n)
{
    if (b == NULL  ||  first < 0  ||  n <= 0  ||  first > b->n_ch - n)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "channel range outside the bank");
    return SPANGPU_OK;
}

int spangpu_txbank_tone(spangpu_txbank_t *b, int first, int n, const spangpu_tone_desc_t *d)
{
    int rc;
    if ((rc = check_range(b, first, n)) != SPANGPU_OK)
        return rc;
    if (d == NULL  ||  b->kind != TXK_TONE_GEN)
        return spangpu_set_error(SPANGPU_ERR_STATE, "spangpu_txbank_tone() is for SPANGPU_TX_TONE_GEN banks");
    TxDescriptor td;
    spg_make_tone_descriptor(td.w, d->f1, d->l1, d->f2, d->l2, d->d1, d->d2, d->d3, d->d4, d->repeat);
    // tone_gen() with nothing but zero-length sections and repeat set never returns in the reference
    if (td.w[8] == 0  &&  td.w[12] != 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "a repeating cadence needs a first section of at least 1 ms");
    td.r2digit = -1;
    td.load = 1;
    TX_TRY(hipSetDevice(b->device));
    int blocks;
    const int threads = launch_cfg(n, &blocks);
    hipLaunchKernelGGL(tx_load_descriptor_kernel, dim3(blocks), dim3(threads), 0, b->stream, b->st, b->n_ch, first, first + n, td);
    TX_TRY(hipGetLastError());
    return SPANGPU_OK;
}

int spangpu_txbank_set_level(spangpu_txbank_t *b, int first, int n, int level, int twist)
{
    int rc;
    if ((rc = check_range(b, first, n)) != SPANGPU_OK)
        return rc;
    if (b->kind != TXK_DTMF)
        return spangpu_set_error(SPANGPU_ERR_STATE, "not a DTMF sender bank");
    // dtmf_tx_set_level(), dtmf.c:621-625
    const float lo = spg_dds_scaling_dbm0f((float) level);
    const float hi = spg_dds_scaling_dbm0f((float) (level + twist));
    int32_t lo_bits, hi_bits;
    memcpy(&lo_bits, &lo, 4);
    memcpy(&hi_bits, &hi, 4);
    TX_TRY(hipSetDevice(b->device));
    int blocks;
    const int threads = launch_cfg(n, &blocks);
    hipLaunchKernelGGL(tx_set_words_kernel, dim3(blocks), dim3(threads), 0, b->stream, b->st, b->n_ch, first, first + n,
                       (int) TX_LOW, lo_bits, (int) TX_HIGH, hi_bits);
    TX_TRY(hipGetLastError());
    return SPANGPU_OK;
}

int spangpu_txbank_set_timing(spangpu_txbank_t *b, int first, int n, int on_time, int off_time)
{
    int rc;
    if ((rc = check_range(b, first, n)) != SPANGPU_OK)
        return rc;
    if (b->kind != TXK_DTMF)
        return spangpu_set_error(SPANGPU_ERR_STATE, "not a DTMF sender bank");
    // dtmf_tx_set_timing(), dtmf.c:628-633
    const int on = ((on_time >= 0)  ?  on_time  :  50)*8000/1000;
    const int off = ((off_time >= 0)  ?  off_time  :  55)*8000/1000;
    TX_TRY(hipSetDevice(b->device));
    int blocks;
    const int threads = launch_cfg(n, &blocks);
    hipLaunchKernelGGL(tx_set_words_kernel, dim3(blocks), dim3(threads), 0, b->stream, b->st, b->n_ch, first, first + n,
                       (int) TX_ON, on, (int) TX_OFF, off);
    TX_TRY(hipGetLastError());
    return SPANGPU_OK;
}

static int put_r2(spangpu_txbank_s *b, int first, int n, char digit)
{
    // r2_mf_tx_put(), bell_r2_mf.c:417-428
    TxDescriptor td;
    memset(&td, 0, sizeof(td));
    const char *at = digit  ?  strchr(k_r2_keys, digit)  :  NULL;
    if (at)
    {
        const int *f = (b->kind == TXK_R2_FWD)  ?  k_r2_fwd_freq  :  k_r2_back_freq;
        int lo, hi;
        pair_of((int) (at - k_r2_keys), lo, hi);
        // bell_r2_mf.c:131-169,447-477: -11 dBm0 each, a 1 ms section repeated for ever
        spg_make_tone_descriptor(td.w, f[lo], -11, f[hi], -11, 1, 0, 0, 0, 1);
        td.load = 1;
        td.r2digit = (unsigned char) digit;
    }
    else
    {
        td.load = 0;
        td.r2digit = 0;
    }
    TX_TRY(hipSetDevice(b->device));
    int blocks;
    const int threads = launch_cfg(n, &blocks);
    hipLaunchKernelGGL(tx_load_descriptor_kernel, dim3(blocks), dim3(threads), 0, b->stream, b->st, b->n_ch, first, first + n, td);
    TX_TRY(hipGetLastError());
    return 0;
}

static int put_common(spangpu_txbank_s *b, int first, int n, const char *digits, int dstride, const int *lens, int len,
                      int *results)
{
    const size_t bytes = lens  ?  (size_t) n*dstride  :  (size_t) len;
    TX_TRY(hipSetDevice(b->device));
    if (bytes > b->digits_cap)
    {
        TX_TRY(hipStreamSynchronize(b->stream));
        (void) hipFree(b->d_digits);
        b->d_digits = NULL;
        b->digits_cap = 0;
        if (hipMalloc(&b->d_digits, bytes) != hipSuccess)
            return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "digit staging");
        b->digits_cap = bytes;
    }
    if (bytes)
        TX_TRY(hipMemcpyAsync(b->d_digits, digits, bytes, hipMemcpyHostToDevice, b->stream));
    if (lens)
        TX_TRY(hipMemcpyAsync(b->d_put_lens, lens, (size_t) n*sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    int blocks;
    const int threads = launch_cfg(n, &blocks);
    hipLaunchKernelGGL(tx_put_kernel, dim3(blocks), dim3(threads), 0, b->stream, b->st, b->n_ch, first, first + n,
                       (const uint8_t *) b->d_digits, lens  ?  dstride  :  0, lens  ?  (const int32_t *) b->d_put_lens  :  NULL,
                       len, b->d_put_res);
    TX_TRY(hipGetLastError());
    int *res = results;
    int *tmp = NULL;
    if (res == NULL)
    {
        if ((tmp = (int *) malloc((size_t) n*sizeof(int))) == NULL)
            return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "malloc");
        res = tmp;
    }
    hipError_t e = hipMemcpyAsync(res, b->d_put_res, (size_t) n*sizeof(int32_t), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(b->stream);       // also: the caller's digit buffers are only borrowed
    int worst = 0;
    if (e == hipSuccess)
    {
        for (int i = 0;  i < n;  i++)
            worst = (res[i] > worst)  ?  res[i]  :  worst;
    }
    free(tmp);
    if (e != hipSuccess)
        return spangpu_set_error(SPANGPU_ERR_HIP, "digit upload failed");
    return worst;
}

int spangpu_txbank_put(spangpu_txbank_t *b, int first, int n, const char *digits, int len)
{
    int rc;
    if ((rc = check_range(b, first, n)) != SPANGPU_OK)
        return rc;
    if (digits == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null digits");
    if (b->kind == TXK_R2_FWD  ||  b->kind == TXK_R2_BACK)
        return put_r2(b, first, n, (len == 0)  ?  '\0'  :  digits[0]);
    if (b->kind == TXK_TONE_GEN)
        return spangpu_set_error(SPANGPU_ERR_STATE, "a tone_gen bank has no digit queue");
    if (len < 0)
        len = (int) strlen(digits);
    if (len == 0)
        return 0;
    return put_common(b, first, n, digits, 0, NULL, len, NULL);
}

int spangpu_txbank_put_each(spangpu_txbank_t *b, int first, int n, const char *digits, int stride, const int *lens, int *results)
{
    int rc;
    if ((rc = check_range(b, first, n)) != SPANGPU_OK)
        return rc;
    if (digits == NULL  ||  lens == NULL  ||  stride <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (b->kind != TXK_DTMF  &&  b->kind != TXK_BELL_MF)
        return spangpu_set_error(SPANGPU_ERR_STATE, "this bank has no digit queue");
    for (int i = 0;  i < n;  i++)
    {
        if (lens[i] < 0  ||  lens[i] > stride)
            return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "lens[i] outside 0..stride");
    }
    return put_common(b, first, n, digits, stride, lens, 0, results);
}

int spangpu_txbank_tx(spangpu_txbank_t *b, int mem_kind, int16_t *pcm, long long stride, int samples, int *lens)
{
    if (b == NULL  ||  pcm == NULL  ||  samples < 0  ||  stride < samples)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (mem_kind != SPANGPU_MEM_HOST  &&  mem_kind != SPANGPU_MEM_DEVICE)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    if (samples == 0)
        return SPANGPU_OK;
    TX_TRY(hipSetDevice(b->device));
    TxLaunch L;
    memset(&L, 0, sizeof(L));
    L.st = b->st;
    L.sine = b->sine;
    L.n_ch = b->n_ch;
    L.samples = samples;
    L.kind = b->kind;
    L.dig = b->dig;
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        const size_t need = (size_t) ((samples + 7) & ~7);
        if (need > b->pcm_cap)
        {
            TX_TRY(hipStreamSynchronize(b->stream));
            (void) hipFree(b->d_pcm);
            b->d_pcm = NULL;
            b->pcm_cap = 0;
            if (hipMalloc(&b->d_pcm, need*b->n_ch*sizeof(int16_t)) != hipSuccess)
                return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "pcm staging");
            b->pcm_cap = need;
        }
        L.pcm = b->d_pcm;
        L.stride = (long long) b->pcm_cap;
        L.lens = b->d_lens;
    }
    else
    {
        L.pcm = pcm;
        L.stride = stride;
        L.lens = lens;
    }
    // 16 channels per wave, four waves per workgroup: even a small bank puts several waves on every SIMD,
    // and the 8 KB sine table copy is shared by 64 channels
    hipLaunchKernelGGL(tx_bank_kernel<kTxChannelsPerWave>, dim3((b->n_ch + kTxChannelsPerWave*kTxWaves - 1)/(kTxChannelsPerWave*kTxWaves)),
                       dim3(64*kTxWaves), 0, b->stream, L);
    TX_TRY(hipGetLastError());
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        TX_TRY(hipMemcpy2DAsync(pcm, (size_t) stride*sizeof(int16_t), b->d_pcm, b->pcm_cap*sizeof(int16_t),
                                (size_t) samples*sizeof(int16_t), b->n_ch, hipMemcpyDeviceToHost, b->stream));
        if (lens)
            TX_TRY(hipMemcpyAsync(lens, b->d_lens, (size_t) b->n_ch*sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
        TX_TRY(hipStreamSynchronize(b->stream));
    }
    return SPANGPU_OK;
}

int spangpu_txbank_get_state(spangpu_txbank_t *b, int channel, int32_t *words)
{
    if (b == NULL  ||  words == NULL  ||  channel < 0  ||  channel >= b->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    TX_TRY(hipSetDevice(b->device));
    TX_TRY(hipMemcpy2DAsync(words, sizeof(int32_t), b->st + channel, (size_t) b->n_ch*sizeof(int32_t), sizeof(int32_t),
                            kTxWords, hipMemcpyDeviceToHost, b->stream));
    TX_TRY(hipStreamSynchronize(b->stream));
    return SPANGPU_OK;
}

}   // extern "C"
