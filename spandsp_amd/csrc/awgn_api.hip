// awgn_api.hip -- C ABI of the noise source banks (include/spangpu.h, "noise source banks"): batched
// awgn_init_dbm0() / awgn().  Device code: awgn_dev.hpp.  No CPU implementation of the generator exists behind
// these entry points; the host only seeds the per-channel state (integer LCG steps and pow(), as awgn.c:82-146).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"
#include "awgn_dev.hpp"

using namespace spg;

extern "C" int spangpu_set_error(int code, const char *msg);

#define AWGN_TRY(expr)                                                                      \
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

struct spangpu_awgn_s
{
    int device;
    int n_ch;
    hipStream_t stream;
    bool own_stream;
    int32_t *st;
    int16_t *d_amp;
    size_t amp_cap;
};

static void put_double(int32_t w[], double v)
{
    memcpy(w, &v, sizeof(v));           // little endian: low word first
}

// ran_init() + awgn_init_dbov(level - DBM0_MAX_POWER): awgn.c:82-105,127-152
static void seed_words(int32_t w[kAwgnWords], int idum, float level)
{
    if (idum < 0)
        idum = -idum;
    int ix1 = (54773 + idum)%259200;
    ix1 = (7141*ix1 + 54773)%259200;
    int ix2 = ix1%134456;
    ix1 = (7141*ix1 + 54773)%259200;
    const int ix3 = ix1%243000;
    for (int j = 0;  j < 97;  j++)
    {
        ix1 = (7141*ix1 + 54773)%259200;
        ix2 = (8121*ix2 + 28411)%134456;
        put_double(&w[AW_R + 2*j], ((double) ix1 + (double) ix2*(1.0/134456.0))*(1.0/259200.0));
    }
    level -= (3.14f + 3.02f);
    put_double(&w[AW_RMS], pow(10.0, level/20.0)*32768.0);
    put_double(&w[AW_AMP2], 0.0);
    w[AW_ODD] = 1;
    w[AW_IX1] = ix1;
    w[AW_IX2] = ix2;
    w[AW_IX3] = ix3;
}

extern "C" {

int spangpu_awgn_create(spangpu_awgn_t **out, int device, int n_channels, const int32_t seeds[], const float levels_dbm0[])
{
    if (out == NULL  ||  n_channels <= 0  ||  seeds == NULL  ||  levels_dbm0 == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *out = NULL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess  ||  count <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    if (device < 0  ||  device >= count)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "device out of range");
    AWGN_TRY(hipSetDevice(device));
    spangpu_awgn_s *b = (spangpu_awgn_s *) calloc(1, sizeof(*b));
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    b->device = device;
    b->n_ch = n_channels;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess)
    {
        free(b);
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipStreamCreate failed");
    }
    b->own_stream = true;
    const size_t words = (size_t) kAwgnWords*n_channels;
    int32_t *host = (int32_t *) malloc(words*sizeof(int32_t));
    if (host == NULL
        ||  hipMalloc(&b->st, words*sizeof(int32_t)) != hipSuccess)
    {
        free(host);
        spangpu_awgn_destroy(b);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "allocation of the noise source bank failed");
    }
    int32_t one[kAwgnWords];
    for (int c = 0;  c < n_channels;  c++)
    {
        seed_words(one, seeds[c], levels_dbm0[c]);
        for (int k = 0;  k < kAwgnWords;  k++)
            host[(size_t) k*n_channels + c] = one[k];
    }
    hipError_t e = hipMemcpy(b->st, host, words*sizeof(int32_t), hipMemcpyHostToDevice);
    free(host);
    if (e != hipSuccess)
    {
        spangpu_awgn_destroy(b);
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    }
    *out = b;
    return SPANGPU_OK;
}

void spangpu_awgn_destroy(spangpu_awgn_t *b)
{
    if (b == NULL)
        return;
    (void) hipSetDevice(b->device);
    if (b->stream)
        (void) hipStreamSynchronize(b->stream);
    (void) hipFree(b->st);
    (void) hipFree(b->d_amp);
    if (b->own_stream  &&  b->stream)
        (void) hipStreamDestroy(b->stream);
    free(b);
}

int spangpu_awgn_channels(const spangpu_awgn_t *b) { return b  ?  b->n_ch  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_awgn_state_words(const spangpu_awgn_t *b) { return b  ?  kAwgnWords  :  SPANGPU_ERR_BAD_ARG; }

int spangpu_awgn_set_stream(spangpu_awgn_t *b, void *stream)
{
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    AWGN_TRY(hipSetDevice(b->device));
    AWGN_TRY(hipStreamSynchronize(b->stream));
    if (b->own_stream)
        (void) hipStreamDestroy(b->stream);
    b->stream = (hipStream_t) stream;
    b->own_stream = false;
    return SPANGPU_OK;
}

int spangpu_awgn_sync(spangpu_awgn_t *b)
{
    if (b == NULL)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    AWGN_TRY(hipSetDevice(b->device));
    AWGN_TRY(hipStreamSynchronize(b->stream));
    return SPANGPU_OK;
}

int spangpu_awgn_reinit(spangpu_awgn_t *b, int channel, int seed, float level_dbm0)
{
    if (b == NULL  ||  channel < 0  ||  channel >= b->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    AWGN_TRY(hipSetDevice(b->device));
    AWGN_TRY(hipStreamSynchronize(b->stream));
    int32_t one[kAwgnWords];
    seed_words(one, seed, level_dbm0);
    AWGN_TRY(hipMemcpy2D(b->st + channel, (size_t) b->n_ch*sizeof(int32_t), one, sizeof(int32_t), sizeof(int32_t),
                         kAwgnWords, hipMemcpyHostToDevice));
    return SPANGPU_OK;
}

int spangpu_awgn_tx(spangpu_awgn_t *b, int mem_kind, int16_t *amp, long long stride, int samples, int mix)
{
    if (b == NULL  ||  amp == NULL  ||  samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (mem_kind != SPANGPU_MEM_HOST  &&  mem_kind != SPANGPU_MEM_DEVICE)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    if (stride <= 0)
        stride = samples;
    if (stride < samples)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "stride < samples");
    AWGN_TRY(hipSetDevice(b->device));
    AwgnLaunch L;
    memset(&L, 0, sizeof(L));
    L.st = b->st;
    L.n_ch = b->n_ch;
    L.samples = samples;
    L.mix = mix  ?  1  :  0;
    const size_t bytes = (size_t) b->n_ch*(size_t) stride*sizeof(int16_t);
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        if (bytes > b->amp_cap)
        {
            AWGN_TRY(hipStreamSynchronize(b->stream));
            (void) hipFree(b->d_amp);
            b->d_amp = NULL;
            b->amp_cap = 0;
            if (hipMalloc(&b->d_amp, bytes) != hipSuccess)
                return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "sample staging buffer");
            b->amp_cap = bytes;
        }
        if (mix)
            AWGN_TRY(hipMemcpyAsync(b->d_amp, amp, bytes, hipMemcpyHostToDevice, b->stream));
        L.amp = b->d_amp;
    }
    else
    {
        L.amp = amp;
    }
    L.stride = stride;
    hipLaunchKernelGGL(awgn_bank_kernel, dim3((b->n_ch + 63)/64), dim3(64), (97*64 + 2*kLogTab)*sizeof(double), b->stream, L);
    AWGN_TRY(hipGetLastError());
    if (mem_kind == SPANGPU_MEM_HOST)
    {
        // the caller's buffer is only borrowed for this call
        AWGN_TRY(hipMemcpyAsync(amp, b->d_amp, bytes, hipMemcpyDeviceToHost, b->stream));
        AWGN_TRY(hipStreamSynchronize(b->stream));
    }
    return samples;
}

int spangpu_awgn_get_state(spangpu_awgn_t *b, int channel, int32_t *words)
{
    if (b == NULL  ||  words == NULL  ||  channel < 0  ||  channel >= b->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    AWGN_TRY(hipSetDevice(b->device));
    AWGN_TRY(hipStreamSynchronize(b->stream));
    AWGN_TRY(hipMemcpy2D(words, sizeof(int32_t), b->st + channel, (size_t) b->n_ch*sizeof(int32_t), sizeof(int32_t),
                         kAwgnWords, hipMemcpyDeviceToHost));
    return SPANGPU_OK;
}

}   // extern "C"
