/*
 * shim_prims.c -- libspangpu_prims.so (NOT part of libspangpu.so: include/spangpu_prims.h says why): the receivers' inner
 * primitives under their spandsp names, for a caller that links them by name:
 *   vec_dot_prodf(), vec_lmsf(), cvec_dot_prodf(), cvec_lmsf()     src/vector_float.c:890-900,982-992   src/complex_vector_float.c:137-150,201-212
 *   periodogram(), _prepare(), _apply(), _generate_coeffs(), _generate_phase_offset(), _freq_error()    src/tone_detect.c:208-312
 *   fixed_sqrt32()   src/math_fixed.c:158-169      dds_lookup_complexf(), dds_complexf(), dds_advancef()   src/dds_float.c:2135-2187
 *   vec_circular_dot_prodf(), vec_circular_lmsf()          src/spandsp/vector_float.h:184,188    src/vector_float.c:932-939,996-1000
 *   cvec_circular_dot_prodf(), cvec_circular_lmsf()        src/spandsp/complex_vector_float.h:159,163   src/complex_vector_float.c:187-196,215-219
 *   power_meter_init/_release/_free/_damping/_update/_rx/_current   src/spandsp/power_meter.h:62-94      src/power_meter.c:44-113
 *   godard_ted_make_descriptor/_free_descriptor/_init/_release/_free/_correction/_rx/_per_baud   src/spandsp/godard.h:85-118   src/godard.c:70-249
 * Each is one item through the batched entry point of csrc/prim_api.hip (a launch per call: the plumbing form, as a
 * one-channel receiver object is -- the receivers themselves run these fused in their kernels, and a caller with many
 * items uses the *_batch entry points).  The arithmetic is the device's, in the reference's order of operations; there is
 * no host implementation behind these names: without a HIP device the float results are NaN, the power meter reads
 * INT32_MIN, and spangpu_last_error() says why.  The device is SPANGPU_DEVICE (environment, default 0).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "spangpu_prims.h"

static int prim_device(void)
{
    const char *e = getenv("SPANGPU_DEVICE");
    return (e  &&  *e)  ?  atoi(e)  :  0;
}

float vec_circular_dot_prodf(const float x[], const float y[], int n, int pos)
{
    float z = 0.0f;
    int32_t p = pos;

    if (spangpu_vec_circular_dot_prodf_batch(prim_device(), x, 0, y, 0, &p, &z, 1, n, SPANGPU_MEM_HOST) < 0)
        return NAN;
    return z;
}

void vec_circular_lmsf(const float x[], float y[], int n, int pos, float error)
{
    int32_t p = pos;

    (void) spangpu_vec_circular_lmsf_batch(prim_device(), x, 0, y, n, &p, &error, 1, n, SPANGPU_MEM_HOST);
}

complexf_t cvec_circular_dot_prodf(const complexf_t x[], const complexf_t y[], int n, int pos)
{
    complexf_t z;
    int32_t p = pos;

    z.re = 0.0f;
    z.im = 0.0f;
    if (spangpu_cvec_circular_dot_prodf_batch(prim_device(), (const float *) x, 0, (const float *) y, 0, &p, (float *) &z, 1, n, SPANGPU_MEM_HOST) < 0)
    {
        z.re = NAN;
        z.im = NAN;
    }
    return z;
}

void cvec_circular_lmsf(const complexf_t x[], complexf_t y[], int n, int pos, const complexf_t *error)
{
    int32_t p = pos;

    if (error == NULL)
        return;
    (void) spangpu_cvec_circular_lmsf_batch(prim_device(), (const float *) x, 0, (float *) y, n, &p, (const float *) error, 1, n, SPANGPU_MEM_HOST);
}

/* power_meter.c:44-63: the caller's storage is used when it gives some (the state is two host words; the arithmetic of
   an update is the device's) */
power_meter_t *power_meter_init(power_meter_t *s, int shift)
{
    if (s == NULL)
    {
        if ((s = (power_meter_t *) malloc(sizeof(*s))) == NULL)
            return NULL;
    }
    s->shift = shift;
    s->reading = 0;
    return s;
}

int power_meter_release(power_meter_t *s)
{
    (void) s;
    return 0;
}

int power_meter_free(power_meter_t *s)
{
    if (s)
        free(s);
    return 0;
}

power_meter_t *power_meter_damping(power_meter_t *s, int shift)
{
    s->shift = shift;
    return s;
}

int32_t power_meter_update(power_meter_t *s, int16_t amp)
{
    int32_t sh = s->shift;

    if (spangpu_power_meter_update_batch(prim_device(), &amp, 1, &s->reading, &sh, 1, 1, SPANGPU_MEM_HOST) < 0)
        return INT32_MIN;
    return s->reading;
}

/* power_meter.c:72-80 (returns 0, as the reference does) */
int32_t power_meter_rx(power_meter_t *s, int16_t amp[], int len)
{
    int32_t sh = s->shift;

    if (len > 0)
        (void) spangpu_power_meter_update_batch(prim_device(), amp, len, &s->reading, &sh, 1, len, SPANGPU_MEM_HOST);
    return 0;
}

int32_t power_meter_current(power_meter_t *s)
{
    return s->reading;
}

/* ---- the Godard timing error detector (godard.c).  The descriptor is table making: its seven coefficients are formed on the
   host with the expressions of godard.c:93-104.  The detector's filters and its per-baud decision run on the device, one
   item through spangpu_godard_ted_rx_batch() / spangpu_godard_ted_per_baud_batch() a call; the state words those take are
   this struct's, in place. ---- */
godard_ted_descriptor_t *godard_ted_make_descriptor(godard_ted_descriptor_t *s, float sample_rate, float baud_rate, float carrier_freq,
                                                    float alpha, float coarse_trigger, float fine_trigger, int coarse_step, int fine_step)
{
    float low_edge;
    float high_edge;

    if (s == NULL)
    {
        if ((s = (godard_ted_descriptor_t *) malloc(sizeof(*s))) == NULL)
            return NULL;
    }
    memset(s, 0, sizeof(*s));
    low_edge = 2.0f*M_PI*(carrier_freq - baud_rate/2.0f)/sample_rate;
    high_edge = 2.0f*M_PI*(carrier_freq + baud_rate/2.0f)/sample_rate;
    s->low_band_edge_coeff[0] = 2.0f*alpha*cosf(low_edge);
    s->low_band_edge_coeff[1] = -alpha*alpha;
    s->low_band_edge_coeff[2] = -alpha*sinf(low_edge);
    s->high_band_edge_coeff[0] = 2.0f*alpha*cosf(high_edge);
    s->high_band_edge_coeff[1] = -alpha*alpha;
    s->high_band_edge_coeff[2] = -alpha*sinf(high_edge);
    s->mixed_band_edges_coeff_3 = -alpha*alpha*(sinf(high_edge)*cosf(low_edge) - sinf(low_edge)*cosf(high_edge));
    s->coarse_trigger = coarse_trigger;
    s->fine_trigger = fine_trigger;
    s->coarse_step = coarse_step;
    s->fine_step = fine_step;
    return s;
}

int godard_ted_free_descriptor(godard_ted_descriptor_t *s)
{
    free(s);
    return 0;
}

int godard_ted_correction(godard_ted_state_t *s)
{
    return s->total_baud_timing_correction;
}

/* the twelve descriptor words the batched entry points read: the struct's eleven and a zero */
static void godard_desc_words(const godard_ted_state_t *s, uint32_t w[12])
{
    memcpy(w, &s->desc, 11*sizeof(uint32_t));
    w[11] = 0;
}

void godard_ted_rx(godard_ted_state_t *s, float sample)
{
    uint32_t w[12];

    godard_desc_words(s, w);
    (void) spangpu_godard_ted_rx_batch(prim_device(), (uint32_t *) s->low_band_edge, w, 0, &sample, 1, 1, 1, SPANGPU_MEM_HOST);
}

int godard_ted_per_baud(godard_ted_state_t *s)
{
    uint32_t w[12];
    int32_t corr = 0;

    godard_desc_words(s, w);
    if (spangpu_godard_ted_per_baud_batch(prim_device(), (uint32_t *) s->low_band_edge, w, 0, &corr, 1, SPANGPU_MEM_HOST) < 0)
        return 0;
    return corr;
}

godard_ted_state_t *godard_ted_init(godard_ted_state_t *s, const godard_ted_descriptor_t *desc)
{
    if (s == NULL)
    {
        if ((s = (godard_ted_state_t *) malloc(sizeof(*s))) == NULL)
            return NULL;
    }
    memset(s, 0, sizeof(*s));
    s->desc = *desc;
    return s;
}

int godard_ted_release(godard_ted_state_t *s)
{
    (void) s;
    return 0;
}

int godard_ted_free(godard_ted_state_t *s)
{
    free(s);
    return 0;
}


/* ---- the plain (non-circular) forms: the circular entry points at position 0 -- vec_circular_dot_prodf(x, y, n, 0) is
   vec_dot_prodf(x, y, n) plus an empty second sum (vector_float.c:932-939), and a sum that started at +0.0f is never -0.0f ---- */
float vec_dot_prodf(const float x[], const float y[], int n)
{
    return vec_circular_dot_prodf(x, y, n, 0);
}

void vec_lmsf(const float x[], float y[], int n, float error)
{
    vec_circular_lmsf(x, y, n, 0, error);
}

complexf_t cvec_dot_prodf(const complexf_t x[], const complexf_t y[], int n)
{
    return cvec_circular_dot_prodf(x, y, n, 0);
}

void cvec_lmsf(const complexf_t x[], complexf_t y[], int n, const complexf_t *error)
{
    cvec_circular_lmsf(x, y, n, 0, error);
}

/* ---- periodograms (tone_detect.c:208-312).  The coefficient set and the phase offset are table making: formed on the host
   with the reference's expressions (libm's cosf / sinf, as there).  The sums and the frequency error are the device's. ---- */
complexf_t periodogram(const complexf_t coeffs[], const complexf_t amp[], int len)
{
    complexf_t z;

    z.re = 0.0f;
    z.im = 0.0f;
    if (len < 2)
        return z;
    if (spangpu_periodogram_batch(prim_device(), (const float *) coeffs, 0, (const float *) amp, 0, (float *) &z, 1, len, SPANGPU_MEM_HOST) < 0)
    {
        z.re = NAN;
        z.im = NAN;
    }
    return z;
}

int periodogram_prepare(complexf_t sum[], complexf_t diff[], const complexf_t amp[], int len)
{
    if (len < 2)
        return 0;
    if (spangpu_periodogram_prepare_batch(prim_device(), (const float *) amp, 0, (float *) sum, (float *) diff, 1, len, SPANGPU_MEM_HOST) < 0)
    {
        for (int i = 0;  i < len/2;  i++)
            sum[i].re = sum[i].im = diff[i].re = diff[i].im = NAN;
    }
    return len/2;
}

complexf_t periodogram_apply(const complexf_t coeffs[], const complexf_t sum[], const complexf_t diff[], int len)
{
    complexf_t z;

    z.re = 0.0f;
    z.im = 0.0f;
    if (len < 2)
        return z;
    if (spangpu_periodogram_apply_batch(prim_device(), (const float *) coeffs, 0, (const float *) sum, (const float *) diff, (float *) &z, 1, len,
                                        SPANGPU_MEM_HOST) < 0)
    {
        z.re = NAN;
        z.im = NAN;
    }
    return z;
}

int periodogram_generate_coeffs(complexf_t coeffs[], float freq, int sample_rate, int window_len)
{
    float window;
    float sum;
    float x;
    int i;

    /* tone_detect.c:258-283: half a Hamming window times the matched phasor, then scaled for unity gain over the whole window */
    sum = 0.0f;
    for (i = 0;  i < window_len/2;  i++)
    {
        window = 0.53836f - 0.46164f*cosf(2.0f*3.1415926535f*i/(window_len - 1.0f));
        x = (i - window_len/2.0f + 0.5f)*freq*2.0f*3.1415926535f/sample_rate;
        coeffs[i].re = cosf(x)*window;
        coeffs[i].im = -sinf(x)*window;
        sum += window;
    }
    sum = 1.0f/(2.0f*sum);
    for (i = 0;  i < window_len/2;  i++)
    {
        coeffs[i].re *= sum;
        coeffs[i].im *= sum;
    }
    return window_len/2;
}

float periodogram_generate_phase_offset(complexf_t *offset, float freq, int sample_rate, int interval)
{
    float x;

    /* tone_detect.c:286-296 */
    x = 2.0f*3.1415926535f*(float) interval/(float) sample_rate;
    offset->re = cosf(freq*x);
    offset->im = sinf(freq*x);
    return 1.0f/x;
}

float periodogram_freq_error(const complexf_t *phase_offset, float scale, const complexf_t *last_result, const complexf_t *result)
{
    float err = NAN;

    if (spangpu_periodogram_freq_error_batch(prim_device(), (const float *) phase_offset, scale, (const float *) last_result, (const float *) result,
                                             &err, 1, SPANGPU_MEM_HOST) < 0)
        return NAN;
    return err;
}

/* ---- SURVEY 8(a) a19's helpers by name: the receivers' tables and device functions, one item a call ---- */
uint16_t fixed_sqrt32(uint32_t x)
{
    uint16_t r = 0;

    if (spangpu_fixed_sqrt32_batch(prim_device(), &x, &r, 1, SPANGPU_MEM_HOST) < 0)
        return 0;
    return r;
}

complexf_t dds_lookup_complexf(uint32_t phase)
{
    complexf_t z;
    int32_t rate = 0;

    if (spangpu_dds_complexf_batch(prim_device(), &phase, &rate, (float *) &z, 1, 1, SPANGPU_MEM_HOST) < 0)
    {
        z.re = NAN;
        z.im = NAN;
    }
    return z;
}

complexf_t dds_complexf(uint32_t *phase_acc, int32_t phase_rate)
{
    complexf_t z;

    if (spangpu_dds_complexf_batch(prim_device(), phase_acc, &phase_rate, (float *) &z, 1, 1, SPANGPU_MEM_HOST) < 0)
    {
        z.re = NAN;
        z.im = NAN;
    }
    return z;
}

void dds_advancef(uint32_t *phase_acc, int32_t phase_rate)
{
    *phase_acc += (uint32_t) phase_rate;        /* dds_float.c:2153-2156: an addition, on the host as on the device */
}
