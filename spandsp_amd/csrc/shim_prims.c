/*
 * shim_prims.c -- the receivers' inner primitives under their spandsp names, for a caller that links them by name:
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

#include "spangpu_spandsp.h"

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
