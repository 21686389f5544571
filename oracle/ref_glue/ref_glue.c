/*
 * ref_glue.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin accessors linked into oracle/_ref/libspandsp_ref.so next to the REAL
 * reference objects (compiled from /root/reference/src where they lie).  This
 * file is our own code: it includes the reference's headers at build time so
 * the Python test harness never has to know the private struct layouts, and it
 * provides C-side event collectors so callbacks do not have to bounce through
 * ctypes.  Nothing here is part of the product.
 */
#include <stdlib.h>
#include <inttypes.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <stdbool.h>
#include <limits.h>

#include "spandsp/telephony.h"
#include "spandsp/alloc.h"
#include "spandsp/logging.h"
#include "spandsp/fast_convert.h"
#include "spandsp/queue.h"
#include "spandsp/complex.h"
#include "spandsp/dds.h"
#include "spandsp/tone_detect.h"
#include "spandsp/tone_generate.h"
#include "spandsp/super_tone_rx.h"
#include "spandsp/dtmf.h"
#include "spandsp/bell_r2_mf.h"
#include "spandsp/saturated.h"
#include "spandsp/dc_restore.h"
#include "spandsp/bit_operations.h"
#include "spandsp/echo.h"
#include "spandsp/async.h"
#include "spandsp/power_meter.h"
#include "spandsp/vector_float.h"
#include "spandsp/complex_vector_float.h"
#include "spandsp/godard.h"
#include "spandsp/v29rx.h"
#include "spandsp/v29tx.h"
#include "spandsp/v27ter_rx.h"
#include "spandsp/v27ter_tx.h"
#include "spandsp/v17rx.h"
#include "spandsp/v17tx.h"
#include "spandsp/awgn.h"

#include "spandsp/private/logging.h"
#include "spandsp/private/queue.h"
#include "spandsp/private/tone_generate.h"
#include "spandsp/private/dtmf.h"
#include "spandsp/private/bell_r2_mf.h"
#include "spandsp/private/super_tone_rx.h"
#include "spandsp/private/echo.h"
#include "spandsp/private/power_meter.h"
#include "spandsp/private/godard.h"
#include "spandsp/private/v29rx.h"
#include "spandsp/private/v27ter_rx.h"
#include "spandsp/private/v17rx.h"

#define GLUE __attribute__((visibility("default")))

/* ---- generic event collector ------------------------------------------- */
/* One record per callback invocation, in call order. */
typedef struct
{
    int32_t kind;       /* 1 = tone report (code, level, delay), 2 = digits string chunk, 3 = bit / status, 4 = segment */
    int32_t a;
    int32_t b;
    int32_t c;
} glue_event_t;

typedef struct
{
    glue_event_t *ev;
    int n;
    int cap;
    char *digits;       /* concatenation of everything delivered to a digits callback */
    int ndigits;
    int capdigits;
} glue_sink_t;

GLUE glue_sink_t *glue_sink_new(void)
{
    glue_sink_t *k = (glue_sink_t *) calloc(1, sizeof(*k));
    k->cap = 1024;
    k->ev = (glue_event_t *) malloc(sizeof(glue_event_t)*k->cap);
    k->capdigits = 1024;
    k->digits = (char *) malloc(k->capdigits);
    return k;
}

GLUE void glue_sink_free(glue_sink_t *k)
{
    free(k->ev);
    free(k->digits);
    free(k);
}

GLUE void glue_sink_clear(glue_sink_t *k)
{
    k->n = 0;
    k->ndigits = 0;
}

GLUE int glue_sink_count(glue_sink_t *k) { return k->n; }
GLUE const glue_event_t *glue_sink_events(glue_sink_t *k) { return k->ev; }
GLUE int glue_sink_ndigits(glue_sink_t *k) { return k->ndigits; }
GLUE const char *glue_sink_digits(glue_sink_t *k) { return k->digits; }

static void sink_push(glue_sink_t *k, int kind, int a, int b, int c)
{
    if (k->n == k->cap)
    {
        k->cap *= 2;
        k->ev = (glue_event_t *) realloc(k->ev, sizeof(glue_event_t)*k->cap);
    }
    k->ev[k->n].kind = kind;
    k->ev[k->n].a = a;
    k->ev[k->n].b = b;
    k->ev[k->n].c = c;
    k->n++;
}

static void sink_tone_report(void *user_data, int code, int level, int delay)
{
    sink_push((glue_sink_t *) user_data, 1, code, level, delay);
}

static void sink_digits(void *user_data, const char *digits, int len)
{
    glue_sink_t *k = (glue_sink_t *) user_data;
    while (k->ndigits + len + 1 > k->capdigits)
    {
        k->capdigits *= 2;
        k->digits = (char *) realloc(k->digits, k->capdigits);
    }
    memcpy(k->digits + k->ndigits, digits, len);
    k->ndigits += len;
    k->digits[k->ndigits] = '\0';
    sink_push(k, 2, len, 0, 0);
}

static void sink_put_bit(void *user_data, int bit)
{
    sink_push((glue_sink_t *) user_data, 3, bit, 0, 0);
}

static void sink_segment(void *user_data, int f1, int f2, int duration)
{
    sink_push((glue_sink_t *) user_data, 4, f1, f2, duration);
}

/* Function-pointer getters, so Python can hand the collectors to xxx_init(). */
GLUE void *glue_fn_tone_report(void) { return (void *) sink_tone_report; }
GLUE void *glue_fn_digits(void) { return (void *) sink_digits; }
GLUE void *glue_fn_put_bit(void) { return (void *) sink_put_bit; }
GLUE void *glue_fn_segment(void) { return (void *) sink_segment; }

/* ---- DTMF --------------------------------------------------------------- */
GLUE dtmf_rx_state_t *glue_dtmf_rx_new(glue_sink_t *k, int use_digits_cb, int use_realtime_cb)
{
    dtmf_rx_state_t *s = dtmf_rx_init(NULL, (use_digits_cb)  ?  sink_digits  :  NULL, k);
    if (use_realtime_cb)
        dtmf_rx_set_realtime_callback(s, sink_tone_report, k);
    return s;
}

/* out[0..7] v2 (row0..3, col0..3), out[8..15] v3, out[16] energy,
   iout: current_sample, duration, last_hit, in_digit, lost_digits, current_digits */
GLUE void glue_dtmf_rx_snapshot(dtmf_rx_state_t *s, float out[17], int32_t iout[6])
{
    int i;
    for (i = 0;  i < 4;  i++)
    {
        out[i] = s->row_out[i].v2;
        out[4 + i] = s->col_out[i].v2;
        out[8 + i] = s->row_out[i].v3;
        out[12 + i] = s->col_out[i].v3;
    }
    out[16] = s->energy;
    iout[0] = s->current_sample;
    iout[1] = s->duration;
    iout[2] = s->last_hit;
    iout[3] = s->in_digit;
    iout[4] = s->lost_digits;
    iout[5] = s->current_digits;
}

GLUE void glue_dtmf_rx_consts(dtmf_rx_state_t *s, float out[11])
{
    int i;
    for (i = 0;  i < 4;  i++)
    {
        out[i] = s->row_out[i].fac;
        out[4 + i] = s->col_out[i].fac;
    }
    out[8] = s->threshold;
    out[9] = s->normal_twist;
    out[10] = s->reverse_twist;
}

/* Batch driver for the CPU baseline: run dtmf_rx() over n channel objects, channel c
   reading amp + c*stride.  One C call per frame keeps Python out of the timed loop. */
GLUE void glue_dtmf_rx_batch(dtmf_rx_state_t **s, const int16_t *amp, int n, long long stride, int samples)
{
    int c;
    for (c = 0;  c < n;  c++)
        dtmf_rx(s[c], amp + c*stride, samples);
}

/* ---- Bell MF / R2 MF ------------------------------------------------------ */
GLUE bell_mf_rx_state_t *glue_bell_mf_rx_new(glue_sink_t *k, int use_digits_cb)
{
    return bell_mf_rx_init(NULL, (use_digits_cb)  ?  sink_digits  :  NULL, k);
}

GLUE void glue_bell_mf_rx_snapshot(bell_mf_rx_state_t *s, float out[18], int32_t iout[8])
{
    int i;
    for (i = 0;  i < 6;  i++)
    {
        out[i] = s->out[i].v2;
        out[6 + i] = s->out[i].v3;
        out[12 + i] = s->out[i].fac;
    }
    iout[0] = s->current_sample;
    for (i = 0;  i < 5;  i++)
        iout[1 + i] = s->hits[i];
    iout[6] = s->lost_digits;
    iout[7] = s->current_digits;
}

GLUE r2_mf_rx_state_t *glue_r2_mf_rx_new(glue_sink_t *k, int fwd, int use_cb)
{
    return r2_mf_rx_init(NULL, fwd, (use_cb)  ?  sink_tone_report  :  NULL, k);
}

GLUE void glue_r2_mf_rx_snapshot(r2_mf_rx_state_t *s, float out[18], int32_t iout[2])
{
    int i;
    for (i = 0;  i < 6;  i++)
    {
        out[i] = s->out[i].v2;
        out[6 + i] = s->out[i].v3;
        out[12 + i] = s->out[i].fac;
    }
    iout[0] = s->current_sample;
    iout[1] = s->current_digit;
}

/* ---- Super tone ----------------------------------------------------------- */
GLUE super_tone_rx_state_t *glue_super_tone_rx_new(super_tone_rx_descriptor_t *desc, glue_sink_t *k, int use_segment_cb)
{
    super_tone_rx_state_t *s = super_tone_rx_init(NULL, desc, sink_tone_report, k);
    if (s  &&  use_segment_cb)
        super_tone_rx_segment_callback(s, sink_segment);
    return s;
}

GLUE int glue_super_tone_desc_bins(super_tone_rx_descriptor_t *desc, float fac[64])
{
    int i;
    for (i = 0;  i < desc->monitored_frequencies;  i++)
        fac[i] = desc->desc[i].fac;
    return desc->monitored_frequencies;
}

/* iout: detected_tone, rotation, state[0].current_sample, then 11*(f1,f2,min_duration) */
GLUE void glue_super_tone_rx_snapshot(super_tone_rx_state_t *s, float *fout, int32_t iout[36])
{
    int i;
    int m = s->desc->monitored_frequencies;
    fout[0] = s->energy;
    for (i = 0;  i < m;  i++)
    {
        fout[1 + i] = s->state[i].v2;
        fout[1 + m + i] = s->state[i].v3;
    }
    iout[0] = s->detected_tone;
    iout[1] = s->rotation;
    iout[2] = (m > 0)  ?  s->state[0].current_sample  :  0;
    for (i = 0;  i < 11;  i++)
    {
        iout[3 + 3*i] = s->segments[i].f1;
        iout[4 + 3*i] = s->segments[i].f2;
        iout[5 + 3*i] = s->segments[i].min_duration;
    }
}

/* ---- Goertzel -------------------------------------------------------------- */
GLUE float glue_goertzel_fac(float freq, int samples)
{
    goertzel_descriptor_t d;
    make_goertzel_descriptor(&d, freq, samples);
    return d.fac;
}

GLUE goertzel_state_t *glue_goertzel_new(float freq, int samples)
{
    goertzel_descriptor_t d;
    make_goertzel_descriptor(&d, freq, samples);
    return goertzel_init(NULL, &d);
}

GLUE void glue_goertzel_snapshot(goertzel_state_t *s, float out[3], int32_t iout[2])
{
    out[0] = s->v2;
    out[1] = s->v3;
    out[2] = s->fac;
    iout[0] = s->samples;
    iout[1] = s->current_sample;
}

/* ---- Echo canceller --------------------------------------------------------- */
/* narrowband_detect() (echo.c:133-139) reads the FIR history up to index 255 whatever
   `taps` is.  Give the reference a zero-filled, padded heap through its own allocator hook
   (span_mem_allocators, alloc.c:142) so that those reads are deterministic (zero). */
static void *padded_alloc(size_t size)
{
    return calloc(1, size + 1024);
}

static void *padded_realloc(void *ptr, size_t size)
{
    return realloc(ptr, size + 1024);
}

GLUE void glue_install_padded_allocator(int on)
{
    if (on)
        span_mem_allocators(padded_alloc, padded_realloc, free, NULL, NULL);
    else
        span_mem_allocators(NULL, NULL, NULL, NULL, NULL);
}

/* Runs n samples through echo_can_update (optionally echo_can_hpf_tx first, as
   tests/echo_tests.c:577-594 does) and stores the clean signal. */
GLUE void glue_echo_run(echo_can_state_t *ec, const int16_t tx[], const int16_t rx[], int16_t clean[], int n, int use_hpf_tx)
{
    int i;
    int16_t t;
    for (i = 0;  i < n;  i++)
    {
        t = tx[i];
        if (use_hpf_tx)
            t = echo_can_hpf_tx(ec, t);
        clean[i] = echo_can_update(ec, t, rx[i]);
    }
}

GLUE int glue_echo_taps(echo_can_state_t *ec) { return ec->taps; }

/* iout layout documented in tests/refutil.py (ECHO_SNAPSHOT_FIELDS) */
GLUE void glue_echo_snapshot(echo_can_state_t *ec, int32_t iout[64], int32_t taps32[], int16_t taps16[], int16_t history[])
{
    int i;
    int j;
    int n = 0;
    for (i = 0;  i < 4;  i++)
        iout[n++] = ec->tx_power[i];
    for (i = 0;  i < 3;  i++)
        iout[n++] = ec->rx_power[i];
    iout[n++] = ec->clean_rx_power;
    iout[n++] = ec->rx_power_threshold;
    iout[n++] = ec->nonupdate_dwell;
    iout[n++] = ec->curr_pos;
    iout[n++] = ec->taps;
    iout[n++] = ec->tap_mask;
    iout[n++] = ec->adaption_mode;
    iout[n++] = ec->supp_test1;
    iout[n++] = ec->supp_test2;
    iout[n++] = ec->supp1;
    iout[n++] = ec->supp2;
    iout[n++] = ec->vad;
    iout[n++] = ec->cng;
    iout[n++] = ec->geigel_max;
    iout[n++] = ec->geigel_lag;
    iout[n++] = ec->dtd_onset;
    iout[n++] = ec->tap_set;
    iout[n++] = ec->tap_rotate_counter;
    iout[n++] = ec->latest_correction;
    iout[n++] = ec->narrowband_count;
    iout[n++] = ec->narrowband_score;
    iout[n++] = ec->fir_state.curr_pos;
    iout[n++] = ec->tx_hpf[0];
    iout[n++] = ec->tx_hpf[1];
    iout[n++] = ec->rx_hpf[0];
    iout[n++] = ec->rx_hpf[1];
    iout[n++] = ec->cng_level;
    iout[n++] = ec->cng_rndnum;
    iout[n++] = ec->cng_filter;
    for (i = 0;  i < ec->taps;  i++)
        taps32[i] = ec->fir_taps32[i];
    for (j = 0;  j < 4;  j++)
    {
        for (i = 0;  i < ec->taps;  i++)
            taps16[j*ec->taps + i] = ec->fir_taps16[j][i];
    }
    for (i = 0;  i < ec->taps;  i++)
        history[i] = ec->fir_state.history[i];
}

/* ---- Modems ----------------------------------------------------------------- */
GLUE v29_rx_state_t *glue_v29_rx_new(int bit_rate, glue_sink_t *k)
{
    return v29_rx_init(NULL, bit_rate, sink_put_bit, k);
}

GLUE v27ter_rx_state_t *glue_v27ter_rx_new(int bit_rate, glue_sink_t *k)
{
    return v27ter_rx_init(NULL, bit_rate, sink_put_bit, k);
}

GLUE v17_rx_state_t *glue_v17_rx_new(int bit_rate, glue_sink_t *k)
{
    return v17_rx_init(NULL, bit_rate, sink_put_bit, k);
}

/* A PRBS bit source for the transmitters: x^15 + x^14 + 1 LFSR, seed in *state. */
static int prbs_get_bit(void *user_data)
{
    uint32_t *st = (uint32_t *) user_data;
    int bit = ((*st >> 14) ^ (*st >> 13)) & 1;
    *st = ((*st << 1) | bit) & 0x7FFF;
    return bit;
}

GLUE void *glue_fn_prbs_get_bit(void) { return (void *) prbs_get_bit; }

GLUE v29_tx_state_t *glue_v29_tx_new(int bit_rate, int tep, uint32_t *prbs_state)
{
    return v29_tx_init(NULL, bit_rate, tep, prbs_get_bit, prbs_state);
}

GLUE v27ter_tx_state_t *glue_v27ter_tx_new(int bit_rate, int tep, uint32_t *prbs_state)
{
    return v27ter_tx_init(NULL, bit_rate, tep, prbs_get_bit, prbs_state);
}

GLUE v17_tx_state_t *glue_v17_tx_new(int bit_rate, int tep, uint32_t *prbs_state)
{
    return v17_tx_init(NULL, bit_rate, tep, prbs_get_bit, prbs_state);
}

GLUE int glue_sizeof(const char *what)
{
    if (strcmp(what, "dtmf_rx_state_t") == 0) return (int) sizeof(dtmf_rx_state_t);
    if (strcmp(what, "bell_mf_rx_state_t") == 0) return (int) sizeof(bell_mf_rx_state_t);
    if (strcmp(what, "r2_mf_rx_state_t") == 0) return (int) sizeof(r2_mf_rx_state_t);
    if (strcmp(what, "goertzel_state_t") == 0) return (int) sizeof(goertzel_state_t);
    if (strcmp(what, "echo_can_state_t") == 0) return (int) sizeof(echo_can_state_t);
    if (strcmp(what, "v29_rx_state_t") == 0) return (int) sizeof(v29_rx_state_t);
    if (strcmp(what, "v27ter_rx_state_t") == 0) return (int) sizeof(v27ter_rx_state_t);
    if (strcmp(what, "v17_rx_state_t") == 0) return (int) sizeof(v17_rx_state_t);
    return -1;
}
