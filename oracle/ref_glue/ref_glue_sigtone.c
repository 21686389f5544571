/*
 * ref_glue_sigtone.c -- TEST INFRASTRUCTURE ONLY.  OUR accessors and drivers over the reference's in-band signalling
 * tone processor (src/sig_tone.c): a snapshot of the receiver's private state in the order of the device
 * layout, a sender whose update-request callback sets the next mode from a script (what a caller's callback does, made
 * repeatable), and the frame loop of the cpu_baseline leg.  Compiled only into oracle/_ref/libspandsp_ref.so; #includes
 * reference headers from /root/reference/src at build time.
 */
#include <stdlib.h>
#include <inttypes.h>
#include <string.h>
#include <stdio.h>
#include <stdbool.h>

#include "spandsp/telephony.h"
#include "spandsp/complex.h"
#include "spandsp/power_meter.h"
#include "spandsp/tone_detect.h"
#include "spandsp/super_tone_rx.h"
#include "spandsp/sig_tone.h"
#include "spandsp/private/power_meter.h"
#include "spandsp/private/sig_tone.h"

#define GLUE __attribute__((visibility("default")))

/* 27 words: tone[3] x {z1[2], z2[2], power}, flat_z[2], flat power, then the counters and flags (floats as their bits) */
GLUE int glue_sigtone_rx_snapshot(const sig_tone_rx_state_t *s, int32_t *out)
{
    int n = 0;
    int j;

    for (j = 0;  j < 3;  j++)
    {
        memcpy(&out[n++], &s->tone[j].notch_z1[0], 4);
        memcpy(&out[n++], &s->tone[j].notch_z1[1], 4);
        memcpy(&out[n++], &s->tone[j].notch_z2[0], 4);
        memcpy(&out[n++], &s->tone[j].notch_z2[1], 4);
        out[n++] = s->tone[j].power.reading;
    }
    memcpy(&out[n++], &s->flat_z[0], 4);
    memcpy(&out[n++], &s->flat_z[1], 4);
    out[n++] = s->flat_power.reading;
    out[n++] = s->tone_persistence_timeout;
    out[n++] = s->last_sample_tone_present;
    out[n++] = s->flat_mode  ?  1  :  0;
    out[n++] = s->flat_mode_timeout;
    out[n++] = s->notch_insertion_timeout;
    out[n++] = s->signalling_state;
    out[n++] = s->signalling_state_duration;
    out[n++] = s->current_notch_filter;
    out[n++] = s->current_rx_tone;
    return n;
}

GLUE void glue_sigtone_rx_thresholds(const sig_tone_rx_state_t *s, int32_t out[3])
{
    out[0] = s->flat_detection_threshold;
    out[1] = s->sharp_detection_threshold;
    out[2] = s->detection_ratio;
}

GLUE int glue_sizeof_sig_tone_rx(void)
{
    return (int) sizeof(sig_tone_rx_state_t);
}

static void quiet_report(void *user_data, int code, int level, int delay)
{
    (void) code;
    (void) level;
    (void) delay;
    ++*(long long *) user_data;
}

GLUE sig_tone_rx_state_t *glue_sigtone_rx_new_quiet(int tone_type, int mode, long long *counter)
{
    sig_tone_rx_state_t *s = sig_tone_rx_init(NULL, tone_type, quiet_report, counter);

    if (s)
        sig_tone_rx_set_mode(s, mode, 0);
    return s;
}

/* CPU baseline helper: `frames` consecutive frames, `loops` times, on n receivers (the receiver writes its frame) */
GLUE void glue_sigtone_rx_batch_frames(sig_tone_rx_state_t **s, int16_t *amp, int n, long long stride, long long frame_stride,
                                       int samples, int frames, int loops)
{
    int c;
    int f;
    int l;

    for (l = 0;  l < loops;  l++)
    {
        for (f = 0;  f < frames;  f++)
        {
            for (c = 0;  c < n;  c++)
                sig_tone_rx(s[c], amp + f*frame_stride + c*stride, samples);
        }
    }
}

/* A receiver whose callback records the report and sets the next mode of a script */
typedef struct
{
    sig_tone_rx_state_t *rx;
    const int32_t *script;
    int script_len;
    int script_pos;
    int32_t *reports;           /* (state, level, duration) triples */
    int n_reports;
    int cap_reports;
} glue_sigtone_rx_t;

static void rx_update(void *user_data, int what, int level, int duration)
{
    glue_sigtone_rx_t *g = (glue_sigtone_rx_t *) user_data;

    if (g->n_reports == g->cap_reports)
    {
        g->cap_reports = g->cap_reports  ?  2*g->cap_reports  :  64;
        g->reports = (int32_t *) realloc(g->reports, sizeof(int32_t)*3*g->cap_reports);
    }
    g->reports[3*g->n_reports] = what;
    g->reports[3*g->n_reports + 1] = level;
    g->reports[3*g->n_reports + 2] = duration;
    g->n_reports++;
    if (g->script_pos < g->script_len)
        sig_tone_rx_set_mode(g->rx, g->script[g->script_pos++], 0);
}

GLUE glue_sigtone_rx_t *glue_sigtone_rx_scripted_new(int tone_type, int mode, const int32_t *script, int n)
{
    glue_sigtone_rx_t *g = (glue_sigtone_rx_t *) calloc(1, sizeof(*g));

    if (g == NULL)
        return NULL;
    if ((g->rx = sig_tone_rx_init(NULL, tone_type, rx_update, g)) == NULL)
    {
        free(g);
        return NULL;
    }
    sig_tone_rx_set_mode(g->rx, mode, 0);
    g->script = script;
    g->script_len = n;
    return g;
}

GLUE void glue_sigtone_rx_scripted_free(glue_sigtone_rx_t *g)
{
    if (g)
    {
        sig_tone_rx_free(g->rx);
        free(g->reports);
        free(g);
    }
}

GLUE int glue_sigtone_rx_scripted(glue_sigtone_rx_t *g, int16_t amp[], int len)
{
    return sig_tone_rx(g->rx, amp, len);
}

GLUE int glue_sigtone_rx_scripted_reports(const glue_sigtone_rx_t *g, const int32_t **reports)
{
    *reports = g->reports;
    return g->n_reports;
}

GLUE const sig_tone_rx_state_t *glue_sigtone_rx_scripted_state(const glue_sigtone_rx_t *g)
{
    return g->rx;
}

typedef struct
{
    sig_tone_tx_state_t *tx;
    const int32_t *script;
    int script_len;
    int script_pos;
    int requests;
} glue_sigtone_tx_t;

static void tx_update(void *user_data, int what, int level, int duration)
{
    glue_sigtone_tx_t *g = (glue_sigtone_tx_t *) user_data;

    (void) level;
    (void) duration;
    if (what != SIG_TONE_TX_UPDATE_REQUEST)
        return;
    g->requests++;
    if (g->script_pos < g->script_len)
    {
        sig_tone_tx_set_mode(g->tx, g->script[2*g->script_pos], g->script[2*g->script_pos + 1]);
        g->script_pos++;
    }
}

GLUE glue_sigtone_tx_t *glue_sigtone_tx_new(int tone_type, const int32_t *script, int n_pairs)
{
    glue_sigtone_tx_t *g = (glue_sigtone_tx_t *) calloc(1, sizeof(*g));

    if (g == NULL)
        return NULL;
    if ((g->tx = sig_tone_tx_init(NULL, tone_type, tx_update, g)) == NULL)
    {
        free(g);
        return NULL;
    }
    g->script = script;
    g->script_len = n_pairs;
    return g;
}

GLUE void glue_sigtone_tx_free(glue_sigtone_tx_t *g)
{
    if (g)
    {
        sig_tone_tx_free(g->tx);
        free(g);
    }
}

GLUE void glue_sigtone_tx_set_mode(glue_sigtone_tx_t *g, int mode, int duration)
{
    sig_tone_tx_set_mode(g->tx, mode, duration);
}

GLUE int glue_sigtone_tx(glue_sigtone_tx_t *g, int16_t amp[], int len)
{
    return sig_tone_tx(g->tx, amp, len);
}

GLUE int glue_sigtone_tx_requests(const glue_sigtone_tx_t *g)
{
    return g->requests;
}

/* 5 words of changing state, then the 6 set at creation: phase rates and the four scalings */
GLUE int glue_sigtone_tx_snapshot(const glue_sigtone_tx_t *g, int32_t *out)
{
    const sig_tone_tx_state_t *s = g->tx;
    int n = 0;

    out[n++] = (int32_t) s->phase_acc[0];
    out[n++] = (int32_t) s->phase_acc[1];
    out[n++] = s->high_low_timer;
    out[n++] = s->current_tx_tone;
    out[n++] = s->current_tx_timeout;
    out[n++] = s->phase_rate[0];
    out[n++] = s->phase_rate[1];
    out[n++] = s->tone_scaling[0][0];
    out[n++] = s->tone_scaling[0][1];
    out[n++] = s->tone_scaling[1][0];
    out[n++] = s->tone_scaling[1][1];
    return n;
}
