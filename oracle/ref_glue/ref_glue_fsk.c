/*
 * ref_glue_fsk.c -- TEST INFRASTRUCTURE ONLY.  OUR accessors over the reference build's FSK
 * receiver (src/fsk.c): the preset table as data and a state snapshot in the word order of
 * spandsp_amd/csrc/fsk_dev.hpp / oracle/fsk_oracle.c.  Compiled only into
 * oracle/_ref/libspandsp_ref.so; #includes reference headers from /root/reference/src at build time.
 */
#include <stdlib.h>
#include <inttypes.h>
#include <string.h>
#include <stdio.h>
#include <stdbool.h>

#include "spandsp/telephony.h"
#include "spandsp/complex.h"
#include "spandsp/async.h"
#include "spandsp/power_meter.h"
#include "spandsp/fsk.h"
#include "spandsp/private/power_meter.h"
#include "spandsp/private/fsk.h"

#define GLUE __attribute__((visibility("default")))

/* freq_zero, freq_one, tx_level, min_level, baud_rate of preset `which` */
GLUE int glue_fsk_preset(int which, int32_t out[5])
{
    if (which < 0  ||  which > FSK_V21CH1_110)
        return -1;
    out[0] = preset_fsk_specs[which].freq_zero;
    out[1] = preset_fsk_specs[which].freq_one;
    out[2] = preset_fsk_specs[which].tx_level;
    out[3] = preset_fsk_specs[which].min_level;
    out[4] = preset_fsk_specs[which].baud_rate;
    return 0;
}

GLUE fsk_rx_state_t *glue_fsk_rx_new(int which, int framing_mode, span_put_bit_func_t put_bit, void *user_data)
{
    return fsk_rx_init(NULL, &preset_fsk_specs[which], framing_mode, put_bit, user_data);
}

GLUE int glue_fsk_rx_restart(fsk_rx_state_t *s, int which, int framing_mode)
{
    return fsk_rx_restart(s, &preset_fsk_specs[which], framing_mode);
}

GLUE fsk_tx_state_t *glue_fsk_tx_new(int which, span_get_bit_func_t get_bit, void *user_data)
{
    return fsk_tx_init(NULL, &preset_fsk_specs[which], get_bit, user_data);
}

/* 28 scalar words, then the correlation window as [slot][tone][re, im] for `correlation_span` slots */
GLUE int glue_fsk_rx_snapshot(const fsk_rx_state_t *s, int32_t *out)
{
    int n = 0;
    int i;
    int j;

    out[n++] = s->baud_rate;
    out[n++] = s->framing_mode;
    out[n++] = s->data_bits;
    out[n++] = s->parity;
    out[n++] = s->stop_bits;
    out[n++] = s->total_data_bits;
    out[n++] = s->carrier_on_power;
    out[n++] = s->carrier_off_power;
    out[n++] = s->power.reading;
    out[n++] = s->last_sample;
    out[n++] = s->signal_present;
    out[n++] = s->phase_rate[0];
    out[n++] = s->phase_rate[1];
    out[n++] = (int32_t) s->phase_acc[0];
    out[n++] = (int32_t) s->phase_acc[1];
    out[n++] = s->correlation_span;
    out[n++] = s->dot[0].re;
    out[n++] = s->dot[0].im;
    out[n++] = s->dot[1].re;
    out[n++] = s->dot[1].im;
    out[n++] = s->buf_ptr;
    out[n++] = s->frame_pos;
    out[n++] = s->frame_in_progress;
    out[n++] = s->baud_phase;
    out[n++] = s->last_bit;
    out[n++] = s->scaling_shift;
    out[n++] = s->parity_errors;
    out[n++] = s->framing_errors;
    for (i = 0;  i < s->correlation_span;  i++)
    {
        for (j = 0;  j < 2;  j++)
        {
            out[n++] = s->window[j][i].re;
            out[n++] = s->window[j][i].im;
        }
    }
    return n;
}

GLUE int glue_sizeof_fsk_rx(void)
{
    return (int) sizeof(fsk_rx_state_t);
}

/* ---- CPU baseline helpers: receivers whose put_bit() only counts, run a frame at a time ---- */
static void count_put_bit(void *user_data, int bit)
{
    (void) bit;
    ++*(long long *) user_data;
}

GLUE fsk_rx_state_t *glue_fsk_rx_new_quiet(int which, int framing_mode, long long *counter)
{
    return fsk_rx_init(NULL, &preset_fsk_specs[which], framing_mode, count_put_bit, counter);
}

GLUE void glue_fsk_rx_batch(fsk_rx_state_t **s, const int16_t *amp, int n, long long stride, int samples)
{
    int c;

    for (c = 0;  c < n;  c++)
        fsk_rx(s[c], amp + c*stride, samples);
}

/* The same over `frames` consecutive frames (frame f of channel c at amp + f*frame_stride + c*stride), `loops` times:
   one call per worker thread keeps the Python harness out of the timed region. */
GLUE void glue_fsk_rx_batch_frames(fsk_rx_state_t **s, const int16_t *amp, int n, long long stride, long long frame_stride,
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
                fsk_rx(s[c], amp + f*frame_stride + c*stride, samples);
        }
    }
}

/* ---- modem connect tones (src/modem_connect_tones.c) ---- */
#include "spandsp/tone_detect.h"
#include "spandsp/super_tone_rx.h"
#include "spandsp/tone_generate.h"
#include "spandsp/modem_connect_tones.h"
#include "spandsp/private/tone_generate.h"
#include "spandsp/private/modem_connect_tones.h"

/* 18 words of the detector proper (floats as their bits), then the V.21 receiver's snapshot when it is in use */
GLUE int glue_mct_rx_snapshot(const modem_connect_tones_rx_state_t *s, int32_t *out)
{
    int n = 0;

    out[n++] = s->tone_type;
    memcpy(&out[n++], &s->znotch_1, 4);
    memcpy(&out[n++], &s->znotch_2, 4);
    memcpy(&out[n++], &s->z15hz_1, 4);
    memcpy(&out[n++], &s->z15hz_2, 4);
    out[n++] = s->notch_level;
    out[n++] = s->channel_level;
    out[n++] = s->am_level;
    out[n++] = s->tone_present;
    out[n++] = s->tone_on;
    out[n++] = s->tone_cycle_duration;
    out[n++] = s->good_cycles;
    out[n++] = s->hit;
    out[n++] = (int32_t) s->raw_bit_stream;
    out[n++] = s->num_bits;
    out[n++] = s->flags_seen;
    out[n++] = s->framing_ok_announced;
    out[n++] = 0;
    if (s->tone_type == MODEM_CONNECT_TONES_FAX_PREAMBLE  ||  s->tone_type == MODEM_CONNECT_TONES_FAX_CED_OR_PREAMBLE)
        n += glue_fsk_rx_snapshot(&s->v21rx, out + n);
    return n;
}

GLUE int glue_sizeof_mct_rx(void)
{
    return (int) sizeof(modem_connect_tones_rx_state_t);
}

/* CPU baseline helper: `frames` consecutive frames, `loops` times, on n detectors (no callback: hits latch) */
GLUE void glue_mct_rx_batch_frames(modem_connect_tones_rx_state_t **s, const int16_t *amp, int n, long long stride,
                                   long long frame_stride, int samples, int frames, int loops)
{
    int c;
    int f;
    int l;

    for (l = 0;  l < loops;  l++)
    {
        for (f = 0;  f < frames;  f++)
        {
            for (c = 0;  c < n;  c++)
                modem_connect_tones_rx(s[c], amp + f*frame_stride + c*stride, samples);
        }
    }
}
