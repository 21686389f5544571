/*
 * sigtone_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's in-band signalling tone processor (SURVEY.md section 8(f)-4 names sig_tone.c
 * beside v18.c and ademco_contactid.c): the 2280 Hz, 2600 Hz and 2400 Hz / 2600 Hz detectors with their notch
 * filters in the media path, and the matching tone sender.  Float build of the library (no SPANDSP_USE_FIXED_POINT),
 * x86-64.
 *
 *   the three tone descriptors             src/sig_tone.c:77-244   (coefficients, timers, thresholds)
 *   sig_tone_rx                            src/sig_tone.c:402-663
 *   sig_tone_rx_set_mode / _init           src/sig_tone.c:666-723
 *   sig_tone_tx                            src/sig_tone.c:246-323
 *   sig_tone_tx_set_mode / _init           src/sig_tone.c:326-382
 *   power_meter_update / _level_dbm0       src/power_meter.c:65-69,82-92
 *   dds_mod / dds_lookup / dds_phase_rate  src/dds_int.c:316-319,340-355,380-387
 *   dds_scaling_dbm0                       src/dds_int.c:328-331
 *   fsaturatef / sat_add16                 src/spandsp/saturated.h:142-149,206-233
 *
 * Arithmetic notes.  The bi-quads are binary32 sums evaluated left to right as written (the library is built with
 * -ffp-contract=off): v = (x*a0 + z0*b1) + z1*b2, then v += (z0*a1 + z1*a2).  power_meter_update() takes an int16_t:
 * handing it a float is a truncating conversion to int and then to 16 bits (cvttss2si + movswl).  The detection
 * ratio is the integer part of 10^(dB/10) + 1.  A report is the tone callback's (state, 0, duration): a kind 1 event.
 *
 * The sender's update request is a callback from inside sig_tone_tx(), and what a caller does in it is to call
 * sig_tone_tx_set_mode(); the restatement (and the glue over the reference, ref_glue_tones.c) take the sequence of
 * such calls as a script of (mode, duration) pairs, one consumed per request.
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define MS(t)       ((t)*8)
#define MAX_POWER   (3.14f + 3.02f)

typedef struct
{
    float a1[3];
    float b1[3];
    float a2[3];
    float b2[3];
} notch_t;

/* sig_tone.c:77-121 (float branch): 2280 Hz, 2400 Hz, 2600 Hz */
static const notch_t notches[3] =
{
    {{0.878906f, 0.439362f, 1.0f}, {0.0f, -0.287627f, -0.883605f}, {0.0f, 0.433228f, 1.0f}, {0.0f, -0.530792f, -0.883605f}},
    {{0.862000f, 0.612055f, 1.0f}, {0.0f, -0.456264f, -0.864899f}, {0.0f, 0.621021f, 1.0f}, {0.0f, -0.690738f, -0.864899f}},
    {{0.862000f, 0.902374f, 1.0f}, {0.0f, -0.732727f, -0.864899f}, {0.0f, 0.910766f, 1.0f}, {0.0f, -0.952393f, -0.864899f}}
};

/* sig_tone.c:123-135 */
static const float flat_a[3] = {0.393676f, -0.5f, -0.5f};
static const float flat_b[3] = {0.0f, -0.261778f, -0.359985f};

typedef struct
{
    int tone_freq[2];
    int tone_amp[2][2];
    int high_low_timeout;
    int sharp_flat_timeout;
    int notch_lag_time;
    int tone_on_check_time;
    int tone_off_check_time;
    int tones;
    int notch[2];               /* index into notches[] */
    int flat;                   /* the flat filter exists */
    float detection_ratio;
    float sharp_detection_threshold;
    float flat_detection_threshold;
} desc_t;

/* sig_tone.c:137-223 */
static const desc_t descs[3] =
{
    {{2280, 0}, {{-10, -20}, {0, 0}}, MS(400), MS(225), MS(225), MS(3), MS(8), 1, {0, 0}, 1, 13.0f, -30.0f, -30.0f},
    {{2600, 0}, {{-8, -8}, {0, 0}},   MS(0),   MS(0),   MS(225), MS(3), MS(8), 1, {2, 0}, 0, 15.6f, -30.0f, -30.0f},
    {{2400, 2600}, {{-8, -8}, {-8, -8}}, MS(0), MS(0),  MS(225), MS(3), MS(8), 2, {1, 2}, 0, 15.6f, -30.0f, -30.0f}
};

static const int present_bits[3] = {0x001, 0x004, 0x001 | 0x004};      /* sig_tone.c:225-230 */
static const int change_bits[3] = {0x002, 0x008, 0x002 | 0x008};       /* sig_tone.c:232-237 */
static const int coeff_sets[3] = {0, 1, 0};                            /* sig_tone.c:239-244 */

static int32_t level_dbm0(float level)
{
    /* power_meter_level_dbm0(), power_meter.c:82-92 */
    float l;

    level -= MAX_POWER;
    if (level > 0.0)
        level = 0.0;
    l = powf(10.0f, level/10.0f)*(32767.0f*32767.0f);
    return (int32_t) l;
}

static int32_t meter(int32_t *reading, int16_t amp)
{
    /* power_meter_update(), power_meter.c:65-69, damping 5 (sig_tone.c:710-712) */
    *reading += ((amp*amp - *reading) >> 5);
    return *reading;
}

static int16_t to_i16(float v)
{
    /* a float handed to an int16_t parameter */
    return (int16_t) (int32_t) v;
}

static int16_t fsat(float famp)
{
    /* fsaturatef(), saturated.h:142-149 */
    if (famp > 32767.0f)
        return 32767;
    if (famp < -32768.0f)
        return -32768;
    return (int16_t) lrintf(famp);
}

int orc_sigtone_rx_sizeof(void)
{
    return (int) sizeof(orc_sigtone_rx_t);
}

int orc_sigtone_rx_init(orc_sigtone_rx_t *s, int tone_type, orc_sink_t *sink)
{
    /* sig_tone_rx_init(), sig_tone.c:672-723 */
    if (tone_type < 1  ||  tone_type > 3)
        return -1;
    memset(s, 0, sizeof(*s));
    s->last_sample_tone_present = -1;
    s->tone_type = tone_type;
    s->sink = sink;
    s->flat_detection_threshold = level_dbm0(descs[tone_type - 1].flat_detection_threshold);
    s->sharp_detection_threshold = level_dbm0(descs[tone_type - 1].sharp_detection_threshold);
    s->detection_ratio = (int32_t) (powf(10.0f, descs[tone_type - 1].detection_ratio/10.0f) + 1.0f);
    return 0;
}

void orc_sigtone_rx_set_mode(orc_sigtone_rx_t *s, int mode)
{
    s->current_rx_tone = mode;          /* sig_tone.c:666-669 */
}

void orc_sigtone_rx_script(orc_sigtone_rx_t *s, const int32_t *modes, int n)
{
    s->script = modes;
    s->script_len = n;
    s->script_pos = 0;
}

void orc_sigtone_rx_thresholds(int tone_type, int32_t out[3])
{
    orc_sigtone_rx_t t;

    out[0] = out[1] = out[2] = 0;
    if (orc_sigtone_rx_init(&t, tone_type, NULL) == 0)
    {
        out[0] = t.flat_detection_threshold;
        out[1] = t.sharp_detection_threshold;
        out[2] = t.detection_ratio;
    }
}

int orc_sigtone_rx(orc_sigtone_rx_t *s, int16_t amp[], int len)
{
    /* sig_tone_rx(), sig_tone.c:402-663 */
    const desc_t *d = &descs[s->tone_type - 1];
    float notched[3] = {0.0f, 0.0f, 0.0f};
    int32_t notch_power[3] = {0, INT32_MAX, INT32_MAX};
    int32_t flat_power;
    int l = (d->tones == 2)  ?  3  :  d->tones;

    for (int i = 0;  i < len;  i++)
    {
        float signal;
        float band;
        float v;
        float x;
        int immediate;

        if (s->signalling_state_duration < INT_MAX)
            s->signalling_state_duration++;
        signal = amp[i];
        for (int j = 0;  j < l;  j++)
        {
            const notch_t *c = &notches[d->notch[coeff_sets[j]]];

            v = signal*c->a1[0] + s->tone[j].z1[0]*c->b1[1] + s->tone[j].z1[1]*c->b1[2];
            x = v;
            v += s->tone[j].z1[0]*c->a1[1] + s->tone[j].z1[1]*c->a1[2];
            s->tone[j].z1[1] = s->tone[j].z1[0];
            s->tone[j].z1[0] = x;
            v += s->tone[j].z2[0]*c->b2[1] + s->tone[j].z2[1]*c->b2[2];
            x = v;
            v += s->tone[j].z2[0]*c->a2[1] + s->tone[j].z2[1]*c->a2[2];
            s->tone[j].z2[1] = s->tone[j].z2[0];
            s->tone[j].z2[0] = x;
            notched[j] = v;
            notch_power[j] = meter(&s->tone[j].power, to_i16(notched[j]));
            if (j == 1)
                signal = notched[j];
        }
        if ((s->signalling_state & (0x001 | 0x004)))
        {
            if (s->flat_mode_timeout  &&  --s->flat_mode_timeout == 0)
                s->flat_mode = 1;
        }
        else
        {
            s->flat_mode_timeout = d->sharp_flat_timeout;
            s->flat_mode = 0;
        }
        immediate = -1;
        if (s->flat_mode)
        {
            band = amp[i];
            if (d->flat)
            {
                v = amp[i]*flat_a[0] + s->flat_z[0]*flat_b[1] + s->flat_z[1]*flat_b[2];
                x = v;
                v += s->flat_z[0]*flat_a[1] + s->flat_z[1]*flat_a[2];
                s->flat_z[1] = s->flat_z[0];
                s->flat_z[0] = x;
                band = v;
            }
            flat_power = meter(&s->flat_power, to_i16(band));
            if ((s->signalling_state & (0x001 | 0x004)))
            {
                if (flat_power < s->flat_detection_threshold)
                {
                    s->signalling_state &= ~present_bits[0];
                    s->signalling_state |= change_bits[0];
                }
            }
            else
            {
                if (flat_power > s->flat_detection_threshold)
                    s->signalling_state |= (present_bits[0] | change_bits[0]);
            }
            if ((s->signalling_state & (0x001 | 0x004)))
            {
                s->notch_insertion_timeout = d->notch_lag_time;
            }
            else
            {
                if (s->notch_insertion_timeout)
                    s->notch_insertion_timeout--;
            }
        }
        else
        {
            flat_power = meter(&s->flat_power, amp[i]);
            if (flat_power >= s->sharp_detection_threshold)
            {
                const int m = (notch_power[0] < notch_power[1])  ?  0  :  1;

                if ((notch_power[m] >> 6)*s->detection_ratio < (flat_power >> 6))
                    immediate = m;
                else if ((notch_power[2] >> 6)*s->detection_ratio < (flat_power >> 7))
                    immediate = 2;
            }
            if ((s->signalling_state & (0x001 | 0x004)))
            {
                if (immediate != s->current_notch_filter)
                {
                    if (--s->tone_persistence_timeout == 0)
                    {
                        s->tone_persistence_timeout = d->tone_on_check_time;
                        s->signalling_state |= ((s->signalling_state & (0x001 | 0x004)) << 1);
                        s->signalling_state &= ~(0x001 | 0x004);
                    }
                }
                else
                {
                    s->tone_persistence_timeout = d->tone_off_check_time;
                }
            }
            else
            {
                if (s->notch_insertion_timeout)
                    s->notch_insertion_timeout--;
                if (immediate >= 0  &&  immediate == s->last_sample_tone_present)
                {
                    if (--s->tone_persistence_timeout == 0)
                    {
                        s->tone_persistence_timeout = d->tone_off_check_time;
                        s->notch_insertion_timeout = d->notch_lag_time;
                        s->signalling_state |= (present_bits[immediate] | change_bits[immediate]);
                        s->current_notch_filter = immediate;
                    }
                }
                else
                {
                    s->tone_persistence_timeout = d->tone_on_check_time;
                }
            }
        }
        if ((s->signalling_state & (0x002 | 0x008)))
        {
            if (s->sink)
                orc_sink_push(s->sink, 1, s->signalling_state, 0, s->signalling_state_duration);
            /* inside the callback: a caller may set the mode, which the media path of this very sample then uses */
            if (s->script_pos < s->script_len)
                s->current_rx_tone = s->script[s->script_pos++];
            s->signalling_state &= ~(0x002 | 0x008);
            s->signalling_state_duration = 0;
        }
        if ((s->current_rx_tone & 0x040))
        {
            if ((s->current_rx_tone & 0x080)  ||  s->notch_insertion_timeout)
                amp[i] = fsat(notched[s->current_notch_filter]);
        }
        else
        {
            amp[i] = 0;
        }
        s->last_sample_tone_present = immediate;
    }
    return len;
}

/* ---- sender ---------------------------------------------------------------------------------------------- */

static int16_t quarter_sine[257];
static int quarter_sine_ready = 0;

static int32_t lookup(uint32_t phase)
{
    /* dds_lookup(), dds_int.c:340-355 */
    uint32_t step;
    int32_t amp;

    if (!quarter_sine_ready)
    {
        for (int i = 0;  i <= 256;  i++)
            quarter_sine[i] = (int16_t) lrint(32767.0*sin(i*3.14159265358979323846/512.0));
        quarter_sine_ready = 1;
    }
    phase >>= 22;
    step = phase & 255;
    if (phase & 256)
        step = 256 - step;
    amp = quarter_sine[step];
    return (phase & 512)  ?  -amp  :  amp;
}

static int16_t sat_add(int16_t x, int16_t y)
{
    const int32_t z = (int32_t) x + y;

    return (int16_t) ((z > 32767)  ?  32767  :  (z < -32768)  ?  -32768  :  z);
}

int orc_sigtone_tx_sizeof(void)
{
    return (int) sizeof(orc_sigtone_tx_t);
}

int orc_sigtone_tx_init(orc_sigtone_tx_t *s, int tone_type, orc_sink_t *sink)
{
    /* sig_tone_tx_init(), sig_tone.c:348-382 */
    if (tone_type < 1  ||  tone_type > 3)
        return -1;
    memset(s, 0, sizeof(*s));
    s->tone_type = tone_type;
    s->sink = sink;
    for (int i = 0;  i < 2;  i++)
    {
        const desc_t *d = &descs[tone_type - 1];

        s->phase_rate[i] = d->tone_freq[i]  ?  (int32_t) ((float) d->tone_freq[i]*65536.0f*65536.0f/8000)  :  0;
        s->tone_scaling[i][0] = (int16_t) (powf(10.0f, ((float) d->tone_amp[i][0] - 3.14f)/20.0f)*32767.0f);
        s->tone_scaling[i][1] = (int16_t) (powf(10.0f, ((float) d->tone_amp[i][1] - 3.14f)/20.0f)*32767.0f);
    }
    return 0;
}

void orc_sigtone_tx_set_mode(orc_sigtone_tx_t *s, int mode, int duration)
{
    /* sig_tone_tx_set_mode(), sig_tone.c:326-345 */
    const int old_tones = s->current_tx_tone & (0x001 | 0x004);
    const int new_tones = mode & (0x001 | 0x004);

    if (new_tones  &&  old_tones != new_tones)
        s->high_low_timer = descs[s->tone_type - 1].high_low_timeout;
    if ((mode & 0x001)  &&  !(s->current_tx_tone & 0x001))
        s->phase_acc[0] = 0;
    if ((mode & 0x004)  &&  !(s->current_tx_tone & 0x004))
        s->phase_acc[1] = 0;
    s->current_tx_tone = mode;
    s->current_tx_timeout = duration;
}

void orc_sigtone_tx_script(orc_sigtone_tx_t *s, const int32_t *script, int n_pairs)
{
    s->script = script;
    s->script_len = n_pairs;
    s->script_pos = 0;
}

int orc_sigtone_tx(orc_sigtone_tx_t *s, int16_t amp[], int len)
{
    /* sig_tone_tx(), sig_tone.c:246-323 */
    const desc_t *d = &descs[s->tone_type - 1];
    int n;

    for (int i = 0;  i < len;  i += n)
    {
        int need_update;
        int high_low;

        if (s->current_tx_timeout)
        {
            if (s->current_tx_timeout <= len - i)
            {
                n = s->current_tx_timeout;
                need_update = 1;
            }
            else
            {
                n = len - i;
                need_update = 0;
            }
            s->current_tx_timeout -= n;
        }
        else
        {
            n = len - i;
            need_update = 0;
        }
        if (!(s->current_tx_tone & 0x010))
            memset(&amp[i], 0, sizeof(int16_t)*n);
        if ((s->current_tx_tone & (0x001 | 0x004)))
        {
            if (s->high_low_timer > 0)
            {
                if (n > s->high_low_timer)
                    n = s->high_low_timer;
                s->high_low_timer -= n;
                high_low = 0;
            }
            else
            {
                high_low = 1;
            }
            for (int k = 0;  k < d->tones;  k++)
            {
                if ((s->current_tx_tone & present_bits[k])  &&  s->phase_rate[k])
                {
                    for (int j = i;  j < i + n;  j++)
                    {
                        /* dds_mod(), dds_int.c:380-387 */
                        const int16_t tone = (int16_t) ((lookup(s->phase_acc[k])*s->tone_scaling[k][high_low]) >> 15);

                        s->phase_acc[k] += (uint32_t) s->phase_rate[k];
                        amp[j] = sat_add(amp[j], tone);
                    }
                }
            }
        }
        if (need_update)
        {
            /* the callback: sig_update(user, SIG_TONE_TX_UPDATE_REQUEST, 0, 0), in which the caller sets the next mode */
            if (s->sink)
                orc_sink_push(s->sink, 1, 0x100, 0, 0);
            if (s->script_pos < s->script_len)
            {
                orc_sigtone_tx_set_mode(s, s->script[2*s->script_pos], s->script[2*s->script_pos + 1]);
                s->script_pos++;
            }
        }
    }
    return len;
}
