/*
 * mct_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's modem connect tone detector: CNG, CED / ANS
 * with and without phase reversals and amplitude modulation, Bell answer tone, calling tone, and the V.21
 * FAX preamble hunter built on fsk_rx.
 *
 *   modem_connect_tones_rx_init    src/modem_connect_tones.c:799-857
 *   modem_connect_tones_rx         src/modem_connect_tones.c:521-785
 *   v21_put_bit (HDLC flag hunt)   src/modem_connect_tones.c:437-518
 *   report_tone_state              src/modem_connect_tones.c:416-435
 *   modem_connect_tones_rx_get     src/modem_connect_tones.c:793-797
 *   power_meter_current_dbm0       src/power_meter.c:114-121
 *
 * Arithmetic notes: the notch / band-pass recurrences are binary32, evaluated left to right (the file
 * includes <tgmath.h>, so the fabs() of modem_connect_tones.c:597 is fabsf()); lfastrintf() truncates on
 * x86-64 (fast_convert.h:184-197).  Reports go to
 * the sink as kind 1 events (a = tone, b = level, c = 0), as the tone callback would see them; without a
 * callback the detector latches `hit`.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define MS(x)   ((x)*8)
#define MAX_POWER   (3.14f + 3.02f)

static void report(orc_mct_t *s, int tone, int level)
{
    if (tone == s->tone_present)
        return;
    if (s->sink)
        orc_sink_push(s->sink, 1, tone, level, 0);
    else if (tone != 0)
        s->hit = tone;
    s->tone_present = tone;
}

static int level_of(const orc_mct_t *s)
{
    /* the expression repeated at modem_connect_tones.c:561,643,659,723,777 */
    const float db = (s->channel_level == 0)  ?  (-96.329f + MAX_POWER)  :  20.0f*log10f(s->channel_level/32768.0f);
    return (int) (long) (db + MAX_POWER + 0.8f);
}

static void preamble_bit(void *user, int bit)
{
    orc_mct_t *s = (orc_mct_t *) user;

    if (bit < 0)
    {
        if (bit == -1  &&  s->tone_present == ORC_MCT_FAX_PREAMBLE)
            report(s, 0, -99);
        if (bit == -1  ||  bit == -2)
        {
            s->raw_bit_stream = 0;
            s->num_bits = 0;
            s->flags_seen = 0;
            s->framing_ok_announced = 0;
        }
        return;
    }
    s->raw_bit_stream = (s->raw_bit_stream << 1) | ((uint32_t) (bit << 8) & 0x100u);
    s->num_bits++;
    if ((s->raw_bit_stream & 0x7F00) == 0x7E00)
    {
        if (s->raw_bit_stream & 0x8000)
        {
            s->flags_seen = 0;          /* HDLC abort */
        }
        else if (s->flags_seen < 5)
        {
            /* flags must be back to back to count */
            if (s->num_bits != 8)
                s->flags_seen = 0;
            if (++s->flags_seen >= 5  &&  !s->framing_ok_announced)
            {
                /* lfastrintf(fsk_rx_signal_power()) */
                const int32_t reading = s->v21.power_reading;
                const float dbm0 = (reading <= 0)  ?  (-96.329f + MAX_POWER)
                                                   :  10.0f*log10f((float) reading/(32767.0f*32767.0f) + 1.0e-10f) + MAX_POWER;
                report(s, ORC_MCT_FAX_PREAMBLE, (int) (long) dbm0);
                s->framing_ok_announced = 1;
            }
        }
        s->num_bits = 0;
    }
    else if (s->flags_seen >= 5  &&  s->num_bits == 8)
    {
        s->framing_ok_announced = 0;
        s->flags_seen = 0;
    }
}

ORC_API int orc_mct_sizeof(void)
{
    return (int) sizeof(orc_mct_t);
}

ORC_API void orc_mct_init(orc_mct_t *s, int tone_type, orc_sink_t *sink)
{
    memset(s, 0, sizeof(*s));
    s->tone_type = tone_type & 0xFFF;
    s->sink = sink;
    if (s->tone_type == ORC_MCT_FAX_PREAMBLE  ||  s->tone_type == ORC_MCT_FAX_CED_OR_PREAMBLE)
    {
        int32_t spec[5];

        orc_fsk_preset(1, spec);                /* FSK_V21CH2 */
        orc_fsk_init(&s->v21, spec, 1);         /* FSK_FRAME_MODE_SYNC */
        orc_fsk_set_signal_cutoff(&s->v21, -45.5f);
    }
    else if (s->tone_type == ORC_MCT_ANS_PR  ||  s->tone_type == ORC_MCT_ANSAM  ||  s->tone_type == ORC_MCT_ANSAM_PR)
    {
        s->tone_type = ORC_MCT_ANS;
    }
}

ORC_API int orc_mct_get(orc_mct_t *s)
{
    const int x = s->hit;

    s->hit = 0;
    return x;
}

/* A notch section: v1 = g*x + a1*z1 - a2*z2;  y = v1 + b1*z1 + z2 (b1 carries its sign). */
static float notch(orc_mct_t *s, float x, float g, float a1, float a2, float b1)
{
    const float v1 = g*x + a1*s->znotch_1 - a2*s->znotch_2;
    const float y = v1 + b1*s->znotch_1 + s->znotch_2;

    s->znotch_2 = s->znotch_1;
    s->znotch_1 = v1;
    return y;
}

/* CNG, Bell answer and calling tone share one decision (modem_connect_tones.c:545-577,711-739,765-781) */
static void single_tone(orc_mct_t *s, int16_t amp, float y, int tone)
{
    const int16_t notched = (int16_t) (long) y;

    s->channel_level += ((abs(amp) - s->channel_level) >> 5);
    s->notch_level += ((abs(notched) - s->notch_level) >> 5);
    if (s->channel_level > 70  &&  s->notch_level*6 < s->channel_level)
    {
        if (s->tone_present != tone)
        {
            if (++s->tone_cycle_duration >= MS(415))
                report(s, tone, level_of(s));
        }
    }
    else
    {
        if (s->tone_present == tone)
            report(s, 0, -99);
        s->tone_cycle_duration = 0;
    }
}

ORC_API int orc_mct_rx(orc_mct_t *s, const int16_t amp[], int len)
{
    int i;

    switch (s->tone_type)
    {
    case ORC_MCT_FAX_CNG:
        for (i = 0;  i < len;  i++)
            single_tone(s, amp[i], notch(s, amp[i], 0.792928f, 1.0018744927985f, 0.54196833412465f, -1.2994747954630f), ORC_MCT_FAX_CNG);
        break;
    case ORC_MCT_BELL_ANS:
        for (i = 0;  i < len;  i++)
            single_tone(s, amp[i], notch(s, amp[i], 0.739651f, -0.257384f, 0.510404f, 0.351437f), ORC_MCT_BELL_ANS);
        break;
    case ORC_MCT_CALLING_TONE:
        for (i = 0;  i < len;  i++)
            single_tone(s, amp[i], notch(s, amp[i], 0.755582f, 0.820887174515f, 0.541968324778f, -1.0456667108f), ORC_MCT_CALLING_TONE);
        break;
    case ORC_MCT_FAX_PREAMBLE:
        orc_fsk_rx_cb(&s->v21, amp, len, preamble_bit, s);
        break;
    case ORC_MCT_FAX_CED_OR_PREAMBLE:
        orc_fsk_rx_cb(&s->v21, amp, len, preamble_bit, s);
        /* fall through */
    case ORC_MCT_ANS:
        for (i = 0;  i < len;  i++)
        {
            const float famp = amp[i];
            float v1;
            float filtered;
            float y;
            int16_t notched;
            int am;

            /* the 15 Hz AM detector, modem_connect_tones.c:593-601 */
            v1 = fabsf(famp) + 1.996667f*s->z15hz_1 - 0.9968004f*s->z15hz_2;
            filtered = 0.001599787f*(v1 - s->z15hz_2);
            s->z15hz_2 = s->z15hz_1;
            s->z15hz_1 = v1;
            s->am_level += abs((int) (long) filtered) - (s->am_level >> 8);
            y = notch(s, famp, 0.7552f, -0.1183852f, 0.5104039f, 0.1567596f);
            notched = (int16_t) (long) y;
            s->channel_level += ((abs(amp[i]) - s->channel_level) >> 5);
            s->notch_level += ((abs(notched) - s->notch_level) >> 4);
            if (s->channel_level <= 70)
            {
                if (s->tone_present != 0)
                    report(s, 0, -99);
                s->tone_cycle_duration = 0;
                s->good_cycles = 0;
                s->tone_on = 0;
                continue;
            }
            s->tone_cycle_duration++;
            am = (s->am_level*15/256 > s->channel_level);
            if (s->notch_level*6 < s->channel_level)
            {
                if (!s->tone_on)
                {
                    if (s->tone_cycle_duration >= MS(450 - 25))
                    {
                        if (++s->good_cycles == 3)
                            report(s, am  ?  ORC_MCT_ANSAM_PR  :  ORC_MCT_ANS_PR, level_of(s));
                    }
                    else
                    {
                        s->good_cycles = 0;
                    }
                    s->tone_cycle_duration = 0;
                }
                else if (s->tone_cycle_duration >= MS(450 + 100))
                {
                    if (s->tone_present == 0)
                        report(s, am  ?  ORC_MCT_ANSAM  :  ORC_MCT_ANS, level_of(s));
                    s->good_cycles = 0;
                    s->tone_cycle_duration = MS(450 + 100);
                }
                s->tone_on = 1;
            }
            else if (s->notch_level*5 > s->channel_level)
            {
                if (s->tone_present == ORC_MCT_ANS)
                {
                    report(s, 0, -99);
                    s->good_cycles = 0;
                }
                else if (s->tone_cycle_duration >= MS(450 + 25))
                {
                    if (s->tone_present == ORC_MCT_ANS_PR  ||  s->tone_present == ORC_MCT_ANSAM_PR)
                        report(s, 0, -99);
                    s->good_cycles = 0;
                }
                s->tone_on = 0;
            }
        }
        break;
    default:
        break;
    }
    return 0;
}
