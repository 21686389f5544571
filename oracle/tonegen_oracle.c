/*
 * tonegen_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's signal sources that sit on the other side of the
 * tone detectors:
 *
 *   tone_gen_descriptor_init   tone_generate.c:60-120
 *   tone_gen_init              tone_generate.c:232-262
 *   tone_gen                   tone_generate.c:128-229   (float build)
 *   dds_phase_ratef            dds_float.c:2109-2112
 *   dds_scaling_dbm0f          dds_float.c:2121-2124  (db_to_amplitude_ratio = powf(10, x/20), telephony.h:141)
 *   dds_modf                   dds_float.c:2167-2174
 *   dtmf_tx / _put / _set_level / _set_timing / _init      dtmf.c:551-660
 *   bell_mf_tx / _put / _init                               bell_r2_mf.c:306-381
 *   r2_mf_tx / _put / _init                                 bell_r2_mf.c:399-487
 *   queue_* byte ring (capacity = len)                      queue.c:60-260
 *
 * The sine table is the one handed to orc_modem_set_tables() (the 2048 entry dds_float.c table).
 */
#include <math.h>
#include <string.h>

#include "oracle.h"
#include "modem_common.h"

#define TX_SAMPLE_RATE      8000
#define TX_MAX_SINE_DBM0    3.14f       /* DBM0_MAX_SINE_POWER, telephony.h */

static int32_t phase_ratef(float hz)
{
    /* dds_float.c:2111 -- all binary32 */
    return (int32_t) (hz*65536.0f*65536.0f/TX_SAMPLE_RATE);
}

static float scaling_dbm0f(float level)
{
    /* dds_float.c:2123 */
    return powf(10.0f, (level - TX_MAX_SINE_DBM0)/20.0f)*32767.0f;
}

static long fast_to_int(float x)
{
    /* lfastrintf() of an x86-64 build is a plain cast, i.e. truncation toward zero
       (spandsp/fast_convert.h:184-197), not lrintf(). */
    return (long) x;
}

static float dds_mod_step(uint32_t *acc, int32_t rate, float scale)
{
    /* dds_float.c:2167-2174 with phase = 0 */
    const float v = orc_modem_T.sine[*acc >> 21]*scale;

    *acc += (uint32_t) rate;
    return v;
}

ORC_API void orc_tone_desc_init(orc_tone_desc_t *d, int f1, int l1, int f2, int l2, int d1, int d2, int d3, int d4,
                                int repeat)
{
    memset(d, 0, sizeof(*d));
    if (f1)
    {
        d->tone[0].phase_rate = phase_ratef((float) f1);
        if (f2 < 0)
            d->tone[0].phase_rate = -d->tone[0].phase_rate;
        d->tone[0].gain = scaling_dbm0f((float) l1);
    }
    if (f2)
    {
        d->tone[1].phase_rate = phase_ratef((float) (f2 < 0  ?  -f2  :  f2));
        d->tone[1].gain = (f2 < 0)  ?  (float) l2/100.0f  :  scaling_dbm0f((float) l2);
    }
    d->duration[0] = d1*TX_SAMPLE_RATE/1000;
    d->duration[1] = d2*TX_SAMPLE_RATE/1000;
    d->duration[2] = d3*TX_SAMPLE_RATE/1000;
    d->duration[3] = d4*TX_SAMPLE_RATE/1000;
    d->repeat = repeat;
}

ORC_API void orc_tone_gen_init(orc_tone_gen_t *g, const orc_tone_desc_t *d)
{
    memset(g, 0, sizeof(*g));
    memcpy(g->tone, d->tone, sizeof(g->tone));
    memcpy(g->duration, d->duration, sizeof(g->duration));
    g->repeat = d->repeat;
}

ORC_API int orc_tone_gen(orc_tone_gen_t *g, int16_t amp[], int max_samples)
{
    int done;

    if (g->current_section < 0)
        return 0;
    done = 0;
    while (done < max_samples)
    {
        /* run to the end of this cadence section or of the caller's buffer, whichever is first */
        int run = g->duration[g->current_section] - g->current_position;

        if (run > max_samples - done)
            run = max_samples - done;
        g->current_position += run;
        if (g->current_section & 1)
        {
            for (  ;  run > 0;  run--)
                amp[done++] = 0;
        }
        else if (g->tone[0].phase_rate < 0)
        {
            /* amplitude modulated pair (tone_generate.c:166-183) */
            for (  ;  run > 0;  run--)
            {
                const float carrier = dds_mod_step(&g->phase[0], -g->tone[0].phase_rate, g->tone[0].gain);
                const float depth = dds_mod_step(&g->phase[1], g->tone[1].phase_rate, g->tone[1].gain);

                amp[done++] = (int16_t) fast_to_int(carrier*(1.0f + depth));
            }
        }
        else
        {
            for (  ;  run > 0;  run--)
            {
                float x = 0.0f;

                for (int i = 0;  i < 4  &&  g->tone[i].phase_rate != 0;  i++)
                    x += dds_mod_step(&g->phase[i], g->tone[i].phase_rate, g->tone[i].gain);
                amp[done++] = (int16_t) fast_to_int(x);
            }
        }
        if (g->current_position >= g->duration[g->current_section])
        {
            g->current_position = 0;
            g->current_section++;
            if (g->current_section > 3  ||  g->duration[g->current_section] == 0)
            {
                if (!g->repeat)
                {
                    g->current_section = -1;
                    break;
                }
                g->current_section = 0;
            }
        }
    }
    return done;
}

/* ---- digit queue: queue.c byte ring with len = 128, so 128 bytes fit ---- */
static int q_put(orc_digit_queue_t *q, const char *digits, int len)
{
    if (len < 0)
    {
        if ((len = (int) strlen(digits)) == 0)
            return 0;
    }
    if (ORC_TX_QUEUE - q->count < len)
        return len - (ORC_TX_QUEUE - q->count);
    for (int i = 0;  i < len;  i++)
        q->data[(q->rd + q->count + i)%ORC_TX_QUEUE] = (uint8_t) digits[i];
    q->count += len;
    return 0;
}

static int q_get(orc_digit_queue_t *q)
{
    int c;

    if (q->count == 0)
        return -1;
    c = q->data[q->rd];
    q->rd = (q->rd + 1)%ORC_TX_QUEUE;
    q->count--;
    return c;
}

/* ---- dtmf_tx ---- */
static const char dtmf_keys[] = "123A456B789C*0#D";
static const int dtmf_rows[4] = {697, 770, 852, 941};
static const int dtmf_cols[4] = {1209, 1336, 1477, 1633};

ORC_API void orc_dtmf_tx_init(orc_dtmf_tx_t *s)
{
    memset(s, 0, sizeof(*s));
    /* dtmf.c:653-658 */
    s->low_level = scaling_dbm0f(-10.0f);
    s->high_level = scaling_dbm0f(-10.0f);
    s->on_time = 50*TX_SAMPLE_RATE/1000;
    s->off_time = 55*TX_SAMPLE_RATE/1000;
    s->tones.current_section = -1;
}

ORC_API void orc_dtmf_tx_set_level(orc_dtmf_tx_t *s, int level, int twist)
{
    s->low_level = scaling_dbm0f((float) level);
    s->high_level = scaling_dbm0f((float) (level + twist));
}

ORC_API void orc_dtmf_tx_set_timing(orc_dtmf_tx_t *s, int on_time, int off_time)
{
    s->on_time = ((on_time >= 0)  ?  on_time  :  50)*TX_SAMPLE_RATE/1000;
    s->off_time = ((off_time >= 0)  ?  off_time  :  55)*TX_SAMPLE_RATE/1000;
}

ORC_API int orc_dtmf_tx_put(orc_dtmf_tx_t *s, const char *digits, int len)
{
    return q_put(&s->queue, digits, len);
}

ORC_API int orc_dtmf_tx(orc_dtmf_tx_t *s, int16_t amp[], int max_samples)
{
    int len = 0;
    int digit;

    if (s->tones.current_section >= 0)
        len = orc_tone_gen(&s->tones, amp, max_samples);
    while (len < max_samples  &&  (digit = q_get(&s->queue)) >= 0)
    {
        const char *at;
        orc_tone_desc_t d;

        if (digit == 0  ||  (at = strchr(dtmf_keys, digit)) == NULL)
            continue;
        /* dtmf.c:529-540 builds the 16 descriptors; :577-581 overrides levels and cadence */
        orc_tone_desc_init(&d, dtmf_rows[(at - dtmf_keys) >> 2], -10, dtmf_cols[(at - dtmf_keys) & 3], -10, 50, 55, 0, 0, 0);
        orc_tone_gen_init(&s->tones, &d);
        s->tones.tone[0].gain = s->low_level;
        s->tones.tone[1].gain = s->high_level;
        s->tones.duration[0] = s->on_time;
        s->tones.duration[1] = s->off_time;
        len += orc_tone_gen(&s->tones, amp + len, max_samples - len);
    }
    return len;
}

/* ---- bell_mf_tx ---- */
static const char bell_keys[] = "1234567890CA*B#";
static const int bell_pairs[15][2] =
{
    {700, 900}, {700, 1100}, {900, 1100}, {700, 1300}, {900, 1300}, {1100, 1300}, {700, 1500}, {900, 1500},
    {1100, 1500}, {1300, 1500}, {700, 1700}, {900, 1700}, {1100, 1700}, {1300, 1700}, {1500, 1700}
};

ORC_API void orc_bell_mf_tx_init(orc_bell_mf_tx_t *s)
{
    memset(s, 0, sizeof(*s));
    s->tones.current_section = -1;
}

ORC_API int orc_bell_mf_tx_put(orc_bell_mf_tx_t *s, const char *digits, int len)
{
    return q_put(&s->queue, digits, len);
}

ORC_API int orc_bell_mf_tx(orc_bell_mf_tx_t *s, int16_t amp[], int max_samples)
{
    int len = 0;
    int digit;

    if (s->tones.current_section >= 0)
        len = orc_tone_gen(&s->tones, amp, max_samples);
    while (len < max_samples  &&  (digit = q_get(&s->queue)) >= 0)
    {
        const char *at;
        orc_tone_desc_t d;
        int k;

        /* (a NUL digit makes the reference index one past its table, bell_r2_mf.c:322; skipped here) */
        if (digit == 0  ||  (at = strchr(bell_keys, digit)) == NULL)
            continue;
        k = (int) (at - bell_keys);
        /* bell_r2_mf.c:104-121: -7 dBm0 each, 68 ms on / 68 ms off, KP ('*') 100 ms on */
        orc_tone_desc_init(&d, bell_pairs[k][0], -7, bell_pairs[k][1], -7, (digit == '*')  ?  100  :  68, 68, 0, 0, 0);
        orc_tone_gen_init(&s->tones, &d);
        len += orc_tone_gen(&s->tones, amp + len, max_samples - len);
    }
    return len;
}

/* ---- r2_mf_tx ---- */
static const char r2_keys[] = "1234567890BCDEF";
static const int r2_fwd_pairs[15][2] =
{
    {1380, 1500}, {1380, 1620}, {1500, 1620}, {1380, 1740}, {1500, 1740}, {1620, 1740}, {1380, 1860}, {1500, 1860},
    {1620, 1860}, {1740, 1860}, {1380, 1980}, {1500, 1980}, {1620, 1980}, {1740, 1980}, {1860, 1980}
};
static const int r2_back_pairs[15][2] =
{
    {1140, 1020}, {1140, 900}, {1020, 900}, {1140, 780}, {1020, 780}, {900, 780}, {1140, 660}, {1020, 660},
    {900, 660}, {780, 660}, {1140, 540}, {1020, 540}, {900, 540}, {780, 540}, {660, 540}
};

ORC_API void orc_r2_mf_tx_init(orc_r2_mf_tx_t *s, int fwd)
{
    memset(s, 0, sizeof(*s));
    s->fwd = fwd;
}

ORC_API int orc_r2_mf_tx_put(orc_r2_mf_tx_t *s, char digit)
{
    const char *at;

    if (digit  &&  (at = strchr(r2_keys, digit)))
    {
        const int (*pairs)[2] = s->fwd  ?  r2_fwd_pairs  :  r2_back_pairs;
        const int k = (int) (at - r2_keys);
        orc_tone_desc_t d;

        /* bell_r2_mf.c:143-181,447-477: -11 dBm0 each, 1 ms section repeated for ever */
        orc_tone_desc_init(&d, pairs[k][0], -11, pairs[k][1], -11, 1, 0, 0, 0, 1);
        orc_tone_gen_init(&s->tone, &d);
        s->digit = digit;
    }
    else
    {
        s->digit = 0;
    }
    return 0;
}

ORC_API int orc_r2_mf_tx(orc_r2_mf_tx_t *s, int16_t amp[], int samples)
{
    if (s->digit == 0)
    {
        memset(amp, 0, sizeof(int16_t)*samples);
        return samples;
    }
    return orc_tone_gen(&s->tone, amp, samples);
}

/* Bench helper: `frames` calls of orc_dtmf_tx() on each of n senders (contiguous array), each call writing
   `samples` samples at amp + channel*stride (the buffer is reused from frame to frame). */
ORC_API long long orc_dtmf_tx_run_batch(orc_dtmf_tx_t *s, int n, int16_t *amp, long long stride, int samples, int frames)
{
    long long total = 0;

    for (int f = 0;  f < frames;  f++)
    {
        for (int c = 0;  c < n;  c++)
            total += orc_dtmf_tx(&s[c], amp + c*stride, samples);
    }
    return total;
}
