/*
 * tone_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's Goertzel bank detectors:
 *   Goertzel primitive   src/tone_detect.c:60-205, src/spandsp/tone_detect.h:129-192
 *   DTMF receiver        src/dtmf.c:104-123 (constants), :132-361 (rx), :363-445
 *   Bell MF receiver     src/bell_r2_mf.c:236-262 (constants), :507-673
 *   R2 MF receiver       src/bell_r2_mf.c:240-276 (constants), :750-880
 *   Super-tone receiver  src/super_tone_rx.c:75-77 (constants), :81-228, :289-490
 *
 * Numeric contract: IEEE binary32, every multiply/add individually rounded
 * (compiled -ffp-contract=off), evaluation order exactly as the reference
 * writes it.  Arrays replace the reference's named goertzel_state_t members;
 * observable actions go to an orc_sink_t instead of user callbacks.
 */
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <math.h>

#include "oracle.h"

/* ------------------------------------------------------------------------- */
/* Event sink                                                                */
/* ------------------------------------------------------------------------- */
orc_sink_t *orc_sink_new(void)
{
    orc_sink_t *k = (orc_sink_t *) calloc(1, sizeof(*k));

    k->cap = 256;
    k->ev = (orc_event_t *) malloc(sizeof(orc_event_t)*k->cap);
    k->captext = 256;
    k->text = (char *) malloc(k->captext);
    k->text[0] = '\0';
    return k;
}

void orc_sink_free(orc_sink_t *k)
{
    if (k)
    {
        free(k->ev);
        free(k->text);
        free(k);
    }
}

void orc_sink_clear(orc_sink_t *k)
{
    k->n = 0;
    k->ntext = 0;
    k->text[0] = '\0';
}

int orc_sink_count(const orc_sink_t *k) { return k->n; }
const orc_event_t *orc_sink_events(const orc_sink_t *k) { return k->ev; }
int orc_sink_ntext(const orc_sink_t *k) { return k->ntext; }
const char *orc_sink_text(const orc_sink_t *k) { return k->text; }

void orc_sink_want_qam(orc_sink_t *k, int on)
{
    k->want_qam = on;
}

void orc_sink_qam(orc_sink_t *k, const float *constel, const float *target, int symbol)
{
    int32_t w[4] = {0, 0, 0, 0};

    if (k == NULL  ||  !k->want_qam)
        return;
    if (constel)
    {
        memcpy(&w[0], &constel[0], 4);
        memcpy(&w[1], &constel[1], 4);
        memcpy(&w[2], &target[0], 4);
        memcpy(&w[3], &target[1], 4);
    }
    orc_sink_push(k, 6, symbol, w[0], w[1]);
    orc_sink_push(k, 7, (constel == NULL), w[2], w[3]);
}

void orc_sink_push(orc_sink_t *k, int kind, int a, int b, int c)
{
    if (k == NULL)
        return;
    if (k->n == k->cap)
    {
        k->cap *= 2;
        k->ev = (orc_event_t *) realloc(k->ev, sizeof(orc_event_t)*k->cap);
    }
    k->ev[k->n].kind = kind;
    k->ev[k->n].a = a;
    k->ev[k->n].b = b;
    k->ev[k->n].c = c;
    k->n++;
}

void orc_sink_text_append(orc_sink_t *k, const char *s, int len)
{
    if (k == NULL)
        return;
    while (k->ntext + len + 1 > k->captext)
    {
        k->captext *= 2;
        k->text = (char *) realloc(k->text, k->captext);
    }
    memcpy(k->text + k->ntext, s, len);
    k->ntext += len;
    k->text[k->ntext] = '\0';
}

/* ------------------------------------------------------------------------- */
/* Goertzel primitive                                                        */
/* ------------------------------------------------------------------------- */

/* tone_detect.c:60-68.  M_PI is a double, so 2.0f*M_PI*(freq/8000.0f) is
   evaluated in double and narrowed to float at the cosf() call. */
float orc_goertzel_fac(float freq)
{
    const double two_pi = 2.0f*3.14159265358979323846264338327;
    float ratio = freq/8000.0f;
    float arg = (float) (two_pi*ratio);

    return 2.0f*cosf(arg);
}

/* One step of the recurrence, tone_detect.h:172-192: v3' = (fac*v2 - v1) + x */
static inline void bin_step(float *v2, float *v3, float fac, float x)
{
    float v1 = *v2;

    *v2 = *v3;
    *v3 = fac*(*v2) - v1 + x;
}

/* tone_detect.c:160-205: push one zero sample, evaluate, reset. */
static inline float bin_finish(float *v2, float *v3, float fac)
{
    float v1 = *v2;
    float r;

    *v2 = *v3;
    *v3 = fac*(*v2) - v1;
    r = (*v3)*(*v3) + (*v2)*(*v2) - (*v2)*(*v3)*fac;
    r *= 2.0f;
    *v2 = 0.0f;
    *v3 = 0.0f;
    return r;
}

/* tone_detect.c:71-121 */
void orc_goertzel_init(orc_goertzel_t *g, float freq, int samples)
{
    g->v2 = 0.0f;
    g->v3 = 0.0f;
    g->fac = orc_goertzel_fac(freq);
    g->samples = samples;
    g->current_sample = 0;
}

/* tone_detect.c:123-156 */
int orc_goertzel_update(orc_goertzel_t *g, const int16_t amp[], int samples)
{
    int room = g->samples - g->current_sample;
    int i;

    if (samples > room)
        samples = room;
    for (i = 0;  i < samples;  i++)
        bin_step(&g->v2, &g->v3, g->fac, (float) amp[i]);
    g->current_sample += samples;
    return samples;
}

/* tone_detect.c:160-205 */
float orc_goertzel_result(orc_goertzel_t *g)
{
    g->current_sample = 0;
    return bin_finish(&g->v2, &g->v3, g->fac);
}

/* ------------------------------------------------------------------------- */
/* DTMF                                                                      */
/* ------------------------------------------------------------------------- */
#define DTMF_BLOCK              102                 /* dtmf.c:71 */
static const float DTMF_THRESHOLD       = 171029200.0f;     /* dtmf.c:104 */
static const float DTMF_NORMAL_TWIST    = 6.309f;           /* dtmf.c:105 */
static const float DTMF_REVERSE_TWIST   = 2.512f;           /* dtmf.c:106 */
static const float DTMF_REL_PEAK_ROW    = 6.309f;           /* dtmf.c:107 */
static const float DTMF_REL_PEAK_COL    = 6.309f;           /* dtmf.c:108 */
static const float DTMF_TO_TOTAL        = 83.868f;          /* dtmf.c:109 */
static const float DTMF_POWER_OFFSET    = 107.255f;         /* dtmf.c:110 */
static const float DTMF_FREQS[8] =                          /* dtmf.c:112-119 */
{
    697.0f, 770.0f, 852.0f, 941.0f, 1209.0f, 1336.0f, 1477.0f, 1633.0f
};
static const char DTMF_KEYS[] = "123A456B789C*0#D";         /* dtmf.c:121 */

int orc_dtmf_sizeof(void) { return (int) sizeof(orc_dtmf_t); }

/* dtmf.c:447-504 */
void orc_dtmf_init(orc_dtmf_t *s, int mode)
{
    int i;

    memset(s, 0, sizeof(*s));
    for (i = 0;  i < 8;  i++)
        s->fac[i] = orc_goertzel_fac(DTMF_FREQS[i]);
    s->threshold = DTMF_THRESHOLD;
    s->normal_twist = DTMF_NORMAL_TWIST;
    s->reverse_twist = DTMF_REVERSE_TWIST;
    s->mode = mode;
}

/* dtmf.c:421-445 */
void orc_dtmf_parms(orc_dtmf_t *s, int filter_dialtone, float twist, float reverse_twist, float threshold)
{
    if (filter_dialtone >= 0)
    {
        s->z350[0] = s->z350[1] = 0.0f;
        s->z440[0] = s->z440[1] = 0.0f;
        s->filter_dialtone = filter_dialtone;
    }
    if (twist >= 0.0f)
        s->normal_twist = powf(10.0f, twist/10.0f);
    if (reverse_twist >= 0.0f)
        s->reverse_twist = powf(10.0f, reverse_twist/10.0f);
    if (threshold > -99.0f)
        s->threshold = (float) ((DTMF_BLOCK*DTMF_BLOCK*32768.0f*32768.0f/2.0f)*powf(10.0f, (threshold - 3.14f)/10.0f));
}

/* Block-end decision, dtmf.c:209-258.  e[0..3] rows, e[4..7] columns. */
static int dtmf_block_decide(const orc_dtmf_t *s, const float e[8])
{
    int best_row = 0;
    int best_col = 0;
    int i;

    for (i = 1;  i < 4;  i++)
    {
        if (e[i] > e[best_row])
            best_row = i;
        if (e[4 + i] > e[4 + best_col])
            best_col = i;
    }
    if (!(e[best_row] >= s->threshold  &&  e[4 + best_col] >= s->threshold))
        return 0;
    if (!(e[4 + best_col] < e[best_row]*s->reverse_twist  &&  e[4 + best_col]*s->normal_twist > e[best_row]))
        return 0;
    for (i = 0;  i < 4;  i++)
    {
        if ((i != best_col  &&  e[4 + i]*DTMF_REL_PEAK_COL > e[4 + best_col])
            ||
            (i != best_row  &&  e[i]*DTMF_REL_PEAK_ROW > e[best_row]))
        {
            return 0;
        }
    }
    if (!((e[best_row] + e[4 + best_col]) > DTMF_TO_TOTAL*s->energy))
        return 0;
    return DTMF_KEYS[(best_row << 2) + best_col];
}

/* dtmf.c:132-361 */
int orc_dtmf_rx(orc_dtmf_t *s, const int16_t amp[], int samples, orc_sink_t *sink, orc_block_t *blocks, int max_blocks)
{
    int pos = 0;
    int nblocks = 0;
    int take;
    int i;
    int j;
    int hit;
    float x;
    float f;
    float v1;
    float e[8];

    while (pos < samples)
    {
        take = DTMF_BLOCK - s->current_sample;
        if (take > samples - pos)
            take = samples - pos;
        for (j = 0;  j < take;  j++)
        {
            x = (float) amp[pos + j];
            if (s->filter_dialtone)
            {
                /* dtmf.c:167-183: two notch biquads, float all the way */
                f = x;
                v1 = 0.98356f*f + 1.8954426f*s->z350[0] - 0.9691396f*s->z350[1];
                f = v1 - 1.9251480f*s->z350[0] + s->z350[1];
                s->z350[1] = s->z350[0];
                s->z350[0] = v1;
                v1 = 0.98456f*f + 1.8529543f*s->z440[0] - 0.9691396f*s->z440[1];
                f = v1 - 1.8819938f*s->z440[0] + s->z440[1];
                s->z440[1] = s->z440[0];
                s->z440[0] = v1;
                x = f;
            }
            s->energy += x*x;
            /* The reference interleaves row/col updates; the bins are independent. */
            for (i = 0;  i < 8;  i++)
                bin_step(&s->v2[i], &s->v3[i], s->fac[i], x);
        }
        if (s->duration < INT_MAX - take)
            s->duration += take;
        s->current_sample += take;
        pos += take;
        if (s->current_sample < DTMF_BLOCK)
            continue;

        for (i = 0;  i < 8;  i++)
            e[i] = bin_finish(&s->v2[i], &s->v3[i], s->fac[i]);
        hit = dtmf_block_decide(s, e);
        if (blocks  &&  nblocks < max_blocks)
        {
            memset(&blocks[nblocks], 0, sizeof(blocks[0]));
            blocks[nblocks].hit = hit;
            blocks[nblocks].total_energy = s->energy;
            memcpy(blocks[nblocks].e, e, sizeof(e));
        }
        /* Debounce, dtmf.c:304-347 */
        if (hit != s->in_digit  &&  s->last_hit != s->in_digit)
        {
            hit = (hit  &&  hit == s->last_hit)  ?  hit  :  0;
            if (s->mode == 2)
            {
                if (s->in_digit  ||  hit)
                {
                    i = (s->in_digit  &&  !hit)  ?  -99  :  (int) (long int) (10.0f*log10f(s->energy) - DTMF_POWER_OFFSET);
                    orc_sink_push(sink, 1, hit, i, s->duration);
                    s->duration = 0;
                }
            }
            else if (hit)
            {
                if (s->current_digits < 128)
                {
                    s->digits[s->current_digits++] = (char) hit;
                    s->digits[s->current_digits] = '\0';
                    if (s->mode == 1)
                    {
                        orc_sink_text_append(sink, s->digits, s->current_digits);
                        orc_sink_push(sink, 2, s->current_digits, 0, 0);
                        s->current_digits = 0;
                    }
                }
                else
                {
                    s->lost_digits++;
                }
            }
            s->in_digit = hit;
        }
        s->last_hit = hit;
        s->energy = 0.0f;
        s->current_sample = 0;
        if (blocks  &&  nblocks < max_blocks)
            blocks[nblocks].aux = s->in_digit;
        nblocks++;
    }
    /* dtmf.c:352-358 */
    if (s->current_digits  &&  s->mode == 1)
    {
        orc_sink_text_append(sink, s->digits, s->current_digits);
        orc_sink_push(sink, 2, s->current_digits, 0, 0);
        s->digits[0] = '\0';
        s->current_digits = 0;
    }
    return nblocks;
}

void orc_dtmf_rx_batch(orc_dtmf_t *s, const int16_t amp[], int n, long long stride, int samples)
{
    int c;

    for (c = 0;  c < n;  c++)
        orc_dtmf_rx(&s[c], amp + c*stride, samples, NULL, NULL, 0);
}

/* dtmf.c:394-408 */
int orc_dtmf_get(orc_dtmf_t *s, char *buf, int max)
{
    if (max > s->current_digits)
        max = s->current_digits;
    if (max > 0)
    {
        memcpy(buf, s->digits, max);
        memmove(s->digits, s->digits + max, s->current_digits - max);
        s->current_digits -= max;
    }
    buf[max] = '\0';
    return max;
}

/* dtmf.c:382-391 */
int orc_dtmf_status(const orc_dtmf_t *s)
{
    if (s->in_digit)
        return s->in_digit;
    if (s->last_hit)
        return 'x';
    return 0;
}

/* dtmf.c:363-379 */
void orc_dtmf_fillin(orc_dtmf_t *s, int samples)
{
    int i;

    (void) samples;
    for (i = 0;  i < 8;  i++)
    {
        s->v2[i] = 0.0f;
        s->v3[i] = 0.0f;
    }
    s->energy = 0.0f;
    s->current_sample = 0;
}

/* ------------------------------------------------------------------------- */
/* Bell MF and R2 MF                                                         */
/* ------------------------------------------------------------------------- */
#define BELL_MF_BLOCK           120                 /* bell_r2_mf.c:204 */
#define R2_MF_BLOCK             133                 /* bell_r2_mf.c:206 */
static const float BELL_MF_THRESHOLD    = 3343803100.0f;    /* bell_r2_mf.c:236 */
static const float BELL_MF_TWIST        = 3.981f;           /* :237 */
static const float BELL_MF_REL_PEAK     = 12.589f;          /* :238 */
static const float R2_MF_THRESHOLD      = 1031766650.0f;    /* :240 */
static const float R2_MF_TWIST          = 5.012f;           /* :241 */
static const float R2_MF_REL_PEAK       = 12.589f;          /* :242 */
static const int BELL_MF_FREQS[6] = {700, 900, 1100, 1300, 1500, 1700};         /* :251-254 */
static const int R2_FWD_FREQS[6] = {1380, 1500, 1620, 1740, 1860, 1980};        /* :264-267 */
static const int R2_BACK_FREQS[6] = {1140, 1020, 900, 780, 660, 540};           /* :269-272 */
static const char BELL_MF_KEYS[] = "1247C-358A--69*---0B----#";                 /* :262 */
static const char R2_MF_KEYS[] = "1247B-358C--69D---0E----F";                   /* :276 */

/* Two strongest of six bins + level, twist and relative peak tests.
   bell_r2_mf.c:556-622 and :793-858 are the same code with different constants.
   Returns the index into the 25-char key table, or -1. */
static int mf_pick_pair(const float e[6], float threshold, float twist, float rel_peak)
{
    int best;
    int second;
    int i;

    if (e[0] > e[1])
    {
        best = 0;
        second = 1;
    }
    else
    {
        best = 1;
        second = 0;
    }
    for (i = 2;  i < 6;  i++)
    {
        if (e[i] >= e[best])
        {
            second = best;
            best = i;
        }
        else if (e[i] >= e[second])
        {
            second = i;
        }
    }
    if (!(e[best] >= threshold
          &&  e[second] >= threshold
          &&  e[best] < e[second]*twist
          &&  e[best]*twist > e[second]))
    {
        return -1;
    }
    for (i = 0;  i < 6;  i++)
    {
        if (i != best  &&  i != second  &&  e[i]*rel_peak >= e[second])
            return -1;
    }
    if (second < best)
    {
        i = best;
        best = second;
        second = i;
    }
    return best*5 + second - 1;
}

int orc_bell_mf_sizeof(void) { return (int) sizeof(orc_bell_mf_t); }

/* bell_r2_mf.c:693-735 */
void orc_bell_mf_init(orc_bell_mf_t *s, int mode)
{
    int i;

    memset(s, 0, sizeof(*s));
    for (i = 0;  i < 6;  i++)
        s->fac[i] = orc_goertzel_fac((float) BELL_MF_FREQS[i]);
    s->mode = mode;
}

/* bell_r2_mf.c:507-673 */
int orc_bell_mf_rx(orc_bell_mf_t *s, const int16_t amp[], int samples, orc_sink_t *sink, orc_block_t *blocks, int max_blocks)
{
    int pos = 0;
    int nblocks = 0;
    int take;
    int i;
    int j;
    int idx;
    int hit;
    int accepted;
    float x;
    float e[6];

    while (pos < samples)
    {
        take = BELL_MF_BLOCK - s->current_sample;
        if (take > samples - pos)
            take = samples - pos;
        for (j = 0;  j < take;  j++)
        {
            x = (float) amp[pos + j];
            for (i = 0;  i < 6;  i++)
                bin_step(&s->v2[i], &s->v3[i], s->fac[i], x);
        }
        s->current_sample += take;
        pos += take;
        if (s->current_sample < BELL_MF_BLOCK)
            continue;

        for (i = 0;  i < 6;  i++)
            e[i] = bin_finish(&s->v2[i], &s->v3[i], s->fac[i]);
        idx = mf_pick_pair(e, BELL_MF_THRESHOLD, BELL_MF_TWIST, BELL_MF_REL_PEAK);
        hit = 0;
        accepted = 0;
        if (idx >= 0)
        {
            hit = (uint8_t) BELL_MF_KEYS[idx];
            /* bell_r2_mf.c:629-635 */
            if (hit == s->hits[4]
                &&  hit == s->hits[3]
                &&  ((hit != '*'  &&  hit != s->hits[2]  &&  hit != s->hits[1])
                     ||
                     (hit == '*'  &&  hit == s->hits[2]  &&  hit != s->hits[1]  &&  hit != s->hits[0])))
            {
                accepted = hit;
                if (s->current_digits < 128)
                {
                    s->digits[s->current_digits++] = (char) hit;
                    s->digits[s->current_digits] = '\0';
                    if (s->mode == 1)
                    {
                        orc_sink_text_append(sink, s->digits, s->current_digits);
                        orc_sink_push(sink, 2, s->current_digits, 0, 0);
                        s->current_digits = 0;
                    }
                }
                else
                {
                    s->lost_digits++;
                }
            }
        }
        if (blocks  &&  nblocks < max_blocks)
        {
            memset(&blocks[nblocks], 0, sizeof(blocks[0]));
            blocks[nblocks].hit = hit;
            blocks[nblocks].aux = accepted;
            memcpy(blocks[nblocks].e, e, sizeof(e));
        }
        s->hits[0] = s->hits[1];
        s->hits[1] = s->hits[2];
        s->hits[2] = s->hits[3];
        s->hits[3] = s->hits[4];
        s->hits[4] = hit;
        s->current_sample = 0;
        nblocks++;
    }
    if (s->current_digits  &&  s->mode == 1)
    {
        orc_sink_text_append(sink, s->digits, s->current_digits);
        orc_sink_push(sink, 2, s->current_digits, 0, 0);
        s->digits[0] = '\0';
        s->current_digits = 0;
    }
    return nblocks;
}

/* bell_r2_mf.c:675-689 */
int orc_bell_mf_get(orc_bell_mf_t *s, char *buf, int max)
{
    if (max > s->current_digits)
        max = s->current_digits;
    if (max > 0)
    {
        memcpy(buf, s->digits, max);
        memmove(s->digits, s->digits + max, s->current_digits - max);
        s->current_digits -= max;
    }
    buf[max] = '\0';
    return max;
}

int orc_r2_mf_sizeof(void) { return (int) sizeof(orc_r2_mf_t); }

/* bell_r2_mf.c:889-937 */
void orc_r2_mf_init(orc_r2_mf_t *s, int fwd, int use_callback)
{
    int i;

    memset(s, 0, sizeof(*s));
    s->fwd = fwd;
    for (i = 0;  i < 6;  i++)
        s->fac[i] = orc_goertzel_fac((float) ((fwd)  ?  R2_FWD_FREQS[i]  :  R2_BACK_FREQS[i]));
    s->use_callback = use_callback;
}

/* bell_r2_mf.c:750-880 */
int orc_r2_mf_rx(orc_r2_mf_t *s, const int16_t amp[], int samples, orc_sink_t *sink, orc_block_t *blocks, int max_blocks)
{
    int pos = 0;
    int nblocks = 0;
    int take;
    int i;
    int j;
    int idx;
    int digit;
    float x;
    float e[6];

    while (pos < samples)
    {
        take = R2_MF_BLOCK - s->current_sample;
        if (take > samples - pos)
            take = samples - pos;
        for (j = 0;  j < take;  j++)
        {
            x = (float) amp[pos + j];
            for (i = 0;  i < 6;  i++)
                bin_step(&s->v2[i], &s->v3[i], s->fac[i], x);
        }
        s->current_sample += take;
        pos += take;
        if (s->current_sample < R2_MF_BLOCK)
            continue;

        for (i = 0;  i < 6;  i++)
            e[i] = bin_finish(&s->v2[i], &s->v3[i], s->fac[i]);
        idx = mf_pick_pair(e, R2_MF_THRESHOLD, R2_MF_TWIST, R2_MF_REL_PEAK);
        digit = (idx >= 0)  ?  R2_MF_KEYS[idx]  :  0;
        /* bell_r2_mf.c:869-876 */
        if (s->current_digit != digit  &&  s->use_callback)
            orc_sink_push(sink, 1, digit, (digit)  ?  -10  :  -99, 0);
        s->current_digit = digit;
        s->current_sample = 0;
        if (blocks  &&  nblocks < max_blocks)
        {
            memset(&blocks[nblocks], 0, sizeof(blocks[0]));
            blocks[nblocks].hit = digit;
            blocks[nblocks].aux = digit;
            memcpy(blocks[nblocks].e, e, sizeof(e));
        }
        nblocks++;
    }
    return nblocks;
}

/* ------------------------------------------------------------------------- */
/* Super tone                                                                */
/* ------------------------------------------------------------------------- */
#define ST_BLOCK                128                 /* private/super_tone_rx.h:29 */
static const float ST_THRESHOLD     = 2104205.6f;   /* super_tone_rx.c:75 */
static const float ST_TWIST         = 3.981f;       /* :76 */
static const float ST_TO_TOTAL      = 1.995f;       /* :77 */

int orc_st_desc_sizeof(void) { return (int) sizeof(orc_st_desc_t); }
int orc_st_sizeof(void) { return (int) sizeof(orc_st_t); }

/* super_tone_rx.c:231-248 */
void orc_st_desc_init(orc_st_desc_t *d)
{
    memset(d, 0, sizeof(*d));
}

/* super_tone_rx.c:81-123.  Note the reference stores the *pitch index* i (not the
   bin number) in pitches[][1] for a merged entry; kept as is. */
static int st_add_freq(orc_st_desc_t *d, int freq)
{
    int i;

    if (freq == 0)
        return -1;
    for (i = 0;  i < d->used_frequencies;  i++)
    {
        if (d->pitches[i][0] == freq)
            return d->pitches[i][1];
    }
    for (i = 0;  i < d->used_frequencies;  i++)
    {
        if ((d->pitches[i][0] - 10) <= freq  &&  freq <= (d->pitches[i][0] + 10))
        {
            d->pitches[d->used_frequencies][0] = freq;
            d->pitches[d->used_frequencies][1] = i;
            d->fac[d->pitches[i][1]] = orc_goertzel_fac((float) (freq + d->pitches[i][0])/2);
            d->used_frequencies++;
            return d->pitches[i][1];
        }
    }
    d->pitches[i][0] = freq;
    d->pitches[i][1] = d->monitored_frequencies;
    d->fac[d->monitored_frequencies++] = orc_goertzel_fac((float) freq);
    d->used_frequencies++;
    return d->pitches[i][1];
}

/* super_tone_rx.c:125-138 */
int orc_st_add_tone(orc_st_desc_t *d)
{
    if (d->tones >= ORC_ST_MAX_TONES)
        return -1;
    d->steps[d->tones] = 0;
    return d->tones++;
}

/* super_tone_rx.c:140-162 */
int orc_st_add_element(orc_st_desc_t *d, int tone, int f1, int f2, int min_ms, int max_ms)
{
    int step = d->steps[tone];

    if (step >= ORC_ST_MAX_STEPS)
        return -1;
    d->list[tone][step].f1 = st_add_freq(d, f1);
    d->list[tone][step].f2 = st_add_freq(d, f2);
    d->list[tone][step].min_duration = min_ms*8;
    d->list[tone][step].max_duration = (max_ms == 0)  ?  0x7FFFFFFF  :  max_ms*8;
    d->steps[tone]++;
    return step;
}

/* super_tone_rx.c:507-554 */
void orc_st_init(orc_st_t *s, const orc_st_desc_t *d, int use_segment_cb)
{
    int i;

    memset(s, 0, sizeof(*s));
    s->desc = d;
    for (i = 0;  i < 11;  i++)
    {
        s->seg[i].f1 = -1;
        s->seg[i].f2 = -1;
        s->seg[i].min_duration = 0;
    }
    s->detected_tone = -1;
    s->use_segment_cb = use_segment_cb;
}

/* super_tone_rx.c:164-228.  `s` supplies the 11-entry segment history. */
static int st_test_cadence(const orc_st_t *s, const orc_st_step_t *pattern, int steps, int rotation)
{
    int i;
    int j;

    if (rotation >= 0)
    {
        j = 0;
        if (steps < 0)
        {
            steps = -steps;
            j = (rotation + steps - 2)%steps;
            if (pattern[j].f1 != s->seg[8].f1  ||  pattern[j].f2 != s->seg[8].f2)
                return 0;
            if (pattern[j].min_duration > s->seg[8].min_duration*ST_BLOCK
                ||  pattern[j].max_duration < s->seg[8].min_duration*ST_BLOCK)
            {
                return 0;
            }
        }
        if (steps)
            j = (rotation + steps - 1)%steps;
        if (pattern[j].f1 != s->seg[9].f1  ||  pattern[j].f2 != s->seg[9].f2)
            return 0;
        if (pattern[j].max_duration < s->seg[9].min_duration*ST_BLOCK)
            return 0;
    }
    else
    {
        for (i = 0;  i < steps;  i++)
        {
            j = i + 10 - steps;
            if (pattern[i].f1 != s->seg[j].f1  ||  pattern[i].f2 != s->seg[j].f2)
                return 0;
            if (pattern[i].min_duration > s->seg[j].min_duration*ST_BLOCK
                ||  pattern[i].max_duration < s->seg[j].min_duration*ST_BLOCK)
            {
                return 0;
            }
        }
    }
    return 1;
}

/* super_tone_rx.c:289-451 */
static void st_chunk(orc_st_t *s, orc_sink_t *sink, orc_block_t *blk)
{
    const orc_st_desc_t *d = s->desc;
    int m = d->monitored_frequencies;
    int i;
    int j;
    int k1;
    int k2;
    float res[64];

    memset(res, 0, sizeof(res));
    if (s->energy < ST_THRESHOLD)
    {
        k1 = -1;
        k2 = -1;
        for (i = 0;  i < m;  i++)
        {
            s->v2[i] = 0.0f;
            s->v3[i] = 0.0f;
        }
    }
    else
    {
        /* The reference's monitored_frequencies < 2 branch (:312-316) never
           resets the bins and so can never finish a second block; the engine
           requires >= 2 monitored frequencies and so does this oracle. */
        for (i = 0;  i < m;  i++)
            res[i] = bin_finish(&s->v2[i], &s->v3[i], d->fac[i]);
        if (res[0] > res[1])
        {
            k1 = 0;
            k2 = 1;
        }
        else
        {
            k1 = 1;
            k2 = 0;
        }
        for (j = 2;  j < m;  j++)
        {
            if (res[j] >= res[k1])
            {
                k2 = k1;
                k1 = j;
            }
            else if (res[j] >= res[k2])
            {
                k2 = j;
            }
        }
        if ((res[k1] + res[k2]) < ST_TO_TOTAL*s->energy)
        {
            k1 = -1;
            k2 = -1;
        }
        else if (res[k1] > ST_TWIST*res[k2])
        {
            k2 = -1;
        }
        else if (k2 < k1)
        {
            j = k1;
            k1 = k2;
            k2 = j;
        }
    }
    if (blk)
    {
        memset(blk, 0, sizeof(*blk));
        blk->hit = k1;
        blk->aux = k2;
        blk->total_energy = s->energy;
        memcpy(blk->e, res, sizeof(float)*m);
    }
    s->current_sample = 0;

    if (k1 != s->seg[10].f1  ||  k2 != s->seg[10].f2)
    {
        s->seg[10].f1 = k1;
        s->seg[10].f2 = k2;
        s->seg[9].min_duration++;
    }
    else
    {
        if (k1 != s->seg[9].f1  ||  k2 != s->seg[9].f2)
        {
            if (s->detected_tone >= 0)
            {
                if (!st_test_cadence(s, d->list[s->detected_tone], -d->steps[s->detected_tone], s->rotation++))
                {
                    s->detected_tone = -1;
                    orc_sink_push(sink, 1, s->detected_tone, -10, 0);
                }
            }
            if (s->use_segment_cb)
                orc_sink_push(sink, 4, s->seg[9].f1, s->seg[9].f2, s->seg[9].min_duration*ST_BLOCK/8);
            memmove(&s->seg[0], &s->seg[1], 9*sizeof(s->seg[0]));
            s->seg[9].f1 = k1;
            s->seg[9].f2 = k2;
            s->seg[9].min_duration = 1;
        }
        else
        {
            if (s->detected_tone >= 0)
            {
                if (!st_test_cadence(s, d->list[s->detected_tone], d->steps[s->detected_tone], s->rotation))
                {
                    s->detected_tone = -1;
                    orc_sink_push(sink, 1, s->detected_tone, -10, 0);
                }
            }
            s->seg[9].min_duration++;
        }
    }
    if (s->detected_tone < 0)
    {
        for (j = 0;  j < d->tones;  j++)
        {
            if (st_test_cadence(s, d->list[j], d->steps[j], -1))
            {
                s->detected_tone = j;
                s->rotation = 0;
                orc_sink_push(sink, 1, s->detected_tone, -10, 0);
                break;
            }
        }
    }
    s->energy = 0.0f;
}

/* super_tone_rx.c:454-490 */
int orc_st_rx(orc_st_t *s, const int16_t amp[], int samples, orc_sink_t *sink, orc_block_t *blocks, int max_blocks)
{
    const orc_st_desc_t *d = s->desc;
    int m = d->monitored_frequencies;
    int pos = 0;
    int nblocks = 0;
    int take;
    int i;
    int j;
    float x;

    while (pos < samples)
    {
        take = ST_BLOCK - s->current_sample;
        if (take > samples - pos)
            take = samples - pos;
        /* Each bin runs over the whole sub-block (goertzel_update per bin, :468-470),
           then the energy is accumulated over the same samples (:471-480). */
        for (i = 0;  i < m;  i++)
        {
            for (j = 0;  j < take;  j++)
                bin_step(&s->v2[i], &s->v3[i], d->fac[i], (float) amp[pos + j]);
        }
        for (j = 0;  j < take;  j++)
        {
            x = (float) amp[pos + j];
            s->energy += x*x;
        }
        s->current_sample += take;
        pos += take;
        if (s->current_sample >= ST_BLOCK)
        {
            st_chunk(s, sink, (blocks  &&  nblocks < max_blocks)  ?  &blocks[nblocks]  :  NULL);
            nblocks++;
        }
    }
    return nblocks;
}


/* ---- G.711 decode (spandsp/g711.h:165-175, :239-252): the front end the tone banks can fuse ---- */
int16_t orc_ulaw_to_linear(uint8_t ulaw)
{
    int t;

    ulaw = (uint8_t) ~ulaw;
    t = (((ulaw & 0x0F) << 3) + 0x84) << (((int) ulaw & 0x70) >> 4);
    return (int16_t) ((ulaw & 0x80)  ?  (0x84 - t)  :  (t - 0x84));
}

int16_t orc_alaw_to_linear(uint8_t alaw)
{
    int i;
    int seg;

    alaw ^= 0x55;
    i = ((alaw & 0x0F) << 4);
    seg = (((int) alaw & 0x70) >> 4);
    if (seg)
        i = (i + 0x108) << (seg - 1);
    else
        i += 8;
    return (int16_t) ((alaw & 0x80)  ?  i  :  -i);
}

/* ---- the Goertzel users outside tone_detect.c: raw block decisions ---------------------------- */

/* src/v18.c:1580-1600 (caller_tone_scan; answerer_tone_scan :1721-1741 is the same): the strongest of the n tone set
   energies by a strict > scan from zero, then the level test against the object's threshold (never assigned in this
   snapshot: 0) and the fraction-of-total-energy test (tone_to_total_energy = 83.868, :192).  0 stands for "no tone" and
   for tone set entry 0 alike, as in the reference. */
ORC_API int orc_v18_tone_decide(const float e[], int n, float total, float threshold)
{
    float best = 0.0f;
    int at = 0;

    for (int i = 0;  i < n;  i++)
    {
        if (e[i] > best)
        {
            best = e[i];
            at = i;
        }
    }
    if (best < threshold  ||  best <= 83.868f*total)
        at = 0;
    return at;
}

/* src/ademco_contactid.c:915-935: 1 = 1400 Hz, 2 = 2300 Hz, 0 = neither (detection_threshold :461, tone_to_total_energy :462) */
ORC_API int orc_ademco_tone_decide(float e1400, float e2300, float total)
{
    int hit = 0;

    if (e1400 > 49728296.6f  ||  e2300 > 49728296.6f)
    {
        if (e1400 > e2300)
        {
            if (e1400 > 45.2233f*total)
                hit = 1;
        }
        else
        {
            if (e2300 > 45.2233f*total)
                hit = 2;
        }
    }
    return hit;
}

/* One detector of either kind over whole blocks: decisions[b] for n_blocks blocks of block_len samples */
ORC_API void orc_tone_functor_blocks(int kind, const float freqs[], int n_freqs, int block_len, float threshold,
                                     const int16_t amp[], int n_blocks, int32_t decisions[])
{
    orc_goertzel_t g[16];
    float e[16];

    for (int i = 0;  i < n_freqs;  i++)
        orc_goertzel_init(&g[i], freqs[i], block_len);
    for (int b = 0;  b < n_blocks;  b++)
    {
        float total = 0.0f;
        for (int j = 0;  j < block_len;  j++)
        {
            const float x = amp[b*block_len + j];
            total += x*x;
        }
        for (int i = 0;  i < n_freqs;  i++)
        {
            orc_goertzel_update(&g[i], amp + b*block_len, block_len);
            e[i] = orc_goertzel_result(&g[i]);
            orc_goertzel_init(&g[i], freqs[i], block_len);
        }
        decisions[b] = (kind == 1)  ?  orc_v18_tone_decide(e, n_freqs, total, threshold)  :  orc_ademco_tone_decide(e[0], e[1], total);
    }
}
