/*
 * shim_tone.c -- host side (plain C) of the spandsp-named tone-detector entry points
 * declared in include/spangpu_spandsp.h.  No signal processing happens here: frames are
 * staged into a bank (include/spangpu.h), the HIP kernels produce one record per
 * completed detection block, and this file replays those records through the caller's
 * callbacks / digit buffers in exactly the order and with exactly the arguments the
 * reference would have used:
 *   DTMF delivery      src/dtmf.c:304-358        Bell MF delivery  src/bell_r2_mf.c:636-672
 *   R2 MF reports      src/bell_r2_mf.c:869-876  super-tone cadence matcher src/super_tone_rx.c:164-228,364-448
 * Without a GPU every init fails (returns NULL): there is no CPU implementation.
 */
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "spangpu_spandsp.h"
#include "spangpu_refstate.h"

#define SUPER_TONE_BINS     128             /* src/spandsp/private/super_tone_rx.h:29 */

/* ------------------------------------------------------------------------------------ */
/* Channel groups                                                                       */
/* ------------------------------------------------------------------------------------ */
struct spangpu_group_s
{
    spangpu_bank_t *bank;
    int kind;
    int n_ch;
    int max_samples;
    int16_t *stage;             /* [n_ch][max_samples] */
    void **handles;             /* per channel: the attached state object or NULL */
    int32_t *lens;              /* per channel: samples staged for the tick being collected (0 = none) */
    int32_t *run;               /* ... and of the tick whose callbacks are being delivered */
    int delivering;             /* a tick's callbacks are being made: staging from inside them waits for the next flush */
    int n_attached;
    int n_staged;
    pthread_mutex_t lock;       /* staging, attach / detach and the tick itself (recursive: callbacks may call back in) */
    spangpu_block_t *blocks;
    int blocks_cap;
    spangpu_tone_params_t params;
};

struct dtmf_rx_state_s
{
    spangpu_group_t *grp;
    int channel;
    int private_grp;
    digits_rx_callback_t digits_callback;
    void *digits_callback_data;
    span_tone_report_func_t realtime_callback;
    void *realtime_callback_data;
    int last_hit;
    int in_digit;
    int lost_digits;
    int current_digits;
    char digits[MAX_DTMF_DIGITS + 1];
    /* bank parameters of a private object (applied when its bank is (re)built) */
    spangpu_tone_params_t params;
    int dirty;
    logging_state_t logging;
};

struct bell_mf_rx_state_s
{
    spangpu_group_t *grp;
    int channel;
    int private_grp;
    digits_rx_callback_t digits_callback;
    void *digits_callback_data;
    int lost_digits;
    int current_digits;
    char digits[MAX_BELL_MF_DIGITS + 1];
};

struct r2_mf_rx_state_s
{
    spangpu_group_t *grp;
    int channel;
    int private_grp;
    span_tone_report_func_t callback;
    void *callback_data;
    int fwd;
    int current_digit;
};

/* Super tone.  A descriptor is two things: a BOOK of every frequency the caller has named, each with the answer
   super_tone_rx_add_element() gives for it (the bin that monitors it), plus the coefficient of every bin; and a list of
   TONES, each a cadence of elements (a bin pair and a duration window).  A detector keeps the last ten confirmed runs
   of a (k1, k2) bin pair in a ring, newest at `head`, plus the pair seen in the last block while it is still unconfirmed. */
typedef struct
{
    int hz;                 /* the frequency as named */
    int answer;             /* what naming it again returns */
} st_name_t;

typedef struct
{
    int f1;                 /* bin pair, -1 = none */
    int f2;
    long long lo;           /* duration window, in samples */
    long long hi;
} st_elem_t;

typedef struct
{
    st_elem_t *elem;
    int n;
    int cap;
} st_tone_t;

struct super_tone_rx_descriptor_s
{
    st_name_t *book;
    int n_names;
    int cap_names;
    float fac[SPANGPU_MAX_BINS];
    int n_bins;             /* may run past SPANGPU_MAX_BINS: such a descriptor cannot be given to a detector */
    st_tone_t *tone;
    int n_tones;
    int cap_tones;
    int owned;
};

#define ST_HISTORY          10

typedef struct
{
    int f1;
    int f2;
    long long blocks;       /* length so far, in 128-sample blocks */
} st_run_t;

struct super_tone_rx_state_s
{
    spangpu_group_t *grp;
    int channel;
    int private_grp;
    super_tone_rx_descriptor_t *desc;
    int tone;               /* cadence being followed, -1 = none */
    int turn;               /* elements of it completed since it was recognised */
    span_tone_report_func_t tone_callback;
    tone_segment_func_t segment_callback;
    void *callback_data;
    st_run_t run[ST_HISTORY];
    int head;               /* index of the current run */
    int seen_f1;            /* the pair of the last block */
    int seen_f2;
};

struct goertzel_state_s
{
    spangpu_bank_t *bank;
    float fac;
    int samples;
    int current_sample;
    int has_pending;
    float pending;
    int owned;
};

static void replay(spangpu_group_t *g, int channel, const spangpu_block_t *b, int n);
static void end_of_call(spangpu_group_t *g, int channel);

spangpu_group_t *spangpu_group_create(int device, int kind, int n_channels, int max_samples,
                                      const spangpu_tone_params_t *params)
{
    spangpu_group_t *g;

    if (n_channels <= 0  ||  max_samples <= 0)
        return NULL;
    if ((g = (spangpu_group_t *) calloc(1, sizeof(*g))) == NULL)
        return NULL;
    g->kind = kind;
    g->n_ch = n_channels;
    g->max_samples = max_samples;
    if (params)
        g->params = *params;
    if (spangpu_bank_create(&g->bank, device, kind, n_channels, &g->params, sizeof(g->params)) != SPANGPU_OK)
    {
        free(g);
        return NULL;
    }
    g->stage = (int16_t *) calloc((size_t) n_channels*max_samples, sizeof(int16_t));
    g->handles = (void **) calloc(n_channels, sizeof(void *));
    g->lens = (int32_t *) calloc(n_channels, sizeof(int32_t));
    g->run = (int32_t *) calloc(n_channels, sizeof(int32_t));
    {
        pthread_mutexattr_t at;

        pthread_mutexattr_init(&at);
        pthread_mutexattr_settype(&at, PTHREAD_MUTEX_RECURSIVE);
        pthread_mutex_init(&g->lock, &at);
        pthread_mutexattr_destroy(&at);
    }
    if (g->stage == NULL  ||  g->handles == NULL  ||  g->lens == NULL  ||  g->run == NULL)
    {
        spangpu_group_destroy(g);
        return NULL;
    }
    return g;
}

int spangpu_group_destroy(spangpu_group_t *g)
{
    if (g == NULL)
        return SPANGPU_OK;
    spangpu_bank_destroy(g->bank);
    free(g->stage);
    free(g->handles);
    free(g->lens);
    free(g->run);
    free(g->blocks);
    pthread_mutex_destroy(&g->lock);
    free(g);
    return SPANGPU_OK;
}

spangpu_bank_t *spangpu_group_bank(spangpu_group_t *g)
{
    return (g)  ?  g->bank  :  NULL;
}

/* Run the tick with the channels that have staged a frame.  The others sit it out -- their detectors are exactly as they
   were, as the reference's are for a channel whose xxx_rx() was not called -- and may stage for the next one.  Returns
   the number of channels that took part. */
static int group_flush_locked_tick(spangpu_group_t *g)
{
    int rc;
    int n;
    int i;
    int start;
    int ch;

    if (g->n_staged == 0)
        return 0;
    rc = spangpu_bank_rx_var(g->bank, g->stage, SPANGPU_MEM_HOST, g->lens, g->max_samples, g->max_samples);
    n = (rc < 0)  ?  rc  :  spangpu_bank_blocks(g->bank, NULL, 0);
    if (n > g->blocks_cap)
    {
        free(g->blocks);
        g->blocks_cap = n + 64;
        if ((g->blocks = (spangpu_block_t *) malloc(sizeof(spangpu_block_t)*g->blocks_cap)) == NULL)
        {
            g->blocks_cap = 0;
            n = SPANGPU_ERR_NO_MEMORY;
        }
    }
    if (n > 0)
        n = spangpu_bank_blocks(g->bank, g->blocks, g->blocks_cap);
    /* The tick is over, whatever came of it: its frames leave the staging area before anything is delivered -- a failure
       must not make every later call a "second frame" or run the same frames again, and a callback that stages a new
       frame finds a clean slate (that frame waits for the next tick). */
    rc = g->n_staged;
    memcpy(g->run, g->lens, sizeof(int32_t)*g->n_ch);
    memset(g->lens, 0, sizeof(int32_t)*g->n_ch);
    g->n_staged = 0;
    if (n < 0)
        return n;
    /* Records arrive in (channel, block) order: replay channel by channel. */
    g->delivering = 1;
    start = 0;
    for (ch = 0;  ch < g->n_ch;  ch++)
    {
        i = start;
        while (i < n  &&  g->blocks[i].channel == ch)
            i++;
        if (g->handles[ch]  &&  g->run[ch] > 0)
        {
            replay(g, ch, &g->blocks[start], i - start);
            end_of_call(g, ch);
        }
        start = i;
    }
    g->delivering = 0;
    return rc;
}

/* The tick(s) that are due.  Callbacks may stage frames (a put_bit handler that answers by feeding its receiver, say): while
   a tick's callbacks run, a flush from inside them does nothing (`delivering`); when they are over, the tick those frames
   complete -- every attached channel has staged again -- runs at once instead of waiting for somebody to ask, so that no
   later xxx_rx() is refused as a second frame of a tick that nobody would ever have run. */
static int group_flush_locked(spangpu_group_t *g)
{
    int total = 0;
    int rc;

    if (g->delivering)
        return 0;
    for (;;)
    {
        if ((rc = group_flush_locked_tick(g)) < 0)
            return rc;
        total += rc;
        if (g->n_staged == 0  ||  g->n_staged < g->n_attached)
            break;
    }
    return total;
}

int spangpu_group_flush(spangpu_group_t *g)
{
    int rc;

    if (g == NULL)
        return SPANGPU_ERR_BAD_ARG;
    pthread_mutex_lock(&g->lock);
    rc = group_flush_locked(g);
    pthread_mutex_unlock(&g->lock);
    return rc;
}

/* Stage one channel's frame (any thread); the tick runs when every attached channel has staged, or when the owner of
   the tick calls spangpu_group_flush() at its deadline.  The frame is copied outside the lock: a channel has one
   submitter, as a spandsp object has. */
static int group_stage(spangpu_group_t *g, int channel, const int16_t amp[], int samples)
{
    int rc;

    if (samples <= 0)
        return 0;
    if (samples > g->max_samples)
        return SPANGPU_ERR_BAD_ARG;
    pthread_mutex_lock(&g->lock);
    if (g->lens[channel])
    {
        pthread_mutex_unlock(&g->lock);
        return SPANGPU_ERR_STATE;       /* second frame before the tick ran */
    }
    pthread_mutex_unlock(&g->lock);
    memcpy(g->stage + (size_t) channel*g->max_samples, amp, sizeof(int16_t)*samples);
    pthread_mutex_lock(&g->lock);
    g->lens[channel] = samples;
    g->n_staged++;
    rc = (g->n_staged >= g->n_attached)  ?  group_flush_locked(g)  :  0;
    pthread_mutex_unlock(&g->lock);
    return rc;
}

static int group_attach(spangpu_group_t *g, int channel, void *handle)
{
    if (g == NULL  ||  channel < 0  ||  channel >= g->n_ch)
        return -1;
    pthread_mutex_lock(&g->lock);
    if (g->handles[channel])
    {
        pthread_mutex_unlock(&g->lock);
        return -1;
    }
    g->handles[channel] = handle;
    g->n_attached++;
    spangpu_bank_reset_channel(g->bank, channel, 0);
    pthread_mutex_unlock(&g->lock);
    return 0;
}

static void group_detach(spangpu_group_t *g, int channel)
{
    if (g == NULL)
        return;
    pthread_mutex_lock(&g->lock);
    if (g->handles[channel])
    {
        g->handles[channel] = NULL;
        g->n_attached--;
        if (g->lens[channel])
        {
            g->lens[channel] = 0;
            g->n_staged--;
        }
        /* the channels that remain may all have been waiting for this one */
        if (g->n_staged > 0  &&  g->n_staged >= g->n_attached)
            group_flush_locked(g);
    }
    pthread_mutex_unlock(&g->lock);
}

/* ------------------------------------------------------------------------------------ */
/* DTMF                                                                                 */
/* ------------------------------------------------------------------------------------ */
static void dtmf_replay(dtmf_rx_state_t *s, const spangpu_block_t *b, int n)
{
    int i;
    int level;

    for (i = 0;  i < n;  i++)
    {
        if (b[i].flags & SPANGPU_BLK_CHANGE)
        {
            if (s->realtime_callback)
            {
                /* dtmf.c:309-316 */
                if (b[i].flags & SPANGPU_BLK_REPORT)
                {
                    if (b[i].flags & SPANGPU_BLK_TONE_OFF)
                        level = -99;
                    else
                        level = (int) (long int) (10.0f*log10f(b[i].energy) - 107.255f);  /* dtmf.c:110,314 */
                    s->realtime_callback(s->realtime_callback_data, b[i].code, level, b[i].duration);
                }
            }
            else if (b[i].code)
            {
                /* dtmf.c:320-340 */
                if (s->current_digits < MAX_DTMF_DIGITS)
                {
                    s->digits[s->current_digits++] = (char) b[i].code;
                    s->digits[s->current_digits] = '\0';
                    if (s->digits_callback)
                    {
                        s->digits_callback(s->digits_callback_data, s->digits, s->current_digits);
                        s->current_digits = 0;
                    }
                }
                else
                {
                    s->lost_digits++;
                }
            }
            s->in_digit = b[i].code;
            s->last_hit = b[i].code;
        }
        else
        {
            s->last_hit = b[i].hit;
        }
    }
}

static void dtmf_end_of_call(dtmf_rx_state_t *s)
{
    /* dtmf.c:352-358 */
    if (s->current_digits  &&  s->digits_callback)
    {
        s->digits_callback(s->digits_callback_data, s->digits, s->current_digits);
        s->digits[0] = '\0';
        s->current_digits = 0;
    }
}

/* A private object owns a one-channel group; its bank is rebuilt when a parameter that
   is compiled into the bank (report mode, dial-tone filter, thresholds) changes. */
static int dtmf_private_rebuild(dtmf_rx_state_t *s, int max_samples)
{
    float f[64];
    int32_t w[4];
    int have_state = 0;
    int nf = 0;

    if (s->grp)
    {
        if (!s->dirty  &&  max_samples <= s->grp->max_samples)
            return 0;
        nf = spangpu_bank_get_state(s->grp->bank, 0, f, 64, w, 4);
        have_state = (nf > 0);
        if (max_samples < s->grp->max_samples)
            max_samples = s->grp->max_samples;
        spangpu_group_destroy(s->grp);
        s->grp = NULL;
    }
    s->params.report_mode = (s->realtime_callback)  ?  SPANGPU_REPORT_REALTIME  :  SPANGPU_REPORT_DIGITS;
    if ((s->grp = spangpu_group_create(0, SPANGPU_DTMF, 1, (max_samples < 160)  ?  160  :  max_samples, &s->params)) == NULL)
        return -1;
    s->grp->handles[0] = s;
    s->grp->n_attached = 1;
    if (have_state)
        spangpu_bank_set_state(s->grp->bank, 0, f, nf, w, 4);
    s->dirty = 0;
    return 0;
}

/* What span_log_init(&s->logging, SPAN_LOG_NONE, NULL) + span_log_set_protocol() leave behind (dtmf.c:468-469,
   logging.c:264-281): logging off, 8 kHz time base, no tag.  The message handler stays NULL: the engine writes no text. */
static void st_logging_init(logging_state_t *lg, const char *protocol)
{
    memset(lg, 0, sizeof(*lg));
    lg->samples_per_second = 8000;
    lg->protocol = protocol;
}

dtmf_rx_state_t *dtmf_rx_init(dtmf_rx_state_t *s, digits_rx_callback_t callback, void *user_data)
{
    int fresh = (s == NULL);

    if (spangpu_device_count() <= 0)
        return NULL;
    if (fresh)
    {
        if ((s = (dtmf_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
            return NULL;
        s->private_grp = 1;
    }
    else if (s->private_grp)
    {
        /* re-initialise in place (dtmf.c:460-467) */
        spangpu_group_destroy(s->grp);
        memset(s, 0, sizeof(*s));
        s->private_grp = 1;
    }
    else
    {
        spangpu_group_t *g = s->grp;
        int ch = s->channel;

        memset(s, 0, sizeof(*s));
        s->grp = g;
        s->channel = ch;
        spangpu_bank_reset_channel(g->bank, ch, 0);
    }
    s->digits_callback = callback;
    s->digits_callback_data = user_data;
    s->dirty = 1;
    st_logging_init(&s->logging, "DTMF");
    if (s->private_grp  &&  dtmf_private_rebuild(s, 160) < 0)
    {
        if (fresh)
            free(s);
        return NULL;
    }
    return s;
}

dtmf_rx_state_t *spangpu_dtmf_rx_attach(spangpu_group_t *g, int channel, digits_rx_callback_t callback, void *user_data)
{
    dtmf_rx_state_t *s;

    if (g == NULL  ||  g->kind != SPANGPU_DTMF)
        return NULL;
    if ((s = (dtmf_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
        return NULL;
    if (group_attach(g, channel, s) < 0)
    {
        free(s);
        return NULL;
    }
    s->grp = g;
    s->channel = channel;
    s->digits_callback = callback;
    s->digits_callback_data = user_data;
    st_logging_init(&s->logging, "DTMF");
    return s;
}

logging_state_t *dtmf_rx_get_logging_state(dtmf_rx_state_t *s)
{
    return &s->logging;
}

int dtmf_rx_release(dtmf_rx_state_t *s)
{
    (void) s;
    return 0;
}

int dtmf_rx_free(dtmf_rx_state_t *s)
{
    if (s == NULL)
        return 0;
    if (s->private_grp)
        spangpu_group_destroy(s->grp);
    else
        group_detach(s->grp, s->channel);
    free(s);
    return 0;
}

void dtmf_rx_set_realtime_callback(dtmf_rx_state_t *s, span_tone_report_func_t callback, void *user_data)
{
    float f[64];
    int32_t w[4];
    int nf;

    s->realtime_callback = callback;
    s->realtime_callback_data = user_data;
    /* dtmf.c:415: the duration restarts */
    if (s->grp  &&  (nf = spangpu_bank_get_state(s->grp->bank, s->channel, f, 64, w, 4)) > 0)
    {
        w[3] = 0;
        spangpu_bank_set_state(s->grp->bank, s->channel, f, nf, w, 4);
    }
    if (s->private_grp)
    {
        s->dirty = 1;
        dtmf_private_rebuild(s, 160);
    }
    /* On a shared group the report mode is a property of the bank (spangpu_tone_params_t). */
}

void dtmf_rx_parms(dtmf_rx_state_t *s, int filter_dialtone, float twist, float reverse_twist, float threshold)
{
    float f[64];
    int32_t w[4];
    int nf;

    if (!s->private_grp)
    {
        /* one channel of a shared bank: the bank keeps thresholds, twists and the filter switch per channel */
        spangpu_tone_params_t p;

        memset(&p, 0, sizeof(p));
        p.filter_dialtone = (filter_dialtone >= 0)  ?  (filter_dialtone != 0)  :  -1;
        p.twist_db = twist;
        p.reverse_twist_db = reverse_twist;
        p.threshold_dbm0 = threshold;
        p.set_mask = SPANGPU_TP_TWIST | SPANGPU_TP_REVERSE_TWIST | SPANGPU_TP_THRESHOLD;    /* the bank applies dtmf.c:436-444's tests */
        pthread_mutex_lock(&s->grp->lock);
        spangpu_bank_set_channel_params(s->grp->bank, s->channel, &p, sizeof(p));
        pthread_mutex_unlock(&s->grp->lock);
        return;
    }
    if (filter_dialtone >= 0)
    {
        /* dtmf.c:428-434: the notch states restart */
        if (s->grp  &&  (nf = spangpu_bank_get_state(s->grp->bank, 0, f, 64, w, 4)) >= 21)
        {
            f[17] = f[18] = f[19] = f[20] = 0.0f;
            spangpu_bank_set_state(s->grp->bank, 0, f, nf, w, 4);
        }
        s->params.filter_dialtone = filter_dialtone;
    }
    if (twist >= 0.0f)
    {
        s->params.twist_db = twist;
        s->params.set_mask |= SPANGPU_TP_TWIST;
    }
    if (reverse_twist >= 0.0f)
    {
        s->params.reverse_twist_db = reverse_twist;
        s->params.set_mask |= SPANGPU_TP_REVERSE_TWIST;
    }
    if (threshold > -99.0f)
    {
        s->params.threshold_dbm0 = threshold;
        s->params.set_mask |= SPANGPU_TP_THRESHOLD;
    }
    s->dirty = 1;
    dtmf_private_rebuild(s, 160);
}

int dtmf_rx(dtmf_rx_state_t *s, const int16_t amp[], int samples)
{
    int pos;
    int n;

    if (s == NULL  ||  s->grp == NULL)
        return -1;
    if (!s->private_grp)
        return (group_stage(s->grp, s->channel, amp, samples) < 0)  ?  -1  :  0;
    /* private object: run the frame now, in pieces no longer than the staging row */
    for (pos = 0;  pos < samples;  pos += n)
    {
        n = samples - pos;
        if (n > s->grp->max_samples)
            n = s->grp->max_samples;
        if (group_stage(s->grp, 0, amp + pos, n) < 0)
            return -1;
    }
    return 0;
}

int dtmf_rx_fillin(dtmf_rx_state_t *s, int samples)
{
    (void) samples;
    if (s  &&  s->grp)
        spangpu_bank_reset_channel(s->grp->bank, s->channel, 1);
    return 0;
}

int dtmf_rx_status(dtmf_rx_state_t *s)
{
    if (s->in_digit)
        return s->in_digit;
    if (s->last_hit)
        return 'x';
    return 0;
}

size_t dtmf_rx_get(dtmf_rx_state_t *s, char *buf, int max)
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

/* ------------------------------------------------------------------------------------ */
/* Bell MF                                                                              */
/* ------------------------------------------------------------------------------------ */
static void bell_replay(bell_mf_rx_state_t *s, const spangpu_block_t *b, int n)
{
    int i;

    for (i = 0;  i < n;  i++)
    {
        if (!(b[i].flags & SPANGPU_BLK_REPORT))
            continue;
        /* bell_r2_mf.c:636-655 */
        if (s->current_digits < MAX_BELL_MF_DIGITS)
        {
            s->digits[s->current_digits++] = (char) b[i].code;
            s->digits[s->current_digits] = '\0';
            if (s->digits_callback)
            {
                s->digits_callback(s->digits_callback_data, s->digits, s->current_digits);
                s->current_digits = 0;
            }
        }
        else
        {
            s->lost_digits++;
        }
    }
}

static void bell_end_of_call(bell_mf_rx_state_t *s)
{
    /* bell_r2_mf.c:665-671 */
    if (s->current_digits  &&  s->digits_callback)
    {
        s->digits_callback(s->digits_callback_data, s->digits, s->current_digits);
        s->digits[0] = '\0';
        s->current_digits = 0;
    }
}

bell_mf_rx_state_t *bell_mf_rx_init(bell_mf_rx_state_t *s, digits_rx_callback_t callback, void *user_data)
{
    if (spangpu_device_count() <= 0)
        return NULL;
    if (s == NULL)
    {
        if ((s = (bell_mf_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
            return NULL;
        s->private_grp = 1;
        if ((s->grp = spangpu_group_create(0, SPANGPU_BELL_MF, 1, 160, NULL)) == NULL)
        {
            free(s);
            return NULL;
        }
        s->grp->handles[0] = s;
        s->grp->n_attached = 1;
    }
    else
    {
        spangpu_bank_reset_channel(s->grp->bank, s->channel, 0);
        s->lost_digits = 0;
        s->current_digits = 0;
        s->digits[0] = '\0';
    }
    s->digits_callback = callback;
    s->digits_callback_data = user_data;
    return s;
}

bell_mf_rx_state_t *spangpu_bell_mf_rx_attach(spangpu_group_t *g, int channel, digits_rx_callback_t callback, void *user_data)
{
    bell_mf_rx_state_t *s;

    if (g == NULL  ||  g->kind != SPANGPU_BELL_MF)
        return NULL;
    if ((s = (bell_mf_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
        return NULL;
    if (group_attach(g, channel, s) < 0)
    {
        free(s);
        return NULL;
    }
    s->grp = g;
    s->channel = channel;
    s->digits_callback = callback;
    s->digits_callback_data = user_data;
    return s;
}

int bell_mf_rx_release(bell_mf_rx_state_t *s)
{
    (void) s;
    return 0;
}

int bell_mf_rx_free(bell_mf_rx_state_t *s)
{
    if (s == NULL)
        return 0;
    if (s->private_grp)
        spangpu_group_destroy(s->grp);
    else
        group_detach(s->grp, s->channel);
    free(s);
    return 0;
}

static int stage_any(spangpu_group_t *g, int private_grp, int channel, const int16_t amp[], int samples)
{
    int pos;
    int n;

    if (g == NULL)
        return -1;
    if (!private_grp)
        return (group_stage(g, channel, amp, samples) < 0)  ?  -1  :  0;
    for (pos = 0;  pos < samples;  pos += n)
    {
        n = samples - pos;
        if (n > g->max_samples)
            n = g->max_samples;
        if (group_stage(g, 0, amp + pos, n) < 0)
            return -1;
    }
    return 0;
}

int bell_mf_rx(bell_mf_rx_state_t *s, const int16_t amp[], int samples)
{
    return stage_any(s->grp, s->private_grp, s->channel, amp, samples);
}

size_t bell_mf_rx_get(bell_mf_rx_state_t *s, char *buf, int max)
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

/* ------------------------------------------------------------------------------------ */
/* R2 MF                                                                                */
/* ------------------------------------------------------------------------------------ */
static void r2_replay(r2_mf_rx_state_t *s, const spangpu_block_t *b, int n)
{
    int i;

    for (i = 0;  i < n;  i++)
    {
        /* bell_r2_mf.c:869-876 */
        if ((b[i].flags & SPANGPU_BLK_REPORT)  &&  s->callback)
            s->callback(s->callback_data, b[i].code, (b[i].code)  ?  -10  :  -99, 0);
        s->current_digit = b[i].code;
    }
}

r2_mf_rx_state_t *r2_mf_rx_init(r2_mf_rx_state_t *s, bool fwd, span_tone_report_func_t callback, void *user_data)
{
    spangpu_tone_params_t p;

    if (spangpu_device_count() <= 0)
        return NULL;
    if (s == NULL)
    {
        if ((s = (r2_mf_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
            return NULL;
        s->private_grp = 1;
        memset(&p, 0, sizeof(p));
        p.r2_fwd = fwd;
        if ((s->grp = spangpu_group_create(0, SPANGPU_R2_MF, 1, 160, &p)) == NULL)
        {
            free(s);
            return NULL;
        }
        s->grp->handles[0] = s;
        s->grp->n_attached = 1;
    }
    else
    {
        spangpu_bank_reset_channel(s->grp->bank, s->channel, 0);
    }
    s->fwd = fwd;
    s->callback = callback;
    s->callback_data = user_data;
    s->current_digit = 0;
    return s;
}

r2_mf_rx_state_t *spangpu_r2_mf_rx_attach(spangpu_group_t *g, int channel, span_tone_report_func_t callback, void *user_data)
{
    r2_mf_rx_state_t *s;

    if (g == NULL  ||  g->kind != SPANGPU_R2_MF)
        return NULL;
    if ((s = (r2_mf_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
        return NULL;
    if (group_attach(g, channel, s) < 0)
    {
        free(s);
        return NULL;
    }
    s->grp = g;
    s->channel = channel;
    s->fwd = g->params.r2_fwd;
    s->callback = callback;
    s->callback_data = user_data;
    return s;
}

int r2_mf_rx_release(r2_mf_rx_state_t *s)
{
    (void) s;
    return 0;
}

int r2_mf_rx_free(r2_mf_rx_state_t *s)
{
    if (s == NULL)
        return 0;
    if (s->private_grp)
        spangpu_group_destroy(s->grp);
    else
        group_detach(s->grp, s->channel);
    free(s);
    return 0;
}

int r2_mf_rx(r2_mf_rx_state_t *s, const int16_t amp[], int samples)
{
    return stage_any(s->grp, s->private_grp, s->channel, amp, samples);
}

int r2_mf_rx_get(r2_mf_rx_state_t *s)
{
    return s->current_digit;
}

/* ------------------------------------------------------------------------------------ */
/* Super tone: descriptor building and cadence matching on the host                      */
/* ------------------------------------------------------------------------------------ */
super_tone_rx_descriptor_t *super_tone_rx_make_descriptor(super_tone_rx_descriptor_t *desc)
{
    const int owned = (desc == NULL);

    if (owned  &&  (desc = (super_tone_rx_descriptor_t *) malloc(sizeof(*desc))) == NULL)
        return NULL;
    memset(desc, 0, sizeof(*desc));
    desc->owned = owned;
    return desc;
}

int super_tone_rx_free_descriptor(super_tone_rx_descriptor_t *desc)
{
    int t;

    if (desc == NULL)
        return 0;
    for (t = 0;  t < desc->n_tones;  t++)
        free(desc->tone[t].elem);
    free(desc->tone);
    free(desc->book);
    if (desc->owned)
        free(desc);
    return 0;
}

/* Room for one more item in a growing array (doubling). */
static int st_room(void **arr, int *cap, int n, size_t item)
{
    void *p;
    int want;

    if (n < *cap)
        return 0;
    want = (*cap)  ?  2*(*cap)  :  8;
    if ((p = realloc(*arr, (size_t) want*item)) == NULL)
        return -1;
    *arr = p;
    *cap = want;
    return 0;
}

static void st_tune_bin(super_tone_rx_descriptor_t *desc, int bin, float hz)
{
    if (bin >= 0  &&  bin < SPANGPU_MAX_BINS)
        desc->fac[bin] = spangpu_goertzel_fac(hz);
}

/* The bin that monitors `hz`, entering it in the book if it is new.  Observable behaviour of the reference's
   resolver (super_tone_rx.c:81-123), which a caller's element numbering depends on:
     - a frequency named before gets the answer recorded for it;
     - one within 10 Hz of an earlier NAME (searched in naming order) shares that name's answer as its bin, the bin is
       re-tuned to the mean of the two, and the new name is recorded with the POSITION of the earlier name in the book as
       its answer -- not the bin; naming it a second time returns that position (a quirk of the reference that is kept);
     - anything else opens a new bin.
   0 Hz means "no tone" (-1). */
static int st_bin_for(super_tone_rx_descriptor_t *desc, int hz)
{
    int i;
    int near = -1;
    st_name_t *e;

    if (hz == 0)
        return -1;
    for (i = 0;  i < desc->n_names;  i++)
    {
        if (desc->book[i].hz == hz)
            return desc->book[i].answer;
        if (near < 0  &&  abs(desc->book[i].hz - hz) <= 10)
            near = i;
    }
    if (st_room((void **) &desc->book, &desc->cap_names, desc->n_names, sizeof(st_name_t)) < 0)
        return -1;
    e = &desc->book[desc->n_names++];
    e->hz = hz;
    if (near >= 0)
    {
        const int bin = desc->book[near].answer;

        e->answer = near;
        st_tune_bin(desc, bin, (float) (hz + desc->book[near].hz)/2);
        return bin;
    }
    e->answer = desc->n_bins;
    st_tune_bin(desc, desc->n_bins, (float) hz);
    return desc->n_bins++;
}

int super_tone_rx_add_tone(super_tone_rx_descriptor_t *desc)
{
    if (st_room((void **) &desc->tone, &desc->cap_tones, desc->n_tones, sizeof(st_tone_t)) < 0)
        return -1;
    memset(&desc->tone[desc->n_tones], 0, sizeof(st_tone_t));
    return desc->n_tones++;
}

/* min / max in milliseconds (max 0 = no upper limit), kept in samples like the reference (super_tone_rx.c:157-158). */
int super_tone_rx_add_element(super_tone_rx_descriptor_t *desc, int tone, int f1, int f2, int min, int max)
{
    st_tone_t *t;
    st_elem_t *e;

    if (tone < 0  ||  tone >= desc->n_tones)
        return -1;
    t = &desc->tone[tone];
    if (st_room((void **) &t->elem, &t->cap, t->n, sizeof(st_elem_t)) < 0)
        return -1;
    e = &t->elem[t->n];
    e->f1 = st_bin_for(desc, f1);           /* in this order: the bins are numbered as they are first named */
    e->f2 = st_bin_for(desc, f2);
    e->lo = 8LL*min;
    e->hi = (max == 0)  ?  0x7FFFFFFFLL  :  8LL*max;
    return t->n++;
}

/* ---- cadence matching (the decisions of super_tone_rx.c:164-228 and :364-448 on the run history) ---- */

/* The run `back` places before the current one (0 = current). */
static const st_run_t *st_past(const super_tone_rx_state_t *s, int back)
{
    return &s->run[(s->head + ST_HISTORY - back)%ST_HISTORY];
}

static int st_same_pair(const st_elem_t *e, const st_run_t *r)
{
    return e->f1 == r->f1  &&  e->f2 == r->f2;
}

static long long st_samples(const st_run_t *r)
{
    return r->blocks*SUPER_TONE_BINS;
}

/* A finished run fits an element when the pair is right and the length lies in the window. */
static int st_fits(const st_elem_t *e, const st_run_t *r)
{
    return st_same_pair(e, r)  &&  e->lo <= st_samples(r)  &&  st_samples(r) <= e->hi;
}

/* Does the newest history spell out the whole cadence, the current run being its last element? */
static int st_spells(const super_tone_rx_state_t *s, const st_tone_t *t)
{
    int i;

    if (t->n > ST_HISTORY)
        return 0;
    for (i = 0;  i < t->n;  i++)
    {
        if (!st_fits(&t->elem[i], st_past(s, t->n - 1 - i)))
            return 0;
    }
    return 1;
}

/* Is the cadence being followed still alive?  `turn` elements of it have gone by since it was recognised (it was
   recognised on its last element), so the current run must be element (turn - 1) mod n and not yet too long; when a run
   has just ended (`run_ended`) the one before it must in addition have been a proper element (turn - 2) mod n. */
static int st_alive(const super_tone_rx_state_t *s, const st_tone_t *t, int turn, int run_ended)
{
    const st_elem_t *e;

    if (t->n <= 0)
        return 0;
    if (run_ended  &&  !st_fits(&t->elem[(turn + t->n - 2)%t->n], st_past(s, 1)))
        return 0;
    e = &t->elem[(turn + t->n - 1)%t->n];
    return st_same_pair(e, st_past(s, 0))  &&  st_samples(st_past(s, 0)) <= e->hi;
}

static void st_lose_tone(super_tone_rx_state_t *s)
{
    s->tone = -1;
    s->tone_callback(s->callback_data, -1, -10, 0);
}

/* One 128-sample block, decided on the device as the bin pair (k1, k2). */
static void st_block(super_tone_rx_state_t *s, int k1, int k2)
{
    const super_tone_rx_descriptor_t *d = s->desc;
    st_run_t *cur = &s->run[s->head];
    const int repeat = (k1 == s->seen_f1  &&  k2 == s->seen_f2);
    int t;

    s->seen_f1 = k1;
    s->seen_f2 = k2;
    if (!repeat)
    {
        /* a pair seen for the first time may be a glitch: it still counts towards the current run */
        cur->blocks++;
    }
    else if (k1 != cur->f1  ||  k2 != cur->f2)
    {
        /* seen twice in a row, and not what the current run is made of: that run is over */
        if (s->tone >= 0)
        {
            const int turn = s->turn++;

            if (!st_alive(s, &d->tone[s->tone], turn, 1))
                st_lose_tone(s);
        }
        if (s->segment_callback)
            s->segment_callback(s->callback_data, cur->f1, cur->f2, (int) (st_samples(cur)/8));
        s->head = (s->head + 1)%ST_HISTORY;
        cur = &s->run[s->head];
        cur->f1 = k1;
        cur->f2 = k2;
        cur->blocks = 1;
    }
    else
    {
        /* more of the same (tested before this block is counted, as the reference does) */
        if (s->tone >= 0  &&  !st_alive(s, &d->tone[s->tone], s->turn, 0))
            st_lose_tone(s);
        cur->blocks++;
    }
    if (s->tone >= 0)
        return;
    for (t = 0;  t < d->n_tones;  t++)
    {
        if (st_spells(s, &d->tone[t]))
        {
            s->tone = t;
            s->turn = 0;
            s->tone_callback(s->callback_data, t, -10, 0);
            break;
        }
    }
}

static void st_replay(super_tone_rx_state_t *s, const spangpu_block_t *b, int n)
{
    int i;

    for (i = 0;  i < n;  i++)
        st_block(s, b[i].hit, b[i].code);
}

static void st_reset(super_tone_rx_state_t *s, super_tone_rx_descriptor_t *desc, span_tone_report_func_t callback, void *user_data)
{
    int i;

    for (i = 0;  i < ST_HISTORY;  i++)
    {
        s->run[i].f1 = -1;
        s->run[i].f2 = -1;
        s->run[i].blocks = 0;
    }
    s->head = ST_HISTORY - 1;
    s->seen_f1 = -1;
    s->seen_f2 = -1;
    s->segment_callback = NULL;
    s->tone_callback = callback;
    s->callback_data = user_data;
    s->desc = desc;
    s->tone = -1;
    s->turn = 0;
}

static void st_params(const super_tone_rx_descriptor_t *desc, spangpu_tone_params_t *p)
{
    int i;

    memset(p, 0, sizeof(*p));
    p->n_bins = desc->n_bins;
    for (i = 0;  i < desc->n_bins  &&  i < SPANGPU_MAX_BINS;  i++)
        p->bin_fac[i] = desc->fac[i];
}

super_tone_rx_state_t *super_tone_rx_init(super_tone_rx_state_t *s, super_tone_rx_descriptor_t *desc,
                                          span_tone_report_func_t callback, void *user_data)
{
    spangpu_tone_params_t p;

    if (desc == NULL  ||  callback == NULL)
        return NULL;                                        /* super_tone_rx.c:514-519 */
    if (desc->n_bins < 2  ||  desc->n_bins > SPANGPU_MAX_BINS)
        return NULL;
    if (spangpu_device_count() <= 0)
        return NULL;
    if (s == NULL)
    {
        if ((s = (super_tone_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
            return NULL;
        s->private_grp = 1;
        st_params(desc, &p);
        if ((s->grp = spangpu_group_create(0, SPANGPU_SUPER_TONE, 1, 160, &p)) == NULL)
        {
            free(s);
            return NULL;
        }
        s->grp->handles[0] = s;
        s->grp->n_attached = 1;
    }
    else
    {
        spangpu_bank_reset_channel(s->grp->bank, s->channel, 0);
    }
    st_reset(s, desc, callback, user_data);
    return s;
}

super_tone_rx_state_t *spangpu_super_tone_rx_attach(spangpu_group_t *g, int channel, super_tone_rx_descriptor_t *desc,
                                                    span_tone_report_func_t callback, void *user_data)
{
    super_tone_rx_state_t *s;

    if (g == NULL  ||  g->kind != SPANGPU_SUPER_TONE  ||  desc == NULL  ||  callback == NULL)
        return NULL;
    if ((s = (super_tone_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
        return NULL;
    if (group_attach(g, channel, s) < 0)
    {
        free(s);
        return NULL;
    }
    s->grp = g;
    s->channel = channel;
    st_reset(s, desc, callback, user_data);
    return s;
}

/* Bank parameters for a group whose channels all use `desc` (one descriptor per bank). */
int spangpu_super_tone_params(const super_tone_rx_descriptor_t *desc, spangpu_tone_params_t *params)
{
    if (desc == NULL  ||  params == NULL  ||  desc->n_bins < 2  ||  desc->n_bins > SPANGPU_MAX_BINS)
        return SPANGPU_ERR_BAD_ARG;
    st_params(desc, params);
    return SPANGPU_OK;
}

/* Give a bank the descriptor's cadences, to be matched on the device (spangpu_bank_set_cadences): for callers that run a
   super-tone bank themselves and want tone reports instead of block records. */
int spangpu_super_tone_cadences(const super_tone_rx_descriptor_t *desc, spangpu_bank_t *bank, int want_segments)
{
    int32_t *counts;
    spangpu_cadence_elem_t *el;
    int t;
    int i;
    int n = 0;
    int rc;

    if (desc == NULL  ||  bank == NULL  ||  desc->n_bins < 2  ||  desc->n_bins > SPANGPU_MAX_BINS)
        return SPANGPU_ERR_BAD_ARG;
    for (t = 0;  t < desc->n_tones;  t++)
        n += desc->tone[t].n;
    counts = (int32_t *) malloc(sizeof(int32_t)*(size_t) (desc->n_tones + 1));
    el = (spangpu_cadence_elem_t *) malloc(sizeof(*el)*(size_t) (n + 1));
    if (counts == NULL  ||  el == NULL)
    {
        free(counts);
        free(el);
        return SPANGPU_ERR_NO_MEMORY;
    }
    n = 0;
    for (t = 0;  t < desc->n_tones;  t++)
    {
        counts[t] = desc->tone[t].n;
        for (i = 0;  i < desc->tone[t].n;  i++, n++)
        {
            const st_elem_t *e = &desc->tone[t].elem[i];

            el[n].f1 = e->f1;
            el[n].f2 = e->f2;
            el[n].min_ms = (int32_t) (e->lo/8);
            el[n].max_ms = (e->hi >= 0x7FFFFFFFLL)  ?  0  :  (int32_t) (e->hi/8);
        }
    }
    rc = spangpu_bank_set_cadences(bank, counts, desc->n_tones, el, want_segments);
    free(counts);
    free(el);
    return rc;
}

int super_tone_rx_release(super_tone_rx_state_t *s)
{
    (void) s;
    return 0;
}

int super_tone_rx_free(super_tone_rx_state_t *s)
{
    if (s == NULL)
        return 0;
    if (s->private_grp)
        spangpu_group_destroy(s->grp);
    else
        group_detach(s->grp, s->channel);
    free(s);
    return 0;
}

void super_tone_rx_tone_callback(super_tone_rx_state_t *s, span_tone_report_func_t callback, void *user_data)
{
    s->tone_callback = callback;
    s->callback_data = user_data;
}

void super_tone_rx_segment_callback(super_tone_rx_state_t *s, tone_segment_func_t callback)
{
    s->segment_callback = callback;
}

int super_tone_rx(super_tone_rx_state_t *s, const int16_t amp[], int samples)
{
    if (stage_any(s->grp, s->private_grp, s->channel, amp, samples) < 0)
        return -1;
    return samples;                                         /* super_tone_rx.c:489 */
}

/* ---- a super-tone receiver in the reference's struct layout (include/spangpu_refstate.h).  The reference keeps the
   current run in segments[9], the nine before it in [8] .. [0], and in [10] the pair seen in the last block
   (super_tone_rx.c:369-409); its Goertzels are driven by goertzel_update(), so their own counters are the block position. */
static int st_ref_bins(const spangpu_ref_super_tone_rx_t *ref)
{
    /* struct super_tone_rx_descriptor_s begins {int used_frequencies; int monitored_frequencies; ...} */
    return (ref->desc)  ?  ((const int *) ref->desc)[1]  :  -1;
}

int spangpu_super_tone_rx_import_state(super_tone_rx_state_t *s, const spangpu_ref_super_tone_rx_t *ref)
{
    float f[2*SPANGPU_MAX_BINS + 8];
    int32_t w[4] = {0, 0, 0, 0};
    int m;
    int nb;
    int nsf;
    int i;
    int rc;

    if (s == NULL  ||  ref == NULL  ||  s->desc == NULL)
        return SPANGPU_ERR_BAD_ARG;
    m = s->desc->n_bins;
    if (m < 2  ||  m > SPANGPU_MAX_BINS  ||  st_ref_bins(ref) != m)
        return SPANGPU_ERR_BAD_ARG;
    if ((rc = spangpu_group_flush(s->grp)) < 0)
        return rc;
    /* the bank keeps nb >= m bins per channel (the kernel's bin count): v2[nb], v3[nb], energy */
    if ((nsf = spangpu_bank_get_state(s->grp->bank, s->channel, f, 2*SPANGPU_MAX_BINS + 8, w, 4)) < 0)
        return nsf;
    nb = (nsf - 1)/2;
    if (nb < m)
        return SPANGPU_ERR_STATE;
    for (i = 0;  i < m;  i++)
    {
        f[i] = ref->state[i].v2;
        f[nb + i] = ref->state[i].v3;
    }
    f[2*nb] = ref->energy;
    w[0] = ref->state[0].current_sample;
    w[1] = w[2] = w[3] = 0;
    if ((rc = spangpu_bank_set_state(s->grp->bank, s->channel, f, nsf, w, 4)) != SPANGPU_OK)
        return rc;
    s->head = ST_HISTORY - 1;
    for (i = 0;  i < ST_HISTORY;  i++)
    {
        s->run[i].f1 = ref->segments[i].f1;
        s->run[i].f2 = ref->segments[i].f2;
        s->run[i].blocks = ref->segments[i].min_duration;
    }
    s->seen_f1 = ref->segments[10].f1;
    s->seen_f2 = ref->segments[10].f2;
    s->tone = ref->detected_tone;
    s->turn = ref->rotation;
    return SPANGPU_OK;
}

int spangpu_super_tone_rx_export_state(super_tone_rx_state_t *s, spangpu_ref_super_tone_rx_t *ref)
{
    float f[2*SPANGPU_MAX_BINS + 8];
    int32_t w[4];
    int m;
    int nb;
    int i;
    int rc;

    if (s == NULL  ||  ref == NULL  ||  s->desc == NULL)
        return SPANGPU_ERR_BAD_ARG;
    m = s->desc->n_bins;
    if (m < 2  ||  m > SPANGPU_MAX_BINS  ||  st_ref_bins(ref) != m)
        return SPANGPU_ERR_BAD_ARG;
    if ((rc = spangpu_group_flush(s->grp)) < 0)
        return rc;
    if ((rc = spangpu_bank_get_state(s->grp->bank, s->channel, f, 2*SPANGPU_MAX_BINS + 8, w, 4)) < 0)
        return rc;
    nb = (rc - 1)/2;
    if (nb < m)
        return SPANGPU_ERR_STATE;
    for (i = 0;  i < m;  i++)
    {
        ref->state[i].v2 = f[i];
        ref->state[i].v3 = f[nb + i];
        ref->state[i].fac = s->desc->fac[i];
        ref->state[i].samples = SUPER_TONE_BINS;
        ref->state[i].current_sample = w[0];
    }
    ref->energy = f[2*nb];
    for (i = 0;  i < ST_HISTORY;  i++)
    {
        const st_run_t *r = st_past(s, ST_HISTORY - 1 - i);

        ref->segments[i].f1 = r->f1;
        ref->segments[i].f2 = r->f2;
        ref->segments[i].min_duration = (int) r->blocks;
    }
    ref->segments[10].f1 = s->seen_f1;
    ref->segments[10].f2 = s->seen_f2;
    ref->detected_tone = s->tone;
    ref->rotation = s->turn;
    return SPANGPU_OK;
}

int super_tone_rx_fillin(super_tone_rx_state_t *s, int samples)
{
    (void) s;
    (void) samples;
    return 0;                                               /* super_tone_rx.c:493-497 */
}

/* ------------------------------------------------------------------------------------ */
/* Goertzel: one bin, one channel, on a generic bank                                     */
/* ------------------------------------------------------------------------------------ */
void make_goertzel_descriptor(goertzel_descriptor_t *t, float freq, int samples)
{
    t->fac = spangpu_goertzel_fac(freq);
    t->samples = samples;
}

goertzel_state_t *goertzel_init(goertzel_state_t *s, goertzel_descriptor_t *t)
{
    spangpu_tone_params_t p;
    int owned = 0;

    if (spangpu_device_count() <= 0  ||  t == NULL)
        return NULL;
    if (s == NULL)
    {
        if ((s = (goertzel_state_t *) calloc(1, sizeof(*s))) == NULL)
            return NULL;
        owned = 1;
    }
    else if (s->bank)
    {
        spangpu_bank_destroy(s->bank);
        owned = s->owned;
    }
    memset(s, 0, sizeof(*s));
    s->owned = owned;
    s->fac = t->fac;
    s->samples = t->samples;
    memset(&p, 0, sizeof(p));
    p.n_bins = 1;
    p.block_len = t->samples;
    p.bin_fac[0] = t->fac;
    if (spangpu_bank_create(&s->bank, 0, SPANGPU_GOERTZEL, 1, &p, sizeof(p)) != SPANGPU_OK)
    {
        if (owned)
            free(s);
        return NULL;
    }
    return s;
}

int goertzel_release(goertzel_state_t *s)
{
    (void) s;
    return 0;
}

int goertzel_free(goertzel_state_t *s)
{
    if (s)
    {
        spangpu_bank_destroy(s->bank);
        if (s->owned)
            free(s);
    }
    return 0;
}

void goertzel_reset(goertzel_state_t *s)
{
    spangpu_bank_reset_channel(s->bank, 0, 0);
    s->current_sample = 0;
    s->has_pending = 0;
}

/* tone_detect.c:123-156: consumes at most the remainder of the block */
int goertzel_update(goertzel_state_t *s, const int16_t amp[], int samples)
{
    float e[8];

    if (samples > s->samples - s->current_sample)
        samples = s->samples - s->current_sample;
    if (samples <= 0)
        return 0;
    if (spangpu_bank_rx(s->bank, amp, SPANGPU_MEM_HOST, SPANGPU_LAYOUT_CHANNEL_MAJOR, samples, samples) < 0)
        return 0;
    s->current_sample += samples;
    if (s->current_sample >= s->samples)
    {
        /* the block completed on the device: its energy is waiting for goertzel_result() */
        if (spangpu_bank_trace(s->bank, e, 8) >= 1)
        {
            s->pending = e[0];
            s->has_pending = 1;
        }
    }
    else
    {
        spangpu_bank_sync(s->bank);
    }
    return samples;
}

/* tone_detect.c:160-205: evaluate (pushing one zero sample) and reset */
float goertzel_result(goertzel_state_t *s)
{
    float e[8];
    float r = 0.0f;

    if (s->has_pending)
    {
        r = s->pending;
    }
    else if (spangpu_bank_force_block(s->bank) == SPANGPU_OK  &&  spangpu_bank_trace(s->bank, e, 8) >= 1)
    {
        r = e[0];
    }
    s->has_pending = 0;
    s->current_sample = 0;
    return r;
}

/* ------------------------------------------------------------------------------------ */
static void replay(spangpu_group_t *g, int channel, const spangpu_block_t *b, int n)
{
    void *h = g->handles[channel];

    switch (g->kind)
    {
    case SPANGPU_DTMF:
        dtmf_replay((dtmf_rx_state_t *) h, b, n);
        break;
    case SPANGPU_BELL_MF:
        bell_replay((bell_mf_rx_state_t *) h, b, n);
        break;
    case SPANGPU_R2_MF:
        r2_replay((r2_mf_rx_state_t *) h, b, n);
        break;
    case SPANGPU_SUPER_TONE:
        st_replay((super_tone_rx_state_t *) h, b, n);
        break;
    }
}

static void end_of_call(spangpu_group_t *g, int channel)
{
    void *h = g->handles[channel];

    switch (g->kind)
    {
    case SPANGPU_DTMF:
        dtmf_end_of_call((dtmf_rx_state_t *) h);
        break;
    case SPANGPU_BELL_MF:
        bell_end_of_call((bell_mf_rx_state_t *) h);
        break;
    }
}
