/*
 * shim_fsk.c -- host side (plain C) of the spandsp-named entry points for the FSK receiver, the modem connect
 * tone detector and the DTMF sender, declared in include/spangpu_spandsp.h: fsk_rx*, modem_connect_tones_rx*,
 * dtmf_tx*.  No signal processing happens here: samples go to (or come from) a bank of include/spangpu.h, the
 * HIP kernel leaves each channel's put_bit / tone report stream in order, and this file replays it through the
 * caller's callbacks as the reference would (src/fsk.c:343-391, src/modem_connect_tones.c:416-435).
 * Without a GPU every init returns NULL: there is no CPU implementation.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "spangpu_spandsp.h"

/* preset_fsk_specs[], src/fsk.c:60-155 */
const fsk_spec_t preset_fsk_specs[] =
{
    {"V21 ch 1", 1080 + 100, 1080 - 100, -14, -30, 300*100},
    {"V21 ch 2", 1750 + 100, 1750 - 100, -14, -30, 300*100},
    {"V23 ch 1", 1700 + 400, 1700 - 400, -14, -30, 1200*100},
    {"V23 ch 2", 420 + 30, 420 - 30, -14, -30, 75*100},
    {"Bell103 ch 1", 1170 - 100, 1170 + 100, -14, -30, 300*100},
    {"Bell103 ch 2", 2125 - 100, 2125 + 100, -14, -30, 300*100},
    {"Bell202", 1700 + 500, 1700 - 500, -14, -30, 1200*100},
    {"Weitbrecht 45.45", 1600 + 200, 1600 - 200, -14, -30, 4545},
    {"Weitbrecht 50", 1600 + 200, 1600 - 200, -14, -30, 50*100},
    {"Weitbrecht 47.6", 1600 + 200, 1600 - 200, -14, -30, 4760},
    {"V21 (110bps) ch 1", 1080 + 100, 1080 - 100, -14, -30, 110*100}
};

/* ---- one staging group for both receiver kinds ---------------------------------------------------------- */
struct spangpu_line_group_s
{
    int is_mct;
    spangpu_fsk_t *fsk;
    spangpu_mct_t *mct;
    spangpu_fsk_spec_t spec;
    int tone_type;
    int n_ch;
    int max_samples;
    int16_t *stage;
    void **handles;
    int32_t *lens;              /* per channel: samples staged for the tick being collected (0 = none) */
    int32_t *run;               /* ... and of the tick whose callbacks are being delivered */
    int delivering;             /* a tick's callbacks are being made: staging from inside them waits for the next flush */
    int n_attached;
    int n_staged;
    pthread_mutex_t lock;       /* staging, attach / detach and the tick itself (recursive: callbacks may call back in) */
};

static spangpu_line_group_t *group_new(int n_channels, int max_samples)
{
    spangpu_line_group_t *g;

    if (n_channels <= 0  ||  max_samples <= 0  ||  (g = (spangpu_line_group_t *) calloc(1, sizeof(*g))) == NULL)
        return NULL;
    g->n_ch = n_channels;
    g->max_samples = max_samples;
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
        spangpu_line_group_destroy(g);
        return NULL;
    }
    return g;
}

spangpu_line_group_t *spangpu_fsk_group_create(int device, const fsk_spec_t *spec, int framing_mode, int n_channels, int max_samples)
{
    spangpu_line_group_t *g;

    if (spec == NULL  ||  (g = group_new(n_channels, max_samples)) == NULL)
        return NULL;
    g->spec.freq_zero = spec->freq_zero;
    g->spec.freq_one = spec->freq_one;
    g->spec.tx_level = spec->tx_level;
    g->spec.min_level = spec->min_level;
    g->spec.baud_rate = spec->baud_rate;
    if (spangpu_fsk_create(&g->fsk, device, n_channels, &g->spec, framing_mode) != SPANGPU_OK)
    {
        spangpu_line_group_destroy(g);
        return NULL;
    }
    return g;
}

spangpu_line_group_t *spangpu_modem_connect_tones_group_create(int device, int tone_type, int use_callbacks, int n_channels,
                                                                int max_samples)
{
    spangpu_line_group_t *g;

    if ((g = group_new(n_channels, max_samples)) == NULL)
        return NULL;
    g->is_mct = 1;
    g->tone_type = tone_type;
    if (spangpu_mct_create(&g->mct, device, tone_type, n_channels, use_callbacks) != SPANGPU_OK)
    {
        spangpu_line_group_destroy(g);
        return NULL;
    }
    return g;
}

int spangpu_line_group_destroy(spangpu_line_group_t *g)
{
    if (g == NULL)
        return 0;
    if (g->fsk)
        spangpu_fsk_destroy(g->fsk);
    if (g->mct)
        spangpu_mct_destroy(g->mct);
    free(g->stage);
    free(g->handles);
    free(g->lens);
    free(g->run);
    pthread_mutex_destroy(&g->lock);
    free(g);
    return 0;
}

/* The tick is over, whatever came of it: its frames leave the staging area before anything is delivered -- a failure must
   not make every later call a "second frame" or run the same frames again, and a callback that stages a new frame finds
   a clean slate (that frame waits for the next tick).  Returns how many frames the tick had. */
static int tick_taken(spangpu_line_group_t *g)
{
    const int took = g->n_staged;

    memcpy(g->run, g->lens, sizeof(int32_t)*g->n_ch);
    memset(g->lens, 0, sizeof(int32_t)*g->n_ch);
    g->n_staged = 0;
    return took;
}

/* Run the tick with the receivers that have staged a frame; the others sit it out, untouched (as the reference's are when
   their xxx_rx() is not called), and may stage for the next one.  Returns how many took part. */
static int line_flush_locked_tick(spangpu_line_group_t *g)
{
    int cap;
    int c;
    int i;
    int n;
    int rc;

    int took;

    if (g->n_staged == 0)
        return 0;
    if (g->is_mct)
    {
        const int32_t *events;
        const int32_t *counts;

        rc = spangpu_mct_rx_var(g->mct, g->stage, SPANGPU_MEM_HOST, g->lens, g->max_samples, g->max_samples);
        cap = (rc < 0)  ?  rc  :  spangpu_mct_events(g->mct, &events, &counts);
        took = tick_taken(g);
        if (cap < 0)
            return cap;
        g->delivering = 1;
        for (c = 0;  c < g->n_ch;  c++)
        {
            modem_connect_tones_rx_state_t *s = (modem_connect_tones_rx_state_t *) g->handles[c];

            n = (counts[c] < cap)  ?  counts[c]  :  cap;
            if (s  &&  s->tone_callback  &&  g->run[c] > 0)
            {
                /* report_tone_state(), modem_connect_tones.c:420-423 */
                for (i = 0;  i < n;  i++)
                    s->tone_callback(s->callback_data, events[((size_t) c*cap + i)*2], events[((size_t) c*cap + i)*2 + 1], 0);
            }
        }
    }
    else
    {
        const int16_t *events;
        const int32_t *counts;

        rc = spangpu_fsk_rx_var(g->fsk, g->stage, SPANGPU_MEM_HOST, g->lens, g->max_samples, g->max_samples);
        cap = (rc < 0)  ?  rc  :  spangpu_fsk_events(g->fsk, &events, &counts);
        took = tick_taken(g);
        if (cap < 0)
            return cap;
        g->delivering = 1;
        for (c = 0;  c < g->n_ch;  c++)
        {
            fsk_rx_state_t *s = (fsk_rx_state_t *) g->handles[c];

            n = (counts[c] < cap)  ?  counts[c]  :  cap;
            if (s  &&  g->run[c] > 0)
            {
                for (i = 0;  i < n;  i++)
                {
                    const int v = events[(size_t) c*cap + i];

                    /* report_status_change(), fsk.c:343-349: the status handler if there is one, else put_bit */
                    if (v < 0  &&  s->status_handler)
                        s->status_handler(s->status_user_data, v);
                    else if (s->put_bit)
                        s->put_bit(s->put_bit_user_data, v);
                }
            }
        }
    }
    g->delivering = 0;
    return took;
}

/* The tick(s) that are due.  Callbacks may stage frames (a put_bit handler that answers by feeding its receiver, say): while
   a tick's callbacks run, a flush from inside them does nothing (`delivering`); when they are over, the tick those frames
   complete -- every attached channel has staged again -- runs at once instead of waiting for somebody to ask, so that no
   later xxx_rx() is refused as a second frame of a tick that nobody would ever have run. */
static int line_flush_locked(spangpu_line_group_t *g)
{
    int total = 0;
    int rc;

    if (g->delivering)
        return 0;
    for (;;)
    {
        if ((rc = line_flush_locked_tick(g)) < 0)
            return rc;
        total += rc;
        if (g->n_staged == 0  ||  g->n_staged < g->n_attached)
            break;
    }
    return total;
}

int spangpu_line_group_flush(spangpu_line_group_t *g)
{
    int rc;

    if (g == NULL)
        return SPANGPU_ERR_BAD_ARG;
    pthread_mutex_lock(&g->lock);
    rc = line_flush_locked(g);
    pthread_mutex_unlock(&g->lock);
    return rc;
}

/* One object's frame: a private bank runs it now (in slices); a shared one stages it (any thread; one submitter per
   receiver) and the tick runs when every attached receiver has staged, or when its owner calls
   spangpu_line_group_flush() at the deadline.  A frame longer than the group was made for, or a second frame for a
   receiver before the tick has run, is refused with -1: nothing is dropped silently. */
static int line_rx(spangpu_line_group_t *g, int channel, int private_grp, const int16_t amp[], int len)
{
    int n;
    int rc;

    if (len <= 0)
        return 0;                           /* as the reference: nothing to do (fsk.c:330 loops over len) */
    if (private_grp)
    {
        if (g->delivering)
            return -1;                      /* called from inside its own callback: refused, not dropped (the staging row is in use) */
        while (len > 0)
        {
            n = (len > g->max_samples)  ?  g->max_samples  :  len;
            memcpy(g->stage, amp, n*sizeof(int16_t));
            g->lens[0] = n;
            g->n_staged = 1;
            spangpu_line_group_flush(g);
            amp += n;
            len -= n;
        }
        return 0;
    }
    if (len > g->max_samples)
        return -1;
    pthread_mutex_lock(&g->lock);
    if (g->lens[channel])
    {
        pthread_mutex_unlock(&g->lock);
        return -1;
    }
    pthread_mutex_unlock(&g->lock);
    memcpy(g->stage + (size_t) channel*g->max_samples, amp, len*sizeof(int16_t));
    pthread_mutex_lock(&g->lock);
    g->lens[channel] = len;
    g->n_staged++;
    rc = (g->n_staged >= g->n_attached)  ?  line_flush_locked(g)  :  0;
    pthread_mutex_unlock(&g->lock);
    return (rc < 0)  ?  -1  :  0;
}

static void line_detach(spangpu_line_group_t *g, int channel, int private_grp)
{
    pthread_mutex_lock(&g->lock);
    g->handles[channel] = NULL;
    g->n_attached--;
    if (g->lens[channel])
    {
        /* its frame of the tick in progress goes with it */
        g->lens[channel] = 0;
        g->n_staged--;
    }
    if (!private_grp  &&  g->n_staged > 0  &&  g->n_staged >= g->n_attached)
        line_flush_locked(g);               /* it was the one the others were waiting for */
    pthread_mutex_unlock(&g->lock);
    if (private_grp)
        spangpu_line_group_destroy(g);
}

/* ---- fsk_rx -------------------------------------------------------------------------------------------- */
static fsk_rx_state_t *fsk_obj(fsk_rx_state_t *s, spangpu_line_group_t *g, int channel, int private_grp, span_put_bit_func_t put_bit,
                               void *user_data)
{
    const int mine = (s != NULL);

    if (mine)
        memset(s, 0, sizeof(*s));
    else if ((s = (fsk_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
        return NULL;
    s->caller_storage = mine;
    s->grp = g;
    s->channel = channel;
    s->private_grp = private_grp;
    s->put_bit = put_bit;
    s->put_bit_user_data = user_data;
    pthread_mutex_lock(&g->lock);
    g->handles[channel] = s;
    g->n_attached++;
    pthread_mutex_unlock(&g->lock);
    return s;
}

fsk_rx_state_t *fsk_rx_init(fsk_rx_state_t *s, const fsk_spec_t *spec, int framing_mode, span_put_bit_func_t put_bit, void *user_data)
{
    spangpu_line_group_t *g;

    /* s != NULL: the caller's storage (fsk.c:725-733); the handle goes there, the receiver itself is a private one-channel bank */
    if ((g = spangpu_fsk_group_create(0, spec, framing_mode, 1, 4096)) == NULL)
        return NULL;
    if ((s = fsk_obj(s, g, 0, 1, put_bit, user_data)) == NULL)
        spangpu_line_group_destroy(g);
    return s;
}

fsk_rx_state_t *spangpu_fsk_rx_attach(spangpu_line_group_t *g, int channel, span_put_bit_func_t put_bit, void *user_data)
{
    if (g == NULL  ||  g->is_mct  ||  channel < 0  ||  channel >= g->n_ch  ||  g->handles[channel])
        return NULL;
    return fsk_obj(NULL, g, channel, 0, put_bit, user_data);
}

int fsk_rx(fsk_rx_state_t *s, const int16_t *amp, int len)
{
    return line_rx(s->grp, s->channel, s->private_grp, amp, len);
}

int fsk_rx_fillin(fsk_rx_state_t *s, int len)
{
    return (spangpu_fsk_fillin(s->grp->fsk, s->channel, len) < 0)  ?  -1  :  0;
}

int fsk_rx_restart(fsk_rx_state_t *s, const fsk_spec_t *spec, int framing_mode)
{
    if (spec == NULL)
        return -1;
    if (spec->freq_zero != s->grp->spec.freq_zero  ||  spec->freq_one != s->grp->spec.freq_one
        ||  spec->baud_rate != s->grp->spec.baud_rate  ||  spec->min_level != s->grp->spec.min_level)
    {
        /* Another modem (fsk.c:670-723 re-initialises everything but the callbacks).  A bank runs one spec -- its window
           length follows the baud rate -- so an object on a shared bank cannot leave it; an object of its own gets a new
           one-channel bank. */
        spangpu_line_group_t *g;

        if (!s->private_grp)
            return -1;
        if ((g = spangpu_fsk_group_create(0, spec, framing_mode, 1, 4096)) == NULL)
            return -1;
        line_detach(s->grp, s->channel, 1);
        s->grp = g;
        s->channel = 0;
        pthread_mutex_lock(&g->lock);
        g->handles[0] = s;
        g->n_attached++;
        pthread_mutex_unlock(&g->lock);
        return 0;
    }
    return (spangpu_fsk_restart(s->grp->fsk, s->channel, framing_mode) < 0)  ?  -1  :  0;
}

int fsk_rx_release(fsk_rx_state_t *s)
{
    /* what ends an object in the caller's storage: its place on the bank (or its private bank) goes, the storage stays */
    if (s  &&  s->grp)
    {
        line_detach(s->grp, s->channel, s->private_grp);
        s->grp = NULL;
    }
    return 0;
}

int fsk_rx_free(fsk_rx_state_t *s)
{
    if (s == NULL)
        return 0;
    fsk_rx_release(s);
    if (!s->caller_storage)
        free(s);
    return 0;
}

void fsk_rx_set_put_bit(fsk_rx_state_t *s, span_put_bit_func_t put_bit, void *user_data)
{
    s->put_bit = put_bit;
    s->put_bit_user_data = user_data;
}

void fsk_rx_set_modem_status_handler(fsk_rx_state_t *s, span_modem_status_func_t handler, void *user_data)
{
    s->status_handler = handler;
    s->status_user_data = user_data;
}

void fsk_rx_set_signal_cutoff(fsk_rx_state_t *s, float cutoff)
{
    spangpu_fsk_set_signal_cutoff(s->grp->fsk, s->channel, cutoff);
}

void fsk_rx_set_frame_parameters(fsk_rx_state_t *s, int data_bits, int parity, int stop_bits)
{
    spangpu_fsk_set_frame_parameters(s->grp->fsk, s->channel, data_bits, parity, stop_bits);
}

/* State word positions (fsk_dev.hpp): 8 power reading, 26 parity errors, 27 framing errors */
static int fsk_word(fsk_rx_state_t *s, int idx, int reset)
{
    int32_t w[28 + 4*128];
    int v;

    if (spangpu_fsk_get_state(s->grp->fsk, s->channel, w) < 0)
        return 0;
    v = w[idx];
    if (reset  &&  v != 0)
    {
        w[idx] = 0;
        spangpu_fsk_set_state(s->grp->fsk, s->channel, w);
    }
    return v;
}

float fsk_rx_signal_power(fsk_rx_state_t *s)
{
    /* power_meter_current_dbm0(), power_meter.c:114-121 */
    const int32_t reading = fsk_word(s, 8, 0);

    if (reading <= 0)
        return -96.329f + (3.14f + 3.02f);
    return 10.0f*log10f((float) reading/(32767.0f*32767.0f) + 1.0e-10f) + (3.14f + 3.02f);
}

int fsk_rx_get_parity_errors(fsk_rx_state_t *s, bool reset)
{
    return fsk_word(s, 26, reset);
}

int fsk_rx_get_framing_errors(fsk_rx_state_t *s, bool reset)
{
    return fsk_word(s, 27, reset);
}

/* ---- modem_connect_tones_rx ---------------------------------------------------------------------------- */
static modem_connect_tones_rx_state_t *mct_obj(modem_connect_tones_rx_state_t *s, spangpu_line_group_t *g, int channel, int private_grp,
                                               span_tone_report_func_t tone_callback, void *user_data)
{
    const int mine = (s != NULL);

    if (mine)
        memset(s, 0, sizeof(*s));
    else if ((s = (modem_connect_tones_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
        return NULL;
    s->caller_storage = mine;
    s->grp = g;
    s->channel = channel;
    s->private_grp = private_grp;
    s->tone_callback = tone_callback;
    s->callback_data = user_data;
    pthread_mutex_lock(&g->lock);
    g->handles[channel] = s;
    g->n_attached++;
    pthread_mutex_unlock(&g->lock);
    return s;
}

modem_connect_tones_rx_state_t *modem_connect_tones_rx_init(modem_connect_tones_rx_state_t *s, int tone_type,
                                                            span_tone_report_func_t tone_callback, void *user_data)
{
    spangpu_line_group_t *g;

    /* s != NULL: the caller's storage (modem_connect_tones.c:823-830) */
    if ((g = spangpu_modem_connect_tones_group_create(0, tone_type, tone_callback != NULL, 1, 4096)) == NULL)
        return NULL;
    if ((s = mct_obj(s, g, 0, 1, tone_callback, user_data)) == NULL)
        spangpu_line_group_destroy(g);
    return s;
}

modem_connect_tones_rx_state_t *spangpu_modem_connect_tones_rx_attach(spangpu_line_group_t *g, int channel,
                                                                      span_tone_report_func_t tone_callback, void *user_data)
{
    if (g == NULL  ||  !g->is_mct  ||  channel < 0  ||  channel >= g->n_ch  ||  g->handles[channel])
        return NULL;
    return mct_obj(NULL, g, channel, 0, tone_callback, user_data);
}

int modem_connect_tones_rx(modem_connect_tones_rx_state_t *s, const int16_t amp[], int len)
{
    return line_rx(s->grp, s->channel, s->private_grp, amp, len);
}

int modem_connect_tones_rx_fillin(modem_connect_tones_rx_state_t *s, int len)
{
    /* modem_connect_tones.c:787-790: nothing to do */
    (void) s;
    (void) len;
    return 0;
}

int modem_connect_tones_rx_get(modem_connect_tones_rx_state_t *s)
{
    const int hit = spangpu_mct_get(s->grp->mct, s->channel);

    return (hit < 0)  ?  MODEM_CONNECT_TONES_NONE  :  hit;
}

int modem_connect_tones_rx_release(modem_connect_tones_rx_state_t *s)
{
    if (s  &&  s->grp)
    {
        line_detach(s->grp, s->channel, s->private_grp);
        s->grp = NULL;
    }
    return 0;
}

int modem_connect_tones_rx_free(modem_connect_tones_rx_state_t *s)
{
    if (s == NULL)
        return 0;
    modem_connect_tones_rx_release(s);
    if (!s->caller_storage)
        free(s);
    return 0;
}

const char *modem_connect_tone_to_str(int tone)
{
    /* modem_connect_tones.c:84-112 */
    static const char *names[] =
    {
        "No tone", "FAX CNG", "ANS or FAX CED", "ANS/", "ANSam", "ANSam/", "FAX preamble", "FAX CED or preamble", "Bell ANS",
        "Calling tone"
    };
    return (tone >= 0  &&  tone <= MODEM_CONNECT_TONES_CALLING_TONE)  ?  names[tone]  :  "???";
}

/* ---- dtmf_tx: one private sender per object ---------------------------------------------------------- */
dtmf_tx_state_t *dtmf_tx_init(dtmf_tx_state_t *s, digits_tx_callback_t callback, void *user_data)
{
    const int mine = (s != NULL);

    if (mine)
        memset(s, 0, sizeof(*s));
    else if ((s = (dtmf_tx_state_t *) calloc(1, sizeof(*s))) == NULL)
        return NULL;
    s->caller_storage = mine;
    s->callback = callback;
    s->callback_data = user_data;
    if (spangpu_txbank_create(&s->bank, 0, SPANGPU_TX_DTMF, 1) != SPANGPU_OK)
    {
        if (!mine)
            free(s);
        return NULL;
    }
    return s;
}

int dtmf_tx_release(dtmf_tx_state_t *s)
{
    if (s  &&  s->bank)
    {
        spangpu_txbank_destroy(s->bank);
        s->bank = NULL;
    }
    return 0;
}

int dtmf_tx_free(dtmf_tx_state_t *s)
{
    if (s)
    {
        dtmf_tx_release(s);
        if (!s->caller_storage)
            free(s);
    }
    return 0;
}

void dtmf_tx_set_level(dtmf_tx_state_t *s, int level, int twist)
{
    spangpu_txbank_set_level(s->bank, 0, 1, level, twist);
}

void dtmf_tx_set_timing(dtmf_tx_state_t *s, int on_time, int off_time)
{
    spangpu_txbank_set_timing(s->bank, 0, 1, on_time, off_time);
}

int dtmf_tx_put(dtmf_tx_state_t *s, const char *digits, int len)
{
    const int rc = spangpu_txbank_put(s->bank, 0, 1, digits, len);

    if (rc == 0)
        s->puts++;
    return (rc < 0)  ?  -1  :  rc;
}

int dtmf_tx(dtmf_tx_state_t *s, int16_t amp[], int max_samples)
{
    int len = 0;

    if (max_samples <= 0)
        return 0;
    for (;;)
    {
        int got = 0;
        int before;

        if (spangpu_txbank_tx(s->bank, SPANGPU_MEM_HOST, amp + len, max_samples - len, max_samples - len, &got) < 0)
            break;
        len += got;
        /* dtmf.c:566-573: the queue ran dry with room left in the buffer -- "see if we can get some more digits": the
           callback answers by calling dtmf_tx_put(), and the sender carries on where it stopped */
        if (len >= max_samples  ||  s->callback == NULL)
            break;
        before = s->puts;
        s->callback(s->callback_data);
        if (s->puts == before)
            break;
    }
    return len;
}

/* ---- bell_mf_tx / r2_mf_tx: one private sender per object (src/bell_r2_mf.c:281-372, 386-462) ----------------- */
bell_mf_tx_state_t *bell_mf_tx_init(bell_mf_tx_state_t *s)
{
    const int mine = (s != NULL);

    if (mine)
        memset(s, 0, sizeof(*s));
    else if ((s = (bell_mf_tx_state_t *) calloc(1, sizeof(*s))) == NULL)
        return NULL;
    s->caller_storage = mine;
    if (spangpu_txbank_create(&s->bank, 0, SPANGPU_TX_BELL_MF, 1) != SPANGPU_OK)
    {
        if (!mine)
            free(s);
        return NULL;
    }
    return s;
}

int bell_mf_tx_release(bell_mf_tx_state_t *s)
{
    if (s  &&  s->bank)
    {
        spangpu_txbank_destroy(s->bank);
        s->bank = NULL;
    }
    return 0;
}

int bell_mf_tx_free(bell_mf_tx_state_t *s)
{
    if (s)
    {
        bell_mf_tx_release(s);
        if (!s->caller_storage)
            free(s);
    }
    return 0;
}

int bell_mf_tx_put(bell_mf_tx_state_t *s, const char *digits, int len)
{
    const int rc = spangpu_txbank_put(s->bank, 0, 1, digits, len);

    return (rc < 0)  ?  -1  :  rc;
}

int bell_mf_tx(bell_mf_tx_state_t *s, int16_t amp[], int max_samples)
{
    int len = 0;

    if (max_samples <= 0  ||  spangpu_txbank_tx(s->bank, SPANGPU_MEM_HOST, amp, max_samples, max_samples, &len) < 0)
        return 0;
    return len;
}

r2_mf_tx_state_t *r2_mf_tx_init(r2_mf_tx_state_t *s, bool fwd)
{
    const int mine = (s != NULL);

    if (mine)
        memset(s, 0, sizeof(*s));
    else if ((s = (r2_mf_tx_state_t *) calloc(1, sizeof(*s))) == NULL)
        return NULL;
    s->caller_storage = mine;
    if (spangpu_txbank_create(&s->bank, 0, fwd  ?  SPANGPU_TX_R2_MF_FWD  :  SPANGPU_TX_R2_MF_BACK, 1) != SPANGPU_OK)
    {
        if (!mine)
            free(s);
        return NULL;
    }
    return s;
}

int r2_mf_tx_release(r2_mf_tx_state_t *s)
{
    if (s  &&  s->bank)
    {
        spangpu_txbank_destroy(s->bank);
        s->bank = NULL;
    }
    return 0;
}

int r2_mf_tx_free(r2_mf_tx_state_t *s)
{
    if (s)
    {
        r2_mf_tx_release(s);
        if (!s->caller_storage)
            free(s);
    }
    return 0;
}

int r2_mf_tx_put(r2_mf_tx_state_t *s, char digit)
{
    /* one signal at a time: the tone of `digit` until the next put, 0 switches it off (bell_r2_mf.c:414-434) */
    (void) spangpu_txbank_put(s->bank, 0, 1, &digit, 1);
    return 0;
}

int r2_mf_tx(r2_mf_tx_state_t *s, int16_t amp[], int samples)
{
    int len = 0;

    if (samples <= 0  ||  spangpu_txbank_tx(s->bank, SPANGPU_MEM_HOST, amp, samples, samples, &len) < 0)
        return 0;
    return len;
}
