/*
 * shim_modem.c -- host side (plain C) of the spandsp-named modem receiver entry points declared in
 * include/spangpu_spandsp.h: v29_rx*, v27ter_rx*, v17_rx*.  No signal processing happens here: samples
 * go to a modem bank (include/spangpu.h, "Modem receiver banks"), the HIP kernel leaves each channel's
 * put_bit / status stream in order, and this file replays it through the caller's callbacks exactly as
 * the reference's report_status_change() / put_bit() would (src/v29rx.c:171-178, :365-397):
 * negative entries go to the modem status handler if one is set, else to put_bit; bits go to put_bit.
 * Without a GPU every init returns NULL: there is no CPU implementation.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "spangpu_spandsp.h"

#define MAX_WORDS   1024

struct spangpu_modem_group_s
{
    spangpu_modem_t *bank;
    int kind;
    int bit_rate;
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
    int qam_tap;                /* some object of the group has a qam report handler: the bank records the reports */
};

typedef struct
{
    int kind;
    spangpu_modem_group_t *grp;
    int channel;
    int private_grp;
    int bit_rate;
    span_put_bit_func_t put_bit;
    void *put_bit_user_data;
    span_modem_status_func_t status_handler;
    void *status_user_data;
    qam_report_handler_t qam_report;
    void *qam_user_data;
    logging_state_t logging;
    uint32_t words[MAX_WORDS];              /* scratch for state reads (equalizer_state() hands out a view) */
    int n_floats;
} modem_obj_t;

struct v29_rx_state_s { modem_obj_t o; };
struct v27ter_rx_state_s { modem_obj_t o; };
struct v17_rx_state_s { modem_obj_t o; };

static int rate_ok(int kind, int bit_rate)
{
    switch (kind)
    {
    case SPANGPU_V29:
        return bit_rate == 9600  ||  bit_rate == 7200  ||  bit_rate == 4800;
    case SPANGPU_V27TER:
        return bit_rate == 4800  ||  bit_rate == 2400;
    case SPANGPU_V17:
        return bit_rate == 14400  ||  bit_rate == 12000  ||  bit_rate == 9600  ||  bit_rate == 7200  ||  bit_rate == 4800;
    }
    return 0;
}

spangpu_modem_group_t *spangpu_modem_group_create(int device, int kind, int n_channels, int bit_rate, int max_samples)
{
    spangpu_modem_group_t *g;

    if (n_channels <= 0  ||  max_samples <= 0  ||  !rate_ok(kind, bit_rate))
        return NULL;
    if ((g = (spangpu_modem_group_t *) calloc(1, sizeof(*g))) == NULL)
        return NULL;
    g->kind = kind;
    g->bit_rate = bit_rate;
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
    if (g->stage == NULL  ||  g->handles == NULL  ||  g->lens == NULL  ||  g->run == NULL
        ||  spangpu_modem_create(&g->bank, device, kind, n_channels, bit_rate) != SPANGPU_OK)
    {
        spangpu_modem_group_destroy(g);
        return NULL;
    }
    return g;
}

int spangpu_modem_group_destroy(spangpu_modem_group_t *g)
{
    if (g == NULL)
        return 0;
    if (g->bank)
        spangpu_modem_destroy(g->bank);
    free(g->stage);
    free(g->handles);
    free(g->lens);
    free(g->run);
    pthread_mutex_destroy(&g->lock);
    free(g);
    return 0;
}

spangpu_modem_t *spangpu_modem_group_bank(spangpu_modem_group_t *g)
{
    return g  ?  g->bank  :  NULL;
}

/* One qam_report() call from its record (include/spangpu.h: spangpu_modem_qam_reports()) */
static void deliver_qam(modem_obj_t *o, const uint32_t *r)
{
    complexf_t constel;
    complexf_t target;

    if (r[1])
    {
        o->qam_report(o->qam_user_data, NULL, NULL, (int) r[2]);
        return;
    }
    memcpy(&constel.re, &r[3], 4);
    memcpy(&constel.im, &r[4], 4);
    memcpy(&target.re, &r[5], 4);
    memcpy(&target.im, &r[6], 4);
    o->qam_report(o->qam_user_data, &constel, &target, (int) r[2]);
}

/* The callbacks of one rx call, in the order the reference makes them: record q comes after q[0] put_bit / status calls */
static void deliver(modem_obj_t *o, const int8_t *ev, int n, const uint32_t *qam, int nq)
{
    int i;
    int q = 0;

    for (i = 0;  i < n;  i++)
    {
        while (q < nq  &&  (int) qam[7*q] <= i)
        {
            if (o->qam_report)
                deliver_qam(o, &qam[7*q]);
            q++;
        }
        if (ev[i] < 0  &&  o->status_handler)
            o->status_handler(o->status_user_data, ev[i]);
        else if (o->put_bit)
            o->put_bit(o->put_bit_user_data, ev[i]);
    }
    for (  ;  q < nq;  q++)
    {
        if (o->qam_report)
            deliver_qam(o, &qam[7*q]);
    }
}

/* Run the tick with the receivers that have staged a frame; the others sit it out, untouched (as the reference's are
   when their xxx_rx() is not called), and may stage for the next one.  Returns how many took part. */
static int group_flush_locked_tick(spangpu_modem_group_t *g)
{
    const int8_t *events;
    const int32_t *counts;
    const uint32_t *qam = NULL;
    const int32_t *qcounts = NULL;
    int cap;
    int qcap = 0;
    int c;
    int rc;

    if (g->n_staged == 0)
        return 0;
    rc = spangpu_modem_rx_var(g->bank, g->stage, SPANGPU_MEM_HOST, g->lens, g->max_samples, g->max_samples);
    /* the put_bit stream comes up packed (a header word and the data bits per channel, status reports as a sparse list) and is
       spread out on the host: spangpu_modem_events_packed() */
    cap = (rc < 0)  ?  rc  :  spangpu_modem_events_packed(g->bank, &events, &counts);
    if (cap >= 0  &&  g->qam_tap)
        qcap = spangpu_modem_qam_reports(g->bank, &qam, &qcounts);
    /* The tick is over whatever happened: its frames are taken off the staging area before anything is delivered, so
       that a failure cannot make every later xxx_rx() a "second frame" (or run the same frames again), and so that a
       callback which stages a new frame sees a clean slate (that frame waits for the next tick). */
    rc = g->n_staged;
    memcpy(g->run, g->lens, sizeof(int32_t)*g->n_ch);
    memset(g->lens, 0, sizeof(int32_t)*g->n_ch);
    g->n_staged = 0;
    if (cap < 0)
        return cap;
    if (qcap < 0)
        return qcap;
    g->delivering = 1;
    for (c = 0;  c < g->n_ch;  c++)
    {
        if (g->handles[c]  &&  g->run[c] > 0)
        {
            deliver((modem_obj_t *) g->handles[c], events + (size_t) c*cap, (counts[c] < cap)  ?  counts[c]  :  cap,
                    qam  ?  qam + (size_t) c*qcap*7  :  NULL, qam  ?  ((qcounts[c] < qcap)  ?  qcounts[c]  :  qcap)  :  0);
        }
    }
    g->delivering = 0;
    return rc;
}

/* The tick(s) that are due.  Callbacks may stage frames (a put_bit handler that answers by feeding its receiver, say): while
   a tick's callbacks run, a flush from inside them does nothing (`delivering`); when they are over, the tick those frames
   complete -- every attached channel has staged again -- runs at once instead of waiting for somebody to ask, so that no
   later xxx_rx() is refused as a second frame of a tick that nobody would ever have run. */
static int group_flush_locked(spangpu_modem_group_t *g)
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

int spangpu_modem_group_flush(spangpu_modem_group_t *g)
{
    int rc;

    if (g == NULL)
        return SPANGPU_ERR_BAD_ARG;
    pthread_mutex_lock(&g->lock);
    rc = group_flush_locked(g);
    pthread_mutex_unlock(&g->lock);
    return rc;
}

static modem_obj_t *obj_new(size_t size, int kind, spangpu_modem_group_t *g, int channel, int private_grp, int bit_rate,
                            span_put_bit_func_t put_bit, void *user_data)
{
    modem_obj_t *o;

    if ((o = (modem_obj_t *) calloc(1, size)) == NULL)
        return NULL;
    o->kind = kind;
    o->grp = g;
    o->channel = channel;
    o->private_grp = private_grp;
    o->bit_rate = bit_rate;
    o->put_bit = put_bit;
    o->put_bit_user_data = user_data;
    /* what span_log_init(.., SPAN_LOG_NONE, NULL) + span_log_set_protocol() leave behind (v29rx.c:1120-1121 and twins) */
    memset(&o->logging, 0, sizeof(o->logging));
    o->logging.samples_per_second = 8000;
    o->logging.protocol = (kind == SPANGPU_V29)  ?  "V.29 RX"  :  (kind == SPANGPU_V27TER)  ?  "V.27ter RX"  :  "V.17 RX";
    spangpu_modem_state_words(kind, &o->n_floats, NULL);
    pthread_mutex_lock(&g->lock);
    g->handles[channel] = o;
    g->n_attached++;
    pthread_mutex_unlock(&g->lock);
    return o;
}

static modem_obj_t *obj_init(size_t size, int kind, int bit_rate, span_put_bit_func_t put_bit, void *user_data)
{
    spangpu_modem_group_t *g;
    modem_obj_t *o;

    if (!rate_ok(kind, bit_rate))
        return NULL;
    /* a private object takes whatever one call hands it, in slices of at most 4096 samples */
    if ((g = spangpu_modem_group_create(0, kind, 1, bit_rate, 4096)) == NULL)
        return NULL;
    if ((o = obj_new(size, kind, g, 0, 1, bit_rate, put_bit, user_data)) == NULL)
        spangpu_modem_group_destroy(g);
    return o;
}

static modem_obj_t *obj_attach(size_t size, int kind, spangpu_modem_group_t *g, int channel,
                               span_put_bit_func_t put_bit, void *user_data)
{
    if (g == NULL  ||  g->kind != kind  ||  channel < 0  ||  channel >= g->n_ch  ||  g->handles[channel])
        return NULL;
    return obj_new(size, kind, g, channel, 0, g->bit_rate, put_bit, user_data);
}

/* The bank records the reports as long as any object of the group wants them (the tap changes the kernel variant, not
   the results) */
static void obj_set_qam(modem_obj_t *o, qam_report_handler_t handler, void *user_data)
{
    spangpu_modem_group_t *g = o->grp;
    int c;
    int any = 0;

    o->qam_report = handler;
    o->qam_user_data = user_data;
    for (c = 0;  c < g->n_ch;  c++)
    {
        if (g->handles[c]  &&  ((modem_obj_t *) g->handles[c])->qam_report)
            any = 1;
    }
    if (any != g->qam_tap)
    {
        g->qam_tap = any;
        spangpu_modem_qam_tap(g->bank, any);
    }
}

static int obj_rx(modem_obj_t *o, const int16_t amp[], int len)
{
    spangpu_modem_group_t *g = o->grp;
    int n;
    int rc;

    if (len <= 0)
        return 0;                           /* as the reference: nothing to do (v29rx.c:867-965 loops over len) */
    if (o->private_grp)
    {
        if (g->delivering)
            return -1;                      /* called from inside its own callback: refused, not dropped (the staging row is in use) */
        while (len > 0)
        {
            n = (len > g->max_samples)  ?  g->max_samples  :  len;
            memcpy(g->stage, amp, n*sizeof(int16_t));
            g->lens[0] = n;
            g->n_staged = 1;
            spangpu_modem_group_flush(g);
            amp += n;
            len -= n;
        }
        return 0;
    }
    /* A shared bank advances in ticks: a frame per receiver that has one (any thread may stage; one submitter per
       receiver, as for a spandsp object).  The tick runs when every attached receiver has staged, or when its owner calls
       spangpu_modem_group_flush() at the deadline.  Nothing is dropped silently: a frame longer than the group was made
       for, or a second frame for a receiver before the tick has run, is refused with -1. */
    if (len > g->max_samples)
        return -1;
    pthread_mutex_lock(&g->lock);
    if (g->lens[o->channel])
    {
        pthread_mutex_unlock(&g->lock);
        return -1;
    }
    pthread_mutex_unlock(&g->lock);
    memcpy(g->stage + (size_t) o->channel*g->max_samples, amp, len*sizeof(int16_t));
    pthread_mutex_lock(&g->lock);
    g->lens[o->channel] = len;
    g->n_staged++;
    rc = (g->n_staged >= g->n_attached)  ?  group_flush_locked(g)  :  0;
    pthread_mutex_unlock(&g->lock);
    return (rc < 0)  ?  -1  :  0;
}

static int obj_free(modem_obj_t *o)
{
    if (o == NULL)
        return 0;
    if (o->grp  &&  o->qam_report)
        obj_set_qam(o, NULL, NULL);
    if (o->grp)
    {
        spangpu_modem_group_t *g = o->grp;

        pthread_mutex_lock(&g->lock);
        g->handles[o->channel] = NULL;
        g->n_attached--;
        if (g->lens[o->channel])
        {
            /* its frame of the tick in progress goes with it */
            g->lens[o->channel] = 0;
            g->n_staged--;
        }
        if (!o->private_grp  &&  g->n_staged > 0  &&  g->n_staged >= g->n_attached)
            group_flush_locked(g);          /* it was the one the others were waiting for */
        pthread_mutex_unlock(&g->lock);
        if (o->private_grp)
            spangpu_modem_group_destroy(g);
    }
    free(o);
    return 0;
}

/* A restart that changes the bit rate of a V.27ter / V.17 private object moves it to a bank of the new rate,
   carrying the words the reference's restart keeps (they are all the words: restart edits in place). */
static int obj_restart(modem_obj_t *o, int bit_rate, int flag)
{
    spangpu_modem_group_t *g = o->grp;
    spangpu_modem_group_t *ng;
    int words;

    if (!rate_ok(o->kind, bit_rate))
        return -1;
    if (o->kind != SPANGPU_V29  &&  bit_rate != g->bit_rate)
    {
        if (!o->private_grp)
            return -1;                      /* a shared bank runs one rate */
        words = spangpu_modem_get_state(g->bank, 0, o->words);
        if (words < 0  ||  (ng = spangpu_modem_group_create(0, o->kind, 1, bit_rate, g->max_samples)) == NULL)
            return -1;
        spangpu_modem_set_state(ng->bank, 0, o->words);
        ng->handles[0] = o;
        ng->n_attached = 1;
        g->handles[0] = NULL;
        spangpu_modem_group_destroy(g);
        o->grp = g = ng;
    }
    o->bit_rate = bit_rate;
    return (spangpu_modem_restart_ex(g->bank, o->channel, bit_rate, flag) < 0)  ?  -1  :  0;
}

static const uint32_t *obj_words(modem_obj_t *o)
{
    if (spangpu_modem_get_state(o->grp->bank, o->channel, o->words) < 0)
        return NULL;
    return o->words;
}

/* dds_frequencyf(), dds_float.c:2115-2118 */
static float phase_rate_hz(int32_t rate)
{
    return (float) rate*8000.0f/(65536.0f*65536.0f);
}

/* power_meter_current_dbm0(), power_meter.c:114-121 (DBM0_MAX_POWER = 3.14 + 3.02) */
static float reading_dbm0(int32_t reading)
{
    if (reading <= 0)
        return -96.329f + (3.14f + 3.02f);
    return 10.0f*log10f((float) reading/(32767.0f*32767.0f) + 1.0e-10f) + (3.14f + 3.02f);
}

/* State word positions the getters need (the "State word map" comments of v29_dev.hpp, v27ter_dev.hpp, v17_dev.hpp) */
enum
{
    V29_F_EQ_COEFF = 40, V29_I_PHASE_RATE = 11, V29_I_POWER = 13, V29_I_TOTAL_CORR = 39,
    V27_F_EQ_COEFF = 33, V27_I_PHASE_RATE = 15, V27_I_POWER = 17, V27_I_TOTAL_CORR = 26,
    V17_F_EQ_COEFF = 40, V17_I_PHASE_RATE = 14, V17_I_POWER = 16, V17_I_TOTAL_CORR = 44
};

#define DEFINE_MODEM(pfx, T, KIND, EQ_F, EQ_LEN, I_RATE, I_POWER, POWER_ADJ)                                         \
T *pfx##_init(T *s, int bit_rate, span_put_bit_func_t put_bit, void *user_data)                                      \
{                                                                                                                    \
    modem_obj_t *o;                                                                                                  \
    if (s)                                                                                                           \
    {                                                                                                                \
        /* re-initialise in place (the reference memset()s the caller's struct) */                                   \
        if (!rate_ok(KIND, bit_rate)  ||  !s->o.private_grp)                                                         \
            return NULL;                                                                                             \
        s->o.put_bit = put_bit;                                                                                      \
        s->o.put_bit_user_data = user_data;                                                                          \
        s->o.status_handler = NULL;                                                                                  \
        if (obj_restart(&s->o, bit_rate, 0) < 0)                                                                     \
            return NULL;                                                                                             \
        spangpu_modem_set_signal_cutoff(s->o.grp->bank, 0, (KIND == SPANGPU_V29)  ?  -28.5f  :  -45.5f);             \
        return s;                                                                                                    \
    }                                                                                                                \
    o = obj_init(sizeof(T), KIND, bit_rate, put_bit, user_data);                                                     \
    return (T *) o;                                                                                                  \
}                                                                                                                    \
T *spangpu_##pfx##_attach(spangpu_modem_group_t *g, int channel, span_put_bit_func_t put_bit, void *user_data)       \
{                                                                                                                    \
    return (T *) obj_attach(sizeof(T), KIND, g, channel, put_bit, user_data);                                        \
}                                                                                                                    \
int pfx(T *s, const int16_t amp[], int len)                                                                          \
{                                                                                                                    \
    return obj_rx(&s->o, amp, len);                                                                                  \
}                                                                                                                    \
int pfx##_fillin(T *s, int len)                                                                                      \
{                                                                                                                    \
    spangpu_modem_fillin(s->o.grp->bank, s->o.channel, len);                                                         \
    return 0;                                                                                                        \
}                                                                                                                    \
int pfx##_release(T *s)                                                                                              \
{                                                                                                                    \
    (void) s;                                                                                                        \
    return 0;                                                                                                        \
}                                                                                                                    \
int pfx##_free(T *s)                                                                                                 \
{                                                                                                                    \
    return obj_free(s  ?  &s->o  :  NULL);                                                                           \
}                                                                                                                    \
void pfx##_set_put_bit(T *s, span_put_bit_func_t put_bit, void *user_data)                                           \
{                                                                                                                    \
    s->o.put_bit = put_bit;                                                                                          \
    s->o.put_bit_user_data = user_data;                                                                              \
}                                                                                                                    \
void pfx##_set_modem_status_handler(T *s, span_modem_status_func_t handler, void *user_data)                         \
{                                                                                                                    \
    s->o.status_handler = handler;                                                                                   \
    s->o.status_user_data = user_data;                                                                               \
}                                                                                                                    \
void pfx##_set_qam_report_handler(T *s, qam_report_handler_t handler, void *user_data)                               \
{                                                                                                                    \
    obj_set_qam(&s->o, handler, user_data);                                                                          \
}                                                                                                                    \
int pfx##_equalizer_state(T *s, complexf_t **coeffs)                                                                 \
{                                                                                                                    \
    const uint32_t *w = obj_words(&s->o);                                                                            \
    *coeffs = w  ?  (complexf_t *) (s->o.words + EQ_F)  :  NULL;                                                     \
    return w  ?  EQ_LEN  :  0;                                                                                       \
}                                                                                                                    \
float pfx##_carrier_frequency(T *s)                                                                                  \
{                                                                                                                    \
    const uint32_t *w = obj_words(&s->o);                                                                            \
    return w  ?  phase_rate_hz((int32_t) w[s->o.n_floats + I_RATE])  :  0.0f;                                        \
}                                                                                                                    \
float pfx##_signal_power(T *s)                                                                                       \
{                                                                                                                    \
    const uint32_t *w = obj_words(&s->o);                                                                            \
    return w  ?  reading_dbm0((int32_t) w[s->o.n_floats + I_POWER]) + POWER_ADJ  :  0.0f;                            \
}                                                                                                                    \
void pfx##_set_signal_cutoff(T *s, float cutoff)                                                                     \
{                                                                                                                    \
    spangpu_modem_set_signal_cutoff(s->o.grp->bank, s->o.channel, cutoff);                                           \
}                                                                                                                    \
logging_state_t *pfx##_get_logging_state(T *s)                                                                       \
{                                                                                                                    \
    return &s->o.logging;                                                                                            \
}

DEFINE_MODEM(v29_rx, v29_rx_state_t, SPANGPU_V29, V29_F_EQ_COEFF, 33, V29_I_PHASE_RATE, V29_I_POWER, 3.98f)
DEFINE_MODEM(v27ter_rx, v27ter_rx_state_t, SPANGPU_V27TER, V27_F_EQ_COEFF, 32, V27_I_PHASE_RATE, V27_I_POWER, 3.98f)
DEFINE_MODEM(v17_rx, v17_rx_state_t, SPANGPU_V17, V17_F_EQ_COEFF, 33, V17_I_PHASE_RATE, V17_I_POWER, 3.98f)

int v29_rx_restart(v29_rx_state_t *s, int bit_rate, bool old_train)
{
    return obj_restart(&s->o, bit_rate, old_train);
}

int v27ter_rx_restart(v27ter_rx_state_t *s, int bit_rate, bool old_train)
{
    return obj_restart(&s->o, bit_rate, old_train);
}

int v17_rx_restart(v17_rx_state_t *s, int bit_rate, int short_train)
{
    return obj_restart(&s->o, bit_rate, short_train);
}

/* v29rx.c:153-156: total correction / (RX_PULSESHAPER_COEFF_SETS*10/3) */
float v29_rx_symbol_timing_correction(v29_rx_state_t *s)
{
    const uint32_t *w = obj_words(&s->o);

    return w  ?  (float) (int32_t) w[s->o.n_floats + V29_I_TOTAL_CORR]/((float) 48*10.0f/3.0f)  :  0.0f;
}

/* v27ter_rx.c:141-147 */
float v27ter_rx_symbol_timing_correction(v27ter_rx_state_t *s)
{
    const uint32_t *w = obj_words(&s->o);
    int steps_per_symbol = (s->o.bit_rate == 4800)  ?  8*5  :  12*20/3;

    return w  ?  (float) (int32_t) w[s->o.n_floats + V27_I_TOTAL_CORR]/(float) steps_per_symbol  :  0.0f;
}

/* v17rx.c:171-174 */
float v17_rx_symbol_timing_correction(v17_rx_state_t *s)
{
    const uint32_t *w = obj_words(&s->o);

    return w  ?  (float) (int32_t) w[s->o.n_floats + V17_I_TOTAL_CORR]/((float) 192*10.0f/3.0f)  :  0.0f;
}
