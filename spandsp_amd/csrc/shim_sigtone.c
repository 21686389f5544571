/*
 * shim_sigtone.c -- the spandsp-named entry points of the in-band signalling tone processor, declared in
 * include/spangpu_spandsp.h: sig_tone_rx_init/_rx/_set_mode/_release/_free and sig_tone_tx_init/_tx/_set_mode/
 * _release/_free (reference: src/spandsp/sig_tone.h:57-176, src/sig_tone.c:246-738).  Host code only: an object is a
 * one-channel bank of include/spangpu.h's signalling tone banks; N channels on one launch per tick are what those banks
 * are for (spangpu_sigtone_rx_create() ...).
 *
 * What a callback may do, kept exact.  The reference calls the receiver's callback from inside sig_tone_rx(), at the
 * sample of the report, and the sender's from inside sig_tone_tx(), at the end of the segment whose duration ran out;
 * either may set a new mode, which then applies to the rest of the same frame.  The sender bank stops a channel at its
 * request by construction.  The receiver here runs the whole frame in one launch and, only if that launch reported
 * anything, puts the state back as it was and runs the frame again in pieces around the reports, calling back in
 * between.
 */
#include <stdlib.h>
#include <string.h>

#include "spangpu.h"
#include "spangpu_spandsp.h"

#define RX_WORDS    27

#define RX_MAGIC    0x53475258u     /* an object this file made (and may therefore re-initialise in place) */
#define TX_MAGIC    0x53475458u

struct sig_tone_rx_state_s
{
    uint32_t magic;
    spangpu_sigtone_rx_t *bank;
    span_tone_report_func_t sig_update;
    void *user_data;
    int16_t *keep;              /* the frame as it came in, for the piecewise second run */
    int keep_cap;
    int in_callback;
    int mode_set;               /* a callback called sig_tone_rx_set_mode() ... */
    int new_mode;               /* ... with this */
};

struct sig_tone_tx_state_s
{
    uint32_t magic;
    spangpu_sigtone_tx_t *bank;
    span_tone_report_func_t sig_update;
    void *user_data;
};

/* ---- receiver --------------------------------------------------------------------------------------------- */

sig_tone_rx_state_t *sig_tone_rx_init(sig_tone_rx_state_t *s, int tone_type, span_tone_report_func_t sig_update, void *user_data)
{
    const int fresh = (s == NULL);

    /* sig_tone.c:679-680: no callback or no such tone type -> NULL.  Caller storage cannot hold state that lives in HBM;
       an object made here is re-initialised in place, as the reference re-initialises whatever it is handed. */
    if (sig_update == NULL  ||  tone_type < 1  ||  tone_type > 3)
        return NULL;
    if (!fresh)
    {
        if (s->magic != RX_MAGIC)
            return NULL;
        spangpu_sigtone_rx_destroy(s->bank);
        free(s->keep);
        memset(s, 0, sizeof(*s));
    }
    else if ((s = (sig_tone_rx_state_t *) calloc(1, sizeof(*s))) == NULL)
    {
        return NULL;
    }
    if (spangpu_sigtone_rx_create(&s->bank, 0, tone_type, 1) != SPANGPU_OK)
    {
        if (fresh)
            free(s);
        return NULL;
    }
    s->magic = RX_MAGIC;
    s->sig_update = sig_update;
    s->user_data = user_data;
    return s;
}

void sig_tone_rx_set_mode(sig_tone_rx_state_t *s, int mode, int duration)
{
    (void) duration;            /* unused by the reference too (sig_tone.c:666-669) */
    if (s == NULL)
        return;
    if (s->in_callback)
    {
        s->mode_set = 1;
        s->new_mode = mode;
    }
    (void) spangpu_sigtone_rx_set_mode(s->bank, 0, mode);
}

/* at most 1024 samples: two reports are 24 samples apart at the least, so 64 slots hold what one call can report */
static int rx_slice(sig_tone_rx_state_t *s, int16_t amp[], int len)
{
    int32_t before[RX_WORDS];
    const int32_t *ev;
    const int32_t *cnt;
    int32_t at[64];
    int n;
    int i;
    int pos;

    if (s == NULL  ||  amp == NULL  ||  len <= 0)
        return 0;
    if (spangpu_sigtone_rx_get_state(s->bank, 0, before) != SPANGPU_OK)
        return 0;
    if (len > s->keep_cap)
    {
        int16_t *k = (int16_t *) realloc(s->keep, sizeof(int16_t)*(size_t) len);

        if (k == NULL)
            return 0;
        s->keep = k;
        s->keep_cap = len;
    }
    memcpy(s->keep, amp, sizeof(int16_t)*(size_t) len);
    if (spangpu_sigtone_rx(s->bank, amp, SPANGPU_MEM_HOST, len, len) != SPANGPU_OK)
        return 0;
    if (spangpu_sigtone_rx_events(s->bank, &ev, &cnt) < 0)
        return 0;
    n = cnt[0];
    if (n == 0)
        return len;
    /* Something was reported: again from the state before (which holds the mode the frame began with), in pieces that
       end in front of the reports -- the mode plays no part in detection, so the reports are where the first run found
       them.  The reference calls back in the middle of a sample: after the detectors, before the media path writes that
       sample (sig_tone.c:627-653), so a mode set in the callback already applies to it.  Hence the sample of a report
       is run on its own, and run again with the new mode if its callback set one. */
    if (n > 64)
        n = 64;
    for (i = 0;  i < n;  i++)
        at[i] = ev[3*i];
    if (spangpu_sigtone_rx_set_state(s->bank, 0, before) != SPANGPU_OK)
        return 0;
    memcpy(amp, s->keep, sizeof(int16_t)*(size_t) len);
    pos = 0;
    for (i = 0;  i <= n;  i++)
    {
        const int upto = (i < n)  ?  at[i]  :  len;
        int32_t here[RX_WORDS];
        int k;

        if (upto > pos)
        {
            if (spangpu_sigtone_rx(s->bank, amp + pos, SPANGPU_MEM_HOST, upto - pos, upto - pos) != SPANGPU_OK)
                return pos;
            pos = upto;
        }
        if (i == n)
            break;
        if (spangpu_sigtone_rx_get_state(s->bank, 0, here) != SPANGPU_OK
            ||  spangpu_sigtone_rx(s->bank, amp + pos, SPANGPU_MEM_HOST, 1, 1) != SPANGPU_OK
            ||  spangpu_sigtone_rx_events(s->bank, &ev, &cnt) < 0)
        {
            return pos;
        }
        s->mode_set = 0;
        s->in_callback = 1;
        for (k = 0;  k < cnt[0];  k++)
            s->sig_update(s->user_data, ev[3*k + 1], 0, ev[3*k + 2]);
        s->in_callback = 0;
        if (s->mode_set)
        {
            here[RX_WORDS - 1] = s->new_mode;
            amp[pos] = s->keep[pos];
            if (spangpu_sigtone_rx_set_state(s->bank, 0, here) != SPANGPU_OK
                ||  spangpu_sigtone_rx(s->bank, amp + pos, SPANGPU_MEM_HOST, 1, 1) != SPANGPU_OK)
            {
                return pos;
            }
        }
        pos++;
    }
    return len;
}

int sig_tone_rx(sig_tone_rx_state_t *s, int16_t amp[], int len)
{
    int done = 0;

    if (s == NULL  ||  amp == NULL  ||  len <= 0)
        return 0;
    while (done < len)
    {
        const int m = (len - done > 1024)  ?  1024  :  (len - done);
        const int got = rx_slice(s, amp + done, m);

        done += got;
        if (got < m)
            break;
    }
    return done;
}

int sig_tone_rx_release(sig_tone_rx_state_t *s)
{
    (void) s;
    return 0;
}

int sig_tone_rx_free(sig_tone_rx_state_t *s)
{
    if (s)
    {
        spangpu_sigtone_rx_destroy(s->bank);
        free(s->keep);
        s->magic = 0;
        free(s);
    }
    return 0;
}

/* ---- sender ----------------------------------------------------------------------------------------------- */

sig_tone_tx_state_t *sig_tone_tx_init(sig_tone_tx_state_t *s, int tone_type, span_tone_report_func_t sig_update, void *user_data)
{
    const int fresh = (s == NULL);

    /* sig_tone.c:352-353 */
    if (sig_update == NULL  ||  tone_type < 1  ||  tone_type > 3)
        return NULL;
    if (!fresh)
    {
        if (s->magic != TX_MAGIC)
            return NULL;
        spangpu_sigtone_tx_destroy(s->bank);
        memset(s, 0, sizeof(*s));
    }
    else if ((s = (sig_tone_tx_state_t *) calloc(1, sizeof(*s))) == NULL)
    {
        return NULL;
    }
    if (spangpu_sigtone_tx_create(&s->bank, 0, tone_type, 1) != SPANGPU_OK)
    {
        if (fresh)
            free(s);
        return NULL;
    }
    s->magic = TX_MAGIC;
    s->sig_update = sig_update;
    s->user_data = user_data;
    return s;
}

void sig_tone_tx_set_mode(sig_tone_tx_state_t *s, int mode, int duration)
{
    if (s  &&  mode >= 0)
        (void) spangpu_sigtone_tx_set_mode(s->bank, 0, mode, duration);
}

int sig_tone_tx(sig_tone_tx_state_t *s, int16_t amp[], int len)
{
    int pending;
    int rounds = 0;

    if (s == NULL  ||  amp == NULL  ||  len <= 0)
        return 0;
    pending = spangpu_sigtone_tx(s->bank, amp, SPANGPU_MEM_HOST, len, len);
    while (pending > 0  &&  rounds++ <= len)
    {
        /* sig_tone.c:316-318: the update request, in which the caller sets what comes next */
        s->sig_update(s->user_data, SPANGPU_SIG_TONE_TX_UPDATE_REQUEST, 0, 0);
        pending = spangpu_sigtone_tx_continue(s->bank, amp, SPANGPU_MEM_HOST, len);
    }
    return (pending < 0)  ?  0  :  len;
}

int sig_tone_tx_release(sig_tone_tx_state_t *s)
{
    (void) s;
    return 0;
}

int sig_tone_tx_free(sig_tone_tx_state_t *s)
{
    if (s)
    {
        spangpu_sigtone_tx_destroy(s->bank);
        s->magic = 0;
        free(s);
    }
    return 0;
}
