/*
 * shim_echo.c -- host side (plain C) of the spandsp-named echo canceller entry points declared in
 * include/spangpu_spandsp.h (reference: src/spandsp/echo.h:145-185, src/echo.c:254-380,421-669).  An object made by
 * echo_can_init() is a private one-channel bank: echo_can_update() is then one kernel launch per SAMPLE -- the plumbing
 * configuration, there for source compatibility; a media loop that wants the GPU's throughput hands whole frames of many
 * channels to spangpu_echo_update() / spangpu_echo_can_update_block().  No arithmetic of the canceller happens here.
 */
#include <stdlib.h>
#include <string.h>

#include "spangpu_spandsp.h"

struct echo_can_state_s
{
    spangpu_echo_t *bank;
    int taps;
    int16_t *snapshot;          /* tap set 0 as echo_can_snapshot() last saw it */
};

echo_can_state_t *echo_can_init(int len, int adaption_mode)
{
    echo_can_state_t *ec;

    if ((ec = (echo_can_state_t *) calloc(1, sizeof(*ec))) == NULL)
        return NULL;
    ec->taps = len;
    if ((ec->snapshot = (int16_t *) calloc((size_t) (len > 0  ?  len  :  1), sizeof(int16_t))) == NULL
        ||
        spangpu_echo_create(&ec->bank, 0, 1, len, adaption_mode) != SPANGPU_OK)
    {
        free(ec->snapshot);
        free(ec);
        return NULL;
    }
    return ec;
}

int echo_can_release(echo_can_state_t *ec)
{
    (void) ec;
    return 0;
}

int echo_can_free(echo_can_state_t *ec)
{
    if (ec)
    {
        spangpu_echo_destroy(ec->bank);
        free(ec->snapshot);
        free(ec);
    }
    return 0;
}

void echo_can_flush(echo_can_state_t *ec)
{
    spangpu_echo_flush(ec->bank, 0);
}

void echo_can_adaption_mode(echo_can_state_t *ec, int adaption_mode)
{
    spangpu_echo_adaption_mode(ec->bank, 0, adaption_mode);
}

/* src/echo.c:376-379: the working tap set (set 0) is copied aside.  The copy lives on the host: the four 16 bit sets of
   the one channel come back in one transfer and the first is kept. */
void echo_can_snapshot(echo_can_state_t *ec)
{
    int16_t *sets;

    if ((sets = (int16_t *) malloc((size_t) 4*ec->taps*sizeof(int16_t))) == NULL)
        return;
    if (spangpu_echo_get_state(ec->bank, 0, NULL, NULL, sets, NULL) == SPANGPU_OK)
        memcpy(ec->snapshot, sets, (size_t) ec->taps*sizeof(int16_t));
    free(sets);
}

int spangpu_echo_can_snapshot_taps(echo_can_state_t *ec, int16_t *out, int max)
{
    int n;

    if (ec == NULL  ||  out == NULL  ||  max < 0)
        return -1;
    n = (max < ec->taps)  ?  max  :  ec->taps;
    memcpy(out, ec->snapshot, (size_t) n*sizeof(int16_t));
    return n;
}

int16_t echo_can_update(echo_can_state_t *ec, int16_t tx, int16_t rx)
{
    int16_t clean = 0;

    spangpu_echo_update(ec->bank, &tx, &rx, &clean, SPANGPU_MEM_HOST, 1, 1, 0);
    return clean;
}

int16_t echo_can_hpf_tx(echo_can_state_t *ec, int16_t tx)
{
    int16_t out = tx;

    spangpu_echo_hpf_tx(ec->bank, &tx, &out, 1, 1);
    return out;
}

/* n samples in one launch: clean[i] = echo_can_update(ec, use_hpf_tx ? echo_can_hpf_tx(ec, tx[i]) : tx[i], rx[i]);
   tx_out (may be NULL) receives the samples the canceller saw on the transmit side. */
int spangpu_echo_can_update_block(echo_can_state_t *ec, const int16_t tx[], const int16_t rx[], int16_t clean[], int16_t tx_out[],
                                  int n, int use_hpf_tx)
{
    if (n <= 0)
        return 0;
    return spangpu_echo_update_tx(ec->bank, tx, rx, clean, tx_out, SPANGPU_MEM_HOST, n, n, use_hpf_tx);
}

spangpu_echo_t *spangpu_echo_can_bank(echo_can_state_t *ec)
{
    return ec  ?  ec->bank  :  NULL;
}
