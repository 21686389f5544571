/* A C caller: a V.29 9600 bps transmission (a one-channel spangpu_modemtx bank: the reference's modulator, bit-exact, include/
 * spangpu.h) through v29_rx() with a put_bit callback in 160-sample frames -- the receiver trains, reports it, and delivers the
 * transmitter's bit stream (a 15-bit LFSR) without an error.  Own code; exits 0 on success. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spangpu_spandsp.h"

#define MAX_BITS 40000

typedef struct
{
    int n_bits;
    int trained_at;
    int failed;
    unsigned char bits[MAX_BITS];
} page_t;

static void put_bit(void *user_data, int bit)
{
    page_t *p = (page_t *) user_data;

    if (bit < 0)
    {
        if (bit == SIG_STATUS_TRAINING_SUCCEEDED)
            p->trained_at = p->n_bits;
        else if (bit == SIG_STATUS_TRAINING_FAILED)
            p->failed = 1;
        return;
    }
    if (p->n_bits < MAX_BITS)
        p->bits[p->n_bits++] = (unsigned char) bit;
}

int main(void)
{
    static page_t page;
    spangpu_modemtx_t *tx = NULL;
    v29_rx_state_t *rx;
    uint32_t seed = 0x1234;
    int16_t amp[160];
    int frame;
    int align;
    int i;
    unsigned st;

    page.trained_at = -1;
    if (spangpu_modemtx_create(&tx, 0, SPANGPU_V29, 1, 9600, 0, &seed) != SPANGPU_OK
        ||  (rx = v29_rx_init(NULL, 9600, put_bit, &page)) == NULL)
    {
        fprintf(stderr, "init failed: %s\n", spangpu_last_error());
        return 2;
    }
    for (frame = 0;  frame < 150;  frame++)     /* 3 s */
    {
        if (spangpu_modemtx_tx(tx, SPANGPU_MEM_HOST, amp, 160, 160) != 160)
            return 3;
        if (v29_rx(rx, amp, 160) != 0)
            return 4;
    }
    printf("v29_page: trained after %d bits, %d data bits, carrier %.2f Hz\n", page.trained_at, page.n_bits, v29_rx_carrier_frequency(rx));
    if (page.failed  ||  page.trained_at < 0  ||  page.n_bits < 20000)
        return 1;
    /* the transmitter's bit source: x^15 + x^14 + 1 from the seed; the receiver joins somewhere in it */
    for (align = 0;  align < 400;  align++)
    {
        int ok = 1;

        st = seed & 0x7FFF;
        for (i = 0;  i < align;  i++)
            st = ((st << 1) | (((st >> 14) ^ (st >> 13)) & 1)) & 0x7FFF;
        for (i = page.trained_at;  i < page.n_bits  &&  ok;  i++)
        {
            const unsigned b = ((st >> 14) ^ (st >> 13)) & 1;
            st = ((st << 1) | b) & 0x7FFF;
            ok = (page.bits[i] == b);
        }
        if (ok)
        {
            printf("v29_page: %d bits equal to the transmitter's from offset %d\n", page.n_bits - page.trained_at, align);
            v29_rx_free(rx);
            spangpu_modemtx_destroy(tx);
            return 0;
        }
    }
    return 5;
}
