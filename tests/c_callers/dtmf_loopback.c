/* A C caller of the spandsp-named entry points (include/spangpu_spandsp.h), compiled -std=c99 -pedantic -Werror and as C++
 * against the headers alone and linked with -lspangpu: "123A456B789C*0#D" through dtmf_tx() and dtmf_rx() in 160-sample
 * frames, as tests/dtmf_rx_tests.c's plumbing case does with the reference.  Own code; exits 0 on success. */
#include <stdio.h>
#include <string.h>

#include "spangpu_spandsp.h"

static int calls = 0;
static char seen[64];

static void on_digits(void *user_data, const char *digits, int len)
{
    (void) user_data;
    if (strlen(seen) + (size_t) len < sizeof(seen))
        strncat(seen, digits, (size_t) len);
    calls++;
}

int main(void)
{
    static const char want[] = "123A456B789C*0#D";
    int16_t amp[160];
    char got[129];
    dtmf_tx_state_t *tx;
    dtmf_rx_state_t *rx;
    size_t n;
    int len;
    int frames = 0;

    if ((tx = dtmf_tx_init(NULL, NULL, NULL)) == NULL  ||  (rx = dtmf_rx_init(NULL, NULL, NULL)) == NULL)
    {
        fprintf(stderr, "init failed: %s\n", spangpu_last_error());
        return 2;
    }
    if (dtmf_tx_put(tx, want, -1) != 0)
        return 3;
    while ((len = dtmf_tx(tx, amp, 160)) > 0)
    {
        if (dtmf_rx(rx, amp, len) != 0)
            return 4;
        frames++;
    }
    memset(amp, 0, sizeof(amp));
    dtmf_rx(rx, amp, 160);
    n = dtmf_rx_get(rx, got, 128);
    got[n] = '\0';
    printf("dtmf_loopback: %d frames, got \"%s\"\n", frames, got);
    if (strcmp(got, want) != 0)
        return 1;
    /* ... and once more through the digits callback */
    dtmf_rx_free(rx);
    if ((rx = dtmf_rx_init(NULL, on_digits, NULL)) == NULL)
        return 5;
    dtmf_tx_put(tx, "42", -1);
    while ((len = dtmf_tx(tx, amp, 160)) > 0)
        dtmf_rx(rx, amp, len);
    memset(amp, 0, sizeof(amp));
    dtmf_rx(rx, amp, 160);
    printf("dtmf_loopback: callback saw \"%s\" in %d calls\n", seen, calls);
    dtmf_rx_free(rx);
    dtmf_tx_free(tx);
    return (strcmp(seen, "42") == 0)  ?  0  :  6;
}
