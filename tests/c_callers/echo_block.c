/* A C caller: echo_can_init() and the block form of echo_can_update() (spangpu_echo_can_update_block(): one object, n samples a
 * call -- the by-name echo_can_update() is a launch per sample, see the header) on a synthetic line: white noise out, an echo
 * through a short FIR path 12 dB down, nobody talking at the near end.  After four seconds the echo return loss enhancement
 * over the last second must be better than 30 dB.  Own code; exits 0 on success. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spangpu_spandsp.h"

#define FRAME 160

static unsigned lcg = 12345u;

static int noise(void)
{
    int i;
    int acc = 0;

    for (i = 0;  i < 4;  i++)
    {
        lcg = lcg*1664525u + 1013904223u;
        acc += (int) ((lcg >> 16) & 0x3FFF) - 0x2000;
    }
    return acc/4;               /* roughly Gaussian, some -15 dBm0 */
}

int main(void)
{
    static const double path[8] = {0.0, 0.12, -0.20, 0.08, 0.05, -0.03, 0.01, 0.005};
    int16_t tx[FRAME];
    int16_t rx[FRAME];
    int16_t clean[FRAME];
    double hist[8];
    double e_rx = 0.0;
    double e_clean = 0.0;
    echo_can_state_t *ec;
    int frame;
    int i;
    int k;

    memset(hist, 0, sizeof(hist));
    if ((ec = echo_can_init(128, ECHO_CAN_USE_ADAPTION)) == NULL)
    {
        fprintf(stderr, "init failed: %s\n", spangpu_last_error());
        return 2;
    }
    for (frame = 0;  frame < 250;  frame++)     /* 5 s */
    {
        for (i = 0;  i < FRAME;  i++)
        {
            double echo = 0.0;

            tx[i] = (int16_t) noise();
            for (k = 7;  k > 0;  k--)
                hist[k] = hist[k - 1];
            hist[0] = tx[i];
            for (k = 0;  k < 8;  k++)
                echo += path[k]*hist[k];
            rx[i] = (int16_t) echo;
        }
        if (spangpu_echo_can_update_block(ec, tx, rx, clean, NULL, FRAME, 0) < 0)
        {
            fprintf(stderr, "update failed: %s\n", spangpu_last_error());
            return 3;
        }
        if (frame >= 200)
        {
            for (i = 0;  i < FRAME;  i++)
            {
                e_rx += (double) rx[i]*rx[i];
                e_clean += (double) clean[i]*clean[i];
            }
        }
    }
    echo_can_free(ec);
    printf("echo_block: ERLE over the last second %.1f dB\n", 10.0*log10(e_rx/(e_clean + 1.0e-9)));
    return (e_rx > 30.0*30.0*FRAME  &&  e_rx > 1000.0*e_clean)  ?  0  :  1;
}
