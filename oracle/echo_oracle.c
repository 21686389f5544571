/*
 * echo_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's G.168 line echo canceller:
 *   echo_can_init / _flush / _adaption_mode   src/echo.c:254-372
 *   echo_can_hpf, echo_can_hpf_tx             src/echo.c:382-419, :663-669
 *   echo_can_update                           src/echo.c:421-661
 *   lms_adapt                                 src/echo.c:177-251
 *   narrowband_detect                         src/echo.c:120-175
 *   fir16()                                   src/spandsp/fir.h:121-183 (scalar path :168-182)
 *
 * All arithmetic is 32-bit two's complement with wrap-around (built -fwrapv, as the
 * reference build in oracle/_ref is) except the 32x9 float autocorrelation of
 * narrowband_detect, which is evaluated in the reference's order, unfused.
 *
 * Two behaviours of this reference snapshot that look like accidents are part of what it
 * computes, and are restated as such (tests/test_oracle_pin.py shows the real build does
 * exactly this):
 *  (1) echo.c:505,566 index fir_taps16[(tap_set - 1)%3].  With tap_set == 0 that is
 *      fir_taps16[-1], which on LP64 is the `history` pointer of the fir16_state_t that
 *      precedes the array in echo_can_state_t (private/echo.h): the "revert" then copies a
 *      tap set over the FIR *history*.
 *  (2) narrowband_detect (echo.c:133-139) walks 32 history samples wrapping at a
 *      hard-coded 256, whatever `taps` is.  For taps < 256 it reads past the history
 *      allocation.  The pin tests give the reference a zero-filled arena through its own
 *      allocator hook (span_mem_allocators, alloc.c:142), so those reads return 0; this
 *      oracle (and the GPU engine) define history[k] = 0 for taps <= k < 256.
 * The `vad` field (echo.c:579-582) is computed here but is not observable through the
 * reference's API.
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define NONUPDATE_DWELL_TIME        600             /* echo.c:115 */
#define MIN_TX_POWER_FOR_ADAPTION   (64*64)         /* echo.c:117 */

int orc_echo_sizeof(void) { return (int) sizeof(orc_echo_t); }

/* echo.c:254-301.  taps must be <= ORC_ECHO_MAX_TAPS. */
int orc_echo_init(orc_echo_t *ec, int taps, int adaption_mode)
{
    if (taps <= 0  ||  taps > ORC_ECHO_MAX_TAPS)
        return -1;
    memset(ec, 0, sizeof(*ec));
    ec->taps = taps;
    ec->curr_pos = taps - 1;
    ec->tap_mask = taps - 1;
    ec->fir_curr_pos = taps - 1;
    ec->rx_power_threshold = 10000000;
    ec->tap_rotate_counter = 1600;
    ec->cng_level = 1000;
    ec->adaption_mode = adaption_mode;
    return 0;
}

void orc_echo_adaption_mode(orc_echo_t *ec, int adaption_mode)
{
    ec->adaption_mode = adaption_mode;
}

/* echo.c:331-372 */
void orc_echo_flush(orc_echo_t *ec)
{
    memset(ec->tx_power, 0, sizeof(ec->tx_power));
    memset(ec->rx_power, 0, sizeof(ec->rx_power));
    ec->clean_rx_power = 0;
    ec->nonupdate_dwell = 0;
    memset(ec->history, 0, sizeof(ec->history));
    ec->fir_curr_pos = ec->taps - 1;
    memset(ec->taps32, 0, sizeof(ec->taps32));
    memset(ec->taps16, 0, sizeof(ec->taps16));
    ec->curr_pos = ec->taps - 1;
    ec->supp_test1 = 0;
    ec->supp_test2 = 0;
    ec->supp1 = 0;
    ec->supp2 = 0;
    ec->vad = 0;
    ec->cng_level = 1000;
    ec->cng_filter = 0;
    ec->geigel_max = 0;
    ec->geigel_lag = 0;
    ec->dtd_onset = 0;
    ec->tap_set = 0;
    ec->tap_rotate_counter = 1600;
    ec->latest_correction = 0;
    memset(ec->last_acf, 0, sizeof(ec->last_acf));
    ec->narrowband_count = 0;
    ec->narrowband_score = 0;
}

/* echo.c:382-419 */
static int16_t hpf(int32_t coeff[2], int16_t amp)
{
    int32_t z;

    z = (int32_t) ((uint32_t) (int32_t) amp << 15);
    z -= (z >> 4);
    coeff[0] += z - (coeff[0] >> 3) - coeff[1];
    coeff[1] = z;
    z = coeff[0] >> 15;
    /* saturate16(), saturated.h:45-63 */
    if (z != (int16_t) z)
        z = (z > 32767)  ?  32767  :  -32768;
    return (int16_t) z;
}

/* echo.c:663-669 */
int16_t orc_echo_hpf_tx(orc_echo_t *ec, int16_t tx)
{
    if (ec->adaption_mode & ORC_ECHO_USE_TX_HPF)
        tx = hpf(ec->tx_hpf, tx);
    return tx;
}

/* bit_operations.h:45-140 */
static int top_bit(uint32_t bits)
{
    int res;

    if (bits == 0)
        return -1;
    res = 0;
    while (bits >>= 1)
        res++;
    return res;
}

/* float -> int32 the way the reference build does it on x86-64 (cvttss2si): out-of-range
   and NaN give INT32_MIN */
static int32_t f2i(float v)
{
    if (!(v < 2147483648.0f)  ||  !(v >= -2147483648.0f))
        return (int32_t) 0x80000000u;
    return (int32_t) v;
}

/* The destination of fir_taps16[idx] for idx in {-1, 0, 1, 2, 3}: -1 is the FIR history
   (see the header comment). */
static int16_t *tap_dest(orc_echo_t *ec, int idx)
{
    return (idx < 0)  ?  ec->history  :  ec->taps16[idx];
}

/* echo.c:120-175 */
static int narrowband_detect(orc_echo_t *ec)
{
    float sf[32];
    float f_acf[9];
    int32_t acf[9];
    float temp;
    float scale;
    int score;
    int i;
    int k;

    k = ec->curr_pos;
    for (i = 0;  i < 32;  i++)
    {
        sf[i] = (k < ec->taps)  ?  ec->history[k]  :  0;     /* history[k] beyond taps reads 0 */
        k++;
        if (k >= 256)
            k = 0;
    }
    for (k = 0;  k < 9;  k++)
    {
        temp = 0;
        for (i = k;  i < 32;  i++)
            temp += sf[i]*sf[i - k];
        f_acf[k] = temp;
    }
    scale = 0x1FFFFFFF/f_acf[0];
    for (k = 0;  k < 9;  k++)
        acf[k] = f2i(f_acf[k]*scale);
    score = 0;
    for (i = 0;  i < 9;  i++)
    {
        if (ec->last_acf[i] >= 0  &&  acf[i] >= 0)
        {
            if ((ec->last_acf[i] >> 1) < acf[i]  &&  acf[i] < (int32_t) ((uint32_t) ec->last_acf[i] << 1))
                score++;
        }
        else if (ec->last_acf[i] < 0  &&  acf[i] < 0)
        {
            if ((ec->last_acf[i] >> 1) > acf[i]  &&  acf[i] > (int32_t) ((uint32_t) ec->last_acf[i] << 1))
                score++;
        }
    }
    memcpy(ec->last_acf, acf, sizeof(acf));
    return score;
}

/* echo.c:232-249: taps32[i] += history[(i + curr_pos) mod taps]*factor */
static void lms_adapt(orc_echo_t *ec, int factor)
{
    int16_t *t16 = ec->taps16[ec->tap_set];
    int i;
    int j;

    for (i = 0;  i < ec->taps;  i++)
    {
        j = i + ec->curr_pos;
        if (j >= ec->taps)
            j -= ec->taps;
        ec->taps32[i] += ec->history[j]*factor;
        t16[i] = (int16_t) (ec->taps32[i] >> 15);
    }
}

/* echo.c:421-661 */
int16_t orc_echo_update(orc_echo_t *ec, int16_t tx, int16_t rx)
{
    const int16_t *coeffs;
    int32_t y;
    int32_t echo_value;
    int clean_rx;
    int nsuppr;
    int score;
    int i;
    int j;
    int src;

    if (ec->adaption_mode & ORC_ECHO_USE_RX_HPF)
        rx = hpf(ec->rx_hpf, rx);
    ec->latest_correction = 0;

    /* fir16(), fir.h:168-183.  fir_state.coeffs follows tap_set only at a rotation (echo.c:526);
       echo_can_flush() resets tap_set but not the pointer, hence the separate fir_set. */
    coeffs = ec->taps16[ec->fir_set];
    ec->history[ec->fir_curr_pos] = tx;
    y = 0;
    for (i = 0;  i < ec->taps;  i++)
    {
        j = i + ec->fir_curr_pos;
        if (j >= ec->taps)
            j -= ec->taps;
        y += coeffs[i]*ec->history[j];
    }
    if (ec->fir_curr_pos <= 0)
        ec->fir_curr_pos = ec->taps;
    ec->fir_curr_pos--;
    echo_value = (int16_t) (y >> 15);

    clean_rx = rx - echo_value;
    if (ec->nonupdate_dwell > 0)
        ec->nonupdate_dwell--;

    /* echo.c:463-469 */
    ec->tx_power[3] += ((abs(tx) - ec->tx_power[3]) >> 5);
    ec->tx_power[2] += ((tx*tx - ec->tx_power[2]) >> 8);
    ec->tx_power[1] += ((tx*tx - ec->tx_power[1]) >> 5);
    ec->tx_power[0] += ((tx*tx - ec->tx_power[0]) >> 3);
    ec->rx_power[1] += ((rx*rx - ec->rx_power[1]) >> 6);
    ec->rx_power[0] += ((rx*rx - ec->rx_power[0]) >> 3);
    ec->clean_rx_power += ((clean_rx*clean_rx - ec->clean_rx_power) >> 6);

    score = 0;
    if (ec->tx_power[0] > MIN_TX_POWER_FOR_ADAPTION)
    {
        if (ec->tx_power[1] > ec->rx_power[0])
        {
            if (ec->nonupdate_dwell == 0)
            {
                if (++ec->narrowband_count >= 160)
                {
                    ec->narrowband_count = 0;
                    score = narrowband_detect(ec);
                    if (score > 6)
                    {
                        if (ec->narrowband_score == 0)
                            memcpy(ec->taps16[3], ec->taps16[(ec->tap_set + 1)%3], ec->taps*sizeof(int16_t));
                        ec->narrowband_score += score;
                    }
                    else
                    {
                        if (ec->narrowband_score > 200)
                        {
                            /* echo.c:504-508 */
                            memcpy(ec->taps16[ec->tap_set], ec->taps16[3], ec->taps*sizeof(int16_t));
                            memcpy(tap_dest(ec, (ec->tap_set - 1)%3), ec->taps16[3], ec->taps*sizeof(int16_t));
                            for (i = 0;  i < ec->taps;  i++)
                                ec->taps32[i] = (int32_t) ((uint32_t) (int32_t) ec->taps16[3][i] << 15);
                            ec->tap_rotate_counter = 1600;
                        }
                        ec->narrowband_score = 0;
                    }
                }
                ec->dtd_onset = 0;
                if (--ec->tap_rotate_counter <= 0)
                {
                    ec->tap_rotate_counter = 1600;
                    ec->tap_set++;
                    if (ec->tap_set > 2)
                        ec->tap_set = 0;
                    ec->fir_set = ec->tap_set;
                }
                if ((ec->adaption_mode & ORC_ECHO_USE_ADAPTION)  &&  ec->narrowband_score == 0)
                {
                    nsuppr = clean_rx;
                    if (tx > 4*ec->tx_power[3])
                        i = top_bit((uint32_t) (int32_t) tx) - 8;
                    else
                        i = top_bit((uint32_t) ec->tx_power[3]) - 8;
                    if (i > 0)
                        nsuppr >>= i;
                    lms_adapt(ec, nsuppr);
                }
            }
        }
        else
        {
            if (!ec->dtd_onset)
            {
                /* echo.c:565-570.  The source set is read before either copy can alias it. */
                src = (ec->tap_set + 1)%3;
                memcpy(ec->taps16[ec->tap_set], ec->taps16[src], ec->taps*sizeof(int16_t));
                memcpy(tap_dest(ec, (ec->tap_set - 1)%3), ec->taps16[src], ec->taps*sizeof(int16_t));
                for (i = 0;  i < ec->taps;  i++)
                    ec->taps32[i] = (int32_t) ((uint32_t) (int32_t) ec->taps16[src][i] << 15);
                ec->tap_rotate_counter = 1600;
                ec->dtd_onset = 1;
            }
            ec->nonupdate_dwell = NONUPDATE_DWELL_TIME;
        }
    }

    /* echo.c:579-591 */
    if (ec->rx_power[1])
        ec->vad = (8000*ec->clean_rx_power)/ec->rx_power[1];
    else
        ec->vad = 0;
    if (ec->rx_power[1] > 2048*2048  &&  ec->clean_rx_power > 4*ec->rx_power[1])
    {
        memset(ec->taps32, 0, ec->taps*sizeof(int32_t));
        for (i = 0;  i < 4;  i++)
            memset(ec->taps16[i], 0, ec->taps*sizeof(int16_t));
    }

    /* echo.c:613-651 */
    if ((ec->adaption_mode & ORC_ECHO_USE_NLP))
    {
        if (ec->rx_power[1] < 30000000)
        {
            if (!ec->cng)
            {
                ec->cng_level = ec->clean_rx_power;
                ec->cng = 1;
            }
            if ((ec->adaption_mode & ORC_ECHO_USE_CNG))
            {
                ec->cng_rndnum = (int) (1664525U*(unsigned int) ec->cng_rndnum + 1013904223U);
                ec->cng_filter = ((ec->cng_rndnum & 0xFFFF) - 32768 + 5*ec->cng_filter) >> 3;
                clean_rx = (ec->cng_filter*ec->cng_level) >> 17;
            }
            else
            {
                clean_rx = 0;
            }
        }
        else
        {
            ec->cng = 0;
        }
    }
    else
    {
        ec->cng = 0;
    }

    /* echo.c:655-658 */
    if (ec->curr_pos <= 0)
        ec->curr_pos = ec->taps;
    ec->curr_pos--;
    return (int16_t) clean_rx;
}

void orc_echo_run(orc_echo_t *ec, const int16_t tx[], const int16_t rx[], int16_t clean[], int n, int use_hpf_tx)
{
    int i;
    int16_t t;

    for (i = 0;  i < n;  i++)
    {
        t = tx[i];
        if (use_hpf_tx)
            t = orc_echo_hpf_tx(ec, t);
        clean[i] = orc_echo_update(ec, t, rx[i]);
    }
}

/* channel c: state s[c], tx + c*stride, rx + c*stride, clean + c*stride */
void orc_echo_run_batch(orc_echo_t *s, const int16_t tx[], const int16_t rx[], int16_t clean[], int n_ch, long long stride, int n, int use_hpf_tx)
{
    int c;

    for (c = 0;  c < n_ch;  c++)
        orc_echo_run(&s[c], tx + c*stride, rx + c*stride, clean + c*stride, n, use_hpf_tx);
}
