/*
 * v17tx_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's V.17 transmitter, the signal source of the V.17 receiver tests
 * (SURVEY.md section 8(f)-1):
 *
 *   v17_tx_init / v17_tx_restart / v17_tx_power      src/v17tx.c:371-483
 *   scramble, training_get, diff_and_convolutional_encode, getbaud     src/v17tx.c:106-293
 *   v17_tx                                           src/v17tx.c:295-369   (float build)
 *
 * Data bits come from the 15 bit LFSR the test glue feeds the reference with (x^15 + x^14 + 1); the
 * end-of-data / shutdown path is therefore never taken.  The pulse shaper is the 10 x 9 table of the V.29
 * transmitter (make_modem_filter gives V.17 the same parameters; orc_v29_tx_set_table()), the constellations
 * and the sine table come through orc_modem_set_tables().
 */
#include <math.h>
#include <string.h>

#include "oracle.h"
#include "modem_common.h"

enum
{
    TEP_B = 480, SEG_1 = 480 + 48, SEG_2 = 480 + 48 + 256, SEG_3 = 480 + 48 + 256 + 2976, SEG_4 = 480 + 48 + 256 + 2976 + 64,
    SHORT_SEG_4 = 480 + 48 + 256 + 38, TRAINING_END = 480 + 48 + 256 + 2976 + 64 + 48,
    SHUTDOWN_A = 480 + 48 + 256 + 2976 + 64 + 48 + 32, SHUTDOWN_END = 480 + 48 + 256 + 2976 + 64 + 48 + 32 + 48
};

extern const float *orc_v29_tx_shaper(void);

static const float abcd[4][2] = {{-6.0f, -2.0f}, {2.0f, -6.0f}, {6.0f, 2.0f}, {-2.0f, 6.0f}};

static const float (*points_of(int bit_rate))[2]
{
    static const int offset[5] = {0, 128, 192, 224, 240};
    int k;

    switch (bit_rate)
    {
    case 14400: k = 0; break;
    case 12000: k = 1; break;
    case 9600: k = 2; break;
    case 7200: k = 3; break;
    default: k = 4; break;
    }
    return (const float (*)[2]) (orc_modem_T.v17_constellation + 2*offset[k]);
}

ORC_API void orc_v17_tx_power(orc_v17_tx_t *s, float power)
{
    /* v17tx.c:371-383; TX_PULSESHAPER_GAIN = 1.0f in the float build */
    s->gain = 0.223f*powf(10.0f, (power - 3.14f)/20.0f)*32768.0f/1.000000f;
}

ORC_API int orc_v17_tx_restart(orc_v17_tx_t *s, int bit_rate, int tep, int short_train)
{
    if (bit_rate != 14400  &&  bit_rate != 12000  &&  bit_rate != 9600  &&  bit_rate != 7200  &&  bit_rate != 4800)
        return -1;
    s->bit_rate = bit_rate;
    s->diff = short_train  ?  0  :  1;
    memset(s->rrc_re, 0, sizeof(s->rrc_re));
    memset(s->rrc_im, 0, sizeof(s->rrc_im));
    s->rrc_step = 0;
    s->convolution = 0;
    s->scramble_reg = 0x2ECDD5;
    s->in_training = 1;
    s->short_train = short_train;
    s->training_step = tep  ?  0  :  SEG_1;
    s->carrier_phase = 0;
    s->baud_phase = 0;
    s->constellation_state = 0;
    return 0;
}

ORC_API int orc_v17_tx_init(orc_v17_tx_t *s, int bit_rate, int tep, uint32_t prbs_seed)
{
    memset(s, 0, sizeof(*s));
    s->prbs = prbs_seed & 0x7FFF;
    s->carrier_phase_rate = (int32_t) (1800.0f*65536.0f*65536.0f/8000);
    orc_v17_tx_power(s, -14.0f);
    return orc_v17_tx_restart(s, bit_rate, tep, 0);
}

static int scramble(orc_v17_tx_t *s, int in_bit)
{
    /* v17tx.c:106-116 with scrambler_tap = 17 */
    const int out = (in_bit ^ (int) (s->scramble_reg >> 17) ^ (int) (s->scramble_reg >> 22)) & 1;

    s->scramble_reg = (s->scramble_reg << 1) | (uint32_t) out;
    return out;
}

static void training_get(orc_v17_tx_t *s, float v[2])
{
    static const int cdba_to_abcd[4] = {2, 3, 1, 0};
    static const int dibit_to_step[4] = {1, 0, 2, 3};
    int bits;
    int shift;

    if (++s->training_step <= SEG_3)
    {
        if (s->training_step <= SEG_2)
        {
            if (s->training_step <= TEP_B)
            {
                v[0] = abcd[0][0];              /* TEP: unmodulated carrier */
                v[1] = abcd[0][1];
            }
            else if (s->training_step <= SEG_1)
            {
                v[0] = v[1] = 0.0f;             /* silence */
            }
            else
            {
                const int k = (s->training_step & 1) ^ 1;       /* ABAB */

                v[0] = abcd[k][0];
                v[1] = abcd[k][1];
            }
            return;
        }
        /* CDBA through the scrambler */
        bits = scramble(s, 1);
        bits = (bits << 1) | scramble(s, 1);
        s->constellation_state = cdba_to_abcd[bits];
        if (s->short_train  &&  s->training_step == SHORT_SEG_4)
            s->training_step = SEG_4;
        v[0] = abcd[s->constellation_state][0];
        v[1] = abcd[s->constellation_state][1];
        return;
    }
    /* the bridge, carrying 0x8880 */
    shift = ((s->training_step - SEG_3 - 1) & 0x7) << 1;
    bits = scramble(s, 0x8880 >> shift);
    bits = (bits << 1) | scramble(s, 0x8880 >> (shift + 1));
    s->constellation_state = (s->constellation_state + dibit_to_step[bits]) & 3;
    v[0] = abcd[s->constellation_state][0];
    v[1] = abcd[s->constellation_state][1];
}

static int encode(orc_v17_tx_t *s, int q)
{
    /* diff_and_convolutional_encode(), v17tx.c:170-222 */
    static const uint8_t diff_4800[4][4] = {{2, 3, 0, 1}, {0, 2, 1, 3}, {3, 1, 2, 0}, {1, 0, 3, 2}};
    static const uint8_t conv[8][4] = {{0, 2, 3, 1}, {4, 7, 5, 6}, {1, 3, 2, 0}, {7, 4, 6, 5}, {2, 0, 1, 3}, {6, 5, 7, 4}, {3, 1, 0, 2},
                                       {5, 6, 4, 7}};

    if (s->bit_rate == 4800)
    {
        s->diff = diff_4800[s->diff][q & 3];
        return s->diff;
    }
    s->diff = (s->diff + (q & 3)) & 3;          /* the V.17 differential encoder table is addition mod 4 */
    s->convolution = conv[s->convolution][s->diff];
    return ((q << 1) & 0x78) | (s->diff << 1) | ((s->convolution >> 2) & 1);
}

static void next_baud(orc_v17_tx_t *s, float v[2])
{
    const int bits_per_symbol = s->bit_rate/2400;
    const float (*pts)[2] = points_of(s->bit_rate);
    int bits = 0;
    int idx;

    if (s->in_training)
    {
        if (s->training_step <= TRAINING_END)
        {
            if (s->training_step < SEG_4)
            {
                training_get(s, v);
                return;
            }
            if (++s->training_step > TRAINING_END)
                s->in_training = 0;
        }
        /* (the shutdown branch needs an end of data, which the LFSR never gives) */
    }
    for (int i = 0;  i < bits_per_symbol;  i++)
    {
        int bit = 1;

        if (!s->in_training)
        {
            bit = ((s->prbs >> 14) ^ (s->prbs >> 13)) & 1;
            s->prbs = ((s->prbs << 1) | (uint32_t) bit) & 0x7FFF;
        }
        bits |= (scramble(s, bit) << i);
    }
    idx = encode(s, bits);
    v[0] = pts[idx][0];
    v[1] = pts[idx][1];
}

ORC_API int orc_v17_tx(orc_v17_tx_t *s, int16_t amp[], int len)
{
    const float (*shaper)[9] = (const float (*)[9]) orc_v29_tx_shaper();
    int i;

    if (s->training_step >= SHUTDOWN_END)
        return 0;
    for (i = 0;  i < len;  i++)
    {
        float v[2];
        float z[2];
        float xre;
        float xim;

        if ((s->baud_phase += 3) >= 10)
        {
            s->baud_phase -= 10;
            next_baud(s, v);
            s->rrc_re[s->rrc_step] = v[0];
            s->rrc_im[s->rrc_step] = v[1];
            if (++s->rrc_step >= 9)
                s->rrc_step = 0;
        }
        xre = circular_dot(s->rrc_re, shaper[9 - s->baud_phase], 9, s->rrc_step);
        xim = circular_dot(s->rrc_im, shaper[9 - s->baud_phase], 9, s->rrc_step);
        dds_complex(s->carrier_phase, z);
        s->carrier_phase += (uint32_t) s->carrier_phase_rate;
        amp[i] = (int16_t) (long) ((xre*z[0] - xim*z[1])*s->gain);
    }
    return len;
}

ORC_API int orc_v17_tx_sizeof(void)
{
    return (int) sizeof(orc_v17_tx_t);
}
