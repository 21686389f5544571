/*
 * v27tertx_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's V.27ter transmitter, the signal source of the V.27ter receiver tests
 * (SURVEY.md section 8(f)-1):
 *
 *   v27ter_tx_init / v27ter_tx_restart / v27ter_tx_power     src/v27ter_tx.c:352-430
 *   scramble, get_scrambled_bit, getbaud                     src/v27ter_tx.c:103-244
 *   v27ter_tx                                                src/v27ter_tx.c:246-350   (float build)
 *
 * Data bits come from the 15 bit LFSR the test glue feeds the reference with (x^15 + x^14 + 1); the
 * end-of-data / shutdown path is therefore never taken.  Pulse shaper tables (4800 bps: 5 phases x 9 taps,
 * 2400 bps: 20 x 9) through orc_v27ter_tx_set_tables(), the sine table through orc_modem_set_tables().
 */
#include <math.h>
#include <string.h>

#include "oracle.h"
#include "modem_common.h"

enum
{
    SEG_1 = 0, SEG_2 = 320, SEG_3 = 320 + 32, SEG_4 = 320 + 32 + 50, SEG_5 = 320 + 32 + 50 + 1074,
    TRAINING_END = 320 + 32 + 50 + 1074 + 8, SHUTDOWN_END = 320 + 32 + 50 + 1074 + 8 + 32
};

static float shaper_4800[5][9];
static float shaper_2400[20][9];

ORC_API void orc_v27ter_tx_set_tables(const float t4800[45], const float t2400[180])
{
    memcpy(shaper_4800, t4800, sizeof(shaper_4800));
    memcpy(shaper_2400, t2400, sizeof(shaper_2400));
}

ORC_API void orc_v27ter_tx_power(orc_v27ter_tx_t *s, float power)
{
    /* v27ter_tx.c:352-364; both TX_PULSESHAPER_xxxx_GAIN are 1.0f in the float build */
    const float gain = powf(10.0f, (power - 3.14f)/20.0f)*32768.0f;

    s->gain_2400 = gain/1.000000f;
    s->gain_4800 = gain/1.000000f;
}

ORC_API int orc_v27ter_tx_restart(orc_v27ter_tx_t *s, int bit_rate, int tep)
{
    if (bit_rate != 4800  &&  bit_rate != 2400)
        return -1;
    s->bit_rate = bit_rate;
    memset(s->rrc_re, 0, sizeof(s->rrc_re));
    memset(s->rrc_im, 0, sizeof(s->rrc_im));
    s->rrc_step = 0;
    s->scramble_reg = 0x3C;
    s->scrambler_pattern_count = 0;
    s->in_training = 1;
    s->training_step = tep  ?  SEG_1  :  SEG_2;
    s->carrier_phase = 0;
    s->baud_phase = 0;
    s->constellation_state = 0;
    return 0;
}

ORC_API int orc_v27ter_tx_init(orc_v27ter_tx_t *s, int bit_rate, int tep, uint32_t prbs_seed)
{
    if (bit_rate != 4800  &&  bit_rate != 2400)
        return -1;
    memset(s, 0, sizeof(*s));
    s->prbs = prbs_seed & 0x7FFF;
    s->carrier_phase_rate = (int32_t) (1800.0f*65536.0f*65536.0f/8000);
    orc_v27ter_tx_power(s, -14.0f);
    return orc_v27ter_tx_restart(s, bit_rate, tep);
}

static int scrambled_bit(orc_v27ter_tx_t *s)
{
    int bit = 1;                /* fake_get_bit() while training */
    int out;

    if (!s->in_training)
    {
        bit = ((s->prbs >> 14) ^ (s->prbs >> 13)) & 1;
        s->prbs = ((s->prbs << 1) | (uint32_t) bit) & 0x7FFF;
    }
    /* scramble(), v27ter_tx.c:103-125: 1 + x^-6 + x^-7 with the guard against long repeating patterns */
    out = (bit ^ (s->scramble_reg >> 5) ^ (s->scramble_reg >> 6)) & 1;
    if (s->scrambler_pattern_count >= 33)
    {
        out ^= 1;
        s->scrambler_pattern_count = 0;
    }
    else if ((((s->scramble_reg >> 7) ^ out) & ((s->scramble_reg >> 8) ^ out) & ((s->scramble_reg >> 11) ^ out) & 1))
    {
        s->scrambler_pattern_count = 0;
    }
    else
    {
        s->scrambler_pattern_count++;
    }
    s->scramble_reg = (s->scramble_reg << 1) | (uint32_t) out;
    return out;
}

static void point(int oct, float v[2])
{
    /* eight phases: 1.414 on the axes, (+-1, +-1) between them (v27ter_tx.c:163-173) */
    static const float pts[8][2] = {{1.414f, 0.0f}, {1.0f, 1.0f}, {0.0f, 1.414f}, {-1.0f, 1.0f}, {-1.414f, 0.0f}, {-1.0f, -1.0f},
                                    {0.0f, -1.414f}, {1.0f, -1.0f}};
    v[0] = pts[oct][0];
    v[1] = pts[oct][1];
}

static void next_baud(orc_v27ter_tx_t *s, float v[2])
{
    static const int steps_4800[8] = {1, 0, 2, 3, 6, 7, 5, 4};
    static const int steps_2400[4] = {0, 2, 6, 4};
    int bits;

    if (s->in_training)
    {
        if (++s->training_step <= SEG_5)
        {
            if (s->training_step <= SEG_2)
            {
                point(0, v);                    /* unmodulated carrier (TEP) */
            }
            else if (s->training_step <= SEG_3)
            {
                v[0] = v[1] = 0.0f;             /* silence */
            }
            else if (s->training_step <= SEG_4)
            {
                s->constellation_state = (s->constellation_state + 4) & 7;      /* regular reversals */
                point(s->constellation_state, v);
            }
            else
            {
                /* scrambled reversals: every third bit of the scrambler */
                bits = scrambled_bit(s) << 2;
                scrambled_bit(s);
                scrambled_bit(s);
                s->constellation_state = (s->constellation_state + bits) & 7;
                point(s->constellation_state, v);
            }
            return;
        }
        if (s->training_step == TRAINING_END + 1)
            s->in_training = 0;
    }
    bits = scrambled_bit(s);
    bits = (bits << 1) | scrambled_bit(s);
    if (s->bit_rate == 4800)
    {
        bits = (bits << 1) | scrambled_bit(s);
        bits = steps_4800[bits];
    }
    else
    {
        bits = steps_2400[bits];
    }
    s->constellation_state = (s->constellation_state + bits) & 7;
    point(s->constellation_state, v);
}

ORC_API int orc_v27ter_tx(orc_v27ter_tx_t *s, int16_t amp[], int len)
{
    int i;

    if (s->training_step >= SHUTDOWN_END)
        return 0;
    for (i = 0;  i < len;  i++)
    {
        const float *coef;
        float v[2];
        float z[2];
        float xre;
        float xim;
        int fresh;

        if (s->bit_rate == 4800)
        {
            if ((fresh = (++s->baud_phase >= 5)))
                s->baud_phase -= 5;
        }
        else
        {
            if ((fresh = ((s->baud_phase += 3) >= 20)))
                s->baud_phase -= 20;
        }
        if (fresh)
        {
            next_baud(s, v);
            s->rrc_re[s->rrc_step] = v[0];
            s->rrc_im[s->rrc_step] = v[1];
            if (++s->rrc_step >= 9)
                s->rrc_step = 0;
        }
        coef = (s->bit_rate == 4800)  ?  shaper_4800[4 - s->baud_phase]  :  shaper_2400[19 - s->baud_phase];
        xre = circular_dot(s->rrc_re, coef, 9, s->rrc_step);
        xim = circular_dot(s->rrc_im, coef, 9, s->rrc_step);
        dds_complex(s->carrier_phase, z);
        s->carrier_phase += (uint32_t) s->carrier_phase_rate;
        amp[i] = (int16_t) (long) ((xre*z[0] - xim*z[1])*((s->bit_rate == 4800)  ?  s->gain_4800  :  s->gain_2400));
    }
    return len;
}

ORC_API int orc_v27ter_tx_sizeof(void)
{
    return (int) sizeof(orc_v27ter_tx_t);
}
