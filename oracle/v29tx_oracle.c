/*
 * v29tx_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's V.29 transmitter, the signal source of the V.29 receiver tests
 * (SURVEY.md section 8(f)-1):
 *
 *   v29_tx_init / v29_tx_restart / v29_tx_power / set_working_gain     src/v29tx.c:286-416
 *   getbaud, get_scrambled_bit                                          src/v29tx.c:97-224
 *   v29_tx                                                              src/v29tx.c:226-284   (float build)
 *   the constellations                                                  src/v29tx_constellation_maps.h
 *   vec_circular_dot_prodf                                              src/vector_float.c (scalar path)
 *   dds_complexf                                                        src/dds_float.c:2184-2191
 *
 * Data bits come from the 15 bit LFSR the test glue feeds the reference with (oracle/ref_glue/ref_glue.c,
 * prbs_get_bit: x^15 + x^14 + 1); the end-of-data / shutdown path (a get_bit callback returning
 * SIG_STATUS_END_OF_DATA) is therefore never taken.  The pulse shaper table (10 phases x 9 taps) and the sine
 * table are handed in through orc_modem_set_tables() / orc_v29_tx_set_table().
 */
#include <math.h>
#include <string.h>

#include "oracle.h"
#include "modem_common.h"

enum
{
    SEG_TEP = 0, SEG_1 = 480, SEG_2 = 480 + 48, SEG_3 = 480 + 48 + 128, SEG_4 = 480 + 48 + 128 + 384,
    TRAINING_END = 480 + 48 + 128 + 384 + 48, SHUTDOWN_END = 480 + 48 + 128 + 384 + 48 + 32
};

static float tx_shaper[10][9];

ORC_API void orc_v29_tx_set_table(const float table[90])
{
    memcpy(tx_shaper, table, sizeof(tx_shaper));
}

/* the V.17 transmitter shapes with the same table (v17tx_oracle.c) */
const float *orc_v29_tx_shaper(void)
{
    return &tx_shaper[0][0];
}

/* The 16 point constellation: index = amplitude bit << 3 | phase octant (v29tx_constellation_maps.h) */
static void point(int idx, float z[2])
{
    static const int8_t axis[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
    const int oct = idx & 7;
    const int diag = oct & 1;
    const float r = (idx & 8)  ?  (diag  ?  3.0f  :  5.0f)  :  (diag  ?  1.0f  :  3.0f);

    z[0] = r*axis[oct][0];
    z[1] = r*axis[oct][1];
}

static void set_gain(orc_v29_tx_t *s)
{
    switch (s->bit_rate)
    {
    case 9600: s->gain = 0.387f*s->base_gain; break;
    case 7200: s->gain = 0.605f*s->base_gain; break;
    case 4800: s->gain = 0.470f*s->base_gain; break;
    }
}

ORC_API void orc_v29_tx_power(orc_v29_tx_t *s, float power)
{
    /* v29tx.c:324-338; TX_PULSESHAPER_GAIN is 1.0f in the float build */
    s->base_gain = powf(10.0f, (power - 3.14f)/20.0f)*32768.0f/1.000000f;
    set_gain(s);
}

ORC_API int orc_v29_tx_restart(orc_v29_tx_t *s, int bit_rate, int tep)
{
    s->bit_rate = bit_rate;
    set_gain(s);
    switch (bit_rate)
    {
    case 9600: s->training_offset = 0; break;
    case 7200: s->training_offset = 2; break;
    case 4800: s->training_offset = 4; break;
    default: return -1;
    }
    memset(s->rrc_re, 0, sizeof(s->rrc_re));
    memset(s->rrc_im, 0, sizeof(s->rrc_im));
    s->rrc_step = 0;
    s->scramble_reg = 0;
    s->training_scramble_reg = 0x2A;
    s->in_training = 1;
    s->training_step = tep  ?  SEG_TEP  :  SEG_1;
    s->carrier_phase = 0;
    s->baud_phase = 0;
    s->constellation_state = 0;
    return 0;
}

ORC_API int orc_v29_tx_init(orc_v29_tx_t *s, int bit_rate, int tep, uint32_t prbs_seed)
{
    if (bit_rate != 9600  &&  bit_rate != 7200  &&  bit_rate != 4800)
        return -1;
    memset(s, 0, sizeof(*s));
    s->prbs = prbs_seed & 0x7FFF;
    s->carrier_phase_rate = (int32_t) (1700.0f*65536.0f*65536.0f/8000);
    orc_v29_tx_power(s, -14.0f);
    return orc_v29_tx_restart(s, bit_rate, tep);
}

static int scrambled_bit(orc_v29_tx_t *s)
{
    int bit = 1;                /* fake_get_bit() while training */
    int out;

    if (!s->in_training)
    {
        bit = ((s->prbs >> 14) ^ (s->prbs >> 13)) & 1;
        s->prbs = ((s->prbs << 1) | (uint32_t) bit) & 0x7FFF;
    }
    out = (bit ^ (s->scramble_reg >> 17) ^ (s->scramble_reg >> 22)) & 1;
    s->scramble_reg = (s->scramble_reg << 1) | (uint32_t) out;
    return out;
}

static void next_baud(orc_v29_tx_t *s, float v[2])
{
    static const int steps_9600[8] = {1, 0, 2, 3, 6, 7, 5, 4};
    static const int steps_4800[4] = {0, 2, 6, 4};
    int bits;
    int amp = 0;

    if (s->in_training)
    {
        if (++s->training_step <= SEG_4)
        {
            if (s->training_step <= SEG_1)
            {
                point(0, v);                    /* TEP: unmodulated carrier */
            }
            else if (s->training_step <= SEG_2)
            {
                v[0] = v[1] = 0.0f;             /* silence */
            }
            else if (s->training_step <= SEG_3)
            {
                /* ABAB: A = 315 deg high (9600) / 315 deg low (7200) / 270 deg low (4800), B = 180 deg low */
                static const int8_t a_idx[3] = {15, 7, 6};
                point((s->training_step & 1)  ?  4  :  a_idx[s->training_offset >> 1], v);
            }
            else
            {
                /* CDCD through the 1 + x^-6 + x^-7 scrambler: C = 0 deg low, D = 135 deg high / low, 90 deg low */
                static const int8_t d_idx[3] = {11, 3, 2};
                const int bit = s->training_scramble_reg & 1;

                s->training_scramble_reg >>= 1;
                s->training_scramble_reg |= (uint32_t) (((bit ^ (int) s->training_scramble_reg) & 1) << 6);
                s->training_scramble_reg &= 0xFF;
                point(bit  ?  d_idx[s->training_offset >> 1]  :  0, v);
            }
            return;
        }
        if (s->training_step == TRAINING_END + 1)
            s->in_training = 0;
    }
    if (s->bit_rate == 9600  &&  scrambled_bit(s))
        amp = 8;
    bits = scrambled_bit(s);
    bits = (bits << 1) | scrambled_bit(s);
    if (s->bit_rate == 4800)
    {
        bits = steps_4800[bits];
    }
    else
    {
        bits = (bits << 1) | scrambled_bit(s);
        bits = steps_9600[bits];
    }
    s->constellation_state = (s->constellation_state + bits) & 7;
    point(amp | s->constellation_state, v);
}

ORC_API int orc_v29_tx(orc_v29_tx_t *s, int16_t amp[], int len)
{
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
        xre = circular_dot(s->rrc_re, tx_shaper[9 - s->baud_phase], 9, s->rrc_step);
        xim = circular_dot(s->rrc_im, tx_shaper[9 - s->baud_phase], 9, s->rrc_step);
        dds_complex(s->carrier_phase, z);
        s->carrier_phase += (uint32_t) s->carrier_phase_rate;
        /* lfastrintf() truncates on x86-64 */
        amp[i] = (int16_t) (long) ((xre*z[0] - xim*z[1])*s->gain);
    }
    return len;
}

ORC_API int orc_v29_tx_sizeof(void)
{
    return (int) sizeof(orc_v29_tx_t);
}
