/*
 * v17_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's V.17 (and V.32bis 4800) receiver, float build (v17rx.c forces
 * the floating point path at :68):
 *   v17_rx / signal_detect / process_half_baud                    src/v17rx.c:600-1358
 *   decode_baud (soft decisions, 8-state trellis, traceback)      src/v17rx.c:396-589
 *   descramble, track_carrier, tune_equalizer, equalizer_*        src/v17rx.c:214-372
 *   v17_rx_restart / v17_rx_init                                  src/v17rx.c:1399-1541
 *   godard_ted_rx / godard_ted_per_baud                           src/godard.c:144-220
 * IAXMODEM_STUFF is #defined at v17rx.c:1: the quick power-drop path of signal_detect and the early
 * hand-over from coarse to fine training (v17rx.c:817-826) are part of the behaviour.
 * Tables (pulse shaper, Godard descriptor, constellations, soft-decision maps) come from the test
 * harness through orc_modem_set_tables().
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"
#include "modem_common.h"

#define T orc_modem_T

#define RRC_SETS        192
#define RRC_LEN         27
#define EQ_LEN          33
#define EQ_PRE          16
#define FAST_DELTA      (0.21f/EQ_LEN)
#define SLOW_DELTA      (0.1f*FAST_DELTA)
#define SEG_1_LEN       256
#define SEG_2_LEN       2976
#define SHORT_SEG_2_LEN 38
#define SEG_3_LEN       64
#define SEG_4A_LEN      15
#define SEG_4_LEN       48
#define BRIDGE_WORD     0x8880
#define TRELLIS_DEPTH   16

enum
{
    ST_NORMAL = 0,
    ST_SYMBOL_ACQUISITION,
    ST_LOG_PHASE,
    ST_SHORT_WAIT_FOR_CDBA,
    ST_WAIT_FOR_CDBA,
    ST_COARSE_TRAIN_ON_CDBA,
    ST_FINE_TRAIN_ON_CDBA,
    ST_SHORT_TRAIN_ON_CDBA_AND_TEST,
    ST_TRAIN_ON_CDBA_AND_TEST,
    ST_BRIDGE,
    ST_TCM_WINDUP,
    ST_TEST_ONES,
    ST_PARKED
};

static const float SPACING[4] = {1.414f, 2.0f, 2.828f, 4.0f};          /* v17rx.c:157-163 */
static const float CDBA[4][2] = {{6.0f, 2.0f}, {-2.0f, 6.0f}, {2.0f, -6.0f}, {-6.0f, -2.0f}};  /* v17rx.c:601-607 */

int orc_v17_sizeof(void) { return (int) sizeof(orc_v17_t); }

static const float (*constellation(const orc_v17_t *s))[2]
{
    static const int offset[5] = {0, 128, 192, 224, 240};
    int k;

    switch (s->bit_rate)
    {
    case 14400: k = 0; break;
    case 12000: k = 1; break;
    case 9600: k = 2; break;
    case 7200: k = 3; break;
    default: k = 4; break;
    }
    return (const float (*)[2]) (T.v17_constellation + 2*offset[k]);
}

static void equalizer_restore(orc_v17_t *s)
{
    memcpy(s->eq_coeff, s->eq_coeff_save, sizeof(s->eq_coeff));
    memset(s->eq_buf, 0, sizeof(s->eq_buf));
    s->eq_delta = SLOW_DELTA;
    s->eq_put_step = RRC_SETS*10/(3*2) - 1;
    s->eq_step = 0;
    s->eq_skip = 0;
}

static void equalizer_reset(orc_v17_t *s)
{
    memset(s->eq_coeff, 0, sizeof(s->eq_coeff));
    s->eq_coeff[EQ_PRE][0] = 3.0f;
    memset(s->eq_buf, 0, sizeof(s->eq_buf));
    s->eq_delta = FAST_DELTA;
    s->eq_put_step = RRC_SETS*10/(3*2) - 1;
    s->eq_step = 0;
    s->eq_skip = 0;
}

/* v17rx.c:1399-1500 */
int orc_v17_restart(orc_v17_t *s, int bit_rate, int short_train)
{
    int i;

    switch (bit_rate)
    {
    case 14400: s->space_map = 0; s->bits_per_symbol = 6; break;
    case 12000: s->space_map = 1; s->bits_per_symbol = 5; break;
    case 9600: s->space_map = 2; s->bits_per_symbol = 4; break;
    case 7200: s->space_map = 3; s->bits_per_symbol = 3; break;
    case 4800: s->space_map = 0; s->bits_per_symbol = 2; break;
    default: return -1;
    }
    s->bit_rate = bit_rate;
    memset(s->rrc_filter, 0, sizeof(s->rrc_filter));
    s->training_error = 0.0f;
    s->rrc_filter_step = 0;
    s->diff = 1;
    s->scramble_reg = 0x2ECDD5;
    s->training_stage = ST_SYMBOL_ACQUISITION;
    s->training_count = 0;
    s->signal_present = 0;
    s->high_sample = 0;
    s->low_samples = 0;
    s->carrier_drop_pending = 0;
    if (short_train != 2)
        s->short_train = (short_train != 0);
    memset(s->last_angles, 0, sizeof(s->last_angles));
    memset(s->diff_angles, 0, sizeof(s->diff_angles));
    for (i = 0;  i < 8;  i++)
        s->distances[i] = 99.0f*1.0f;
    memset(s->full_path_to_past_state_locations, 0, sizeof(s->full_path_to_past_state_locations));
    memset(s->past_state_locations, 0, sizeof(s->past_state_locations));
    s->distances[0] = 0;
    s->trellis_ptr = 14;
    s->carrier_phase = 0;
    s->power_reading = 0;
    if (s->short_train)
    {
        s->carrier_phase_rate = s->carrier_phase_rate_save;
        equalizer_restore(s);
        s->agc_scaling = s->agc_scaling_save;
        s->carrier_track_i = 0.0f;
        s->carrier_track_p = 40000.0f;
    }
    else
    {
        s->carrier_phase_rate = (int32_t) (1800.0f*65536.0f*65536.0f/8000);
        equalizer_reset(s);
        s->agc_scaling_save = 0.0f;
        s->agc_scaling = (2.17f/1.000000f)/735.0f;
        s->carrier_track_i = 5000.0f;
        s->carrier_track_p = 40000.0f;
    }
    s->last_sample = 0;
    memset(s->g_low, 0, sizeof(s->g_low));
    memset(s->g_high, 0, sizeof(s->g_high));
    memset(s->g_dc, 0, sizeof(s->g_dc));
    s->g_baud_phase = 0.0f;
    s->g_total_correction = 0;
    s->baud_half = 0;
    return 0;
}

/* v17rx.c:1502-1535 */
int orc_v17_init(orc_v17_t *s, int bit_rate)
{
    switch (bit_rate)
    {
    case 14400: case 12000: case 9600: case 7200: case 4800:
        break;
    default:
        return -1;
    }
    memset(s, 0, sizeof(*s));
    s->short_train = 0;
    s->scrambler_tap = 18 - 1;
    s->carrier_on_power = (int32_t) (level_dbm0(-45.5f + 2.5f)*0.4f);
    s->carrier_off_power = (int32_t) (level_dbm0(-45.5f - 2.5f)*0.4f);
    s->carrier_phase_rate_save = (int32_t) (1800.0f*65536.0f*65536.0f/8000);
    return orc_v17_restart(s, bit_rate, s->short_train);
}

static void report_status(orc_sink_t *sink, int status)
{
    orc_sink_push(sink, 3, status, 0, 0);
}

static void godard_rx(orc_v17_t *s, float sample)
{
    float v;

    v = s->g_low[0]*T.v17_godard[0] + s->g_low[1]*T.v17_godard[1] + sample;
    s->g_low[1] = s->g_low[0];
    s->g_low[0] = v;
    v = s->g_high[0]*T.v17_godard[3] + s->g_high[1]*T.v17_godard[4] + sample;
    s->g_high[1] = s->g_high[0];
    s->g_high[0] = v;
}

static int godard_per_baud(orc_v17_t *s)
{
    float v;
    float p;
    int i;

    v = s->g_low[1]*s->g_high[0]*T.v17_godard[2]
      - s->g_low[0]*s->g_high[1]*T.v17_godard[5]
      + s->g_low[1]*s->g_high[1]*T.v17_godard[6];
    p = v - s->g_dc[1];
    s->g_dc[1] = s->g_dc[0];
    s->g_dc[0] = v;
    s->g_baud_phase -= p;
    v = fabsf(s->g_baud_phase);
    if (v > T.v17_godard_fine_trigger)
    {
        i = (v > T.v17_godard_coarse_trigger)  ?  T.v17_godard_coarse_step  :  T.v17_godard_fine_step;
        if (s->g_baud_phase < 0.0f)
            i = -i;
        s->g_total_correction += i;
        return i;
    }
    return 0;
}

static void track_carrier(orc_v17_t *s, const float z[2], const float target[2])
{
    float error;

    error = z[1]*target[0] - z[0]*target[1];
    s->carrier_phase_rate += (int32_t) (s->carrier_track_i*error);
    s->carrier_phase += (uint32_t) (int32_t) (s->carrier_track_p*error);
}

static void tune_equalizer(orc_v17_t *s, const float z[2], const float target[2])
{
    float err_re;
    float err_im;

    err_re = (target[0] - z[0])*s->eq_delta;
    err_im = (target[1] - z[1])*s->eq_delta;
    ccircular_lms((const float (*)[2]) s->eq_buf, s->eq_coeff, EQ_LEN, s->eq_step, err_re, err_im);
}

/* v17rx.c:336-349 */
static int descramble(orc_v17_t *s, int in_bit)
{
    int out_bit;

    in_bit &= 1;
    out_bit = (in_bit ^ (s->scramble_reg >> s->scrambler_tap) ^ (s->scramble_reg >> (23 - 1))) & 1;
    s->scramble_reg <<= 1;
    if (s->training_stage > ST_NORMAL  &&  s->training_stage < ST_TCM_WINDUP)
        s->scramble_reg |= out_bit;
    else
        s->scramble_reg |= (in_bit & 1);
    return out_bit;
}

static void put_bit(orc_v17_t *s, orc_sink_t *sink, int bit)
{
    int out_bit = descramble(s, bit);

    if (s->training_stage == ST_NORMAL)
        orc_sink_push(sink, 3, out_bit, 0, 0);
}

static float dist_sq(const float x[2], const float y[2])
{
    return (x[0] - y[0])*(x[0] - y[0]) + (x[1] - y[1])*(x[1] - y[1]);
}

/* v17rx.c:396-589 */
static int decode_baud(orc_v17_t *s, orc_sink_t *sink, const float z[2])
{
    static const uint8_t v32bis_4800_differential_decoder[4][4] = {{2, 3, 0, 1}, {0, 2, 1, 3}, {3, 1, 2, 0}, {1, 0, 3, 2}};
    static const uint8_t v17_differential_decoder[4][4] = {{0, 1, 2, 3}, {3, 0, 1, 2}, {2, 3, 0, 1}, {1, 2, 3, 0}};
    static const uint8_t tcm_paths[8][4] =
    {
        {0, 6, 2, 4}, {6, 0, 4, 2}, {2, 4, 0, 6}, {4, 2, 6, 0}, {1, 3, 7, 5}, {5, 7, 3, 1}, {7, 5, 1, 3}, {3, 1, 5, 7}
    };
    const float (*con)[2] = constellation(s);
    const uint8_t *cell;
    float distances[8];
    float new_distances[8];
    float min;
    int nearest;
    int i;
    int j;
    int k;
    int re;
    int im;
    int raw;
    int min_index;
    int set;
    int constellation_state;

    re = (int) ((z[0] + 9.0f)*2.0f);
    im = (int) ((z[1] + 9.0f)*2.0f);
    if (re > 35)
        re = 35;
    else if (re < 0)
        re = 0;
    if (im > 35)
        im = 35;
    else if (im < 0)
        im = 0;
    if (s->bits_per_symbol == 2)
    {
        constellation_state = T.v17_map_4800[re*36 + im];
        raw = v32bis_4800_differential_decoder[s->diff][constellation_state];
        s->diff = constellation_state;
        put_bit(s, sink, raw);
        put_bit(s, sink, raw >> 1);
        return constellation_state;
    }
    cell = T.v17_maps + ((s->space_map*36 + re)*36 + im)*8;
    min = 9999999.0f;
    min_index = 0;
    for (i = 0;  i < 8;  i++)
    {
        nearest = cell[i];
        distances[i] = dist_sq(con[nearest], z);
        if (min > distances[i])
        {
            min = distances[i];
            min_index = i;
        }
    }
    constellation_state = cell[min_index];
    track_carrier(s, z, con[constellation_state]);

    if (++s->trellis_ptr >= TRELLIS_DEPTH)
        s->trellis_ptr = 0;
    for (i = 0;  i < 8;  i++)
    {
        set = i >> 2;
        min = distances[tcm_paths[i][0]] + s->distances[set];
        min_index = 0;
        for (j = 1;  j < 4;  j++)
        {
            k = (j << 1) + set;
            if (min > distances[tcm_paths[i][j]] + s->distances[k])
            {
                min = distances[tcm_paths[i][j]] + s->distances[k];
                min_index = j;
            }
        }
        k = (min_index << 1) + set;
        new_distances[i] = s->distances[k]*0.9f + distances[tcm_paths[i][min_index]]*0.1f;
        s->full_path_to_past_state_locations[s->trellis_ptr][i] = cell[tcm_paths[i][min_index]];
        s->past_state_locations[s->trellis_ptr][i] = k;
    }
    memcpy(s->distances, new_distances, sizeof(s->distances));

    min = s->distances[0];
    min_index = 0;
    for (i = 1;  i < 8;  i++)
    {
        if (min > s->distances[i])
        {
            min = s->distances[i];
            min_index = i;
        }
    }
    k = min_index;
    for (i = 0, j = s->trellis_ptr;  i < TRELLIS_DEPTH - 1;  i++)
    {
        k = s->past_state_locations[j][k];
        if (--j < 0)
            j = TRELLIS_DEPTH - 1;
    }
    nearest = s->full_path_to_past_state_locations[j][k] >> 1;

    raw = (nearest & 0x3C) | v17_differential_decoder[s->diff][nearest & 0x03];
    s->diff = nearest & 0x03;
    for (i = 0;  i < s->bits_per_symbol;  i++)
    {
        put_bit(s, sink, raw);
        raw >>= 1;
    }
    return constellation_state;
}

static void spin(orc_v17_t *s, uint32_t phase_step)
{
    float p;
    float zz[2];
    float t;
    int i;

    p = phase_step*2.0f*3.1415926f/(65536.0f*65536.0f);         /* dds_phase_to_radians, dds_float.c:2103 */
    zz[0] = orc_cosf(p);
    zz[1] = -orc_sinf(p);
    for (i = 0;  i < EQ_LEN;  i++)
    {
        t = s->eq_buf[i][0]*zz[0] - s->eq_buf[i][1]*zz[1];
        s->eq_buf[i][1] = s->eq_buf[i][0]*zz[1] + s->eq_buf[i][1]*zz[0];
        s->eq_buf[i][0] = t;
    }
    s->carrier_phase += phase_step;
}

static void park(orc_v17_t *s, orc_sink_t *sink, int clear_agc)
{
    if (clear_agc)
        s->agc_scaling_save = 0.0f;
    s->training_stage = ST_PARKED;
    report_status(sink, -5);                                    /* SIG_STATUS_TRAINING_FAILED */
}

/* DDS_PHASE(), spandsp/dds.h:32: degrees to a 32 bit phase, evaluated in float */
#define DDS_PHASE_F(deg)    ((int32_t) ((uint32_t) ((((deg) < 0.0f)  ?  (360.0f + (deg))  :  (deg))*65536.0f*65536.0f/360.0f)))

/* v17rx.c:592-1130 */
static void process_half_baud(orc_v17_t *s, orc_sink_t *sink, const float sample[2])
{
    const float (*con)[2] = constellation(s);
    float z[2];
    float zz[2];
    static const float zero[2] = {0.0f, 0.0f};
    const float *target;
    int symbol = 0;                 /* the reference's local constellation_state, what qam_report() is given */
    int bit;
    int i;
    int j;
    uint32_t phase_step;
    int32_t angle;
    int32_t ang;
    int cs;

    s->eq_buf[s->eq_step][0] = sample[0];
    s->eq_buf[s->eq_step][1] = sample[1];
    if (++s->eq_step >= EQ_LEN)
        s->eq_step = 0;
    if ((s->baud_half ^= 1))
        return;
    s->eq_put_step += godard_per_baud(s);
    ccircular_dot((const float (*)[2]) s->eq_buf, (const float (*)[2]) s->eq_coeff, EQ_LEN, s->eq_step, z);

    switch (s->training_stage)
    {
    case ST_NORMAL:
        symbol = decode_baud(s, sink, z);
        target = con[symbol];
        break;
    case ST_SYMBOL_ACQUISITION:
        target = zero;
        if (++s->training_count >= 100)
        {
            s->training_stage = ST_LOG_PHASE;
            memset(s->diff_angles, 0, sizeof(s->diff_angles));
            s->last_angles[0] = arctan2_i(z[1], z[0]);
            if (s->agc_scaling_save == 0.0f)
                s->agc_scaling_save = s->agc_scaling;
        }
        break;
    case ST_LOG_PHASE:
        target = zero;
        angle = arctan2_i(z[1], z[0]);
        s->training_count = 1;
        if (s->short_train)
        {
            if ((uint32_t) (angle - s->last_angles[0]) < (uint32_t) DDS_PHASE_F(180.0f))
            {
                angle = s->last_angles[0];
                s->last_angles[0] = DDS_PHASE_F(270.0f + 18.433f);
                s->last_angles[1] = DDS_PHASE_F(180.0f + 18.433f);
            }
            else
            {
                s->last_angles[0] = DDS_PHASE_F(180.0f + 18.433f);
                s->last_angles[1] = DDS_PHASE_F(270.0f + 18.433f);
            }
            phase_step = (uint32_t) angle - (uint32_t) DDS_PHASE_F(180.0f + 18.433f);
            s->carrier_track_p = 500000.0f;
            spin(s, phase_step);
            s->training_stage = ST_SHORT_WAIT_FOR_CDBA;
        }
        else
        {
            s->last_angles[1] = angle;
            s->training_stage = ST_WAIT_FOR_CDBA;
        }
        break;
    case ST_WAIT_FOR_CDBA:
        target = zero;
        angle = arctan2_i(z[1], z[0]);
        i = s->training_count + 1;
        ang = (int32_t) ((uint32_t) angle - (uint32_t) s->last_angles[i & 1]);
        s->last_angles[i & 1] = angle;
        s->diff_angles[i & 0xF] = (int32_t) ((uint32_t) s->diff_angles[(i - 2) & 0xF] + (uint32_t) (ang >> 4));
        if ((ang > DDS_PHASE_F(90.0f)  ||  ang < DDS_PHASE_F(-90.0f))  &&  s->training_count >= 13)
        {
            i = (s->training_count - 8) & ~1;
            if (i > 1)
            {
                j = i & 0xF;
                ang = (int32_t) ((uint32_t) s->diff_angles[j] + (uint32_t) s->diff_angles[j | 0x1])/(i - 1);
                s->carrier_phase_rate += 3*16*(ang/20);
            }
            if (s->carrier_phase_rate < (int32_t) ((1800.0f - 20.0f)*65536.0f*65536.0f/8000)
                ||
                s->carrier_phase_rate > (int32_t) ((1800.0f + 20.0f)*65536.0f*65536.0f/8000))
            {
                park(s, sink, 1);
                break;
            }
            phase_step = (uint32_t) angle - (uint32_t) DDS_PHASE_F(18.433f);
            spin(s, phase_step);
            bit = descramble(s, 1);
            bit = (bit << 1) | descramble(s, 1);
            target = CDBA[bit];
            s->training_count = 1;
            s->training_stage = ST_COARSE_TRAIN_ON_CDBA;
            report_status(sink, -3);                            /* SIG_STATUS_TRAINING_IN_PROGRESS */
            break;
        }
        if (++s->training_count > SEG_1_LEN)
            park(s, sink, 1);
        break;
    case ST_COARSE_TRAIN_ON_CDBA:
        bit = descramble(s, 1);
        bit = (bit << 1) | descramble(s, 1);
        target = CDBA[bit];
        track_carrier(s, z, target);
        tune_equalizer(s, z, target);
        zz[0] = z[0] - target[0];
        zz[1] = z[1] - target[1];
        s->training_error = zz[0]*zz[0] + zz[1]*zz[1];
        if (++s->training_count == SEG_2_LEN - 2000  ||  s->training_error < 1.0f*1.0f  ||  s->training_error > 200.0f*1.0f)
        {
            s->eq_delta = SLOW_DELTA;
            s->carrier_track_i = 1000.0f;
            s->training_stage = ST_FINE_TRAIN_ON_CDBA;
        }
        break;
    case ST_FINE_TRAIN_ON_CDBA:
        bit = descramble(s, 1);
        bit = (bit << 1) | descramble(s, 1);
        target = CDBA[bit];
        track_carrier(s, z, target);
        tune_equalizer(s, z, target);
        if (++s->training_count >= SEG_2_LEN - 48)
        {
            s->training_error = 0.0f;
            s->carrier_track_i = 100.0f;
            s->carrier_track_p = 500000.0f;
            s->training_stage = ST_TRAIN_ON_CDBA_AND_TEST;
        }
        break;
    case ST_TRAIN_ON_CDBA_AND_TEST:
        bit = descramble(s, 1);
        bit = (bit << 1) | descramble(s, 1);
        target = CDBA[bit];
        if (++s->training_count < SEG_2_LEN - 20)
        {
            track_carrier(s, z, target);
            tune_equalizer(s, z, target);
            zz[0] = z[0] - target[0];
            zz[1] = z[1] - target[1];
            s->training_error += (zz[0]*zz[0] + zz[1]*zz[1]);
        }
        else if (s->training_count >= SEG_2_LEN)
        {
            if (s->training_error < 20.0f*1.414f*SPACING[s->space_map])
            {
                s->training_error = 0.0f;
                s->training_count = 0;
                s->training_stage = ST_BRIDGE;
            }
            else
            {
                park(s, sink, 1);
            }
        }
        break;
    case ST_BRIDGE:
        descramble(s, BRIDGE_WORD >> ((s->training_count & 0x7) << 1));
        descramble(s, BRIDGE_WORD >> (((s->training_count & 0x7) << 1) + 1));
        target = z;
        if (++s->training_count >= SEG_3_LEN)
        {
            s->training_error = 0.0f;
            s->training_count = 0;
            if (s->bits_per_symbol == 2)
            {
                s->diff = (s->short_train)  ?  0  :  1;
                s->training_stage = ST_TEST_ONES;
            }
            else
            {
                s->training_stage = ST_TCM_WINDUP;
            }
        }
        break;
    case ST_SHORT_WAIT_FOR_CDBA:
        angle = arctan2_i(z[1], z[0]);
        ang = (int32_t) ((uint32_t) angle - (uint32_t) s->last_angles[s->training_count & 1]);
        if (ang > DDS_PHASE_F(90.0f)  ||  ang < DDS_PHASE_F(-90.0f))
        {
            bit = descramble(s, 1);
            bit = (bit << 1) | descramble(s, 1);
            target = CDBA[bit];
            s->training_error = 0.0f;
            s->training_count = 1;
            s->training_stage = ST_SHORT_TRAIN_ON_CDBA_AND_TEST;
        }
        else
        {
            target = CDBA[(s->training_count & 1) + 2];
            track_carrier(s, z, target);
            if (++s->training_count > SEG_1_LEN)
                park(s, sink, 0);
        }
        break;
    case ST_SHORT_TRAIN_ON_CDBA_AND_TEST:
        bit = descramble(s, 1);
        bit = (bit << 1) | descramble(s, 1);
        target = CDBA[bit];
        track_carrier(s, z, target);
        if (s->training_count > 8)
        {
            zz[0] = z[0] - target[0];
            zz[1] = z[1] - target[1];
            s->training_error += (zz[0]*zz[0] + zz[1]*zz[1]);
        }
        if (++s->training_count >= SHORT_SEG_2_LEN)
        {
            s->carrier_track_i = 100.0f;
            s->carrier_track_p = 500000.0f;
            if (s->training_error < (SHORT_SEG_2_LEN - 8)*4.0f*1.0f*SPACING[s->space_map])
            {
                s->training_count = 0;
                if (s->bits_per_symbol == 2)
                {
                    s->diff = (s->short_train)  ?  0  :  1;
                    s->training_error = 0.0f;
                    s->training_stage = ST_TEST_ONES;
                }
                else
                {
                    s->training_stage = ST_TCM_WINDUP;
                }
                report_status(sink, -3);
            }
            else
            {
                park(s, sink, 0);
            }
        }
        break;
    case ST_TCM_WINDUP:
        cs = decode_baud(s, sink, z);
        symbol = cs;
        target = con[cs];
        zz[0] = z[0] - con[cs][0];
        zz[1] = z[1] - con[cs][1];
        s->training_error += (zz[0]*zz[0] + zz[1]*zz[1]);
        if (++s->training_count >= SEG_4A_LEN)
        {
            s->training_error = 0.0f;
            s->training_count = 0;
            s->diff = (s->short_train)  ?  0  :  1;
            s->training_stage = ST_TEST_ONES;
        }
        break;
    case ST_TEST_ONES:
        cs = decode_baud(s, sink, z);
        symbol = cs;
        target = con[cs];
        zz[0] = z[0] - con[cs][0];
        zz[1] = z[1] - con[cs][1];
        s->training_error += (zz[0]*zz[0] + zz[1]*zz[1]);
        if (++s->training_count >= SEG_4_LEN)
        {
            if (s->training_error < SEG_4_LEN*1.0f*1.0f*SPACING[s->space_map])
            {
                report_status(sink, -4);                        /* SIG_STATUS_TRAINING_SUCCEEDED */
                s->signal_present = 60;
                memcpy(s->eq_coeff_save, s->eq_coeff, sizeof(s->eq_coeff));
                s->carrier_phase_rate_save = s->carrier_phase_rate;
                s->short_train = 1;
                s->training_stage = ST_NORMAL;
            }
            else
            {
                park(s, sink, !s->short_train);
            }
        }
        break;
    default:
        target = zero;
        break;
    }
    orc_sink_qam(sink, z, target, symbol);                      /* v17rx.c:1117-1131 */
}

/* v17rx.c:1133-1210 */
static int signal_detect(orc_v17_t *s, orc_sink_t *sink, int16_t amp)
{
    int16_t diff;
    int16_t x;
    int32_t power;

    x = amp >> 1;
    diff = (int16_t) (x - s->last_sample);
    s->last_sample = x;
    s->power_reading += ((diff*diff - s->power_reading) >> 4);
    power = s->power_reading;
    diff = (int16_t) abs(diff);
    if (10*diff < s->high_sample)
    {
        if (++s->low_samples > 120)
        {
            s->power_reading = 0;
            s->high_sample = 0;
            s->low_samples = 0;
        }
    }
    else
    {
        s->low_samples = 0;
        if (diff > s->high_sample)
            s->high_sample = diff;
    }
    if (s->signal_present > 0)
    {
        if (s->carrier_drop_pending  ||  power < s->carrier_off_power)
        {
            if (--s->signal_present <= 0)
            {
                orc_v17_restart(s, s->bit_rate, s->short_train);
                report_status(sink, -1);
                return 0;
            }
            s->carrier_drop_pending = 1;
        }
    }
    else
    {
        if (power < s->carrier_on_power)
            return 0;
        s->signal_present = 1;
        s->carrier_drop_pending = 0;
        report_status(sink, -2);
    }
    return power;
}

/* v17rx.c:1212-1318 */
int orc_v17_rx(orc_v17_t *s, const int16_t amp[], int len, orc_sink_t *sink)
{
    float z[2];
    float zz[2];
    float sample[2];
    float v;
    int32_t power;
    int root_power;
    int step;
    int i;

    for (i = 0;  i < len;  i++)
    {
        s->rrc_filter[s->rrc_filter_step] = amp[i];
        if (++s->rrc_filter_step >= RRC_LEN)
            s->rrc_filter_step = 0;
        if ((power = signal_detect(s, sink, amp[i])) == 0)
            continue;
        if (s->training_stage == ST_PARKED)
            continue;
        s->eq_put_step -= RRC_SETS;
        step = -s->eq_put_step;
        if (step < 0)
            step += RRC_SETS;
        if (step < 0)
            step = 0;
        else if (step > RRC_SETS - 1)
            step = RRC_SETS - 1;
        v = circular_dot(s->rrc_filter, T.v17_re + step*RRC_LEN, RRC_LEN, s->rrc_filter_step);
        sample[0] = v*s->agc_scaling;
        godard_rx(s, sample[0]);
        if (s->eq_put_step <= 0)
        {
            if (s->agc_scaling_save == 0.0f)
            {
                if ((root_power = fixed_sqrt32((uint32_t) power)) == 0)
                    root_power = 1;
                s->agc_scaling = (2.17f/1.000000f)/root_power;
            }
            v = circular_dot(s->rrc_filter, T.v17_im + step*RRC_LEN, RRC_LEN, s->rrc_filter_step);
            sample[1] = v*s->agc_scaling;
            dds_complex(s->carrier_phase, z);
            zz[0] = sample[0]*z[0] - sample[1]*z[1];
            zz[1] = -sample[0]*z[1] - sample[1]*z[0];
            s->eq_put_step += RRC_SETS*10/(3*2);
            process_half_baud(s, sink, zz);
        }
        s->carrier_phase += (uint32_t) s->carrier_phase_rate;
    }
    return 0;
}
