/*
 * v27ter_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's V.27ter receiver (float build):
 *   v27ter_rx / signal_detect / process_half_baud / decode_baud   src/v27ter_rx.c:441-1028
 *   symbol_sync (Gardner)                                         src/v27ter_rx.c:486-528
 *   find_quadrant / find_octant / descramble                      src/v27ter_rx.c:320-432
 *   equalizer_reset / track_carrier / tune_equalizer              src/v27ter_rx.c:197-313
 *   v27ter_rx_restart / v27ter_rx_init / set_signal_cutoff        src/v27ter_rx.c:161-167, :1091-1195
 * This snapshot of the reference #defines IAXMODEM_STUFF at v27ter_rx.c:1, so the quick
 * power-drop path of signal_detect is part of the behaviour.  v27ter_rx_restart() never
 * stores its old_train argument (v27ter_rx.c:1091-1160 tests s->old_train, which only
 * memset() ever writes), so every restart is a full retrain; that is kept.
 * cosf/sinf of the one-off phase spin (v27ter_rx.c:661-667): the libm restatement in modem_common.h.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"
#include "modem_common.h"

#define T orc_modem_T

#define RRC_LEN         27
#define EQ_LEN          32              /* V27TER_EQUALIZER_LEN */
#define EQ_PRE          16              /* V27TER_EQUALIZER_PRE_LEN */
#define EQ_DELTA        0.25f
#define SETS_4800       8
#define SETS_2400       12
#define SEG_3_LEN       50
#define SEG_5_LEN       1074
#define SEG_6_LEN       8

enum
{
    ST_NORMAL = 0,
    ST_SYMBOL_ACQUISITION,
    ST_LOG_PHASE,
    ST_WAIT_FOR_HOP,
    ST_TRAIN_ON_ABAB,
    ST_TEST_ONES,
    ST_PARKED
};

static const float CONSTEL[8][2] =         /* v27ter_rx.c:125-134 */
{
    { 1.414f,  0.0f}, { 1.0f,  1.0f}, { 0.0f,  1.414f}, {-1.0f,  1.0f},
    {-1.414f,  0.0f}, {-1.0f, -1.0f}, { 0.0f, -1.414f}, { 1.0f, -1.0f}
};

int orc_v27ter_sizeof(void) { return (int) sizeof(orc_v27ter_t); }

static void equalizer_reset(orc_v27ter_t *s)
{
    memset(s->eq_coeff, 0, sizeof(s->eq_coeff));
    s->eq_coeff[EQ_PRE + 1][0] = 1.414f;
    s->eq_coeff[EQ_PRE + 1][1] = 0.0f;
    memset(s->eq_buf, 0, sizeof(s->eq_buf));
    s->eq_delta = EQ_DELTA/EQ_LEN;
    s->eq_put_step = (s->bit_rate == 4800)  ?  SETS_4800*5/2  :  SETS_2400*20/(3*2);
    s->eq_step = 0;
}

static void equalizer_restore(orc_v27ter_t *s)
{
    memcpy(s->eq_coeff, s->eq_coeff_save, sizeof(s->eq_coeff));
    memset(s->eq_buf, 0, sizeof(s->eq_buf));
    s->eq_delta = EQ_DELTA/EQ_LEN;
    s->eq_put_step = (s->bit_rate == 4800)  ?  (SETS_4800*5/2 - 1)  :  (SETS_2400*20/(3*2) - 1);
    s->eq_step = 0;
}

/* v27ter_rx.c:1091-1160 */
int orc_v27ter_restart(orc_v27ter_t *s, int bit_rate, int old_train)
{
    (void) old_train;                       /* the reference ignores it too */
    if (bit_rate != 4800  &&  bit_rate != 2400)
        return -1;
    s->bit_rate = bit_rate;
    memset(s->rrc_filter, 0, sizeof(s->rrc_filter));
    s->training_error = 0.0f;
    s->rrc_filter_step = 0;
    s->scramble_reg = 0x3C;
    s->scrambler_pattern_count = 0;
    s->training_stage = ST_SYMBOL_ACQUISITION;
    s->training_bc = 0;
    s->training_count = 0;
    s->signal_present = 0;
    s->high_sample = 0;
    s->low_samples = 0;
    s->carrier_drop_pending = 0;
    memset(s->diff_angles, 0, sizeof(s->diff_angles));
    s->carrier_phase = 0;
    s->carrier_track_i = 200000.0f;
    s->carrier_track_p = 10000000.0f;
    s->power_reading = 0;
    s->constellation_state = 0;
    if (s->old_train)
    {
        s->carrier_phase_rate = s->carrier_phase_rate_save;
        s->agc_scaling = s->agc_scaling_save;
        equalizer_restore(s);
    }
    else
    {
        s->carrier_phase_rate = (int32_t) (1800.0f*65536.0f*65536.0f/8000);
        s->agc_scaling = (1.414f/1.000000f)/283.0f;
        equalizer_reset(s);
    }
    s->eq_skip = 0;
    s->last_sample = 0;
    s->gardner_integrate = 0;
    s->total_baud_timing_correction = 0;
    s->gardner_step = 512;
    s->baud_half = 0;
    return 0;
}

/* v27ter_rx.c:1162-1190 */
int orc_v27ter_init(orc_v27ter_t *s, int bit_rate)
{
    if (bit_rate != 4800  &&  bit_rate != 2400)
        return -1;
    memset(s, 0, sizeof(*s));
    s->carrier_on_power = (int32_t) (level_dbm0(-45.5f + 2.5f)*0.4f);
    s->carrier_off_power = (int32_t) (level_dbm0(-45.5f - 2.5f)*0.4f);
    return orc_v27ter_restart(s, bit_rate, 0);
}

static void report_status(orc_sink_t *sink, int status)
{
    orc_sink_push(sink, 3, status, 0, 0);
}

static void track_carrier(orc_v27ter_t *s, const float z[2], const float target[2])
{
    float error;

    error = z[1]*target[0] - z[0]*target[1];
    s->carrier_phase_rate += (int32_t) (s->carrier_track_i*error);
    s->carrier_phase += (uint32_t) (int32_t) (s->carrier_track_p*error);
}

static void tune_equalizer(orc_v27ter_t *s, const float z[2], const float target[2])
{
    float err_re;
    float err_im;

    err_re = (target[0] - z[0])*s->eq_delta;
    err_im = (target[1] - z[1])*s->eq_delta;
    ccircular_lms((const float (*)[2]) s->eq_buf, s->eq_coeff, EQ_LEN, s->eq_step, err_re, err_im);
}

/* v27ter_rx.c:320-331 */
static int find_quadrant(const float z[2])
{
    int b1 = (z[1] > z[0]);
    int b2 = (z[1] < -z[0]);

    return (b2 << 1) | (b1 ^ b2);
}

/* v27ter_rx.c:334-377 */
static int find_octant(const float z[2])
{
    float abs_re = fabsf(z[0]);
    float abs_im = fabsf(z[1]);
    int b1;
    int b2;

    if (abs_im*1.0f > abs_re*0.4142136f  &&  abs_im*1.0f < abs_re*2.4142136f)
    {
        b1 = (z[0] < 0.0f);
        b2 = (z[1] < 0.0f);
        return (b2 << 2) | ((b1 ^ b2) << 1) | 1;
    }
    b1 = (z[1] > z[0]);
    b2 = (z[1] < -z[0]);
    return (b2 << 2) | ((b1 ^ b2) << 1);
}

/* v27ter_rx.c:380-414 */
static int descramble(orc_v27ter_t *s, int in_bit)
{
    int out_bit;
    int training = (s->training_stage > ST_NORMAL  &&  s->training_stage < ST_TEST_ONES);

    in_bit &= 1;
    out_bit = (in_bit ^ (s->scramble_reg >> 5) ^ (s->scramble_reg >> 6)) & 1;
    if (s->scrambler_pattern_count >= 33)
    {
        out_bit ^= 1;
        s->scrambler_pattern_count = 0;
    }
    else
    {
        if (training)
        {
            s->scrambler_pattern_count = 0;
        }
        else
        {
            if ((((s->scramble_reg >> 7) ^ in_bit) & ((s->scramble_reg >> 8) ^ in_bit) & ((s->scramble_reg >> 11) ^ in_bit) & 1))
                s->scrambler_pattern_count = 0;
            else
                s->scrambler_pattern_count++;
        }
    }
    s->scramble_reg <<= 1;
    if (training)
        s->scramble_reg |= out_bit;
    else
        s->scramble_reg |= in_bit;
    return out_bit;
}

static void put_bit(orc_v27ter_t *s, orc_sink_t *sink, int bit)
{
    int out_bit = descramble(s, bit);

    if (s->training_stage == ST_NORMAL)
        orc_sink_push(sink, 3, out_bit, 0, 0);
}

/* v27ter_rx.c:441-484 */
static void decode_baud(orc_v27ter_t *s, orc_sink_t *sink, const float z[2])
{
    static const uint8_t phase_steps_4800[8] = {4, 0, 2, 6, 7, 3, 1, 5};
    static const uint8_t phase_steps_2400[4] = {0, 2, 3, 1};
    int nearest;
    int raw_bits;

    if (s->bit_rate == 2400)
    {
        nearest = find_quadrant(z);
        raw_bits = phase_steps_2400[(nearest - s->constellation_state) & 3];
        put_bit(s, sink, raw_bits);
        put_bit(s, sink, raw_bits >> 1);
        s->constellation_state = nearest;
        nearest <<= 1;
    }
    else
    {
        nearest = find_octant(z);
        raw_bits = phase_steps_4800[(nearest - s->constellation_state) & 7];
        put_bit(s, sink, raw_bits);
        put_bit(s, sink, raw_bits >> 1);
        put_bit(s, sink, raw_bits >> 2);
        s->constellation_state = nearest;
    }
    track_carrier(s, z, CONSTEL[nearest]);
    if (--s->eq_skip <= 0)
    {
        s->eq_skip = 100;
        tune_equalizer(s, z, CONSTEL[nearest]);
    }
}

/* v27ter_rx.c:486-528 */
static void symbol_sync(orc_v27ter_t *s, orc_sink_t *sink)
{
    float p;
    float q;

    p = s->eq_buf[(s->eq_step - 3) & (EQ_LEN - 1)][0] - s->eq_buf[(s->eq_step - 1) & (EQ_LEN - 1)][0];
    p *= s->eq_buf[(s->eq_step - 2) & (EQ_LEN - 1)][0];
    q = s->eq_buf[(s->eq_step - 3) & (EQ_LEN - 1)][1] - s->eq_buf[(s->eq_step - 1) & (EQ_LEN - 1)][1];
    q *= s->eq_buf[(s->eq_step - 2) & (EQ_LEN - 1)][1];
    s->gardner_integrate += (p + q > 0)  ?  s->gardner_step  :  -s->gardner_step;
    if (abs(s->gardner_integrate) >= 128)
    {
        s->eq_put_step += (s->gardner_integrate/128);
        s->total_baud_timing_correction += (s->gardner_integrate/128);
        orc_sink_qam(sink, NULL, NULL, s->gardner_integrate);       /* v27ter_rx.c:517-518 */
        s->gardner_integrate = 0;
    }
}

/* v27ter_rx.c:531-777 */
static void process_half_baud(orc_v27ter_t *s, orc_sink_t *sink, const float sample[2])
{
    static const int abab_pos[2] = {0, 4};
    static const float zero[2] = {0.0f, 0.0f};
    const float *target;
    float z[2];
    float zz[2];
    float p;
    float t;
    int i;
    int j;
    int32_t angle;
    int32_t ang;
    int cs;

    s->eq_buf[s->eq_step][0] = sample[0];
    s->eq_buf[s->eq_step][1] = sample[1];
    if (++s->eq_step >= EQ_LEN)
        s->eq_step = 0;
    if ((s->baud_half ^= 1))
        return;
    symbol_sync(s, sink);
    ccircular_dot((const float (*)[2]) s->eq_buf, (const float (*)[2]) s->eq_coeff, EQ_LEN, s->eq_step, z);

    switch (s->training_stage)
    {
    case ST_NORMAL:
        decode_baud(s, sink, z);
        cs = (s->bit_rate == 4800)  ?  s->constellation_state  :  (s->constellation_state << 1);
        target = CONSTEL[cs];
        break;
    case ST_SYMBOL_ACQUISITION:
        target = zero;
        if (++s->training_count >= 30)
        {
            s->gardner_step = 32;
            s->training_stage = ST_LOG_PHASE;
            memset(s->diff_angles, 0, sizeof(s->diff_angles));
            s->last_angles[0] = arctan2_i(z[1], z[0]);
        }
        break;
    case ST_LOG_PHASE:
        target = zero;
        s->last_angles[1] = arctan2_i(z[1], z[0]);
        s->training_count = 1;
        s->training_stage = ST_WAIT_FOR_HOP;
        break;
    case ST_WAIT_FOR_HOP:
        target = zero;
        angle = arctan2_i(z[1], z[0]);
        i = s->training_count + 1;
        ang = (int32_t) ((uint32_t) angle - (uint32_t) s->last_angles[i & 1]);
        s->last_angles[i & 1] = angle;
        s->diff_angles[i & 0xF] = (int32_t) ((uint32_t) s->diff_angles[(i - 2) & 0xF] + (uint32_t) (ang >> 4));
        if ((ang > 0x20000000  ||  ang < (int32_t) 0xE0000000u)  &&  s->training_count >= 13)
        {
            i = (s->training_count - 8) & ~1;
            if (i > 1)
            {
                j = i & 0xF;
                ang = (int32_t) ((uint32_t) s->diff_angles[j] + (uint32_t) s->diff_angles[j | 0x1])/(i - 1);
                if (s->bit_rate == 4800)
                    s->carrier_phase_rate += 16*(ang/10);
                else
                    s->carrier_phase_rate += 3*16*(ang/40);
            }
            if (s->carrier_phase_rate < (int32_t) ((1800.0f - 20.0f)*65536.0f*65536.0f/8000)
                ||
                s->carrier_phase_rate > (int32_t) ((1800.0f + 20.0f)*65536.0f*65536.0f/8000))
            {
                s->training_stage = ST_PARKED;
                report_status(sink, -5);                    /* SIG_STATUS_TRAINING_FAILED */
                break;
            }
            angle = (int32_t) ((uint32_t) angle + 0x80000000u);     /* DDS_PHASE(180.0f) */
            p = ((uint32_t) angle)*2.0f*3.1415926f/(65536.0f*65536.0f);     /* dds_phase_to_radians */
            zz[0] = orc_cosf(p);
            zz[1] = -orc_sinf(p);
            for (i = 0;  i < EQ_LEN;  i++)
            {
                t = s->eq_buf[i][0]*zz[0] - s->eq_buf[i][1]*zz[1];
                s->eq_buf[i][1] = s->eq_buf[i][0]*zz[1] + s->eq_buf[i][1]*zz[0];
                s->eq_buf[i][0] = t;
            }
            s->carrier_phase += (uint32_t) angle;
            s->gardner_step = 2;
            s->training_bc = 1;
            s->training_bc ^= descramble(s, 1);
            descramble(s, 1);
            descramble(s, 1);
            s->constellation_state = abab_pos[s->training_bc];
            target = CONSTEL[s->constellation_state];
            s->training_count = 1;
            s->training_stage = ST_TRAIN_ON_ABAB;
            report_status(sink, -3);                        /* SIG_STATUS_TRAINING_IN_PROGRESS */
        }
        else if (++s->training_count > SEG_3_LEN)
        {
            s->training_stage = ST_PARKED;
            report_status(sink, -5);
        }
        break;
    case ST_TRAIN_ON_ABAB:
        s->training_bc ^= descramble(s, 1);
        descramble(s, 1);
        descramble(s, 1);
        s->constellation_state = abab_pos[s->training_bc];
        target = CONSTEL[s->constellation_state];
        track_carrier(s, z, target);
        tune_equalizer(s, z, target);
        s->carrier_track_i = 400.0f + (200000.0f - 400.0f)*(float) (SEG_5_LEN - s->training_count)/(float) SEG_5_LEN;
        s->carrier_track_p = 1000000.0f + (10000000.0f - 1000000.0f)*(float) (SEG_5_LEN - s->training_count)/(float) SEG_5_LEN;
        if (++s->training_count >= SEG_5_LEN)
        {
            s->constellation_state = (s->bit_rate == 4800)  ?  4  :  2;
            s->training_count = 0;
            s->training_stage = ST_TEST_ONES;
        }
        break;
    case ST_TEST_ONES:
        decode_baud(s, sink, z);
        cs = (s->bit_rate == 4800)  ?  s->constellation_state  :  (s->constellation_state << 1);
        target = CONSTEL[cs];
        zz[0] = z[0] - CONSTEL[cs][0];
        zz[1] = z[1] - CONSTEL[cs][1];
        s->training_error += (zz[0]*zz[0] + zz[1]*zz[1]);
        if (++s->training_count >= SEG_6_LEN)
        {
            if ((s->bit_rate == 4800  &&  s->training_error < (float) SEG_6_LEN*0.25f)
                ||
                (s->bit_rate == 2400  &&  s->training_error < (float) SEG_6_LEN*0.5f))
            {
                report_status(sink, -4);                    /* SIG_STATUS_TRAINING_SUCCEEDED */
                s->signal_present = (s->bit_rate == 4800)  ?  90  :  120;
                s->training_stage = ST_NORMAL;
                memcpy(s->eq_coeff_save, s->eq_coeff, sizeof(s->eq_coeff));
                s->carrier_phase_rate_save = s->carrier_phase_rate;
                s->agc_scaling_save = s->agc_scaling;
            }
            else
            {
                s->training_stage = ST_PARKED;
                report_status(sink, -5);
            }
        }
        break;
    default:
        target = zero;
        break;
    }
    orc_sink_qam(sink, z, target, s->constellation_state);      /* v27ter_rx.c:765-777 */
}

/* v27ter_rx.c:779-861 */
static int signal_detect(orc_v27ter_t *s, orc_sink_t *sink, int16_t amp)
{
    int16_t diff;
    int16_t x;
    int32_t power;

    x = amp >> 1;
    diff = (int16_t) (x - s->last_sample);
    s->last_sample = x;
    s->power_reading += ((diff*diff - s->power_reading) >> 4);      /* power_meter_update, shift 4 */
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
                orc_v27ter_restart(s, s->bit_rate, 0);
                report_status(sink, -1);                    /* SIG_STATUS_CARRIER_DOWN */
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
        report_status(sink, -2);                            /* SIG_STATUS_CARRIER_UP */
    }
    return power;
}

/* v27ter_rx.c:863-1028 (the two bit-rate branches differ only in these constants) */
int orc_v27ter_rx(orc_v27ter_t *s, const int16_t amp[], int len, orc_sink_t *sink)
{
    const int sets = (s->bit_rate == 4800)  ?  SETS_4800  :  SETS_2400;
    const int put_add = (s->bit_rate == 4800)  ?  SETS_4800*5/2  :  SETS_2400*20/(3*2);
    const float *tre = (s->bit_rate == 4800)  ?  T.v27_4800_re  :  T.v27_2400_re;
    const float *tim = (s->bit_rate == 4800)  ?  T.v27_4800_im  :  T.v27_2400_im;
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
        if ((s->eq_put_step -= sets) <= 0)
        {
            if (s->training_stage == ST_SYMBOL_ACQUISITION)
            {
                if ((root_power = fixed_sqrt32((uint32_t) power)) == 0)
                    root_power = 1;
                s->agc_scaling = (1.414f/1.000000f)/root_power;
            }
            step = -s->eq_put_step;
            if (step > sets - 1)
                step = sets - 1;
            v = circular_dot(s->rrc_filter, tre + step*RRC_LEN, RRC_LEN, s->rrc_filter_step);
            sample[0] = v*s->agc_scaling;
            v = circular_dot(s->rrc_filter, tim + step*RRC_LEN, RRC_LEN, s->rrc_filter_step);
            sample[1] = v*s->agc_scaling;
            dds_complex(s->carrier_phase, z);
            zz[0] = sample[0]*z[0] - sample[1]*z[1];
            zz[1] = -sample[0]*z[1] - sample[1]*z[0];
            s->eq_put_step += put_add;
            process_half_baud(s, sink, zz);
        }
        s->carrier_phase += (uint32_t) s->carrier_phase_rate;
    }
    return 0;
}
