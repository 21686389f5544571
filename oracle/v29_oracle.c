/*
 * v29_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's V.29 receiver and the primitives under it:
 *   v29_rx / signal_detect / process_half_baud / decode_baud   src/v29rx.c:400-965
 *   track_carrier, tune_equalizer, equalizer_reset/save/restore src/v29rx.c:197-331
 *   descramble, scrambled_training_bit                          src/v29rx.c:350-378
 *   v29_rx_restart / v29_rx_init / v29_rx_set_signal_cutoff     src/v29rx.c:163-169, :1019-1131
 *   vec_circular_dot_prodf                                      src/vector_float.c:890-939
 *   cvec_circular_dot_prodf, cvec_circular_lmsf                 src/complex_vector_float.c:137-219
 *   godard_ted_rx / godard_ted_per_baud                         src/godard.c:144-220
 *   power_meter_update, power_meter_level_dbm0                  src/power_meter.c:65-92
 *   fixed_sqrt32                                                src/math_fixed.c:158-169
 *   dds_lookup_complexf, dds_phase_to_radians                   src/dds_float.c:2103,2135,2177
 *   arctan2                                                     src/spandsp/arctan2.h:47-80
 *
 * Tables (polyphase RRC, sine, sqrt, Godard descriptor) are handed in by the test harness
 * (orc_modem_set_tables); they are data taken from the reference build (tests/golden/
 * modem_tables.npz) -- the product builds its own and is tested against the same data.
 *
 * The phase "spin" at the end of training (v29rx.c:618-623) calls libm's cosf/sinf: restated in
 * modem_common.h (glibc's sincosf algorithm, verified against libm over the whole argument range),
 * so this file, the device code and the reference build all produce the same bits.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"
#include "modem_common.h"

orc_modem_tables_t orc_modem_T;
#define T orc_modem_T

void orc_modem_set_tables(const orc_modem_tables_t *t)
{
    T = *t;
}

#define RRC_SETS        48              /* RX_PULSESHAPER_COEFF_SETS, generated v29rx_rrc.h */
#define RRC_LEN         27              /* V29_RX_FILTER_STEPS, private/v29rx.h */
#define EQ_LEN          33              /* V29_EQUALIZER_LEN */
#define EQ_PRE          16              /* V29_EQUALIZER_PRE_LEN */
#define EQ_DELTA        0.21f           /* v29rx.c:97 */
#define LMS_LEAK        0.9999f         /* complex_vector_float.c:199 */

enum
{
    ST_NORMAL = 0,
    ST_SYMBOL_ACQUISITION,
    ST_LOG_PHASE,
    ST_WAIT_FOR_CDCD,
    ST_TRAIN_ON_CDCD,
    ST_TRAIN_ON_CDCD_AND_TEST,
    ST_TEST_ONES,
    ST_PARKED
};

static const float CONSTEL[16][2] =        /* v29tx_constellation_maps.h:58-77 */
{
    { 3.0f,  0.0f}, { 1.0f,  1.0f}, { 0.0f,  3.0f}, {-1.0f,  1.0f},
    {-3.0f,  0.0f}, {-1.0f, -1.0f}, { 0.0f, -3.0f}, { 1.0f, -1.0f},
    { 5.0f,  0.0f}, { 3.0f,  3.0f}, { 0.0f,  5.0f}, {-3.0f,  3.0f},
    {-5.0f,  0.0f}, {-3.0f, -3.0f}, { 0.0f, -5.0f}, { 3.0f, -3.0f}
};

/* v29rx.c:119-143, packed as one row per string: value = char - 'a' */
static const char *SPACE_MAP[20] =
{
    "nnnnnnmmmmmmmmllllll", "nnnnnnnmmmmmmlllllll", "nnnnnnneeeeeelllllll", "nnnnnnneeeeeelllllll",
    "nnnnnnneeeeeelllllll", "nnnnnnnfeeeedlllllll", "onnnnnffffddddlllllk", "oogggfffffdddddccckk",
    "ooggggffffddddcccckk", "ooggggffffddddcccckk", "oogggghhhhbbbbcccckk", "oogggghhhhbbbbcccckk",
    "ooggghhhhhbbbbbccckk", "oppppphhhhbbbbjjjjjk", "ppppppphaaaabjjjjjjj", "pppppppaaaaaajjjjjjj",
    "pppppppaaaaaajjjjjjj", "pppppppaaaaaajjjjjjj", "pppppppiiiiiijjjjjjj", "ppppppiiiiiiiijjjjjj"
};

int orc_v29_sizeof(void) { return (int) sizeof(orc_v29_t); }

/* test hooks for the libm restatement in modem_common.h */
float orc_trig_cosf(float x) { return orc_cosf(x); }
float orc_trig_sinf(float x) { return orc_sinf(x); }

/* v29rx.c:163-169 */
void orc_v29_set_signal_cutoff(orc_v29_t *s, float cutoff)
{
    s->carrier_on_power = (int32_t) (level_dbm0(cutoff + 2.5f)*0.4f);
    s->carrier_off_power = (int32_t) (level_dbm0(cutoff - 2.5f)*0.4f);
}

/* v29rx.c:223-246 / :204-221 */
static void equalizer_reset(orc_v29_t *s)
{
    memset(s->eq_coeff, 0, sizeof(s->eq_coeff));
    s->eq_coeff[EQ_PRE][0] = 3.0f;
    s->eq_coeff[EQ_PRE][1] = 0.0f;
    memset(s->eq_buf, 0, sizeof(s->eq_buf));
    s->eq_delta = EQ_DELTA/EQ_LEN;
    s->eq_put_step = RRC_SETS*10/(3*2) - 1;
    s->eq_step = 0;
}

static void equalizer_restore(orc_v29_t *s)
{
    memcpy(s->eq_coeff, s->eq_coeff_save, sizeof(s->eq_coeff));
    memset(s->eq_buf, 0, sizeof(s->eq_buf));
    s->eq_delta = EQ_DELTA/EQ_LEN;
    s->eq_put_step = RRC_SETS*10/(3*2) - 1;
    s->eq_step = 0;
}

/* v29rx.c:1019-1098 */
int orc_v29_restart(orc_v29_t *s, int bit_rate, int old_train)
{
    switch (bit_rate)
    {
    case 9600:
        s->training_cd = 0;
        break;
    case 7200:
        s->training_cd = 2;
        break;
    case 4800:
        s->training_cd = 4;
        break;
    default:
        return -1;
    }
    s->bit_rate = bit_rate;
    memset(s->rrc_filter, 0, sizeof(s->rrc_filter));
    s->rrc_filter_step = 0;
    s->scramble_reg = 0;
    s->training_scramble_reg = 0x2A;
    s->training_stage = ST_SYMBOL_ACQUISITION;
    s->training_count = 0;
    s->signal_present = 0;
    s->high_sample = 0;                     /* IAXMODEM_STUFF is #defined at v29rx.c:1 */
    s->low_samples = 0;
    s->carrier_drop_pending = 0;
    s->old_train = old_train;
    memset(s->diff_angles, 0, sizeof(s->diff_angles));
    s->carrier_phase = 0;
    s->power_reading = 0;                   /* power_meter_init(&s->power, 4) */
    s->constellation_state = 0;
    if (s->old_train)
    {
        s->carrier_phase_rate = s->carrier_phase_rate_save;
        equalizer_restore(s);
        s->agc_scaling = s->agc_scaling_save;
    }
    else
    {
        s->carrier_phase_rate = (int32_t) (1700.0f*65536.0f*65536.0f/8000);     /* DDS_PHASE_RATE, dds.h:31 */
        equalizer_reset(s);
        s->agc_scaling_save = 0.0f;
        s->agc_scaling = (1.25f/1.0f)/735.0f;
    }
    s->carrier_track_i = 8000.0f;
    s->carrier_track_p = 8000000.0f;
    s->last_sample = 0;
    s->eq_skip = 0;
    /* godard_ted_init(), godard.c:223-236 */
    memset(s->g_low, 0, sizeof(s->g_low));
    memset(s->g_high, 0, sizeof(s->g_high));
    memset(s->g_dc, 0, sizeof(s->g_dc));
    s->g_baud_phase = 0.0f;
    s->g_total_correction = 0;
    s->baud_half = 0;
    return 0;
}

/* v29rx.c:1100-1131 */
int orc_v29_init(orc_v29_t *s, int bit_rate)
{
    if (bit_rate != 9600  &&  bit_rate != 7200  &&  bit_rate != 4800)
        return -1;
    memset(s, 0, sizeof(*s));
    orc_v29_set_signal_cutoff(s, -28.5f);
    return orc_v29_restart(s, bit_rate, 0);
}

static void report_status(orc_sink_t *sink, int status)
{
    orc_sink_push(sink, 3, status, 0, 0);       /* v29rx.c:171-178: status through put_bit */
}

/* godard.c:144-162 */
static void godard_rx(orc_v29_t *s, float sample)
{
    float v;

    v = s->g_low[0]*T.godard[0] + s->g_low[1]*T.godard[1] + sample;
    s->g_low[1] = s->g_low[0];
    s->g_low[0] = v;
    v = s->g_high[0]*T.godard[3] + s->g_high[1]*T.godard[4] + sample;
    s->g_high[1] = s->g_high[0];
    s->g_high[0] = v;
}

/* godard.c:165-220 (fine trigger 30, coarse trigger 1000, steps 1 / 5) */
static int godard_per_baud(orc_v29_t *s)
{
    float v;
    float p;
    int i;

    v = s->g_low[1]*s->g_high[0]*T.godard[2]
      - s->g_low[0]*s->g_high[1]*T.godard[5]
      + s->g_low[1]*s->g_high[1]*T.godard[6];
    p = v - s->g_dc[1];
    s->g_dc[1] = s->g_dc[0];
    s->g_dc[0] = v;
    s->g_baud_phase -= p;
    v = fabsf(s->g_baud_phase);
    if (v > T.godard_fine_trigger)
    {
        i = (v > T.godard_coarse_trigger)  ?  T.godard_coarse_step  :  T.godard_fine_step;
        if (s->g_baud_phase < 0.0f)
            i = -i;
        s->g_total_correction += i;
        return i;
    }
    return 0;
}

/* v29rx.c:297-331 */
static void track_carrier(orc_v29_t *s, const float z[2], const float target[2])
{
    float error;

    error = z[1]*target[0] - z[0]*target[1];
    s->carrier_phase_rate += (int32_t) (s->carrier_track_i*error);
    s->carrier_phase += (uint32_t) (int32_t) (s->carrier_track_p*error);
}

/* v29rx.c:281-291 + cvec_circular_lmsf (complex_vector_float.c:201-219) */
static void tune_equalizer(orc_v29_t *s, const float z[2], const float target[2])
{
    float err_re;
    float err_im;
    int pos = s->eq_step;
    int i;
    int k;

    err_re = (target[0] - z[0])*s->eq_delta;
    err_im = (target[1] - z[1])*s->eq_delta;
    for (i = 0;  i < EQ_LEN;  i++)
    {
        k = pos + i;
        if (k >= EQ_LEN)
            k -= EQ_LEN;
        /* y[i] pairs with x[(pos + i) mod n] */
        s->eq_coeff[i][0] = s->eq_coeff[i][0]*LMS_LEAK + (s->eq_buf[k][1]*err_im + s->eq_buf[k][0]*err_re);
        s->eq_coeff[i][1] = s->eq_coeff[i][1]*LMS_LEAK + (s->eq_buf[k][0]*err_im - s->eq_buf[k][1]*err_re);
    }
}

/* cvec_circular_dot_prodf, complex_vector_float.c:137-196 */
static void equalizer_get(const orc_v29_t *s, float z[2])
{
    float a_re = 0.0f;
    float a_im = 0.0f;
    float b_re = 0.0f;
    float b_im = 0.0f;
    int pos = s->eq_step;
    int i;

    for (i = 0;  i < EQ_LEN - pos;  i++)
    {
        a_re += (s->eq_buf[pos + i][0]*s->eq_coeff[i][0] - s->eq_buf[pos + i][1]*s->eq_coeff[i][1]);
        a_im += (s->eq_buf[pos + i][0]*s->eq_coeff[i][1] + s->eq_buf[pos + i][1]*s->eq_coeff[i][0]);
    }
    for (i = 0;  i < pos;  i++)
    {
        b_re += (s->eq_buf[i][0]*s->eq_coeff[EQ_LEN - pos + i][0] - s->eq_buf[i][1]*s->eq_coeff[EQ_LEN - pos + i][1]);
        b_im += (s->eq_buf[i][0]*s->eq_coeff[EQ_LEN - pos + i][1] + s->eq_buf[i][1]*s->eq_coeff[EQ_LEN - pos + i][0]);
    }
    z[0] = a_re + b_re;
    z[1] = a_im + b_im;
}

/* v29rx.c:350-362 */
static int scrambled_training_bit(orc_v29_t *s)
{
    int bit;

    bit = s->training_scramble_reg & 1;
    s->training_scramble_reg >>= 1;
    if (bit ^ (s->training_scramble_reg & 1))
        s->training_scramble_reg |= 0x40;
    return bit;
}

/* v29rx.c:365-397 */
static void put_bit(orc_v29_t *s, orc_sink_t *sink, int bit)
{
    int out_bit;

    bit &= 1;
    out_bit = (bit ^ (s->scramble_reg >> (18 - 1)) ^ (s->scramble_reg >> (23 - 1))) & 1;
    s->scramble_reg = (s->scramble_reg << 1) | bit;
    if (s->training_stage == ST_NORMAL)
        orc_sink_push(sink, 3, out_bit, 0, 0);
}

/* v29rx.c:400-481 */
static void decode_baud(orc_v29_t *s, orc_sink_t *sink, const float z[2])
{
    static const uint8_t phase_steps_9600[8] = {4, 0, 2, 6, 7, 3, 1, 5};
    static const uint8_t phase_steps_4800[4] = {0, 2, 3, 1};
    int nearest;
    int raw_bits;
    int i;
    int re;
    int im;
    int b1;
    int b2;

    if (s->bit_rate == 4800)
    {
        b1 = (z[1] > z[0]);
        b2 = (z[1] < -z[0]);
        nearest = ((b2 << 1) | (b1 ^ b2)) << 1;
        raw_bits = phase_steps_4800[((nearest - s->constellation_state) >> 1) & 3];
        put_bit(s, sink, raw_bits);
        put_bit(s, sink, raw_bits >> 1);
    }
    else
    {
        re = (int) ((z[0] + 5.0f)*2.0f);
        im = (int) ((z[1] + 5.0f)*2.0f);
        if (re > 19)
            re = 19;
        else if (re < 0)
            re = 0;
        if (im > 19)
            im = 19;
        else if (im < 0)
            im = 0;
        nearest = SPACE_MAP[re][im] - 'a';
        if (s->bit_rate == 9600)
            put_bit(s, sink, nearest >> 3);
        else
            nearest &= 7;
        raw_bits = phase_steps_9600[(nearest - s->constellation_state) & 7];
        for (i = 0;  i < 3;  i++)
        {
            put_bit(s, sink, raw_bits);
            raw_bits >>= 1;
        }
    }
    track_carrier(s, z, CONSTEL[nearest]);
    if (--s->eq_skip <= 0)
    {
        s->eq_skip = 10;
        tune_equalizer(s, z, CONSTEL[nearest]);
    }
    s->constellation_state = nearest;
}

static void park(orc_v29_t *s, orc_sink_t *sink)
{
    s->agc_scaling_save = 0.0f;
    s->training_stage = ST_PARKED;
    report_status(sink, -5);                        /* SIG_STATUS_TRAINING_FAILED */
}

/* v29rx.c:484-786 */
static void process_half_baud(orc_v29_t *s, orc_sink_t *sink, const float sample[2])
{
    static const int cdcd_pos[6] = {0, 11, 0, 3, 0, 2};
    static const float zero[2] = {0.0f, 0.0f};
    float z[2];
    float zz[2];
    float p;
    float c;
    float sn;
    float t;
    const float *target;
    int bit;
    int i;
    int j;
    int32_t angle;
    int32_t ang;

    s->eq_buf[s->eq_step][0] = sample[0];
    s->eq_buf[s->eq_step][1] = sample[1];
    if (++s->eq_step >= EQ_LEN)
        s->eq_step = 0;
    if ((s->baud_half ^= 1))
        return;
    s->eq_put_step += godard_per_baud(s);
    equalizer_get(s, z);

    switch (s->training_stage)
    {
    case ST_NORMAL:
        decode_baud(s, sink, z);
        target = CONSTEL[s->constellation_state];
        break;
    case ST_SYMBOL_ACQUISITION:
        target = zero;
        if (++s->training_count >= 60)
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
        s->last_angles[1] = arctan2_i(z[1], z[0]);
        s->training_count = 1;
        s->training_stage = ST_WAIT_FOR_CDCD;
        break;
    case ST_WAIT_FOR_CDCD:
        target = zero;
        angle = arctan2_i(z[1], z[0]);
        i = s->training_count + 1;
        ang = (int32_t) ((uint32_t) angle - (uint32_t) s->last_angles[i & 1]);
        s->last_angles[i & 1] = angle;
        s->diff_angles[i & 0xF] = (int32_t) ((uint32_t) s->diff_angles[(i - 2) & 0xF] + (uint32_t) (ang >> 4));
        /* DDS_PHASE(45.0f) = 0x20000000, DDS_PHASE(-45.0f) = (int32_t) 0xE0000000 (dds.h:32) */
        if ((ang > 0x20000000  ||  ang < (int32_t) 0xE0000000u)  &&  s->training_count >= 13)
        {
            i = (s->training_count - 8) & ~1;
            if (i > 1)
            {
                j = i & 0xF;
                ang = (int32_t) ((uint32_t) s->diff_angles[j] + (uint32_t) s->diff_angles[j | 0x1])/(i - 1);
                s->carrier_phase_rate += 3*16*(ang/20);
            }
            /* plausibility: +-20 Hz around 1700 Hz (v29rx.c:596-598) */
            if (s->carrier_phase_rate < (int32_t) ((1700.0f - 20.0f)*65536.0f*65536.0f/8000)
                ||  s->carrier_phase_rate > (int32_t) ((1700.0f + 20.0f)*65536.0f*65536.0f/8000))
            {
                park(s, sink);
                break;
            }
            /* spin the equaliser buffer and the carrier (v29rx.c:618-624); see the header note on cos/sin */
            p = ((uint32_t) angle)*2.0f*3.1415926f/(65536.0f*65536.0f);        /* dds_phase_to_radians */
            c = orc_cosf(p);
            sn = -orc_sinf(p);
            zz[0] = c;
            zz[1] = sn;
            for (i = 0;  i < EQ_LEN;  i++)
            {
                t = s->eq_buf[i][0]*zz[0] - s->eq_buf[i][1]*zz[1];
                s->eq_buf[i][1] = s->eq_buf[i][0]*zz[1] + s->eq_buf[i][1]*zz[0];
                s->eq_buf[i][0] = t;
            }
            s->carrier_phase += (uint32_t) angle;
            bit = scrambled_training_bit(s);
            s->constellation_state = cdcd_pos[s->training_cd + bit];
            target = CONSTEL[s->constellation_state];
            s->training_count = 1;
            s->training_stage = ST_TRAIN_ON_CDCD;
            report_status(sink, -3);                /* SIG_STATUS_TRAINING_IN_PROGRESS */
            break;
        }
        if (++s->training_count > 128)              /* V29_TRAINING_SEG_2_LEN */
            park(s, sink);
        break;
    case ST_TRAIN_ON_CDCD:
        bit = scrambled_training_bit(s);
        s->constellation_state = cdcd_pos[s->training_cd + bit];
        target = CONSTEL[s->constellation_state];
        track_carrier(s, z, target);
        tune_equalizer(s, z, target);
        if (++s->training_count >= 384 - 48)
        {
            s->training_stage = ST_TRAIN_ON_CDCD_AND_TEST;
            s->training_error = 0.0f;
            s->carrier_track_i = 200.0f;
            s->carrier_track_p = 1000000.0f;
        }
        break;
    case ST_TRAIN_ON_CDCD_AND_TEST:
        bit = scrambled_training_bit(s);
        s->constellation_state = cdcd_pos[s->training_cd + bit];
        target = CONSTEL[s->constellation_state];
        track_carrier(s, z, target);
        tune_equalizer(s, z, target);
        zz[0] = z[0] - target[0];
        zz[1] = z[1] - target[1];
        s->training_error += zz[0]*zz[0] + zz[1]*zz[1];
        if (++s->training_count >= 384)
        {
            if (s->training_error < 48.0f*2.0f)
            {
                s->training_error = 0.0f;
                s->training_count = 0;
                s->constellation_state = 0;
                s->training_stage = ST_TEST_ONES;
            }
            else
            {
                park(s, sink);
            }
        }
        break;
    case ST_TEST_ONES:
        decode_baud(s, sink, z);
        target = CONSTEL[s->constellation_state];
        zz[0] = z[0] - target[0];
        zz[1] = z[1] - target[1];
        s->training_error += zz[0]*zz[0] + zz[1]*zz[1];
        if (++s->training_count >= 48)
        {
            if (s->training_error < 48.0f*1.0f)
            {
                report_status(sink, -4);            /* SIG_STATUS_TRAINING_SUCCEEDED */
                s->signal_present = 60;
                s->training_stage = ST_NORMAL;
                memcpy(s->eq_coeff_save, s->eq_coeff, sizeof(s->eq_coeff));
                s->carrier_phase_rate_save = s->carrier_phase_rate;
                s->agc_scaling_save = s->agc_scaling;
            }
            else
            {
                park(s, sink);
            }
        }
        break;
    case ST_PARKED:
    default:
        target = zero;
        break;
    }
    orc_sink_qam(sink, z, target, s->constellation_state);      /* v29rx.c:769-783 */
}

/* v29rx.c:788-865.  Returns the power, 0 meaning "skip this sample". */
static int32_t signal_detect(orc_v29_t *s, orc_sink_t *sink, int16_t amp)
{
    int16_t diff;
    int16_t x;
    int32_t power;

    x = amp >> 1;
    diff = (int16_t) (x - s->last_sample);
    s->last_sample = x;
    s->power_reading += ((diff*diff - s->power_reading) >> 4);      /* power_meter_update, shift 4 */
    power = s->power_reading;
    /* "Quick power drop fudge", v29rx.c:802-823 (IAXMODEM_STUFF is #defined at v29rx.c:1) */
    diff = (int16_t) abs(diff);
    if (10*diff < s->high_sample)
    {
        if (++s->low_samples > 120)
        {
            s->power_reading = 0;                   /* power_meter_init(&s->power, 4) */
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
                orc_v29_restart(s, s->bit_rate, 0);
                report_status(sink, -1);            /* SIG_STATUS_CARRIER_DOWN */
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
        report_status(sink, -2);                    /* SIG_STATUS_CARRIER_UP */
    }
    return power;
}

/* v29rx.c:867-965 */
int orc_v29_rx(orc_v29_t *s, const int16_t amp[], int len, orc_sink_t *sink)
{
    float v;
    float sample[2];
    float zz[2];
    float zre;
    float zim;
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
        v = circular_dot(s->rrc_filter, T.rrc_re + step*RRC_LEN, RRC_LEN, s->rrc_filter_step);
        sample[0] = v*s->agc_scaling;
        godard_rx(s, sample[0]);
        if (s->eq_put_step <= 0)
        {
            if (s->agc_scaling_save == 0.0f)
            {
                if ((root_power = fixed_sqrt32((uint32_t) power)) == 0)
                    root_power = 1;
                s->agc_scaling = (1.25f/1.0f)/root_power;
            }
            v = circular_dot(s->rrc_filter, T.rrc_im + step*RRC_LEN, RRC_LEN, s->rrc_filter_step);
            sample[1] = v*s->agc_scaling;
            /* dds_lookup_complexf(), dds_float.c:2135,2177 */
            zre = T.sine[(uint32_t) (s->carrier_phase + (1u << 30)) >> (32 - 11)];
            zim = T.sine[s->carrier_phase >> (32 - 11)];
            zz[0] = sample[0]*zre - sample[1]*zim;
            zz[1] = -sample[0]*zim - sample[1]*zre;
            s->eq_put_step += RRC_SETS*10/(3*2);
            process_half_baud(s, sink, zz);
        }
        s->carrier_phase += (uint32_t) s->carrier_phase_rate;
    }
    return 0;
}
