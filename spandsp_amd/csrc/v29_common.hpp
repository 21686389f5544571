// v29_common.hpp -- what the two V.29 receiver kernels (v29_dev.hpp: one channel per lane; v29_quad.hpp: four lanes per
// channel) and the host-side lane emulator of the tests (tests/emul) share: the state word map, the constant tables, the
// launch record and the scalar helpers restated from the reference (citations at each).
#pragma once

#include "quad_ctx.hpp"

namespace spg {

typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int kV29Floats = 238;             // float words of state per channel (layout: V29 word map below)
constexpr int kV29Ints = 43;
constexpr int kV29Words = kV29Floats + kV29Ints;
constexpr int kRrcSets = 48;
constexpr int kRrcLen = 27;
constexpr int kEqLen = 33;
constexpr int kPcmTile = 80;                            // samples of PCM staged per lane at a time

// State word map (identical to the reference-ordered snapshot the tests use):
//   floats: 0 agc_scaling, 1 agc_scaling_save, 2 eq_delta, 3 training_error, 4 carrier_track_p,
//           5 carrier_track_i, 6-7 godard low[2], 8-9 godard high[2], 10-11 godard dc[2],
//           12 baud_phase, 13-39 rrc_filter[27], 40-105 eq_coeff[33][2], 106-171 eq_coeff_save,
//           172-237 eq_buf[33][2]
//   ints:   0 bit_rate, 1 rrc_filter_step, 2 scramble_reg, 3 training_scramble_reg, 4 training_cd,
//           5 old_train, 6 training_stage, 7 training_count, 8 last_sample, 9 signal_present,
//           10 carrier_phase, 11 carrier_phase_rate, 12 carrier_phase_rate_save, 13 power reading,
//           14 carrier_on_power, 15 carrier_off_power, 16 eq_step, 17 eq_put_step, 18 eq_skip,
//           19 baud_half, 20-21 last_angles, 22-37 diff_angles, 38 constellation_state,
//           39 total timing correction, 40 high_sample, 41 low_samples, 42 carrier_drop_pending
enum
{
    VF_AGC = 0, VF_AGC_SAVE, VF_EQ_DELTA, VF_TRAIN_ERR, VF_TRACK_P, VF_TRACK_I,
    VF_GLOW = 6, VF_GHIGH = 8, VF_GDC = 10, VF_BAUD_PHASE = 12, VF_RRC = 13, VF_EQ_COEFF = 40,
    VF_EQ_SAVE = 106, VF_EQ_BUF = 172
};
enum
{
    VI_BIT_RATE = 0, VI_RRC_STEP, VI_SCRAMBLE, VI_TRAIN_SCRAMBLE, VI_TRAINING_CD, VI_OLD_TRAIN, VI_STAGE,
    VI_TRAIN_COUNT, VI_LAST_SAMPLE, VI_SIGNAL_PRESENT, VI_CARRIER_PHASE, VI_PHASE_RATE, VI_PHASE_RATE_SAVE,
    VI_POWER, VI_ON_POWER, VI_OFF_POWER, VI_EQ_STEP, VI_EQ_PUT_STEP, VI_EQ_SKIP, VI_BAUD_HALF,
    VI_LAST_ANGLES = 20, VI_DIFF_ANGLES = 22, VI_CONSTEL = 38, VI_TOTAL_CORR = 39, VI_HIGH_SAMPLE = 40,
    VI_LOW_SAMPLES = 41, VI_DROP_PENDING = 42
};

enum
{
    V29_NORMAL = 0, V29_SYMBOL_ACQUISITION, V29_LOG_PHASE, V29_WAIT_FOR_CDCD, V29_TRAIN_ON_CDCD,
    V29_TRAIN_ON_CDCD_AND_TEST, V29_TEST_ONES, V29_PARKED
};

struct V29Tables
{
    float rrc_re[kRrcSets*kRrcLen];
    float rrc_im[kRrcSets*kRrcLen];
    float sine[2048];
    float godard[7];
    float coarse_trigger;
    float fine_trigger;
    int coarse_step;
    int fine_step;
    uint16_t sqrt_tab[194];
    uint8_t space_map[400];
};

struct V29Launch
{
    const int16_t *amp;
    long long stride;
    int samples;
    const int32_t *lens;        // nullptr, or samples per channel in this call (<= samples; 0 = the channel sits it out)
    int n_ch;
    uint32_t *state;            // [kV29Words][n_ch]
    int8_t *events;             // [n_ch][ev_cap]: 0/1 bits and negative SIG_STATUS_* codes, in order
    int32_t *ev_count;          // [n_ch]
    int ev_cap;
    uint32_t *qam;              // QAM variant: [n_ch][qam_cap][7] qam_report records (include/spangpu.h), else unused
    int32_t *qam_count;         // [n_ch]
    int qam_cap;
    const V29Tables *tab;
};

// cosf()/sinf() of glibc >= 2.28 (sysdeps/ieee754/flt-32/sincosf.h: reduction by pi/2 and a polynomial, all in
// double, one rounding to float), which is what the reference build's libm computes; checked on the host against
// libm for every float in [0, 2*pi], the only range the receivers use.
SPG_FN_NOINLINE float spg_sincosf(float y, bool want_cos)
{
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    double x = (double) y;
    int n = want_cos  ?  1  :  0;
    double sg = 1.0;
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7FF;
    if (top < 0x3F4)
    {
        if (top < 0x398)
            return want_cos  ?  1.0f  :  y;
    }
    else
    {
        const double r = x*0x1.45F306DC9C883p+23;
        const int q = ((int32_t) r + 0x800000) >> 24;
        x = x - (double) q*0x1.921FB54442D18p0;
        const double x_red = x;
        if (((q & 3) == 1)  ||  ((q & 3) == 2))
            x = -x;
        if (q & 2)
            sg = -1.0;
        n = want_cos  ?  (q ^ 1)  :  q;
        const double x2 = x_red*x_red;
        if ((n & 1) == 0)
        {
            const double x3 = x*x2;
            const double t1 = s2 + x2*s3;
            const double x7 = x3*x2;
            const double s = x + x3*s1;
            return (float) (s + x7*t1);
        }
        const double x4 = x2*x2;
        const double k2 = sg*c3 + x2*(sg*c4);
        const double k1 = sg*c0 + x2*(sg*c1);
        const double x6 = x4*x2;
        const double c = k1 + x4*(sg*c2);
        return (float) (c + x6*k2);
    }
    const double x2 = x*x;
    if ((n & 1) == 0)
    {
        const double x3 = x*x2;
        const double t1 = s2 + x2*s3;
        const double x7 = x3*x2;
        const double s = x + x3*s1;
        return (float) (s + x7*t1);
    }
    const double x4 = x2*x2;
    const double k2 = c3 + x2*c4;
    const double k1 = c0 + x2*c1;
    const double x6 = x4*x2;
    const double c = k1 + x4*c2;
    return (float) (c + x6*k2);
}

SPG_FN int32_t v29_f2i(float v)
{
    // (int32_t) of the reference build (x86-64 cvttss2si): NaN / out of range -> INT32_MIN
    if (!(v < 2147483648.0f)  ||  !(v >= -2147483648.0f))
        return (int32_t) 0x80000000u;
    return (int32_t) v;
}

// spandsp/arctan2.h:47-80
SPG_FN int32_t v29_arctan2(float y, float x)
{
    if (y == 0.0f)
        return (x < 0.0f)  ?  (int32_t) 0x80000000u  :  0;
    if (x == 0.0f)
        return (y < 0.0f)  ?  (int32_t) 0xc0000000u  :  0x40000000;
    const float abs_y = fabsf(y);
    float angle;
    if (x < 0.0f)
        angle = 3.0f - (x + abs_y)/(abs_y - x);
    else
        angle = 1.0f - (x - abs_y)/(abs_y + x);
    angle *= 536870912.0f;
    if (y < 0.0f)
        angle = -angle;
    return v29_f2i(angle);
}

}   // namespace spg
