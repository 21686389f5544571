// v27ter_common.hpp -- what the two V.27ter receiver kernels (v27ter_dev.hpp: one channel per lane; v27ter_quad.hpp: four
// lanes per channel) and the host-side lane emulator of the tests share: the state word map, the tables, the launch record.
#pragma once

#include "v29_common.hpp"

namespace spg {

constexpr int kV27Floats = 225;
constexpr int kV27Ints = 45;
constexpr int kV27Words = kV27Floats + kV27Ints;
constexpr int kV27EqLen = 32;
constexpr int kV27MaxSets = 12;

// State word map (reference-ordered snapshot used by the tests):
//   floats: 0 agc_scaling, 1 agc_scaling_save, 2 eq_delta, 3 training_error, 4 carrier_track_p, 5 carrier_track_i,
//           6-32 rrc_filter[27], 33-96 eq_coeff[32][2], 97-160 eq_coeff_save, 161-224 eq_buf
//   ints:   0 bit_rate, 1 rrc_filter_step, 2 scramble_reg, 3 scrambler_pattern_count, 4 training_bc, 5 old_train,
//           6 training_stage, 7 training_count, 8 last_sample, 9 signal_present, 10 carrier_drop_pending, 11 low_samples,
//           12 high_sample, 13 constellation_state, 14 carrier_phase, 15 carrier_phase_rate, 16 carrier_phase_rate_save,
//           17 power reading, 18 carrier_on_power, 19 carrier_off_power, 20 eq_step, 21 eq_put_step, 22 eq_skip,
//           23 baud_half, 24 gardner_integrate, 25 gardner_step, 26 total timing correction, 27-28 last_angles,
//           29-44 diff_angles
enum
{
    WF_AGC = 0, WF_AGC_SAVE, WF_EQ_DELTA, WF_TRAIN_ERR, WF_TRACK_P, WF_TRACK_I,
    WF_RRC = 6, WF_EQ_COEFF = 33, WF_EQ_SAVE = 97, WF_EQ_BUF = 161
};
enum
{
    WI_BIT_RATE = 0, WI_RRC_STEP, WI_SCRAMBLE, WI_PATTERN_COUNT, WI_TRAINING_BC, WI_OLD_TRAIN, WI_STAGE, WI_TRAIN_COUNT,
    WI_LAST_SAMPLE, WI_SIGNAL_PRESENT, WI_DROP_PENDING, WI_LOW_SAMPLES, WI_HIGH_SAMPLE, WI_CONSTEL, WI_CARRIER_PHASE,
    WI_PHASE_RATE, WI_PHASE_RATE_SAVE, WI_POWER, WI_ON_POWER, WI_OFF_POWER, WI_EQ_STEP, WI_EQ_PUT_STEP, WI_EQ_SKIP,
    WI_BAUD_HALF, WI_GARDNER_INT, WI_GARDNER_STEP, WI_TOTAL_CORR, WI_LAST_ANGLES = 27, WI_DIFF_ANGLES = 29
};

enum
{
    V27_NORMAL = 0, V27_SYMBOL_ACQUISITION, V27_LOG_PHASE, V27_WAIT_FOR_HOP, V27_TRAIN_ON_ABAB, V27_TEST_ONES, V27_PARKED
};

struct V27Tables
{
    float re4800[8*kRrcLen];
    float im4800[8*kRrcLen];
    float re2400[12*kRrcLen];
    float im2400[12*kRrcLen];
    float sine[2048];
    uint16_t sqrt_tab[194];
};

struct V27Launch
{
    const int16_t *amp;
    long long stride;
    int samples;
    const int32_t *lens;        // nullptr, or samples per channel in this call (<= samples; 0 = the channel sits it out)
    int n_ch;
    int bit_rate;               // bank-wide: 4800 or 2400
    uint32_t *state;            // [kV27Words][n_ch]
    int8_t *events;
    int32_t *ev_count;
    int ev_cap;
    uint32_t *qam;              // QAM variant: [n_ch][qam_cap][7] qam_report records (include/spangpu.h), else unused
    int32_t *qam_count;         // [n_ch]
    int qam_cap;
    const V27Tables *tab;
};

}   // namespace spg
