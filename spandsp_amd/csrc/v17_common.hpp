// v17_common.hpp -- what the two V.17 receiver kernels (v17_dev.hpp: one channel per lane; v17_quad.hpp: four lanes per
// channel) and the host-side lane emulator of the tests share: the state word map, the constant tables, the launch record.
#pragma once

#include "v29_common.hpp"

namespace spg {

constexpr int kV17Floats = 246;
constexpr int kV17Ints = 301;
constexpr int kV17Words = kV17Floats + kV17Ints;
constexpr int kV17Sets = 192;

// State word map: floats 0-237 as the V.29 map (v29_dev.hpp), 238-245 trellis distances[8];
//   ints: 0 bit_rate, 1 rrc_filter_step, 2 diff, 3 scramble_reg, 4 scrambler_tap, 5 short_train, 6 training_stage,
//         7 training_count, 8 last_sample, 9 signal_present, 10 carrier_drop_pending, 11 low_samples, 12 high_sample,
//         13 carrier_phase, 14 carrier_phase_rate, 15 carrier_phase_rate_save, 16 power reading, 17 carrier_on_power,
//         18 carrier_off_power, 19 eq_step, 20 eq_put_step, 21 eq_skip, 22 baud_half, 23-24 last_angles,
//         25-40 diff_angles, 41 space_map, 42 bits_per_symbol, 43 trellis_ptr, 44 total timing correction,
//         45-172 full_path_to_past_state_locations[16][8], 173-300 past_state_locations[16][8]
enum
{
    XF_DIST = 238
};
enum
{
    XI_BIT_RATE = 0, XI_RRC_STEP, XI_DIFF, XI_SCRAMBLE, XI_SCRAMBLER_TAP, XI_SHORT_TRAIN, XI_STAGE, XI_TRAIN_COUNT,
    XI_LAST_SAMPLE, XI_SIGNAL_PRESENT, XI_DROP_PENDING, XI_LOW_SAMPLES, XI_HIGH_SAMPLE, XI_CARRIER_PHASE, XI_PHASE_RATE,
    XI_PHASE_RATE_SAVE, XI_POWER, XI_ON_POWER, XI_OFF_POWER, XI_EQ_STEP, XI_EQ_PUT_STEP, XI_EQ_SKIP, XI_BAUD_HALF,
    XI_LAST_ANGLES = 23, XI_DIFF_ANGLES = 25, XI_SPACE_MAP = 41, XI_BITS_PER_SYMBOL = 42, XI_TRELLIS_PTR = 43,
    XI_TOTAL_CORR = 44, XI_FULL_PATH = 45, XI_PAST_STATE = 173
};

enum
{
    V17_NORMAL = 0, V17_SYMBOL_ACQUISITION, V17_LOG_PHASE, V17_SHORT_WAIT_FOR_CDBA, V17_WAIT_FOR_CDBA,
    V17_COARSE_TRAIN_ON_CDBA, V17_FINE_TRAIN_ON_CDBA, V17_SHORT_TRAIN_ON_CDBA_AND_TEST, V17_TRAIN_ON_CDBA_AND_TEST,
    V17_BRIDGE, V17_TCM_WINDUP, V17_TEST_ONES, V17_PARKED
};

struct V17Tables
{
    float rrc_re[kRrcLen*kV17Sets];         // [tap][phase]
    float rrc_im[kRrcLen*kV17Sets];
    float rrc_q[kRrcLen*kV17Sets*2];        // the same as {re, im} pairs, [tap][phase] (the four-lanes-per-channel kernel)
    float sine[2048];
    float godard[7];
    float coarse_trigger;
    float fine_trigger;
    int coarse_step;
    int fine_step;
    float con[128*2];                       // the bank's constellation, {re, im}
    uint32_t map[36*36*2];                  // the bank's soft-decision map: 8 bytes per cell (36*36 bytes at 4800 bps)
    uint16_t sqrt_tab[194];
};

struct V17Launch
{
    const int16_t *amp;
    long long stride;
    int samples;
    const int32_t *lens;        // nullptr, or samples per channel in this call (<= samples; 0 = the channel sits it out)
    int n_ch;
    int bit_rate;
    uint32_t *state;            // [kV17Words][n_ch]
    int8_t *events;
    int32_t *ev_count;
    int ev_cap;
    uint32_t *qam;              // QAM variant: [n_ch][qam_cap][7] qam_report records (include/spangpu.h), else unused
    int32_t *qam_count;         // [n_ch]
    int qam_cap;
    const V17Tables *tab;
};

// DDS_PHASE(), spandsp/dds.h:32 (float arithmetic)
#define V17_DDS_PHASE(deg)  ((int32_t) ((uint32_t) ((((deg) < 0.0f)  ?  (360.0f + (deg))  :  (deg))*65536.0f*65536.0f/360.0f)))

}   // namespace spg
