/*
 * spangpu_refstate.h -- a channel's state in the REFERENCE's own struct layout, for moving live calls between a host
 * running spandsp and a bank on the GPU (either way) and for differential tests.
 *
 * The structs below mirror, field for field, the private structs of the reference's float build:
 *   spangpu_ref_goertzel_t      struct goertzel_state_s     src/spandsp/tone_detect.h:45-58
 *   spangpu_ref_dtmf_rx_t       struct dtmf_rx_state_s      src/spandsp/private/dtmf.h:54-117
 *   spangpu_ref_fir16_t         fir16_state_t               src/spandsp/fir.h:64-70
 *   spangpu_ref_echo_can_t      struct echo_can_state_s     src/spandsp/private/echo.h:37-89
 * so a pointer to a detector made by the reference (dtmf_rx_init(), echo_can_init()) can be passed as it is.  An import
 * takes the signal-processing fields; an export writes them and leaves the fields that belong to the caller's side of
 * the object alone (callbacks and their data, the collected digits, the logging descriptor, the pointers of the echo
 * canceller's arrays -- through which the arrays themselves are written).  tests/test_refstate_gpu.py checks sizes and
 * offsets against the reference build and runs calls that change sides in mid-stream.
 */
#if !defined(SPANGPU_REFSTATE_H)
#define SPANGPU_REFSTATE_H

#include "spangpu_spandsp.h"

#if defined(__cplusplus)
extern "C" {
#endif

typedef struct
{
    float v2;
    float v3;
    float fac;
    int samples;
    int current_sample;
} spangpu_ref_goertzel_t;

typedef struct
{
    digits_rx_callback_t digits_callback;
    void *digits_callback_data;
    span_tone_report_func_t realtime_callback;
    void *realtime_callback_data;
    bool filter_dialtone;
    float z350[2];
    float z440[2];
    float normal_twist;
    float reverse_twist;
    float threshold;
    float energy;
    spangpu_ref_goertzel_t row_out[4];
    spangpu_ref_goertzel_t col_out[4];
    uint8_t last_hit;
    uint8_t in_digit;
    int current_sample;
    int duration;
    int lost_digits;
    int current_digits;
    char digits[MAX_DTMF_DIGITS + 1];
    logging_state_t logging;
} spangpu_ref_dtmf_rx_t;

typedef struct
{
    int taps;
    int curr_pos;
    const int16_t *coeffs;
    int16_t *history;
} spangpu_ref_fir16_t;

typedef struct
{
    int tx_power[4];
    int rx_power[3];
    int clean_rx_power;
    int rx_power_threshold;
    int nonupdate_dwell;
    int curr_pos;
    int taps;
    int tap_mask;
    int adaption_mode;
    int32_t supp_test1;
    int32_t supp_test2;
    int32_t supp1;
    int32_t supp2;
    int vad;
    int cng;
    int16_t geigel_max;
    int geigel_lag;
    int dtd_onset;
    int tap_set;
    int tap_rotate_counter;
    int32_t latest_correction;
    int32_t last_acf[28];
    int narrowband_count;
    int narrowband_score;
    spangpu_ref_fir16_t fir_state;
    int16_t *fir_taps16[4];
    int32_t *fir_taps32;
    int32_t tx_hpf[2];
    int32_t rx_hpf[2];
    int cng_level;
    int cng_rndnum;
    int cng_filter;
    int16_t *snapshot;
} spangpu_ref_echo_can_t;

/* One channel of a DTMF bank <-> a reference detector.  Import: Goertzel states, block energy and phase, notch filter
   states, debounce state (last_hit, in_digit), duration, and the detector's own thresholds, twists and dial tone filter
   switch (the bank then carries them per channel, as spangpu_bank_set_channel_params() does).  Export: the same fields. */
SPANGPU_API int spangpu_dtmf_import_state(spangpu_bank_t *bank, int channel, const spangpu_ref_dtmf_rx_t *s);
SPANGPU_API int spangpu_dtmf_export_state(spangpu_bank_t *bank, int channel, spangpu_ref_dtmf_rx_t *s);

/* One channel of an echo canceller bank <-> a reference canceller of the same length.  The arrays travel through the
   struct's own pointers (fir_taps16[0..3], fir_taps32, fir_state.history: `taps` entries each, as echo_can_init() makes
   them); fir_state.coeffs names the set the FIR runs on.  On export the pointers are not changed, the arrays are
   written, and fir_state.coeffs is set to the struct's own fir_taps16[] entry. */
SPANGPU_API int spangpu_echo_import_state(spangpu_echo_t *bank, int channel, const spangpu_ref_echo_can_t *ec);
SPANGPU_API int spangpu_echo_export_state(spangpu_echo_t *bank, int channel, spangpu_ref_echo_can_t *ec);

#if defined(__cplusplus)
}
#endif

#endif
