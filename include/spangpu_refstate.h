/*
 * spangpu_refstate.h -- a channel's state in the REFERENCE's own struct layout, for moving live calls between a host
 * running spandsp and a bank on the GPU (either way) and for differential tests.
 *
 * The structs below mirror, field for field, the private structs of the reference's float build:
 *   spangpu_ref_goertzel_t      struct goertzel_state_s     src/spandsp/tone_detect.h:45-58
 *   spangpu_ref_dtmf_rx_t       struct dtmf_rx_state_s      src/spandsp/private/dtmf.h:54-117
 *   spangpu_ref_fir16_t         fir16_state_t               src/spandsp/fir.h:64-70
 *   spangpu_ref_echo_can_t      struct echo_can_state_s     src/spandsp/private/echo.h:37-89
 *   spangpu_ref_bell_mf_rx_t    struct bell_mf_rx_state_s   src/spandsp/private/bell_r2_mf.h:62-84
 *   spangpu_ref_r2_mf_rx_t      struct r2_mf_rx_state_s     src/spandsp/private/bell_r2_mf.h:102-116
 *   spangpu_ref_v29_rx_t        struct v29_rx_state_s       src/spandsp/private/v29rx.h:56-226
 *   spangpu_ref_v27ter_rx_t     struct v27ter_rx_state_s    src/spandsp/private/v27ter_rx.h:57-210
 *   spangpu_ref_v17_rx_t        struct v17_rx_state_s       src/spandsp/private/v17rx.h:64-254
 *   spangpu_ref_super_tone_rx_t struct super_tone_rx_state_s src/spandsp/private/super_tone_rx.h:50-62
 *   spangpu_ref_fsk_rx_t        struct fsk_rx_state_s       src/spandsp/private/fsk.h:58-115
 *   spangpu_ref_mct_rx_t        struct modem_connect_tones_rx_state_s   src/spandsp/private/modem_connect_tones.h:58-112
 *   spangpu_ref_sig_tone_rx_t   struct sig_tone_rx_state_s  src/spandsp/private/sig_tone.h:163-236
 * so a pointer to a detector made by the reference (dtmf_rx_init(), echo_can_init(), v29_rx_init()) can be passed as it is.  An import
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

/* ---- Bell MF and MFC/R2 detectors ------------------------------------------------------------------------ */
typedef struct
{
    digits_rx_callback_t digits_callback;
    void *digits_callback_data;
    spangpu_ref_goertzel_t out[6];
    uint8_t hits[5];
    int current_sample;
    int lost_digits;
    int current_digits;
    char digits[MAX_BELL_MF_DIGITS + 1];
} spangpu_ref_bell_mf_rx_t;                 /* struct bell_mf_rx_state_s, src/spandsp/private/bell_r2_mf.h:62-84 */

typedef struct
{
    span_tone_report_func_t callback;
    void *callback_data;
    bool fwd;
    spangpu_ref_goertzel_t out[6];
    int current_sample;
    int current_digit;
} spangpu_ref_r2_mf_rx_t;                   /* struct r2_mf_rx_state_s, src/spandsp/private/bell_r2_mf.h:102-116 */

/* On banks of kind SPANGPU_BELL_MF / SPANGPU_R2_MF.  What moves: the six Goertzel states, the block position, and
   the hit history (Bell MF) or the digit being reported (R2); an R2 detector must be of the bank's direction (fwd). */
SPANGPU_API int spangpu_bell_mf_import_state(spangpu_bank_t *bank, int channel, const spangpu_ref_bell_mf_rx_t *s);
SPANGPU_API int spangpu_bell_mf_export_state(spangpu_bank_t *bank, int channel, spangpu_ref_bell_mf_rx_t *s);
SPANGPU_API int spangpu_r2_mf_import_state(spangpu_bank_t *bank, int channel, const spangpu_ref_r2_mf_rx_t *s);
SPANGPU_API int spangpu_r2_mf_export_state(spangpu_bank_t *bank, int channel, spangpu_ref_r2_mf_rx_t *s);

/* ---- V.29 receiver ---------------------------------------------------------------------------------------- */
typedef struct
{
    float low_band_edge_coeff[3];
    float high_band_edge_coeff[3];
    float mixed_band_edges_coeff_3;
    float coarse_trigger;
    float fine_trigger;
    int coarse_step;
    int fine_step;
} spangpu_ref_godard_descriptor_t;          /* godard_ted_descriptor_t, src/spandsp/godard.h:30-66 */

typedef struct
{
    spangpu_ref_godard_descriptor_t desc;
    float low_band_edge[2];
    float high_band_edge[2];
    float dc_filter[2];
    float baud_phase;
    int total_baud_timing_correction;
} spangpu_ref_godard_t;                     /* struct godard_ted_state_s, src/spandsp/private/godard.h:28-55 */

typedef struct
{
    int shift;
    int32_t reading;
} spangpu_ref_power_meter_t;                /* struct power_meter_s, src/spandsp/private/power_meter.h:33-40 */

typedef struct
{
    int bit_rate;
    span_put_bit_func_t put_bit;
    void *put_bit_user_data;
    span_modem_status_func_t status_handler;
    void *status_user_data;
    qam_report_handler_t qam_report;
    void *qam_user_data;
    float agc_scaling;
    float agc_scaling_save;
    float eq_delta;
    complexf_t eq_coeff[33];
    complexf_t eq_coeff_save[33];
    complexf_t eq_buf[33];
    float training_error;
    float carrier_track_p;
    float carrier_track_i;
    float rrc_filter[27];
    spangpu_ref_godard_t godard;
    int rrc_filter_step;
    uint32_t scramble_reg;
    uint8_t training_scramble_reg;
    int training_cd;
    bool old_train;
    int training_stage;
    int training_count;
    int16_t last_sample;
    int signal_present;
    int carrier_drop_pending;
    int low_samples;
    int16_t high_sample;
    uint32_t carrier_phase;
    int32_t carrier_phase_rate;
    int32_t carrier_phase_rate_save;
    spangpu_ref_power_meter_t power;
    int32_t carrier_on_power;
    int32_t carrier_off_power;
    int eq_step;
    int eq_put_step;
    int eq_skip;
    int baud_half;
    int32_t last_angles[2];
    int32_t diff_angles[16];
    int constellation_state;
    logging_state_t logging;
} spangpu_ref_v29_rx_t;                     /* struct v29_rx_state_s (float build), src/spandsp/private/v29rx.h:56-226 */

/* A V.29 receiver -- trained equaliser, carrier and timing loops, scrambler, training stage and all -- in or out of a
   channel of a bank made with spangpu_modem_create(SPANGPU_V29, ...).  The callbacks, the logging descriptor and the
   Godard descriptor (a constant of the modem) are not taken on import and are left alone on export. */
SPANGPU_API int spangpu_v29_import_state(spangpu_modem_t *bank, int channel, const spangpu_ref_v29_rx_t *s);
SPANGPU_API int spangpu_v29_export_state(spangpu_modem_t *bank, int channel, spangpu_ref_v29_rx_t *s);

/* ---- V.27ter receiver ------------------------------------------------------------------------------------- */
typedef struct
{
    int bit_rate;
    span_put_bit_func_t put_bit;
    void *put_bit_user_data;
    span_modem_status_func_t status_handler;
    void *status_user_data;
    qam_report_handler_t qam_report;
    void *qam_user_data;
    float agc_scaling;
    float agc_scaling_save;
    float eq_delta;
    complexf_t eq_coeff[32];
    complexf_t eq_coeff_save[32];
    complexf_t eq_buf[32];
    float training_error;
    float carrier_track_p;
    float carrier_track_i;
    float rrc_filter[27];
    int rrc_filter_step;
    uint32_t scramble_reg;
    int scrambler_pattern_count;
    int training_bc;
    bool old_train;
    int training_stage;
    int training_count;
    int16_t last_sample;
    int signal_present;
    int carrier_drop_pending;
    int low_samples;
    int16_t high_sample;
    int constellation_state;
    uint32_t carrier_phase;
    int32_t carrier_phase_rate;
    int32_t carrier_phase_rate_save;
    spangpu_ref_power_meter_t power;
    int32_t carrier_on_power;
    int32_t carrier_off_power;
    int eq_step;
    int eq_put_step;
    int eq_skip;
    int baud_half;
    int gardner_integrate;
    int gardner_step;
    int total_baud_timing_correction;
    int32_t last_angles[2];
    int32_t diff_angles[16];
    logging_state_t logging;
} spangpu_ref_v27ter_rx_t;                  /* struct v27ter_rx_state_s (float build), src/spandsp/private/v27ter_rx.h:57-210 */

/* As for V.29, on a bank made with spangpu_modem_create(SPANGPU_V27TER, ...).  A V.27ter bank runs one bit rate (its
   pulse shaping tables are per rate): a receiver of the other rate is refused (SPANGPU_ERR_BAD_ARG). */
SPANGPU_API int spangpu_v27ter_import_state(spangpu_modem_t *bank, int channel, const spangpu_ref_v27ter_rx_t *s);
SPANGPU_API int spangpu_v27ter_export_state(spangpu_modem_t *bank, int channel, spangpu_ref_v27ter_rx_t *s);

/* ---- V.17 receiver ---------------------------------------------------------------------------------------- */
typedef struct
{
    int bit_rate;
    span_put_bit_func_t put_bit;
    void *put_bit_user_data;
    span_modem_status_func_t status_handler;
    void *status_user_data;
    qam_report_handler_t qam_report;
    void *qam_user_data;
    float agc_scaling;
    float agc_scaling_save;
    float eq_delta;
    complexf_t eq_coeff[33];
    complexf_t eq_coeff_save[33];
    complexf_t eq_buf[33];
    float training_error;
    float carrier_track_p;
    float carrier_track_i;
    float rrc_filter[27];
    const complexf_t *constellation;        /* into the reference library's own tables: never read, never written here */
    spangpu_ref_godard_t godard;
    int rrc_filter_step;
    int diff;
    uint32_t scramble_reg;
    int scrambler_tap;
    bool short_train;
    int training_stage;
    int training_count;
    int16_t last_sample;
    int signal_present;
    int carrier_drop_pending;
    int low_samples;
    int16_t high_sample;
    uint32_t carrier_phase;
    int32_t carrier_phase_rate;
    int32_t carrier_phase_rate_save;
    spangpu_ref_power_meter_t power;
    int32_t carrier_on_power;
    int32_t carrier_off_power;
    int eq_step;
    int eq_put_step;
    int eq_skip;
    int baud_half;
    int32_t last_angles[2];
    int32_t diff_angles[16];
    int space_map;
    int bits_per_symbol;
    int trellis_ptr;
    int full_path_to_past_state_locations[16][8];
    int past_state_locations[16][8];
    float distances[8];
    logging_state_t logging;
} spangpu_ref_v17_rx_t;                     /* struct v17_rx_state_s (float build), src/spandsp/private/v17rx.h:64-254 */

/* As for V.27ter, on a bank made with spangpu_modem_create(SPANGPU_V17, ...): one bit rate per bank, a receiver of
   another rate is refused.  The trellis decoder's survivor memory and path metrics travel with the rest. */
SPANGPU_API int spangpu_v17_import_state(spangpu_modem_t *bank, int channel, const spangpu_ref_v17_rx_t *s);
SPANGPU_API int spangpu_v17_export_state(spangpu_modem_t *bank, int channel, spangpu_ref_v17_rx_t *s);

/* ---- FSK receiver ----------------------------------------------------------------------------------------- */
typedef struct
{
    int32_t re;
    int32_t im;
} spangpu_ref_complexi32_t;                 /* complexi32_t, src/spandsp/complex.h:99-105 */

typedef struct
{
    int baud_rate;
    int framing_mode;
    int data_bits;
    int parity;
    int stop_bits;
    int total_data_bits;
    span_put_bit_func_t put_bit;
    void *put_bit_user_data;
    span_modem_status_func_t status_handler;
    void *status_user_data;
    int32_t carrier_on_power;
    int32_t carrier_off_power;
    spangpu_ref_power_meter_t power;
    int16_t last_sample;
    int signal_present;
    int32_t phase_rate[2];
    uint32_t phase_acc[2];
    int correlation_span;
    spangpu_ref_complexi32_t window[2][128];
    spangpu_ref_complexi32_t dot[2];
    int buf_ptr;
    int frame_pos;
    uint16_t frame_in_progress;
    int baud_phase;
    int last_bit;
    int scaling_shift;
    int parity_errors;
    int framing_errors;
} spangpu_ref_fsk_rx_t;                     /* struct fsk_rx_state_s, src/spandsp/private/fsk.h:58-115 */

/* On a bank made with spangpu_fsk_create(): the receiver must be of the bank's spec (baud rate and the two
   frequencies); framing mode, frame parameters and the power cutoffs travel with the receiver. */
SPANGPU_API int spangpu_fsk_import_state(spangpu_fsk_t *bank, int channel, const spangpu_ref_fsk_rx_t *s);
SPANGPU_API int spangpu_fsk_export_state(spangpu_fsk_t *bank, int channel, spangpu_ref_fsk_rx_t *s);

/* ---- modem connect tone detector and signalling tone receiver ------------------------------------------------ */
typedef struct
{
    int tone_type;
    bool real_time_reports;
    span_tone_report_func_t tone_callback;
    void *callback_data;
    float znotch_1;
    float znotch_2;
    float z15hz_1;
    float z15hz_2;
    int32_t notch_level;
    int32_t channel_level;
    int32_t am_level;
    int chunk_remainder;
    int tone_present;
    int tone_on;
    int tone_cycle_duration;
    int good_cycles;
    int hit;
    spangpu_ref_fsk_rx_t v21rx;
    unsigned int raw_bit_stream;
    int num_bits;
    int flags_seen;
    bool framing_ok_announced;
} spangpu_ref_mct_rx_t;                     /* struct modem_connect_tones_rx_state_s, src/spandsp/private/modem_connect_tones.h:58-112 */

typedef struct
{
    span_tone_report_func_t sig_update;
    void *user_data;
    const void *desc;                       /* into the reference library's own tables: never read, never written here */
    int current_rx_tone;
    int high_low_timer;
    int current_notch_filter;
    struct
    {
        float notch_z1[2];
        float notch_z2[2];
        spangpu_ref_power_meter_t power;
    } tone[3];
    float flat_z[2];
    spangpu_ref_power_meter_t flat_power;
    int tone_persistence_timeout;
    int last_sample_tone_present;
    int32_t flat_detection_threshold;
    int32_t sharp_detection_threshold;
    int32_t detection_ratio;
    bool flat_mode;
    bool notch_enabled;
    int flat_mode_timeout;
    int notch_insertion_timeout;
    int signalling_state;
    int signalling_state_duration;
} spangpu_ref_sig_tone_rx_t;                /* struct sig_tone_rx_state_s (float build), src/spandsp/private/sig_tone.h:163-236 */

/* The detector must be of the bank's tone type (after modem_connect_tones_rx_init()'s folding of the ANS variants);
   the V.21 receiver of the preamble hunting types travels inside it. */
SPANGPU_API int spangpu_mct_import_state(spangpu_mct_t *bank, int channel, const spangpu_ref_mct_rx_t *s);
SPANGPU_API int spangpu_mct_export_state(spangpu_mct_t *bank, int channel, spangpu_ref_mct_rx_t *s);
/* The receiver must have the bank's thresholds (that is: its tone type); its mode travels with it. */
SPANGPU_API int spangpu_sig_tone_rx_import_state(spangpu_sigtone_rx_t *bank, int channel, const spangpu_ref_sig_tone_rx_t *s);
SPANGPU_API int spangpu_sig_tone_rx_export_state(spangpu_sigtone_rx_t *bank, int channel, spangpu_ref_sig_tone_rx_t *s);

/* ---- super-tone receiver -------------------------------------------------------------------------------------- */
typedef struct
{
    int f1;
    int f2;
    int recognition_duration;
    int min_duration;
    int max_duration;
} spangpu_ref_super_tone_rx_segment_t;      /* struct super_tone_rx_segment_s, src/spandsp/private/super_tone_rx.h:29-36 */

typedef struct
{
    const void *desc;                       /* the reference's own descriptor: only its bin count is looked at */
    float energy;
    int detected_tone;
    int rotation;
    span_tone_report_func_t tone_callback;
    tone_segment_func_t segment_callback;
    void *callback_data;
    spangpu_ref_super_tone_rx_segment_t segments[11];
    spangpu_ref_goertzel_t state[];
} spangpu_ref_super_tone_rx_t;              /* struct super_tone_rx_state_s, src/spandsp/private/super_tone_rx.h:50-62 */

/* Between a receiver of the reference and a receiver made by this library's super_tone_rx_init() (or attached to a
   group) on a descriptor built by the same sequence of add_tone / add_element calls: the Goertzel states and the
   block's energy and position move to / from the channel in HBM, the cadence bookkeeping (the last ten runs, the pair
   seen last, the tone being followed) to / from the host object. */
SPANGPU_API int spangpu_super_tone_rx_import_state(super_tone_rx_state_t *s, const spangpu_ref_super_tone_rx_t *ref);
SPANGPU_API int spangpu_super_tone_rx_export_state(super_tone_rx_state_t *s, spangpu_ref_super_tone_rx_t *ref);

/* sizeof() of the mirror of the reference struct of that name ("dtmf_rx_state_t", "goertzel_state_t",
   "echo_can_state_t", "bell_mf_rx_state_t", "r2_mf_rx_state_t", "v29_rx_state_t", "v27ter_rx_state_t", "v17_rx_state_t", "fsk_rx_state_t", "modem_connect_tones_rx_state_t",
   "sig_tone_rx_state_t"), -1 for any other: what the tests hold against the reference build's own sizeof */
SPANGPU_API int spangpu_refstate_sizeof(const char *what);

#if defined(__cplusplus)
}
#endif

#endif
