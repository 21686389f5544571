/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or
 * executed from the product path (spandsp_amd/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * A plain-C, single-channel, sample-serial CPU restatement of the reference's
 * algorithms for the hot path named in BASELINE.json (Goertzel banks and their
 * decision logic, the vector primitives, the V.29 receiver, the G.168 echo
 * canceller).  Every function cites the reference file:line it follows
 * (paths relative to /root/reference/).  Build: strict IEEE binary32,
 * `gcc -O2 -ffp-contract=off -fwrapv` (see oracle/Makefile).
 *
 * PARITY PINNING: tests/test_oracle_pin.py checks every function here against
 * (a) the real reference compiled into oracle/_ref/libspandsp_ref.so when
 * that file is present, and (b) the golden vectors under tests/golden/ that
 * were generated from that build by tests/golden/make_golden.py.
 */
#if !defined(SPANDSP_AMD_ORACLE_H)
#define SPANDSP_AMD_ORACLE_H

#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#define ORC_API __attribute__((visibility("default")))

/* One record per observable action, in the order the reference would perform
   it.  kind: 1 = tone report callback (a=code, b=level, c=duration/delay)
              2 = digits callback (a=len; characters appended to the sink text)
              3 = put_bit (a=bit or negative SIG_STATUS_*)
              4 = super-tone segment callback (a=f1, b=f2, c=duration ms)
   Same layout as glue_event_t in oracle/ref_glue/ref_glue.c so the two
   streams compare with memcmp. */
typedef struct
{
    int32_t kind;
    int32_t a;
    int32_t b;
    int32_t c;
} orc_event_t;

typedef struct
{
    orc_event_t *ev;
    int n;
    int cap;
    char *text;
    int ntext;
    int captext;
    int want_qam;           /* modem receivers: also record the qam_report_handler_t calls (kinds 6 and 7) */
} orc_sink_t;

ORC_API orc_sink_t *orc_sink_new(void);
ORC_API void orc_sink_free(orc_sink_t *k);
ORC_API void orc_sink_clear(orc_sink_t *k);
ORC_API int orc_sink_count(const orc_sink_t *k);
ORC_API const orc_event_t *orc_sink_events(const orc_sink_t *k);
ORC_API int orc_sink_ntext(const orc_sink_t *k);
ORC_API const char *orc_sink_text(const orc_sink_t *k);
void orc_sink_push(orc_sink_t *k, int kind, int a, int b, int c);
/* The modems' qam_report(user, constel, target, symbol) call (v29rx.c:769-783): two events, kind 6 = {symbol, bits of
   constel.re, constel.im} and kind 7 = {1 if the pointers were NULL, bits of target.re, target.im}; only when wanted. */
ORC_API void orc_sink_want_qam(orc_sink_t *k, int on);
void orc_sink_qam(orc_sink_t *k, const float *constel, const float *target, int symbol);
void orc_sink_text_append(orc_sink_t *k, const char *s, int len);

/* Per-block trace record for the tone detectors (what the GPU kernels also
   emit in debug mode): the raw block decision before debouncing plus the
   Goertzel energies the decision saw. */
#define ORC_MAX_BINS 64
typedef struct
{
    int32_t hit;            /* raw block result (ASCII code, 0 = none); super-tone: k1 */
    int32_t aux;            /* DTMF: in_digit after the block; Bell: accepted digit or 0; R2: current digit; super-tone: k2 */
    float total_energy;     /* sum x*x over the block (DTMF, super-tone), else 0 */
    float e[ORC_MAX_BINS];  /* goertzel_result per bin (NaN-free; bins not evaluated are left 0) */
} orc_block_t;

/* ---- Goertzel primitive ---------------------------------------------------- */
ORC_API float orc_goertzel_fac(float freq);
ORC_API int16_t orc_ulaw_to_linear(uint8_t ulaw);
ORC_API int16_t orc_alaw_to_linear(uint8_t alaw);
typedef struct
{
    float v2;
    float v3;
    float fac;
    int samples;
    int current_sample;
} orc_goertzel_t;
ORC_API void orc_goertzel_init(orc_goertzel_t *g, float freq, int samples);
ORC_API int orc_goertzel_update(orc_goertzel_t *g, const int16_t amp[], int samples);
ORC_API float orc_goertzel_result(orc_goertzel_t *g);

/* ---- DTMF -------------------------------------------------------------------- */
typedef struct
{
    float fac[8];           /* row 0..3 then col 0..3 */
    float v2[8];
    float v3[8];
    float energy;
    float threshold;
    float normal_twist;
    float reverse_twist;
    float z350[2];
    float z440[2];
    int filter_dialtone;
    int current_sample;
    int duration;
    int last_hit;
    int in_digit;
    int mode;               /* 0 = buffer digits (dtmf_rx_get), 1 = digits callback, 2 = realtime callback */
    int lost_digits;
    int current_digits;
    char digits[129];
} orc_dtmf_t;

ORC_API int orc_dtmf_sizeof(void);
ORC_API void orc_dtmf_init(orc_dtmf_t *s, int mode);
ORC_API void orc_dtmf_parms(orc_dtmf_t *s, int filter_dialtone, float twist, float reverse_twist, float threshold);
/* blocks (may be NULL): receives one orc_block_t per completed 102-sample block,
   up to max_blocks; returns the number of completed blocks. */
ORC_API int orc_dtmf_rx(orc_dtmf_t *s, const int16_t amp[], int samples, orc_sink_t *sink, orc_block_t *blocks, int max_blocks);
/* n channel objects laid out contiguously (orc_dtmf_sizeof() apart); channel c reads amp + c*stride */
ORC_API void orc_dtmf_rx_batch(orc_dtmf_t *s, const int16_t amp[], int n, long long stride, int samples);
ORC_API int orc_dtmf_get(orc_dtmf_t *s, char *buf, int max);
ORC_API int orc_dtmf_status(const orc_dtmf_t *s);
ORC_API void orc_dtmf_fillin(orc_dtmf_t *s, int samples);

/* ---- Bell MF / R2 MF ------------------------------------------------------------ */
typedef struct
{
    float fac[6];
    float v2[6];
    float v3[6];
    int hits[5];
    int current_sample;
    int mode;               /* 0 = buffer digits, 1 = digits callback */
    int lost_digits;
    int current_digits;
    char digits[129];
} orc_bell_mf_t;

ORC_API int orc_bell_mf_sizeof(void);
ORC_API void orc_bell_mf_init(orc_bell_mf_t *s, int mode);
ORC_API int orc_bell_mf_rx(orc_bell_mf_t *s, const int16_t amp[], int samples, orc_sink_t *sink, orc_block_t *blocks, int max_blocks);
ORC_API int orc_bell_mf_get(orc_bell_mf_t *s, char *buf, int max);

typedef struct
{
    float fac[6];
    float v2[6];
    float v3[6];
    int fwd;
    int current_sample;
    int current_digit;
    int use_callback;
} orc_r2_mf_t;

ORC_API int orc_r2_mf_sizeof(void);
ORC_API void orc_r2_mf_init(orc_r2_mf_t *s, int fwd, int use_callback);
ORC_API int orc_r2_mf_rx(orc_r2_mf_t *s, const int16_t amp[], int samples, orc_sink_t *sink, orc_block_t *blocks, int max_blocks);

/* ---- Super tone --------------------------------------------------------------------- */
#define ORC_ST_MAX_TONES    32
#define ORC_ST_MAX_STEPS    16
typedef struct
{
    int f1;
    int f2;
    int min_duration;       /* in samples (ms*8) */
    int max_duration;
} orc_st_step_t;

typedef struct
{
    int used_frequencies;
    int monitored_frequencies;
    int pitches[64][2];
    float fac[64];
    int tones;
    int steps[ORC_ST_MAX_TONES];
    orc_st_step_t list[ORC_ST_MAX_TONES][ORC_ST_MAX_STEPS];
} orc_st_desc_t;

typedef struct
{
    const orc_st_desc_t *desc;
    float energy;
    int detected_tone;
    int rotation;
    int current_sample;
    int use_segment_cb;
    struct
    {
        int f1;
        int f2;
        int min_duration;
    } seg[11];
    float v2[64];
    float v3[64];
} orc_st_t;

ORC_API int orc_st_desc_sizeof(void);
ORC_API int orc_st_sizeof(void);
ORC_API void orc_st_desc_init(orc_st_desc_t *d);
ORC_API int orc_st_add_tone(orc_st_desc_t *d);
ORC_API int orc_st_add_element(orc_st_desc_t *d, int tone, int f1, int f2, int min_ms, int max_ms);
ORC_API void orc_st_init(orc_st_t *s, const orc_st_desc_t *d, int use_segment_cb);
ORC_API int orc_st_rx(orc_st_t *s, const int16_t amp[], int samples, orc_sink_t *sink, orc_block_t *blocks, int max_blocks);

/* ---- G.168 echo canceller ------------------------------------------------------------- */
#define ORC_ECHO_MAX_TAPS       1024
#define ORC_ECHO_USE_ADAPTION   0x01        /* src/spandsp/echo.h:118-127 */
#define ORC_ECHO_USE_NLP        0x02
#define ORC_ECHO_USE_CNG        0x04
#define ORC_ECHO_USE_CLIP       0x08
#define ORC_ECHO_USE_SUPPRESSOR 0x10
#define ORC_ECHO_USE_TX_HPF     0x20
#define ORC_ECHO_USE_RX_HPF     0x40
#define ORC_ECHO_DISABLE        0x80

typedef struct
{
    /* scalar words, in the order tests read them (see oracle/restated.py ECHO_FIELDS) */
    int32_t tx_power[4];
    int32_t rx_power[3];
    int32_t clean_rx_power;
    int32_t rx_power_threshold;
    int32_t nonupdate_dwell;
    int32_t curr_pos;
    int32_t taps;
    int32_t tap_mask;
    int32_t adaption_mode;
    int32_t supp_test1;
    int32_t supp_test2;
    int32_t supp1;
    int32_t supp2;
    int32_t vad;
    int32_t cng;
    int32_t geigel_max;
    int32_t geigel_lag;
    int32_t dtd_onset;
    int32_t tap_set;
    int32_t tap_rotate_counter;
    int32_t latest_correction;
    int32_t narrowband_count;
    int32_t narrowband_score;
    int32_t fir_curr_pos;
    int32_t tx_hpf[2];
    int32_t rx_hpf[2];
    int32_t cng_level;
    int32_t cng_rndnum;
    int32_t cng_filter;
    int32_t fir_set;            /* which tap set fir_state.coeffs points at */
    int32_t last_acf[9];
    int32_t taps32[ORC_ECHO_MAX_TAPS];
    int16_t taps16[4][ORC_ECHO_MAX_TAPS];
    int16_t history[ORC_ECHO_MAX_TAPS];
} orc_echo_t;

ORC_API int orc_echo_sizeof(void);
ORC_API int orc_echo_init(orc_echo_t *ec, int taps, int adaption_mode);
ORC_API void orc_echo_adaption_mode(orc_echo_t *ec, int adaption_mode);
ORC_API void orc_echo_flush(orc_echo_t *ec);
ORC_API int16_t orc_echo_hpf_tx(orc_echo_t *ec, int16_t tx);
ORC_API int16_t orc_echo_update(orc_echo_t *ec, int16_t tx, int16_t rx);
ORC_API void orc_echo_run(orc_echo_t *ec, const int16_t tx[], const int16_t rx[], int16_t clean[], int n, int use_hpf_tx);
ORC_API void orc_echo_run_batch(orc_echo_t *s, const int16_t tx[], const int16_t rx[], int16_t clean[], int n_ch, long long stride, int n, int use_hpf_tx);

/* ---- modem receivers ------------------------------------------------------------------------ */
typedef struct
{
    const float *rrc_re;        /* [48][27] V.29 rx pulse shaper, real part */
    const float *rrc_im;
    const float *sine;          /* [2048] dds_float.c sine table */
    const uint16_t *sqrt_tab;   /* [193] fixed_sqrt_table */
    float godard[7];            /* low[3], high[3], mixed */
    float godard_coarse_trigger;
    float godard_fine_trigger;
    int godard_coarse_step;
    int godard_fine_step;
    const float *v27_4800_re;   /* [8][27]  V.27ter 4800 bps rx pulse shaper */
    const float *v27_4800_im;
    const float *v27_2400_re;   /* [12][27] V.27ter 2400 bps rx pulse shaper */
    const float *v27_2400_im;
    const float *v17_re;        /* [192][27] V.17 rx pulse shaper */
    const float *v17_im;
    float v17_godard[7];
    float v17_godard_coarse_trigger;
    float v17_godard_fine_trigger;
    int v17_godard_coarse_step;
    int v17_godard_fine_step;
    const float *v17_constellation;     /* [128 + 64 + 32 + 16 + 4][2]: 14400, 12000, 9600, 7200, 4800 bps */
    const uint8_t *v17_maps;            /* [4][36][36][8] */
    const uint8_t *v17_map_4800;        /* [36][36] */
} orc_modem_tables_t;

ORC_API void orc_modem_set_tables(const orc_modem_tables_t *t);

typedef struct
{
    /* floats first, then ints (tests read the struct as words; see oracle/restated.py V29_LAYOUT) */
    float agc_scaling;
    float agc_scaling_save;
    float eq_delta;
    float training_error;
    float carrier_track_p;
    float carrier_track_i;
    float g_low[2];
    float g_high[2];
    float g_dc[2];
    float g_baud_phase;
    float rrc_filter[27];
    float eq_coeff[33][2];
    float eq_coeff_save[33][2];
    float eq_buf[33][2];
    int32_t bit_rate;
    int32_t rrc_filter_step;
    uint32_t scramble_reg;
    int32_t training_scramble_reg;
    int32_t training_cd;
    int32_t old_train;
    int32_t training_stage;
    int32_t training_count;
    int32_t last_sample;
    int32_t signal_present;
    uint32_t carrier_phase;
    int32_t carrier_phase_rate;
    int32_t carrier_phase_rate_save;
    int32_t power_reading;
    int32_t carrier_on_power;
    int32_t carrier_off_power;
    int32_t eq_step;
    int32_t eq_put_step;
    int32_t eq_skip;
    int32_t baud_half;
    int32_t last_angles[2];
    int32_t diff_angles[16];
    int32_t constellation_state;
    int32_t g_total_correction;
    int32_t high_sample;
    int32_t low_samples;
    int32_t carrier_drop_pending;
} orc_v29_t;

ORC_API float orc_trig_cosf(float x);
ORC_API float orc_trig_sinf(float x);
ORC_API int orc_v29_sizeof(void);
ORC_API int orc_v29_init(orc_v29_t *s, int bit_rate);
ORC_API int orc_v29_restart(orc_v29_t *s, int bit_rate, int old_train);
ORC_API void orc_v29_set_signal_cutoff(orc_v29_t *s, float cutoff);
ORC_API int orc_v29_rx(orc_v29_t *s, const int16_t amp[], int len, orc_sink_t *sink);

typedef struct
{
    /* floats first, then ints */
    float agc_scaling;
    float agc_scaling_save;
    float eq_delta;
    float training_error;
    float carrier_track_p;
    float carrier_track_i;
    float rrc_filter[27];
    float eq_coeff[32][2];
    float eq_coeff_save[32][2];
    float eq_buf[32][2];
    int32_t bit_rate;
    int32_t rrc_filter_step;
    uint32_t scramble_reg;
    int32_t scrambler_pattern_count;
    int32_t training_bc;
    int32_t old_train;
    int32_t training_stage;
    int32_t training_count;
    int32_t last_sample;
    int32_t signal_present;
    int32_t carrier_drop_pending;
    int32_t low_samples;
    int32_t high_sample;
    int32_t constellation_state;
    uint32_t carrier_phase;
    int32_t carrier_phase_rate;
    int32_t carrier_phase_rate_save;
    int32_t power_reading;
    int32_t carrier_on_power;
    int32_t carrier_off_power;
    int32_t eq_step;
    int32_t eq_put_step;
    int32_t eq_skip;
    int32_t baud_half;
    int32_t gardner_integrate;
    int32_t gardner_step;
    int32_t total_baud_timing_correction;
    int32_t last_angles[2];
    int32_t diff_angles[16];
} orc_v27ter_t;

typedef struct
{
    /* floats first, then ints */
    float agc_scaling;
    float agc_scaling_save;
    float eq_delta;
    float training_error;
    float carrier_track_p;
    float carrier_track_i;
    float g_low[2];
    float g_high[2];
    float g_dc[2];
    float g_baud_phase;
    float rrc_filter[27];
    float eq_coeff[33][2];
    float eq_coeff_save[33][2];
    float eq_buf[33][2];
    float distances[8];
    int32_t bit_rate;
    int32_t rrc_filter_step;
    int32_t diff;
    uint32_t scramble_reg;
    int32_t scrambler_tap;
    int32_t short_train;
    int32_t training_stage;
    int32_t training_count;
    int32_t last_sample;
    int32_t signal_present;
    int32_t carrier_drop_pending;
    int32_t low_samples;
    int32_t high_sample;
    uint32_t carrier_phase;
    int32_t carrier_phase_rate;
    int32_t carrier_phase_rate_save;
    int32_t power_reading;
    int32_t carrier_on_power;
    int32_t carrier_off_power;
    int32_t eq_step;
    int32_t eq_put_step;
    int32_t eq_skip;
    int32_t baud_half;
    int32_t last_angles[2];
    int32_t diff_angles[16];
    int32_t space_map;
    int32_t bits_per_symbol;
    int32_t trellis_ptr;
    int32_t g_total_correction;
    int32_t full_path_to_past_state_locations[16][8];
    int32_t past_state_locations[16][8];
} orc_v17_t;

#define ORC_V17_FLOATS      246
#define ORC_V17_INTS        301

ORC_API int orc_v17_sizeof(void);
ORC_API int orc_v17_init(orc_v17_t *s, int bit_rate);
ORC_API int orc_v17_restart(orc_v17_t *s, int bit_rate, int short_train);
ORC_API int orc_v17_rx(orc_v17_t *s, const int16_t amp[], int len, orc_sink_t *sink);

#define ORC_V27TER_FLOATS   225
#define ORC_V27TER_INTS     45

ORC_API int orc_v27ter_sizeof(void);
ORC_API int orc_v27ter_init(orc_v27ter_t *s, int bit_rate);
ORC_API int orc_v27ter_restart(orc_v27ter_t *s, int bit_rate, int old_train);
ORC_API int orc_v27ter_rx(orc_v27ter_t *s, const int16_t amp[], int len, orc_sink_t *sink);

/* ---- signal sources: tone_gen and the digit senders built on it (tonegen_oracle.c) ---- */
typedef struct
{
    int32_t phase_rate;
    float gain;
} orc_tone_t;

typedef struct
{
    orc_tone_t tone[4];
    int32_t duration[4];
    int32_t repeat;
} orc_tone_desc_t;

typedef struct
{
    orc_tone_t tone[4];
    uint32_t phase[4];
    int32_t duration[4];
    int32_t repeat;
    int32_t current_section;
    int32_t current_position;
} orc_tone_gen_t;

#define ORC_TX_QUEUE    128

typedef struct
{
    uint8_t data[ORC_TX_QUEUE];
    int32_t rd;
    int32_t count;
} orc_digit_queue_t;

typedef struct
{
    orc_tone_gen_t tones;
    float low_level;
    float high_level;
    int32_t on_time;
    int32_t off_time;
    orc_digit_queue_t queue;
} orc_dtmf_tx_t;

typedef struct
{
    orc_tone_gen_t tones;
    orc_digit_queue_t queue;
} orc_bell_mf_tx_t;

typedef struct
{
    orc_tone_gen_t tone;
    int32_t fwd;
    int32_t digit;
} orc_r2_mf_tx_t;

ORC_API void orc_tone_desc_init(orc_tone_desc_t *d, int f1, int l1, int f2, int l2, int d1, int d2, int d3, int d4,
                                int repeat);
ORC_API void orc_tone_gen_init(orc_tone_gen_t *g, const orc_tone_desc_t *d);
ORC_API int orc_tone_gen(orc_tone_gen_t *g, int16_t amp[], int max_samples);
ORC_API void orc_dtmf_tx_init(orc_dtmf_tx_t *s);
ORC_API void orc_dtmf_tx_set_level(orc_dtmf_tx_t *s, int level, int twist);
ORC_API void orc_dtmf_tx_set_timing(orc_dtmf_tx_t *s, int on_time, int off_time);
ORC_API int orc_dtmf_tx_put(orc_dtmf_tx_t *s, const char *digits, int len);
ORC_API int orc_dtmf_tx(orc_dtmf_tx_t *s, int16_t amp[], int max_samples);
ORC_API long long orc_dtmf_tx_run_batch(orc_dtmf_tx_t *s, int n, int16_t *amp, long long stride, int samples, int frames);
ORC_API void orc_bell_mf_tx_init(orc_bell_mf_tx_t *s);
ORC_API int orc_bell_mf_tx_put(orc_bell_mf_tx_t *s, const char *digits, int len);
ORC_API int orc_bell_mf_tx(orc_bell_mf_tx_t *s, int16_t amp[], int max_samples);
ORC_API void orc_r2_mf_tx_init(orc_r2_mf_tx_t *s, int fwd);
ORC_API int orc_r2_mf_tx_put(orc_r2_mf_tx_t *s, char digit);
ORC_API int orc_r2_mf_tx(orc_r2_mf_tx_t *s, int16_t amp[], int samples);

/* ---- FSK receiver (fsk_oracle.c).  Field order = the snapshot word order of ref_glue_fsk.c and of the
   device state (spandsp_amd/csrc/fsk_dev.hpp): 28 scalars then the window as [slot][tone][re, im]. ---- */
#define ORC_FSK_MAX_WINDOW  128
#define ORC_FSK_SCALARS     28

typedef struct
{
    int32_t baud_rate;
    int32_t framing_mode;       /* 0 async, 1 sync, 2 framed (fsk.h:124-129) */
    int32_t data_bits;
    int32_t parity;             /* 0 none, 1 even, 2 odd, 3 mark, 4 space (async.h:151-157) */
    int32_t stop_bits;
    int32_t total_data_bits;
    int32_t carrier_on_power;
    int32_t carrier_off_power;
    int32_t power_reading;
    int32_t last_sample;
    int32_t signal_present;
    int32_t phase_rate[2];
    uint32_t phase_acc[2];
    int32_t correlation_span;
    int32_t dot[2][2];
    int32_t buf_ptr;
    int32_t frame_pos;
    int32_t frame_in_progress;
    int32_t baud_phase;
    int32_t last_bit;
    int32_t scaling_shift;
    int32_t parity_errors;
    int32_t framing_errors;
    int32_t window[ORC_FSK_MAX_WINDOW][2][2];
} orc_fsk_t;

ORC_API int orc_fsk_sizeof(void);
ORC_API int orc_fsk_preset(int which, int32_t out[5]);
ORC_API int orc_fsk_init(orc_fsk_t *s, const int32_t spec[5], int framing_mode);
ORC_API int orc_fsk_restart(orc_fsk_t *s, const int32_t spec[5], int framing_mode);
ORC_API void orc_fsk_set_signal_cutoff(orc_fsk_t *s, float cutoff);
ORC_API void orc_fsk_set_frame_parameters(orc_fsk_t *s, int data_bits, int parity, int stop_bits);
typedef void (*orc_put_bit_t)(void *user, int bit);
ORC_API int orc_fsk_rx(orc_fsk_t *s, const int16_t amp[], int len, orc_sink_t *sink);
ORC_API int orc_fsk_rx_cb(orc_fsk_t *s, const int16_t amp[], int len, orc_put_bit_t put, void *user);
ORC_API int orc_fsk_fillin(orc_fsk_t *s, int len);

/* ---- V.29 transmitter (v29tx_oracle.c); field order = glue_v29_tx_snapshot() = the device state ---- */
typedef struct
{
    int32_t bit_rate;
    float base_gain;
    float gain;
    float rrc_re[9];
    float rrc_im[9];
    int32_t rrc_step;
    uint32_t scramble_reg;
    uint32_t training_scramble_reg;
    int32_t in_training;
    int32_t training_step;
    int32_t training_offset;
    uint32_t carrier_phase;
    int32_t carrier_phase_rate;
    int32_t baud_phase;
    int32_t constellation_state;
    uint32_t prbs;              /* the data source: x^15 + x^14 + 1 */
} orc_v29_tx_t;

#define ORC_V29_TX_WORDS    32

ORC_API int orc_v29_tx_sizeof(void);
ORC_API void orc_v29_tx_set_table(const float table[90]);
ORC_API int orc_v29_tx_init(orc_v29_tx_t *s, int bit_rate, int tep, uint32_t prbs_seed);
ORC_API int orc_v29_tx_restart(orc_v29_tx_t *s, int bit_rate, int tep);
ORC_API void orc_v29_tx_power(orc_v29_tx_t *s, float power);
ORC_API int orc_v29_tx(orc_v29_tx_t *s, int16_t amp[], int len);

/* ---- V.27ter transmitter (v27tertx_oracle.c); same word positions as the V.29 one ---- */
typedef struct
{
    int32_t bit_rate;
    float gain_2400;
    float gain_4800;
    float rrc_re[9];
    float rrc_im[9];
    int32_t rrc_step;
    uint32_t scramble_reg;
    int32_t scrambler_pattern_count;
    int32_t in_training;
    int32_t training_step;
    int32_t unused;
    uint32_t carrier_phase;
    int32_t carrier_phase_rate;
    int32_t baud_phase;
    int32_t constellation_state;
    uint32_t prbs;
} orc_v27ter_tx_t;

ORC_API int orc_v27ter_tx_sizeof(void);
ORC_API void orc_v27ter_tx_set_tables(const float t4800[45], const float t2400[180]);
ORC_API int orc_v27ter_tx_init(orc_v27ter_tx_t *s, int bit_rate, int tep, uint32_t prbs_seed);
ORC_API int orc_v27ter_tx_restart(orc_v27ter_tx_t *s, int bit_rate, int tep);
ORC_API void orc_v27ter_tx_power(orc_v27ter_tx_t *s, float power);
ORC_API int orc_v27ter_tx(orc_v27ter_tx_t *s, int16_t amp[], int len);

/* ---- V.17 transmitter (v17tx_oracle.c) ---- */
typedef struct
{
    int32_t bit_rate;
    float gain;
    int32_t diff;
    float rrc_re[9];
    float rrc_im[9];
    int32_t rrc_step;
    uint32_t scramble_reg;
    int32_t convolution;
    int32_t in_training;
    int32_t training_step;
    int32_t short_train;
    uint32_t carrier_phase;
    int32_t carrier_phase_rate;
    int32_t baud_phase;
    int32_t constellation_state;
    uint32_t prbs;
} orc_v17_tx_t;

ORC_API int orc_v17_tx_sizeof(void);
ORC_API int orc_v17_tx_init(orc_v17_tx_t *s, int bit_rate, int tep, uint32_t prbs_seed);
ORC_API int orc_v17_tx_restart(orc_v17_tx_t *s, int bit_rate, int tep, int short_train);
ORC_API void orc_v17_tx_power(orc_v17_tx_t *s, float power);
ORC_API int orc_v17_tx(orc_v17_tx_t *s, int16_t amp[], int len);

/* ---- AWGN (awgn_oracle.c); field order = the device state (doubles as two words, low first) ---- */
typedef struct
{
    double rms;
    double amp2;
    int32_t odd;
    int32_t ix1;
    int32_t ix2;
    int32_t ix3;
    double r[97];
} orc_awgn_t;

/* raw block decisions of the Goertzel users outside tone_detect.c (tone_oracle.c) */
ORC_API int orc_v18_tone_decide(const float e[], int n, float total, float threshold);
ORC_API int orc_ademco_tone_decide(float e1400, float e2300, float total);
ORC_API void orc_tone_functor_blocks(int kind, const float freqs[], int n_freqs, int block_len, float threshold,
                                     const int16_t amp[], int n_blocks, int32_t decisions[]);
ORC_API int orc_awgn_sizeof(void);
/* GNU libc's binary64 log() as the x86-64 FMA build computes it (glibc_log.c) -- what awgn() calls */
ORC_API double orc_glibc_log(double x);
ORC_API void orc_glibc_log_block(const double x[], double y[], int n);
ORC_API void orc_awgn_init_dbm0(orc_awgn_t *s, int idum, float level);
ORC_API int16_t orc_awgn(orc_awgn_t *s);
ORC_API void orc_awgn_block(orc_awgn_t *s, int16_t out[], int n);

/* ---- modem connect tones (mct_oracle.c) ---- */
#define ORC_MCT_FAX_CNG             1
#define ORC_MCT_ANS                 2
#define ORC_MCT_ANS_PR              3
#define ORC_MCT_ANSAM               4
#define ORC_MCT_ANSAM_PR            5
#define ORC_MCT_FAX_PREAMBLE        6
#define ORC_MCT_FAX_CED_OR_PREAMBLE 7
#define ORC_MCT_BELL_ANS            8
#define ORC_MCT_CALLING_TONE        9

/* The first 18 words are the detector's own state in the order of the device layout (mct_dev.hpp). */
typedef struct
{
    int32_t tone_type;
    float znotch_1;
    float znotch_2;
    float z15hz_1;
    float z15hz_2;
    int32_t notch_level;
    int32_t channel_level;
    int32_t am_level;
    int32_t tone_present;
    int32_t tone_on;
    int32_t tone_cycle_duration;
    int32_t good_cycles;
    int32_t hit;
    uint32_t raw_bit_stream;
    int32_t num_bits;
    int32_t flags_seen;
    int32_t framing_ok_announced;
    int32_t pad;
    orc_fsk_t v21;
    orc_sink_t *sink;
} orc_mct_t;

#define ORC_MCT_WORDS   18

ORC_API int orc_mct_sizeof(void);
ORC_API void orc_mct_init(orc_mct_t *s, int tone_type, orc_sink_t *sink);
ORC_API int orc_mct_rx(orc_mct_t *s, const int16_t amp[], int len);
ORC_API int orc_mct_get(orc_mct_t *s);

/* ---- in-band signalling tones (sigtone_oracle.c) ---- */
#define ORC_SIG_TONE_2280HZ         1
#define ORC_SIG_TONE_2600HZ         2
#define ORC_SIG_TONE_2400HZ_2600HZ  3

/* The first 27 words are the receiver's own state in the order of the device layout (sigtone_dev.hpp). */
typedef struct
{
    struct
    {
        float z1[2];
        float z2[2];
        int32_t power;
    } tone[3];
    float flat_z[2];
    int32_t flat_power;
    int32_t tone_persistence_timeout;
    int32_t last_sample_tone_present;
    int32_t flat_mode;
    int32_t flat_mode_timeout;
    int32_t notch_insertion_timeout;
    int32_t signalling_state;
    int32_t signalling_state_duration;
    int32_t current_notch_filter;
    int32_t current_rx_tone;
    int32_t tone_type;
    int32_t flat_detection_threshold;
    int32_t sharp_detection_threshold;
    int32_t detection_ratio;
    orc_sink_t *sink;
    const int32_t *script;      /* modes, one set from inside each report (what a caller's callback may do) */
    int32_t script_len;
    int32_t script_pos;
} orc_sigtone_rx_t;

#define ORC_SIGTONE_RX_WORDS    27

ORC_API int orc_sigtone_rx_sizeof(void);
ORC_API int orc_sigtone_rx_init(orc_sigtone_rx_t *s, int tone_type, orc_sink_t *sink);
ORC_API void orc_sigtone_rx_set_mode(orc_sigtone_rx_t *s, int mode);
ORC_API void orc_sigtone_rx_script(orc_sigtone_rx_t *s, const int32_t *modes, int n);
ORC_API void orc_sigtone_rx_thresholds(int tone_type, int32_t out[3]);
ORC_API int orc_sigtone_rx(orc_sigtone_rx_t *s, int16_t amp[], int len);

/* The first 5 words are the sender's own state in the order of the device layout. */
typedef struct
{
    uint32_t phase_acc[2];
    int32_t high_low_timer;
    int32_t current_tx_tone;
    int32_t current_tx_timeout;
    int32_t phase_rate[2];
    int16_t tone_scaling[2][2];
    int32_t tone_type;
    const int32_t *script;      /* (mode, duration) pairs, one consumed per update request */
    int32_t script_len;
    int32_t script_pos;
    orc_sink_t *sink;
} orc_sigtone_tx_t;

#define ORC_SIGTONE_TX_WORDS    5

ORC_API int orc_sigtone_tx_sizeof(void);
ORC_API int orc_sigtone_tx_init(orc_sigtone_tx_t *s, int tone_type, orc_sink_t *sink);
ORC_API void orc_sigtone_tx_set_mode(orc_sigtone_tx_t *s, int mode, int duration);
ORC_API void orc_sigtone_tx_script(orc_sigtone_tx_t *s, const int32_t *script, int n_pairs);
ORC_API int orc_sigtone_tx(orc_sigtone_tx_t *s, int16_t amp[], int len);

#if defined(__cplusplus)
}
#endif

#endif
