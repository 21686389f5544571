/*
 * spangpu_spandsp.h -- the spandsp-named, per-channel C entry points of libspangpu.so.
 *
 * Source-compatible with the reference's public tone-detector API: same names, argument
 * order, return values and callback types, so an existing caller (FreeSWITCH mod_spandsp,
 * adsi.c, v18.c ...) re-links unchanged.  State objects are opaque (the reference exposes
 * them only through its private headers).  Behind each object is one channel slot of a
 * GPU bank (include/spangpu.h):
 *
 *   - xxx_rx_init(NULL, ...) makes a PRIVATE one-channel bank: xxx_rx() then runs the frame
 *     on the GPU and replays callbacks before it returns -- same observable behaviour, per
 *     call, as the reference (this is the plumbing configuration);
 *   - spangpu_group_create() + spangpu_xxx_rx_attach() put N objects on ONE bank: each
 *     xxx_rx() stages its frame (from any thread; one submitter per object), and when every
 *     attached channel has staged -- or when the tick's owner calls spangpu_group_flush() at
 *     its deadline, with whoever staged: a silent or late channel stalls nobody and is itself
 *     untouched -- a single kernel launch advances the channels that take part and their
 *     callbacks are replayed in channel order, each channel's events in sample order.  This
 *     is the production configuration (thousands of channels / launch).
 *
 * Reference declarations being replaced (relative to the reference tree):
 *   dtmf_rx_init/_release/_free            src/spandsp/dtmf.h:216-228     src/dtmf.c:447-519
 *   dtmf_rx                                src/spandsp/dtmf.h:177         src/dtmf.c:132-361
 *   dtmf_rx_fillin/_status/_get            src/spandsp/dtmf.h:185-201     src/dtmf.c:363-408
 *   dtmf_rx_parms/_set_realtime_callback   src/spandsp/dtmf.h:153-170     src/dtmf.c:410-445
 *   bell_mf_rx_init/_rx/_get/_release/_free  src/spandsp/bell_r2_mf.h:199-228  src/bell_r2_mf.c:507-748
 *   r2_mf_rx_init/_rx/_get/_release/_free  src/spandsp/bell_r2_mf.h:236-266  src/bell_r2_mf.c:750-951
 *   super_tone_rx_* (descriptor + detector)  src/spandsp/super_tone_rx.h:76-164  src/super_tone_rx.c:81-568
 *   goertzel_*  / make_goertzel_descriptor src/spandsp/tone_detect.h:86-124  src/tone_detect.c:60-205
 *   echo_can_init/_release/_free/_flush/_adaption_mode/_update/_hpf_tx
 *                                          src/spandsp/echo.h:145-185     src/echo.c:254-380,421-669
 *   v29_rx_init/_restart/_release/_free/_set_put_bit/_set_modem_status_handler/_set_qam_report_handler/_rx/_fillin/_equalizer_state/
 *     _carrier_frequency/_symbol_timing_correction/_signal_power/_set_signal_cutoff
 *                                          src/spandsp/v29rx.h:151-244    src/v29rx.c:139-196,867-1148
 *   v27ter_rx_* (same set)                 src/spandsp/v27ter_rx.h:71-165 src/v27ter_rx.c:136-170,863-1210
 *   v17_rx_* (same set)                    src/spandsp/v17rx.h:236-333    src/v17rx.c:165-212,1212-1541
 * Callback types: digits_rx_callback_t (dtmf.h:76), span_tone_report_func_t and
 * tone_segment_func_t (super_tone_rx.h:56-58), span_put_bit_func_t and span_modem_status_func_t
 * (async.h:123,131), qam_report_handler_t (v29rx.h:130).  xxx_rx_get_logging_state() hands out a
 * logging_state_t with the layout spandsp's span_log_*() functions work on (private/logging.h:32-42).
 *   filter_create/_delete/_step, cfilter_create/_delete/_step
 *                                          src/spandsp/complex_filters.h:62-68   src/complex_filters.c:39-118
 *   echo_can_snapshot                      src/spandsp/echo.h:185          src/echo.c:376-379
 */
#if !defined(SPANGPU_SPANDSP_H)
#define SPANGPU_SPANDSP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "spangpu.h"

#if defined(__cplusplus)
extern "C" {
#endif

/* The logging descriptor the reference's receivers carry (src/spandsp/private/logging.h:32-42): same fields in the same
   order, so that a caller that links spandsp's logging.c can hand the pointer xxx_rx_get_logging_state() returns to
   span_log_set_level() / span_log_set_tag() as it does today.  The engine itself never writes log text. */
typedef void (*message_handler_func_t)(void *user_data, int level, const char *text);
typedef struct logging_state_s
{
    int level;
    int samples_per_second;
    int64_t elapsed_samples;
    const char *tag;
    const char *protocol;
    message_handler_func_t span_message;
    void *user_data;
} logging_state_t;

typedef void (*digits_rx_callback_t)(void *user_data, const char *digits, int len);
typedef void (*span_tone_report_func_t)(void *user_data, int code, int level, int delay);
typedef void (*tone_segment_func_t)(void *data, int f1, int f2, int duration);

#define MAX_DTMF_DIGITS         128
#define MAX_BELL_MF_DIGITS      128

typedef struct dtmf_rx_state_s dtmf_rx_state_t;
typedef struct bell_mf_rx_state_s bell_mf_rx_state_t;
typedef struct r2_mf_rx_state_s r2_mf_rx_state_t;
typedef struct super_tone_rx_descriptor_s super_tone_rx_descriptor_t;
typedef struct super_tone_rx_state_s super_tone_rx_state_t;
typedef struct goertzel_state_s goertzel_state_t;
typedef struct
{
    float fac;
    int samples;
} goertzel_descriptor_t;

/* ---- modem receivers ------------------------------------------------------------------ */
typedef void (*span_put_bit_func_t)(void *user_data, int bit);
typedef void (*span_modem_status_func_t)(void *user_data, int status);

typedef struct
{
    float re;
    float im;
} complexf_t;

/* spandsp/v29rx.h:130: the per-baud constellation tap (constel and target are NULL for V.27ter's timing hop report) */
typedef void (*qam_report_handler_t)(void *user_data, const complexf_t *constel, const complexf_t *target, int symbol);

/* spandsp/async.h:66-103: the status codes a receiver passes through put_bit / the status handler */
enum
{
    SIG_STATUS_CARRIER_DOWN = -1,
    SIG_STATUS_CARRIER_UP = -2,
    SIG_STATUS_TRAINING_IN_PROGRESS = -3,
    SIG_STATUS_TRAINING_SUCCEEDED = -4,
    SIG_STATUS_TRAINING_FAILED = -5
};

typedef struct v29_rx_state_s v29_rx_state_t;
typedef struct v27ter_rx_state_s v27ter_rx_state_t;
typedef struct v17_rx_state_s v17_rx_state_t;
typedef struct spangpu_modem_group_s spangpu_modem_group_t;

/* N receivers of one kind (SPANGPU_V29 / _V27TER / _V17) and bit rate on one bank; as for the tone groups, each
   xxx_rx() stages its frame and the last attached channel of the tick (or spangpu_modem_group_flush()) launches. */
SPANGPU_API spangpu_modem_group_t *spangpu_modem_group_create(int device, int kind, int n_channels, int bit_rate, int max_samples);
SPANGPU_API int spangpu_modem_group_destroy(spangpu_modem_group_t *g);
SPANGPU_API int spangpu_modem_group_flush(spangpu_modem_group_t *g);
SPANGPU_API spangpu_modem_t *spangpu_modem_group_bank(spangpu_modem_group_t *g);

SPANGPU_API v29_rx_state_t *v29_rx_init(v29_rx_state_t *s, int bit_rate, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API v29_rx_state_t *spangpu_v29_rx_attach(spangpu_modem_group_t *g, int channel, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API int v29_rx(v29_rx_state_t *s, const int16_t amp[], int len);
SPANGPU_API int v29_rx_fillin(v29_rx_state_t *s, int len);
SPANGPU_API int v29_rx_release(v29_rx_state_t *s);
SPANGPU_API int v29_rx_free(v29_rx_state_t *s);
SPANGPU_API void v29_rx_set_put_bit(v29_rx_state_t *s, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API void v29_rx_set_modem_status_handler(v29_rx_state_t *s, span_modem_status_func_t handler, void *user_data);
SPANGPU_API void v29_rx_set_qam_report_handler(v29_rx_state_t *s, qam_report_handler_t handler, void *user_data);
SPANGPU_API int v29_rx_equalizer_state(v29_rx_state_t *s, complexf_t **coeffs);
SPANGPU_API float v29_rx_carrier_frequency(v29_rx_state_t *s);
SPANGPU_API float v29_rx_symbol_timing_correction(v29_rx_state_t *s);
SPANGPU_API float v29_rx_signal_power(v29_rx_state_t *s);
SPANGPU_API logging_state_t *v29_rx_get_logging_state(v29_rx_state_t *s);          /* src/spandsp/v29rx.h:177 */
SPANGPU_API void v29_rx_set_signal_cutoff(v29_rx_state_t *s, float cutoff);

SPANGPU_API v27ter_rx_state_t *v27ter_rx_init(v27ter_rx_state_t *s, int bit_rate, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API v27ter_rx_state_t *spangpu_v27ter_rx_attach(spangpu_modem_group_t *g, int channel, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API int v27ter_rx(v27ter_rx_state_t *s, const int16_t amp[], int len);
SPANGPU_API int v27ter_rx_fillin(v27ter_rx_state_t *s, int len);
SPANGPU_API int v27ter_rx_release(v27ter_rx_state_t *s);
SPANGPU_API int v27ter_rx_free(v27ter_rx_state_t *s);
SPANGPU_API void v27ter_rx_set_put_bit(v27ter_rx_state_t *s, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API void v27ter_rx_set_modem_status_handler(v27ter_rx_state_t *s, span_modem_status_func_t handler, void *user_data);
SPANGPU_API void v27ter_rx_set_qam_report_handler(v27ter_rx_state_t *s, qam_report_handler_t handler, void *user_data);
SPANGPU_API int v27ter_rx_equalizer_state(v27ter_rx_state_t *s, complexf_t **coeffs);
SPANGPU_API float v27ter_rx_carrier_frequency(v27ter_rx_state_t *s);
SPANGPU_API float v27ter_rx_symbol_timing_correction(v27ter_rx_state_t *s);
SPANGPU_API float v27ter_rx_signal_power(v27ter_rx_state_t *s);
SPANGPU_API logging_state_t *v27ter_rx_get_logging_state(v27ter_rx_state_t *s);
SPANGPU_API void v27ter_rx_set_signal_cutoff(v27ter_rx_state_t *s, float cutoff);

SPANGPU_API v17_rx_state_t *v17_rx_init(v17_rx_state_t *s, int bit_rate, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API v17_rx_state_t *spangpu_v17_rx_attach(spangpu_modem_group_t *g, int channel, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API int v17_rx(v17_rx_state_t *s, const int16_t amp[], int len);
SPANGPU_API int v17_rx_fillin(v17_rx_state_t *s, int len);
SPANGPU_API int v17_rx_release(v17_rx_state_t *s);
SPANGPU_API int v17_rx_free(v17_rx_state_t *s);
SPANGPU_API void v17_rx_set_put_bit(v17_rx_state_t *s, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API void v17_rx_set_modem_status_handler(v17_rx_state_t *s, span_modem_status_func_t handler, void *user_data);
SPANGPU_API void v17_rx_set_qam_report_handler(v17_rx_state_t *s, qam_report_handler_t handler, void *user_data);
SPANGPU_API int v17_rx_equalizer_state(v17_rx_state_t *s, complexf_t **coeffs);
SPANGPU_API float v17_rx_carrier_frequency(v17_rx_state_t *s);
SPANGPU_API float v17_rx_symbol_timing_correction(v17_rx_state_t *s);
SPANGPU_API float v17_rx_signal_power(v17_rx_state_t *s);
SPANGPU_API logging_state_t *v17_rx_get_logging_state(v17_rx_state_t *s);
SPANGPU_API void v17_rx_set_signal_cutoff(v17_rx_state_t *s, float cutoff);

SPANGPU_API int v29_rx_restart(v29_rx_state_t *s, int bit_rate, bool old_train);
SPANGPU_API int v27ter_rx_restart(v27ter_rx_state_t *s, int bit_rate, bool old_train);
SPANGPU_API int v17_rx_restart(v17_rx_state_t *s, int bit_rate, int short_train);

/* ---- echo canceller ---------------------------------------------------------------------- */
/* src/spandsp/echo.h:120-131 */
enum
{
    ECHO_CAN_USE_ADAPTION = 0x01,
    ECHO_CAN_USE_NLP = 0x02,
    ECHO_CAN_USE_CNG = 0x04,
    ECHO_CAN_USE_CLIP = 0x08,
    ECHO_CAN_USE_SUPPRESSOR = 0x10,
    ECHO_CAN_USE_TX_HPF = 0x20,
    ECHO_CAN_USE_RX_HPF = 0x40,
    ECHO_CAN_DISABLE = 0x80
};

typedef struct echo_can_state_s echo_can_state_t;

/* len (taps) must be 32, 64, 128 or 256.  echo_can_update() is one kernel launch per sample: source compatibility only;
   use spangpu_echo_can_update_block() (one object, n samples) or spangpu_echo_update() (N channels) for throughput. */
SPANGPU_API echo_can_state_t *echo_can_init(int len, int adaption_mode);
SPANGPU_API int echo_can_release(echo_can_state_t *ec);
SPANGPU_API int echo_can_free(echo_can_state_t *ec);
SPANGPU_API void echo_can_flush(echo_can_state_t *ec);
/* echo_can_snapshot(): keep a copy of tap set 0 as it is now (src/echo.c:376-379; the reference keeps it in a private
   field).  spangpu_echo_can_snapshot_taps() hands the copy out: taps int16 values, returns the count. */
SPANGPU_API void echo_can_snapshot(echo_can_state_t *ec);
SPANGPU_API int spangpu_echo_can_snapshot_taps(echo_can_state_t *ec, int16_t *out, int max);
SPANGPU_API void echo_can_adaption_mode(echo_can_state_t *ec, int adaption_mode);
/* (a kernel launch per sample: build with -DSPANGPU_WARN_SLOW_CALLS to have the compiler point at every call site) */
#if defined(SPANGPU_WARN_SLOW_CALLS)
#define SPANGPU_SLOW_CALL(msg) __attribute__((deprecated(msg)))
#else
#define SPANGPU_SLOW_CALL(msg)
#endif
SPANGPU_API int16_t echo_can_update(echo_can_state_t *ec, int16_t tx, int16_t rx)
    SPANGPU_SLOW_CALL("one kernel launch per sample: use spangpu_echo_can_update_block() or spangpu_echo_update()");
SPANGPU_API int16_t echo_can_hpf_tx(echo_can_state_t *ec, int16_t tx)
    SPANGPU_SLOW_CALL("one kernel launch per sample: use spangpu_echo_can_update_block(..., use_hpf_tx = 1)");
SPANGPU_API int spangpu_echo_can_update_block(echo_can_state_t *ec, const int16_t tx[], const int16_t rx[], int16_t clean[],
                                              int16_t tx_out[], int n, int use_hpf_tx);
SPANGPU_API spangpu_echo_t *spangpu_echo_can_bank(echo_can_state_t *ec);

/* ---- channel groups: N spandsp objects on one GPU bank ----------------------------- */
typedef struct spangpu_group_s spangpu_group_t;

/* kind: SPANGPU_DTMF / _BELL_MF / _R2_MF / _SUPER_TONE.  params may be NULL (defaults).
   max_samples bounds the samples one xxx_rx() call may carry (e.g. 160). */
SPANGPU_API spangpu_group_t *spangpu_group_create(int device, int kind, int n_channels, int max_samples,
                                                  const spangpu_tone_params_t *params);
SPANGPU_API int spangpu_group_destroy(spangpu_group_t *g);
/* Run the tick now: every attached channel must have staged a frame of the same length.
   Returns the number of channels processed, or a negative SPANGPU_ERR_*. */
SPANGPU_API int spangpu_group_flush(spangpu_group_t *g);
SPANGPU_API spangpu_bank_t *spangpu_group_bank(spangpu_group_t *g);

SPANGPU_API dtmf_rx_state_t *spangpu_dtmf_rx_attach(spangpu_group_t *g, int channel,
                                                    digits_rx_callback_t callback, void *user_data);
SPANGPU_API bell_mf_rx_state_t *spangpu_bell_mf_rx_attach(spangpu_group_t *g, int channel,
                                                          digits_rx_callback_t callback, void *user_data);
SPANGPU_API r2_mf_rx_state_t *spangpu_r2_mf_rx_attach(spangpu_group_t *g, int channel,
                                                      span_tone_report_func_t callback, void *user_data);
SPANGPU_API super_tone_rx_state_t *spangpu_super_tone_rx_attach(spangpu_group_t *g, int channel,
                                                                super_tone_rx_descriptor_t *desc,
                                                                span_tone_report_func_t callback, void *user_data);

/* Bank parameters (bin count and taps) for a super-tone group whose channels all use `desc`. */
SPANGPU_API int spangpu_super_tone_params(const super_tone_rx_descriptor_t *desc, spangpu_tone_params_t *params);
/* ... and its cadences, for spangpu_bank_cadence_events() (matched on the device; include/spangpu.h) */
SPANGPU_API int spangpu_super_tone_cadences(const super_tone_rx_descriptor_t *desc, spangpu_bank_t *bank, int want_segments);

/* ---- DTMF (src/spandsp/dtmf.h) -------------------------------------------------------- */
SPANGPU_API dtmf_rx_state_t *dtmf_rx_init(dtmf_rx_state_t *s, digits_rx_callback_t callback, void *user_data);
SPANGPU_API int dtmf_rx_release(dtmf_rx_state_t *s);
SPANGPU_API int dtmf_rx_free(dtmf_rx_state_t *s);
SPANGPU_API void dtmf_rx_set_realtime_callback(dtmf_rx_state_t *s, span_tone_report_func_t callback, void *user_data);
SPANGPU_API void dtmf_rx_parms(dtmf_rx_state_t *s, int filter_dialtone, float twist, float reverse_twist, float threshold);
SPANGPU_API int dtmf_rx(dtmf_rx_state_t *s, const int16_t amp[], int samples);
SPANGPU_API int dtmf_rx_fillin(dtmf_rx_state_t *s, int samples);
SPANGPU_API int dtmf_rx_status(dtmf_rx_state_t *s);
SPANGPU_API size_t dtmf_rx_get(dtmf_rx_state_t *s, char *digits, int max);
SPANGPU_API logging_state_t *dtmf_rx_get_logging_state(dtmf_rx_state_t *s);        /* src/spandsp/dtmf.h:206 */

/* ---- Bell MF / R2 MF (src/spandsp/bell_r2_mf.h) ------------------------------------------ */
SPANGPU_API bell_mf_rx_state_t *bell_mf_rx_init(bell_mf_rx_state_t *s, digits_rx_callback_t callback, void *user_data);
SPANGPU_API int bell_mf_rx_release(bell_mf_rx_state_t *s);
SPANGPU_API int bell_mf_rx_free(bell_mf_rx_state_t *s);
SPANGPU_API int bell_mf_rx(bell_mf_rx_state_t *s, const int16_t amp[], int samples);
SPANGPU_API size_t bell_mf_rx_get(bell_mf_rx_state_t *s, char *buf, int max);

SPANGPU_API r2_mf_rx_state_t *r2_mf_rx_init(r2_mf_rx_state_t *s, bool fwd, span_tone_report_func_t callback, void *user_data);
SPANGPU_API int r2_mf_rx_release(r2_mf_rx_state_t *s);
SPANGPU_API int r2_mf_rx_free(r2_mf_rx_state_t *s);
SPANGPU_API int r2_mf_rx(r2_mf_rx_state_t *s, const int16_t amp[], int samples);
SPANGPU_API int r2_mf_rx_get(r2_mf_rx_state_t *s);

/* ---- Super tone (src/spandsp/super_tone_rx.h) ----------------------------------------------- */
SPANGPU_API super_tone_rx_descriptor_t *super_tone_rx_make_descriptor(super_tone_rx_descriptor_t *desc);
SPANGPU_API int super_tone_rx_free_descriptor(super_tone_rx_descriptor_t *desc);
SPANGPU_API int super_tone_rx_add_tone(super_tone_rx_descriptor_t *desc);
SPANGPU_API int super_tone_rx_add_element(super_tone_rx_descriptor_t *desc, int tone, int f1, int f2, int min, int max);
SPANGPU_API super_tone_rx_state_t *super_tone_rx_init(super_tone_rx_state_t *s, super_tone_rx_descriptor_t *desc,
                                                      span_tone_report_func_t callback, void *user_data);
SPANGPU_API int super_tone_rx_release(super_tone_rx_state_t *s);
SPANGPU_API int super_tone_rx_free(super_tone_rx_state_t *s);
SPANGPU_API void super_tone_rx_tone_callback(super_tone_rx_state_t *s, span_tone_report_func_t callback, void *user_data);
SPANGPU_API void super_tone_rx_segment_callback(super_tone_rx_state_t *s, tone_segment_func_t callback);
SPANGPU_API int super_tone_rx(super_tone_rx_state_t *s, const int16_t amp[], int samples);
SPANGPU_API int super_tone_rx_fillin(super_tone_rx_state_t *s, int samples);

/* ---- Goertzel (src/spandsp/tone_detect.h) ------------------------------------------------------ */
SPANGPU_API void make_goertzel_descriptor(goertzel_descriptor_t *t, float freq, int samples);
SPANGPU_API goertzel_state_t *goertzel_init(goertzel_state_t *s, goertzel_descriptor_t *t);
SPANGPU_API int goertzel_release(goertzel_state_t *s);
SPANGPU_API int goertzel_free(goertzel_state_t *s);

/* ---- complex_filters.h: a filter instance around a caller-supplied step function (src/complex_filters.c:39-118).  The
   arithmetic of a filter is its fspec_t's fsf callback; these entry points only own the state it works on.  Host side only. */
typedef struct filter_s filter_t;
typedef float (*filter_step_func_t)(filter_t *fi, float x);
typedef struct
{
    int nz;
    int np;
    filter_step_func_t fsf;
} fspec_t;
struct filter_s
{
    fspec_t *fs;
    float sum;
    int ptr;                    /* only for moving average filters */
    float v[];
};
typedef struct
{
    filter_t *ref;
    filter_t *imf;
} cfilter_t;
SPANGPU_API filter_t *filter_create(fspec_t *fs);
SPANGPU_API void filter_delete(filter_t *fi);
SPANGPU_API float filter_step(filter_t *fi, float x);
SPANGPU_API cfilter_t *cfilter_create(fspec_t *fs);
SPANGPU_API void cfilter_delete(cfilter_t *cfi);
SPANGPU_API complexf_t cfilter_step(cfilter_t *cfi, const complexf_t *z);
/* The receivers' inner primitives under their spandsp names (vec_*, cvec_*, power_meter_*, godard_ted_*, periodogram*) are
 * NOT in libspangpu.so: libspandsp calls them from inside its own modules (fsk, v22bis, the modem receivers ...), and a process
 * that loads both libraries would have those internal calls bound here.  They live in libspangpu_prims.so, which a caller
 * links only when it wants them: include/spangpu_prims.h. */
SPANGPU_API void goertzel_reset(goertzel_state_t *s);
SPANGPU_API int goertzel_update(goertzel_state_t *s, const int16_t amp[], int samples);
SPANGPU_API float goertzel_result(goertzel_state_t *s);

/* ---- FSK receiver, modem connect tones, DTMF sender (csrc/shim_fsk.c) ---------------------------------
 * Reference declarations being replaced:
 *   fsk_rx_init/_restart/_rx/_fillin/_release/_free/_set_put_bit/_set_modem_status_handler/_set_signal_cutoff/
 *     _set_frame_parameters/_signal_power/_get_parity_errors/_get_framing_errors, fsk_spec_t, preset_fsk_specs[]
 *                                          src/spandsp/fsk.h:85-260      src/fsk.c:60-155,270-742
 *   modem_connect_tones_rx_init/_rx/_fillin/_get/_release/_free, modem_connect_tone_to_str
 *                                          src/spandsp/modem_connect_tones.h:57-200  src/modem_connect_tones.c:84-112,521-871
 *   dtmf_tx_init/_put/dtmf_tx/_set_level/_set_timing/_release/_free
 *                                          src/spandsp/dtmf.h:112-148    src/dtmf.c:551-676
 * As with the other families: xxx_init(NULL, ...) makes a private one-channel bank; a spangpu_line_group_t puts N
 * receivers on one bank and one launch per tick.  Storage supplied by the caller (s != NULL) cannot be used --
 * the state lives in HBM -- so such a call returns NULL.  fsk_rx_restart() accepts the spec the object was made
 * with (a bank runs one baud rate); a dtmf_tx digits callback is not replayed (init with one returns NULL).
 */
typedef struct
{
    const char *name;
    int freq_zero;
    int freq_one;
    int tx_level;
    int min_level;
    int baud_rate;
} fsk_spec_t;

enum
{
    FSK_V21CH1 = 0, FSK_V21CH2, FSK_V23CH1, FSK_V23CH2, FSK_BELL103CH1, FSK_BELL103CH2, FSK_BELL202, FSK_WEITBRECHT_4545,
    FSK_WEITBRECHT_50, FSK_WEITBRECHT_476, FSK_V21CH1_110
};

enum
{
    FSK_FRAME_MODE_ASYNC = 0,
    FSK_FRAME_MODE_SYNC = 1,
    FSK_FRAME_MODE_FRAMED = 2
};

enum
{
    ASYNC_PARITY_NONE = 0,
    ASYNC_PARITY_EVEN,
    ASYNC_PARITY_ODD,
    ASYNC_PARITY_MARK,
    ASYNC_PARITY_SPACE
};

enum
{
    MODEM_CONNECT_TONES_NONE = 0,
    MODEM_CONNECT_TONES_FAX_CNG = 1,
    MODEM_CONNECT_TONES_ANS = 2,
    MODEM_CONNECT_TONES_ANS_PR = 3,
    MODEM_CONNECT_TONES_ANSAM = 4,
    MODEM_CONNECT_TONES_ANSAM_PR = 5,
    MODEM_CONNECT_TONES_FAX_PREAMBLE = 6,
    MODEM_CONNECT_TONES_FAX_CED_OR_PREAMBLE = 7,
    MODEM_CONNECT_TONES_BELL_ANS = 8,
    MODEM_CONNECT_TONES_CALLING_TONE = 9,
    MODEM_CONNECT_TONES_REAL_TIME_REPORTS = 0x1000
};
#define MODEM_CONNECT_TONES_FAX_CED MODEM_CONNECT_TONES_ANS

typedef void (*digits_tx_callback_t)(void *user_data);

typedef struct fsk_rx_state_s fsk_rx_state_t;
typedef struct modem_connect_tones_rx_state_s modem_connect_tones_rx_state_t;
typedef struct dtmf_tx_state_s dtmf_tx_state_t;
typedef struct spangpu_line_group_s spangpu_line_group_t;

/* The objects themselves, for callers that bring their own storage (xxx_init(&my_state, ...), ended with xxx_release();
   spandsp keeps the like under spandsp/private/).  They are handles: the signal state lives in HBM.  Nothing in them is
   for the caller to touch. */
struct fsk_rx_state_s
{
    spangpu_line_group_t *grp;
    int channel;
    int private_grp;
    span_put_bit_func_t put_bit;
    void *put_bit_user_data;
    span_modem_status_func_t status_handler;
    void *status_user_data;
    int caller_storage;
};
struct modem_connect_tones_rx_state_s
{
    spangpu_line_group_t *grp;
    int channel;
    int private_grp;
    span_tone_report_func_t tone_callback;
    void *callback_data;
    int caller_storage;
    int tone_type;
};
struct dtmf_tx_state_s
{
    spangpu_txbank_t *bank;
    digits_tx_callback_t callback;
    void *callback_data;
    int puts;                   /* dtmf_tx_put() calls that queued digits (the callback's way of answering) */
    int caller_storage;
};

SPANGPU_API extern const fsk_spec_t preset_fsk_specs[];

SPANGPU_API spangpu_line_group_t *spangpu_fsk_group_create(int device, const fsk_spec_t *spec, int framing_mode, int n_channels,
                                                           int max_samples);
SPANGPU_API spangpu_line_group_t *spangpu_modem_connect_tones_group_create(int device, int tone_type, int use_callbacks,
                                                                           int n_channels, int max_samples);
SPANGPU_API int spangpu_line_group_destroy(spangpu_line_group_t *g);
SPANGPU_API int spangpu_line_group_flush(spangpu_line_group_t *g);

SPANGPU_API fsk_rx_state_t *fsk_rx_init(fsk_rx_state_t *s, const fsk_spec_t *spec, int framing_mode, span_put_bit_func_t put_bit,
                                        void *user_data);
SPANGPU_API fsk_rx_state_t *spangpu_fsk_rx_attach(spangpu_line_group_t *g, int channel, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API int fsk_rx_restart(fsk_rx_state_t *s, const fsk_spec_t *spec, int framing_mode);
SPANGPU_API int fsk_rx(fsk_rx_state_t *s, const int16_t *amp, int len);
SPANGPU_API int fsk_rx_fillin(fsk_rx_state_t *s, int len);
SPANGPU_API int fsk_rx_release(fsk_rx_state_t *s);
SPANGPU_API int fsk_rx_free(fsk_rx_state_t *s);
SPANGPU_API void fsk_rx_set_put_bit(fsk_rx_state_t *s, span_put_bit_func_t put_bit, void *user_data);
SPANGPU_API void fsk_rx_set_modem_status_handler(fsk_rx_state_t *s, span_modem_status_func_t handler, void *user_data);
SPANGPU_API void fsk_rx_set_signal_cutoff(fsk_rx_state_t *s, float cutoff);
SPANGPU_API void fsk_rx_set_frame_parameters(fsk_rx_state_t *s, int data_bits, int parity, int stop_bits);
SPANGPU_API float fsk_rx_signal_power(fsk_rx_state_t *s);
SPANGPU_API int fsk_rx_get_parity_errors(fsk_rx_state_t *s, bool reset);
SPANGPU_API int fsk_rx_get_framing_errors(fsk_rx_state_t *s, bool reset);

SPANGPU_API modem_connect_tones_rx_state_t *modem_connect_tones_rx_init(modem_connect_tones_rx_state_t *s, int tone_type,
                                                                        span_tone_report_func_t tone_callback, void *user_data);
SPANGPU_API modem_connect_tones_rx_state_t *spangpu_modem_connect_tones_rx_attach(spangpu_line_group_t *g, int channel,
                                                                                  span_tone_report_func_t tone_callback,
                                                                                  void *user_data);
SPANGPU_API int modem_connect_tones_rx(modem_connect_tones_rx_state_t *s, const int16_t amp[], int len);
SPANGPU_API int modem_connect_tones_rx_fillin(modem_connect_tones_rx_state_t *s, int len);
SPANGPU_API int modem_connect_tones_rx_get(modem_connect_tones_rx_state_t *s);
SPANGPU_API int modem_connect_tones_rx_release(modem_connect_tones_rx_state_t *s);
SPANGPU_API int modem_connect_tones_rx_free(modem_connect_tones_rx_state_t *s);
SPANGPU_API const char *modem_connect_tone_to_str(int tone);

/* ---- in-band signalling tones (csrc/shim_sigtone.c) ------------------------------------------------------
 * Reference declarations being replaced:
 *   sig_tone_rx_init/_rx/_set_mode/_release/_free, sig_tone_tx_init/_tx/_set_mode/_release/_free
 *                                          src/spandsp/sig_tone.h:57-176   src/sig_tone.c:246-738
 * Tone types SIG_TONE_2280HZ .. SIG_TONE_2400HZ_2600HZ (1..3) and the mode / report bits are the SPANGPU_SIG_TONE_*
 * values of spangpu.h.  An object is a private one-channel bank (N channels per launch: spangpu_sigtone_rx_create()
 * etc.); caller storage (s != NULL) returns NULL, an object made by these functions is re-initialised in place.  Callbacks arrive from inside the call, in order, and a
 * mode set in one applies to the rest of the same frame, as in the reference.
 */
typedef struct sig_tone_rx_state_s sig_tone_rx_state_t;
typedef struct sig_tone_tx_state_s sig_tone_tx_state_t;

SPANGPU_API sig_tone_rx_state_t *sig_tone_rx_init(sig_tone_rx_state_t *s, int tone_type, span_tone_report_func_t sig_update,
                                                  void *user_data);
SPANGPU_API int sig_tone_rx(sig_tone_rx_state_t *s, int16_t amp[], int len);
SPANGPU_API void sig_tone_rx_set_mode(sig_tone_rx_state_t *s, int mode, int duration);
SPANGPU_API int sig_tone_rx_release(sig_tone_rx_state_t *s);
SPANGPU_API int sig_tone_rx_free(sig_tone_rx_state_t *s);
SPANGPU_API sig_tone_tx_state_t *sig_tone_tx_init(sig_tone_tx_state_t *s, int tone_type, span_tone_report_func_t sig_update,
                                                  void *user_data);
SPANGPU_API int sig_tone_tx(sig_tone_tx_state_t *s, int16_t amp[], int len);
SPANGPU_API void sig_tone_tx_set_mode(sig_tone_tx_state_t *s, int mode, int duration);
SPANGPU_API int sig_tone_tx_release(sig_tone_tx_state_t *s);
SPANGPU_API int sig_tone_tx_free(sig_tone_tx_state_t *s);

SPANGPU_API dtmf_tx_state_t *dtmf_tx_init(dtmf_tx_state_t *s, digits_tx_callback_t callback, void *user_data);
SPANGPU_API int dtmf_tx_release(dtmf_tx_state_t *s);
SPANGPU_API int dtmf_tx_free(dtmf_tx_state_t *s);
SPANGPU_API void dtmf_tx_set_level(dtmf_tx_state_t *s, int level, int twist);
SPANGPU_API void dtmf_tx_set_timing(dtmf_tx_state_t *s, int on_time, int off_time);
SPANGPU_API int dtmf_tx_put(dtmf_tx_state_t *s, const char *digits, int len);
SPANGPU_API int dtmf_tx(dtmf_tx_state_t *s, int16_t amp[], int max_samples);

/* Bell MF and MFC/R2 senders (src/spandsp/bell_r2_mf.h:137-191, src/bell_r2_mf.c:281-372,386-462): a private one-channel
   signal source bank each; caller-supplied storage (s != NULL) is not supported (state lives in HBM) */
typedef struct bell_mf_tx_state_s bell_mf_tx_state_t;
typedef struct r2_mf_tx_state_s r2_mf_tx_state_t;
struct bell_mf_tx_state_s
{
    spangpu_txbank_t *bank;
    int caller_storage;
};
struct r2_mf_tx_state_s
{
    spangpu_txbank_t *bank;
    int caller_storage;
};
SPANGPU_API bell_mf_tx_state_t *bell_mf_tx_init(bell_mf_tx_state_t *s);
SPANGPU_API int bell_mf_tx_release(bell_mf_tx_state_t *s);
SPANGPU_API int bell_mf_tx_free(bell_mf_tx_state_t *s);
SPANGPU_API int bell_mf_tx_put(bell_mf_tx_state_t *s, const char *digits, int len);
SPANGPU_API int bell_mf_tx(bell_mf_tx_state_t *s, int16_t amp[], int max_samples);
SPANGPU_API r2_mf_tx_state_t *r2_mf_tx_init(r2_mf_tx_state_t *s, bool fwd);
SPANGPU_API int r2_mf_tx_release(r2_mf_tx_state_t *s);
SPANGPU_API int r2_mf_tx_free(r2_mf_tx_state_t *s);
SPANGPU_API int r2_mf_tx_put(r2_mf_tx_state_t *s, char digit);
SPANGPU_API int r2_mf_tx(r2_mf_tx_state_t *s, int16_t amp[], int samples);

#if defined(__cplusplus)
}
#endif

#endif
