/*
 * spangpu.h -- C ABI of libspangpu.so: the MI355X (gfx950) batched engine behind
 * spandsp's tone-detect / modem-demod / echo-cancel entry points.
 *
 * This is the drop-in boundary: plain C, pointers and sizes only.  A "bank" owns
 * the device-resident (HBM, structure-of-arrays) state of N independent 8 kHz
 * channels of one detector kind; one spangpu_bank_rx() call advances every
 * channel of the bank by one frame with one kernel launch.  The per-channel
 * spandsp-named entry points (dtmf_rx() ...) in include/spangpu_spandsp.h are a
 * thin C shim over this ABI.
 *
 * What each entry point replaces in the reference (paths relative to the
 * reference tree):
 *   spangpu_bank_create(SPANGPU_DTMF)       dtmf_rx_init()            src/dtmf.c:447-504, src/spandsp/dtmf.h:216
 *   spangpu_bank_rx() on a DTMF bank        dtmf_rx() x N channels    src/dtmf.c:132-361,  src/spandsp/dtmf.h:177
 *   spangpu_bank_create(SPANGPU_BELL_MF)    bell_mf_rx_init()         src/bell_r2_mf.c:693-735
 *   spangpu_bank_rx() on a Bell MF bank     bell_mf_rx() x N          src/bell_r2_mf.c:507-673
 *   spangpu_bank_create(SPANGPU_R2_MF)      r2_mf_rx_init()           src/bell_r2_mf.c:889-937
 *   spangpu_bank_rx() on an R2 MF bank      r2_mf_rx() x N            src/bell_r2_mf.c:750-880
 *   spangpu_bank_create(SPANGPU_SUPER_TONE) super_tone_rx_init()      src/super_tone_rx.c:507-554
 *   spangpu_bank_rx() on a super-tone bank  super_tone_rx() x N       src/super_tone_rx.c:454-490 (+ :289-362 on device,
 *                                                                     cadence matcher :164-228,:364-448 in the host shim)
 *   spangpu_bank_create(SPANGPU_GOERTZEL)   goertzel_init() x bins    src/tone_detect.c:71-92
 *   spangpu_bank_rx() on a Goertzel bank    goertzel_update()/goertzel_result() x N x bins   src/tone_detect.c:123-205
 *                                           plus the block's total energy: the tone front ends of src/v18.c:1546-1600 and
 *                                           src/ademco_contactid.c:890-935 are this bank + a few comparisons on the host
 *   spangpu_bank_reset_channel()            xxx_rx_init() on a live object / dtmf_rx_fillin()  src/dtmf.c:363-379
 *
 * Threading: a bank is single-submitter (like a spandsp state object); distinct
 * banks are independent.  All work of a bank is ordered on one HIP stream.
 * Errors: negative int codes, never exceptions; a missing GPU / HIP runtime is
 * SPANGPU_ERR_NO_DEVICE -- there is NO CPU fallback in this library.
 */
#if !defined(SPANGPU_H)
#define SPANGPU_H

#include <stddef.h>
#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#define SPANGPU_API __attribute__((visibility("default")))

/* ---- error codes ------------------------------------------------------------ */
#define SPANGPU_OK                  0
#define SPANGPU_ERR_NO_DEVICE       (-1)    /* no HIP device / runtime: the product path fails loudly */
#define SPANGPU_ERR_BAD_ARG         (-2)
#define SPANGPU_ERR_NO_MEMORY       (-3)
#define SPANGPU_ERR_HIP             (-4)    /* a HIP call failed; see spangpu_last_error() */
#define SPANGPU_ERR_STATE           (-5)    /* call not valid in the bank's current state */
#define SPANGPU_ERR_UNSUPPORTED     (-6)

/* ---- bank kinds ------------------------------------------------------------- */
#define SPANGPU_DTMF                1
#define SPANGPU_BELL_MF             2
#define SPANGPU_R2_MF               3
#define SPANGPU_SUPER_TONE          4
#define SPANGPU_GOERTZEL            5
#define SPANGPU_V29                 6
#define SPANGPU_V27TER              7
#define SPANGPU_V17                 8
#define SPANGPU_ECHO                9

/* ---- where a caller's frame buffer lives / how it is laid out ------------------ */
#define SPANGPU_MEM_HOST            0       /* pageable or pinned host memory: copied H2D on the bank stream */
#define SPANGPU_MEM_DEVICE          1       /* already resident in HBM (e.g. a torch tensor's data_ptr) */

#define SPANGPU_LAYOUT_CHANNEL_MAJOR 0      /* amp[channel][sample]: what N spandsp callers naturally hold */
#define SPANGPU_LAYOUT_SAMPLE_MAJOR  1      /* amp[sample][channel]: pre-interleaved */

#define SPANGPU_MAX_BINS            64      /* bins per channel in one Goertzel / super-tone bank (super_tone_rx.h:44: 64 pitches) */

/* ---- report modes for DTMF (which of the reference's delivery paths is replayed) */
#define SPANGPU_REPORT_DIGITS       0       /* digits[] buffer / digits_rx_callback_t   (dtmf.c:318-340) */
#define SPANGPU_REPORT_REALTIME     2       /* span_tone_report_func_t on/off reports    (dtmf.c:309-316) */

typedef struct spangpu_bank_s spangpu_bank_t;

/* Parameters of a tone-detector bank.  Zero-initialise, then set what you need. */
typedef struct
{
    int32_t report_mode;        /* DTMF: SPANGPU_REPORT_*                                              */
    int32_t filter_dialtone;    /* DTMF: 1 = apply the 350/440 Hz notches (dtmf.c:167-183)             */
    float twist_db;             /* DTMF: <= 0 keeps the default 8 dB  (dtmf_rx_parms, dtmf.c:421-445)  */
    float reverse_twist_db;     /* DTMF: <= 0 keeps the default 4 dB                                   */
    float threshold_dbm0;       /* DTMF: 0 or <= -99 keeps the default -42 dBm0                        */
    int32_t r2_fwd;             /* R2 MF: 1 = forward tone set, 0 = backward                           */
    int32_t n_bins;             /* GOERTZEL / SUPER_TONE: bins per channel (<= SPANGPU_MAX_BINS)       */
    int32_t block_len;          /* GOERTZEL: samples per block (super-tone is fixed at 128)            */
    float bin_fac[SPANGPU_MAX_BINS]; /* GOERTZEL / SUPER_TONE: 2cos(2 pi f/8000) per bin, as
                                   make_goertzel_descriptor() computes it (tone_detect.c:60-68);
                                   spangpu_goertzel_fac() below reproduces it on the host             */
    int32_t trace;              /* 1 = also write per-block Goertzel energies (parity / diagnostics)   */
    int32_t set_mask;           /* DTMF: SPANGPU_TP_* -- which of twist_db / reverse_twist_db / threshold_dbm0 are
                                   meant as given, 0 dB and 0 dBm0 included, under dtmf_rx_parms()'s own tests
                                   (twists >= 0, threshold > -99).  0 = the rules in the field comments above.   */
    int32_t functor;            /* GOERTZEL: SPANGPU_FUNCTOR_* -- which raw block decision goes into a block's `hit`     */
    float functor_threshold;    /* GOERTZEL, SPANGPU_FUNCTOR_V18: the object's level threshold (v18_state_t.threshold)     */
} spangpu_tone_params_t;

/* Raw block decisions of the reference's Goertzel users outside tone_detect.c, made on the device from a generic bank's
   bin energies and block energy (the caller's debounce / protocol logic stays on the host, fed by spangpu_block_t.hit):
     SPANGPU_FUNCTOR_V18     src/v18.c:1580-1600 (and :1721-1741): bins = the nine tone set frequencies, block_len 102;
                             hit = index of the tone seen, 0 = none (and tone 0: the reference does not tell them apart)
     SPANGPU_FUNCTOR_ADEMCO  src/ademco_contactid.c:915-935: bins = 1400 Hz, 2300 Hz, block_len 55; hit = 1, 2 or 0 */
#define SPANGPU_FUNCTOR_NONE        0
#define SPANGPU_FUNCTOR_V18         1
#define SPANGPU_FUNCTOR_ADEMCO      2

#define SPANGPU_TP_TWIST            0x01
#define SPANGPU_TP_REVERSE_TWIST    0x02
#define SPANGPU_TP_THRESHOLD        0x04

/* One completed detection block of one channel, decoded from the device records. */
typedef struct
{
    int32_t channel;
    int32_t block;              /* index of the block within the last spangpu_bank_rx() call           */
    int32_t hit;                /* raw block decision: DTMF/Bell/R2 ASCII code or 0; super-tone k1     */
    int32_t code;               /* DTMF: debounced digit state after the block; Bell: accepted digit   */
                                /* or 0; R2: current digit; super-tone: k2                             */
    int32_t flags;              /* SPANGPU_BLK_*                                                       */
    int32_t duration;           /* DTMF realtime mode: duration reported with the event                */
    float energy;               /* DTMF / super-tone: total block energy (sum x*x)                     */
} spangpu_block_t;

#define SPANGPU_BLK_VALID           0x01    /* a block completed in this slot                          */
#define SPANGPU_BLK_CHANGE          0x02    /* DTMF: debouncer changed state (dtmf.c:304)              */
#define SPANGPU_BLK_REPORT          0x04    /* DTMF realtime: a report is due (dtmf.c:312); Bell: digit
                                               accepted (bell_r2_mf.c:629-655); R2: digit changed      */
#define SPANGPU_BLK_TONE_OFF        0x08    /* DTMF realtime: the report is a tone-off (level -99)     */

/* ---- library -------------------------------------------------------------------- */
SPANGPU_API int spangpu_device_count(void);
/* Measurement aid: the streaming read rate a plain kernel reaches on this device (bytes > 256 MB for an HBM figure). */
SPANGPU_API int spangpu_probe_stream_read(int device, size_t bytes, int reps, double *gb_per_s);
SPANGPU_API const char *spangpu_last_error(void);
SPANGPU_API const char *spangpu_version(void);
SPANGPU_API float spangpu_goertzel_fac(float freq_hz);
/* Tuning knob for the tone banks: lanes per channel used by the kernels (results are
   identical): 1 = one channel per lane, 2 = a channel's bins split over two lanes (more
   wavefronts for small banks), 0 = choose from the bank size (default). */
SPANGPU_API int spangpu_tune_lanes_per_channel(int lpc);
/* Tuning knob for the tone banks: which kernel family serves spangpu_bank_rx() (results are identical):
   0 = choose per call (a streaming kernel for channel-major frames with 16-byte aligned rows -- with a loader wave
   per workgroup for banks of up to 393 216 channels, without for larger ones -- and the general kernel otherwise),
   1 = always the general kernel, 2 = streaming with loader waves where eligible, 3 = streaming without. */
SPANGPU_API int spangpu_tune_tone_kernel(int variant);

/* ---- banks ------------------------------------------------------------------------ */
SPANGPU_API int spangpu_bank_create(spangpu_bank_t **bank, int device, int kind, int n_channels,
                                    const void *params, size_t params_size);
SPANGPU_API int spangpu_bank_destroy(spangpu_bank_t *bank);
SPANGPU_API int spangpu_bank_kind(const spangpu_bank_t *bank);
SPANGPU_API int spangpu_bank_device(const spangpu_bank_t *bank);
SPANGPU_API int spangpu_bank_channels(const spangpu_bank_t *bank);
/* Use a caller-owned HIP stream (hipStream_t passed as void*); NULL = the bank's own stream. */
SPANGPU_API int spangpu_bank_set_stream(spangpu_bank_t *bank, void *hip_stream);
SPANGPU_API void *spangpu_bank_get_stream(spangpu_bank_t *bank);
/* Queue mode.  queues = 2: every launch the streaming kernel serves is cut in two ranges of channels, the second on a stream
   (hardware queue) the bank owns, so that one half's launch boundary, start burst and write-back lie under the other half's
   steady state; 1: one launch on the bank's stream (the default); 0: the library's choice (two from 131 072 channels: that is
   where it pays, profiles/r5_probe_mq.log -- but only for a bank on its own stream: on a caller's stream the library cannot
   see what produces the frames, and chooses one).  Returns the number of queues now in use, or a negative error.  Results are those
   of one launch, bit for bit.  Ordering: consecutive spangpu_bank_rx*() calls with nothing else between them run the two
   queues free of each other (no event per tick: that is the gain) -- so in queue mode a device-resident frame must be complete
   when the call is made, or have been put on the bank's stream before the last spangpu_bank_get_stream() / _join().  Every
   spangpu_bank_* call that reads results, edits state, waits or hands out the stream joins the second queue into the bank's
   stream first, and the launch after it starts behind whatever the bank's stream then holds.  A caller that puts work of its
   own on the bank's stream behind a launch (an event, a collective reading the records buffer) calls spangpu_bank_join() first. */
SPANGPU_API int spangpu_bank_set_queues(spangpu_bank_t *bank, int queues);
SPANGPU_API int spangpu_bank_join(spangpu_bank_t *bank);

/* Advance every channel by `samples` samples.  `amp` is int16 PCM; for
   CHANNEL_MAJOR, channel c starts at amp + c*stride (stride in samples, must be a
   multiple of 8 and >= samples rounded up to 8 for device buffers); for
   SAMPLE_MAJOR, sample s of channel c is amp[s*stride + c].  Asynchronous: returns
   after the launch is queued on the bank stream.  Returns 0, like dtmf_rx(). */
SPANGPU_API int spangpu_bank_rx(spangpu_bank_t *bank, const int16_t *amp, int mem, int layout,
                                int samples, long long stride);
/* spangpu_bank_rx() for G.711 input: one A-law / u-law byte per sample, channel-major, decoded on the device
   (alaw_to_linear / ulaw_to_linear, src/spandsp/g711.h:165-175,239-252).  stride in bytes (= samples). */
#define SPANGPU_G711_ALAW           1
#define SPANGPU_G711_ULAW           2
SPANGPU_API int spangpu_bank_rx_g711(spangpu_bank_t *bank, const uint8_t *codes, int mem, int law, int samples, long long stride);
/* A tick in which not every channel has a frame (or frames differ in length): channel c takes part with lens[c]
   samples of row c of amp[n_channels][stride]; 0 = the channel sits the call out, its detector exactly as it was (what
   the reference does for a channel whose xxx_rx() is not called).  lens[] is host memory; amp as `mem` says. */
SPANGPU_API int spangpu_bank_rx_var(spangpu_bank_t *bank, const int16_t *amp, int mem, const int32_t *lens, int max_samples,
                                    long long stride);
/* dtmf_rx_parms() (src/dtmf.c:421-445) for ONE channel of a DTMF bank: filter_dialtone (< 0 = leave), twist_db,
   reverse_twist_db, threshold_dbm0 under set_mask; the other fields of params are ignored. */
SPANGPU_API int spangpu_bank_set_channel_params(spangpu_bank_t *bank, int channel, const spangpu_tone_params_t *params,
                                                size_t params_size);

/* Advance several banks (<= 4, same device, device-resident channel-major frames) in one call.  Banks on ONE stream share ONE
   kernel launch: a tick of a mixed population of small banks then pays the launch and ramp-up cost once.  DTMF (no dial-tone
   filter), Bell MF, R2 MF and super-tone banks can share a launch.  Banks that were given streams of their own
   (spangpu_bank_set_stream) get a launch each, on their streams: with banks large enough to fill the chip between them the
   free-running hardware queues overlap one bank's launch boundary and start burst with the others' steady state (BASELINE
   configs[2], 131 072 channels in three banks: 23.7 us a tick as one launch, 18.4 us on three streams; the caller joins the
   streams when it reads results: spangpu_bank_blocks() etc. wait on the bank's own stream).  Banks are grouped by stream:
   those that share one (spangpu_bank_set_stream with the same stream) share a launch on it.  strides may be NULL (= samples). */
SPANGPU_API int spangpu_banks_rx(spangpu_bank_t *const *banks, const int16_t *const *amps, int n_banks, int samples,
                                 const long long *strides);
/* Gives every bank of the list (at most 4) a stream of its own (owned by the bank) on a hardware queue that is not another
   bank's: the runtime binds fresh streams to its few hardware queues in an order a caller does not control, and two banks
   that land on one queue run one after the other.  Candidate streams are probed in pairs with a 200 us spin kernel each (on
   two queues they end together) until every bank has one that runs beside all the others': a few milliseconds, once.
   Returns the number of banks for which that was proven.  Use before spangpu_banks_rx() for its launch-per-bank form. */
SPANGPU_API int spangpu_banks_own_queues(spangpu_bank_t *const *banks, int n_banks);
/* Evaluate the current (partial) block of every channel now and restart it: what
   goertzel_result() called mid-block does (tone_detect.c:160-205).  The results are read
   with spangpu_bank_blocks() / spangpu_bank_trace() as after spangpu_bank_rx(). */
SPANGPU_API int spangpu_bank_force_block(spangpu_bank_t *bank);
/* Wait for everything queued on the bank's stream. */
SPANGPU_API int spangpu_bank_sync(spangpu_bank_t *bank);
/* After spangpu_bank_rx(): copy the block records of the last call to the host
   and decode them in (channel, block) order.  Only VALID slots are returned. `out` may be
   NULL to query the count.  Returns the number of records or a negative error. */
SPANGPU_API int spangpu_bank_blocks(spangpu_bank_t *bank, spangpu_block_t *out, int max);
/* Queue a device-to-device copy of the last call's raw record words ([blocks][channel] uint32,
   hit | code<<8 | flags<<16) into a caller-owned device buffer (e.g. the send buffer of an RCCL
   gather).  Returns the number of bytes, or a negative error. */
/* Following launches write their block records directly into a caller-owned device buffer (e.g. an RCCL send
   buffer); NULL restores the bank's own.  Size: ceil(samples/block) * n_channels words per frame to come. */
SPANGPU_API int spangpu_bank_set_records_buffer(spangpu_bank_t *bank, void *device_buffer, size_t bytes);
SPANGPU_API long long spangpu_bank_copy_records(spangpu_bank_t *bank, void *dst_device, size_t dst_bytes);
/* The digits of the last spangpu_bank_rx() as a compact device-resident list, written on the bank's stream: dst[0] = how
   many blocks accepted a digit (SPANGPU_BLK_CHANGE with a non-zero code), dst[1 + i] = channel | code << 20 | block << 28
   for the first cap_entries of them, in no particular order (dst holds 1 + cap_entries words; a count above cap_entries
   says the list was cut).  A separate small kernel over the records of the last launch (two stream operations). */
SPANGPU_API int spangpu_bank_digit_events(spangpu_bank_t *bank, uint32_t *dst_device, int cap_entries);

/* ---- The pipelined host-buffer path (csrc/feed_api.hip) -----------------------------------------------------------------
   For a caller whose frames live in host memory (what N spandsp callers hold between dtmf_rx() calls): a ring of `depth`
   slots, each a pinned host buffer the caller's receive path writes a frame into, so that tick t + 1's copy to the device
   overlaps tick t's kernel and the way back of its digits.  Per tick:
       buf = spangpu_feed_acquire(feed);   write the frame: channel c at buf + c*spangpu_feed_stride(feed) samples
       spangpu_feed_commit(feed, samples); returns at once
       n = spangpu_feed_collect(feed, &entries);   (as late as `depth` - 1 ticks later) the digits of the oldest tick:
                                                   channel | digit << 20 | block << 28, as spangpu_bank_digit_events()
   law = 0: 16 bit linear PCM; SPANGPU_G711_ALAW / _ULAW: one byte per sample, decoded on the device (half the PCIe volume).
   Replaces the per-call sequence of dtmf_rx() + dtmf_rx_get() (src/dtmf.c:164-347, 481-519) for a whole bank. */
/* `device` must be the bank's own device (refused otherwise).  The feed uses the bank until it is destroyed: destroy the feed
   first.  max_samples may not exceed what spangpu_bank_digit_events() can report in one tick (16 blocks per channel). */
typedef struct spangpu_feed_s spangpu_feed_t;
SPANGPU_API int spangpu_feed_create(spangpu_feed_t **feed, spangpu_bank_t *bank, int device, int max_samples, int law, int depth);
SPANGPU_API int spangpu_feed_destroy(spangpu_feed_t *feed);
SPANGPU_API long long spangpu_feed_stride(const spangpu_feed_t *feed);
SPANGPU_API void *spangpu_feed_acquire(spangpu_feed_t *feed);
SPANGPU_API int spangpu_feed_commit(spangpu_feed_t *feed, int samples);
SPANGPU_API int spangpu_feed_collect(spangpu_feed_t *feed, const uint32_t **entries);
SPANGPU_API int spangpu_feed_outstanding(const spangpu_feed_t *feed);
/* A caller's tick loop in C, for measurements: `ticks` x { acquire (the slots keep the frames they hold), commit, collect
   with `lag` ticks (1 .. depth - 1) between a commit and the collect of its digits }, then the rest collected.
   *elapsed_ms = wall time of the loop, *digits = entries collected. */
SPANGPU_API int spangpu_feed_run(spangpu_feed_t *feed, int samples, int ticks, int lag, double *elapsed_ms, long long *digits);
/* One byte per block and channel, written by the detector kernel itself beside its records (no extra launch): launches
   from now on fill dev_ptr as digits[block][channel] = the digit the block delivered, 0 = none (DTMF: digit accepted; Bell
   MF / R2 MF: the digit of a report).  What a multi-GPU run gathers: a quarter of the record words.  NULL turns it off. */
SPANGPU_API int spangpu_bank_set_digits_buffer(spangpu_bank_t *bank, void *dev_ptr, size_t bytes);
/* ... into n_slices slices of slice_bytes, successive launches filling successive slices round and round: one call sets up
   a reporting interval of n_slices steps. */
SPANGPU_API int spangpu_bank_set_digits_ring(spangpu_bank_t *bank, void *dev_ptr, size_t slice_bytes, int n_slices);
/* Super-tone banks: the cadence matching of super_tone_rx() (src/super_tone_rx.c:164-228 test_cadence() and :364-448, the
   tail of super_tone_chunk()) on the device, for callers that want tone reports and not block records.  The descriptor is
   given as super_tone_rx_add_tone() / _add_element() build it (src/super_tone_rx.c:125-162): tone_elems[t] = how many
   elements tone t has, elems = all of them in order, f1 / f2 = the BIN numbers the frequencies resolved to (-1 = none, the
   numbering of the bank's bin_fac[]), min_ms / max_ms in milliseconds (max 0 = no upper limit).  Giving cadences again
   replaces the tones and keeps every channel's run history (a tone being followed is forgotten: its number belongs to
   the old set).  Events of one launch, per channel in the order the reference
   would call back, two words each at events[(slot*n_channels + channel)*2], counts[channel] of them:
     word 0 = kind | (f1 + 1) << 8 | (f2 + 1) << 16 | block << 24, word 1 = tone (kind 1) or milliseconds (kind 3)
     SPANGPU_CADENCE_TONE_ON   tone_callback(user, tone, -10, 0): the newest runs spell out a tone
     SPANGPU_CADENCE_TONE_OFF  tone_callback(user, -1, -10, 0): the cadence followed broke
     SPANGPU_CADENCE_SEGMENT   segment_callback(user, f1, f2, ms): a run ended (only with want_segments)
   The streaming detector kernel matches the cadences itself, in its epilogue; for launches it does not serve (sample-major
   or unaligned frames, ragged lengths, more than 16 bins, spangpu_banks_rx()) spangpu_bank_cadence_run() queues a matcher
   launch over the records of the last spangpu_bank_rx*() on the bank's stream (once per launch; the next spangpu_bank_rx*()
   does it itself if nobody has, so every block is counted whichever kernel served its frame -- only the events of a launch
   nobody asked about are not kept).  It returns the slots per channel; spangpu_bank_cadence_events() does that if it was not done, waits and hands out pinned host copies;
   spangpu_bank_cadence_device() the device buffers themselves (for a gather). */
#define SPANGPU_CADENCE_TONE_ON     1
#define SPANGPU_CADENCE_TONE_OFF    2
#define SPANGPU_CADENCE_SEGMENT     3
typedef struct
{
    int32_t f1;
    int32_t f2;
    int32_t min_ms;
    int32_t max_ms;
} spangpu_cadence_elem_t;
SPANGPU_API int spangpu_bank_set_cadences(spangpu_bank_t *bank, const int32_t *tone_elems, int n_tones,
                                          const spangpu_cadence_elem_t *elems, int want_segments);
SPANGPU_API int spangpu_bank_cadence_run(spangpu_bank_t *bank);
SPANGPU_API int spangpu_bank_cadence_events(spangpu_bank_t *bank, const uint32_t **events, const int32_t **counts);
SPANGPU_API int spangpu_bank_cadence_device(spangpu_bank_t *bank, const uint32_t **events_dev, const int32_t **counts_dev);
/* The same events as one compact list (what a host consumer of a large bank wants: a tick of 65 536 lines has a few hundred
   to a few thousand reports, the slot arrays are 3 MB): returns how many, *list = that many (channel, word 0, word 1) triples
   in pinned host memory; a channel's events are together and in order, the channels in no particular order.  Runs the
   matcher if that was not done, waits. */
SPANGPU_API int spangpu_bank_cadence_list(spangpu_bank_t *bank, const uint32_t **list);
/* One channel's history back to that of a new detector (channel -1: all of them). */
SPANGPU_API int spangpu_bank_cadence_reset(spangpu_bank_t *bank, int channel);
/* A channel's matcher state as words: 0-1 the pair of the last block, 2 tone followed (-1 none), 3 turn (elements gone by
   since it was recognised), 4-13 the pairs of the ten newest runs (f1 & 0xFFFF | f2 << 16), the current one first, 14-23
   their lengths in blocks. */
SPANGPU_API int spangpu_bank_cadence_state_words(void);
SPANGPU_API int spangpu_bank_cadence_get_state(spangpu_bank_t *bank, int channel, int32_t *words);
SPANGPU_API int spangpu_bank_cadence_set_state(spangpu_bank_t *bank, int channel, const int32_t *words);
/* Parity / diagnostics tap: per-block Goertzel energies of the last call, laid out
   [block][bin][channel] (bank created with trace=1).  Returns blocks-per-call. */
SPANGPU_API int spangpu_bank_trace(spangpu_bank_t *bank, float *energies, size_t max_floats);
/* Re-initialise one channel (what xxx_rx_init() on a live object does), or only
   its filters (what dtmf_rx_fillin() does, dtmf.c:363-379) when fillin_only != 0. */
SPANGPU_API int spangpu_bank_reset_channel(spangpu_bank_t *bank, int channel, int fillin_only);
/* Import / export the device state of one channel as flat arrays (migration and
   differential tests).  Layout documented per kind in DESIGN.md. */
SPANGPU_API int spangpu_bank_get_state(spangpu_bank_t *bank, int channel, float *fstate, int max_f,
                                       int32_t *istate, int max_i);
SPANGPU_API int spangpu_bank_set_state(spangpu_bank_t *bank, int channel, const float *fstate, int n_f,
                                       const int32_t *istate, int n_i);
/* Time, in milliseconds, spent by the kernels of the last spangpu_bank_rx() call
   (HIP events on the bank stream); negative if unavailable. */
SPANGPU_API float spangpu_bank_last_kernel_ms(spangpu_bank_t *bank);
/* Bracket each spangpu_bank_rx() launch with HIP events on the bank stream (off by default). */
SPANGPU_API int spangpu_bank_set_timing(spangpu_bank_t *bank, int on);
/* Bins compiled into the kernel this bank uses (>= the requested n_bins; trace stride). */
SPANGPU_API int spangpu_bank_bins(const spangpu_bank_t *bank);

/* ---- echo canceller banks --------------------------------------------------------------
 * N independent G.168 line echo cancellers, state resident in HBM (per channel: 32-bit
 * LMS taps, four 16-bit tap sets, FIR history, ~40 control words).  Replaces, per channel:
 *   spangpu_echo_create()         echo_can_init()            src/echo.c:254-301, src/spandsp/echo.h:145
 *   spangpu_echo_update()         echo_can_hpf_tx() + echo_can_update() per sample
 *                                                            src/echo.c:421-669, src/spandsp/echo.h:176-183
 *   spangpu_echo_adaption_mode()  echo_can_adaption_mode()   src/echo.c:324-328
 *   spangpu_echo_flush()          echo_can_flush()           src/echo.c:331-372
 *   spangpu_echo_destroy()        echo_can_free()            src/echo.c:309-322
 * The adaption-mode bits are the reference's (src/spandsp/echo.h:118-127). */
typedef struct spangpu_echo_s spangpu_echo_t;

#define SPANGPU_ECHO_SCALARS        48      /* control words per channel in get/set_state */

SPANGPU_API int spangpu_echo_create(spangpu_echo_t **ec, int device, int n_channels, int taps, int adaption_mode);
SPANGPU_API int spangpu_echo_destroy(spangpu_echo_t *ec);
SPANGPU_API int spangpu_echo_channels(const spangpu_echo_t *ec);
SPANGPU_API int spangpu_echo_taps(const spangpu_echo_t *ec);
SPANGPU_API int spangpu_echo_lanes_per_channel(const spangpu_echo_t *ec);     /* the kernel mapping the bank runs: 16, 8, 4 or 2 */
SPANGPU_API int spangpu_echo_set_stream(spangpu_echo_t *ec, void *hip_stream);
SPANGPU_API int spangpu_echo_sync(spangpu_echo_t *ec);
/* Run `samples` samples of every channel: clean[c][i] = echo_can_update(ec_c, tx'[c][i], rx[c][i])
   with tx' = echo_can_hpf_tx(ec_c, tx) when use_hpf_tx != 0 (the calling sequence of
   tests/echo_tests.c:577-594).  Channel c's samples start at tx/rx/clean + c*stride. */
SPANGPU_API int spangpu_echo_update(spangpu_echo_t *ec, const int16_t *tx, const int16_t *rx, int16_t *clean,
                                    int mem, int samples, long long stride, int use_hpf_tx);
/* As spangpu_echo_update(); tx_out (may be NULL) receives the transmit samples after echo_can_hpf_tx(), i.e. what the
   caller has to send to the line when ECHO_CAN_USE_TX_HPF is set (src/echo.c:663-669). */
SPANGPU_API int spangpu_echo_update_tx(spangpu_echo_t *ec, const int16_t *tx, const int16_t *rx, int16_t *clean, int16_t *tx_out,
                                       int mem, int samples, long long stride, int use_hpf_tx);
/* echo_can_hpf_tx() on its own, for callers that filter tx before they have the matching rx (host buffers). */
/* Tuning / A-B testing: lanes per channel of echo banks created from now on (0 = by length, 4, 8 or 16); results are identical. */
SPANGPU_API int spangpu_tune_echo_lanes_per_channel(int lanes);
SPANGPU_API int spangpu_echo_hpf_tx(spangpu_echo_t *ec, const int16_t *tx, int16_t *out, int samples, long long stride);
SPANGPU_API int spangpu_echo_adaption_mode(spangpu_echo_t *ec, int channel, int adaption_mode);
SPANGPU_API int spangpu_echo_flush(spangpu_echo_t *ec, int channel);
/* Per-channel line statistics (the result a multi-GPU echo run gathers: SURVEY 8(d)-5).  Once enabled, every update
   also accumulates, per channel, the energy of the received signal and of the cleaned signal -- ERLE = 10 log10 of their
   ratio, what level_measurements_update() of tests/echo_tests.c:577-594 watches -- and the CRC-32 (zlib) of the clean
   stream as int16 little-endian bytes, for exactness checks against the CPU path. */
typedef struct
{
    uint64_t sum_rx2;           /* sum of rx[i]^2 since the sums were last reset        */
    uint64_t sum_clean2;        /* sum of clean[i]^2                                    */
    uint32_t crc;               /* CRC-32 of every clean sample since the CRC was reset */
    uint32_t samples;           /* samples in the sums                                  */
} spangpu_echo_stats_t;
#define SPANGPU_ECHO_STATS_SUMS     1
#define SPANGPU_ECHO_STATS_CRC      2
/* enable: 0 off; 1 energy sums and the CRC of the clean stream, by a pass of their own behind every update; 2 the energy sums
   only (what ERLE needs), added up by the update kernel itself from the samples it already holds -- no second pass. */
SPANGPU_API int spangpu_echo_stats(spangpu_echo_t *ec, int enable);
SPANGPU_API int spangpu_echo_stats_reset(spangpu_echo_t *ec, int what);    /* SPANGPU_ECHO_STATS_SUMS | _CRC */
SPANGPU_API int spangpu_echo_stats_get(spangpu_echo_t *ec, int first_channel, int n, spangpu_echo_stats_t *out);
/* ERLE in dB of every channel from the sums, into erle_db[n_channels] (host or device memory: a device buffer can be the
   send buffer of an RCCL gather).  0 while nothing was measured, 120 when the residue is exactly zero. */
SPANGPU_API int spangpu_echo_erle(spangpu_echo_t *ec, float *erle_db, int mem);
/* One channel's state in the reference's terms: control words (order: DESIGN.md), taps32[taps],
   taps16[4][taps], FIR history[taps] in its physical circular order.  Any pointer may be NULL. */
SPANGPU_API int spangpu_echo_get_state(spangpu_echo_t *ec, int channel, int32_t *scalars, int32_t *taps32,
                                       int16_t *taps16, int16_t *history);
SPANGPU_API int spangpu_echo_set_state(spangpu_echo_t *ec, int channel, const int32_t *scalars, const int32_t *taps32,
                                       const int16_t *taps16, const int16_t *history);

/* ---- Modem receiver banks --------------------------------------------------------------
 * N independent V.29 (9600 / 7200 / 4800 bps), V.27ter (4800 / 2400 bps) or V.17 (14400 / 12000 /
 * 9600 / 7200 bps, plus V.32bis 4800) receivers, state resident in HBM.  Replaces, per channel:
 *   spangpu_modem_create()   v29_rx_init(NULL, bit_rate, put_bit, user)      src/v29rx.c:1100-1131, src/spandsp/v29rx.h:151
 *                            v27ter_rx_init(NULL, bit_rate, put_bit, user)   src/v27ter_rx.c:1162-1190, src/spandsp/v27ter_rx.h:84
 *                            v17_rx_init(NULL, bit_rate, put_bit, user)      src/v17rx.c:1502-1535, src/spandsp/v17rx.h:241
 *   spangpu_modem_rx()       v29_rx(s, amp, len)                             src/v29rx.c:867-965,  src/spandsp/v29rx.h:187
 *                            v27ter_rx(s, amp, len)                          src/v27ter_rx.c:863-1028, src/spandsp/v27ter_rx.h:120
 *                            v17_rx(s, amp, len)                             src/v17rx.c:1212-1318, src/spandsp/v17rx.h:279
 *   spangpu_modem_restart()  v29_rx_restart / v27ter_rx_restart(s, rate, false)   src/v29rx.c:1019, src/v27ter_rx.c:1091
 *   spangpu_modem_events()   the put_bit / status callback stream            src/v29rx.c:171-178,380-397
 * The demodulated bit stream and the SIG_STATUS_* events (spandsp/async.h:66-103) are delivered
 * per channel, in the order the reference would have made its callbacks: for channel c,
 * counts[c] entries at events + c*cap, each 0/1 (a descrambled data bit) or a negative
 * SIG_STATUS_* code. */
typedef struct spangpu_modem_s spangpu_modem_t;

SPANGPU_API int spangpu_modem_create(spangpu_modem_t **modem, int device, int kind, int n_channels, int bit_rate);
SPANGPU_API int spangpu_modem_destroy(spangpu_modem_t *modem);
SPANGPU_API int spangpu_modem_channels(const spangpu_modem_t *modem);
SPANGPU_API int spangpu_modem_set_stream(spangpu_modem_t *modem, void *hip_stream);
SPANGPU_API int spangpu_modem_sync(spangpu_modem_t *modem);
SPANGPU_API int spangpu_modem_rx(spangpu_modem_t *modem, const int16_t *amp, int mem, int samples, long long stride);
/* A tick in which not every receiver has a frame (or frames differ in length): channel c takes lens[c] samples of row c;
   0 = it sits the call out, untouched.  lens[] is host memory. */
SPANGPU_API int spangpu_modem_rx_var(spangpu_modem_t *modem, const int16_t *amp, int mem, const int32_t *lens, int max_samples,
                                     long long stride);
SPANGPU_API int spangpu_modem_events(spangpu_modem_t *modem, const int8_t **events, const int32_t **counts);
/* The last call's events device to device (for a gather across GPUs): dst = int32 counts[n_ch], int8 events[n_ch][per_channel].
   Asynchronous on the bank's stream. */
SPANGPU_API int spangpu_modem_copy_events(spangpu_modem_t *modem, void *dev_dst, size_t dst_bytes, int per_channel);
/* xxx_rx_set_qam_report_handler() (src/v29rx.c:1149, v27ter_rx.c:1204, v17rx.c:1535): with the tap on, every channel's
   qam_report(user, constel, target, symbol) calls (v29rx.c:769-783, v27ter_rx.c:517,765-777, v17rx.c:1117-1131) of an rx
   call are recorded: counts[c] records of seven words at records + c*cap*7 = {put_bit / status calls before it in
   this rx call, 1 if constel and target were NULL, symbol, constel re, im, target re, im as binary32 bits}.
   spangpu_modem_qam_reports() returns cap.  Bit-exact with the reference's float build. */
SPANGPU_API int spangpu_modem_qam_tap(spangpu_modem_t *modem, int enable);
SPANGPU_API int spangpu_modem_qam_reports(spangpu_modem_t *modem, const uint32_t **records, const int32_t **counts);
SPANGPU_API int spangpu_modem_state_words(int kind, int *n_floats, int *n_ints);
SPANGPU_API int spangpu_modem_get_state(spangpu_modem_t *modem, int channel, uint32_t *words);
SPANGPU_API int spangpu_modem_set_state(spangpu_modem_t *modem, int channel, const uint32_t *words);
SPANGPU_API int spangpu_modem_restart(spangpu_modem_t *modem, int channel);
/* v29_rx_restart(s, rate, old_train) / v27ter_rx_restart(s, rate, old_train) / v17_rx_restart(s, rate, short_train) */
SPANGPU_API int spangpu_modem_restart_ex(spangpu_modem_t *modem, int channel, int bit_rate, int train_flag);
/* xxx_rx_fillin(s, len) (v29rx.c:967) and xxx_rx_set_signal_cutoff(s, cutoff) (v29rx.c:163) */
SPANGPU_API int spangpu_modem_fillin(spangpu_modem_t *modem, int channel, int len);
/* channel -1: every channel of the bank (fax_modems.c:416 gives each of its V.29 receivers -45.5 dBm0; the default, -28.5 dBm0
 * (v29rx.c:1129), leaves a line below some -26 dBm0 undetected) */
SPANGPU_API int spangpu_modem_set_signal_cutoff(spangpu_modem_t *modem, int channel, float cutoff_dbm0);
/* ... and every channel with a cutoff of its own: cutoff_dbm0[n_channels] */
SPANGPU_API int spangpu_modem_set_signal_cutoffs(spangpu_modem_t *modem, const float *cutoff_dbm0);
/* The constant tables the modem receivers use, as built by this library (host code; see modem_api.hip for `which`). */
SPANGPU_API int spangpu_modem_table(int which, float *out, int max);
/* Tuning / A-B testing: how the receiver kernels map channels to lanes from now on: 0 = by bank size (four lanes per channel
   below 65 536 channels), 1 = one channel per lane, 4 = four lanes per channel, 16 channels per wavefront (8 is accepted
   and means 4).  Results are identical. */
SPANGPU_API int spangpu_tune_modem_mapping(int mapping);
SPANGPU_API int spangpu_v17_rx_maps(uint8_t *maps, uint8_t *map_4800);

/* ---- signal source banks ------------------------------
 * N independent cadenced tone generators / digit senders, one per channel, state in HBM.
 * One spangpu_txbank_tx() call writes one frame of every channel with one kernel launch, so a
 * benchmark or a loop-back test can synthesise its input where the detector banks read it.
 * Output is bit-exact with the reference's float build on x86-64 (lfastrintf() truncates there).
 *   spangpu_txbank_create(SPANGPU_TX_TONE_GEN) + _tone()   tone_gen_descriptor_init()+tone_gen_init()
 *                                                          src/tone_generate.c:60-120,232-262
 *   spangpu_txbank_tx() on a tone_gen bank                 tone_gen() x N          src/tone_generate.c:128-229
 *   SPANGPU_TX_DTMF: _create/_put/_set_level/_set_timing/_tx   dtmf_tx_init/_put/_set_level/_set_timing/dtmf_tx
 *                                                          src/dtmf.c:551-660
 *   SPANGPU_TX_BELL_MF: _create/_put/_tx                   bell_mf_tx_init/_put/bell_mf_tx   src/bell_r2_mf.c:306-381
 *   SPANGPU_TX_R2_MF_FWD/_BACK: _create/_put/_tx           r2_mf_tx_init/_put/r2_mf_tx       src/bell_r2_mf.c:399-487
 * Differences from the reference, both at the edge of a frame: samples past the returned length are
 * zero-filled (the reference leaves them untouched), and digits_tx_callback_t (a host callback asked for
 * more digits when the queue runs dry, dtmf.c:566-573) is not replayed -- refill with _put() between frames.
 */
#define SPANGPU_TX_TONE_GEN         1
#define SPANGPU_TX_DTMF             2
#define SPANGPU_TX_BELL_MF          3
#define SPANGPU_TX_R2_MF_FWD        4
#define SPANGPU_TX_R2_MF_BACK       5

typedef struct spangpu_txbank_s spangpu_txbank_t;

/* The arguments of tone_gen_descriptor_init() (src/spandsp/tone_generate.h): two tones in Hz with levels in
   dBm0 (f2 < 0: tone 1 amplitude-modulated by |f2| at l2 percent), four cadence times in ms, repeat flag. */
typedef struct
{
    int f1;
    int l1;
    int f2;
    int l2;
    int d1;
    int d2;
    int d3;
    int d4;
    int repeat;
} spangpu_tone_desc_t;

SPANGPU_API int spangpu_txbank_create(spangpu_txbank_t **bank, int device, int kind, int n_channels);
SPANGPU_API void spangpu_txbank_destroy(spangpu_txbank_t *bank);
SPANGPU_API int spangpu_txbank_channels(const spangpu_txbank_t *bank);
SPANGPU_API int spangpu_txbank_set_stream(spangpu_txbank_t *bank, void *hip_stream);
SPANGPU_API int spangpu_txbank_sync(spangpu_txbank_t *bank);
/* tone_gen_init() of one descriptor on channels [first, first + n) of a SPANGPU_TX_TONE_GEN bank */
SPANGPU_API int spangpu_txbank_tone(spangpu_txbank_t *bank, int first, int n, const spangpu_tone_desc_t *desc);
/* dtmf_tx_set_level(level, twist) / dtmf_tx_set_timing(on_ms, off_ms) on channels [first, first + n) */
SPANGPU_API int spangpu_txbank_set_level(spangpu_txbank_t *bank, int first, int n, int level, int twist);
SPANGPU_API int spangpu_txbank_set_timing(spangpu_txbank_t *bank, int first, int n, int on_time, int off_time);
/* dtmf_tx_put()/bell_mf_tx_put(digits, len) with the same digits on every channel of the range (len < 0: strlen);
   r2_mf_tx_put(digits[0]) on an R2 bank.  Returns 0, or -- as the reference does per channel -- the number of
   characters that did not fit (the largest over the range; channels without room queue nothing). */
SPANGPU_API int spangpu_txbank_put(spangpu_txbank_t *bank, int first, int n, const char *digits, int len);
/* Per-channel digits: channel first + i gets digits[i*stride .. i*stride + lens[i]); results[i] (may be NULL) is
   that channel's xxx_tx_put() return value. */
SPANGPU_API int spangpu_txbank_put_each(spangpu_txbank_t *bank, int first, int n, const char *digits, int stride,
                                        const int *lens, int *results);
/* One frame of every channel: pcm[channel*stride + i], i < samples; lens[channel] (may be NULL) = what the
   reference's xxx_tx() would have returned.  pcm and lens live where mem_kind says. */
SPANGPU_API int spangpu_txbank_tx(spangpu_txbank_t *bank, int mem_kind, int16_t *pcm, long long stride, int samples, int *lens);
/* Test / checkpoint access to one channel's generator state (layout: txgen_dev.hpp) */
SPANGPU_API int spangpu_txbank_state_words(void);
SPANGPU_API int spangpu_txbank_get_state(spangpu_txbank_t *bank, int channel, int32_t *words);

/* ---- FSK receiver banks --------------------------------
 * N independent non-coherent FSK receivers of one modem spec (V.21 channel 2 in synchronous mode is the
 * HDLC control channel that runs beside V.17 / V.29 / V.27ter in a FAX front end,
 * src/fax_modems.c:213-323).  Integer arithmetic throughout: results are bit-exact with the reference.
 *   spangpu_fsk_preset()                 preset_fsk_specs[]                      src/fsk.c:60-155
 *   spangpu_fsk_create()                 fsk_rx_init(NULL, spec, mode, put_bit, user)   src/fsk.c:723-742
 *   spangpu_fsk_rx()                     fsk_rx(s, amp, len) x N                 src/fsk.c:393-622
 *   spangpu_fsk_events()                 the put_bit stream: bits, SIG_STATUS_CARRIER_UP (-2) / _DOWN (-1), and in
 *                                        framed mode the received characters    src/fsk.c:343-391
 *   spangpu_fsk_restart()                fsk_rx_restart(s, spec, mode)           src/fsk.c:660-720
 *   spangpu_fsk_set_signal_cutoff()      fsk_rx_set_signal_cutoff()              src/fsk.c:270-276
 *   spangpu_fsk_set_frame_parameters()   fsk_rx_set_frame_parameters()           src/fsk.c:300-316
 *   spangpu_fsk_fillin()                 fsk_rx_fillin()                         src/fsk.c:625-657
 *   state words 26 / 27                  fsk_rx_get_parity_errors() / _framing_errors()   src/fsk.c:318-340
 * A bank runs one spec (its correlation window length is a property of the baud rate); the framing mode and
 * the frame parameters are per channel.
 */
#define SPANGPU_FSK_V21CH1          0
#define SPANGPU_FSK_V21CH2          1
#define SPANGPU_FSK_V23CH1          2
#define SPANGPU_FSK_V23CH2          3
#define SPANGPU_FSK_BELL103CH1      4
#define SPANGPU_FSK_BELL103CH2      5
#define SPANGPU_FSK_BELL202         6
#define SPANGPU_FSK_WEITBRECHT_4545 7
#define SPANGPU_FSK_WEITBRECHT_50   8
#define SPANGPU_FSK_WEITBRECHT_476  9
#define SPANGPU_FSK_V21CH1_110      10

#define SPANGPU_FSK_FRAME_MODE_ASYNC    0
#define SPANGPU_FSK_FRAME_MODE_SYNC     1
#define SPANGPU_FSK_FRAME_MODE_FRAMED   2

typedef struct spangpu_fsk_s spangpu_fsk_t;

/* fsk_spec_t (src/spandsp/fsk.h:85-100) without the name */
typedef struct
{
    int freq_zero;      /* Hz */
    int freq_one;
    int tx_level;       /* dBm0 (unused by the receiver) */
    int min_level;      /* dBm0: the carrier detect cutoff */
    int baud_rate;      /* baud x 100 */
} spangpu_fsk_spec_t;

SPANGPU_API int spangpu_fsk_preset(int which, spangpu_fsk_spec_t *spec);
SPANGPU_API int spangpu_fsk_create(spangpu_fsk_t **fsk, int device, int n_channels, const spangpu_fsk_spec_t *spec,
                                   int framing_mode);
SPANGPU_API void spangpu_fsk_destroy(spangpu_fsk_t *fsk);
SPANGPU_API int spangpu_fsk_channels(const spangpu_fsk_t *fsk);
SPANGPU_API int spangpu_fsk_set_stream(spangpu_fsk_t *fsk, void *hip_stream);
SPANGPU_API int spangpu_fsk_sync(spangpu_fsk_t *fsk);
SPANGPU_API int spangpu_fsk_rx(spangpu_fsk_t *fsk, const int16_t *amp, int mem, int samples, long long stride);
/* a tick in which channels are missing or bring short frames: lens[c] samples of row c; 0 = untouched (host array) */
SPANGPU_API int spangpu_fsk_rx_var(spangpu_fsk_t *fsk, const int16_t *amp, int mem, const int32_t *lens, int max_samples, long long stride);
/* events[channel*cap + i], i < counts[channel]; returns cap.  Valid until the next call on this bank. */
SPANGPU_API int spangpu_fsk_events(spangpu_fsk_t *fsk, const int16_t **events, const int32_t **counts);
/* Tuning / A-B testing: how many wavefronts work on 64 receivers (FSK banks, connect-tone banks and signalling-tone
   receiver banks made or run from now on): 0 = the library's choice, 1 = the whole receiver in one lane of one wavefront,
   2 = the receiver cut into two instruction streams on two wavefronts (fsk_dev.hpp).  Results are identical. */
SPANGPU_API int spangpu_tune_fsk_waves(int waves);
SPANGPU_API int spangpu_fsk_state_words(const spangpu_fsk_t *fsk);
SPANGPU_API int spangpu_fsk_get_state(spangpu_fsk_t *fsk, int channel, int32_t *words);
SPANGPU_API int spangpu_fsk_set_state(spangpu_fsk_t *fsk, int channel, const int32_t *words);
SPANGPU_API int spangpu_fsk_restart(spangpu_fsk_t *fsk, int channel, int framing_mode);
SPANGPU_API int spangpu_fsk_set_signal_cutoff(spangpu_fsk_t *fsk, int channel, float cutoff_dbm0);
SPANGPU_API int spangpu_fsk_set_frame_parameters(spangpu_fsk_t *fsk, int channel, int data_bits, int parity, int stop_bits);
SPANGPU_API int spangpu_fsk_fillin(spangpu_fsk_t *fsk, int channel, int len);

/* ---- modem connect tone banks ---------------------------
 * N detectors of one tone type: FAX CNG (1100 Hz), CED / ANS (2100 Hz) with its phase-reversal and 15 Hz AM
 * variants, Bell answer tone (2225 Hz), calling tone (1300 Hz), and the V.21 FAX preamble (HDLC flags on the
 * V.21 channel 2 bit stream), as run at the head of a FAX or data call beside the receivers above.
 *   spangpu_mct_create()     modem_connect_tones_rx_init(NULL, tone_type, callback, user)   src/modem_connect_tones.c:799-857
 *   spangpu_mct_rx()         modem_connect_tones_rx(s, amp, len) x N                       src/modem_connect_tones.c:521-785
 *   spangpu_mct_events()     the span_tone_report_func_t calls (tone, level, 0)             src/modem_connect_tones.c:416-435
 *   spangpu_mct_get()        modem_connect_tones_rx_get()                                   src/modem_connect_tones.c:793-797
 * tone_type takes the MODEM_CONNECT_TONES_* codes of src/spandsp/modem_connect_tones.h:57-90 (below).
 * use_callback = 0 replays the no-callback behaviour (reports latch `hit`, read with spangpu_mct_get()).
 */
#define SPANGPU_MCT_NONE                    0
#define SPANGPU_MCT_FAX_CNG                 1
#define SPANGPU_MCT_ANS                     2
#define SPANGPU_MCT_ANS_PR                  3
#define SPANGPU_MCT_ANSAM                   4
#define SPANGPU_MCT_ANSAM_PR                5
#define SPANGPU_MCT_FAX_PREAMBLE            6
#define SPANGPU_MCT_FAX_CED_OR_PREAMBLE     7
#define SPANGPU_MCT_BELL_ANS                8
#define SPANGPU_MCT_CALLING_TONE            9

typedef struct spangpu_mct_s spangpu_mct_t;

SPANGPU_API int spangpu_mct_create(spangpu_mct_t **mct, int device, int tone_type, int n_channels, int use_callback);
SPANGPU_API void spangpu_mct_destroy(spangpu_mct_t *mct);
SPANGPU_API int spangpu_mct_channels(const spangpu_mct_t *mct);
SPANGPU_API int spangpu_mct_set_stream(spangpu_mct_t *mct, void *hip_stream);
SPANGPU_API int spangpu_mct_sync(spangpu_mct_t *mct);
SPANGPU_API int spangpu_mct_rx(spangpu_mct_t *mct, const int16_t *amp, int mem, int samples, long long stride);
SPANGPU_API int spangpu_mct_rx_var(spangpu_mct_t *mct, const int16_t *amp, int mem, const int32_t *lens, int max_samples, long long stride);
/* events[(channel*cap + i)*2 + {0: tone, 1: level}], i < counts[channel]; returns cap.  Valid until the next call. */
SPANGPU_API int spangpu_mct_events(spangpu_mct_t *mct, const int32_t **events, const int32_t **counts);
SPANGPU_API int spangpu_mct_get(spangpu_mct_t *mct, int channel);
SPANGPU_API int spangpu_mct_state_words(const spangpu_mct_t *mct);
SPANGPU_API int spangpu_mct_get_state(spangpu_mct_t *mct, int channel, int32_t *words);
SPANGPU_API int spangpu_mct_set_state(spangpu_mct_t *mct, int channel, const int32_t *words);

/* ---- signalling tone banks (sig_tone.c) -----------------
 * N in-band signalling tone receivers, or senders, of one tone type: 2280 Hz (AC15 and relatives), 2600 Hz, or
 * 2400 Hz / 2600 Hz (SS5).  A receiver detects the tone(s) -- notch filters as guard filters, a sharp detector that
 * turns into a flat one on sustained tone, persistence checks -- and rewrites the frame in place: muted, passed, or
 * passed with the tone notched out.  A sender mutes or passes the frame and adds the tone(s), high level first.
 *   spangpu_sigtone_rx_create()    sig_tone_rx_init(NULL, tone_type, callback, user)    src/sig_tone.c:672-723
 *   spangpu_sigtone_rx_set_mode()  sig_tone_rx_set_mode(s, mode, duration)              src/sig_tone.c:666-669
 *   spangpu_sigtone_rx()           sig_tone_rx(s, amp, len) x N                         src/sig_tone.c:402-663
 *   spangpu_sigtone_rx_events()    the span_tone_report_func_t calls (signalling_state, 0, duration), :627-635
 *   spangpu_sigtone_tx_create()    sig_tone_tx_init(NULL, tone_type, callback, user)    src/sig_tone.c:348-382
 *   spangpu_sigtone_tx_set_mode()  sig_tone_tx_set_mode(s, mode, duration)              src/sig_tone.c:326-345
 *   spangpu_sigtone_tx()           sig_tone_tx(s, amp, len) x N                         src/sig_tone.c:246-323
 * tone_type and the mode bits are those of src/spandsp/sig_tone.h:57-88 (below).
 *
 * The reference's sender calls back from inside sig_tone_tx() when a mode's duration runs out
 * (SIG_TONE_TX_UPDATE_REQUEST), and the caller sets the next mode in that callback.  A bank stops each such channel at
 * that sample: spangpu_sigtone_tx() returns the number of channels whose callback is due, spangpu_sigtone_tx_requests()
 * says which, the caller sets their next modes and calls spangpu_sigtone_tx_continue() on the same frame until 0 comes
 * back.  (A caller that never sets a duration never sees a request.)
 */
#define SPANGPU_SIG_TONE_2280HZ             1
#define SPANGPU_SIG_TONE_2600HZ             2
#define SPANGPU_SIG_TONE_2400HZ_2600HZ      3

#define SPANGPU_SIG_TONE_1_PRESENT          0x001
#define SPANGPU_SIG_TONE_1_CHANGE           0x002
#define SPANGPU_SIG_TONE_2_PRESENT          0x004
#define SPANGPU_SIG_TONE_2_CHANGE           0x008
#define SPANGPU_SIG_TONE_TX_PASSTHROUGH     0x010
#define SPANGPU_SIG_TONE_RX_PASSTHROUGH     0x040
#define SPANGPU_SIG_TONE_RX_FILTER_TONE     0x080
#define SPANGPU_SIG_TONE_TX_UPDATE_REQUEST  0x100

typedef struct spangpu_sigtone_rx_s spangpu_sigtone_rx_t;
typedef struct spangpu_sigtone_tx_s spangpu_sigtone_tx_t;

SPANGPU_API int spangpu_sigtone_rx_create(spangpu_sigtone_rx_t **bank, int device, int tone_type, int n_channels);
SPANGPU_API void spangpu_sigtone_rx_destroy(spangpu_sigtone_rx_t *bank);
SPANGPU_API int spangpu_sigtone_rx_channels(const spangpu_sigtone_rx_t *bank);
SPANGPU_API int spangpu_sigtone_rx_set_stream(spangpu_sigtone_rx_t *bank, void *hip_stream);
SPANGPU_API int spangpu_sigtone_rx_sync(spangpu_sigtone_rx_t *bank);
/* channel < 0: every channel */
SPANGPU_API int spangpu_sigtone_rx_set_mode(spangpu_sigtone_rx_t *bank, int channel, int mode);
/* amp is read AND written: [n_channels] rows of `samples`, `stride` apart */
SPANGPU_API int spangpu_sigtone_rx(spangpu_sigtone_rx_t *bank, int16_t *amp, int mem, int samples, long long stride);
SPANGPU_API int spangpu_sigtone_rx_var(spangpu_sigtone_rx_t *bank, int16_t *amp, int mem, const int32_t *lens, int max_samples, long long stride);
/* events[(channel*cap + i)*3 + {0: sample of the call, 1: signalling_state, 2: duration}], i < counts[channel];
   returns cap.  Valid until the next call. */
SPANGPU_API int spangpu_sigtone_rx_events(spangpu_sigtone_rx_t *bank, const int32_t **events, const int32_t **counts);
SPANGPU_API int spangpu_sigtone_rx_state_words(const spangpu_sigtone_rx_t *bank);
SPANGPU_API int spangpu_sigtone_rx_get_state(spangpu_sigtone_rx_t *bank, int channel, int32_t *words);
SPANGPU_API int spangpu_sigtone_rx_set_state(spangpu_sigtone_rx_t *bank, int channel, const int32_t *words);
/* flat_detection_threshold, sharp_detection_threshold, detection_ratio as sig_tone_rx_init() computes them */
SPANGPU_API int spangpu_sigtone_rx_thresholds(const spangpu_sigtone_rx_t *bank, int32_t out[3]);

SPANGPU_API int spangpu_sigtone_tx_create(spangpu_sigtone_tx_t **bank, int device, int tone_type, int n_channels);
SPANGPU_API void spangpu_sigtone_tx_destroy(spangpu_sigtone_tx_t *bank);
SPANGPU_API int spangpu_sigtone_tx_channels(const spangpu_sigtone_tx_t *bank);
SPANGPU_API int spangpu_sigtone_tx_set_stream(spangpu_sigtone_tx_t *bank, void *hip_stream);
/* channel < 0: every channel */
SPANGPU_API int spangpu_sigtone_tx_set_mode(spangpu_sigtone_tx_t *bank, int channel, int mode, int duration);
/* host arrays of n_channels entries; channels whose modes[] entry is negative keep theirs */
SPANGPU_API int spangpu_sigtone_tx_set_modes(spangpu_sigtone_tx_t *bank, const int32_t *modes, const int32_t *durations);
/* both return the number of channels stopped at an update request (0: the frame is done), or a negative error */
SPANGPU_API int spangpu_sigtone_tx(spangpu_sigtone_tx_t *bank, int16_t *amp, int mem, int samples, long long stride);
SPANGPU_API int spangpu_sigtone_tx_continue(spangpu_sigtone_tx_t *bank, int16_t *amp, int mem, long long stride);
SPANGPU_API int spangpu_sigtone_tx_requests(spangpu_sigtone_tx_t *bank, const int32_t **request, const int32_t **stopped);
SPANGPU_API int spangpu_sigtone_tx_state_words(const spangpu_sigtone_tx_t *bank);
SPANGPU_API int spangpu_sigtone_tx_get_state(spangpu_sigtone_tx_t *bank, int channel, int32_t *words);

/* ---- modem transmitter banks ---------------------------
 * N V.29, V.27ter or V.17 modulators as device-side signal sources for the receiver banks: training (optionally with
 * the talker echo protection tone), then scrambled data.  Bit-exact with the reference's float build on x86-64.
 *   spangpu_modemtx_create(SPANGPU_V29)      v29_tx_init(NULL, bit_rate, tep, get_bit, user)      src/v29tx.c:406-434
 *   spangpu_modemtx_create(SPANGPU_V27TER)   v27ter_tx_init(NULL, bit_rate, tep, get_bit, user)   src/v27ter_tx.c:411-437
 *   spangpu_modemtx_create(SPANGPU_V17)      v17_tx_init(NULL, bit_rate, tep, get_bit, user)      src/v17tx.c:452-483
 *   spangpu_modemtx_tx()                     v29_tx() / v27ter_tx() / v17_tx() x N   src/v29tx.c:226-284, src/v27ter_tx.c:246-350,
 *                                                                                     src/v17tx.c:295-369
 *   spangpu_modemtx_power()                  v29_tx_power() / v27ter_tx_power()     src/v29tx.c:322-338, src/v27ter_tx.c:352-364
 *   spangpu_modemtx_restart()                v29_tx_restart() / v27ter_tx_restart() src/v29tx.c:365-404, src/v27ter_tx.c:384-409
 *   spangpu_modemtx_restart_ex()             v17_tx_restart(s, rate, tep, short_train)  src/v17tx.c:397-450 (and _power: :371-383)
 * The data bits of channel c come from a 15 bit LFSR (x^15 + x^14 + 1) seeded with seeds[c] (NULL: a seed per
 * channel is derived from its index); a get_bit() callback, and with it the end-of-data shutdown sequence, is
 * not replayed.
 */
typedef struct spangpu_modemtx_s spangpu_modemtx_t;

SPANGPU_API int spangpu_modemtx_create(spangpu_modemtx_t **tx, int device, int modem, int n_channels, int bit_rate, int tep,
                                       const uint32_t *seeds);
SPANGPU_API void spangpu_modemtx_destroy(spangpu_modemtx_t *tx);
SPANGPU_API int spangpu_modemtx_channels(const spangpu_modemtx_t *tx);
SPANGPU_API int spangpu_modemtx_set_stream(spangpu_modemtx_t *tx, void *hip_stream);
SPANGPU_API int spangpu_modemtx_sync(spangpu_modemtx_t *tx);
SPANGPU_API int spangpu_modemtx_power(spangpu_modemtx_t *tx, int channel, float power_dbm0);
/* Every channel's level (xxx_tx_power(), as above) and carrier frequency in Hz in one call; either array may be NULL (left
 * as it is).  The carrier frequency is a line model, not a reference API: v29_tx_init() fixes the carrier at 1700 Hz
 * (src/v29tx.c:431), v27ter / v17 at 1800 Hz; a few hertz either side stand for the frequency shift of a carrier system
 * between the modems (1700 Hz +- 7 Hz). */
SPANGPU_API int spangpu_modemtx_line(spangpu_modemtx_t *tx, const float *power_dbm0, const float *carrier_hz);
SPANGPU_API int spangpu_modemtx_restart(spangpu_modemtx_t *tx, int channel, int bit_rate, int tep);
SPANGPU_API int spangpu_modemtx_restart_ex(spangpu_modemtx_t *tx, int channel, int bit_rate, int tep, int short_train);
/* pcm[channel*stride + i], i < samples, where mem says; returns samples */
SPANGPU_API int spangpu_modemtx_tx(spangpu_modemtx_t *tx, int mem, int16_t *pcm, long long stride, int samples);
SPANGPU_API int spangpu_modemtx_state_words(void);
SPANGPU_API int spangpu_modemtx_get_state(spangpu_modemtx_t *tx, int channel, int32_t *words);
/* The pulse shaper tables as built by this library (host code): which = 0 V.29, 1 V.27ter 4800 bps, 2 V.27ter 2400 bps */
SPANGPU_API int spangpu_modemtx_table(int which, float *out, int max);

/* ------------------------------------------------------------------------------------------------------------
 * Noise source banks: batched awgn() -- one independent Gaussian noise generator per channel (Numerical Recipes
 * ran1 + polar Box-Muller in binary64, as the reference), written to, or mixed with saturation into, a
 * [channel][stride] int16 buffer.
 *   spangpu_awgn_create() / _reinit()   awgn_init_dbm0(NULL, idum, level)      src/awgn.c:127-152 (ran_init :82-105)
 *   spangpu_awgn_tx()                   awgn(s) x samples x N                  src/awgn.c:168-195
 * Exactness: every int16 and the carried half pair equal the reference's.  The generator, the accept / reject sequence
 * and the arithmetic are IEEE binary64 operations rounded as the reference's compiled code rounds them; the one library
 * call of the path, log(), is GNU libc's table-driven routine (not correctly rounded: its bits are part of the
 * reference's output) restated on the device operation by operation, as the x86-64 FMA build of glibc computes it
 * (csrc/glibc_log_dev.hpp).
 */
typedef struct spangpu_awgn_s spangpu_awgn_t;

SPANGPU_API int spangpu_awgn_create(spangpu_awgn_t **bank, int device, int n_channels, const int32_t seeds[],
                                    const float levels_dbm0[]);
SPANGPU_API void spangpu_awgn_destroy(spangpu_awgn_t *bank);
SPANGPU_API int spangpu_awgn_channels(const spangpu_awgn_t *bank);
SPANGPU_API int spangpu_awgn_set_stream(spangpu_awgn_t *bank, void *hip_stream);
SPANGPU_API int spangpu_awgn_sync(spangpu_awgn_t *bank);
SPANGPU_API int spangpu_awgn_reinit(spangpu_awgn_t *bank, int channel, int seed, float level_dbm0);
/* mix = 0: pcm[channel*stride + i] = awgn();  mix = 1: pcm[...] = saturate16(pcm[...] + awgn());  returns samples */
SPANGPU_API int spangpu_awgn_tx(spangpu_awgn_t *bank, int mem, int16_t *pcm, long long stride, int samples, int mix);
SPANGPU_API int spangpu_awgn_state_words(const spangpu_awgn_t *bank);
SPANGPU_API int spangpu_awgn_get_state(spangpu_awgn_t *bank, int channel, int32_t *words);

/* ---- The pipelined host path of echo canceller and modem receiver banks (csrc/feed_api.hip, round 4) -----------------------
   echo_can_update() returns the cleaned sample to its caller (src/echo.c:421-661) and v29_rx() delivers bits through
   put_bit (src/v29rx.c:867-965; negative "bits" are SIG_STATUS_* reports, spandsp/async.h:66-103): for these paths the way
   back over PCIe is part of the product.  A slot has a pinned buffer each way; a tick's H2D copy, kernel and D2H copy run on
   three streams, so tick t + 1's input crosses the link downwards while tick t's kernel runs and tick t - 1's output
   crosses it upwards (full duplex): a tick costs max(H2D, kernel, D2H).
   Echo: per tick  spangpu_echo_feed_acquire(feed, &tx, &rx); write channel c at tx / rx + c*stride samples (int16) or bytes
   (law = SPANGPU_G711_ALAW / _ULAW: decoded on the device, the clean signal encoded on the device -- linear_to_alaw /
   linear_to_ulaw of spandsp/g711.h:124-237 -- half the volume each way);  spangpu_echo_feed_commit(feed, samples);
   n = spangpu_echo_feed_collect(feed, &clean)  (up to depth - 1 ticks later): the clean rows in the same format. */
typedef struct spangpu_xfeed_s spangpu_echo_feed_t;
typedef struct spangpu_xfeed_s spangpu_modem_feed_t;
SPANGPU_API int spangpu_echo_feed_create(spangpu_echo_feed_t **feed, spangpu_echo_t *ec, int max_samples, int law, int depth, int use_hpf_tx);
SPANGPU_API int spangpu_echo_feed_destroy(spangpu_echo_feed_t *feed);
SPANGPU_API long long spangpu_echo_feed_stride(const spangpu_echo_feed_t *feed);
SPANGPU_API int spangpu_echo_feed_acquire(spangpu_echo_feed_t *feed, void **tx, void **rx);
SPANGPU_API int spangpu_echo_feed_commit(spangpu_echo_feed_t *feed, int samples);
SPANGPU_API int spangpu_echo_feed_collect(spangpu_echo_feed_t *feed, const void **clean);
SPANGPU_API int spangpu_echo_feed_outstanding(const spangpu_echo_feed_t *feed);
SPANGPU_API int spangpu_echo_feed_run(spangpu_echo_feed_t *feed, int samples, int ticks, int lag, double *elapsed_ms);
/* The put_bit stream of the last spangpu_modem_rx() in packed form, device to device on the bank's stream (24
   bytes of bits per channel and 160-sample frame for V.29 9600, against one byte per put_bit call in spangpu_modem_events()):
   row c of packed_device[n_ch][words_per_channel]: word 0 = data bits | status reports << 16 of the call (bit 31: the call's
   event buffer overflowed), then the data bits LSB first; status_device[0] = status reports of the whole bank, then pairs
   {channel, data bits delivered before it | (code & 0xFFFF) << 16}, a channel's reports in call order.
   spangpu_modem_packed_words(): row length that holds every bit a call of `samples` samples can deliver at `bit_rate`.
   spangpu_modem_unpack_events() (host code): the packed form back into the calls -- events[n_ch][cap] int8 and counts[n_ch]
   exactly as spangpu_modem_events() delivers them -- for a shim that replays put_bit / status callbacks. */
SPANGPU_API int spangpu_modem_packed_words(int bit_rate, int samples);
/* spangpu_modem_events() by way of the packed form: same answer, a twentieth of the bytes over PCIe (what the shim's groups call) */
SPANGPU_API int spangpu_modem_events_packed(spangpu_modem_t *modem, const int8_t **events, const int32_t **counts);
SPANGPU_API int spangpu_modem_pack_events(spangpu_modem_t *modem, uint32_t *packed_device, int words_per_channel, uint32_t *status_device, int status_cap);
SPANGPU_API int spangpu_modem_unpack_events(const uint32_t *packed, int words_per_channel, const uint32_t *status, int status_cap, int n_ch,
                                            int8_t *events, int cap, int32_t *counts);
SPANGPU_API void *spangpu_modem_get_stream(spangpu_modem_t *modem);
SPANGPU_API int spangpu_modem_bit_rate(const spangpu_modem_t *modem);
SPANGPU_API int spangpu_modem_device(const spangpu_modem_t *modem);
SPANGPU_API void *spangpu_echo_get_stream(spangpu_echo_t *ec);
SPANGPU_API int spangpu_echo_device(const spangpu_echo_t *ec);
/* Modem: per tick  buf = spangpu_modem_feed_acquire(feed); write channel c's PCM at buf + c*stride samples;
   spangpu_modem_feed_commit(feed, samples);  n = spangpu_modem_feed_collect(feed, &packed, &status): the tick's put_bit
   stream packed as above ([n_ch][spangpu_modem_feed_words_per_channel()] rows, status list of
   spangpu_modem_feed_status_cap() entries).  max_bit_rate sizes the rows (a bank's channels may run different rates). */
SPANGPU_API int spangpu_modem_feed_create(spangpu_modem_feed_t **feed, spangpu_modem_t *modem, int max_samples, int max_bit_rate, int depth);
SPANGPU_API int spangpu_modem_feed_destroy(spangpu_modem_feed_t *feed);
SPANGPU_API long long spangpu_modem_feed_stride(const spangpu_modem_feed_t *feed);
SPANGPU_API int spangpu_modem_feed_words_per_channel(const spangpu_modem_feed_t *feed);
SPANGPU_API int spangpu_modem_feed_status_cap(const spangpu_modem_feed_t *feed);
SPANGPU_API void *spangpu_modem_feed_acquire(spangpu_modem_feed_t *feed);
SPANGPU_API int spangpu_modem_feed_commit(spangpu_modem_feed_t *feed, int samples);
SPANGPU_API int spangpu_modem_feed_collect(spangpu_modem_feed_t *feed, const uint32_t **packed, const uint32_t **status);
SPANGPU_API int spangpu_modem_feed_outstanding(const spangpu_modem_feed_t *feed);
SPANGPU_API int spangpu_modem_feed_run(spangpu_modem_feed_t *feed, int samples, int ticks, int lag, double *elapsed_ms, long long *bits);

/* ---- One logical tone bank over several devices (csrc/shard_api.hip, round 4) ---------------------------------
   Channels shard as contiguous ranges, one bank, stream and digit buffer per entry of devices[] (a device may be named
   twice: two shards on one GPU); nothing is exchanged between compute steps; the one exchange is the gather of the digit
   byte of every block and channel to the first shard's device -- written by each shard's detector kernel into a buffer on
   its own device and copied device to device (hipMemcpyPeerAsync, xGMI between peers) behind the kernel on the shard's
   stream.  One host thread drives all shards: every call queues work and returns.
       spangpu_shard_create(&sh, devices, n, SPANGPU_DTMF, n_channels, 160, &params, sizeof(params));
       per tick: spangpu_shard_rx(sh, amp_per_shard, 160, 160);   amp_per_shard[i]: shard i's rows on shard i's device
                 spangpu_shard_digits_device(sh, stream, &digits, &dev, &max_blocks)  or  spangpu_shard_digits_host(sh, out, bytes)
   What it stands for on a host running the reference: the loop over all its dtmf_rx() objects and their dtmf_rx_get()s. */
/* How a shard's results reach the collecting device, and where it sits (spangpu_shard_info() and its echo / modem twins). */
#define SPANGPU_LINK_SAME           0       /* the shard is on the collecting device */
#define SPANGPU_LINK_PEER           1       /* peer access enabled: hipMemcpyPeerAsync goes device to device (xGMI) */
#define SPANGPU_LINK_STAGED         2       /* no peer access (not possible, or enabling it failed): the copy is staged through the host -- correct, slow */
typedef struct
{
    int device;
    int first_channel;
    int n_channels;
    int collect_device;
    int link;                   /* SPANGPU_LINK_* */
    int forced_peer_copy;       /* spangpu_tune_force_peer_copy() is on */
} spangpu_shard_info_t;
/* Debug knob: shards on the collecting device itself send their results with hipMemcpyPeerAsync too (source and destination
   device equal), so that the multi-device code path runs on a one-GPU box.  Returns the previous setting. */
SPANGPU_API int spangpu_tune_force_peer_copy(int on);
typedef struct spangpu_shard_s spangpu_shard_t;
SPANGPU_API int spangpu_shard_create(spangpu_shard_t **shard, const int *devices, int n_devices, int kind, int n_channels, int max_samples,
                                     const void *params, size_t params_size);
SPANGPU_API int spangpu_shard_destroy(spangpu_shard_t *shard);
SPANGPU_API int spangpu_shard_count(const spangpu_shard_t *shard);
SPANGPU_API int spangpu_shard_channels(const spangpu_shard_t *shard);
SPANGPU_API int spangpu_shard_range(const spangpu_shard_t *shard, int i, int *device, int *first_channel, int *n_channels);
SPANGPU_API int spangpu_shard_info(const spangpu_shard_t *shard, int i, spangpu_shard_info_t *info);
SPANGPU_API spangpu_bank_t *spangpu_shard_bank(spangpu_shard_t *shard, int i);
SPANGPU_API int spangpu_shard_rx(spangpu_shard_t *shard, const int16_t *const *amp, int samples, long long stride);
SPANGPU_API int spangpu_shard_digits_device(spangpu_shard_t *shard, void *hip_stream, const uint8_t **digits, int *collect_device, int *max_blocks);
SPANGPU_API int spangpu_shard_digits_host(spangpu_shard_t *shard, uint8_t *out, size_t out_bytes);
SPANGPU_API int spangpu_shard_sync(spangpu_shard_t *shard);
/* (The gathered bytes have two slots used in turn: a step's bytes stay whole until the step after next is queued.)

   BASELINE configs[4]'s own object -- the echo cancellers of N lines over several devices (echo.c:421-661 per line): contiguous
   channel ranges, one update launch per device and step on the shard's stream, and per reporting interval the ERLE of every line
   (10 log10(sum rx^2 / sum clean^2) since the last reset, as tests/echo_tests.c:577-594 measures it) computed on each shard's
   device and gathered to the first shard's device (hipMemcpyPeerAsync: xGMI where the devices are peers).
       spangpu_echo_shard_create(&es, devices, n, n_channels, 128, SPANGPU_ECHO_USE_ADAPTION-style mode bits);
       per step:   spangpu_echo_shard_update(es, tx_per_shard, rx_per_shard, clean_per_shard, 160, 160);   rows on each shard's device
       per second: spangpu_echo_shard_report(es, 1);  spangpu_echo_shard_erle_device(es, stream, &erle, &dev)  or  _erle_host(es, out, n) */
typedef struct spangpu_echo_shard_s spangpu_echo_shard_t;
SPANGPU_API int spangpu_echo_shard_create(spangpu_echo_shard_t **shard, const int *devices, int n_devices, int n_channels, int taps, int adaption_mode);
SPANGPU_API int spangpu_echo_shard_destroy(spangpu_echo_shard_t *shard);
SPANGPU_API int spangpu_echo_shard_count(const spangpu_echo_shard_t *shard);
SPANGPU_API int spangpu_echo_shard_range(const spangpu_echo_shard_t *shard, int i, int *device, int *first_channel, int *n_channels);
SPANGPU_API int spangpu_echo_shard_info(const spangpu_echo_shard_t *shard, int i, spangpu_shard_info_t *info);
SPANGPU_API spangpu_echo_t *spangpu_echo_shard_bank(spangpu_echo_shard_t *shard, int i);
SPANGPU_API int spangpu_echo_shard_update(spangpu_echo_shard_t *shard, const int16_t *const *tx, const int16_t *const *rx, int16_t *const *clean,
                                          int samples, long long stride);
SPANGPU_API int spangpu_echo_shard_report(spangpu_echo_shard_t *shard, int reset);
SPANGPU_API int spangpu_echo_shard_erle_device(spangpu_echo_shard_t *shard, void *hip_stream, const float **erle_db, int *collect_device);
SPANGPU_API int spangpu_echo_shard_erle_host(spangpu_echo_shard_t *shard, float *out, size_t out_floats);
SPANGPU_API int spangpu_echo_shard_sync(spangpu_echo_shard_t *shard);
/* Modem receivers (SPANGPU_V29 / _V27TER / _V17) over several devices: per step the put_bit streams of every channel -- what
   spangpu_modem_copy_events() lays out, int32 counts[n_i] then int8 events[n_i][events_per_channel] per shard -- gathered to
   the first shard's device; spangpu_modem_shard_events_host() hands them out in the whole bank's channel order. */
typedef struct spangpu_modem_shard_s spangpu_modem_shard_t;
SPANGPU_API int spangpu_modem_shard_create(spangpu_modem_shard_t **shard, const int *devices, int n_devices, int kind, int n_channels, int bit_rate,
                                           int events_per_channel);
SPANGPU_API int spangpu_modem_shard_destroy(spangpu_modem_shard_t *shard);
SPANGPU_API int spangpu_modem_shard_range(const spangpu_modem_shard_t *shard, int i, int *device, int *first_channel, int *n_channels);
SPANGPU_API int spangpu_modem_shard_info(const spangpu_modem_shard_t *shard, int i, spangpu_shard_info_t *info);
SPANGPU_API spangpu_modem_t *spangpu_modem_shard_bank(spangpu_modem_shard_t *shard, int i);
SPANGPU_API int spangpu_modem_shard_rx(spangpu_modem_shard_t *shard, const int16_t *const *amp, int samples, long long stride);
SPANGPU_API int spangpu_modem_shard_events_host(spangpu_modem_shard_t *shard, int32_t *counts, int8_t *events);
SPANGPU_API int spangpu_modem_shard_sync(spangpu_modem_shard_t *shard);

/* ---- The receivers' inner primitives as batched entry points of their own (csrc/prim_api.hip) ----
   N independent items per launch, one lane each, in the reference's scalar order of operations (every product and sum rounded
   by itself): bit-exact with the reference's strict build on any input.  Rows are item-major; a stride of 0 shares one row
   among all items; complex values are {re, im} float pairs and their strides count complex elements; mem says where ALL the
   arrays of a call live (SPANGPU_MEM_HOST: copied in and out; SPANGPU_MEM_DEVICE: used in place, the call waits for the kernel).
     spangpu_vec_circular_dot_prodf_batch    vec_circular_dot_prodf(x, y, n, pos)      src/vector_float.c:890-900,932-939
     spangpu_vec_circular_lmsf_batch         vec_circular_lmsf(x, y, n, pos, error)     src/vector_float.c:942,982-1000
     spangpu_cvec_circular_dot_prodf_batch   cvec_circular_dot_prodf(x, y, n, pos)     src/complex_vector_float.c:137-150,187-196
     spangpu_cvec_circular_lmsf_batch        cvec_circular_lmsf(x, y, n, pos, &error)   src/complex_vector_float.c:201-219
     spangpu_power_meter_update_batch        power_meter_update() over a row of samples  src/power_meter.c:65-70
     spangpu_godard_ted_rx_batch             godard_ted_rx() over a row of samples       src/godard.c:144-162
     spangpu_godard_ted_per_baud_batch       godard_ted_per_baud()                       src/godard.c:165-220
   A Godard timing error detector's item is eight state words {low_band_edge[2], high_band_edge[2], dc_filter[2], baud_phase (floats),
   total_baud_timing_correction (int32)} and twelve descriptor words {low_band_edge_coeff[3], high_band_edge_coeff[3],
   mixed_band_edges_coeff_3, coarse_trigger, fine_trigger (floats), coarse_step, fine_step (int32), 0}: the reference's structs
   (src/spandsp/godard.h:57-77, private/godard.h:29-54) word for word.  A descriptor stride of 0 shares one among all items. */
SPANGPU_API int spangpu_vec_circular_dot_prodf_batch(int device, const float *x, long long x_stride, const float *y, long long y_stride,
                                                     const int32_t *pos, float *z, int items, int n, int mem);
SPANGPU_API int spangpu_vec_circular_lmsf_batch(int device, const float *x, long long x_stride, float *y, long long y_stride,
                                                const int32_t *pos, const float *error, int items, int n, int mem);
SPANGPU_API int spangpu_cvec_circular_dot_prodf_batch(int device, const float *x, long long x_stride, const float *y, long long y_stride,
                                                      const int32_t *pos, float *z, int items, int n, int mem);
SPANGPU_API int spangpu_cvec_circular_lmsf_batch(int device, const float *x, long long x_stride, float *y, long long y_stride,
                                                 const int32_t *pos, const float *error, int items, int n, int mem);
SPANGPU_API int spangpu_power_meter_update_batch(int device, const int16_t *amp, long long stride, int32_t *reading, const int32_t *shift,
                                                 int items, int n, int mem);
SPANGPU_API int spangpu_godard_ted_rx_batch(int device, uint32_t *state, const uint32_t *desc, long long desc_stride, const float *samples,
                                            long long stride, int items, int n, int mem);
SPANGPU_API int spangpu_godard_ted_per_baud_batch(int device, uint32_t *state, const uint32_t *desc, long long desc_stride, int32_t *correction,
                                                  int items, int mem);
/* A position outside a row (pos < 0 or pos > n) is refused with SPANGPU_ERR_BAD_ARG when the arrays are host memory; in
   device memory nobody reads them beforehand: such an item's result is NaN (dot products) and its y row is left alone (LMS). */

/* ---- ... and the rest of them (csrc/prim2_api.hip; SURVEY 8(a) a19 and the periodograms of section 2's Goertzel core) ----
     spangpu_periodogram_batch               periodogram(coeffs, amp, len)                        src/tone_detect.c:208-225
     spangpu_periodogram_prepare_batch       periodogram_prepare(sum, diff, amp, len)             src/tone_detect.c:228-239
     spangpu_periodogram_apply_batch         periodogram_apply(coeffs, sum, diff, len)            src/tone_detect.c:242-255
     spangpu_periodogram_freq_error_batch    periodogram_freq_error(&offset, scale, &last, &now)  src/tone_detect.c:299-310
     spangpu_fixed_sqrt32_batch              fixed_sqrt32(x)                                      src/math_fixed.c:158-169
     spangpu_dds_complexf_batch              dds_complexf(&phase_acc, phase_rate) n times (n = 1, rate 0: dds_lookup_complexf(phase);
                                             the accumulator moves as dds_advancef() moves it)   src/dds_float.c:2135-2187
     spangpu_arctan2_batch                   arctan2(y, x)                                        src/spandsp/arctan2.h:47-80
   Same conventions as above.  periodogram rows: coeffs [len/2] complex (periodogram_generate_coeffs(), host table making, by name
   in libspangpu_prims), amp [len] complex; prepare writes sum and diff as [items][len/2] complex and returns len/2.  The square
   root and the phasor use the tables the receiver kernels use (csrc/modem_tables.c), arctan2 the receivers' device function. */
SPANGPU_API int spangpu_periodogram_batch(int device, const float *coeffs, long long c_stride, const float *amp, long long a_stride, float *out,
                                          int items, int len, int mem);
SPANGPU_API int spangpu_periodogram_prepare_batch(int device, const float *amp, long long a_stride, float *sum, float *diff, int items, int len, int mem);
SPANGPU_API int spangpu_periodogram_apply_batch(int device, const float *coeffs, long long c_stride, const float *sum, const float *diff, float *out,
                                                int items, int len, int mem);
SPANGPU_API int spangpu_periodogram_freq_error_batch(int device, const float *phase_offset, float scale, const float *last_result,
                                                     const float *result, float *out, int items, int mem);
SPANGPU_API int spangpu_fixed_sqrt32_batch(int device, const uint32_t *x, uint16_t *out, int items, int mem);
SPANGPU_API int spangpu_dds_complexf_batch(int device, uint32_t *phase_acc, const int32_t *phase_rate, float *out, int items, int n, int mem);
SPANGPU_API int spangpu_arctan2_batch(int device, const float *y, const float *x, int32_t *out, int items, int mem);

#if defined(__cplusplus)
}
#endif

#endif
