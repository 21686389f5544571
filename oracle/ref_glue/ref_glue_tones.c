/*
 * ref_glue_tones.c -- TEST INFRASTRUCTURE ONLY.  OUR drivers over the two Goertzel users of the reference build outside
 * tone_detect.c: the caller-side tone scan of src/v18.c (:1528-1667) and the handshake tone detector of
 * the Ademco Contact ID sender (src/ademco_contactid.c:871-1115).  Both decide once per Goertzel block and keep the raw
 * block decision in their state (v18: in_tone, assigned whenever it differs; ademco: last_hit), which is what these
 * drivers hand out, block by block.  Compiled only into oracle/_ref/libspandsp_ref.so; #includes reference headers from
 * /root/reference/src at build time.
 */
#include <stdlib.h>
#include <inttypes.h>
#include <string.h>
#include <stdio.h>
#include <stdbool.h>

#include "spandsp/telephony.h"
#include "spandsp/logging.h"
#include "spandsp/queue.h"
#include "spandsp/async.h"
#include "spandsp/complex.h"
#include "spandsp/dds.h"
#include "spandsp/power_meter.h"
#include "spandsp/tone_detect.h"
#include "spandsp/tone_generate.h"
#include "spandsp/super_tone_rx.h"
#include "spandsp/dtmf.h"
#include "spandsp/fsk.h"
#include "spandsp/modem_connect_tones.h"
#include "spandsp/v18.h"
#include "spandsp/ademco_contactid.h"
#include "spandsp/private/logging.h"
#include "spandsp/private/queue.h"
#include "spandsp/private/power_meter.h"
#include "spandsp/private/tone_generate.h"
#include "spandsp/private/async.h"
#include "spandsp/private/fsk.h"
#include "spandsp/private/dtmf.h"
#include "spandsp/private/modem_connect_tones.h"
#include "spandsp/private/v18.h"
#include "spandsp/private/ademco_contactid.h"

#define GLUE __attribute__((visibility("default")))

static void v18_status(void *user_data, int status)
{
    (void) user_data;
    (void) status;
}

static void v18_msg(void *user_data, const uint8_t *msg, int len)
{
    (void) user_data;
    (void) msg;
    (void) len;
}

/* n_blocks blocks of 102 samples through v18_rx() with the receiver held in its first originating state (where it runs
   caller_tone_scan()); tone[b] = in_tone after block b.  Returns the level threshold the object carries. */
GLUE float glue_v18_tone_blocks(const int16_t amp[], int n_blocks, int32_t tone[])
{
    v18_state_t *s;
    float threshold;
    int b;

    if ((s = v18_init(NULL, true, V18_MODE_WEITBRECHT_5BIT_4545, V18_AUTOMODING_GLOBAL, v18_msg, NULL, v18_status, NULL)) == NULL)
        return -1.0f;
    threshold = s->threshold;
    for (b = 0;  b < n_blocks;  b++)
    {
        s->rx_state = V18_RX_STATE_ORIGINATING_1;      /* a confirmed tone moves the receiver on */
        s->rx_suppression_timer = 0;
        v18_rx(s, amp + 102*b, 102);
        tone[b] = s->in_tone;
    }
    v18_free(s);
    return threshold;
}

static void ademco_report(void *user_data, int tone, int level, int duration)
{
    (void) user_data;
    (void) tone;
    (void) level;
    (void) duration;
}

/* n_blocks blocks of 55 samples through ademco_contactid_sender_rx(); hit[b] = last_hit after block b */
GLUE int glue_ademco_tone_blocks(const int16_t amp[], int n_blocks, int32_t hit[])
{
    ademco_contactid_sender_state_t *s;
    int b;

    if ((s = ademco_contactid_sender_init(NULL, ademco_report, NULL)) == NULL)
        return -1;
    for (b = 0;  b < n_blocks;  b++)
    {
        ademco_contactid_sender_rx(s, amp + 55*b, 55);
        hit[b] = s->last_hit;
    }
    ademco_contactid_sender_free(s);
    return 0;
}
