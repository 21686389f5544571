/*
 * fsk_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's non-coherent FSK receiver, the
 * detector that runs beside the fast modems in a FAX front end (V.21 channel 2, synchronous):
 *
 *   fsk_rx_init / fsk_rx_restart       src/fsk.c:660-742
 *   fsk_rx_set_signal_cutoff           src/fsk.c:270-276   (power_meter_level_dbm0, power_meter.c:82-92)
 *   fsk_rx_set_frame_parameters        src/fsk.c:300-316
 *   fsk_rx                             src/fsk.c:393-622
 *   put_frame                          src/fsk.c:352-391   (parity8, bit_operations.h:284-288)
 *   fsk_rx_fillin                      src/fsk.c:625-657
 *   dds_phase_rate / dds_lookup        src/dds_int.c (quarter wave table of 257 int16 = round(32767 sin(i pi/512)))
 *   power_meter_update                 src/power_meter.c:65-69
 *   preset_fsk_specs[]                 src/fsk.c:60-155
 *
 * Everything here is integer arithmetic.  put_bit() calls, status reports and framed characters go to the
 * sink as kind 3 events in the order the reference makes them (no status handler installed: status
 * changes go through put_bit, fsk.c:343-349).
 */
#include <math.h>
#include <string.h>

#include "oracle.h"
#include "modem_common.h"

#define RATE_X100   (8000*100)

static int16_t quarter_sine[257];
static int quarter_sine_ready = 0;

static void make_quarter_sine(void)
{
    for (int i = 0;  i <= 256;  i++)
        quarter_sine[i] = (int16_t) lrint(32767.0*sin(i*3.14159265358979323846/512.0));
    quarter_sine_ready = 1;
}

static int32_t lookup(uint32_t phase)
{
    /* dds_lookup(), dds_int.c */
    uint32_t step;
    int32_t amp;

    phase >>= 22;
    step = phase & 255;
    if (phase & 256)
        step = 256 - step;
    amp = quarter_sine[step];
    return (phase & 512)  ?  -amp  :  amp;
}

static const int32_t presets[11][5] =
{
    {1180, 980, -14, -30, 30000},       /* V21 ch 1 */
    {1850, 1650, -14, -30, 30000},      /* V21 ch 2 */
    {2100, 1300, -14, -30, 120000},     /* V23 ch 1 */
    {450, 390, -14, -30, 7500},         /* V23 ch 2 */
    {1070, 1270, -14, -30, 30000},      /* Bell103 ch 1 */
    {2025, 2225, -14, -30, 30000},      /* Bell103 ch 2 */
    {2200, 1200, -14, -30, 120000},     /* Bell202 */
    {1800, 1400, -14, -30, 4545},       /* Weitbrecht 45.45 */
    {1800, 1400, -14, -30, 5000},       /* Weitbrecht 50 */
    {1800, 1400, -14, -30, 4760},       /* Weitbrecht 47.6 */
    {1180, 980, -14, -30, 11000}        /* V21 (110bps) ch 1 */
};

ORC_API int orc_fsk_preset(int which, int32_t out[5])
{
    if (which < 0  ||  which > 10)
        return -1;
    memcpy(out, presets[which], sizeof(presets[0]));
    return 0;
}

ORC_API void orc_fsk_set_signal_cutoff(orc_fsk_t *s, float cutoff)
{
    s->carrier_on_power = level_dbm0(cutoff + 2.5f - 5.3f);
    s->carrier_off_power = level_dbm0(cutoff - 2.5f - 5.3f);
}

ORC_API void orc_fsk_set_frame_parameters(orc_fsk_t *s, int data_bits, int parity, int stop_bits)
{
    if (s->framing_mode != 2)
        return;
    s->data_bits = data_bits;
    s->parity = parity;
    s->stop_bits = stop_bits;
    s->total_data_bits = data_bits + (parity != 0  ?  1  :  0);
}

ORC_API int orc_fsk_restart(orc_fsk_t *s, const int32_t spec[5], int framing_mode)
{
    int chop;

    if (!quarter_sine_ready)
        make_quarter_sine();
    s->baud_rate = spec[4];
    s->framing_mode = framing_mode;
    if (framing_mode == 2)
        orc_fsk_set_frame_parameters(s, 8, 0, 1);
    orc_fsk_set_signal_cutoff(s, (float) spec[3]);
    s->phase_rate[0] = (int32_t) ((float) spec[0]*65536.0f*65536.0f/8000);
    s->phase_rate[1] = (int32_t) ((float) spec[1]*65536.0f*65536.0f/8000);
    s->phase_acc[0] = 0;
    s->phase_acc[1] = 0;
    s->last_sample = 0;
    s->correlation_span = RATE_X100/spec[4];
    if (s->correlation_span > ORC_FSK_MAX_WINDOW)
        s->correlation_span = ORC_FSK_MAX_WINDOW;
    s->scaling_shift = 0;
    for (chop = s->correlation_span;  chop != 0;  chop >>= 1)
        s->scaling_shift++;
    s->baud_phase = 0;
    s->frame_pos = -2;
    s->frame_in_progress = 0;
    s->last_bit = 0;
    s->power_reading = 0;           /* power_meter_init(&s->power, 4) */
    s->signal_present = 0;
    return 0;
}

ORC_API int orc_fsk_init(orc_fsk_t *s, const int32_t spec[5], int framing_mode)
{
    memset(s, 0, sizeof(*s));
    return orc_fsk_restart(s, spec, framing_mode);
}

static void deliver_frame(orc_fsk_t *s, uint32_t frame, orc_put_bit_t put, void *user)
{
    /* put_frame(), fsk.c:352-391; frame is the 16 bit shift register */
    if (s->parity != 0)
    {
        const uint32_t sent = (frame >> 15) & 1;
        uint32_t want;
        uint32_t x;

        frame = (frame & 0x7FFF) >> (16 - s->total_data_bits);
        x = frame & 0xFF;                   /* parity8() takes a uint8_t */
        x = (x ^ (x >> 4)) & 0x0F;
        x = (0x6996 >> x) & 1;
        switch (s->parity)
        {
        case 2: want = x ^ 1; break;        /* ASYNC_PARITY_ODD */
        case 1: want = x; break;            /* ASYNC_PARITY_EVEN */
        case 3: want = 1; break;            /* ASYNC_PARITY_MARK */
        default: want = 0; break;           /* ASYNC_PARITY_SPACE */
        }
        if (sent == want)
            put(user, (int) frame);
        else
            s->parity_errors++;
    }
    else
    {
        put(user, (int) (frame >> (16 - s->total_data_bits)));
    }
}

ORC_API int orc_fsk_rx_cb(orc_fsk_t *s, const int16_t amp[], int len, orc_put_bit_t put, void *user)
{
    int ptr = s->buf_ptr;

    for (int i = 0;  i < len;  i++)
    {
        int32_t sum[2];
        int32_t power;
        int16_t x;
        int state;

        for (int j = 0;  j < 2;  j++)
        {
            int32_t *slot = s->window[ptr][j];
            const int32_t c = lookup(s->phase_acc[j] + (1u << 30));
            const int32_t q = lookup(s->phase_acc[j]);
            int32_t d;

            s->phase_acc[j] += (uint32_t) s->phase_rate[j];
            s->dot[j][0] -= slot[0];
            s->dot[j][1] -= slot[1];
            slot[0] = (c*amp[i]) >> s->scaling_shift;
            slot[1] = (q*amp[i]) >> s->scaling_shift;
            s->dot[j][0] += slot[0];
            s->dot[j][1] += slot[1];
            d = s->dot[j][0] >> 15;
            sum[j] = d*d;
            d = s->dot[j][1] >> 15;
            sum[j] += d*d;
        }
        /* fsk.c:425-431: power behind a one-tap DC blocker */
        x = amp[i] >> 1;
        {
            const int16_t diff = (int16_t) (x - s->last_sample);
            s->power_reading += ((diff*diff - s->power_reading) >> 4);
        }
        power = s->power_reading;
        s->last_sample = x;
        if (s->signal_present)
        {
            if (power < s->carrier_off_power)
            {
                if (--s->signal_present <= 0)
                {
                    put(user, -1);                          /* SIG_STATUS_CARRIER_DOWN */
                    s->baud_phase = 0;
                    continue;                               /* note: the window slot is not advanced */
                }
            }
        }
        else
        {
            if (power < s->carrier_on_power)
            {
                s->baud_phase = 0;
                continue;
            }
            if (s->baud_phase < (s->correlation_span >> 1) - 30)
            {
                s->baud_phase++;
                continue;
            }
            s->signal_present = 1;
            s->baud_phase = 0;
            s->frame_pos = -2;
            s->frame_in_progress = 0;
            s->last_bit = 0;
            put(user, -2);                                  /* SIG_STATUS_CARRIER_UP */
        }
        state = (sum[0] < sum[1]);
        if (s->framing_mode == 1)
        {
            /* synchronous, fsk.c:489-512 */
            if (s->last_bit != state)
            {
                s->last_bit = state;
                if (s->baud_phase < RATE_X100/2)
                    s->baud_phase += (s->baud_rate >> 3);
                else
                    s->baud_phase -= (s->baud_rate >> 3);
            }
            if ((s->baud_phase += s->baud_rate) >= RATE_X100)
            {
                s->baud_phase -= RATE_X100;
                put(user, state);
            }
        }
        else if (s->framing_mode == 0)
        {
            /* asynchronous, fsk.c:513-537 */
            if (s->last_bit != state)
            {
                s->last_bit = state;
                s->baud_phase = RATE_X100/2;
            }
            if ((s->baud_phase += s->baud_rate) >= RATE_X100)
            {
                s->baud_phase -= RATE_X100;
                put(user, state);
            }
        }
        else if (s->frame_pos == -2)
        {
            /* framed, fsk.c:538-614: hunting for a start bit */
            if (state == 0)
            {
                s->baud_phase = 8000*(100 - 40)/2;
                s->frame_pos = -1;
                s->frame_in_progress = 0;
                s->last_bit = -1;
            }
        }
        else if (s->frame_pos == -1)
        {
            if (state != 0)
            {
                s->frame_pos = -2;
            }
            else if ((s->baud_phase += s->baud_rate) >= RATE_X100)
            {
                s->frame_pos = 0;
                s->last_bit = state;
            }
        }
        else if ((s->baud_phase += s->baud_rate) >= 8000*(100 - 40))
        {
            if (s->last_bit < 0)
                s->last_bit = state;
            if (s->last_bit != state)
            {
                s->frame_pos = -2;
                s->framing_errors++;
            }
            else if (s->baud_phase >= RATE_X100)
            {
                if (s->frame_pos++ > s->total_data_bits)
                {
                    if (state == 1)
                        deliver_frame(s, (uint32_t) s->frame_in_progress, put, user);
                    else
                        s->framing_errors++;
                    s->frame_pos = -2;
                }
                else
                {
                    s->frame_in_progress = ((s->frame_in_progress >> 1) | (state << 15)) & 0xFFFF;
                }
                s->baud_phase -= RATE_X100;
                s->last_bit = -1;
            }
        }
        if (++ptr >= s->correlation_span)
            ptr = 0;
    }
    s->buf_ptr = ptr;
    return 0;
}

static void sink_put_bit(void *user, int bit)
{
    orc_sink_push((orc_sink_t *) user, 3, bit, 0, 0);
}

ORC_API int orc_fsk_rx(orc_fsk_t *s, const int16_t amp[], int len, orc_sink_t *sink)
{
    return orc_fsk_rx_cb(s, amp, len, sink_put_bit, sink);
}

ORC_API int orc_fsk_fillin(orc_fsk_t *s, int len)
{
    /* fsk.c:625-657 -- note that buf_ptr does not move */
    const int ptr = s->buf_ptr;

    for (int i = 0;  i < len;  i++)
    {
        for (int j = 0;  j < 2;  j++)
        {
            s->dot[j][0] -= s->window[ptr][j][0];
            s->dot[j][1] -= s->window[ptr][j][1];
            s->phase_acc[j] += (uint32_t) s->phase_rate[j];
            s->window[ptr][j][0] = 0;
            s->window[ptr][j][1] = 0;
        }
    }
    return 0;
}

ORC_API int orc_fsk_sizeof(void)
{
    return (int) sizeof(orc_fsk_t);
}
