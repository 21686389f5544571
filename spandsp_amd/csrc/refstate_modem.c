/*
 * refstate_modem.c -- a modem, FSK, connect tone or signalling tone receiver's state in the reference's own struct layout (include/spangpu_refstate.h) in and
 * out of a bank channel.  Host code over the bank's word-level state access: the 238 float and 43 integer words of a
 * channel are, in this order, the fields listed below (the order of the bank's get_state / set_state, which the parity
 * tests compare word for word with the reference's struct).
 */
#include <string.h>

#include "spangpu.h"
#include "spangpu_refstate.h"

#define NF  238
#define NI  43

static uint32_t fbits(float v)
{
    uint32_t u;

    memcpy(&u, &v, 4);
    return u;
}

static float bitsf(uint32_t u)
{
    float v;

    memcpy(&v, &u, 4);
    return v;
}

int spangpu_v29_import_state(spangpu_modem_t *bank, int channel, const spangpu_ref_v29_rx_t *s)
{
    uint32_t w[NF + NI + 512];
    uint32_t *f = w;
    uint32_t *iw = w + NF;
    int n = 0;
    int i;

    if (bank == NULL  ||  s == NULL)
        return SPANGPU_ERR_BAD_ARG;
    /* what is there says how many words this bank keeps per channel: only a V.29 bank keeps 281 */
    if (spangpu_modem_get_state(bank, channel, w) != NF + NI)
        return SPANGPU_ERR_BAD_ARG;
    f[n++] = fbits(s->agc_scaling);
    f[n++] = fbits(s->agc_scaling_save);
    f[n++] = fbits(s->eq_delta);
    f[n++] = fbits(s->training_error);
    f[n++] = fbits(s->carrier_track_p);
    f[n++] = fbits(s->carrier_track_i);
    f[n++] = fbits(s->godard.low_band_edge[0]);
    f[n++] = fbits(s->godard.low_band_edge[1]);
    f[n++] = fbits(s->godard.high_band_edge[0]);
    f[n++] = fbits(s->godard.high_band_edge[1]);
    f[n++] = fbits(s->godard.dc_filter[0]);
    f[n++] = fbits(s->godard.dc_filter[1]);
    f[n++] = fbits(s->godard.baud_phase);
    for (i = 0;  i < 27;  i++)
        f[n++] = fbits(s->rrc_filter[i]);
    for (i = 0;  i < 33;  i++)
    {
        f[n++] = fbits(s->eq_coeff[i].re);
        f[n++] = fbits(s->eq_coeff[i].im);
    }
    for (i = 0;  i < 33;  i++)
    {
        f[n++] = fbits(s->eq_coeff_save[i].re);
        f[n++] = fbits(s->eq_coeff_save[i].im);
    }
    for (i = 0;  i < 33;  i++)
    {
        f[n++] = fbits(s->eq_buf[i].re);
        f[n++] = fbits(s->eq_buf[i].im);
    }
    n = 0;
    iw[n++] = (uint32_t) s->bit_rate;
    iw[n++] = (uint32_t) s->rrc_filter_step;
    iw[n++] = s->scramble_reg;
    iw[n++] = s->training_scramble_reg;
    iw[n++] = (uint32_t) s->training_cd;
    iw[n++] = s->old_train  ?  1  :  0;
    iw[n++] = (uint32_t) s->training_stage;
    iw[n++] = (uint32_t) s->training_count;
    iw[n++] = (uint32_t) (int32_t) s->last_sample;
    iw[n++] = (uint32_t) s->signal_present;
    iw[n++] = s->carrier_phase;
    iw[n++] = (uint32_t) s->carrier_phase_rate;
    iw[n++] = (uint32_t) s->carrier_phase_rate_save;
    iw[n++] = (uint32_t) s->power.reading;
    iw[n++] = (uint32_t) s->carrier_on_power;
    iw[n++] = (uint32_t) s->carrier_off_power;
    iw[n++] = (uint32_t) s->eq_step;
    iw[n++] = (uint32_t) s->eq_put_step;
    iw[n++] = (uint32_t) s->eq_skip;
    iw[n++] = (uint32_t) s->baud_half;
    iw[n++] = (uint32_t) s->last_angles[0];
    iw[n++] = (uint32_t) s->last_angles[1];
    for (i = 0;  i < 16;  i++)
        iw[n++] = (uint32_t) s->diff_angles[i];
    iw[n++] = (uint32_t) s->constellation_state;
    iw[n++] = (uint32_t) s->godard.total_baud_timing_correction;
    iw[n++] = (uint32_t) (int32_t) s->high_sample;
    iw[n++] = (uint32_t) s->low_samples;
    iw[n++] = (uint32_t) s->carrier_drop_pending;
    return spangpu_modem_set_state(bank, channel, w);
}

int spangpu_v29_export_state(spangpu_modem_t *bank, int channel, spangpu_ref_v29_rx_t *s)
{
    uint32_t w[NF + NI + 512];
    const uint32_t *f = w;
    const uint32_t *iw = w + NF;
    int n = 0;
    int i;

    if (bank == NULL  ||  s == NULL)
        return SPANGPU_ERR_BAD_ARG;
    if (spangpu_modem_get_state(bank, channel, w) != NF + NI)
        return SPANGPU_ERR_BAD_ARG;
    s->agc_scaling = bitsf(f[n++]);
    s->agc_scaling_save = bitsf(f[n++]);
    s->eq_delta = bitsf(f[n++]);
    s->training_error = bitsf(f[n++]);
    s->carrier_track_p = bitsf(f[n++]);
    s->carrier_track_i = bitsf(f[n++]);
    s->godard.low_band_edge[0] = bitsf(f[n++]);
    s->godard.low_band_edge[1] = bitsf(f[n++]);
    s->godard.high_band_edge[0] = bitsf(f[n++]);
    s->godard.high_band_edge[1] = bitsf(f[n++]);
    s->godard.dc_filter[0] = bitsf(f[n++]);
    s->godard.dc_filter[1] = bitsf(f[n++]);
    s->godard.baud_phase = bitsf(f[n++]);
    for (i = 0;  i < 27;  i++)
        s->rrc_filter[i] = bitsf(f[n++]);
    for (i = 0;  i < 33;  i++)
    {
        s->eq_coeff[i].re = bitsf(f[n++]);
        s->eq_coeff[i].im = bitsf(f[n++]);
    }
    for (i = 0;  i < 33;  i++)
    {
        s->eq_coeff_save[i].re = bitsf(f[n++]);
        s->eq_coeff_save[i].im = bitsf(f[n++]);
    }
    for (i = 0;  i < 33;  i++)
    {
        s->eq_buf[i].re = bitsf(f[n++]);
        s->eq_buf[i].im = bitsf(f[n++]);
    }
    n = 0;
    s->bit_rate = (int) iw[n++];
    s->rrc_filter_step = (int) iw[n++];
    s->scramble_reg = iw[n++];
    s->training_scramble_reg = (uint8_t) iw[n++];
    s->training_cd = (int) iw[n++];
    s->old_train = (iw[n++] != 0);
    s->training_stage = (int) iw[n++];
    s->training_count = (int) iw[n++];
    s->last_sample = (int16_t) iw[n++];
    s->signal_present = (int) iw[n++];
    s->carrier_phase = iw[n++];
    s->carrier_phase_rate = (int32_t) iw[n++];
    s->carrier_phase_rate_save = (int32_t) iw[n++];
    s->power.reading = (int32_t) iw[n++];
    s->carrier_on_power = (int32_t) iw[n++];
    s->carrier_off_power = (int32_t) iw[n++];
    s->eq_step = (int) iw[n++];
    s->eq_put_step = (int) iw[n++];
    s->eq_skip = (int) iw[n++];
    s->baud_half = (int) iw[n++];
    s->last_angles[0] = (int32_t) iw[n++];
    s->last_angles[1] = (int32_t) iw[n++];
    for (i = 0;  i < 16;  i++)
        s->diff_angles[i] = (int32_t) iw[n++];
    s->constellation_state = (int) iw[n++];
    s->godard.total_baud_timing_correction = (int) iw[n++];
    s->high_sample = (int16_t) iw[n++];
    s->low_samples = (int) iw[n++];
    s->carrier_drop_pending = (int) iw[n++];
    return SPANGPU_OK;
}

/* ---- V.27ter: 225 float words and 45 integer words ---------------------------------------------------------- */
#define NF27    225
#define NI27    45

int spangpu_v27ter_import_state(spangpu_modem_t *bank, int channel, const spangpu_ref_v27ter_rx_t *s)
{
    uint32_t w[NF27 + NI27 + 512];
    uint32_t *f = w;
    uint32_t *iw = w + NF27;
    int n = 0;
    int i;

    if (bank == NULL  ||  s == NULL)
        return SPANGPU_ERR_BAD_ARG;
    if (spangpu_modem_get_state(bank, channel, w) != NF27 + NI27)
        return SPANGPU_ERR_BAD_ARG;
    if ((int) iw[0] != s->bit_rate)
        return SPANGPU_ERR_BAD_ARG;             /* the bank's tables are those of its own rate */
    f[n++] = fbits(s->agc_scaling);
    f[n++] = fbits(s->agc_scaling_save);
    f[n++] = fbits(s->eq_delta);
    f[n++] = fbits(s->training_error);
    f[n++] = fbits(s->carrier_track_p);
    f[n++] = fbits(s->carrier_track_i);
    for (i = 0;  i < 27;  i++)
        f[n++] = fbits(s->rrc_filter[i]);
    for (i = 0;  i < 32;  i++)
    {
        f[n++] = fbits(s->eq_coeff[i].re);
        f[n++] = fbits(s->eq_coeff[i].im);
    }
    for (i = 0;  i < 32;  i++)
    {
        f[n++] = fbits(s->eq_coeff_save[i].re);
        f[n++] = fbits(s->eq_coeff_save[i].im);
    }
    for (i = 0;  i < 32;  i++)
    {
        f[n++] = fbits(s->eq_buf[i].re);
        f[n++] = fbits(s->eq_buf[i].im);
    }
    n = 0;
    iw[n++] = (uint32_t) s->bit_rate;
    iw[n++] = (uint32_t) s->rrc_filter_step;
    iw[n++] = s->scramble_reg;
    iw[n++] = (uint32_t) s->scrambler_pattern_count;
    iw[n++] = (uint32_t) s->training_bc;
    iw[n++] = s->old_train  ?  1  :  0;
    iw[n++] = (uint32_t) s->training_stage;
    iw[n++] = (uint32_t) s->training_count;
    iw[n++] = (uint32_t) (int32_t) s->last_sample;
    iw[n++] = (uint32_t) s->signal_present;
    iw[n++] = (uint32_t) s->carrier_drop_pending;
    iw[n++] = (uint32_t) s->low_samples;
    iw[n++] = (uint32_t) (int32_t) s->high_sample;
    iw[n++] = (uint32_t) s->constellation_state;
    iw[n++] = s->carrier_phase;
    iw[n++] = (uint32_t) s->carrier_phase_rate;
    iw[n++] = (uint32_t) s->carrier_phase_rate_save;
    iw[n++] = (uint32_t) s->power.reading;
    iw[n++] = (uint32_t) s->carrier_on_power;
    iw[n++] = (uint32_t) s->carrier_off_power;
    iw[n++] = (uint32_t) s->eq_step;
    iw[n++] = (uint32_t) s->eq_put_step;
    iw[n++] = (uint32_t) s->eq_skip;
    iw[n++] = (uint32_t) s->baud_half;
    iw[n++] = (uint32_t) s->gardner_integrate;
    iw[n++] = (uint32_t) s->gardner_step;
    iw[n++] = (uint32_t) s->total_baud_timing_correction;
    iw[n++] = (uint32_t) s->last_angles[0];
    iw[n++] = (uint32_t) s->last_angles[1];
    for (i = 0;  i < 16;  i++)
        iw[n++] = (uint32_t) s->diff_angles[i];
    return spangpu_modem_set_state(bank, channel, w);
}

int spangpu_v27ter_export_state(spangpu_modem_t *bank, int channel, spangpu_ref_v27ter_rx_t *s)
{
    uint32_t w[NF27 + NI27 + 512];
    const uint32_t *f = w;
    const uint32_t *iw = w + NF27;
    int n = 0;
    int i;

    if (bank == NULL  ||  s == NULL)
        return SPANGPU_ERR_BAD_ARG;
    if (spangpu_modem_get_state(bank, channel, w) != NF27 + NI27)
        return SPANGPU_ERR_BAD_ARG;
    s->agc_scaling = bitsf(f[n++]);
    s->agc_scaling_save = bitsf(f[n++]);
    s->eq_delta = bitsf(f[n++]);
    s->training_error = bitsf(f[n++]);
    s->carrier_track_p = bitsf(f[n++]);
    s->carrier_track_i = bitsf(f[n++]);
    for (i = 0;  i < 27;  i++)
        s->rrc_filter[i] = bitsf(f[n++]);
    for (i = 0;  i < 32;  i++)
    {
        s->eq_coeff[i].re = bitsf(f[n++]);
        s->eq_coeff[i].im = bitsf(f[n++]);
    }
    for (i = 0;  i < 32;  i++)
    {
        s->eq_coeff_save[i].re = bitsf(f[n++]);
        s->eq_coeff_save[i].im = bitsf(f[n++]);
    }
    for (i = 0;  i < 32;  i++)
    {
        s->eq_buf[i].re = bitsf(f[n++]);
        s->eq_buf[i].im = bitsf(f[n++]);
    }
    n = 0;
    s->bit_rate = (int) iw[n++];
    s->rrc_filter_step = (int) iw[n++];
    s->scramble_reg = iw[n++];
    s->scrambler_pattern_count = (int) iw[n++];
    s->training_bc = (int) iw[n++];
    s->old_train = (iw[n++] != 0);
    s->training_stage = (int) iw[n++];
    s->training_count = (int) iw[n++];
    s->last_sample = (int16_t) iw[n++];
    s->signal_present = (int) iw[n++];
    s->carrier_drop_pending = (int) iw[n++];
    s->low_samples = (int) iw[n++];
    s->high_sample = (int16_t) iw[n++];
    s->constellation_state = (int) iw[n++];
    s->carrier_phase = iw[n++];
    s->carrier_phase_rate = (int32_t) iw[n++];
    s->carrier_phase_rate_save = (int32_t) iw[n++];
    s->power.reading = (int32_t) iw[n++];
    s->carrier_on_power = (int32_t) iw[n++];
    s->carrier_off_power = (int32_t) iw[n++];
    s->eq_step = (int) iw[n++];
    s->eq_put_step = (int) iw[n++];
    s->eq_skip = (int) iw[n++];
    s->baud_half = (int) iw[n++];
    s->gardner_integrate = (int) iw[n++];
    s->gardner_step = (int) iw[n++];
    s->total_baud_timing_correction = (int) iw[n++];
    s->last_angles[0] = (int32_t) iw[n++];
    s->last_angles[1] = (int32_t) iw[n++];
    for (i = 0;  i < 16;  i++)
        s->diff_angles[i] = (int32_t) iw[n++];
    return SPANGPU_OK;
}

/* ---- V.17: 246 float words and 301 integer words ------------------------------------------------------------ */
#define NF17    246
#define NI17    301

int spangpu_v17_import_state(spangpu_modem_t *bank, int channel, const spangpu_ref_v17_rx_t *s)
{
    uint32_t w[NF17 + NI17 + 64];
    uint32_t *f = w;
    uint32_t *iw = w + NF17;
    int n = 0;
    int i;

    if (bank == NULL  ||  s == NULL)
        return SPANGPU_ERR_BAD_ARG;
    if (spangpu_modem_get_state(bank, channel, w) != NF17 + NI17)
        return SPANGPU_ERR_BAD_ARG;
    if ((int) iw[0] != s->bit_rate)
        return SPANGPU_ERR_BAD_ARG;             /* the bank's constellation and space map are those of its own rate */
    f[n++] = fbits(s->agc_scaling);
    f[n++] = fbits(s->agc_scaling_save);
    f[n++] = fbits(s->eq_delta);
    f[n++] = fbits(s->training_error);
    f[n++] = fbits(s->carrier_track_p);
    f[n++] = fbits(s->carrier_track_i);
    f[n++] = fbits(s->godard.low_band_edge[0]);
    f[n++] = fbits(s->godard.low_band_edge[1]);
    f[n++] = fbits(s->godard.high_band_edge[0]);
    f[n++] = fbits(s->godard.high_band_edge[1]);
    f[n++] = fbits(s->godard.dc_filter[0]);
    f[n++] = fbits(s->godard.dc_filter[1]);
    f[n++] = fbits(s->godard.baud_phase);
    for (i = 0;  i < 27;  i++)
        f[n++] = fbits(s->rrc_filter[i]);
    for (i = 0;  i < 33;  i++)
    {
        f[n++] = fbits(s->eq_coeff[i].re);
        f[n++] = fbits(s->eq_coeff[i].im);
    }
    for (i = 0;  i < 33;  i++)
    {
        f[n++] = fbits(s->eq_coeff_save[i].re);
        f[n++] = fbits(s->eq_coeff_save[i].im);
    }
    for (i = 0;  i < 33;  i++)
    {
        f[n++] = fbits(s->eq_buf[i].re);
        f[n++] = fbits(s->eq_buf[i].im);
    }
    for (i = 0;  i < 8;  i++)
        f[n++] = fbits(s->distances[i]);
    n = 0;
    iw[n++] = (uint32_t) s->bit_rate;
    iw[n++] = (uint32_t) s->rrc_filter_step;
    iw[n++] = (uint32_t) s->diff;
    iw[n++] = s->scramble_reg;
    iw[n++] = (uint32_t) s->scrambler_tap;
    iw[n++] = s->short_train  ?  1  :  0;
    iw[n++] = (uint32_t) s->training_stage;
    iw[n++] = (uint32_t) s->training_count;
    iw[n++] = (uint32_t) (int32_t) s->last_sample;
    iw[n++] = (uint32_t) s->signal_present;
    iw[n++] = (uint32_t) s->carrier_drop_pending;
    iw[n++] = (uint32_t) s->low_samples;
    iw[n++] = (uint32_t) (int32_t) s->high_sample;
    iw[n++] = s->carrier_phase;
    iw[n++] = (uint32_t) s->carrier_phase_rate;
    iw[n++] = (uint32_t) s->carrier_phase_rate_save;
    iw[n++] = (uint32_t) s->power.reading;
    iw[n++] = (uint32_t) s->carrier_on_power;
    iw[n++] = (uint32_t) s->carrier_off_power;
    iw[n++] = (uint32_t) s->eq_step;
    iw[n++] = (uint32_t) s->eq_put_step;
    iw[n++] = (uint32_t) s->eq_skip;
    iw[n++] = (uint32_t) s->baud_half;
    iw[n++] = (uint32_t) s->last_angles[0];
    iw[n++] = (uint32_t) s->last_angles[1];
    for (i = 0;  i < 16;  i++)
        iw[n++] = (uint32_t) s->diff_angles[i];
    iw[n++] = (uint32_t) s->space_map;
    iw[n++] = (uint32_t) s->bits_per_symbol;
    iw[n++] = (uint32_t) s->trellis_ptr;
    iw[n++] = (uint32_t) s->godard.total_baud_timing_correction;
    for (i = 0;  i < 16*8;  i++)
        iw[n++] = (uint32_t) s->full_path_to_past_state_locations[i >> 3][i & 7];
    for (i = 0;  i < 16*8;  i++)
        iw[n++] = (uint32_t) s->past_state_locations[i >> 3][i & 7];
    return spangpu_modem_set_state(bank, channel, w);
}

int spangpu_v17_export_state(spangpu_modem_t *bank, int channel, spangpu_ref_v17_rx_t *s)
{
    uint32_t w[NF17 + NI17 + 64];
    const uint32_t *f = w;
    const uint32_t *iw = w + NF17;
    int n = 0;
    int i;

    if (bank == NULL  ||  s == NULL)
        return SPANGPU_ERR_BAD_ARG;
    if (spangpu_modem_get_state(bank, channel, w) != NF17 + NI17)
        return SPANGPU_ERR_BAD_ARG;
    if ((int) iw[0] != s->bit_rate)
        return SPANGPU_ERR_BAD_ARG;             /* (the struct's constellation pointer is the one of its own rate) */
    s->agc_scaling = bitsf(f[n++]);
    s->agc_scaling_save = bitsf(f[n++]);
    s->eq_delta = bitsf(f[n++]);
    s->training_error = bitsf(f[n++]);
    s->carrier_track_p = bitsf(f[n++]);
    s->carrier_track_i = bitsf(f[n++]);
    s->godard.low_band_edge[0] = bitsf(f[n++]);
    s->godard.low_band_edge[1] = bitsf(f[n++]);
    s->godard.high_band_edge[0] = bitsf(f[n++]);
    s->godard.high_band_edge[1] = bitsf(f[n++]);
    s->godard.dc_filter[0] = bitsf(f[n++]);
    s->godard.dc_filter[1] = bitsf(f[n++]);
    s->godard.baud_phase = bitsf(f[n++]);
    for (i = 0;  i < 27;  i++)
        s->rrc_filter[i] = bitsf(f[n++]);
    for (i = 0;  i < 33;  i++)
    {
        s->eq_coeff[i].re = bitsf(f[n++]);
        s->eq_coeff[i].im = bitsf(f[n++]);
    }
    for (i = 0;  i < 33;  i++)
    {
        s->eq_coeff_save[i].re = bitsf(f[n++]);
        s->eq_coeff_save[i].im = bitsf(f[n++]);
    }
    for (i = 0;  i < 33;  i++)
    {
        s->eq_buf[i].re = bitsf(f[n++]);
        s->eq_buf[i].im = bitsf(f[n++]);
    }
    for (i = 0;  i < 8;  i++)
        s->distances[i] = bitsf(f[n++]);
    n = 1;                                      /* (bit_rate: checked above, as it is) */
    s->rrc_filter_step = (int) iw[n++];
    s->diff = (int) iw[n++];
    s->scramble_reg = iw[n++];
    s->scrambler_tap = (int) iw[n++];
    s->short_train = (iw[n++] != 0);
    s->training_stage = (int) iw[n++];
    s->training_count = (int) iw[n++];
    s->last_sample = (int16_t) iw[n++];
    s->signal_present = (int) iw[n++];
    s->carrier_drop_pending = (int) iw[n++];
    s->low_samples = (int) iw[n++];
    s->high_sample = (int16_t) iw[n++];
    s->carrier_phase = iw[n++];
    s->carrier_phase_rate = (int32_t) iw[n++];
    s->carrier_phase_rate_save = (int32_t) iw[n++];
    s->power.reading = (int32_t) iw[n++];
    s->carrier_on_power = (int32_t) iw[n++];
    s->carrier_off_power = (int32_t) iw[n++];
    s->eq_step = (int) iw[n++];
    s->eq_put_step = (int) iw[n++];
    s->eq_skip = (int) iw[n++];
    s->baud_half = (int) iw[n++];
    s->last_angles[0] = (int32_t) iw[n++];
    s->last_angles[1] = (int32_t) iw[n++];
    for (i = 0;  i < 16;  i++)
        s->diff_angles[i] = (int32_t) iw[n++];
    s->space_map = (int) iw[n++];
    s->bits_per_symbol = (int) iw[n++];
    s->trellis_ptr = (int) iw[n++];
    s->godard.total_baud_timing_correction = (int) iw[n++];
    for (i = 0;  i < 16*8;  i++)
        s->full_path_to_past_state_locations[i >> 3][i & 7] = (int) iw[n++];
    for (i = 0;  i < 16*8;  i++)
        s->past_state_locations[i >> 3][i & 7] = (int) iw[n++];
    return SPANGPU_OK;
}

/* ---- FSK: 28 scalar words, then the correlation window (4 words per position) ---------------------------------- */
#define FSK_SCALARS 28

static int fsk_to_words(const spangpu_ref_fsk_rx_t *s, int32_t *w)
{
    int n = 0;
    int i;
    int j;

    w[n++] = s->baud_rate;
    w[n++] = s->framing_mode;
    w[n++] = s->data_bits;
    w[n++] = s->parity;
    w[n++] = s->stop_bits;
    w[n++] = s->total_data_bits;
    w[n++] = s->carrier_on_power;
    w[n++] = s->carrier_off_power;
    w[n++] = s->power.reading;
    w[n++] = s->last_sample;
    w[n++] = s->signal_present;
    w[n++] = s->phase_rate[0];
    w[n++] = s->phase_rate[1];
    w[n++] = (int32_t) s->phase_acc[0];
    w[n++] = (int32_t) s->phase_acc[1];
    w[n++] = s->correlation_span;
    w[n++] = s->dot[0].re;
    w[n++] = s->dot[0].im;
    w[n++] = s->dot[1].re;
    w[n++] = s->dot[1].im;
    w[n++] = s->buf_ptr;
    w[n++] = s->frame_pos;
    w[n++] = s->frame_in_progress;
    w[n++] = s->baud_phase;
    w[n++] = s->last_bit;
    w[n++] = s->scaling_shift;
    w[n++] = s->parity_errors;
    w[n++] = s->framing_errors;
    for (i = 0;  i < s->correlation_span;  i++)
    {
        for (j = 0;  j < 2;  j++)
        {
            w[n++] = s->window[j][i].re;
            w[n++] = s->window[j][i].im;
        }
    }
    return n;
}

static int fsk_from_words(spangpu_ref_fsk_rx_t *s, const int32_t *w)
{
    int n = 0;
    int i;
    int j;

    s->baud_rate = w[n++];
    s->framing_mode = w[n++];
    s->data_bits = w[n++];
    s->parity = w[n++];
    s->stop_bits = w[n++];
    s->total_data_bits = w[n++];
    s->carrier_on_power = w[n++];
    s->carrier_off_power = w[n++];
    s->power.reading = w[n++];
    s->last_sample = (int16_t) w[n++];
    s->signal_present = w[n++];
    s->phase_rate[0] = w[n++];
    s->phase_rate[1] = w[n++];
    s->phase_acc[0] = (uint32_t) w[n++];
    s->phase_acc[1] = (uint32_t) w[n++];
    s->correlation_span = w[n++];
    s->dot[0].re = w[n++];
    s->dot[0].im = w[n++];
    s->dot[1].re = w[n++];
    s->dot[1].im = w[n++];
    s->buf_ptr = w[n++];
    s->frame_pos = w[n++];
    s->frame_in_progress = (uint16_t) w[n++];
    s->baud_phase = w[n++];
    s->last_bit = w[n++];
    s->scaling_shift = w[n++];
    s->parity_errors = w[n++];
    s->framing_errors = w[n++];
    for (i = 0;  i < s->correlation_span;  i++)
    {
        for (j = 0;  j < 2;  j++)
        {
            s->window[j][i].re = w[n++];
            s->window[j][i].im = w[n++];
        }
    }
    return n;
}

int spangpu_fsk_import_state(spangpu_fsk_t *bank, int channel, const spangpu_ref_fsk_rx_t *s)
{
    int32_t w[FSK_SCALARS + 4*128];

    if (bank == NULL  ||  s == NULL  ||  s->correlation_span < 1  ||  s->correlation_span > 128)
        return SPANGPU_ERR_BAD_ARG;
    if (spangpu_fsk_get_state(bank, channel, w) != SPANGPU_OK)
        return SPANGPU_ERR_BAD_ARG;
    /* a bank runs one baud rate and one pair of frequencies: the window length and the oscillators say which */
    if (w[0] != s->baud_rate  ||  w[11] != s->phase_rate[0]  ||  w[12] != s->phase_rate[1]  ||  w[15] != s->correlation_span)
        return SPANGPU_ERR_BAD_ARG;
    (void) fsk_to_words(s, w);
    return spangpu_fsk_set_state(bank, channel, w);
}

int spangpu_fsk_export_state(spangpu_fsk_t *bank, int channel, spangpu_ref_fsk_rx_t *s)
{
    int32_t w[FSK_SCALARS + 4*128];

    if (bank == NULL  ||  s == NULL)
        return SPANGPU_ERR_BAD_ARG;
    if (spangpu_fsk_get_state(bank, channel, w) != SPANGPU_OK)
        return SPANGPU_ERR_BAD_ARG;
    if (w[15] < 1  ||  w[15] > 128)
        return SPANGPU_ERR_STATE;
    (void) fsk_from_words(s, w);
    return SPANGPU_OK;
}

/* ---- modem connect tones: 18 words of the detector, then the V.21 receiver's words where it runs ------------------ */
#define MCT_WORDS   18

int spangpu_mct_import_state(spangpu_mct_t *bank, int channel, const spangpu_ref_mct_rx_t *s)
{
    int32_t w[MCT_WORDS + FSK_SCALARS + 4*128];
    int words;
    int n = 0;

    if (bank == NULL  ||  s == NULL)
        return SPANGPU_ERR_BAD_ARG;
    words = spangpu_mct_state_words(bank);
    if (words < MCT_WORDS  ||  words > MCT_WORDS + FSK_SCALARS + 4*128  ||  spangpu_mct_get_state(bank, channel, w) != SPANGPU_OK)
        return SPANGPU_ERR_BAD_ARG;
    if (w[0] != s->tone_type)
        return SPANGPU_ERR_BAD_ARG;
    if (words > MCT_WORDS  &&  (s->v21rx.correlation_span != w[MCT_WORDS + 15]  ||  s->v21rx.baud_rate != w[MCT_WORDS]))
        return SPANGPU_ERR_BAD_ARG;
    w[n++] = s->tone_type;
    w[n++] = (int32_t) fbits(s->znotch_1);
    w[n++] = (int32_t) fbits(s->znotch_2);
    w[n++] = (int32_t) fbits(s->z15hz_1);
    w[n++] = (int32_t) fbits(s->z15hz_2);
    w[n++] = s->notch_level;
    w[n++] = s->channel_level;
    w[n++] = s->am_level;
    w[n++] = s->tone_present;
    w[n++] = s->tone_on;
    w[n++] = s->tone_cycle_duration;
    w[n++] = s->good_cycles;
    w[n++] = s->hit;
    w[n++] = (int32_t) s->raw_bit_stream;
    w[n++] = s->num_bits;
    w[n++] = s->flags_seen;
    w[n++] = s->framing_ok_announced  ?  1  :  0;
    w[n++] = 0;
    if (words > MCT_WORDS)
        (void) fsk_to_words(&s->v21rx, w + MCT_WORDS);
    return spangpu_mct_set_state(bank, channel, w);
}

int spangpu_mct_export_state(spangpu_mct_t *bank, int channel, spangpu_ref_mct_rx_t *s)
{
    int32_t w[MCT_WORDS + FSK_SCALARS + 4*128];
    int words;
    int n = 1;

    if (bank == NULL  ||  s == NULL)
        return SPANGPU_ERR_BAD_ARG;
    words = spangpu_mct_state_words(bank);
    if (words < MCT_WORDS  ||  words > MCT_WORDS + FSK_SCALARS + 4*128  ||  spangpu_mct_get_state(bank, channel, w) != SPANGPU_OK)
        return SPANGPU_ERR_BAD_ARG;
    if (w[0] != s->tone_type)
        return SPANGPU_ERR_BAD_ARG;             /* (the struct must have been initialised for the same tone type) */
    s->znotch_1 = bitsf((uint32_t) w[n++]);
    s->znotch_2 = bitsf((uint32_t) w[n++]);
    s->z15hz_1 = bitsf((uint32_t) w[n++]);
    s->z15hz_2 = bitsf((uint32_t) w[n++]);
    s->notch_level = w[n++];
    s->channel_level = w[n++];
    s->am_level = w[n++];
    s->tone_present = w[n++];
    s->tone_on = w[n++];
    s->tone_cycle_duration = w[n++];
    s->good_cycles = w[n++];
    s->hit = w[n++];
    s->raw_bit_stream = (unsigned int) w[n++];
    s->num_bits = w[n++];
    s->flags_seen = w[n++];
    s->framing_ok_announced = (w[n++] != 0);
    if (words > MCT_WORDS)
    {
        /* the V.21 receiver's callbacks point into the detector that owns it: they stay */
        span_put_bit_func_t put_bit = s->v21rx.put_bit;
        void *put_bit_user_data = s->v21rx.put_bit_user_data;
        span_modem_status_func_t status_handler = s->v21rx.status_handler;
        void *status_user_data = s->v21rx.status_user_data;

        (void) fsk_from_words(&s->v21rx, w + MCT_WORDS);
        s->v21rx.put_bit = put_bit;
        s->v21rx.put_bit_user_data = put_bit_user_data;
        s->v21rx.status_handler = status_handler;
        s->v21rx.status_user_data = status_user_data;
    }
    return SPANGPU_OK;
}

/* ---- signalling tone receiver: 27 words ----------------------------------------------------------------------------- */
#define SIG_WORDS   27

int spangpu_sig_tone_rx_import_state(spangpu_sigtone_rx_t *bank, int channel, const spangpu_ref_sig_tone_rx_t *s)
{
    int32_t w[SIG_WORDS];
    int32_t thr[3];
    int n = 0;
    int j;

    if (bank == NULL  ||  s == NULL  ||  spangpu_sigtone_rx_thresholds(bank, thr) != SPANGPU_OK)
        return SPANGPU_ERR_BAD_ARG;
    /* a bank runs one tone type; what tells the types apart in the receiver's own fields is the detection ratio
       (2280 Hz against the 2600 Hz types) and, between those two, which tones a report may name -- the caller says which
       bank by handing the channel over; the thresholds are checked */
    if (thr[0] != s->flat_detection_threshold  ||  thr[1] != s->sharp_detection_threshold  ||  thr[2] != s->detection_ratio)
        return SPANGPU_ERR_BAD_ARG;
    for (j = 0;  j < 3;  j++)
    {
        w[n++] = (int32_t) fbits(s->tone[j].notch_z1[0]);
        w[n++] = (int32_t) fbits(s->tone[j].notch_z1[1]);
        w[n++] = (int32_t) fbits(s->tone[j].notch_z2[0]);
        w[n++] = (int32_t) fbits(s->tone[j].notch_z2[1]);
        w[n++] = s->tone[j].power.reading;
    }
    w[n++] = (int32_t) fbits(s->flat_z[0]);
    w[n++] = (int32_t) fbits(s->flat_z[1]);
    w[n++] = s->flat_power.reading;
    w[n++] = s->tone_persistence_timeout;
    w[n++] = s->last_sample_tone_present;
    w[n++] = s->flat_mode  ?  1  :  0;
    w[n++] = s->flat_mode_timeout;
    w[n++] = s->notch_insertion_timeout;
    w[n++] = s->signalling_state;
    w[n++] = s->signalling_state_duration;
    w[n++] = s->current_notch_filter;
    w[n++] = s->current_rx_tone;
    return spangpu_sigtone_rx_set_state(bank, channel, w);
}

int spangpu_sig_tone_rx_export_state(spangpu_sigtone_rx_t *bank, int channel, spangpu_ref_sig_tone_rx_t *s)
{
    int32_t w[SIG_WORDS];
    int32_t thr[3];
    int n = 0;
    int j;

    if (bank == NULL  ||  s == NULL  ||  spangpu_sigtone_rx_thresholds(bank, thr) != SPANGPU_OK)
        return SPANGPU_ERR_BAD_ARG;
    if (thr[0] != s->flat_detection_threshold  ||  thr[1] != s->sharp_detection_threshold  ||  thr[2] != s->detection_ratio)
        return SPANGPU_ERR_BAD_ARG;             /* (the struct must have been initialised for the bank's tone type) */
    if (spangpu_sigtone_rx_get_state(bank, channel, w) != SPANGPU_OK)
        return SPANGPU_ERR_BAD_ARG;
    for (j = 0;  j < 3;  j++)
    {
        s->tone[j].notch_z1[0] = bitsf((uint32_t) w[n++]);
        s->tone[j].notch_z1[1] = bitsf((uint32_t) w[n++]);
        s->tone[j].notch_z2[0] = bitsf((uint32_t) w[n++]);
        s->tone[j].notch_z2[1] = bitsf((uint32_t) w[n++]);
        s->tone[j].power.reading = w[n++];
    }
    s->flat_z[0] = bitsf((uint32_t) w[n++]);
    s->flat_z[1] = bitsf((uint32_t) w[n++]);
    s->flat_power.reading = w[n++];
    s->tone_persistence_timeout = w[n++];
    s->last_sample_tone_present = w[n++];
    s->flat_mode = (w[n++] != 0);
    s->flat_mode_timeout = w[n++];
    s->notch_insertion_timeout = w[n++];
    s->signalling_state = w[n++];
    s->signalling_state_duration = w[n++];
    s->current_notch_filter = w[n++];
    s->current_rx_tone = w[n++];
    return SPANGPU_OK;
}

/* sizeof() of the mirrors above and in spangpu_refstate.h, for the tests to hold against the reference build's own */
int spangpu_refstate_sizeof(const char *what)
{
    if (what == NULL)
        return -1;
    if (strcmp(what, "dtmf_rx_state_t") == 0)
        return (int) sizeof(spangpu_ref_dtmf_rx_t);
    if (strcmp(what, "goertzel_state_t") == 0)
        return (int) sizeof(spangpu_ref_goertzel_t);
    if (strcmp(what, "echo_can_state_t") == 0)
        return (int) sizeof(spangpu_ref_echo_can_t);
    if (strcmp(what, "bell_mf_rx_state_t") == 0)
        return (int) sizeof(spangpu_ref_bell_mf_rx_t);
    if (strcmp(what, "r2_mf_rx_state_t") == 0)
        return (int) sizeof(spangpu_ref_r2_mf_rx_t);
    if (strcmp(what, "v29_rx_state_t") == 0)
        return (int) sizeof(spangpu_ref_v29_rx_t);
    if (strcmp(what, "v27ter_rx_state_t") == 0)
        return (int) sizeof(spangpu_ref_v27ter_rx_t);
    if (strcmp(what, "v17_rx_state_t") == 0)
        return (int) sizeof(spangpu_ref_v17_rx_t);
    if (strcmp(what, "fsk_rx_state_t") == 0)
        return (int) sizeof(spangpu_ref_fsk_rx_t);
    if (strcmp(what, "modem_connect_tones_rx_state_t") == 0)
        return (int) sizeof(spangpu_ref_mct_rx_t);
    if (strcmp(what, "sig_tone_rx_state_t") == 0)
        return (int) sizeof(spangpu_ref_sig_tone_rx_t);
    return -1;
}
