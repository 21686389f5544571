/*
 * ref_glue_v17.c -- TEST INFRASTRUCTURE ONLY.  OUR accessors over the reference build's V.17
 * receiver: its constant tables as data and a state snapshot in the word order of orc_v17_t
 * (oracle/oracle.h).  Kept apart from ref_glue.c because the V.17 and V.29 generated headers
 * use the same identifiers.  Compiled only into oracle/_ref/libspandsp_ref.so; #includes
 * reference headers from /root/reference/src at build time.
 */
#include <stdlib.h>
#include <inttypes.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <stdbool.h>

#include "spandsp/telephony.h"
#include "spandsp/logging.h"
#include "spandsp/complex.h"
#include "spandsp/async.h"
#include "spandsp/power_meter.h"
#include "spandsp/godard.h"
#include "spandsp/v29rx.h"
#include "spandsp/v17rx.h"
#include "spandsp/private/logging.h"
#include "spandsp/private/power_meter.h"
#include "spandsp/private/godard.h"
#include "spandsp/private/v17rx.h"

#define FP_SCALE(x)                     (x)
#define FP_CONSTELLATION_SCALE(x)       (x)
#include "v17_v32bis_rx_rrc.h"
#include "v17_v32bis_tx_constellation_maps.h"
#include "v17_v32bis_rx_constellation_maps.h"
#include "v17_v32bis_rx_godard.h"

#define GLUE __attribute__((visibility("default")))

GLUE void glue_v17_tables(float re[192*27], float im[192*27], float godard[9], int32_t steps[2])
{
    int i;
    int j;

    for (j = 0;  j < 192;  j++)
    {
        for (i = 0;  i < 27;  i++)
        {
            re[j*27 + i] = rx_pulseshaper_re[j][i];
            im[j*27 + i] = rx_pulseshaper_im[j][i];
        }
    }
    for (i = 0;  i < 3;  i++)
    {
        godard[i] = godard_desc.low_band_edge_coeff[i];
        godard[3 + i] = godard_desc.high_band_edge_coeff[i];
    }
    godard[6] = godard_desc.mixed_band_edges_coeff_3;
    godard[7] = godard_desc.coarse_trigger;
    godard[8] = godard_desc.fine_trigger;
    steps[0] = godard_desc.coarse_step;
    steps[1] = godard_desc.fine_step;
}

/* out: 128 + 64 + 32 + 16 + 4 points, {re, im} each */
GLUE void glue_v17_constellations(float out[244*2])
{
    int i;
    int n = 0;

    for (i = 0;  i < 128;  i++, n++) { out[2*n] = v17_v32bis_14400_constellation[i].re; out[2*n + 1] = v17_v32bis_14400_constellation[i].im; }
    for (i = 0;  i < 64;  i++, n++) { out[2*n] = v17_v32bis_12000_constellation[i].re; out[2*n + 1] = v17_v32bis_12000_constellation[i].im; }
    for (i = 0;  i < 32;  i++, n++) { out[2*n] = v17_v32bis_9600_constellation[i].re; out[2*n + 1] = v17_v32bis_9600_constellation[i].im; }
    for (i = 0;  i < 16;  i++, n++) { out[2*n] = v17_v32bis_7200_constellation[i].re; out[2*n + 1] = v17_v32bis_7200_constellation[i].im; }
    for (i = 0;  i < 4;  i++, n++) { out[2*n] = v17_v32bis_4800_constellation[i].re; out[2*n + 1] = v17_v32bis_4800_constellation[i].im; }
}

GLUE void glue_v17_constel_maps(uint8_t maps[4*36*36*8], uint8_t map4800[36*36])
{
    memcpy(maps, constel_maps, sizeof(constel_maps));
    memcpy(map4800, constel_map_4800, sizeof(constel_map_4800));
}

/* Same word order as orc_v17_t: 246 floats, 301 ints */
GLUE void glue_v17_rx_snapshot(v17_rx_state_t *s, float f[246], int32_t w[301])
{
    int i;
    int n = 0;

    f[n++] = s->agc_scaling;
    f[n++] = s->agc_scaling_save;
    f[n++] = s->eq_delta;
    f[n++] = s->training_error;
    f[n++] = s->carrier_track_p;
    f[n++] = s->carrier_track_i;
    f[n++] = s->godard.low_band_edge[0];
    f[n++] = s->godard.low_band_edge[1];
    f[n++] = s->godard.high_band_edge[0];
    f[n++] = s->godard.high_band_edge[1];
    f[n++] = s->godard.dc_filter[0];
    f[n++] = s->godard.dc_filter[1];
    f[n++] = s->godard.baud_phase;
    for (i = 0;  i < 27;  i++)
        f[n++] = s->rrc_filter[i];
    for (i = 0;  i < 33;  i++) { f[n++] = s->eq_coeff[i].re; f[n++] = s->eq_coeff[i].im; }
    for (i = 0;  i < 33;  i++) { f[n++] = s->eq_coeff_save[i].re; f[n++] = s->eq_coeff_save[i].im; }
    for (i = 0;  i < 33;  i++) { f[n++] = s->eq_buf[i].re; f[n++] = s->eq_buf[i].im; }
    for (i = 0;  i < 8;  i++)
        f[n++] = s->distances[i];
    n = 0;
    w[n++] = s->bit_rate;
    w[n++] = s->rrc_filter_step;
    w[n++] = s->diff;
    w[n++] = (int32_t) s->scramble_reg;
    w[n++] = s->scrambler_tap;
    w[n++] = s->short_train;
    w[n++] = s->training_stage;
    w[n++] = s->training_count;
    w[n++] = s->last_sample;
    w[n++] = s->signal_present;
    w[n++] = s->carrier_drop_pending;
    w[n++] = s->low_samples;
    w[n++] = s->high_sample;
    w[n++] = (int32_t) s->carrier_phase;
    w[n++] = s->carrier_phase_rate;
    w[n++] = s->carrier_phase_rate_save;
    w[n++] = s->power.reading;
    w[n++] = s->carrier_on_power;
    w[n++] = s->carrier_off_power;
    w[n++] = s->eq_step;
    w[n++] = s->eq_put_step;
    w[n++] = s->eq_skip;
    w[n++] = s->baud_half;
    w[n++] = s->last_angles[0];
    w[n++] = s->last_angles[1];
    for (i = 0;  i < 16;  i++)
        w[n++] = s->diff_angles[i];
    w[n++] = s->space_map;
    w[n++] = s->bits_per_symbol;
    w[n++] = s->trellis_ptr;
    w[n++] = s->godard.total_baud_timing_correction;
    for (i = 0;  i < 16*8;  i++)
        w[n++] = s->full_path_to_past_state_locations[i >> 3][i & 7];
    for (i = 0;  i < 16*8;  i++)
        w[n++] = s->past_state_locations[i >> 3][i & 7];
}

/* ---- V.17 transmitter: its pulse shaper table and a state snapshot (word order of oracle/v17tx_oracle.c) ---- */
#include "spandsp/v17tx.h"
#include "spandsp/private/v17tx.h"
#include "v17_v32bis_tx_rrc.h"

GLUE void glue_v17_tx_table(float out[10*9])
{
    int i;
    int j;

    for (j = 0;  j < 10;  j++)
    {
        for (i = 0;  i < 9;  i++)
            out[j*9 + i] = tx_pulseshaper[j][i];
    }
}

GLUE int glue_v17_tx_snapshot(const v17_tx_state_t *s, uint32_t *out)
{
    int n = 0;
    int i;

    out[n++] = (uint32_t) s->bit_rate;
    memcpy(&out[n++], &s->gain, 4);
    out[n++] = (uint32_t) s->diff;
    for (i = 0;  i < V17_TX_FILTER_STEPS;  i++)
        memcpy(&out[n++], &s->rrc_filter_re[i], 4);
    for (i = 0;  i < V17_TX_FILTER_STEPS;  i++)
        memcpy(&out[n++], &s->rrc_filter_im[i], 4);
    out[n++] = (uint32_t) s->rrc_filter_step;
    out[n++] = s->scramble_reg;
    out[n++] = (uint32_t) s->convolution;
    out[n++] = (uint32_t) s->in_training;
    out[n++] = (uint32_t) s->training_step;
    out[n++] = (uint32_t) s->short_train;
    out[n++] = s->carrier_phase;
    out[n++] = (uint32_t) s->carrier_phase_rate;
    out[n++] = (uint32_t) s->baud_phase;
    out[n++] = (uint32_t) s->constellation_state;
    return n;
}
