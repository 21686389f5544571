/* modem_tables.h -- constant tables of the modem receivers, built on the host at load time
   (see modem_tables.c for the recipes and their reference citations). */
#if !defined(SPG_MODEM_TABLES_H)
#define SPG_MODEM_TABLES_H

#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#define SPG_SINE_LEN    2048

void spg_make_sine_table(float out[SPG_SINE_LEN]);
void spg_make_sqrt_table(uint16_t out[193]);
/* re/im: [coeff_sets][coeffs_per_filter] */
int spg_make_rx_pulseshaper(int coeff_sets, int coeffs_per_filter, double carrier_hz, double baud_rate,
                            double excess_bandwidth, float *re, float *im);

/* out: [coeff_sets][coeffs_per_filter] */
int spg_make_tx_pulseshaper(int coeff_sets, int coeffs_per_filter, double excess_bandwidth, float *out);

void spg_make_godard(double carrier, double baud_rate, double alpha, float out[7]);
void spg_make_v29_space_map(uint8_t out[400]);
int spg_v17_constellation_size(int bit_rate);
int spg_make_v17_constellation(int bit_rate, int8_t out[][2]);
void spg_make_v17_rx_maps(uint8_t maps[4*36*36*8], uint8_t map_4800[36*36]);

/* Tone generator descriptors (tone_gen_descriptor_init(), tone_generate.c:60-120).  Called with run-time
   arguments from another translation unit so that powf() is libm's, as it is in the reference. */
int32_t spg_dds_phase_ratef(float hz);
float spg_dds_scaling_dbm0f(float level);
float spg_db_to_amplitude_ratio(float db);
/* out: rate[4], gain[4] (as float bits), duration[4], repeat = 13 words */
void spg_make_tone_descriptor(int32_t out[13], int f1, int l1, int f2, int l2, int d1, int d2, int d3, int d4, int repeat);

#if defined(__cplusplus)
}
#endif

#endif
