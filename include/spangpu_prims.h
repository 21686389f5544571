/*
 * spangpu_prims.h -- libspangpu_prims.so: the receivers' inner primitives under their spandsp names, an OPT-IN library.
 *
 * Why a library of its own: these are names libspandsp itself calls from inside its modules -- vec_circular_dot_prodf() and
 * cvec_circular_lmsf() from the modem receivers, power_meter_update() from fsk.c, v22bis, modem_connect_tones.c, ... some 57
 * calls per sample in all.  Exported from libspangpu.so they would, in a process that also loads libspandsp (INTEGRATION.md
 * envisages one: logging.c and the modules outside the hot path stay with libspandsp), capture those internal calls according
 * to link order -- each becoming a hipMalloc, two copies and a launch.  So libspangpu.so exports none of them, and a caller
 * that wants the by-name forms links libspangpu_prims.so explicitly (-lspangpu_prims -lspangpu), AFTER deciding that no
 * libspandsp in the process should see them (or with -Bsymbolic / a version script on its own side).  INTEGRATION.md section 5.
 *
 * Each call is one item through the batched entry points of libspangpu.so (csrc/prim_api.hip, csrc/prim2_api.hip; a launch per
 * call: the plumbing form -- the receivers run these fused in their kernels, a caller with many items uses spangpu_*_batch()).
 * No host arithmetic behind the device functions: without a HIP device float results are NaN, power_meter_update() returns
 * INT32_MIN, godard_ted_per_baud() returns 0 and leaves the state as it was.  Table making (descriptors, coefficient sets) is
 * host code, as in the reference.  The device is SPANGPU_DEVICE (environment, default 0).
 *
 * Reference declarations being replaced:
 *   vec_dot_prodf, vec_circular_dot_prodf, vec_lmsf, vec_circular_lmsf
 *                                                 src/spandsp/vector_float.h:166-188          src/vector_float.c:890-900,932-939,982-1000
 *   cvec_dot_prodf, cvec_circular_dot_prodf, cvec_lmsf, cvec_circular_lmsf
 *                                                 src/spandsp/complex_vector_float.h:141-163  src/complex_vector_float.c:137-150,187-219
 *   power_meter_t, power_meter_init/_release/_free/_damping/_update/_rx/_current
 *                                                 src/spandsp/power_meter.h:34-94, private/power_meter.h:33-40   src/power_meter.c:44-113
 *   godard_ted_descriptor_t, godard_ted_state_t, godard_ted_make_descriptor/_free_descriptor/_init/_release/_free/_correction/_rx/_per_baud
 *                                                 src/spandsp/godard.h:57-124, private/godard.h:29-54            src/godard.c:70-249
 *     (the float build's structs, field for field)
 *   periodogram, periodogram_prepare, periodogram_apply, periodogram_generate_coeffs, periodogram_generate_phase_offset,
 *   periodogram_freq_error                        src/spandsp/tone_detect.h:202-249           src/tone_detect.c:208-312
 *   fixed_sqrt32                                  src/spandsp/math_fixed.h                    src/math_fixed.c:158-169
 *   dds_lookup_complexf, dds_complexf, dds_advancef   src/spandsp/dds.h                       src/dds_float.c:2135-2187
 */
#if !defined(SPANGPU_PRIMS_H)
#define SPANGPU_PRIMS_H

#include "spangpu_spandsp.h"

#if !defined(SPANGPU_PRIMS_API)
#define SPANGPU_PRIMS_API __attribute__((visibility("default")))
#endif

#if defined(__cplusplus)
extern "C" {
#endif

typedef struct power_meter_s
{
    int shift;
    int32_t reading;
} power_meter_t;
typedef struct godard_ted_descriptor_s
{
    float low_band_edge_coeff[3];
    float high_band_edge_coeff[3];
    float mixed_band_edges_coeff_3;
    float coarse_trigger;
    float fine_trigger;
    int coarse_step;
    int fine_step;
} godard_ted_descriptor_t;
typedef struct godard_ted_state_s
{
    godard_ted_descriptor_t desc;
    float low_band_edge[2];
    float high_band_edge[2];
    float dc_filter[2];
    float baud_phase;
    int total_baud_timing_correction;
} godard_ted_state_t;
SPANGPU_PRIMS_API float vec_circular_dot_prodf(const float x[], const float y[], int n, int pos);
SPANGPU_PRIMS_API void vec_circular_lmsf(const float x[], float y[], int n, int pos, float error);
SPANGPU_PRIMS_API complexf_t cvec_circular_dot_prodf(const complexf_t x[], const complexf_t y[], int n, int pos);
SPANGPU_PRIMS_API void cvec_circular_lmsf(const complexf_t x[], complexf_t y[], int n, int pos, const complexf_t *error);
SPANGPU_PRIMS_API power_meter_t *power_meter_init(power_meter_t *s, int shift);
SPANGPU_PRIMS_API int power_meter_release(power_meter_t *s);
SPANGPU_PRIMS_API int power_meter_free(power_meter_t *s);
SPANGPU_PRIMS_API power_meter_t *power_meter_damping(power_meter_t *s, int shift);
SPANGPU_PRIMS_API int32_t power_meter_update(power_meter_t *s, int16_t amp);
SPANGPU_PRIMS_API int32_t power_meter_rx(power_meter_t *s, int16_t amp[], int len);
SPANGPU_PRIMS_API int32_t power_meter_current(power_meter_t *s);
SPANGPU_PRIMS_API godard_ted_descriptor_t *godard_ted_make_descriptor(godard_ted_descriptor_t *desc, float sample_rate, float baud_rate, float carrier_freq,
                                                                float alpha, float coarse_trigger, float fine_trigger, int coarse_step, int fine_step);
SPANGPU_PRIMS_API int godard_ted_free_descriptor(godard_ted_descriptor_t *s);
SPANGPU_PRIMS_API int godard_ted_correction(godard_ted_state_t *s);
SPANGPU_PRIMS_API void godard_ted_rx(godard_ted_state_t *s, float sample);
SPANGPU_PRIMS_API int godard_ted_per_baud(godard_ted_state_t *s);
SPANGPU_PRIMS_API godard_ted_state_t *godard_ted_init(godard_ted_state_t *s, const godard_ted_descriptor_t *desc);
SPANGPU_PRIMS_API int godard_ted_release(godard_ted_state_t *s);
SPANGPU_PRIMS_API int godard_ted_free(godard_ted_state_t *s);
SPANGPU_PRIMS_API float vec_dot_prodf(const float x[], const float y[], int n);
SPANGPU_PRIMS_API void vec_lmsf(const float x[], float y[], int n, float error);
SPANGPU_PRIMS_API complexf_t cvec_dot_prodf(const complexf_t x[], const complexf_t y[], int n);
SPANGPU_PRIMS_API void cvec_lmsf(const complexf_t x[], complexf_t y[], int n, const complexf_t *error);
SPANGPU_PRIMS_API complexf_t periodogram(const complexf_t coeffs[], const complexf_t amp[], int len);
SPANGPU_PRIMS_API int periodogram_prepare(complexf_t sum[], complexf_t diff[], const complexf_t amp[], int len);
SPANGPU_PRIMS_API complexf_t periodogram_apply(const complexf_t coeffs[], const complexf_t sum[], const complexf_t diff[], int len);
SPANGPU_PRIMS_API int periodogram_generate_coeffs(complexf_t coeffs[], float freq, int sample_rate, int window_len);
SPANGPU_PRIMS_API float periodogram_generate_phase_offset(complexf_t *offset, float freq, int sample_rate, int interval);
SPANGPU_PRIMS_API float periodogram_freq_error(const complexf_t *phase_offset, float scale, const complexf_t *last_result, const complexf_t *result);
SPANGPU_PRIMS_API uint16_t fixed_sqrt32(uint32_t x);
SPANGPU_PRIMS_API complexf_t dds_lookup_complexf(uint32_t phase);
SPANGPU_PRIMS_API complexf_t dds_complexf(uint32_t *phase_acc, int32_t phase_rate);
SPANGPU_PRIMS_API void dds_advancef(uint32_t *phase_acc, int32_t phase_rate);

#if defined(__cplusplus)
}
#endif

#endif
