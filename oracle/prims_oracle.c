/*
 * prims_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).  Restatement of the primitives that round 6 gave entry points
 * of their own (spandsp_amd/csrc/prim2_api.hip, shim_prims.c), each citing the reference lines it follows:
 *   periodogram, _prepare, _apply, _generate_coeffs, _generate_phase_offset, _freq_error     src/tone_detect.c:208-312
 *   vec_dot_prodf, vec_lmsf                        src/vector_float.c:890-900, 942, 982-992
 *   cvec_dot_prodf, cvec_lmsf                      src/complex_vector_float.c:137-150, 199-212
 *   fixed_sqrt32 / arctan2 / dds_complexf over arrays: the static restatements of modem_common.h (math_fixed.c:158-169,
 *   spandsp/arctan2.h:47-80, dds_float.c:2135-2187) that the receiver oracles use, on the tables orc_modem_set_tables() was given
 * Pinned to the reference compiled from its own sources in tests/test_oracle_pin.py (test_prims2_live) and to
 * tests/golden/prims2.npz.  cosf / sinf are the C library's, as in the reference.
 */
#include <math.h>
#include <stdint.h>

#include "oracle.h"
#include "modem_common.h"

/* tone_detect.c:208-225 */
ORC_API void orc_periodogram(const float coeffs[][2], const float amp[][2], int len, float out[2])
{
    float re = 0.0f;
    float im = 0.0f;
    int i;

    for (i = 0;  i < len/2;  i++)
    {
        const float sre = amp[i][0] + amp[len - 1 - i][0];
        const float sim = amp[i][1] + amp[len - 1 - i][1];
        const float dre = amp[i][0] - amp[len - 1 - i][0];
        const float dim = amp[i][1] - amp[len - 1 - i][1];
        re += (coeffs[i][0]*sre - coeffs[i][1]*dim);
        im += (coeffs[i][0]*sim + coeffs[i][1]*dre);
    }
    out[0] = re;
    out[1] = im;
}

/* tone_detect.c:228-239 */
ORC_API int orc_periodogram_prepare(float sum[][2], float diff[][2], const float amp[][2], int len)
{
    int i;

    for (i = 0;  i < len/2;  i++)
    {
        sum[i][0] = amp[i][0] + amp[len - 1 - i][0];
        sum[i][1] = amp[i][1] + amp[len - 1 - i][1];
        diff[i][0] = amp[i][0] - amp[len - 1 - i][0];
        diff[i][1] = amp[i][1] - amp[len - 1 - i][1];
    }
    return len/2;
}

/* tone_detect.c:242-255 */
ORC_API void orc_periodogram_apply(const float coeffs[][2], const float sum[][2], const float diff[][2], int len, float out[2])
{
    float re = 0.0f;
    float im = 0.0f;
    int i;

    for (i = 0;  i < len/2;  i++)
    {
        re += (coeffs[i][0]*sum[i][0] - coeffs[i][1]*diff[i][1]);
        im += (coeffs[i][0]*sum[i][1] + coeffs[i][1]*diff[i][0]);
    }
    out[0] = re;
    out[1] = im;
}

/* tone_detect.c:258-283 */
ORC_API int orc_periodogram_generate_coeffs(float coeffs[][2], float freq, int sample_rate, int window_len)
{
    float window;
    float sum = 0.0f;
    float x;
    int i;

    for (i = 0;  i < window_len/2;  i++)
    {
        window = 0.53836f - 0.46164f*cosf(2.0f*3.1415926535f*i/(window_len - 1.0f));
        x = (i - window_len/2.0f + 0.5f)*freq*2.0f*3.1415926535f/sample_rate;
        coeffs[i][0] = cosf(x)*window;
        coeffs[i][1] = -sinf(x)*window;
        sum += window;
    }
    sum = 1.0f/(2.0f*sum);
    for (i = 0;  i < window_len/2;  i++)
    {
        coeffs[i][0] *= sum;
        coeffs[i][1] *= sum;
    }
    return window_len/2;
}

/* tone_detect.c:286-296 */
ORC_API float orc_periodogram_generate_phase_offset(float offset[2], float freq, int sample_rate, int interval)
{
    const float x = 2.0f*3.1415926535f*(float) interval/(float) sample_rate;

    offset[0] = cosf(freq*x);
    offset[1] = sinf(freq*x);
    return 1.0f/x;
}

/* tone_detect.c:299-310 (complex_mulf(): spandsp/complex.h) */
ORC_API float orc_periodogram_freq_error(const float phase_offset[2], float scale, const float last_result[2], const float result[2])
{
    const float pre = last_result[0]*phase_offset[0] - last_result[1]*phase_offset[1];
    const float pim = last_result[0]*phase_offset[1] + last_result[1]*phase_offset[0];

    return scale*(result[1]*pre - result[0]*pim)/(result[0]*result[0] + result[1]*result[1]);
}

/* vector_float.c:890-900 */
ORC_API float orc_vec_dot_prodf(const float x[], const float y[], int n)
{
    float z = 0.0f;
    int i;

    for (i = 0;  i < n;  i++)
        z += x[i]*y[i];
    return z;
}

/* vector_float.c:982-992 (LMS_LEAK_RATE 0.9999f, :942) */
ORC_API void orc_vec_lmsf(const float x[], float y[], int n, float error)
{
    int i;

    for (i = 0;  i < n;  i++)
        y[i] = y[i]*0.9999f + x[i]*error;
}

/* complex_vector_float.c:137-150 */
ORC_API void orc_cvec_dot_prodf(const float x[][2], const float y[][2], int n, float z[2])
{
    float re = 0.0f;
    float im = 0.0f;
    int i;

    for (i = 0;  i < n;  i++)
    {
        re += (x[i][0]*y[i][0] - x[i][1]*y[i][1]);
        im += (x[i][0]*y[i][1] + x[i][1]*y[i][0]);
    }
    z[0] = re;
    z[1] = im;
}

/* complex_vector_float.c:199-212 */
ORC_API void orc_cvec_lmsf(const float x[][2], float y[][2], int n, const float error[2])
{
    int i;

    for (i = 0;  i < n;  i++)
    {
        y[i][0] = y[i][0]*0.9999f + (x[i][1]*error[1] + x[i][0]*error[0]);
        y[i][1] = y[i][1]*0.9999f + (x[i][0]*error[1] - x[i][1]*error[0]);
    }
}

ORC_API void orc_fixed_sqrt32_batch(const uint32_t *x, uint16_t *out, int n)
{
    int i;

    for (i = 0;  i < n;  i++)
        out[i] = (uint16_t) fixed_sqrt32(x[i]);
}

ORC_API void orc_arctan2_batch(const float *y, const float *x, int32_t *out, int n)
{
    int i;

    for (i = 0;  i < n;  i++)
        out[i] = arctan2_i(y[i], x[i]);
}

/* dds_complexf(), dds_float.c:2179-2187, n times per item */
ORC_API void orc_dds_complexf_batch(uint32_t *phase_acc, const int32_t *phase_rate, float *out, int items, int n)
{
    int i;
    int k;

    for (i = 0;  i < items;  i++)
    {
        for (k = 0;  k < n;  k++)
        {
            dds_complex(phase_acc[i], &out[2*((size_t) i*n + k)]);
            phase_acc[i] += (uint32_t) phase_rate[i];
        }
    }
}
