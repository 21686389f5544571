/*
 * modem_common.h -- TEST INFRASTRUCTURE ONLY (see oracle.h).  Primitives shared by the modem
 * receiver restatements (v29_oracle.c, v27ter_oracle.c, v17_oracle.c).
 */
#if !defined(ORC_MODEM_COMMON_H)
#define ORC_MODEM_COMMON_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "oracle.h"

extern orc_modem_tables_t orc_modem_T;

/* power_meter.c:82-92 */
static inline int32_t level_dbm0(float level)
{
    float l;

    level -= (3.14f + 3.02f);
    if (level > 0.0)
        level = 0.0;
    l = powf(10.0f, level/10.0f)*(32767.0f*32767.0f);
    return (int32_t) l;
}

/* vector_float.c:890-939 */
static inline float circular_dot(const float x[], const float y[], int n, int pos)
{
    float z = 0.0f;
    float z1 = 0.0f;
    int i;

    for (i = 0;  i < n - pos;  i++)
        z += x[pos + i]*y[i];
    for (i = 0;  i < pos;  i++)
        z1 += x[i]*y[n - pos + i];
    z += z1;
    return z;
}

/* spandsp/arctan2.h:47-80 */
static inline int32_t arctan2_i(float y, float x)
{
    float abs_y;
    float angle;

    if (y == 0.0f)
        return (x < 0.0f)  ?  (int32_t) 0x80000000u  :  0;
    if (x == 0.0f)
        return (y < 0.0f)  ?  (int32_t) 0xc0000000u  :  0x40000000;
    abs_y = fabsf(y);
    if (x < 0.0f)
        angle = 3.0f - (x + abs_y)/(abs_y - x);
    else
        angle = 1.0f - (x - abs_y)/(abs_y + x);
    angle *= 536870912.0f;
    if (y < 0.0f)
        angle = -angle;
    return (int32_t) angle;
}

/* math_fixed.c:158-169 */
static inline int fixed_sqrt32(uint32_t x)
{
    int shift;
    int top;

    if (x == 0)
        return 0;
    top = 31;
    while (!(x & 0x80000000u))
    {
        x <<= 1;
        top--;
    }
    x >>= (31 - top);
    shift = 30 - (top & ~1);
    x <<= shift;
    return orc_modem_T.sqrt_tab[((x >> 24) & 0xFF) - 64] >> (shift >> 1);
}


/* cosf()/sinf() as the reference build gets them from its C library.  The reference calls libm's float trig once
   per training (the phase "spin", v29rx.c:618-623, v27ter_rx.c:661-667, v17rx.c); libm is a third-party dependency
   that is not in /root/reference, so its algorithm is restated here: glibc >= 2.28 (the reference build's libm is
   glibc 2.35), sysdeps/ieee754/flt-32/{s_sinf.c, s_cosf.c, sincosf.h} -- the ARM optimized-routines sincosf: reduce
   by pi/2 in double, then a degree-7 / degree-8 polynomial in double, rounded once to float.  This restatement was
   compared with the container's libm for EVERY float in [0, 2*pi] (1 086 918 620 values, cos and sin): 0 mismatches
   (tests/test_oracle_pin.py::test_trig_restatement samples it; the phase argument is always in [0, 2*pi)). */
typedef struct
{
    double c0, c1, c2, c3, c4;
    double s1, s2, s3;
} orc_sincos_poly_t;

static inline float orc_sincos_eval(double x, double x2, int negate_cos, int n)
{
    static const orc_sincos_poly_t P =
    {
        0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,
        -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13
    };
    double x3, x4, x6, x7, s, c, c1, c2, s1;
    double sg = negate_cos  ?  -1.0  :  1.0;

    if ((n & 1) == 0)
    {
        x3 = x*x2;
        s1 = P.s2 + x2*P.s3;
        x7 = x3*x2;
        s = x + x3*P.s1;
        return (float) (s + x7*s1);
    }
    x4 = x2*x2;
    c2 = sg*P.c3 + x2*(sg*P.c4);
    c1 = sg*P.c0 + x2*(sg*P.c1);
    x6 = x4*x2;
    c = c1 + x4*(sg*P.c2);
    return (float) (c + x6*c2);
}

static inline float orc_sincosf(float y, int want_cos)
{
    static const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    double x = y;
    double r;
    uint32_t top;
    int n;

    memcpy(&top, &y, 4);
    top = (top >> 20) & 0x7FF;
    if (top < 0x3F4)                        /* |y| < pi/4 */
    {
        if (top < 0x398)                    /* |y| < 2^-12 */
            return want_cos  ?  1.0f  :  y;
        return orc_sincos_eval(x, x*x, 0, want_cos);
    }
    r = x*0x1.45F306DC9C883p+23;
    n = ((int32_t) r + 0x800000) >> 24;
    x = x - n*0x1.921FB54442D18p0;
    return orc_sincos_eval(x*sign[n & 3], x*x, n & 2, want_cos  ?  (n ^ 1)  :  n);
}

static inline float orc_cosf(float y) { return orc_sincosf(y, 1); }
static inline float orc_sinf(float y) { return orc_sincosf(y, 0); }

/* dds_lookup_complexf (dds_float.c:2135-2177): {cos, sin} of a 32 bit phase from the 2048 entry sine table */
static inline void dds_complex(uint32_t phase, float z[2])
{
    z[0] = orc_modem_T.sine[(uint32_t) (phase + (1u << 30)) >> 21];
    z[1] = orc_modem_T.sine[phase >> 21];
}

/* cvec_circular_dot_prodf, complex_vector_float.c:137-196 */
static inline void ccircular_dot(const float x[][2], const float y[][2], int n, int pos, float z[2])
{
    float a_re = 0.0f;
    float a_im = 0.0f;
    float b_re = 0.0f;
    float b_im = 0.0f;
    int i;

    for (i = 0;  i < n - pos;  i++)
    {
        a_re += (x[pos + i][0]*y[i][0] - x[pos + i][1]*y[i][1]);
        a_im += (x[pos + i][0]*y[i][1] + x[pos + i][1]*y[i][0]);
    }
    for (i = 0;  i < pos;  i++)
    {
        b_re += (x[i][0]*y[n - pos + i][0] - x[i][1]*y[n - pos + i][1]);
        b_im += (x[i][0]*y[n - pos + i][1] + x[i][1]*y[n - pos + i][0]);
    }
    z[0] = a_re + b_re;
    z[1] = a_im + b_im;
}

/* cvec_circular_lmsf, complex_vector_float.c:199-219 (leak 0.9999) */
static inline void ccircular_lms(const float x[][2], float y[][2], int n, int pos, float err_re, float err_im)
{
    int i;
    int k;

    for (i = 0;  i < n;  i++)
    {
        k = pos + i;
        if (k >= n)
            k -= n;
        y[i][0] = y[i][0]*0.9999f + (x[k][1]*err_im + x[k][0]*err_re);
        y[i][1] = y[i][1]*0.9999f + (x[k][0]*err_im - x[k][1]*err_re);
    }
}

#endif
