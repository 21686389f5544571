/*
 * awgn_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's noise source:
 *
 *   ran_init / ran (Numerical Recipes ran1: three LCGs and a 97 entry shuffle table)   src/awgn.c:82-131
 *   awgn_init_dbm0 / awgn_init_dbov                                                    src/awgn.c:133-152
 *   awgn (polar Box-Muller, one pair per two calls)                                    src/awgn.c:168-195
 *   fsaturate                                                                          src/spandsp/saturated.h:152-159
 *
 * pow() (once, at init) and sqrt() (correctly rounded everywhere) are libm's, as in the reference; log() is the
 * restatement of GNU libc's routine in glibc_log.c, so that the result does not depend on which build of log() the
 * host's C library selects.
 */
#include <math.h>
#include <string.h>

#include "oracle.h"

#define M1  259200
#define IA1 7141
#define IC1 54773
#define RM1 (1.0/(double) M1)
#define M2  134456
#define IA2 8121
#define IC2 28411
#define RM2 (1.0/(double) M2)
#define M3  243000
#define IA3 4561
#define IC3 51349

ORC_API void orc_awgn_init_dbm0(orc_awgn_t *s, int idum, float level)
{
    memset(s, 0, sizeof(*s));
    if (idum < 0)
        idum = -idum;
    s->ix1 = (IC1 + idum)%M1;
    s->ix1 = (IA1*s->ix1 + IC1)%M1;
    s->ix2 = s->ix1%M2;
    s->ix1 = (IA1*s->ix1 + IC1)%M1;
    s->ix3 = s->ix1%M3;
    for (int j = 0;  j < 97;  j++)
    {
        s->ix1 = (IA1*s->ix1 + IC1)%M1;
        s->ix2 = (IA2*s->ix2 + IC2)%M2;
        s->r[j] = (s->ix1 + s->ix2*RM2)*RM1;
    }
    /* awgn_init_dbm0 -> awgn_init_dbov(level - DBM0_MAX_POWER): the subtraction is binary32 */
    level -= (3.14f + 3.02f);
    s->rms = pow(10.0, level/20.0)*32768.0;
    s->amp2 = 0.0;
    s->odd = 1;
}

static double uniform(orc_awgn_t *s)
{
    double t;
    int j;

    s->ix1 = (IA1*s->ix1 + IC1)%M1;
    s->ix2 = (IA2*s->ix2 + IC2)%M2;
    s->ix3 = (IA3*s->ix3 + IC3)%M3;
    j = (97*s->ix3)/M3;
    if (j > 96  ||  j < 0)
        return -1.0;
    t = s->r[j];
    s->r[j] = (s->ix1 + s->ix2*RM2)*RM1;
    return t;
}

ORC_API int16_t orc_awgn(orc_awgn_t *s)
{
    double amp;

    s->odd = !s->odd;
    if (s->odd)
    {
        amp = s->amp2;
    }
    else
    {
        double v1;
        double v2;
        double r;

        do
        {
            v1 = 2.0*uniform(s) - 1.0;
            v2 = 2.0*uniform(s) - 1.0;
            r = v1*v1 + v2*v2;
        }
        while (r >= 1.0);
        r = sqrt(-2.0*orc_glibc_log(r)/r);
        s->amp2 = v1*r;
        amp = v2*r;
    }
    amp *= s->rms;
    if (amp > 32767.0)
        return 32767;
    if (amp < -32768.0)
        return -32768;
    return (int16_t) lrint(amp);
}

ORC_API void orc_awgn_block(orc_awgn_t *s, int16_t out[], int n)
{
    for (int i = 0;  i < n;  i++)
        out[i] = orc_awgn(s);
}

ORC_API int orc_awgn_sizeof(void)
{
    return (int) sizeof(orc_awgn_t);
}
