/*
 * modem_tables.c -- host-side construction of the constant tables the modem receivers use.
 * The reference gets them from build-time generator programs and from a literal table;
 * here they are computed once at library load, following the same recipes so that every
 * entry is bit-identical (tests/test_modem_tables.py compares against the reference build):
 *
 *   sine table        src/dds_float.c:51-2101: sin(2*pi*i/2048) written with 8 decimals
 *   fixed_sqrt_table  src/make_math_fixed_tables.c (193 entries, sqrt(i/256)*65536 + 0.5)
 *   RX pulse shaper   src/make_modem_filter.c:158-268 (make_rx_filter) over
 *                     src/filter_tools.c:122-181 (compute_raised_cosine_filter): a root raised
 *                     cosine designed by frequency sampling on 8192 points, normalised to unity
 *                     DC gain per polyphase set, modulated to the carrier, and printed with
 *                     "%15.10f" -- the decimal rounding is part of the recipe, because the
 *                     receivers use what the C compiler parsed back from that text.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "modem_tables.h"

#define SEQ_LEN     8192                    /* filter_tools.c:57 */

/* Round through the decimal text the generator prints, then to float as a C compiler would. */
static float via_text(double v, int decimals)
{
    char buf[64];

    snprintf(buf, sizeof(buf), "%.*f", decimals, v);
    return strtof(buf, NULL);
}

void spg_make_sine_table(float out[SPG_SINE_LEN])
{
    int i;

    for (i = 0;  i < SPG_SINE_LEN;  i++)
        out[i] = via_text(sin(2.0*3.14159265358979323846*(double) i/(double) SPG_SINE_LEN), 8);
}

void spg_make_sqrt_table(uint16_t out[193])
{
    int i;
    int ival;

    for (i = 64;  i <= 256;  i++)
    {
        ival = (int) (sqrt(i/256.0)*65536.0 + 0.5);
        if (ival > 65535)
            ival = 65535;
        out[i - 64] = (uint16_t) ival;
    }
}

/* Radix-2 decimation-in-time transform with kernel exp(+j*theta) over SEQ_LEN points, using the
   generator's own twiddle table (filter_tools.c:107-119: angles built from the truncated pi
   3.1415926535, which is visible at the 10th decimal of the coefficients, so it is kept). */
typedef struct
{
    double re;
    double im;
} cplx_t;

static cplx_t twiddle[SEQ_LEN/2];

static void dit(cplx_t *data, cplx_t *tmp, int n)
{
    int half = n/2;
    int stride = SEQ_LEN/n;
    int i;
    cplx_t w;
    cplx_t t;

    if (n <= 1)
        return;
    for (i = 0;  i < half;  i++)
    {
        tmp[i] = data[2*i];
        tmp[half + i] = data[2*i + 1];
    }
    dit(tmp, data, half);
    dit(tmp + half, data + half, half);
    for (i = 0;  i < half;  i++)
    {
        w = twiddle[i*stride];
        t.re = w.re*tmp[half + i].re - w.im*tmp[half + i].im;
        t.im = w.re*tmp[half + i].im + w.im*tmp[half + i].re;
        data[i].re = tmp[i].re + t.re;
        data[i].im = tmp[i].im + t.im;
        data[half + i].re = tmp[i].re - t.re;
        data[half + i].im = tmp[i].im - t.im;
    }
}

/* filter_tools.c:122-181 with root = true, sinc_compensate = false */
static void root_raised_cosine(double coeffs[], int len, double alpha, double beta)
{
    static cplx_t vec[SEQ_LEN];
    static cplx_t tmp[SEQ_LEN];
    double f;
    double f1;
    double f2;
    double tau;
    double x;
    int i;
    int h;

    f1 = (1.0 - beta)*alpha;
    f2 = (1.0 + beta)*alpha;
    tau = 0.5/alpha;
    for (i = 0;  i <= SEQ_LEN/2;  i++)
    {
        f = (double) i/(double) SEQ_LEN;
        if (f <= f1)
            x = 1.0;
        else if (f <= f2)
            x = 0.5*(1.0 + cos((3.1415926535*tau/beta)*(f - f1)));
        else
            x = 0.0;
        vec[i].re = sqrt(x)*tau;
        vec[i].im = 0.0;
    }
    for (i = 1;  i < SEQ_LEN/2;  i++)
        vec[SEQ_LEN - i] = vec[i];
    for (i = 0;  i < SEQ_LEN/2;  i++)
    {
        x = (2.0*3.1415926535*i)/(double) SEQ_LEN;
        twiddle[i].re = cos(x);
        twiddle[i].im = sin(x);
    }
    dit(vec, tmp, SEQ_LEN);
    h = (len - 1)/2;
    for (i = 0;  i < len;  i++)
        coeffs[i] = vec[(SEQ_LEN - h + i)%SEQ_LEN].re/(double) SEQ_LEN;
}

/* make_modem_filter.c:158-268 */
int spg_make_rx_pulseshaper(int coeff_sets, int coeffs_per_filter, double carrier_hz, double baud_rate,
                            double excess_bandwidth, float *re, float *im)
{
    double *coeffs;
    double alpha;
    double gain;
    double carrier;
    int total;
    int i;
    int j;
    int m;
    int x;

    total = coeff_sets*coeffs_per_filter + 1;
    if ((coeffs = (double *) malloc(sizeof(double)*total)) == NULL)
        return -1;
    alpha = baud_rate/(2.0*(double) (coeff_sets*8000.0));
    carrier = carrier_hz*(2.0*3.1415926535/8000.0);
    root_raised_cosine(coeffs, total, alpha, excess_bandwidth);
    gain = 0.0;
    for (i = coeff_sets/2;  i < total;  i += coeff_sets)
        gain += coeffs[i];
    for (i = 0;  i < total;  i++)
        coeffs[i] /= gain;
    for (j = 0;  j < coeff_sets;  j++)
    {
        for (i = 0;  i < coeffs_per_filter;  i++)
        {
            m = i - (coeffs_per_filter >> 1);
            x = i*coeff_sets + j;
            re[j*coeffs_per_filter + i] = via_text(coeffs[x]*cos(carrier*m), 10);
            im[j*coeffs_per_filter + i] = via_text(coeffs[x]*sin(carrier*m), 10);
        }
    }
    free(coeffs);
    return 0;
}

/* make_modem_godard_descriptor.c:59-78, printed with "%10.6f" (:186-196) */
void spg_make_godard(double carrier, double baud_rate, double alpha, float out[7])
{
    const double pi = 3.14159265358979323846;
    double low_edge = 2.0*pi*(carrier - baud_rate/2.0)/8000.0;
    double high_edge = 2.0*pi*(carrier + baud_rate/2.0)/8000.0;

    out[0] = via_text(2.0*alpha*cos(low_edge), 6);
    out[1] = via_text(-alpha*alpha, 6);
    out[2] = via_text(-alpha*sin(low_edge), 6);
    out[3] = via_text(2.0*alpha*cos(high_edge), 6);
    out[4] = via_text(-alpha*alpha, 6);
    out[5] = via_text(-alpha*sin(high_edge), 6);
    out[6] = via_text(-alpha*alpha*(sin(high_edge)*cos(low_edge) - sin(low_edge)*cos(high_edge)), 6);
}

/* The V.29 9600 bps decision regions (v29rx.c:119-143): for each half-unit cell of the
   [-5, 5) x [-5, 5) plane, the index of the constellation point it decodes to.  One row of
   the 20 x 20 map per string, value = character - 'a'. */
void spg_make_v29_space_map(uint8_t out[400])
{
    static const char *rows[20] =
    {
        "nnnnnnmmmmmmmmllllll", "nnnnnnnmmmmmmlllllll", "nnnnnnneeeeeelllllll", "nnnnnnneeeeeelllllll",
        "nnnnnnneeeeeelllllll", "nnnnnnnfeeeedlllllll", "onnnnnffffddddlllllk", "oogggfffffdddddccckk",
        "ooggggffffddddcccckk", "ooggggffffddddcccckk", "oogggghhhhbbbbcccckk", "oogggghhhhbbbbcccckk",
        "ooggghhhhhbbbbbccckk", "oppppphhhhbbbbjjjjjk", "ppppppphaaaabjjjjjjj", "pppppppaaaaaajjjjjjj",
        "pppppppaaaaaajjjjjjj", "pppppppaaaaaajjjjjjj", "pppppppiiiiiijjjjjjj", "ppppppiiiiiiiijjjjjj"
    };
    int i;
    int j;

    for (i = 0;  i < 20;  i++)
    {
        for (j = 0;  j < 20;  j++)
            out[i*20 + j] = (uint8_t) (rows[i][j] - 'a');
    }
}
