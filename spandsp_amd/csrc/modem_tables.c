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
#include <string.h>

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

/* ---- V.17 / V.32bis signal space ------------------------------------------------------------
 * The trellis coded constellations of ITU-T V.17 (figures 2 to 5) have four-fold rotational symmetry: within each
 * group of eight points that share the uncoded bits Q, the three coded bits Y select one of two seed points turned
 * by 0, 90, 180 or 270 degrees.  Seeds per group: {re, im} for Y = 0 and for Y = 1; the turns of the first seed sit at
 * Y = 0, 7, 4, 3 and those of the second at Y = 1, 6, 5, 2.  (Index order as the reference's
 * v17_v32bis_tx_constellation_maps.h, which tests/test_modem_tables.py checks this against.) */
static const int8_t v17_seeds_14400[16][4] =
{
    {-8, -3, 9, 2}, {-8, 1, 9, -2}, {-4, -3, 5, 2}, {-4, 1, 5, -2}, {4, -3, -3, 2}, {4, 1, -3, -2}, {0, -3, 1, 2}, {0, 1, 1, -2},
    {8, -3, -7, 2}, {8, 1, -7, -2}, {-4, -7, 5, 6}, {-4, 5, 5, -6}, {4, -7, -3, 6}, {4, 5, -3, -6}, {0, -7, 1, 6}, {0, 5, 1, -6}
};
static const int8_t v17_seeds_12000[8][4] =
{
    {7, 1, -5, -1}, {3, -3, -1, 3}, {7, -7, -5, 7}, {-1, -7, 3, 7}, {3, 5, -1, -5}, {-1, 1, 3, -1}, {-5, 5, 7, -5}, {-5, -3, 7, 3}
};
static const int8_t v17_seeds_9600[4][4] = {{-8, 2, -6, -4}, {0, 2, -6, 4}, {0, -6, 2, -4}, {8, 2, 2, 4}};
static const int8_t v17_seeds_7200[2][4] = {{6, -6, -2, 6}, {-2, 2, 6, -2}};

int spg_v17_constellation_size(int bit_rate)
{
    switch (bit_rate)
    {
    case 14400: return 128;
    case 12000: return 64;
    case 9600: return 32;
    case 7200: return 16;
    case 4800: return 4;
    }
    return -1;
}

/* out[n][2] = {re, im}; returns n */
int spg_make_v17_constellation(int bit_rate, int8_t out[][2])
{
    static const int turn_a[4] = {0, 7, 4, 3};
    static const int turn_b[4] = {1, 6, 5, 2};
    const int8_t (*seeds)[4];
    int n = spg_v17_constellation_size(bit_rate);
    int g;
    int k;
    int re;
    int im;
    int t;

    switch (bit_rate)
    {
    case 14400: seeds = v17_seeds_14400; break;
    case 12000: seeds = v17_seeds_12000; break;
    case 9600: seeds = v17_seeds_9600; break;
    case 7200: seeds = v17_seeds_7200; break;
    case 4800:
        /* V.32bis 4800 bps: the four points of the V.17 training constellation, v17rx.c:1335 */
        out[0][0] = -6; out[0][1] = -2;
        out[1][0] = -2; out[1][1] = 6;
        out[2][0] = 2; out[2][1] = -6;
        out[3][0] = 6; out[3][1] = 2;
        return 4;
    default:
        return -1;
    }
    for (g = 0;  g < n/8;  g++)
    {
        re = seeds[g][0];
        im = seeds[g][1];
        for (k = 0;  k < 4;  k++)
        {
            out[8*g + turn_a[k]][0] = (int8_t) re;
            out[8*g + turn_a[k]][1] = (int8_t) im;
            t = re;  re = -im;  im = t;             /* +90 degrees */
        }
        re = seeds[g][2];
        im = seeds[g][3];
        for (k = 0;  k < 4;  k++)
        {
            out[8*g + turn_b[k]][0] = (int8_t) re;
            out[8*g + turn_b[k]][1] = (int8_t) im;
            t = re;  re = -im;  im = t;
        }
    }
    return n;
}

/* The receiver's soft decision maps (v17rx.c:440-470 indexes constel_maps[space][re][im][i] with re, im = the half-unit
   cell of the plane [-9, 9) x [-9, 9) and i = the trellis subset): for each cell and each subset, the constellation
   point of that subset (index & 7 == i) nearest to the cell.  The nearest point is found from the cell centre nudged
   towards (-inf, -inf), taking the higher numbered point on a remaining tie -- except in the cells listed below, where
   two points are exactly equidistant along the diagonal and the reference's table holds the lower numbered one
   (packed space << 15 | re << 9 | im << 3 | i).  tests/test_modem_tables.py checks the result against the reference's
   table (CRC-32, and entry by entry when the reference build is present). */
static const uint32_t v17_map_lower_on_tie[100] =
{
    0x03913, 0x03916, 0x03B1B, 0x03B1E, 0x04112, 0x04117, 0x0431A, 0x0431F, 0x044E0, 0x044E1, 0x04504, 0x04505, 0x046E8,
    0x046E9, 0x0470C, 0x0470D, 0x09087, 0x0928F, 0x09472, 0x09474, 0x09497, 0x0967A, 0x0967C, 0x0969F, 0x09882, 0x09884,
    0x098A7, 0x09A8A, 0x09A8C, 0x09AAF, 0x09C33, 0x09C50, 0x09C56, 0x09CB3, 0x09CD0, 0x09CD6, 0x09CF1, 0x09E3B, 0x09E58,
    0x09E5E, 0x09EBB, 0x09ED8, 0x09EDE, 0x09EF9, 0x0A043, 0x0A060, 0x0A066, 0x0A0C3, 0x0A0E6, 0x0A24B, 0x0A268, 0x0A26E,
    0x0A2CB, 0x0A2EE, 0x0A453, 0x0A4D3, 0x0A4F6, 0x0A65B, 0x0A6DB, 0x0A6FE, 0x0A863, 0x0AA6B, 0x0AC77, 0x0AE7F, 0x0B087,
    0x0B28F, 0x0B472, 0x0B474, 0x0B497, 0x0B67A, 0x0B67C, 0x0B69F, 0x0B882, 0x0B8A7, 0x0BA8A, 0x0BAAF, 0x0BC71, 0x0BC92,
    0x0BD14, 0x0BE79, 0x0BE9A, 0x0BF1C, 0x0C4F0, 0x0C6F8, 0x10803, 0x10A0B, 0x144F4, 0x146FC, 0x18C74, 0x18E7C, 0x1969C,
    0x198A4, 0x19C30, 0x19E38, 0x1A658, 0x1A860, 0x1B514, 0x1B71C, 0x1C4D0, 0x1C6D8
};

void spg_make_v17_rx_maps(uint8_t maps[4*36*36*8], uint8_t map_4800[36*36])
{
    static const int rates[4] = {14400, 12000, 9600, 7200};
    int8_t pts[128][2];
    int space;
    int n;
    int re;
    int im;
    int i;
    int k;
    int best;
    int second;
    long long x;
    long long y;
    long long d;
    long long dmin;
    unsigned e;

    for (space = 0;  space < 4;  space++)
    {
        n = spg_make_v17_constellation(rates[space], pts);
        for (re = 0;  re < 36;  re++)
        {
            for (im = 0;  im < 36;  im++)
            {
                /* units of 1/4000: cell centre = re/2 - 9 + 1/4, nudged by -1/4000 in both axes */
                x = re*2000 - 36000 + 1000 - 1;
                y = im*2000 - 36000 + 1000 - 1;
                for (i = 0;  i < 8;  i++)
                {
                    best = -1;
                    second = -1;
                    dmin = 0;
                    for (k = i;  k < n;  k += 8)
                    {
                        d = (pts[k][0]*4000LL - x)*(pts[k][0]*4000LL - x) + (pts[k][1]*4000LL - y)*(pts[k][1]*4000LL - y);
                        if (best < 0  ||  d < dmin)
                        {
                            dmin = d;
                            best = k;
                            second = -1;
                        }
                        else if (d == dmin)
                        {
                            second = best;
                            best = k;
                        }
                    }
                    maps[((space*36 + re)*36 + im)*8 + i] = (uint8_t) best;
                    (void) second;
                }
            }
        }
    }
    /* the diagonal ties the reference resolves the other way */
    for (e = 0;  e < sizeof(v17_map_lower_on_tie)/sizeof(v17_map_lower_on_tie[0]);  e++)
    {
        const uint32_t v = v17_map_lower_on_tie[e];
        space = (int) (v >> 15);
        re = (int) ((v >> 9) & 0x3F);
        im = (int) ((v >> 3) & 0x3F);
        i = (int) (v & 7);
        n = spg_make_v17_constellation(rates[space], pts);
        x = re*2000 - 36000 + 1000 - 1;
        y = im*2000 - 36000 + 1000 - 1;
        best = -1;
        dmin = 0;
        for (k = i;  k < n;  k += 8)
        {
            d = (pts[k][0]*4000LL - x)*(pts[k][0]*4000LL - x) + (pts[k][1]*4000LL - y)*(pts[k][1]*4000LL - y);
            if (best < 0  ||  d < dmin)
            {
                dmin = d;
                best = k;
            }
        }
        maps[((space*36 + re)*36 + im)*8 + i] = (uint8_t) best;
    }
    /* 4800 bps, no trellis: plain nearest of the four points to the cell centre (no ties occur) */
    spg_make_v17_constellation(4800, pts);
    for (re = 0;  re < 36;  re++)
    {
        for (im = 0;  im < 36;  im++)
        {
            x = re*2000 - 36000 + 1000;
            y = im*2000 - 36000 + 1000;
            best = 0;
            dmin = -1;
            for (k = 0;  k < 4;  k++)
            {
                d = (pts[k][0]*4000LL - x)*(pts[k][0]*4000LL - x) + (pts[k][1]*4000LL - y)*(pts[k][1]*4000LL - y);
                if (dmin < 0  ||  d < dmin)
                {
                    dmin = d;
                    best = k;
                }
            }
            map_4800[re*36 + im] = (uint8_t) best;
        }
    }
}


/* The transmit pulse shaper of make_modem_filter.c:52-152 (make_tx_filter): a root raised cosine at baseband,
   coeff_sets interpolation phases of coeffs_per_filter taps, unity DC gain, printed with ten decimals. */
int spg_make_tx_pulseshaper(int coeff_sets, int coeffs_per_filter, double excess_bandwidth, float *out)
{
    double *coeffs;
    double gain;
    int total;
    int i;
    int j;

    total = coeff_sets*coeffs_per_filter + 1;
    if ((coeffs = (double *) malloc(sizeof(double)*total)) == NULL)
        return -1;
    /* alpha = baud_rate/(2*coeff_sets*baud_rate) */
    root_raised_cosine(coeffs, total, 1.0/(2.0*(double) coeff_sets), excess_bandwidth);
    gain = 0.0;
    for (i = coeff_sets/2;  i < total;  i += coeff_sets)
        gain += coeffs[i];
    for (i = 0;  i < total;  i++)
        coeffs[i] /= gain;
    for (j = 0;  j < coeff_sets;  j++)
    {
        for (i = 0;  i < coeffs_per_filter;  i++)
            out[j*coeffs_per_filter + i] = via_text(coeffs[i*coeff_sets + j], 10);
    }
    free(coeffs);
    return 0;
}

/* ---- tone generator descriptors ---------------------------------------------------------------- */
int32_t spg_dds_phase_ratef(float hz)
{
    /* dds_phase_ratef(), dds_float.c:2109-2112: binary32 throughout */
    return (int32_t) (hz*65536.0f*65536.0f/8000);
}

float spg_db_to_amplitude_ratio(float db)
{
    /* db_to_amplitude_ratio(), telephony.h:141 */
    return powf(10.0f, db/20.0f);
}

float spg_dds_scaling_dbm0f(float level)
{
    /* dds_scaling_dbm0f(), dds_float.c:2121-2124; DBM0_MAX_SINE_POWER = 3.14f */
    return powf(10.0f, (level - 3.14f)/20.0f)*32767.0f;
}

void spg_make_tone_descriptor(int32_t out[13], int f1, int l1, int f2, int l2, int d1, int d2, int d3, int d4, int repeat)
{
    float g[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int i;

    for (i = 0;  i < 13;  i++)
        out[i] = 0;
    if (f1)
    {
        out[0] = spg_dds_phase_ratef((float) f1);
        if (f2 < 0)
            out[0] = -out[0];
        g[0] = spg_dds_scaling_dbm0f((float) l1);
    }
    if (f2)
    {
        out[1] = spg_dds_phase_ratef((float) abs(f2));
        g[1] = (f2 < 0)  ?  (float) l2/100.0f  :  spg_dds_scaling_dbm0f((float) l2);
    }
    memcpy(&out[4], g, sizeof(g));
    out[8] = d1*8000/1000;
    out[9] = d2*8000/1000;
    out[10] = d3*8000/1000;
    out[11] = d4*8000/1000;
    out[12] = repeat;
}
