// modemtx_dev.hpp -- device side of the modem transmitter banks: N V.29 or V.27ter
// modulators, one channel per lane, as the signal sources of the receiver banks.
//
// What is restated (paths relative to the reference tree; float build, x86-64):
//   v29_tx()                   src/v29tx.c:226-284
//   getbaud(), get_scrambled_bit()   src/v29tx.c:103-224   (training segments, data scrambler, differential phase map)
//   v27ter_tx()                src/v27ter_tx.c:246-350   (1600 baud at 4800 bps, 1200 baud at 2400 bps)
//   getbaud(), scramble()      src/v27ter_tx.c:103-244   (the scrambler with its guard against repeating patterns)
//   vec_circular_dot_prodf()   src/vector_float.c (scalar path): two partial sums, split where the ring wraps
//   dds_complexf()             src/dds_float.c:2184-2191
//   lfastrintf()               a truncating cast on x86-64 (spandsp/fast_convert.h:184-197)
//
// Data bits come from a per-channel 15 bit LFSR (x^15 + x^14 + 1, the source the test harness of the oracle feeds
// the reference with); a caller's get_bit() callback, and with it the end-of-data shutdown, is not replayed.
//
// The nine-symbol pulse shaping buffer is kept oldest-first in registers (a shift per baud); the ring position
// of the reference only decides where its dot product splits into two partial sums, and that split is
// reproduced per lane with selects, so the float sums are formed in the reference's order.

#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spg
{

enum
{
    VT_BIT_RATE = 0,
    VT_BASE_GAIN = 1,       // float (V.27ter: gain_2400)
    VT_GAIN = 2,            // float (V.27ter: gain_4800)
    VT_RRC_RE = 3,          // 9 floats, ring order (as the reference keeps them)
    VT_RRC_IM = 12,         // 9 floats
    VT_RRC_STEP = 21,
    VT_SCRAMBLE = 22,
    VT_TRAIN_SCRAMBLE = 23, // V.27ter: scrambler_pattern_count
    VT_IN_TRAINING = 24,
    VT_TRAINING_STEP = 25,
    VT_TRAINING_OFFSET = 26,
    VT_CARRIER_PHASE = 27,
    VT_CARRIER_RATE = 28,
    VT_BAUD_PHASE = 29,
    VT_CONSTELLATION = 30,
    VT_PRBS = 31,
    kV29TxWords = 32
};

constexpr int kVtSeg1 = 480;
constexpr int kVtSeg2 = kVtSeg1 + 48;
constexpr int kVtSeg3 = kVtSeg2 + 128;
constexpr int kVtSeg4 = kVtSeg3 + 384;
constexpr int kVtEnd = kVtSeg4 + 48;
constexpr int kVtShutdownEnd = kVtEnd + 32;

constexpr int kTxV29 = 0;
constexpr int kTxV27ter = 1;

// V.27ter training, in symbols (v27ter_tx.c:82-96)
constexpr int kV27Seg2 = 320;
constexpr int kV27Seg3 = kV27Seg2 + 32;
constexpr int kV27Seg4 = kV27Seg3 + 50;
constexpr int kV27Seg5 = kV27Seg4 + 1074;
constexpr int kV27End = kV27Seg5 + 8;
constexpr int kV27ShutdownEnd = kV27End + 32;

struct V29TxLaunch
{
    int32_t *st;                // [kV29TxWords][n_ch]
    const float *sine;          // [2048]
    const float *shaper;        // V.29: [10][9]; V.27ter: [5][9] (4800 bps) then [20][9] (2400 bps)
    int16_t *pcm;               // [n_ch][stride]
    long long stride;
    int n_ch;
    int samples;
    int vec;
};

// v29tx_constellation_maps.h: index = amplitude bit << 3 | phase octant
__device__ __forceinline__ void v29tx_point(int idx, float &re, float &im)
{
    const int oct = idx & 7;
    const bool diag = (oct & 1) != 0;
    const float r = (idx & 8)  ?  (diag  ?  3.0f  :  5.0f)  :  (diag  ?  1.0f  :  3.0f);
    // cos / sin signs of the octant
    const float cx = (oct == 2  ||  oct == 6)  ?  0.0f  :  ((oct >= 3  &&  oct <= 5)  ?  -1.0f  :  1.0f);
    const float cy = (oct == 0  ||  oct == 4)  ?  0.0f  :  ((oct >= 5)  ?  -1.0f  :  1.0f);
    re = r*cx;
    im = r*cy;
}

// v27ter_tx.c:163-173: eight phases, 1.414 on the axes
__device__ __forceinline__ void v27tx_point(int oct, float &re, float &im)
{
    const bool diag = (oct & 1) != 0;
    const float r = diag  ?  1.0f  :  1.414f;
    const float cx = (oct == 2  ||  oct == 6)  ?  0.0f  :  ((oct >= 3  &&  oct <= 5)  ?  -1.0f  :  1.0f);
    const float cy = (oct == 0  ||  oct == 4)  ?  0.0f  :  ((oct >= 5)  ?  -1.0f  :  1.0f);
    re = r*cx;
    im = r*cy;
}

template <int KIND>
__global__ __launch_bounds__(64) void modemtx_bank_kernel(const V29TxLaunch L)
{
    __shared__ float sine[2048];
    __shared__ float shaper[25][9];
    const int lane = threadIdx.x;
    const int ch = blockIdx.x*64 + lane;

    for (int i = lane;  i < 2048;  i += 64)
        sine[i] = L.sine[i];
    for (int i = lane;  i < ((KIND == kTxV29)  ?  90  :  225);  i += 64)
        (&shaper[0][0])[i] = L.shaper[i];
    __syncthreads();
    if (ch >= L.n_ch)
        return;

    int32_t *st = L.st + ch;
    const size_t n = (size_t) L.n_ch;
    const int bit_rate = st[VT_BIT_RATE*n];
    // V.27ter keeps one gain per rate: word 1 for 2400 bps, word 2 for 4800 bps
    const float gain = __int_as_float(st[((KIND == kTxV27ter  &&  bit_rate == 2400)  ?  VT_BASE_GAIN  :  VT_GAIN)*n]);
    int rrc_step = st[VT_RRC_STEP*n];
    uint32_t scramble = (uint32_t) st[VT_SCRAMBLE*n];
    uint32_t train_scramble = (uint32_t) st[VT_TRAIN_SCRAMBLE*n];
    int in_training = st[VT_IN_TRAINING*n];
    int training_step = st[VT_TRAINING_STEP*n];
    const int training_offset = st[VT_TRAINING_OFFSET*n];
    uint32_t carrier_phase = (uint32_t) st[VT_CARRIER_PHASE*n];
    const uint32_t carrier_rate = (uint32_t) st[VT_CARRIER_RATE*n];
    int baud_phase = st[VT_BAUD_PHASE*n];
    int constellation = st[VT_CONSTELLATION*n];
    uint32_t prbs = (uint32_t) st[VT_PRBS*n];
    // ring -> oldest first: aged[i] = ring[(rrc_step + i) mod 9]
    float are[9];
    float aim[9];
#pragma unroll
    for (int i = 0;  i < 9;  i++)
    {
        int at = rrc_step + i;
        at = (at >= 9)  ?  (at - 9)  :  at;
        are[i] = __int_as_float(st[(size_t) (VT_RRC_RE + at)*n]);
        aim[i] = __int_as_float(st[(size_t) (VT_RRC_IM + at)*n]);
    }

    auto scrambled_bit = [&]() -> int
    {
        // get_scrambled_bit(), v29tx.c:103-123 / v27ter_tx.c:127-146; while training the source is fake_get_bit() = 1
        int bit = 1;
        if (!in_training)
        {
            bit = (int) (((prbs >> 14) ^ (prbs >> 13)) & 1u);
            prbs = ((prbs << 1) | (uint32_t) bit) & 0x7FFFu;
        }
        if (KIND == kTxV29)
        {
            const int out = (int) (((uint32_t) bit ^ (scramble >> 17) ^ (scramble >> 22)) & 1u);
            scramble = (scramble << 1) | (uint32_t) out;
            return out;
        }
        // scramble(), v27ter_tx.c:103-125: 1 + x^-6 + x^-7, inverted after 33 bits without the guard pattern
        uint32_t out = ((uint32_t) bit ^ (scramble >> 5) ^ (scramble >> 6)) & 1u;
        if ((int) train_scramble >= 33)
        {
            out ^= 1u;
            train_scramble = 0;
        }
        else if ((((scramble >> 7) ^ out) & ((scramble >> 8) ^ out) & ((scramble >> 11) ^ out) & 1u))
        {
            train_scramble = 0;
        }
        else
        {
            train_scramble++;
        }
        scramble = (scramble << 1) | out;
        return (int) out;
    };

    int16_t *row = L.pcm + (size_t) ch*L.stride;
    const bool silent = (training_step >= ((KIND == kTxV29)  ?  kVtShutdownEnd  :  kV27ShutdownEnd));    // nothing more is sent
    for (int base = 0;  base < L.samples;  base += 8)
    {
        uint32_t pk[4] = {0u, 0u, 0u, 0u};
        const int todo = (L.samples - base < 8)  ?  (L.samples - base)  :  8;
#pragma unroll
        for (int j = 0;  j < 8;  j++)
        {
            int v = 0;
            if (j < todo  &&  !silent)
            {
                bool fresh;
                if (KIND == kTxV29)
                {
                    baud_phase += 3;
                    fresh = (baud_phase >= 10);
                    baud_phase -= fresh  ?  10  :  0;
                }
                else if (bit_rate == 4800)
                {
                    baud_phase += 1;
                    fresh = (baud_phase >= 5);
                    baud_phase -= fresh  ?  5  :  0;
                }
                else
                {
                    baud_phase += 3;
                    fresh = (baud_phase >= 20);
                    baud_phase -= fresh  ?  20  :  0;
                }
                if (fresh)
                {
                    float vre = 0.0f;
                    float vim = 0.0f;
                    if (KIND == kTxV29)
                    {
                        // getbaud(), v29tx.c:126-224
                        bool have = false;
                        if (in_training)
                        {
                            training_step++;
                            if (training_step <= kVtSeg4)
                            {
                                have = true;
                                if (training_step <= kVtSeg1)
                                {
                                    v29tx_point(0, vre, vim);               // TEP: unmodulated carrier
                                }
                                else if (training_step <= kVtSeg2)
                                {
                                    vre = 0.0f;                             // silence
                                    vim = 0.0f;
                                }
                                else if (training_step <= kVtSeg3)
                                {
                                    // ABAB: A = 315 deg high / 315 deg low / 270 deg low by rate, B = 180 deg low
                                    const int a = (training_offset == 0)  ?  15  :  ((training_offset == 2)  ?  7  :  6);
                                    v29tx_point((training_step & 1)  ?  4  :  a, vre, vim);
                                }
                                else
                                {
                                    // CDCD through the 1 + x^-6 + x^-7 training scrambler
                                    const int d = (training_offset == 0)  ?  11  :  ((training_offset == 2)  ?  3  :  2);
                                    const uint32_t bit = train_scramble & 1u;
                                    train_scramble >>= 1;
                                    train_scramble |= ((bit ^ train_scramble) & 1u) << 6;
                                    train_scramble &= 0xFFu;
                                    v29tx_point(bit  ?  d  :  0, vre, vim);
                                }
                            }
                            else if (training_step == kVtEnd + 1)
                            {
                                in_training = 0;
                            }
                        }
                        if (!have)
                        {
                            int amp = 0;
                            if (bit_rate == 9600)
                                amp = scrambled_bit()  ?  8  :  0;
                            int bits = scrambled_bit();
                            bits = (bits << 1) | scrambled_bit();
                            if (bit_rate == 4800)
                            {
                                bits = (0x4620 >> (bits*4)) & 7;                            // {0, 2, 6, 4}
                            }
                            else
                            {
                                bits = (bits << 1) | scrambled_bit();
                                bits = (int) ((0x45763201u >> (bits*4)) & 7u);              // {1, 0, 2, 3, 6, 7, 5, 4}
                            }
                            constellation = (constellation + bits) & 7;
                            v29tx_point(amp | constellation, vre, vim);
                        }
                    }
                    else
                    {
                        // getbaud(), v27ter_tx.c:148-244
                        bool have = false;
                        if (in_training)
                        {
                            training_step++;
                            if (training_step <= kV27Seg5)
                            {
                                have = true;
                                if (training_step <= kV27Seg2)
                                {
                                    v27tx_point(0, vre, vim);           // unmodulated carrier (TEP)
                                }
                                else if (training_step <= kV27Seg3)
                                {
                                    vre = 0.0f;                         // silence
                                    vim = 0.0f;
                                }
                                else if (training_step <= kV27Seg4)
                                {
                                    constellation = (constellation + 4) & 7;    // regular reversals
                                    v27tx_point(constellation, vre, vim);
                                }
                                else
                                {
                                    // scrambled reversals: every third bit of the scrambler
                                    const int bits = scrambled_bit() << 2;
                                    (void) scrambled_bit();
                                    (void) scrambled_bit();
                                    constellation = (constellation + bits) & 7;
                                    v27tx_point(constellation, vre, vim);
                                }
                            }
                            else if (training_step == kV27End + 1)
                            {
                                in_training = 0;
                            }
                        }
                        if (!have)
                        {
                            int bits = scrambled_bit();
                            bits = (bits << 1) | scrambled_bit();
                            if (bit_rate == 4800)
                            {
                                bits = (bits << 1) | scrambled_bit();
                                bits = (int) ((0x45763201u >> (bits*4)) & 7u);          // {1, 0, 2, 3, 6, 7, 5, 4}
                            }
                            else
                            {
                                bits = (0x4620 >> (bits*4)) & 7;                        // {0, 2, 6, 4}
                            }
                            constellation = (constellation + bits) & 7;
                            v27tx_point(constellation, vre, vim);
                        }
                    }
                    // the new symbol overwrites the oldest: shift, newest last
#pragma unroll
                    for (int i = 0;  i < 8;  i++)
                    {
                        are[i] = are[i + 1];
                        aim[i] = aim[i + 1];
                    }
                    are[8] = vre;
                    aim[8] = vim;
                    rrc_step = (rrc_step + 1 >= 9)  ?  0  :  (rrc_step + 1);
                }
                // vec_circular_dot_prodf(ring, shaper[9 - baud_phase], 9, rrc_step): the first partial sum runs over the
                // 9 - rrc_step oldest entries, the second over the rest; each from 0.0f, then added
                const float *coef = (KIND == kTxV29)  ?  shaper[9 - baud_phase]
                                                      :  ((bit_rate == 4800)  ?  shaper[4 - baud_phase]  :  shaper[5 + 19 - baud_phase]);
                const int split = 9 - rrc_step;
                float zre = 0.0f;
                float zim = 0.0f;
                float z1re = 0.0f;
                float z1im = 0.0f;
#pragma unroll
                for (int i = 0;  i < 9;  i++)
                {
                    const float c = coef[i];
                    const float pre = __fmul_rn(are[i], c);
                    const float pim = __fmul_rn(aim[i], c);
                    const bool first = (i < split);
                    zre = first  ?  __fadd_rn(zre, pre)  :  zre;
                    zim = first  ?  __fadd_rn(zim, pim)  :  zim;
                    z1re = first  ?  z1re  :  __fadd_rn(z1re, pre);
                    z1im = first  ?  z1im  :  __fadd_rn(z1im, pim);
                }
                const float xre = __fadd_rn(zre, z1re);
                const float xim = __fadd_rn(zim, z1im);
                const float cre = sine[(uint32_t) (carrier_phase + (1u << 30)) >> 21];
                const float cim = sine[carrier_phase >> 21];
                carrier_phase += carrier_rate;
                const float famp = __fsub_rn(__fmul_rn(xre, cre), __fmul_rn(xim, cim));
                v = (int) __fmul_rn(famp, gain);
            }
            pk[j >> 1] |= ((uint32_t) v & 0xFFFFu) << ((j & 1)*16);
        }
        if (L.vec  &&  todo == 8)
        {
            *reinterpret_cast<uint4 *>(row + base) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
        else
        {
            for (int j = 0;  j < todo;  j++)
                row[base + j] = (int16_t) (pk[j >> 1] >> ((j & 1)*16));
        }
    }

    // oldest-first -> ring
#pragma unroll
    for (int i = 0;  i < 9;  i++)
    {
        int at = rrc_step + i;
        at = (at >= 9)  ?  (at - 9)  :  at;
        st[(size_t) (VT_RRC_RE + at)*n] = __float_as_int(are[i]);
        st[(size_t) (VT_RRC_IM + at)*n] = __float_as_int(aim[i]);
    }
    st[VT_RRC_STEP*n] = rrc_step;
    st[VT_SCRAMBLE*n] = (int32_t) scramble;
    st[VT_TRAIN_SCRAMBLE*n] = (int32_t) train_scramble;
    st[VT_IN_TRAINING*n] = in_training;
    st[VT_TRAINING_STEP*n] = training_step;
    st[VT_CARRIER_PHASE*n] = (int32_t) carrier_phase;
    st[VT_BAUD_PHASE*n] = baud_phase;
    st[VT_CONSTELLATION*n] = constellation;
    st[VT_PRBS*n] = (int32_t) prbs;
}

}   // namespace spg
