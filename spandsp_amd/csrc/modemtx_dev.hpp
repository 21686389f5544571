// modemtx_dev.hpp -- device side of the modem transmitter banks: N V.29, V.27ter or
// V.17 modulators, one channel per lane, as the signal sources of the receiver banks.
//
// What is restated (paths relative to the reference tree; float build, x86-64):
//   v29_tx()                   src/v29tx.c:226-284
//   getbaud(), get_scrambled_bit()   src/v29tx.c:103-224   (training segments, data scrambler, differential phase map)
//   v27ter_tx()                src/v27ter_tx.c:246-350   (1600 baud at 4800 bps, 1200 baud at 2400 bps)
//   getbaud(), scramble()      src/v27ter_tx.c:103-244   (the scrambler with its guard against repeating patterns)
//   v17_tx()                   src/v17tx.c:295-369
//   training_get(), diff_and_convolutional_encode(), getbaud()   src/v17tx.c:106-293   (long / short training, the
//                              bridge, differential + convolutional encoding into the 4 .. 128 point constellations)
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
    VT_BASE_GAIN = 1,       // float (V.27ter: gain_2400; V.17: gain)
    VT_GAIN = 2,            // float (V.27ter: gain_4800; V.17: int diff)
    VT_RRC_RE = 3,          // 9 floats, ring order (as the reference keeps them)
    VT_RRC_IM = 12,         // 9 floats
    VT_RRC_STEP = 21,
    VT_SCRAMBLE = 22,
    VT_TRAIN_SCRAMBLE = 23, // V.27ter: scrambler_pattern_count; V.17: convolution
    VT_IN_TRAINING = 24,
    VT_TRAINING_STEP = 25,
    VT_TRAINING_OFFSET = 26,    // V.17: short_train
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
constexpr int kTxV17 = 2;

// V.17 training, in symbols (v17tx.c:86-103)
constexpr int kV17TepB = 480;
constexpr int kV17Seg1 = kV17TepB + 48;
constexpr int kV17Seg2 = kV17Seg1 + 256;
constexpr int kV17Seg3 = kV17Seg2 + 2976;
constexpr int kV17Seg4 = kV17Seg3 + 64;
constexpr int kV17ShortSeg4 = kV17Seg2 + 38;
constexpr int kV17End = kV17Seg4 + 48;
constexpr int kV17ShutdownEnd = kV17End + 32 + 48;

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
    const float *shaper;        // V.29, V.17: [10][9]; V.27ter: [5][9] (4800 bps) then [20][9] (2400 bps)
    const float *constel;       // V.17: [128 + 64 + 32 + 16 + 4][2] (14400 .. 4800 bps) then the ABCD points [4][2]
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
    __shared__ float constel[(KIND == kTxV17)  ?  248  :  1][2];
    const int lane = threadIdx.x;
    const int ch = blockIdx.x*64 + lane;

    for (int i = lane;  i < 2048;  i += 64)
        sine[i] = L.sine[i];
    for (int i = lane;  i < ((KIND == kTxV27ter)  ?  225  :  90);  i += 64)
        (&shaper[0][0])[i] = L.shaper[i];
    if (KIND == kTxV17)
    {
        for (int i = lane;  i < 496;  i += 64)
            (&constel[0][0])[i] = L.constel[i];
    }
    __syncthreads();
    if (ch >= L.n_ch)
        return;

    int32_t *st = L.st + ch;
    const size_t n = (size_t) L.n_ch;
    const int bit_rate = st[VT_BIT_RATE*n];
    // V.27ter keeps one gain per rate: word 1 for 2400 bps, word 2 for 4800 bps
    const float gain = __int_as_float(st[(((KIND == kTxV27ter  &&  bit_rate == 2400)  ||  KIND == kTxV17)  ?  VT_BASE_GAIN  :  VT_GAIN)*n]);
    int diff = (KIND == kTxV17)  ?  st[VT_GAIN*n]  :  0;
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

    // scramble(), v17tx.c:106-116 (scrambler_tap = 17)
    auto scramble17 = [&](int in_bit) -> int
    {
        const int out = (int) (((uint32_t) in_bit ^ (scramble >> 17) ^ (scramble >> 22)) & 1u);
        scramble = (scramble << 1) | (uint32_t) out;
        return out;
    };
    // the constellation of this lane's rate inside `constel`
    const int pts_at = (bit_rate == 14400)  ?  0  :  ((bit_rate == 12000)  ?  128  :  ((bit_rate == 9600)  ?  192  :  ((bit_rate == 7200)  ?  224  :  240)));

    int16_t *row = L.pcm + (size_t) ch*L.stride;
    const bool silent = (training_step >= ((KIND == kTxV29)  ?  kVtShutdownEnd  :  ((KIND == kTxV27ter)  ?  kV27ShutdownEnd  :  kV17ShutdownEnd)));
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
                if (KIND == kTxV29  ||  KIND == kTxV17)
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
                    else if (KIND == kTxV17)
                    {
                        // getbaud() / training_get(), v17tx.c:118-293; training_offset holds short_train, train_scramble
                        // the convolutional encoder state
                        bool have = false;
                        if (in_training  &&  training_step <= kV17End)
                        {
                            if (training_step < kV17Seg4)
                            {
                                have = true;
                                training_step++;
                                int k;
                                if (training_step <= kV17Seg2)
                                {
                                    // TEP carrier (A), silence, then ABAB
                                    k = (training_step <= kV17TepB)  ?  0  :  ((training_step & 1) ^ 1);
                                    if (training_step > kV17TepB  &&  training_step <= kV17Seg1)
                                        k = -1;
                                }
                                else if (training_step <= kV17Seg3)
                                {
                                    // CDBA through the scrambler
                                    int bits = scramble17(1);
                                    bits = (bits << 1) | scramble17(1);
                                    constellation = (0x0132 >> (bits*4)) & 3;                   // {2, 3, 1, 0}
                                    if (training_offset  &&  training_step == kV17ShortSeg4)
                                        training_step = kV17Seg4;
                                    k = constellation;
                                }
                                else
                                {
                                    // the bridge, carrying 0x8880
                                    const int shift = ((training_step - kV17Seg3 - 1) & 7) << 1;
                                    int bits = scramble17((0x8880 >> shift) & 1);
                                    bits = (bits << 1) | scramble17((0x8880 >> (shift + 1)) & 1);
                                    constellation = (constellation + ((0x3201 >> (bits*4)) & 3)) & 3;  // {1, 0, 2, 3}
                                    k = constellation;
                                }
                                vre = (k < 0)  ?  0.0f  :  constel[244 + ((k < 0)  ?  0  :  k)][0];
                                vim = (k < 0)  ?  0.0f  :  constel[244 + ((k < 0)  ?  0  :  k)][1];
                            }
                            else
                            {
                                training_step++;
                                if (training_step > kV17End)
                                    in_training = 0;
                            }
                        }
                        if (!have)
                        {
                            const int nbits = (bit_rate == 14400)  ?  6  :  ((bit_rate == 12000)  ?  5  :  ((bit_rate == 9600)  ?  4  :  ((bit_rate == 7200)  ?  3  :  2)));
                            int q = 0;
                            for (int i = 0;  i < nbits;  i++)
                            {
                                int bit = 1;
                                if (!in_training)
                                {
                                    bit = (int) (((prbs >> 14) ^ (prbs >> 13)) & 1u);
                                    prbs = ((prbs << 1) | (uint32_t) bit) & 0x7FFFu;
                                }
                                q |= scramble17(bit) << i;
                            }
                            // diff_and_convolutional_encode(), v17tx.c:170-222
                            int idx = 0;
                            if (bit_rate != 4800)
                            {
                                diff = (diff + (q & 3)) & 3;        // v17_differential_encoder: addition mod 4
                                // v17_convolutional_encoder[8][4], one byte per row, two bits... kept as a table of nibbles
                                const uint32_t row4 = (train_scramble == 0)  ?  0x1320u  :  (train_scramble == 1)  ?  0x6574u  :  (train_scramble == 2)  ?  0x0231u
                                                      :  (train_scramble == 3)  ?  0x5647u  :  (train_scramble == 4)  ?  0x3102u  :  (train_scramble == 5)  ?  0x4756u
                                                      :  (train_scramble == 6)  ?  0x2013u  :  0x7465u;
                                train_scramble = (row4 >> (diff*4)) & 7u;
                                idx = ((q << 1) & 0x78) | (diff << 1) | (int) ((train_scramble >> 2) & 1u);
                            }
                            else
                            {
                                // v32bis_4800_differential_encoder[diff][q]: {2,3,0,1},{0,2,1,3},{3,1,2,0},{1,0,3,2}
                                const uint32_t d4 = (diff == 0)  ?  0x1032u  :  (diff == 1)  ?  0x3120u  :  (diff == 2)  ?  0x0213u  :  0x2301u;
                                diff = (int) ((d4 >> ((q & 3)*4)) & 3u);
                                idx = diff;
                            }
                            vre = constel[pts_at + idx][0];
                            vim = constel[pts_at + idx][1];
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
                const float *coef = (KIND != kTxV27ter)  ?  shaper[9 - baud_phase]
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
    if (KIND == kTxV17)
        st[VT_GAIN*n] = diff;
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
