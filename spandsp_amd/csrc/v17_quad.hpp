// v17_quad.hpp -- the V.17 (and V.32bis 4800) receiver with FOUR LANES PER CHANNEL (reference: src/v17rx.c:214-1358,
// src/godard.c:144-220; state word map and tables: v17_common.hpp).  The mapping is that of v29_quad.hpp, whose header
// explains it -- a round is one baud of a channel, its samples dealt over the lanes of a DPP quad for the pulse
// shaping filters (quad_round_front.inc, shared with V.29: the two receivers' sample paths differ in constants only),
// the equaliser's inner product as four chains, the LMS update by taps -- plus what V.17 adds:
//   * the 192-phase pulse shaper (41 KB as {re, im} pairs) stays in global memory, [tap][phase]: a lane's 27 coefficient
//     pairs are requested as soon as the round's plan is known;
//   * the trellis decoder (v17rx.c:396-589): lane r forms the distances to candidate points r and r + 4 and the new
//     metrics of states r and r + 4 (the four predecessors of a state are per-lane constants; the distances they need
//     are picked from a little LDS array); the survivor words are OR-ed together over the quad, the 16-deep survivor
//     memory is packed in LDS per channel, and the walk back through it is replicated;
//   * the equaliser delay line is kept as [B | 0 | B] (no third copy: with the survivor memory a channel's LDS has to
//     stay under 2.2 KB for 64 channels and the tables to share a CU); the LMS update picks its entry from one of the
//     two copies.
#pragma once

#include "v17_common.hpp"
#include "v29_quad.hpp"

namespace spg {

struct V17QuadTables                                    // per workgroup, in LDS
{
    float sine[2048];
    float con[256];
    uint32_t map[36*36*2];
};

constexpr int kQuad17EqStride = 100;                    // pairs: [B | 0 | B] = 99 (+1); 200 words = 3*64 + 8
constexpr int kQuad17TrellisStride = 57;                // words: 16 + 32 + 8 (+1)

struct V17QuadChan                                      // per channel, in LDS (strides: see v29_quad.hpp)
{
    uint32_t *pcm;                                      // [kV29QuadTile/2]
    float2 *rrc;                                        // [54]
    float2 *u;                                          // [99] eq_buf in ring order: [B | 0 | B]
    float *taps;                                        // [99] {re, im, -re} per tap
    uint32_t *trellis;                                  // [16] 8 x 3 bit predecessor states per step, [16][2] 8 x 1 byte points,
                                                        // [8] the baud's distances to the eight candidate points
};

SPG_FN void v17_quad_tables(V17QuadTables &T, const V17Tables &TB, int tid, int n)
{
    for (int i = tid;  i < 2048;  i += n)
        T.sine[i] = TB.sine[i];
    for (int i = tid;  i < 256;  i += n)
        T.con[i] = TB.con[i];
    for (int i = tid;  i < 36*36*2;  i += n)
        T.map[i] = TB.map[i];
}

template <class Q>
SPG_FN void v17_quad_run(Q &q, const V17Launch &L, const int ch, const V17QuadTables &T, const V17QuadChan C)
{
    const int role = q.role();
    const V17Tables &TB = *L.tab;
    const float2 *g_rrc = (const float2 *) TB.rrc_q;
    const float g0 = TB.godard[0];
    const float g1 = TB.godard[1];
    const float g2 = TB.godard[2];
    const float g3 = TB.godard[3];
    const float g4 = TB.godard[4];
    const float g5 = TB.godard[5];
    const float g6 = TB.godard[6];
    const float fine_trigger = TB.fine_trigger;
    const float coarse_trigger = TB.coarse_trigger;
    const int fine_step = TB.fine_step;
    const int coarse_step = TB.coarse_step;

    // bank-wide constants of the bit rate (v17rx.c:1403-1436)
    const int bit_rate = L.bit_rate;
    const int bits_per_symbol = (bit_rate == 14400)  ?  6  :  (bit_rate == 12000)  ?  5  :  (bit_rate == 9600)  ?  4  :  (bit_rate == 7200)  ?  3  :  2;
    const int space_map = (bit_rate == 12000)  ?  1  :  (bit_rate == 9600)  ?  2  :  (bit_rate == 7200)  ?  3  :  0;
    const float spacing = (space_map == 0)  ?  1.414f  :  (space_map == 1)  ?  2.0f  :  (space_map == 2)  ?  2.828f  :  4.0f;

    const size_t N = (size_t) L.n_ch;
    const int mylen = L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;
    auto ldf = [&](int w) { return __uint_as_float(L.state[(size_t) w*N + ch]); };
    auto ldi = [&](int w) { return (int32_t) L.state[(size_t) (kV17Floats + w)*N + ch]; };
    // a NaN goes back as x86's (v29_dev.hpp)
    auto stf = [&](int w, float v) { L.state[(size_t) w*N + ch] = (v != v)  ?  0xFFC00000u  :  __float_as_uint(v); };
    auto sti = [&](int w, int32_t v) { L.state[(size_t) (kV17Floats + w)*N + ch] = (uint32_t) v; };

#define PAST(t)     C.trellis[(t)]
#define FULL(t, h)  C.trellis[16 + 2*(t) + (h)]
#define DIST(i)     (((float *) C.trellis)[48 + (i)])
    // tcm_paths (v17rx.c:452-462), a row as four 3 bit numbers; this lane forms the metrics of states role and role + 4
    constexpr uint32_t kTpRow[8] =
    {
        0u | 6u << 3 | 2u << 6 | 4u << 9, 6u | 0u << 3 | 4u << 6 | 2u << 9, 2u | 4u << 3 | 0u << 6 | 6u << 9, 4u | 2u << 3 | 6u << 6 | 0u << 9,
        1u | 3u << 3 | 7u << 6 | 5u << 9, 5u | 7u << 3 | 3u << 6 | 1u << 9, 7u | 5u << 3 | 1u << 6 | 3u << 9, 3u | 1u << 3 | 5u << 6 | 7u << 9
    };
    const uint32_t tp_row_a = (role == 0)  ?  kTpRow[0]  :  (role == 1)  ?  kTpRow[1]  :  (role == 2)  ?  kTpRow[2]  :  kTpRow[3];
    const uint32_t tp_row_b = (role == 0)  ?  kTpRow[4]  :  (role == 1)  ?  kTpRow[5]  :  (role == 2)  ?  kTpRow[6]  :  kTpRow[7];

    // ---- state: scalars replicated in the four lanes, arrays into LDS (dealt over the lanes) ---------------------------
    float agc_scaling = ldf(VF_AGC);
    float agc_scaling_save = ldf(VF_AGC_SAVE);
    float eq_delta = ldf(VF_EQ_DELTA);
    float training_error = ldf(VF_TRAIN_ERR);
    float carrier_track_p = ldf(VF_TRACK_P);
    float carrier_track_i = ldf(VF_TRACK_I);
    float glow0 = ldf(VF_GLOW);
    float glow1 = ldf(VF_GLOW + 1);
    float ghigh0 = ldf(VF_GHIGH);
    float ghigh1 = ldf(VF_GHIGH + 1);
    float gdc0 = ldf(VF_GDC);
    float gdc1 = ldf(VF_GDC + 1);
    float baud_phase = ldf(VF_BAUD_PHASE);
    for (int i = role;  i < kRrcLen;  i += 4)
    {
        const float v = ldf(VF_RRC + i);
        C.rrc[i] = make_float2(v, 0.0f);
        C.rrc[kRrcLen + i] = make_float2(0.0f, v);
    }
    for (int i = role;  i < kEqLen;  i += 4)
    {
        const float cr = ldf(VF_EQ_COEFF + 2*i);
        const float ci = ldf(VF_EQ_COEFF + 2*i + 1);
        C.taps[3*i] = cr;
        C.taps[3*i + 1] = ci;
        C.taps[3*i + 2] = -cr;
        const float2 x = make_float2(ldf(VF_EQ_BUF + 2*i), ldf(VF_EQ_BUF + 2*i + 1));
        C.u[i] = x;
        C.u[kEqLen + i] = make_float2(0.0f, 0.0f);
        C.u[2*kEqLen + i] = x;
    }
    float sd[8];
    SPG_UNROLL
    for (int i = 0;  i < 8;  i++)
        sd[i] = ldf(XF_DIST + i);
    for (int t = role;  t < 16;  t += 4)
    {
        uint32_t p = 0;
        uint32_t f0 = 0;
        uint32_t f1 = 0;
        for (int i = 0;  i < 8;  i++)
        {
            p |= ((uint32_t) ldi(XI_PAST_STATE + 8*t + i) & 7u) << (3*i);
            const uint32_t b = (uint32_t) ldi(XI_FULL_PATH + 8*t + i) & 0xFFu;
            if (i < 4)
                f0 |= b << (8*i);
            else
                f1 |= b << (8*(i - 4));
        }
        PAST(t) = p;
        FULL(t, 0) = f0;
        FULL(t, 1) = f1;
    }
    int rrc_step = ldi(XI_RRC_STEP);
    int diff = ldi(XI_DIFF);
    uint32_t scramble_reg = (uint32_t) ldi(XI_SCRAMBLE);
    int short_train = ldi(XI_SHORT_TRAIN);
    int stage = ldi(XI_STAGE);
    int training_count = ldi(XI_TRAIN_COUNT);
    int last_sample = ldi(XI_LAST_SAMPLE);
    int signal_present = ldi(XI_SIGNAL_PRESENT);
    int drop_pending = ldi(XI_DROP_PENDING);
    int low_samples = ldi(XI_LOW_SAMPLES);
    int high_sample = ldi(XI_HIGH_SAMPLE);
    uint32_t carrier_phase = (uint32_t) ldi(XI_CARRIER_PHASE);
    int32_t carrier_phase_rate = ldi(XI_PHASE_RATE);
    int32_t carrier_phase_rate_save = ldi(XI_PHASE_RATE_SAVE);
    int32_t power_reading = ldi(XI_POWER);
    const int32_t carrier_on_power = ldi(XI_ON_POWER);
    const int32_t carrier_off_power = ldi(XI_OFF_POWER);
    int eq_step = ldi(XI_EQ_STEP);
    int eq_put_step = ldi(XI_EQ_PUT_STEP);
    int eq_skip = ldi(XI_EQ_SKIP);
    int baud_half = ldi(XI_BAUD_HALF);
    int32_t last_angle0 = ldi(XI_LAST_ANGLES);
    int32_t last_angle1 = ldi(XI_LAST_ANGLES + 1);
    int trellis_ptr = ldi(XI_TRELLIS_PTR);
    int total_corr = ldi(XI_TOTAL_CORR);
    // diff_angles[16] stays in the state array (every lane of the quad stores the same value and reads it back)
    auto diff_ld = [&](int k) { return ldi(XI_DIFF_ANGLES + (k & 0xF)); };
    auto diff_st = [&](int k, int32_t v) { sti(XI_DIFF_ANGLES + (k & 0xF), v); };

    int8_t *evp = L.events + (size_t) ch*L.ev_cap;
    int n_ev = 0;
    auto emit = [&](int v)
    {
        if (role == 0  &&  n_ev < L.ev_cap)
            evp[n_ev] = (int8_t) v;
        n_ev++;
    };

    // v17_rx_restart(s, s->bit_rate, s->short_train), v17rx.c:1399-1500: all four lanes, the same stores
    auto restart = [&]()
    {
        for (int i = 0;  i < 2*kRrcLen;  i++)
            C.rrc[i] = make_float2(0.0f, 0.0f);
        training_error = 0.0f;
        rrc_step = 0;
        diff = 1;
        scramble_reg = 0x2ECDD5;
        stage = V17_SYMBOL_ACQUISITION;
        training_count = 0;
        signal_present = 0;
        high_sample = 0;
        low_samples = 0;
        drop_pending = 0;
        last_angle0 = 0;
        last_angle1 = 0;
        for (int k = 0;  k < 16;  k++)
            diff_st(k, 0);
        SPG_UNROLL
        for (int i = 0;  i < 8;  i++)
            sd[i] = 99.0f*1.0f;
        sd[0] = 0.0f;
        for (int t = 0;  t < 16;  t++)
        {
            PAST(t) = 0;
            FULL(t, 0) = 0;
            FULL(t, 1) = 0;
        }
        trellis_ptr = 14;
        carrier_phase = 0;
        power_reading = 0;
        for (int i = 0;  i < 3*kEqLen;  i++)
            C.u[i] = make_float2(0.0f, 0.0f);
        eq_put_step = kV17Sets*10/(3*2) - 1;
        eq_step = 0;
        eq_skip = 0;
        if (short_train)
        {
            carrier_phase_rate = carrier_phase_rate_save;
            for (int i = 0;  i < kEqLen;  i++)
            {
                const float cr = ldf(VF_EQ_SAVE + 2*i);
                C.taps[3*i] = cr;
                C.taps[3*i + 1] = ldf(VF_EQ_SAVE + 2*i + 1);
                C.taps[3*i + 2] = -cr;
            }
            eq_delta = 0.1f*(0.21f/kEqLen);
            agc_scaling = agc_scaling_save;
            carrier_track_i = 0.0f;
            carrier_track_p = 40000.0f;
        }
        else
        {
            carrier_phase_rate = v29_f2i(1800.0f*65536.0f*65536.0f/8000);
            for (int i = 0;  i < kEqLen;  i++)
            {
                const float cr = (i == 16)  ?  3.0f  :  0.0f;
                C.taps[3*i] = cr;
                C.taps[3*i + 1] = 0.0f;
                C.taps[3*i + 2] = -cr;
            }
            eq_delta = 0.21f/kEqLen;
            agc_scaling_save = 0.0f;
            agc_scaling = (2.17f/1.000000f)/735.0f;
            carrier_track_i = 5000.0f;
            carrier_track_p = 40000.0f;
        }
        last_sample = 0;
        glow0 = glow1 = ghigh0 = ghigh1 = gdc0 = gdc1 = 0.0f;
        baud_phase = 0.0f;
        total_corr = 0;
        baud_half = 0;
    };

    // track_carrier() and tune_equalizer(): requested by the stage logic, carried out once after it (see v29_dev.hpp)
    bool do_track = false;
    bool do_tune = false;
    bool do_save = false;
    float tgt_re = 0.0f;
    float tgt_im = 0.0f;
    float use_track_i = 0.0f;
    float use_track_p = 0.0f;
    float use_delta = 0.0f;
    auto track_carrier = [&](float tre, float tim)
    {
        do_track = true;
        tgt_re = tre;
        tgt_im = tim;
        use_track_i = carrier_track_i;
        use_track_p = carrier_track_p;
    };
    auto tune_equalizer = [&](float tre, float tim)
    {
        do_tune = true;
        tgt_re = tre;
        tgt_im = tim;
        use_delta = eq_delta;
    };
    // v17rx.c:336-349 (scrambler_tap is 18 - 1: v17_rx_init never changes it)
    auto descramble = [&](int in_bit)
    {
        in_bit &= 1;
        const int out_bit = (in_bit ^ (int) (scramble_reg >> 17) ^ (int) (scramble_reg >> 22)) & 1;
        const bool training = (stage > V17_NORMAL  &&  stage < V17_TCM_WINDUP);
        scramble_reg = (scramble_reg << 1) | (uint32_t) (training  ?  out_bit  :  in_bit);
        return out_bit;
    };
    auto put_bit = [&](int bit)
    {
        const int out_bit = descramble(bit);
        if (stage == V17_NORMAL)
            emit(out_bit);
    };
    auto cdba = [&](int bit, float &tre, float &tim)
    {
        // v17rx.c:601-607: {6, 2}, {-2, 6}, {2, -6}, {-6, -2}
        tre = (bit == 0)  ?  6.0f  :  (bit == 1)  ?  -2.0f  :  (bit == 2)  ?  2.0f  :  -6.0f;
        tim = (bit == 0)  ?  2.0f  :  (bit == 1)  ?  6.0f  :  (bit == 2)  ?  -6.0f  :  -2.0f;
    };
    // rotate the equaliser delay line (each lane every fourth entry, both copies) and the carrier
    auto spin = [&](uint32_t phase_step)
    {
        const float p = phase_step*2.0f*3.1415926f/(65536.0f*65536.0f);
        const float zc = spg_sincosf(p, true);
        const float zs = -spg_sincosf(p, false);
        for (int k = role;  k < kEqLen;  k += 4)
        {
            const float2 xv = C.u[k];
            const float2 r = make_float2(xv.x*zc - xv.y*zs, xv.x*zs + xv.y*zc);
            C.u[k] = r;
            C.u[2*kEqLen + k] = r;
        }
        carrier_phase += phase_step;
    };
    auto park = [&](bool clear_agc)
    {
        if (clear_agc)
            agc_scaling_save = 0.0f;
        stage = V17_PARKED;
        emit(-5);                                           // SIG_STATUS_TRAINING_FAILED
    };

    // the accumulated distance of predecessor j of a state of set `set`: distances[(j << 1) + set]
    auto sd_pick = [&](const int set, const int j) -> float { return sd[(j << 1) + set]; };
    // n data bits through the descrambler (put_bit(), v17rx.c:351-367; first bit = bit 0 of raw), stored together.  The
    // taps (17, 22) lie beyond the six bits of a baud, so every output bit only needs the register as it was before.
    auto put_bits = [&](const int raw, const int n)
    {
        const bool training = (stage > V17_NORMAL  &&  stage < V17_TCM_WINDUP);
        if (training)
        {
            for (int i = 0;  i < n;  i++)
                put_bit(raw >> i);
            return;
        }
        // bit j of the baud goes with the register's bits 17 - j and 22 - j: the baud's six bits against the bit-reversed six at
        // 12 and 17 (a reversal of six bits = two look-ups of three in a word of nibbles), then one multiply to give every bit
        // of a nibble a byte of its own
        const uint32_t rev3 = 0x73516240u;
        auto rev6 = [&](const uint32_t v) -> uint32_t { return (((rev3 >> (4*(v & 7u))) & 7u) << 3) | ((rev3 >> (4*((v >> 3) & 7u))) & 7u); };
        const uint32_t o6 = ((uint32_t) raw & 0x3Fu) ^ rev6(((scramble_reg >> 12) ^ (scramble_reg >> 17)) & 0x3Fu);
        const uint32_t lo = ((o6 & 0xFu)*0x00204081u) & 0x01010101u;
        const uint32_t hi = ((o6 >> 4)*0x00204081u) & 0x01010101u;
        const uint32_t rev = rev6((uint32_t) raw & 0x3Fu) >> (6 - n);
        scramble_reg = (scramble_reg << n) | rev;
        if (stage == V17_NORMAL)
        {
            if (role == 0)
            {
                if (n_ev + 8 <= L.ev_cap)
                {
                    const uint32_t w2[2] = {lo, hi};
                    __builtin_memcpy(evp + n_ev, w2, 8);    // bytes past the baud's bits are overwritten by the next
                }
                else
                {
                    for (int jb = 0;  jb < n;  jb++)
                    {
                        if (n_ev + jb < L.ev_cap)
                            evp[n_ev + jb] = (int8_t) ((((jb < 4)  ?  lo  :  hi) >> (8*(jb & 3))) & 1u);
                    }
                }
            }
            n_ev += n;
        }
    };

    // decode_baud(), v17rx.c:396-589.  Returns the constellation point the carrier loop tracked.
    auto decode_baud = [&](float zre, float zim) -> int
    {
        int re = v29_f2i((zre + 9.0f)*2.0f);
        int im = v29_f2i((zim + 9.0f)*2.0f);
        re = max(0, min(35, re));
        im = max(0, min(35, im));
        if (bits_per_symbol == 2)
        {
            const int cell = re*36 + im;
            const int cs = (int) ((T.map[cell >> 2] >> (8*(cell & 3))) & 0xFF);
            // v32bis_4800_differential_decoder[diff][cs] = {{2,3,0,1},{0,2,1,3},{3,1,2,0},{1,0,3,2}}, 2 bits per entry
            constexpr uint32_t dec = (2u | 3u << 2 | 0u << 4 | 1u << 6) | (0u | 2u << 2 | 1u << 4 | 3u << 6) << 8
                                   | (3u | 1u << 2 | 2u << 4 | 0u << 6) << 16 | (1u | 0u << 2 | 3u << 4 | 2u << 6) << 24;
            const int raw = (int) ((dec >> (2*(diff*4 + cs))) & 3u);
            diff = cs;
            put_bits(raw, 2);
            return cs;
        }
        float dist[8];
        const uint32_t c_lo = T.map[(re*36 + im)*2];
        const uint32_t c_hi = T.map[(re*36 + im)*2 + 1];
        // the distances to the eight candidate points: this lane's two, then all eight on every lane
        {
            const int cell_a = (int) ((c_lo >> (8*role)) & 0xFF);
            const int cell_b = (int) ((c_hi >> (8*role)) & 0xFF);
            const float dxa = T.con[2*cell_a] - zre;
            const float dya = T.con[2*cell_a + 1] - zim;
            const float dxb = T.con[2*cell_b] - zre;
            const float dyb = T.con[2*cell_b + 1] - zim;
            const float my_da = dxa*dxa + dya*dya;
            const float my_db = dxb*dxb + dyb*dyb;
            DIST(role) = my_da;
            DIST(role + 4) = my_db;
            dist[0] = q.template bcast<0>(my_da, 21);
            dist[1] = q.template bcast<1>(my_da, 22);
            dist[2] = q.template bcast<2>(my_da, 23);
            dist[3] = q.template bcast<3>(my_da, 24);
            dist[4] = q.template bcast<0>(my_db, 25);
            dist[5] = q.template bcast<1>(my_db, 26);
            dist[6] = q.template bcast<2>(my_db, 27);
            dist[7] = q.template bcast<3>(my_db, 28);
        }
        auto cell_of = [&](int i) { return (int) ((((i < 4)  ?  c_lo  :  c_hi) >> (8*(i & 3))) & 0xFF); };
        float mn = 9999999.0f;
        int cs = (int) (c_lo & 0xFF);
        SPG_UNROLL
        for (int i = 0;  i < 8;  i++)
        {
            if (mn > dist[i])
            {
                mn = dist[i];
                cs = (int) ((((i < 4)  ?  c_lo  :  c_hi) >> (8*(i & 3))) & 0xFF);
            }
        }
        track_carrier(T.con[2*cs], T.con[2*cs + 1]);

        if (++trellis_ptr >= 16)
            trellis_ptr = 0;
        // the new metric of one state: the best of its four predecessors (v17rx.c:496-541).  rows = the state's row of
        // tcm_paths as four 3 bit numbers; set = state >> 2
        auto one_state = [&](const uint32_t rows, const int set, float &nd_out, uint32_t &kk_out, uint32_t &cell_out)
        {
            const int t0 = (int) (rows & 7u);
            float sel_d = DIST(t0);
            float sel_sd = sd_pick(set, 0);
            float best = sel_d + sel_sd;
            int sel_t = t0;
            int kk = set;
            SPG_UNROLL
            for (int j = 1;  j < 4;  j++)
            {
                const int tj = (int) ((rows >> (3*j)) & 7u);
                const float dj = DIST(tj);
                const float sj = sd_pick(set, j);
                const float t = dj + sj;
                const bool better = (best > t);
                best = better  ?  t  :  best;
                sel_sd = better  ?  sj  :  sel_sd;
                sel_d = better  ?  dj  :  sel_d;
                sel_t = better  ?  tj  :  sel_t;
                kk = better  ?  ((j << 1) + set)  :  kk;
            }
            nd_out = sel_sd*0.9f + sel_d*0.1f;
            kk_out = (uint32_t) kk;
            cell_out = (uint32_t) cell_of(sel_t);
        };
        q.sync(9);
        float nd_a;
        float nd_b;
        uint32_t kk_a;
        uint32_t kk_b;
        uint32_t cell_a2;
        uint32_t cell_b2;
        one_state(tp_row_a, 0, nd_a, kk_a, cell_a2);
        one_state(tp_row_b, 1, nd_b, kk_b, cell_b2);
        uint32_t pw = (kk_a << (3*role)) | (kk_b << (3*(role + 4)));
        uint32_t f_lo = cell_a2 << (8*role);
        uint32_t f_hi = cell_b2 << (8*role);
        pw |= q.swap1(pw, 1);
        f_lo |= q.swap1(f_lo, 2);
        f_hi |= q.swap1(f_hi, 3);
        pw |= q.swap2(pw, 4);
        f_lo |= q.swap2(f_lo, 5);
        f_hi |= q.swap2(f_hi, 6);
        PAST(trellis_ptr) = pw;
        FULL(trellis_ptr, 0) = f_lo;
        FULL(trellis_ptr, 1) = f_hi;
        sd[0] = q.template bcast<0>(nd_a, 31);
        sd[1] = q.template bcast<1>(nd_a, 32);
        sd[2] = q.template bcast<2>(nd_a, 33);
        sd[3] = q.template bcast<3>(nd_a, 34);
        sd[4] = q.template bcast<0>(nd_b, 35);
        sd[5] = q.template bcast<1>(nd_b, 36);
        sd[6] = q.template bcast<2>(nd_b, 37);
        sd[7] = q.template bcast<3>(nd_b, 38);
        q.sync(10);
        mn = sd[0];
        int k = 0;
        SPG_UNROLL
        for (int i = 1;  i < 8;  i++)
        {
            if (mn > sd[i])
            {
                mn = sd[i];
                k = i;
            }
        }
        // trace back: the fifteen words of the path are requested together -- where they lie does not depend on the state
        // the walk is in -- and the walk itself is then register work
        uint32_t pth[15];
        SPG_UNROLL
        for (int i = 0;  i < 15;  i++)
            pth[i] = PAST((trellis_ptr - i) & 15);
        SPG_UNROLL
        for (int i = 0;  i < 15;  i++)
            k = (int) ((pth[i] >> (3*k)) & 7u);
        const int j = (trellis_ptr - 15) & 15;
        const int nearest = (int) ((FULL(j, k >> 2) >> (8*(k & 3))) & 0xFF) >> 1;
        int raw = (nearest & 0x3C) | (((nearest & 3) - diff) & 3);      // v17_differential_decoder[diff][nearest & 3]
        diff = nearest & 3;
        put_bits(raw, bits_per_symbol);
        return cs;
    };

    SPG_PROF_DECL();
    const int16_t *src = L.amp + (size_t) ch*L.stride;
    SPG_LOADS_DONE();
    q.sync(1);
    for (int tile = 0;  tile < L.samples;  tile += kV29QuadTile)
    {
    const int tn = max(0, min(kV29QuadTile, mylen - tile));
    // ---- stage the channel's stretch of PCM: pcm[k] = samples 2k, 2k+1 of the tile (16-byte pieces dealt over the lanes)
    {
        const int16_t *row = src + tile;
        const bool wide = ((((uintptr_t) row) & 15) == 0)  &&  (tn == kV29QuadTile);
        if (wide)
        {
            for (int k = role;  k < kV29QuadTile/8;  k += 4)
            {
                const int4 v = ((const int4 *) row)[k];
                C.pcm[4*k + 0] = (uint32_t) v.x;
                C.pcm[4*k + 1] = (uint32_t) v.y;
                C.pcm[4*k + 2] = (uint32_t) v.z;
                C.pcm[4*k + 3] = (uint32_t) v.w;
            }
        }
        else
        {
            for (int k = role;  k < (tn + 1)/2;  k += 4)
            {
                const uint32_t lo = (uint16_t) row[2*k];
                const uint32_t hi = (2*k + 1 < tn)  ?  (uint16_t) row[2*k + 1]  :  0u;
                C.pcm[k] = lo | (hi << 16);
            }
        }
    }
    q.sync(2);
    int pos = 0;
    // The round loop, bottom tested (the compiler does not rotate a loop whose test is a cross-lane operation, and with the
    // test at the top it kept the loop-carried state in two register sets: some forty copies at the head of every round)
    if (q.any(pos < tn, 1)) do
    {
        SPG_PROF_STAMP(0);
#define QF_SETS                 kV17Sets
#define QF_PUT_ADD              (kV17Sets*10/(3*2))
#define QF_PARKED               V17_PARKED
#define QF_AGC_NUM              (2.17f/1.000000f)
#define QF_TILE                 kV29QuadTile
#define QF_EQLEN                kEqLen
#define QF_RRC_COEF(tap)        g_rrc[(tap)*kV17Sets + my_step]
#define QF_EQ_THIRD_COPY(k, h)  do { } while (0)
#include "quad_round_front.inc"
#undef QF_SETS
#undef QF_PUT_ADD
#undef QF_PARKED
#undef QF_AGC_NUM
#undef QF_TILE
#undef QF_EQLEN
#undef QF_RRC_COEF
#undef QF_EQ_THIRD_COPY

        // ---- the baud (process_half_baud() of the second T/2 instant, v17rx.c:592-1130) ---------------------------------
        if (baud_done)
        {
            carrier_phase -= (uint32_t) carrier_phase_rate;
            {
                // godard_ted_per_baud(), godard.c:165-220
                float cv = glow1*ghigh0*g2 - glow0*ghigh1*g5 + glow1*ghigh1*g6;
                const float p = cv - gdc1;
                gdc1 = gdc0;
                gdc0 = cv;
                baud_phase -= p;
                cv = fabsf(baud_phase);
                if (cv > fine_trigger)
                {
                    int i = (cv > coarse_trigger)  ?  coarse_step  :  fine_step;
                    if (baud_phase < 0.0f)
                        i = -i;
                    total_corr += i;
                    eq_put_step += i;
                }
            }
            // equalizer_get(): cvec_circular_dot_prodf, one chain per lane (see v29_quad.hpp)
            float zre;
            float zim;
            {
                const float2 *x = &C.u[eq_step + ((role & 2)  ?  kEqLen  :  0)];
                const float *c = &C.taps[role & 1];
                float acc = 0.0f;
                SPG_UNROLL
                for (int i0 = 0;  i0 < kEqLen;  i0 += 11)
                {
                    float2 xs[11];
                    float ca[11];
                    float cb[11];
                    SPG_UNROLL
                    for (int i = 0;  i < 11;  i++)
                    {
                        xs[i] = x[i0 + i];
                        ca[i] = c[3*(i0 + i)];
                        cb[i] = c[3*(i0 + i) + 1];
                    }
                    SPG_UNROLL
                    for (int i = 0;  i < 11;  i++)
                        acc += xs[i].x*ca[i] - xs[i].y*cb[i];
                }
                float z = acc + q.swap2(acc, 1);
                if (q.any(!(fabsf(z) < __builtin_inff()), 5))
                {
                    // not finite: again, with every term of the other part selected away instead of multiplied by zero
                    const int split = kEqLen - eq_step;
                    float acc2 = 0.0f;
                    for (int i = 0;  i < kEqLen;  i++)
                    {
                        int k = eq_step + i;
                        k = (k >= kEqLen)  ?  (k - kEqLen)  :  k;
                        const float2 xv = C.u[k];
                        const float p = xv.x*c[3*i] - xv.y*c[3*i + 1];
                        const bool mine = (role & 2)  ?  (i >= split)  :  (i < split);
                        acc2 += mine  ?  p  :  0.0f;
                    }
                    const float z2 = acc2 + q.swap2(acc2, 2);
                    if (!(fabsf(z) < __builtin_inff()))
                        z = z2;
                }
                zre = q.template bcast<0>(z, 9);
                zim = q.template bcast<1>(z, 10);
            }

            SPG_PROF_STAMP(6);
            float tre;
            float tim;
            do_track = false;
            do_tune = false;
            do_save = false;
            int cs = 0;
            if (stage == V17_NORMAL  ||  stage == V17_TCM_WINDUP  ||  stage == V17_TEST_ONES)
                cs = decode_baud(zre, zim);
            switch (stage)
            {
            case V17_NORMAL:
                break;
            case V17_SYMBOL_ACQUISITION:
                if (++training_count >= 100)
                {
                    stage = V17_LOG_PHASE;
                    for (int k = 0;  k < 16;  k++)
                        diff_st(k, 0);
                    last_angle0 = v29_arctan2(zim, zre);
                    if (agc_scaling_save == 0.0f)
                        agc_scaling_save = agc_scaling;
                }
                break;
            case V17_LOG_PHASE:
            {
                int32_t angle = v29_arctan2(zim, zre);
                training_count = 1;
                if (short_train)
                {
                    if ((uint32_t) angle - (uint32_t) last_angle0 < (uint32_t) V17_DDS_PHASE(180.0f))
                    {
                        angle = last_angle0;
                        last_angle0 = V17_DDS_PHASE(270.0f + 18.433f);
                        last_angle1 = V17_DDS_PHASE(180.0f + 18.433f);
                    }
                    else
                    {
                        last_angle0 = V17_DDS_PHASE(180.0f + 18.433f);
                        last_angle1 = V17_DDS_PHASE(270.0f + 18.433f);
                    }
                    carrier_track_p = 500000.0f;
                    spin((uint32_t) angle - (uint32_t) V17_DDS_PHASE(180.0f + 18.433f));
                    stage = V17_SHORT_WAIT_FOR_CDBA;
                }
                else
                {
                    last_angle1 = angle;
                    stage = V17_WAIT_FOR_CDBA;
                }
                break;
            }
            case V17_WAIT_FOR_CDBA:
            {
                const int32_t angle = v29_arctan2(zim, zre);
                int i = training_count + 1;
                const int32_t prev = (i & 1)  ?  last_angle1  :  last_angle0;
                int32_t ang = (int32_t) ((uint32_t) angle - (uint32_t) prev);
                if (i & 1)
                    last_angle1 = angle;
                else
                    last_angle0 = angle;
                diff_st(i, (int32_t) ((uint32_t) diff_ld(i - 2) + (uint32_t) (ang >> 4)));
                if ((ang > V17_DDS_PHASE(90.0f)  ||  ang < V17_DDS_PHASE(-90.0f))  &&  training_count >= 13)
                {
                    i = (training_count - 8) & ~1;
                    if (i > 1)
                    {
                        const int jj = i & 0xF;
                        ang = (int32_t) ((uint32_t) diff_ld(jj) + (uint32_t) diff_ld(jj | 1))/(i - 1);
                        carrier_phase_rate += 3*16*(ang/20);
                    }
                    if (carrier_phase_rate < v29_f2i((1800.0f - 20.0f)*65536.0f*65536.0f/8000)
                        ||  carrier_phase_rate > v29_f2i((1800.0f + 20.0f)*65536.0f*65536.0f/8000))
                    {
                        park(true);
                        break;
                    }
                    spin((uint32_t) angle - (uint32_t) V17_DDS_PHASE(18.433f));
                    descramble(1);
                    descramble(1);
                    training_count = 1;
                    stage = V17_COARSE_TRAIN_ON_CDBA;
                    emit(-3);                           // SIG_STATUS_TRAINING_IN_PROGRESS
                    break;
                }
                if (++training_count > 256)
                    park(true);
                break;
            }
            case V17_COARSE_TRAIN_ON_CDBA:
            {
                int bit = descramble(1);
                bit = (bit << 1) | descramble(1);
                cdba(bit, tre, tim);
                track_carrier(tre, tim);
                tune_equalizer(tre, tim);
                const float ere = zre - tre;
                const float eim = zim - tim;
                training_error = ere*ere + eim*eim;
                if (++training_count == 2976 - 2000  ||  training_error < 1.0f*1.0f  ||  training_error > 200.0f*1.0f)
                {
                    eq_delta = 0.1f*(0.21f/kEqLen);
                    carrier_track_i = 1000.0f;
                    stage = V17_FINE_TRAIN_ON_CDBA;
                }
                break;
            }
            case V17_FINE_TRAIN_ON_CDBA:
            {
                int bit = descramble(1);
                bit = (bit << 1) | descramble(1);
                cdba(bit, tre, tim);
                track_carrier(tre, tim);
                tune_equalizer(tre, tim);
                if (++training_count >= 2976 - 48)
                {
                    training_error = 0.0f;
                    carrier_track_i = 100.0f;
                    carrier_track_p = 500000.0f;
                    stage = V17_TRAIN_ON_CDBA_AND_TEST;
                }
                break;
            }
            case V17_TRAIN_ON_CDBA_AND_TEST:
            {
                int bit = descramble(1);
                bit = (bit << 1) | descramble(1);
                cdba(bit, tre, tim);
                if (++training_count < 2976 - 20)
                {
                    track_carrier(tre, tim);
                    tune_equalizer(tre, tim);
                    const float ere = zre - tre;
                    const float eim = zim - tim;
                    training_error += (ere*ere + eim*eim);
                }
                else if (training_count >= 2976)
                {
                    if (training_error < 20.0f*1.414f*spacing)
                    {
                        training_error = 0.0f;
                        training_count = 0;
                        stage = V17_BRIDGE;
                    }
                    else
                    {
                        park(true);
                    }
                }
                break;
            }
            case V17_BRIDGE:
                descramble(0x8880 >> ((training_count & 0x7) << 1));
                descramble(0x8880 >> (((training_count & 0x7) << 1) + 1));
                if (++training_count >= 64)
                {
                    training_error = 0.0f;
                    training_count = 0;
                    if (bits_per_symbol == 2)
                    {
                        diff = short_train  ?  0  :  1;
                        stage = V17_TEST_ONES;
                    }
                    else
                    {
                        stage = V17_TCM_WINDUP;
                    }
                }
                break;
            case V17_SHORT_WAIT_FOR_CDBA:
            {
                const int32_t angle = v29_arctan2(zim, zre);
                const int32_t prev = (training_count & 1)  ?  last_angle1  :  last_angle0;
                const int32_t ang = (int32_t) ((uint32_t) angle - (uint32_t) prev);
                if (ang > V17_DDS_PHASE(90.0f)  ||  ang < V17_DDS_PHASE(-90.0f))
                {
                    descramble(1);
                    descramble(1);
                    training_error = 0.0f;
                    training_count = 1;
                    stage = V17_SHORT_TRAIN_ON_CDBA_AND_TEST;
                }
                else
                {
                    cdba((training_count & 1) + 2, tre, tim);
                    track_carrier(tre, tim);
                    if (++training_count > 256)
                        park(false);
                }
                break;
            }
            case V17_SHORT_TRAIN_ON_CDBA_AND_TEST:
            {
                int bit = descramble(1);
                bit = (bit << 1) | descramble(1);
                cdba(bit, tre, tim);
                track_carrier(tre, tim);
                if (training_count > 8)
                {
                    const float ere = zre - tre;
                    const float eim = zim - tim;
                    training_error += (ere*ere + eim*eim);
                }
                if (++training_count >= 38)
                {
                    carrier_track_i = 100.0f;
                    carrier_track_p = 500000.0f;
                    if (training_error < (38 - 8)*4.0f*1.0f*spacing)
                    {
                        training_count = 0;
                        if (bits_per_symbol == 2)
                        {
                            diff = short_train  ?  0  :  1;
                            training_error = 0.0f;
                            stage = V17_TEST_ONES;
                        }
                        else
                        {
                            stage = V17_TCM_WINDUP;
                        }
                        emit(-3);
                    }
                    else
                    {
                        park(false);
                    }
                }
                break;
            }
            case V17_TCM_WINDUP:
            {
                const float ere = zre - T.con[2*cs];
                const float eim = zim - T.con[2*cs + 1];
                training_error += (ere*ere + eim*eim);
                if (++training_count >= 15)
                {
                    training_error = 0.0f;
                    training_count = 0;
                    diff = short_train  ?  0  :  1;
                    stage = V17_TEST_ONES;
                }
                break;
            }
            case V17_TEST_ONES:
            {
                const float ere = zre - T.con[2*cs];
                const float eim = zim - T.con[2*cs + 1];
                training_error += (ere*ere + eim*eim);
                if (++training_count >= 48)
                {
                    if (training_error < 48*1.0f*1.0f*spacing)
                    {
                        emit(-4);                       // SIG_STATUS_TRAINING_SUCCEEDED
                        signal_present = 60;
                        do_save = true;                 // taps and carrier rate, once this baud's updates are in
                        short_train = 1;
                        stage = V17_NORMAL;
                    }
                    else
                    {
                        park(!short_train);
                    }
                }
                break;
            }
            default:
                break;
            }
            if (do_track)
            {
                const float error = zim*tgt_re - zre*tgt_im;
                carrier_phase_rate += v29_f2i(use_track_i*error);
                carrier_phase += (uint32_t) v29_f2i(use_track_p*error);
            }
            SPG_PROF_STAMP(7);
            q.sync(6);
            if (do_tune)
            {
                // cvec_circular_lmsf: tap i goes with the entry i places on from the ring position; lane r takes taps r, r + 4, ...
                const float ere = (tgt_re - zre)*use_delta;
                const float eim = (tgt_im - zim)*use_delta;
                SPG_UNROLL
                for (int j = 0;  j < (kEqLen + 3)/4;  j++)
                {
                    const int i = role + 4*j;
                    if (i < kEqLen)
                    {
                        int k = eq_step + i;
                        k = (k >= kEqLen)  ?  (k + kEqLen)  :  k;           // into the second copy of B
                        const float2 xv = C.u[k];
                        const f32x2v c0 = {C.taps[3*i], C.taps[3*i + 1]};
                        // {xi*eim + xr*ere, xr*eim - xi*ere}
                        const f32x2v u = (f32x2v) {xv.y, xv.x}*(f32x2v) {eim, eim};
                        const f32x2v w = (f32x2v) {xv.x, xv.y}*(f32x2v) {ere, ere};
                        const f32x2v c = c0*(f32x2v) {0.9999f, 0.9999f} + (u + (f32x2v) {w.x, -w.y});
                        C.taps[3*i] = c.x;
                        C.taps[3*i + 1] = c.y;
                        C.taps[3*i + 2] = -c.x;
                    }
                }
            }
            SPG_PROF_STAMP(8);
            q.sync(7);
            if (do_save)
            {
                carrier_phase_rate_save = carrier_phase_rate;
                for (int k = role;  k < kEqLen;  k += 4)
                {
                    stf(VF_EQ_SAVE + 2*k, C.taps[3*k]);
                    stf(VF_EQ_SAVE + 2*k + 1, C.taps[3*k + 1]);
                }
            }
            carrier_phase += (uint32_t) carrier_phase_rate;     // dds_advancef() with the rate the baud left behind
        }
    } while (q.any(pos < tn, 11));
    }

    // ---- write back (arrays dealt over the lanes, scalars by the first) ------------------------------------------------
    SPG_PROF_STAMP(9);
    SPG_PROF_FLUSH();
    q.sync(8);
    for (int i = role;  i < kRrcLen;  i += 4)
        stf(VF_RRC + i, C.rrc[i].x);
    for (int i = role;  i < kEqLen;  i += 4)
    {
        stf(VF_EQ_COEFF + 2*i, C.taps[3*i]);
        stf(VF_EQ_COEFF + 2*i + 1, C.taps[3*i + 1]);
        const float2 x = C.u[i];
        stf(VF_EQ_BUF + 2*i, x.x);
        stf(VF_EQ_BUF + 2*i + 1, x.y);
    }
    for (int t = role;  t < 16;  t += 4)
    {
        const uint32_t p = PAST(t);
        const uint32_t f0 = FULL(t, 0);
        const uint32_t f1 = FULL(t, 1);
        for (int i = 0;  i < 8;  i++)
        {
            sti(XI_PAST_STATE + 8*t + i, (int32_t) ((p >> (3*i)) & 7u));
            sti(XI_FULL_PATH + 8*t + i, (int32_t) (((i < 4  ?  f0  :  f1) >> (8*(i & 3))) & 0xFFu));
        }
    }
    if (role == 0)
    {
        stf(VF_AGC, agc_scaling);
        stf(VF_AGC_SAVE, agc_scaling_save);
        stf(VF_EQ_DELTA, eq_delta);
        stf(VF_TRAIN_ERR, training_error);
        stf(VF_TRACK_P, carrier_track_p);
        stf(VF_TRACK_I, carrier_track_i);
        stf(VF_GLOW, glow0);
        stf(VF_GLOW + 1, glow1);
        stf(VF_GHIGH, ghigh0);
        stf(VF_GHIGH + 1, ghigh1);
        stf(VF_GDC, gdc0);
        stf(VF_GDC + 1, gdc1);
        stf(VF_BAUD_PHASE, baud_phase);
        SPG_UNROLL
        for (int i = 0;  i < 8;  i++)
            stf(XF_DIST + i, sd[i]);
        sti(XI_RRC_STEP, rrc_step);
        sti(XI_DIFF, diff);
        sti(XI_SCRAMBLE, (int32_t) scramble_reg);
        sti(XI_SHORT_TRAIN, short_train);
        sti(XI_STAGE, stage);
        sti(XI_TRAIN_COUNT, training_count);
        sti(XI_LAST_SAMPLE, last_sample);
        sti(XI_SIGNAL_PRESENT, signal_present);
        sti(XI_DROP_PENDING, drop_pending);
        sti(XI_LOW_SAMPLES, low_samples);
        sti(XI_HIGH_SAMPLE, high_sample);
        sti(XI_CARRIER_PHASE, (int32_t) carrier_phase);
        sti(XI_PHASE_RATE, carrier_phase_rate);
        sti(XI_PHASE_RATE_SAVE, carrier_phase_rate_save);
        sti(XI_POWER, power_reading);
        sti(XI_EQ_STEP, eq_step);
        sti(XI_EQ_PUT_STEP, eq_put_step);
        sti(XI_EQ_SKIP, eq_skip);
        sti(XI_BAUD_HALF, baud_half);
        sti(XI_LAST_ANGLES, last_angle0);
        sti(XI_LAST_ANGLES + 1, last_angle1);
        sti(XI_TRELLIS_PTR, trellis_ptr);
        sti(XI_TOTAL_CORR, total_corr);
        L.ev_count[ch] = n_ev;
    }
#undef PAST
#undef FULL
#undef DIST
}

#if !defined(SPG_HOST_EMUL)

// CPW channels per wave (4*CPW live lanes), WPB waves per workgroup sharing the tables.
template <int CPW, int WPB>
__global__ __launch_bounds__(64*WPB)
void v17_quad_kernel(const V17Launch L)
{
    __shared__ V17QuadTables T;
    __shared__ uint32_t s_pcm[WPB*CPW*kQuadPcmStride];
    __shared__ float2 s_rrc[WPB*CPW*kQuadRrcStride];
    __shared__ float2 s_u[WPB*CPW*kQuad17EqStride];
    __shared__ float s_taps[WPB*CPW*kQuadTapStride];
    __shared__ uint32_t s_trellis[WPB*CPW*kQuad17TrellisStride];
    v17_quad_tables(T, *L.tab, (int) threadIdx.x, 64*WPB);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = (int) (threadIdx.x >> 6);
    const int cw = lane >> 2;
    const int ch = (blockIdx.x*WPB + wv)*CPW + cw;
    if (cw >= CPW  ||  ch >= L.n_ch)
        return;
    QuadDev q{lane & 3};
    const int slot = wv*CPW + cw;
    const V17QuadChan C = {s_pcm + slot*kQuadPcmStride, s_rrc + slot*kQuadRrcStride, s_u + slot*kQuad17EqStride,
                           s_taps + slot*kQuadTapStride, s_trellis + slot*kQuad17TrellisStride};
    v17_quad_run(q, L, ch, T, C);
}

#endif

}   // namespace spg
