// v27ter_quad.hpp -- the V.27ter receiver with FOUR LANES PER CHANNEL (reference: src/v27ter_rx.c:197-1028; state word
// map and tables: v27ter_common.hpp; the lane group: quad_ctx.hpp, the reasons for it: v29_quad.hpp).
//
// V.27ter has no per-sample filter (its symbol timing is a Gardner detector on the equaliser's delay line), so between
// T/2 instants a sample costs only the carrier detector.  A round of the main loop is one baud of a channel, taken as
// the one-lane kernel takes it -- every channel runs its own samples up to ITS next T/2 instant, then all channels of
// the wave take the T/2 instant together, twice, then the baud -- and the lanes of a channel share the work where
// the reference's arithmetic leaves a choice:
//   * at a T/2 instant the pulse shaping filter runs twice, real and imaginary (27 taps each, one chain of packed
//     multiply-adds: the two parts of vec_circular_dot_prodf() in the halves of a packed register): even lanes form the
//     real sum, odd lanes the imaginary one, and they swap;
//   * the equaliser's complex inner product (32 taps) is four chains -- real / imaginary x the two parts of
//     cvec_circular_dot_prodf() -- one per lane; the delay line sits in LDS as [B | 0 | B], taps as {re, im, -re}
//     (v29_quad.hpp);
//   * the LMS update is dealt by taps, lane r takes taps r, r + 4, ...;
//   * carrier detection, AGC, the Gardner detector, slicer, descrambler and the training state machine are replicated.
#pragma once

#include "v27ter_common.hpp"
#include "v29_quad.hpp"

namespace spg {

struct V27QuadTables                                    // per workgroup, in LDS
{
    float2 rrc[kRrcLen*kV27MaxSets];                    // [tap][set] {re, im} of the bank's bit rate
    float sine[2048];
};

constexpr int kQuad27EqStride = 100;                    // pairs: [B | 0 | B] = 96 (+4); 200 words = 3*64 + 8

SPG_FN void v27_quad_tables(V27QuadTables &T, const V27Tables &TB, const bool fast, int tid, int n)
{
    const int sets = fast  ?  8  :  12;
    const float *sre = fast  ?  TB.re4800  :  TB.re2400;
    const float *sim = fast  ?  TB.im4800  :  TB.im2400;
    for (int i = tid;  i < sets*kRrcLen;  i += n)
    {
        const int set = i/kRrcLen;
        const int tap = i - set*kRrcLen;
        T.rrc[tap*kV27MaxSets + set] = make_float2(sre[i], sim[i]);
    }
    for (int i = tid;  i < 2048;  i += n)
        T.sine[i] = TB.sine[i];
}

template <class Q>
SPG_FN void v27_quad_run(Q &q, const V27Launch &L, const int ch, const V27QuadTables &T, const V29QuadChan C)
{
    constexpr int EQN = kV27EqLen;
    const int role = q.role();
    const V27Tables &TB = *L.tab;
    const bool fast = (L.bit_rate == 4800);
    const int sets = fast  ?  8  :  12;
    const int put_add = fast  ?  8*5/2  :  12*20/(3*2);

    const size_t N = (size_t) L.n_ch;
    const int mylen = L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;
    auto ldf = [&](int w) { return __uint_as_float(L.state[(size_t) w*N + ch]); };
    auto ldi = [&](int w) { return (int32_t) L.state[(size_t) (kV27Floats + w)*N + ch]; };
    // a NaN goes back as x86's (v29_dev.hpp)
    auto stf = [&](int w, float v) { L.state[(size_t) w*N + ch] = (v != v)  ?  0xFFC00000u  :  __float_as_uint(v); };
    auto sti = [&](int w, int32_t v) { L.state[(size_t) (kV27Floats + w)*N + ch] = (uint32_t) v; };

    // ---- state: scalars replicated in the four lanes, arrays into LDS (dealt over the lanes) ---------------------------
    float agc_scaling = ldf(WF_AGC);
    float agc_scaling_save = ldf(WF_AGC_SAVE);
    const float eq_delta = ldf(WF_EQ_DELTA);
    float training_error = ldf(WF_TRAIN_ERR);
    float carrier_track_p = ldf(WF_TRACK_P);
    float carrier_track_i = ldf(WF_TRACK_I);
    for (int i = role;  i < kRrcLen;  i += 4)
    {
        const float v = ldf(WF_RRC + i);
        C.rrc[i] = make_float2(v, 0.0f);
        C.rrc[kRrcLen + i] = make_float2(0.0f, v);
    }
    for (int i = role;  i < EQN;  i += 4)
    {
        const float cr = ldf(WF_EQ_COEFF + 2*i);
        const float ci = ldf(WF_EQ_COEFF + 2*i + 1);
        C.taps[3*i] = cr;
        C.taps[3*i + 1] = ci;
        C.taps[3*i + 2] = -cr;
        const float2 x = make_float2(ldf(WF_EQ_BUF + 2*i), ldf(WF_EQ_BUF + 2*i + 1));
        C.u[i] = x;
        C.u[EQN + i] = make_float2(0.0f, 0.0f);
        C.u[2*EQN + i] = x;
    }
    int rrc_step = ldi(WI_RRC_STEP);
    uint32_t scramble_reg = (uint32_t) ldi(WI_SCRAMBLE);
    int pattern_count = ldi(WI_PATTERN_COUNT);
    int training_bc = ldi(WI_TRAINING_BC);
    int stage = ldi(WI_STAGE);
    int training_count = ldi(WI_TRAIN_COUNT);
    int last_sample = ldi(WI_LAST_SAMPLE);
    int signal_present = ldi(WI_SIGNAL_PRESENT);
    int drop_pending = ldi(WI_DROP_PENDING);
    int low_samples = ldi(WI_LOW_SAMPLES);
    int high_sample = ldi(WI_HIGH_SAMPLE);
    int constellation_state = ldi(WI_CONSTEL);
    uint32_t carrier_phase = (uint32_t) ldi(WI_CARRIER_PHASE);
    int32_t carrier_phase_rate = ldi(WI_PHASE_RATE);
    int32_t carrier_phase_rate_save = ldi(WI_PHASE_RATE_SAVE);
    int32_t power_reading = ldi(WI_POWER);
    const int32_t carrier_on_power = ldi(WI_ON_POWER);
    const int32_t carrier_off_power = ldi(WI_OFF_POWER);
    int eq_step = ldi(WI_EQ_STEP);
    int eq_put_step = ldi(WI_EQ_PUT_STEP);
    int eq_skip = ldi(WI_EQ_SKIP);
    int baud_half = ldi(WI_BAUD_HALF);
    int gardner_integrate = ldi(WI_GARDNER_INT);
    int gardner_step = ldi(WI_GARDNER_STEP);
    int total_corr = ldi(WI_TOTAL_CORR);
    int32_t last_angle0 = ldi(WI_LAST_ANGLES);
    int32_t last_angle1 = ldi(WI_LAST_ANGLES + 1);
    auto diff_ld = [&](int k) { return ldi(WI_DIFF_ANGLES + (k & 0xF)); };
    auto diff_st = [&](int k, int32_t v) { sti(WI_DIFF_ANGLES + (k & 0xF), v); };

    int8_t *evp = L.events + (size_t) ch*L.ev_cap;
    int n_ev = 0;
    auto emit = [&](int v)
    {
        if (role == 0  &&  n_ev < L.ev_cap)
            evp[n_ev] = (int8_t) v;
        n_ev++;
    };

    // v27ter_rx_restart() as the receive path reaches it (v27ter_rx.c:1091-1160; s->old_train is never set): all four
    // lanes, the same stores
    auto restart = [&]()
    {
        for (int i = 0;  i < 2*kRrcLen;  i++)
            C.rrc[i] = make_float2(0.0f, 0.0f);
        training_error = 0.0f;
        rrc_step = 0;
        scramble_reg = 0x3C;
        pattern_count = 0;
        stage = V27_SYMBOL_ACQUISITION;
        training_bc = 0;
        training_count = 0;
        signal_present = 0;
        high_sample = 0;
        low_samples = 0;
        drop_pending = 0;
        for (int k = 0;  k < 16;  k++)
            diff_st(k, 0);
        carrier_phase = 0;
        carrier_track_i = 200000.0f;
        carrier_track_p = 10000000.0f;
        power_reading = 0;
        constellation_state = 0;
        carrier_phase_rate = v29_f2i(1800.0f*65536.0f*65536.0f/8000);
        agc_scaling = (1.414f/1.000000f)/283.0f;
        for (int i = 0;  i < EQN;  i++)
        {
            const float cr = (i == 17)  ?  1.414f  :  0.0f;                 // V27TER_EQUALIZER_PRE_LEN + 1
            C.taps[3*i] = cr;
            C.taps[3*i + 1] = 0.0f;
            C.taps[3*i + 2] = -cr;
        }
        for (int i = 0;  i < 3*EQN;  i++)
            C.u[i] = make_float2(0.0f, 0.0f);
        eq_put_step = put_add;
        eq_step = 0;
        eq_skip = 0;
        last_sample = 0;
        gardner_integrate = 0;
        total_corr = 0;
        gardner_step = 512;
        baud_half = 0;
    };

    bool do_track = false;
    bool do_tune = false;
    bool do_save = false;
    float tgt_re = 0.0f;
    float tgt_im = 0.0f;
    float use_track_i = 0.0f;
    float use_track_p = 0.0f;
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
    };
    // v27ter_rx.c:380-414
    auto descramble = [&](int in_bit)
    {
        const bool training = (stage > V27_NORMAL  &&  stage < V27_TEST_ONES);
        in_bit &= 1;
        int out_bit = (in_bit ^ (int) (scramble_reg >> 5) ^ (int) (scramble_reg >> 6)) & 1;
        const uint32_t m = ((scramble_reg >> 7) ^ (uint32_t) in_bit) & ((scramble_reg >> 8) ^ (uint32_t) in_bit)
                         & ((scramble_reg >> 11) ^ (uint32_t) in_bit) & 1u;
        const bool guard = (pattern_count >= 33);
        out_bit ^= guard  ?  1  :  0;
        pattern_count = (guard  ||  training  ||  m)  ?  0  :  (pattern_count + 1);
        scramble_reg = (scramble_reg << 1) | (uint32_t) (training  ?  out_bit  :  in_bit);
        return out_bit;
    };
    auto put_bit = [&](int bit)
    {
        const int out_bit = descramble(bit);
        if (stage == V27_NORMAL)
            emit(out_bit);
    };
    auto target_of = [&](int k, float &tre, float &tim)
    {
        // v27ter_constellation[8], v27ter_rx.c:125-134
        const float mag = (k & 1)  ?  1.0f  :  1.414f;
        const int qd = k >> 1;                              // 0: +re, 1: +im, 2: -re, 3: -im (even k); diagonals for odd k
        if (k & 1)
        {
            tre = (qd == 0  ||  qd == 3)  ?  1.0f  :  -1.0f;
            tim = (qd == 0  ||  qd == 1)  ?  1.0f  :  -1.0f;
        }
        else
        {
            tre = (qd == 0)  ?  mag  :  (qd == 2)  ?  -mag  :  0.0f;
            tim = (qd == 1)  ?  mag  :  (qd == 3)  ?  -mag  :  0.0f;
        }
    };
    // v27ter_rx.c:441-484
    auto decode_baud = [&](float zre, float zim)
    {
        int nearest;
        if (!fast)
        {
            const int b1 = (zim > zre);
            const int b2 = (zim < -zre);
            nearest = (b2 << 1) | (b1 ^ b2);
            const int raw_bits = (0x1320 >> (4*((nearest - constellation_state) & 3))) & 0xF;      // {0, 2, 3, 1}
            put_bit(raw_bits);
            put_bit(raw_bits >> 1);
            constellation_state = nearest;
            nearest <<= 1;
        }
        else
        {
            const float abs_re = fabsf(zre);
            const float abs_im = fabsf(zim);
            if (abs_im*1.0f > abs_re*0.4142136f  &&  abs_im*1.0f < abs_re*2.4142136f)
            {
                const int b1 = (zre < 0.0f);
                const int b2 = (zim < 0.0f);
                nearest = (b2 << 2) | ((b1 ^ b2) << 1) | 1;
            }
            else
            {
                const int b1 = (zim > zre);
                const int b2 = (zim < -zre);
                nearest = (b2 << 2) | ((b1 ^ b2) << 1);
            }
            const int raw_bits = (int) ((0x51376204u >> (4*((nearest - constellation_state) & 7))) & 0xF);  // {4,0,2,6,7,3,1,5}
            put_bit(raw_bits);
            put_bit(raw_bits >> 1);
            put_bit(raw_bits >> 2);
            constellation_state = nearest;
        }
        float tre;
        float tim;
        target_of(nearest, tre, tim);
        track_carrier(tre, tim);
        if (--eq_skip <= 0)
        {
            eq_skip = 100;
            tune_equalizer(tre, tim);
        }
    };
    auto park = [&]()
    {
        stage = V27_PARKED;
        emit(-5);                                           // SIG_STATUS_TRAINING_FAILED
    };

    SPG_PROF_DECL();
    const int16_t *src = L.amp + (size_t) ch*L.stride;
    SPG_LOADS_DONE();
    q.sync(1);
    for (int tile = 0;  tile < L.samples;  tile += kV29QuadTile)
    {
    const int tn = max(0, min(kV29QuadTile, mylen - tile));
    // ---- stage the channel's stretch of PCM (16-byte pieces dealt over the lanes) ----
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
    for (;;)
    {
    // One round = one baud of every channel: two T/2 instants (a channel that enters the round in the middle of its baud
    // sits out the first), then the baud, with all channels in step.
    bool any_ready = false;
    bool restarted = false;
    bool baud_done = false;
    bool merged = false;
    // ---- The calm baud.  A channel at the start of a baud whose carrier is up and not about to drop sends every sample on,
    // so the whole baud follows from eq_put_step: k1 samples to its first T/2 instant, k2 more to its second (up to four
    // each), the step of the pulse shaper at either, where the delay line stands, the carrier phase.  Then the samples of
    // both halves are taken together -- lane r prepares samples r and r + 4, the carrier detector's recurrences run over
    // them in order on copies (if the power falls under the carrier-off threshold on any channel of the wave, or a channel
    // is elsewhere in its life, nothing of this is kept and the round goes the long way below) -- and the two instants'
    // shaping filters run side by side: lanes 0 / 1 the real and imaginary sum of the first, lanes 2 / 3 of the second.
    // The second half's samples overwrite the oldest entries of the first instant's window (the delay line is exactly one
    // window long), at most four of them: every lane reads the four oldest taps of its window before they are written.
    {
        const int E = eq_put_step;
        const int k1 = 1 + ((E > sets)  ?  1  :  0) + ((E > 2*sets)  ?  1  :  0) + ((E > 3*sets)  ?  1  :  0);
        const int E1 = E - k1*sets + put_add;
        const int k2 = 1 + ((E1 > sets)  ?  1  :  0) + ((E1 > 2*sets)  ?  1  :  0) + ((E1 > 3*sets)  ?  1  :  0);
        const int need = k1 + k2;
        const bool idle = (pos >= tn);
        const bool act = !idle  &&  (signal_present > 0)  &&  (drop_pending == 0)  &&  (stage != V27_PARKED)  &&  (stage != V27_SYMBOL_ACQUISITION)
                         &&  (baud_half == 0)  &&  (E <= 4*sets)  &&  (E1 <= 4*sets)  &&  (pos + need <= tn);
        if (!q.any(!(act  ||  idle), 40)  &&  q.any(act, 41))
        {
            // lane r: samples r and r + 4 of the baud
            const int cand0 = min(pos + role, kV29QuadTile - 1);
            const int cand1 = min(pos + role + 4, kV29QuadTile - 1);
            const uint32_t pw0 = C.pcm[cand0 >> 1];
            const uint32_t pw1 = C.pcm[cand1 >> 1];
            const int amp0 = (int) (short) ((cand0 & 1)  ?  (pw0 >> 16)  :  (pw0 & 0xFFFF));
            const int amp1 = (int) (short) ((cand1 & 1)  ?  (pw1 >> 16)  :  (pw1 & 0xFFFF));
            const int x0 = amp0 >> 1;
            const int x1 = amp1 >> 1;
            const int bef0 = q.prev1(x0, 42);
            const int bef1 = q.prev1(x1, 43);
            const int top0 = q.template bcast<3>(x0, 44);
            const int dif0 = (int) (short) (x0 - ((role == 0)  ?  last_sample  :  bef0));
            const int dif1 = (int) (short) (x1 - ((role == 0)  ?  top0  :  bef1));
            const int sq0 = dif0*dif0;
            const int sq1 = dif1*dif1;
            const int ad0 = (int) (short) abs(dif0);
            const int ad1 = (int) (short) abs(dif1);
            const int at0 = (ad0 << 3) + (ad0 << 1);
            const int at1 = (ad1 << 3) + (ad1 << 1);
            int t_pr = power_reading;
            int t_high = high_sample;
            int t_low = low_samples;
            int badf = 0;
            const int off1 = max(carrier_off_power, 1);
            const int m = act  ?  need  :  0;
            auto step = [&](const int k, const int sq, const int ad, const int ad10)
            {
                if (k < m)
                {
                    const int pwr = t_pr + ((sq - t_pr) >> 4);
                    badf |= (pwr < off1)  ?  1  :  0;
                    const bool low = (ad10 < t_high);
                    const int low_inc = t_low + 1;
                    const bool wipe = low  &&  (low_inc > 120);
                    t_pr = wipe  ?  0  :  pwr;
                    t_high = low  ?  (wipe  ?  0  :  t_high)  :  max(t_high, ad);
                    t_low = low  ?  (wipe  ?  0  :  low_inc)  :  0;
                }
            };
            step(0, q.template bcast<0>(sq0, 45), q.template bcast<0>(ad0, 46), q.template bcast<0>(at0, 47));
            step(1, q.template bcast<1>(sq0, 48), q.template bcast<1>(ad0, 49), q.template bcast<1>(at0, 50));
            step(2, q.template bcast<2>(sq0, 51), q.template bcast<2>(ad0, 52), q.template bcast<2>(at0, 53));
            step(3, q.template bcast<3>(sq0, 54), q.template bcast<3>(ad0, 55), q.template bcast<3>(at0, 56));
            step(4, q.template bcast<0>(sq1, 57), q.template bcast<0>(ad1, 58), q.template bcast<0>(at1, 59));
            step(5, q.template bcast<1>(sq1, 60), q.template bcast<1>(ad1, 61), q.template bcast<1>(at1, 62));
            if (q.any(m > 6, 63))
            {
                step(6, q.template bcast<2>(sq1, 64), q.template bcast<2>(ad1, 65), q.template bcast<2>(at1, 66));
                step(7, q.template bcast<3>(sq1, 67), q.template bcast<3>(ad1, 68), q.template bcast<3>(at1, 69));
            }
            if (!q.any(act  &&  badf != 0, 70))
            {
                merged = true;
                // the sample the baud ends on, for the next difference
                const int e0 = q.template bcast<0>((need > 4)  ?  x1  :  x0, 71);
                const int e1 = q.template bcast<1>((need > 5)  ?  x1  :  x0, 72);
                const int e2 = q.template bcast<2>((need > 6)  ?  x1  :  x0, 73);
                const int e3 = q.template bcast<3>((need > 7)  ?  x1  :  x0, 74);
                const int lastq = (need - 1) & 3;
                const int x_end = (lastq == 0)  ?  e0  :  (lastq == 1)  ?  e1  :  (lastq == 2)  ?  e2  :  e3;
                // the first half's samples into the delay line
                const int j0 = role;
                const int j1 = role + 4;
                int i0 = rrc_step + j0;
                i0 = (i0 >= kRrcLen)  ?  (i0 - kRrcLen)  :  i0;
                int i1 = rrc_step + j1;
                i1 = (i1 >= kRrcLen)  ?  (i1 - kRrcLen)  :  i1;
                const float f0 = (float) amp0;
                const float f1 = (float) amp1;
                if (act  &&  j0 < k1)
                {
                    C.rrc[i0].x = f0;
                    C.rrc[kRrcLen + i0].y = f0;
                }
                if (act  &&  j1 < k1)
                {
                    C.rrc[i1].x = f1;
                    C.rrc[kRrcLen + i1].y = f1;
                }
                q.sync(75);
                // this lane's instant: where the delay line stands then, the shaper's step, the carrier phase
                const bool second = (role & 2) != 0;
                const int taken = second  ?  need  :  k1;
                int rs = rrc_step + taken;
                rs = (rs >= kRrcLen)  ?  (rs - kRrcLen)  :  rs;
                const int e_aft = second  ?  (E1 - k2*sets)  :  (E - k1*sets);
                const int stp = min(-e_aft, sets - 1);
                const float *y = ((const float *) &T.rrc[act  ?  stp  :  0]) + (role & 1);
                const float2 *xw = &C.rrc[act  ?  rs  :  0];
                float2 xo[4];
                SPG_UNROLL
                for (int i = 0;  i < 4;  i++)
                    xo[i] = xw[i];
                q.sync(76);
                if (act  &&  j0 >= k1  &&  j0 < need)
                {
                    C.rrc[i0].x = f0;
                    C.rrc[kRrcLen + i0].y = f0;
                }
                if (act  &&  j1 >= k1  &&  j1 < need)
                {
                    C.rrc[i1].x = f1;
                    C.rrc[kRrcLen + i1].y = f1;
                }
                q.sync(77);
                float v;
                {
                    f32x2v a = {0.0f, 0.0f};
                    {
                        float ys[4];
                        SPG_UNROLL
                        for (int i = 0;  i < 4;  i++)
                            ys[i] = y[2*i*kV27MaxSets];
                        SPG_UNROLL
                        for (int i = 0;  i < 4;  i++)
                            a += (f32x2v) {xo[i].x, xo[i].y}*(f32x2v) {ys[i], ys[i]};
                    }
                    SPG_UNROLL
                    for (int i0b = 4;  i0b < kRrcLen;  i0b += 8)
                    {
                        float2 xs[8];
                        float ys[8];
                        SPG_UNROLL
                        for (int i = 0;  i < 8;  i++)
                        {
                            if (i0b + i < kRrcLen)
                            {
                                xs[i] = xw[i0b + i];
                                ys[i] = y[2*(i0b + i)*kV27MaxSets];
                            }
                        }
                        SPG_UNROLL
                        for (int i = 0;  i < 8;  i++)
                        {
                            if (i0b + i < kRrcLen)
                                a += (f32x2v) {xs[i].x, xs[i].y}*(f32x2v) {ys[i], ys[i]};
                        }
                    }
                    v = a.x + a.y;
                }
                const float s_mine = v*agc_scaling;
                const float s_other = q.swap1f(s_mine, 78);
                if (act)
                {
                    const float sre = (role & 1)  ?  s_other  :  s_mine;
                    const float sim = (role & 1)  ?  s_mine  :  s_other;
                    const uint32_t cp = carrier_phase + (uint32_t) (taken - 1)*(uint32_t) carrier_phase_rate;
                    const float dre = T.sine[(uint32_t) (cp + (1u << 30)) >> 21];
                    const float dim = T.sine[cp >> 21];
                    const float2 h = make_float2(sre*dre - sim*dim, -sre*dim - sim*dre);
                    const int e_at = (eq_step + (second  ?  1  :  0)) & (EQN - 1);
                    if ((role & 1) == 0)
                    {
                        C.u[e_at] = h;
                        C.u[2*EQN + e_at] = h;
                    }
                    power_reading = t_pr;
                    high_sample = t_high;
                    low_samples = t_low;
                    last_sample = x_end;
                    int rs2 = rrc_step + need;
                    rrc_step = (rs2 >= kRrcLen)  ?  (rs2 - kRrcLen)  :  rs2;
                    pos += need;
                    eq_put_step = E1 - k2*sets + put_add;
                    eq_step = (eq_step + 2) & (EQN - 1);
                    carrier_phase += (uint32_t) need*(uint32_t) carrier_phase_rate;
                    any_ready = true;
                    baud_done = true;
                }
                q.sync(79);
            }
        }
    }
    for (int half = merged  ?  2  :  0;  half < 2;  half++)
    {
    const bool take = (half == 1)  ||  (baud_half == 0);
    SPG_PROF_STAMP(0);
    // ---- phase A: every channel runs its own samples up to its next T/2 instant (replicated; selects, not branches) ----
    bool ready = false;
    bool restart_pending = false;
    int power = 0;
    while (q.any(take  &&  !ready  &&  !restart_pending  &&  pos < tn, 1))
    {
        const bool want = take  &&  !ready  &&  !restart_pending  &&  pos < tn;
        // -- The calm stretch.  A channel whose carrier is up and not about to drop (and that is not parked) sends every sample
        // on, so how many samples it is to the T/2 instant follows from eq_put_step alone: up to four of them are taken
        // together -- one per lane for the arithmetic that does not depend on the sample before, the power estimate in
        // order on copies -- and if the power stays above the carrier-off threshold on every channel of the wave, that
        // is the stretch.  Otherwise nothing of it is kept and the loop takes one sample the long way.
        {
            const int cand = min(pos + role, kV29QuadTile - 1);
            const uint32_t pwc = C.pcm[cand >> 1];
            const int amp_c = (int) (short) ((cand & 1)  ?  (pwc >> 16)  :  (pwc & 0xFFFF));
            const float my_ampf = (float) amp_c;
            const int my_x = amp_c >> 1;
            const int before = q.prev1(my_x, 2);
            const int dif = (int) (short) (my_x - ((role == 0)  ?  last_sample  :  before));
            const int my_sq = dif*dif;
            const int my_ad = (int) (short) abs(dif);
            const int my_ad10 = (my_ad << 3) + (my_ad << 1);
            const int x0 = q.template bcast<0>(my_x, 21);
            const int x1 = q.template bcast<1>(my_x, 22);
            const int x2 = q.template bcast<2>(my_x, 23);
            const int x3 = q.template bcast<3>(my_x, 24);
            const int sq0 = q.template bcast<0>(my_sq, 25);
            const int sq1 = q.template bcast<1>(my_sq, 26);
            const int sq2 = q.template bcast<2>(my_sq, 27);
            const int sq3 = q.template bcast<3>(my_sq, 28);
            const int ad0 = q.template bcast<0>(my_ad, 29);
            const int ad1 = q.template bcast<1>(my_ad, 30);
            const int ad2 = q.template bcast<2>(my_ad, 31);
            const int ad3 = q.template bcast<3>(my_ad, 32);
            const int adt0 = q.template bcast<0>(my_ad10, 33);
            const int adt1 = q.template bcast<1>(my_ad10, 34);
            const int adt2 = q.template bcast<2>(my_ad10, 35);
            const int adt3 = q.template bcast<3>(my_ad10, 36);
            const bool calm0 = (signal_present > 0)  &&  (drop_pending == 0)  &&  (stage != V27_PARKED);
            const int E = eq_put_step;
            const int kn = max(1, ((E > 0)  ?  1  :  0) + ((E > sets)  ?  1  :  0) + ((E > 2*sets)  ?  1  :  0) + ((E > 3*sets)  ?  1  :  0));
            const int m = want  ?  min(kn, min(tn - pos, 4))  :  0;
            int t_pr = power_reading;
            int t_high = high_sample;
            int t_low = low_samples;
            int t_power = power;
            int badf = 0;
            const int off1 = max(carrier_off_power, 1);
            auto calm_sample = [&](const int k, const int sq, const int ad, const int ad10)
            {
                if (k < m)
                {
                    const int pwr = t_pr + ((sq - t_pr) >> 4);
                    badf |= (pwr < off1)  ?  1  :  0;
                    const bool low = (ad10 < t_high);
                    const int low_inc = t_low + 1;
                    const bool wipe = low  &&  (low_inc > 120);
                    t_pr = wipe  ?  0  :  pwr;
                    t_high = low  ?  (wipe  ?  0  :  t_high)  :  max(t_high, ad);
                    t_low = low  ?  (wipe  ?  0  :  low_inc)  :  0;
                    t_power = pwr;
                }
            };
            calm_sample(0, sq0, ad0, adt0);
            calm_sample(1, sq1, ad1, adt1);
            calm_sample(2, sq2, ad2, adt2);
            calm_sample(3, sq3, ad3, adt3);
            if (!q.any(want  &&  (!calm0  ||  badf != 0), 9))
            {
                if (role < m)
                {
                    int idx = rrc_step + role;
                    idx = (idx >= kRrcLen)  ?  (idx - kRrcLen)  :  idx;
                    C.rrc[idx].x = my_ampf;
                    C.rrc[kRrcLen + idx].y = my_ampf;
                }
                if (m > 0)
                {
                    power_reading = t_pr;
                    high_sample = t_high;
                    low_samples = t_low;
                    power = t_power;
                    last_sample = (m >= 4)  ?  x3  :  (m == 3)  ?  x2  :  (m == 2)  ?  x1  :  x0;
                    int rs = rrc_step + m;
                    rrc_step = (rs >= kRrcLen)  ?  (rs - kRrcLen)  :  rs;
                    pos += m;
                    eq_put_step = E - sets*m;
                    ready = (eq_put_step <= 0);
                    carrier_phase += (uint32_t) (ready  ?  (m - 1)  :  m)*(uint32_t) carrier_phase_rate;
                }
                q.sync(11);
                continue;
            }
        }
        if (want)
        {
            const uint32_t pw = C.pcm[pos >> 1];
            const int amp = (int) (short) ((pos & 1)  ?  (pw >> 16)  :  (pw & 0xFFFF));
            pos++;
            // v27ter_rx(), v27ter_rx.c:862-1028: the sample into the delay line (the zero halves of the pairs stay)
            const float ampf = (float) amp;
            C.rrc[rrc_step].x = ampf;
            C.rrc[kRrcLen + rrc_step].y = ampf;
            rrc_step = (rrc_step == kRrcLen - 1)  ?  0  :  (rrc_step + 1);
            // signal_detect(), v27ter_rx.c:779-861 (IAXMODEM_STUFF is #defined at v27ter_rx.c:1)
            const int x = amp >> 1;
            const int diff = (int) (short) (x - last_sample);
            last_sample = x;
            const int pwr = power_reading + ((diff*diff - power_reading) >> 4);
            const int ad = (int) (short) abs(diff);
            const bool low = (10*ad < high_sample);
            const int low_inc = low_samples + 1;
            const bool wipe = low  &&  (low_inc > 120);
            power_reading = wipe  ?  0  :  pwr;
            high_sample = low  ?  (wipe  ?  0  :  high_sample)  :  max(high_sample, ad);
            low_samples = low  ?  (wipe  ?  0  :  low_inc)  :  0;
            // A channel whose carrier is up and stays up (and that is not parked) sends the sample on: when that holds for
            // every channel of the wave taking a sample now, the carrier detector's state machine has nothing to do
            const bool calm = (signal_present > 0)  &&  (drop_pending == 0)  &&  (stage != V27_PARKED)
                              &&  (pwr >= carrier_off_power)  &&  (pwr != 0);
            if (!q.any(!calm, 8))
            {
                power = pwr;
                eq_put_step -= sets;
                ready = (eq_put_step <= 0);
                carrier_phase += ready  ?  0u  :  (uint32_t) carrier_phase_rate;
            }
            else
            {
                const bool present = (signal_present > 0);
                const bool dropping = present  &&  ((drop_pending != 0)  ||  (pwr < carrier_off_power));
                const bool down = dropping  &&  (signal_present <= 1);
                const bool up = !present  &&  (pwr >= carrier_on_power);
                signal_present = up  ?  1  :  (dropping  ?  (signal_present - 1)  :  signal_present);
                drop_pending = up  ?  0  :  (dropping  ?  1  :  drop_pending);
                if (q.any(up  ||  down, 2))
                {
                    if (up)
                        emit(-2);                               // SIG_STATUS_CARRIER_UP
                    if (down)
                        emit(-1);                               // SIG_STATUS_CARRIER_DOWN
                }
                restart_pending = down;
                const bool acc = !down  &&  (present  ||  up)  &&  pwr != 0  &&  stage != V27_PARKED;
                power = acc  ?  pwr  :  power;
                eq_put_step -= acc  ?  sets  :  0;
                ready = acc  &&  (eq_put_step <= 0);
                carrier_phase += (acc  &&  !ready)  ?  (uint32_t) carrier_phase_rate  :  0u;
            }
        }
    }
    SPG_PROF_STAMP(2);
    if (q.any(restart_pending, 3))
    {
        if (restart_pending)
        {
            restart();
            restarted = true;
        }
    }
    q.sync(3);
    // ---- phase B: the T/2 instant, for all channels that reached one ----
    if (q.any(ready, 4))
    {
        if (ready)
        {
            any_ready = true;
            if (stage == V27_SYMBOL_ACQUISITION)
            {
                // fixed_sqrt32(), math_fixed.c:158-169
                int root_power;
                {
                    uint32_t xx = (uint32_t) power;
                    const int top = 31 - __builtin_clz(xx);
                    const int shift = 30 - (top & ~1);
                    xx <<= shift;
                    root_power = TB.sqrt_tab[((xx >> 24) & 0xFF) - 64] >> (shift >> 1);     // training only: from global memory
                }
                if (root_power == 0)
                    root_power = 1;
                agc_scaling = (1.414f/1.000000f)/(float) root_power;
            }
            const int step = min(-eq_put_step, sets - 1);
            // vec_circular_dot_prodf(rrc_filter, coeffs[step], 27, rrc_step): even lanes the real filter, odd the imaginary
            float v;
            {
                const float *y = ((const float *) &T.rrc[step]) + (role & 1);
                const float2 *xw = &C.rrc[rrc_step];
                f32x2v a = {0.0f, 0.0f};
                SPG_UNROLL
                for (int i0 = 0;  i0 < kRrcLen;  i0 += 9)
                {
                    float2 xs[9];
                    float ys[9];
                    SPG_UNROLL
                    for (int i = 0;  i < 9;  i++)
                    {
                        xs[i] = xw[i0 + i];
                        ys[i] = y[2*(i0 + i)*kV27MaxSets];
                    }
                    SPG_UNROLL
                    for (int i = 0;  i < 9;  i++)
                        a += (f32x2v) {xs[i].x, xs[i].y}*(f32x2v) {ys[i], ys[i]};
                }
                v = a.x + a.y;
            }
            const float s_mine = v*agc_scaling;
            const float s_other = q.swap1f(s_mine, 1);
            const float sre = (role & 1)  ?  s_other  :  s_mine;
            const float sim = (role & 1)  ?  s_mine  :  s_other;
            const float dre = T.sine[(uint32_t) (carrier_phase + (1u << 30)) >> 21];
            const float dim = T.sine[carrier_phase >> 21];
            const float hre = sre*dre - sim*dim;
            const float him = -sre*dim - sim*dre;
            eq_put_step += put_add;
            // ---- process_half_baud(), v27ter_rx.c:531-777: the new entry into the delay line (both copies) ----
            const float2 h = make_float2(hre, him);
            C.u[eq_step] = h;
            C.u[2*EQN + eq_step] = h;
            eq_step = (eq_step + 1) & (EQN - 1);
            baud_half ^= 1;
            baud_done = baud_done  ||  (baud_half == 0);
            carrier_phase += (uint32_t) carrier_phase_rate;
        }
    }
    SPG_PROF_STAMP(5);
    q.sync(4);
    }
    if (!q.any(any_ready  ||  restarted, 5))
        break;
    if (!q.any(baud_done, 6))
        continue;
    // ---- phase C: the baud, for every channel that completed one in this round ----
    if (baud_done)
    {
        carrier_phase -= (uint32_t) carrier_phase_rate;
        {
            // symbol_sync(), v27ter_rx.c:486-528: the three newest entries of the delay line
            const float2 n1 = C.u[(eq_step - 1) & (EQN - 1)];
            const float2 n2 = C.u[(eq_step - 2) & (EQN - 1)];
            const float2 n3 = C.u[(eq_step - 3) & (EQN - 1)];
            float p = n3.x - n1.x;
            p *= n2.x;
            float qv = n3.y - n1.y;
            qv *= n2.y;
            gardner_integrate += (p + qv > 0.0f)  ?  gardner_step  :  -gardner_step;
            if (abs(gardner_integrate) >= 128)
            {
                eq_put_step += gardner_integrate/128;
                total_corr += gardner_integrate/128;
                gardner_integrate = 0;
            }
        }
        // equalizer_get(): cvec_circular_dot_prodf, one chain per lane (v29_quad.hpp)
        float zre;
        float zim;
        {
            const float2 *x = &C.u[eq_step + ((role & 2)  ?  EQN  :  0)];
            const float *c = &C.taps[role & 1];
            float acc = 0.0f;
            SPG_UNROLL
            for (int i0 = 0;  i0 < EQN;  i0 += 8)
            {
                float2 xs[8];
                float ca[8];
                float cb[8];
                SPG_UNROLL
                for (int i = 0;  i < 8;  i++)
                {
                    xs[i] = x[i0 + i];
                    ca[i] = c[3*(i0 + i)];
                    cb[i] = c[3*(i0 + i) + 1];
                }
                SPG_UNROLL
                for (int i = 0;  i < 8;  i++)
                    acc += xs[i].x*ca[i] - xs[i].y*cb[i];
            }
            float z = acc + q.swap2(acc, 1);
            if (q.any(!(fabsf(z) < __builtin_inff()), 7))
            {
                // not finite: again, with every term of the other part selected away instead of multiplied by zero
                const int split = EQN - eq_step;
                float acc2 = 0.0f;
                for (int i = 0;  i < EQN;  i++)
                {
                    const float2 xv = C.u[(eq_step + i) & (EQN - 1)];
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
        do_track = false;
        do_tune = false;
        do_save = false;
        if (stage == V27_NORMAL  ||  stage == V27_TEST_ONES)
            decode_baud(zre, zim);
        switch (stage)
        {
        case V27_NORMAL:
            break;
        case V27_SYMBOL_ACQUISITION:
            if (++training_count >= 30)
            {
                gardner_step = 32;
                stage = V27_LOG_PHASE;
                for (int k = 0;  k < 16;  k++)
                    diff_st(k, 0);
                last_angle0 = v29_arctan2(zim, zre);
            }
            break;
        case V27_LOG_PHASE:
            last_angle1 = v29_arctan2(zim, zre);
            training_count = 1;
            stage = V27_WAIT_FOR_HOP;
            break;
        case V27_WAIT_FOR_HOP:
        {
            int32_t angle = v29_arctan2(zim, zre);
            int i = training_count + 1;
            const int32_t prev = (i & 1)  ?  last_angle1  :  last_angle0;
            int32_t ang = (int32_t) ((uint32_t) angle - (uint32_t) prev);
            if (i & 1)
                last_angle1 = angle;
            else
                last_angle0 = angle;
            diff_st(i, (int32_t) ((uint32_t) diff_ld(i - 2) + (uint32_t) (ang >> 4)));
            if ((ang > 0x20000000  ||  ang < (int32_t) 0xE0000000u)  &&  training_count >= 13)
            {
                i = (training_count - 8) & ~1;
                if (i > 1)
                {
                    const int jj = i & 0xF;
                    ang = (int32_t) ((uint32_t) diff_ld(jj) + (uint32_t) diff_ld(jj | 1))/(i - 1);
                    if (fast)
                        carrier_phase_rate += 16*(ang/10);
                    else
                        carrier_phase_rate += 3*16*(ang/40);
                }
                if (carrier_phase_rate < v29_f2i((1800.0f - 20.0f)*65536.0f*65536.0f/8000)
                    ||  carrier_phase_rate > v29_f2i((1800.0f + 20.0f)*65536.0f*65536.0f/8000))
                {
                    park();
                    break;
                }
                angle = (int32_t) ((uint32_t) angle + 0x80000000u);
                const float p = ((uint32_t) angle)*2.0f*3.1415926f/(65536.0f*65536.0f);
                const float zc = spg_sincosf(p, true);
                const float zs = -spg_sincosf(p, false);
                for (int k = role;  k < EQN;  k += 4)
                {
                    const float2 xv = C.u[k];
                    const float2 r = make_float2(xv.x*zc - xv.y*zs, xv.x*zs + xv.y*zc);
                    C.u[k] = r;
                    C.u[2*EQN + k] = r;
                }
                carrier_phase += (uint32_t) angle;
                gardner_step = 2;
                training_bc = 1;
                training_bc ^= descramble(1);
                descramble(1);
                descramble(1);
                constellation_state = training_bc  ?  4  :  0;
                training_count = 1;
                stage = V27_TRAIN_ON_ABAB;
                emit(-3);                           // SIG_STATUS_TRAINING_IN_PROGRESS
            }
            else if (++training_count > 50)
            {
                park();
            }
            break;
        }
        case V27_TRAIN_ON_ABAB:
        {
            training_bc ^= descramble(1);
            descramble(1);
            descramble(1);
            constellation_state = training_bc  ?  4  :  0;
            const float tre = training_bc  ?  -1.414f  :  1.414f;
            track_carrier(tre, 0.0f);
            tune_equalizer(tre, 0.0f);
            carrier_track_i = 400.0f + (200000.0f - 400.0f)*(float) (1074 - training_count)/(float) 1074;
            carrier_track_p = 1000000.0f + (10000000.0f - 1000000.0f)*(float) (1074 - training_count)/(float) 1074;
            if (++training_count >= 1074)
            {
                constellation_state = fast  ?  4  :  2;
                training_count = 0;
                stage = V27_TEST_ONES;
            }
            break;
        }
        case V27_TEST_ONES:
        {
            float tre;
            float tim;
            target_of(fast  ?  constellation_state  :  (constellation_state << 1), tre, tim);
            const float dre2 = zre - tre;
            const float dim2 = zim - tim;
            training_error += (dre2*dre2 + dim2*dim2);
            if (++training_count >= 8)
            {
                if (training_error < (fast  ?  8.0f*0.25f  :  8.0f*0.5f))
                {
                    emit(-4);                       // SIG_STATUS_TRAINING_SUCCEEDED
                    signal_present = fast  ?  90  :  120;
                    stage = V27_NORMAL;
                    do_save = true;                 // taps and carrier rate, once this baud's updates are in
                    agc_scaling_save = agc_scaling;
                }
                else
                {
                    park();
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
            const float ere = (tgt_re - zre)*eq_delta;
            const float eim = (tgt_im - zim)*eq_delta;
            SPG_UNROLL
            for (int j = 0;  j < EQN/4;  j++)
            {
                const int i = role + 4*j;
                const float2 xv = C.u[(eq_step + i) & (EQN - 1)];
                const f32x2v c0 = {C.taps[3*i], C.taps[3*i + 1]};
                // {xi*eim + xr*ere, xr*eim - xi*ere}
                const f32x2v u = (f32x2v) {xv.y, xv.x}*(f32x2v) {eim, eim};
                const f32x2v w = (f32x2v) {xv.x, xv.y}*(f32x2v) {ere, ere};
                const f32x2v cn = c0*(f32x2v) {0.9999f, 0.9999f} + (u + (f32x2v) {w.x, -w.y});
                C.taps[3*i] = cn.x;
                C.taps[3*i + 1] = cn.y;
                C.taps[3*i + 2] = -cn.x;
            }
        }
        SPG_PROF_STAMP(8);
        q.sync(7);
        if (do_save)
        {
            carrier_phase_rate_save = carrier_phase_rate;
            for (int k = role;  k < EQN;  k += 4)
            {
                stf(WF_EQ_SAVE + 2*k, C.taps[3*k]);
                stf(WF_EQ_SAVE + 2*k + 1, C.taps[3*k + 1]);
            }
        }
        carrier_phase += (uint32_t) carrier_phase_rate;     // dds_advancef() with the rate the baud left behind
    }
    }
    }

    SPG_PROF_STAMP(9);
    SPG_PROF_FLUSH();
    // ---- write back (arrays dealt over the lanes, scalars by the first) ----
    q.sync(8);
    for (int i = role;  i < kRrcLen;  i += 4)
        stf(WF_RRC + i, C.rrc[i].x);
    for (int i = role;  i < EQN;  i += 4)
    {
        stf(WF_EQ_COEFF + 2*i, C.taps[3*i]);
        stf(WF_EQ_COEFF + 2*i + 1, C.taps[3*i + 1]);
        const float2 x = C.u[i];
        stf(WF_EQ_BUF + 2*i, x.x);
        stf(WF_EQ_BUF + 2*i + 1, x.y);
    }
    if (role == 0)
    {
        stf(WF_AGC, agc_scaling);
        stf(WF_AGC_SAVE, agc_scaling_save);
        stf(WF_TRAIN_ERR, training_error);
        stf(WF_TRACK_P, carrier_track_p);
        stf(WF_TRACK_I, carrier_track_i);
        sti(WI_RRC_STEP, rrc_step);
        sti(WI_SCRAMBLE, (int32_t) scramble_reg);
        sti(WI_PATTERN_COUNT, pattern_count);
        sti(WI_TRAINING_BC, training_bc);
        sti(WI_STAGE, stage);
        sti(WI_TRAIN_COUNT, training_count);
        sti(WI_LAST_SAMPLE, last_sample);
        sti(WI_SIGNAL_PRESENT, signal_present);
        sti(WI_DROP_PENDING, drop_pending);
        sti(WI_LOW_SAMPLES, low_samples);
        sti(WI_HIGH_SAMPLE, high_sample);
        sti(WI_CONSTEL, constellation_state);
        sti(WI_CARRIER_PHASE, (int32_t) carrier_phase);
        sti(WI_PHASE_RATE, carrier_phase_rate);
        sti(WI_PHASE_RATE_SAVE, carrier_phase_rate_save);
        sti(WI_POWER, power_reading);
        sti(WI_EQ_STEP, eq_step);
        sti(WI_EQ_PUT_STEP, eq_put_step);
        sti(WI_EQ_SKIP, eq_skip);
        sti(WI_BAUD_HALF, baud_half);
        sti(WI_GARDNER_INT, gardner_integrate);
        sti(WI_GARDNER_STEP, gardner_step);
        sti(WI_TOTAL_CORR, total_corr);
        sti(WI_LAST_ANGLES, last_angle0);
        sti(WI_LAST_ANGLES + 1, last_angle1);
        L.ev_count[ch] = n_ev;
    }
}

#if !defined(SPG_HOST_EMUL)

template <int CPW, int WPB>
__global__ __launch_bounds__(64*WPB)
void v27ter_quad_kernel(const V27Launch L)
{
    __shared__ V27QuadTables T;
    __shared__ uint32_t s_pcm[WPB*CPW*kQuadPcmStride];
    __shared__ float2 s_rrc[WPB*CPW*kQuadRrcStride];
    __shared__ float2 s_u[WPB*CPW*kQuad27EqStride];
    __shared__ float s_taps[WPB*CPW*kQuadTapStride];
    v27_quad_tables(T, *L.tab, L.bit_rate == 4800, (int) threadIdx.x, 64*WPB);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = (int) (threadIdx.x >> 6);
    const int cw = lane >> 2;
    const int ch = (blockIdx.x*WPB + wv)*CPW + cw;
    if (cw >= CPW  ||  ch >= L.n_ch)
        return;
    QuadDev q{lane & 3};
    const int slot = wv*CPW + cw;
    const V29QuadChan C = {s_pcm + slot*kQuadPcmStride, s_rrc + slot*kQuadRrcStride, s_u + slot*kQuad27EqStride, s_taps + slot*kQuadTapStride};
    v27_quad_run(q, L, ch, T, C);
}

#endif

}   // namespace spg
