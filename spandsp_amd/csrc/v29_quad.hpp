// v29_quad.hpp -- the V.29 receiver with FOUR LANES PER CHANNEL (reference: src/v29rx.c:400-965, src/godard.c:144-220,
// src/vector_float.c:890-939, src/complex_vector_float.c:137-219; state word map and helpers: v29_common.hpp).
//
// Why: a bank of 16 384 channels (BASELINE configs[3]) is 16 channels per SIMD.  With one channel per lane
// (v29_dev.hpp) a lone wave per SIMD walks the whole receiver -- about 670 instructions per sample -- with 16 of its 64
// lanes alive, and the launch takes as long as that one instruction stream.  Here a channel owns a DPP quad, and the
// work of one baud is dealt over its four lanes wherever the reference's arithmetic leaves a choice:
//   * a ROUND of the main loop is one baud of a channel: its three or four samples are taken together.  Lane r owns
//     sample r of the round: it forms that sample's root raised cosine inner products, real and imaginary at once
//     (27 taps, two independent chains of packed multiply-adds on the same delay line window: the two parts of
//     vec_circular_dot_prodf() ride in the halves of a packed register, as in v29_dev.hpp).  The polyphase row of every
//     sample of a baud is known when the baud starts, because only the baud's own timing decision moves it;
//   * the four windows of a round differ by one sample each.  The delay line keeps the reference's 27 entry ring: the
//     three oldest taps of each lane's window are read BEFORE the round's samples are written over them, the other 24
//     after;
//   * the T/2 spaced equaliser input of the (two) T/2 instants of the round is formed by the lanes that own those
//     samples, side by side (carrier phase at a sample = phase at the start of the round + accepted samples x rate);
//   * the equaliser's complex inner product (33 taps) is four chains -- real / imaginary x the two parts of
//     cvec_circular_dot_prodf() -- one per lane, each in the reference's order.  The delay line sits in LDS in the
//     reference's ring order as [B | 0 | B | B]: a lane of the first part reads 33 entries from the ring position on and
//     runs into the zeros, a lane of the second part starts 33 further on and runs out of them (adding +0 is exact: a
//     sum that starts at +0 is never -0).  Taps are stored {re, im, -re}: the imaginary lanes read one word further on,
//     and the same three instructions (a*c - b*d) give re*re - im*im on one lane and re*im + im*re on its neighbour.
//     Should a product with a padding zero be NaN (a tap at infinity: the receiver has been fed garbage), the result is
//     not finite and the sum is redone with every term selected instead of padded;
//     (round 5: the taps a lane multiplies with live in its registers -- `tc` below -- and LDS keeps the copy the update's lanes
//     exchange through; what a lone wave pays for in such a sum is the LDS words it reads, tools/probe_lds.hip)
//   * the LMS update is independent per tap: lane r takes taps r, r + 4, ... -- or, when only a few channels of the wave are due
//     (calls that did not start together), the wave's lanes take a tap each of one such channel after the other;
//   * everything scalar (carrier detect, AGC, Godard filters, the training state machine, descrambler) is replicated
//     in the four lanes, so every decision is uniform over the quad and no value ever has to be sent back.
// Results are the reference's bit for bit: tests/test_quad_emul.py runs THIS source on the host (four fibers per
// channel, quad_ctx.hpp) against the oracle, tests/test_v29_gpu.py runs it on the GPU.
#pragma once

#include "v29_common.hpp"

namespace spg {

// Phase timing for the builder (tools/quad_prof.py; never in the product build): cycles between stamps, summed per wave.
#if defined(SPG_QUAD_PROF)  &&  !defined(SPG_HOST_EMUL)
__device__ unsigned long long spg_quad_prof[16];
#define SPG_PROF_DECL()     unsigned long long prof_last = __builtin_readcyclecounter(); unsigned int prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define SPG_PROF_STAMP(k)   do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[k] += (unsigned int) (now_ - prof_last); prof_last = now_; } while (0)
#define SPG_PROF_FLUSH()    do { if ((threadIdx.x & 63) == 0) { for (int k_ = 0;  k_ < 10;  k_++) atomicAdd(&spg_quad_prof[k_], (unsigned long long) prof_acc[k_]); } } while (0)
#else
#define SPG_PROF_DECL()     do { } while (0)
#define SPG_PROF_STAMP(k)   do { } while (0)
#define SPG_PROF_FLUSH()    do { } while (0)
#endif

constexpr int kV29QuadTile = 160;                       // samples of PCM staged per channel at a time (a whole frame)

struct V29QuadTables                                    // per workgroup, in LDS
{
    float2 rrc[kRrcLen*kRrcSets];                       // [tap][set] {re, im}
    float sine[2048];
    float konst[32];                                    // v29tx_constellation_maps.h:58-77
    uint8_t space_map[400];
};

// Per channel, in LDS.  The strides from channel to channel are chosen so that the lanes of a half wave (eight channels)
// that read the same element of their own channel fall into different banks: 4-byte accesses see 32 banks (strides of
// 4 and 17 words: a quad touches at most three consecutive words of the taps, one of the PCM), 8-byte accesses 64
// (strides of 8 x odd words: a quad touches up to four consecutive pairs).
constexpr int kQuadPcmStride = 81;                      // words: 160 samples of PCM (+1)
constexpr int kQuadRrcStride = 60;                      // pairs: 2 x 27 (+6); 120 words = 64 + 8*7
constexpr int kQuadEqStride = 4*kEqLen;                 // pairs: 264 words = 4*64 + 8
constexpr int kQuadTapStride = 100;                     // words: 33 x {re, im, -re} (+1); 100 = 3*32 + 4

struct V29QuadChan
{
    uint32_t *pcm;                                      // [kV29QuadTile/2]
    float2 *rrc;                                        // [54] pair k < 27: {x[k], 0}; pair 27 + k: {0, x[k]}
    float2 *u;                                          // [132] eq_buf in ring order: [B | 0 | B | B]
    float *taps;                                        // [99] {re, im, -re} per tap
};

// tables -> LDS, by all threads of the workgroup (tid of n)
SPG_FN void v29_quad_tables(V29QuadTables &T, const V29Tables &TB, int tid, int n)
{
    for (int i = tid;  i < kRrcSets*kRrcLen;  i += n)
    {
        const int set = i/kRrcLen;
        const int tap = i - set*kRrcLen;
        T.rrc[tap*kRrcSets + set] = make_float2(TB.rrc_re[i], TB.rrc_im[i]);
    }
    for (int i = tid;  i < 2048;  i += n)
        T.sine[i] = TB.sine[i];
    for (int i = tid;  i < 400;  i += n)
        T.space_map[i] = TB.space_map[i];
    for (int i = tid;  i < 16;  i += n)
    {
        const float re[16] = {3, 1, 0, -1, -3, -1, 0, 1, 5, 3, 0, -3, -5, -3, 0, 3};
        const float im[16] = {0, 1, 3, 1, 0, -1, -3, -1, 0, 3, 5, 3, 0, -3, -5, -3};
        T.konst[2*i] = re[i];
        T.konst[2*i + 1] = im[i];
    }
}

template <class Q>
SPG_FN void v29_quad_run(Q &q, const V29Launch &L, const int ch, const V29QuadTables &T, const V29QuadChan C)
{
    const int role = q.role();
    const V29Tables &TB = *L.tab;
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

    const size_t N = (size_t) L.n_ch;
    const int mylen = L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;
    auto ldf = [&](int w) { return __uint_as_float(L.state[(size_t) w*N + ch]); };
    auto ldi = [&](int w) { return (int32_t) L.state[(size_t) (kV29Floats + w)*N + ch]; };
    // a NaN goes back as x86's (v29_dev.hpp)
    auto stf = [&](int w, float v) { L.state[(size_t) w*N + ch] = (v != v)  ?  0xFFC00000u  :  __float_as_uint(v); };
    auto sti = [&](int w, int32_t v) { L.state[(size_t) (kV29Floats + w)*N + ch] = (uint32_t) v; };

    // ---- state: scalars replicated in the four lanes, arrays into LDS (dealt over the lanes) ---------------------------
    float agc_scaling = ldf(VF_AGC);
    float agc_scaling_save = ldf(VF_AGC_SAVE);
    const float eq_delta = ldf(VF_EQ_DELTA);
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
        C.u[3*kEqLen + i] = x;
    }
    const int bit_rate = ldi(VI_BIT_RATE);
    int rrc_step = ldi(VI_RRC_STEP);
    uint32_t scramble_reg = (uint32_t) ldi(VI_SCRAMBLE);
    int training_scramble_reg = ldi(VI_TRAIN_SCRAMBLE);
    const int training_cd = ldi(VI_TRAINING_CD);
    int old_train = ldi(VI_OLD_TRAIN);
    int stage = ldi(VI_STAGE);
    int training_count = ldi(VI_TRAIN_COUNT);
    int last_sample = ldi(VI_LAST_SAMPLE);
    int signal_present = ldi(VI_SIGNAL_PRESENT);
    uint32_t carrier_phase = (uint32_t) ldi(VI_CARRIER_PHASE);
    int32_t carrier_phase_rate = ldi(VI_PHASE_RATE);
    int32_t carrier_phase_rate_save = ldi(VI_PHASE_RATE_SAVE);
    int32_t power_reading = ldi(VI_POWER);
    const int32_t carrier_on_power = ldi(VI_ON_POWER);
    const int32_t carrier_off_power = ldi(VI_OFF_POWER);
    int eq_step = ldi(VI_EQ_STEP);
    int eq_put_step = ldi(VI_EQ_PUT_STEP);
    int eq_skip = ldi(VI_EQ_SKIP);
    int baud_half = ldi(VI_BAUD_HALF);
    int32_t last_angle0 = ldi(VI_LAST_ANGLES);
    int32_t last_angle1 = ldi(VI_LAST_ANGLES + 1);
    int constellation_state = ldi(VI_CONSTEL);
    int total_corr = ldi(VI_TOTAL_CORR);
    int high_sample = ldi(VI_HIGH_SAMPLE);
    int low_samples = ldi(VI_LOW_SAMPLES);
    int drop_pending = ldi(VI_DROP_PENDING);
    // diff_angles[16] is only touched during WAIT_FOR_CDCD: it stays in the state array (every lane of the quad stores
    // the same value and reads it back)
    auto diff_ld = [&](int k) { return ldi(VI_DIFF_ANGLES + (k & 0xF)); };
    auto diff_st = [&](int k, int32_t v) { sti(VI_DIFF_ANGLES + (k & 0xF), v); };

    int8_t *evp = L.events + (size_t) ch*L.ev_cap;
    int n_ev = 0;
    auto emit = [&](int v)
    {
        if (role == 0  &&  n_ev < L.ev_cap)
            evp[n_ev] = (int8_t) v;
        n_ev++;
    };

    // The equaliser's taps as this lane multiplies with them ({re, im} on even lanes, {im, -re} on odd ones), in registers: in
    // data mode they change with every tenth baud (eq_skip), the sum that reads them runs every baud, and what a lone wave pays
    // for in that sum is the LDS words it reads (tools/probe_lds.hip: 35 cycles a term with both operands from LDS, 23 with the
    // taps in registers and the products as one packed multiply).  LDS keeps the copy the LMS update's lanes exchange through;
    // whoever changes it there (the update, a restart) has the registers loaded again.
    f32x2v tc[kEqLen];
    auto load_taps = [&]()
    {
        SPG_UNROLL
        for (int i = 0;  i < kEqLen;  i++)
            tc[i] = (f32x2v) {C.taps[3*i + (role & 1)], C.taps[3*i + (role & 1) + 1]};
    };

    // v29_rx_restart(s, bit_rate, false), v29rx.c:1019-1098: all four lanes, the same stores
    auto restart = [&]()
    {
        for (int i = 0;  i < 2*kRrcLen;  i++)
            C.rrc[i] = make_float2(0.0f, 0.0f);
        rrc_step = 0;
        scramble_reg = 0;
        training_scramble_reg = 0x2A;
        stage = V29_SYMBOL_ACQUISITION;
        training_count = 0;
        signal_present = 0;
        high_sample = 0;
        low_samples = 0;
        drop_pending = 0;
        old_train = 0;
        for (int k = 0;  k < 16;  k++)
            diff_st(k, 0);
        carrier_phase = 0;
        power_reading = 0;
        constellation_state = 0;
        carrier_phase_rate = v29_f2i(1700.0f*65536.0f*65536.0f/8000);
        for (int i = 0;  i < kEqLen;  i++)
        {
            const float cr = (i == 16)  ?  3.0f  :  0.0f;                   // V29_EQUALIZER_PRE_LEN
            C.taps[3*i] = cr;
            C.taps[3*i + 1] = 0.0f;
            C.taps[3*i + 2] = -cr;
        }
        load_taps();
        for (int i = 0;  i < 4*kEqLen;  i++)
            C.u[i] = make_float2(0.0f, 0.0f);
        eq_put_step = kRrcSets*10/(3*2) - 1;
        eq_step = 0;
        agc_scaling_save = 0.0f;
        agc_scaling = (1.25f/1.0f)/735.0f;
        carrier_track_i = 8000.0f;
        carrier_track_p = 8000000.0f;
        last_sample = 0;
        eq_skip = 0;
        glow0 = glow1 = ghigh0 = ghigh1 = gdc0 = gdc1 = 0.0f;
        baud_phase = 0.0f;
        total_corr = 0;
        baud_half = 0;
    };

    // track_carrier() and tune_equalizer() (v29rx.c:281-331): requested by the stage logic, carried out once after it
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
    auto put_bit = [&](int bit)
    {
        // v29rx.c:365-397
        bit &= 1;
        const int out_bit = (bit ^ (int) (scramble_reg >> 17) ^ (int) (scramble_reg >> 22)) & 1;
        scramble_reg = (scramble_reg << 1) | (uint32_t) bit;
        if (stage == V29_NORMAL)
            emit(out_bit);
    };
    auto scrambled_training_bit = [&]()
    {
        // v29rx.c:350-362
        const int bit = training_scramble_reg & 1;
        training_scramble_reg >>= 1;
        if (bit ^ (training_scramble_reg & 1))
            training_scramble_reg |= 0x40;
        return bit;
    };
    auto decode_baud = [&](float zre, float zim)
    {
        // v29rx.c:400-481
        int nearest;
        if (bit_rate == 4800)
        {
            const int b1 = (zim > zre);
            const int b2 = (zim < -zre);
            nearest = ((b2 << 1) | (b1 ^ b2)) << 1;
            const int idx = ((nearest - constellation_state) >> 1) & 3;
            const int raw_bits = (0x1320 >> (4*idx)) & 0xF;            // phase_steps_4800 = {0, 2, 3, 1}
            put_bit(raw_bits);
            put_bit(raw_bits >> 1);
        }
        else
        {
            int re = v29_f2i((zre + 5.0f)*2.0f);
            int im = v29_f2i((zim + 5.0f)*2.0f);
            re = max(0, min(19, re));
            im = max(0, min(19, im));
            nearest = T.space_map[re*20 + im];
            if (bit_rate == 9600)
                put_bit(nearest >> 3);
            else
                nearest &= 7;
            const int idx = (nearest - constellation_state) & 7;
            int raw_bits = (int) ((0x51376204u >> (4*idx)) & 0xF);  // phase_steps_9600 = {4,0,2,6,7,3,1,5}
            put_bit(raw_bits);
            put_bit(raw_bits >> 1);
            put_bit(raw_bits >> 2);
        }
        const float tre = T.konst[2*nearest];
        const float tim = T.konst[2*nearest + 1];
        track_carrier(tre, tim);
        if (--eq_skip <= 0)
        {
            eq_skip = 10;
            tune_equalizer(tre, tim);
        }
        constellation_state = nearest;
    };
    auto park = [&]()
    {
        agc_scaling_save = 0.0f;
        stage = V29_PARKED;
        emit(-5);                                           // SIG_STATUS_TRAINING_FAILED
    };

    SPG_PROF_DECL();
    const int16_t *src = L.amp + (size_t) ch*L.stride;
    SPG_LOADS_DONE();
    q.sync(1);
    load_taps();
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
#define QF_SETS                 kRrcSets
#define QF_PUT_ADD              (kRrcSets*10/(3*2))
#define QF_PARKED               V29_PARKED
#define QF_AGC_NUM              (1.25f/1.0f)
#define QF_TILE                 kV29QuadTile
#define QF_EQLEN                kEqLen
#define QF_RRC_COEF(tap)        T.rrc[(tap)*kRrcSets + my_step]
#define QF_EQ_THIRD_COPY(k, h)  C.u[3*kEqLen + (k)] = (h)
#include "quad_round_front.inc"
#undef QF_SETS
#undef QF_PUT_ADD
#undef QF_PARKED
#undef QF_AGC_NUM
#undef QF_TILE
#undef QF_EQLEN
#undef QF_RRC_COEF
#undef QF_EQ_THIRD_COPY

        // ---- the baud (process_half_baud() of the second T/2 instant, v29rx.c:484-786) ---------------------------------
        do_track = false;
        do_tune = false;
        do_save = false;
        float zre = 0.0f;                                   // the equaliser's output of this baud
        float zim = 0.0f;
        if (baud_done)
        {
            // the reference advances the carrier phase after the baud's processing, with the rate that may just have
            // changed: take the last sample's advance back, redo it at the end
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
            // equalizer_get(): cvec_circular_dot_prodf (complex_vector_float.c:137-196), one chain per lane:
            // role 0 / 1 = real / imaginary of the part from the ring position to the end, 2 / 3 = of the wrapped part
            {
                const float2 *x = &C.u[eq_step + ((role & 2)  ?  kEqLen  :  0)];
                const float *c = &C.taps[role & 1];
                float acc = 0.0f;
                SPG_UNROLL
                for (int i0 = 0;  i0 < kEqLen;  i0 += 11)
                {
                    float2 xs[11];
                    SPG_UNROLL
                    for (int i = 0;  i < 11;  i++)
                        xs[i] = x[i0 + i];
                    SPG_UNROLL
                    for (int i = 0;  i < 11;  i++)
                    {
                        const f32x2v p = (f32x2v) {xs[i].x, xs[i].y}*tc[i0 + i];
                        acc += p.x - p.y;
                    }
                    // SPG_SCHED_FENCE();
                }
                float z = acc + q.swap2(acc, 1);
                if (q.any(!(fabsf(z) < __builtin_inff()), 5))
                {
                    // not finite: a tap may be, and then a padding zero times it was NaN where the reference has no
                    // term at all.  Again, with every term of the other part selected away instead of multiplied by zero
                    const int split = kEqLen - eq_step;
                    const float2 *xx = &C.u[2*kEqLen + eq_step];
                    float acc2 = 0.0f;
                    for (int i = 0;  i < kEqLen;  i++)
                    {
                        const float2 xv = xx[i];
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
            if (!q.any(stage != V29_NORMAL, 9))
            {
                // -- every channel of the wave carries data: decode_baud() (v29rx.c:400-481) and put_bit() (:365-397)
                // in one straight piece, the baud's bits descrambled together and stored with one access
                int nearest;
                int raw;                                    // the baud's bits in put_bit() order, first = bit 0
                int nbits;
                if (bit_rate == 4800)
                {
                    const int b1 = (zim > zre);
                    const int b2 = (zim < -zre);
                    nearest = ((b2 << 1) | (b1 ^ b2)) << 1;
                    const int idx = ((nearest - constellation_state) >> 1) & 3;
                    raw = (0x1320 >> (4*idx)) & 0x3;       // phase_steps_4800 = {0, 2, 3, 1}
                    nbits = 2;
                }
                else
                {
                    int re = v29_f2i((zre + 5.0f)*2.0f);
                    int im = v29_f2i((zim + 5.0f)*2.0f);
                    re = max(0, min(19, re));
                    im = max(0, min(19, im));
                    nearest = T.space_map[re*20 + im];
                    const bool full = (bit_rate == 9600);
                    const int first = (nearest >> 3) & 1;
                    nearest = full  ?  nearest  :  (nearest & 7);
                    const int idx = (nearest - constellation_state) & 7;
                    const int three = (int) ((0x51376204u >> (4*idx)) & 0x7);     // phase_steps_9600 = {4,0,2,6,7,3,1,5}
                    raw = full  ?  ((three << 1) | first)  :  three;
                    nbits = full  ?  4  :  3;
                }
                // the self synchronising descrambler: the taps (17, 22) lie beyond the four bits of a baud, so each output
                // bit only needs the register as it was before the baud -- bit j of the baud (put_bit() order) goes with the
                // register's bits 17 - j and 22 - j: the baud's bits as a nibble against the bit-reversed nibbles at 14 and 19
                const uint32_t rev4 = 0xF7B3D591u;          // bit reversal of a nibble: nibble n - 8 of this word for n = 8 .. 15,
                const uint32_t rev4lo = 0xE6A2C480u;        // nibble n of this one for n = 0 .. 7
                const uint32_t taps = ((scramble_reg >> 14) ^ (scramble_reg >> 19)) & 0xFu;
                const uint32_t taps_r = (((taps & 8u)  ?  rev4  :  rev4lo) >> (4*(taps & 7u))) & 0xFu;
                // one byte per bit, the first lowest: bit k of the nibble to bit 8k
                const uint32_t out = ((((uint32_t) raw & 0xFu) ^ taps_r)*0x00204081u) & 0x01010101u;
                {
                    // put_bit order: bit 0 first, so it ends up highest in the register
                    const uint32_t rw = (uint32_t) raw & 0xFu;
                    uint32_t rev = (((rw & 8u)  ?  rev4  :  rev4lo) >> (4*(rw & 7u))) & 0xFu;
                    rev >>= 4 - nbits;
                    scramble_reg = (scramble_reg << nbits) | rev;
                }
                if (role == 0)
                {
                    if (n_ev + 4 <= L.ev_cap)
                    {
                        __builtin_memcpy(evp + n_ev, &out, 4);      // bytes past the baud's bits are overwritten by the next
                    }
                    else
                    {
                        for (int j = 0;  j < nbits;  j++)
                        {
                            if (n_ev + j < L.ev_cap)
                                evp[n_ev + j] = (int8_t) ((out >> (8*j)) & 1u);
                        }
                    }
                }
                n_ev += nbits;
                const float tre = T.konst[2*nearest];
                const float tim = T.konst[2*nearest + 1];
                do_track = true;
                tgt_re = tre;
                tgt_im = tim;
                use_track_i = carrier_track_i;
                use_track_p = carrier_track_p;
                const bool tune_now = (eq_skip <= 1);
                eq_skip = tune_now  ?  10  :  (eq_skip - 1);
                do_tune = tune_now;
                constellation_state = nearest;
            }
            else
            {
            if (stage == V29_NORMAL  ||  stage == V29_TEST_ONES)
                decode_baud(zre, zim);
            switch (stage)
            {
            case V29_NORMAL:
                break;
            case V29_SYMBOL_ACQUISITION:
                if (++training_count >= 60)
                {
                    stage = V29_LOG_PHASE;
                    for (int k = 0;  k < 16;  k++)
                        diff_st(k, 0);
                    last_angle0 = v29_arctan2(zim, zre);
                    if (agc_scaling_save == 0.0f)
                        agc_scaling_save = agc_scaling;
                }
                break;
            case V29_LOG_PHASE:
                last_angle1 = v29_arctan2(zim, zre);
                training_count = 1;
                stage = V29_WAIT_FOR_CDCD;
                break;
            case V29_WAIT_FOR_CDCD:
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
                if ((ang > 0x20000000  ||  ang < (int32_t) 0xE0000000u)  &&  training_count >= 13)
                {
                    i = (training_count - 8) & ~1;
                    if (i > 1)
                    {
                        const int jj = i & 0xF;
                        ang = (int32_t) ((uint32_t) diff_ld(jj) + (uint32_t) diff_ld(jj | 1))/(i - 1);
                        carrier_phase_rate += 3*16*(ang/20);
                    }
                    if (carrier_phase_rate < v29_f2i((1700.0f - 20.0f)*65536.0f*65536.0f/8000)
                        ||  carrier_phase_rate > v29_f2i((1700.0f + 20.0f)*65536.0f*65536.0f/8000))
                    {
                        park();
                        break;
                    }
                    // v29rx.c:618-624: spin the equaliser delay line (each lane every fourth entry) and the carrier
                    const float p = ((uint32_t) angle)*2.0f*3.1415926f/(65536.0f*65536.0f);
                    const float zc = spg_sincosf(p, true);
                    const float zs = -spg_sincosf(p, false);
                    for (int k = role;  k < kEqLen;  k += 4)
                    {
                        const float2 xv = C.u[k];
                        const float2 r = make_float2(xv.x*zc - xv.y*zs, xv.x*zs + xv.y*zc);
                        C.u[k] = r;
                        C.u[2*kEqLen + k] = r;
                        C.u[3*kEqLen + k] = r;
                    }
                    carrier_phase += (uint32_t) angle;
                    const int bit = scrambled_training_bit();
                    constellation_state = (0x002030B0 >> (4*(training_cd + bit))) & 0xF;   // cdcd_pos = {0,11,0,3,0,2}
                    training_count = 1;
                    stage = V29_TRAIN_ON_CDCD;
                    emit(-3);                           // SIG_STATUS_TRAINING_IN_PROGRESS
                    break;
                }
                if (++training_count > 128)
                    park();
                break;
            }
            case V29_TRAIN_ON_CDCD:
            {
                const int bit = scrambled_training_bit();
                constellation_state = (0x002030B0 >> (4*(training_cd + bit))) & 0xF;
                const float tre = T.konst[2*constellation_state];
                const float tim = T.konst[2*constellation_state + 1];
                track_carrier(tre, tim);
                tune_equalizer(tre, tim);
                if (++training_count >= 384 - 48)
                {
                    stage = V29_TRAIN_ON_CDCD_AND_TEST;
                    training_error = 0.0f;
                    carrier_track_i = 200.0f;
                    carrier_track_p = 1000000.0f;
                }
                break;
            }
            case V29_TRAIN_ON_CDCD_AND_TEST:
            {
                const int bit = scrambled_training_bit();
                constellation_state = (0x002030B0 >> (4*(training_cd + bit))) & 0xF;
                const float tre = T.konst[2*constellation_state];
                const float tim = T.konst[2*constellation_state + 1];
                track_carrier(tre, tim);
                tune_equalizer(tre, tim);
                const float dre2 = zre - tre;
                const float dim2 = zim - tim;
                training_error += dre2*dre2 + dim2*dim2;
                if (++training_count >= 384)
                {
                    if (training_error < 48.0f*2.0f)
                    {
                        training_error = 0.0f;
                        training_count = 0;
                        constellation_state = 0;
                        stage = V29_TEST_ONES;
                    }
                    else
                    {
                        park();
                    }
                }
                break;
            }
            case V29_TEST_ONES:
            {
                const float tre = T.konst[2*constellation_state];
                const float tim = T.konst[2*constellation_state + 1];
                const float dre2 = zre - tre;
                const float dim2 = zim - tim;
                training_error += dre2*dre2 + dim2*dim2;
                if (++training_count >= 48)
                {
                    if (training_error < 48.0f*1.0f)
                    {
                        emit(-4);                       // SIG_STATUS_TRAINING_SUCCEEDED
                        signal_present = 60;
                        stage = V29_NORMAL;
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
            }
            if (do_track)
            {
                const float error = zim*tgt_re - zre*tgt_im;
                carrier_phase_rate += v29_f2i(use_track_i*error);
                carrier_phase += (uint32_t) v29_f2i(use_track_p*error);
            }
            carrier_phase += (uint32_t) carrier_phase_rate;     // dds_advancef() with the rate the baud left behind
        }
        SPG_PROF_STAMP(7);
        q.sync(6);
        // ---- tune_equalizer(): cvec_circular_lmsf (complex_vector_float.c:201-219), tap i with the entry i places on from the ring
        // position, for the channels whose stage logic asked for it this baud -----------------------------------------------------
        if (q.any(do_tune, 12))
        {
            const float lms_ere = (tgt_re - zre)*eq_delta;      // (meaningful on the lanes that update)
            const float lms_eim = (tgt_im - zim)*eq_delta;
            bool lms_done = false;
#if !defined(SPG_HOST_EMUL)
            {
                // In data mode a channel updates on every tenth baud.  When its neighbours in the wavefront are at other bauds of
                // their ten -- calls that did not start together -- the update below would run for the whole wave with four lanes
                // in work, on nearly every round.  A few channels at a time are therefore updated by the WAVE: one lane per tap,
                // the channel's delay line and taps addressed in LDS from the lane's own (a constant stride per channel), its error
                // terms and ring position read from its lanes' registers.  The same expressions, one tap per lane instead of nine in
                // turn.  (16 384 x 160 with the channels' starts spread over a frame: 167.6 -> 155.7 us a launch.)
                const unsigned long long tuning = __ballot(do_tune) & 0x1111111111111111ull;      // lane 0 of every quad that updates
                const int n_tuning = __popcll(tuning);
                if (n_tuning <= 4  &&  __popcll(__ballot(1)) == 64)
                {
                    const int wl = (int) (threadIdx.x & 63);
                    unsigned long long left = tuning;
                    do
                    {
                        const int src = __ffsll((long long) left) - 1;
                        left &= left - 1;
                        const float ere = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lms_ere), src));
                        const float eim = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lms_eim), src));
                        const int at = __builtin_amdgcn_readlane(eq_step, src);
                        const int dq = (src >> 2) - (wl >> 2);                  // that channel's arrays from this lane's
                        if (wl < kEqLen)
                        {
                            const float2 xv = (C.u + dq*kQuadEqStride)[2*kEqLen + at + wl];
                            float *tp = C.taps + dq*kQuadTapStride + 3*wl;
                            const f32x2v c0 = {tp[0], tp[1]};
                            const f32x2v u = (f32x2v) {xv.y, xv.x}*(f32x2v) {eim, eim};
                            const f32x2v w = (f32x2v) {xv.x, xv.y}*(f32x2v) {ere, ere};
                            const f32x2v c = c0*(f32x2v) {0.9999f, 0.9999f} + (u + (f32x2v) {w.x, -w.y});
                            tp[0] = c.x;
                            tp[1] = c.y;
                            tp[2] = -c.x;
                        }
                    }
                    while (left != 0);
                    lms_done = true;
                }
            }
#endif
            if (do_tune  &&  !lms_done)
            {
                // lane r takes taps r, r + 4, ...
                const float2 *x = &C.u[2*kEqLen + eq_step];
                SPG_UNROLL
                for (int j = 0;  j < (kEqLen + 3)/4;  j++)
                {
                    const int i = role + 4*j;
                    if (i < kEqLen)
                    {
                        const float2 xv = x[i];
                        const f32x2v c0 = {C.taps[3*i], C.taps[3*i + 1]};
                        // {xi*eim + xr*ere, xr*eim - xi*ere}
                        const f32x2v u = (f32x2v) {xv.y, xv.x}*(f32x2v) {lms_eim, lms_eim};
                        const f32x2v w = (f32x2v) {xv.x, xv.y}*(f32x2v) {lms_ere, lms_ere};
                        const f32x2v c = c0*(f32x2v) {0.9999f, 0.9999f} + (u + (f32x2v) {w.x, -w.y});
                        C.taps[3*i] = c.x;
                        C.taps[3*i + 1] = c.y;
                        C.taps[3*i + 2] = -c.x;
                    }
                }
            }
            SPG_PROF_STAMP(8);
            q.sync(7);
            if (do_tune)
                load_taps();
        }
        if (q.any(do_save, 13))
        {
            if (do_save)
            {
                carrier_phase_rate_save = carrier_phase_rate;
                for (int k = role;  k < kEqLen;  k += 4)
                {
                    stf(VF_EQ_SAVE + 2*k, C.taps[3*k]);
                    stf(VF_EQ_SAVE + 2*k + 1, C.taps[3*k + 1]);
                }
            }
        }
    } while (q.any(pos < tn, 11));
    }

    SPG_PROF_STAMP(9);
    SPG_PROF_FLUSH();
    // ---- write back (arrays dealt over the lanes, scalars by the first) ------------------------------------------------
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
    if (role == 0)
    {
        stf(VF_AGC, agc_scaling);
        stf(VF_AGC_SAVE, agc_scaling_save);
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
        sti(VI_RRC_STEP, rrc_step);
        sti(VI_SCRAMBLE, (int32_t) scramble_reg);
        sti(VI_TRAIN_SCRAMBLE, training_scramble_reg);
        sti(VI_OLD_TRAIN, old_train);
        sti(VI_STAGE, stage);
        sti(VI_TRAIN_COUNT, training_count);
        sti(VI_LAST_SAMPLE, last_sample);
        sti(VI_SIGNAL_PRESENT, signal_present);
        sti(VI_CARRIER_PHASE, (int32_t) carrier_phase);
        sti(VI_PHASE_RATE, carrier_phase_rate);
        sti(VI_PHASE_RATE_SAVE, carrier_phase_rate_save);
        sti(VI_POWER, power_reading);
        sti(VI_EQ_STEP, eq_step);
        sti(VI_EQ_PUT_STEP, eq_put_step);
        sti(VI_EQ_SKIP, eq_skip);
        sti(VI_BAUD_HALF, baud_half);
        sti(VI_LAST_ANGLES, last_angle0);
        sti(VI_LAST_ANGLES + 1, last_angle1);
        sti(VI_CONSTEL, constellation_state);
        sti(VI_TOTAL_CORR, total_corr);
        sti(VI_HIGH_SAMPLE, high_sample);
        sti(VI_LOW_SAMPLES, low_samples);
        sti(VI_DROP_PENDING, drop_pending);
        L.ev_count[ch] = n_ev;
    }
}

#if !defined(SPG_HOST_EMUL)

// CPW channels per wave (4*CPW live lanes), WPB waves per workgroup sharing the tables.
template <int CPW, int WPB>
__global__ __launch_bounds__(64*WPB)
void v29_quad_kernel(const V29Launch L)
{
    __shared__ V29QuadTables T;
    __shared__ uint32_t s_pcm[WPB*CPW*kQuadPcmStride];
    __shared__ float2 s_rrc[WPB*CPW*kQuadRrcStride];
    __shared__ float2 s_u[WPB*CPW*kQuadEqStride];
    __shared__ float s_taps[WPB*CPW*kQuadTapStride];
    v29_quad_tables(T, *L.tab, (int) threadIdx.x, 64*WPB);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = (int) (threadIdx.x >> 6);
    const int cw = lane >> 2;
    const int ch = (blockIdx.x*WPB + wv)*CPW + cw;
    if (cw >= CPW  ||  ch >= L.n_ch)
        return;
    QuadDev q{lane & 3};
    const int slot = wv*CPW + cw;
    const V29QuadChan C = {s_pcm + slot*kQuadPcmStride, s_rrc + slot*kQuadRrcStride, s_u + slot*kQuadEqStride, s_taps + slot*kQuadTapStride};
    v29_quad_run(q, L, ch, T, C);
}

#endif

}   // namespace spg
