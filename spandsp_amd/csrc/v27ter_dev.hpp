// v27ter_dev.hpp -- device side of the batched V.27ter receiver (reference: src/v27ter_rx.c:197-1028;
// primitives as in v29_dev.hpp).  Same mapping and baud-aligned execution as the V.29 bank: one channel per lane,
// RRC delay line, PCM tile and equaliser taps index-major in LDS, the equaliser delay line in VGPRs in age order, the
// reference's summation order kept.
// Differences from V.29 that shape the kernel: the pulse-shaping filter only runs at the T/2
// instants (no per-sample timing-error filter; symbol timing is a Gardner detector on the
// equaliser delay line), 32 taps, an 8-point PSK slicer, and a descrambler with the V.27ter
// repeated-pattern guard.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "v29_dev.hpp"

#include "v27ter_common.hpp"

namespace spg {

// WPB, TILE, PK16: several waves per workgroup sharing the tables, a short PCM tile and the RRC delay line as packed
// int16 pairs, for banks of full waves -- see v29_bank_kernel.
template <int CPW, bool QAM = false, int WPB = 1, int TILE = kPcmTile, bool PK16 = false>
__global__ __launch_bounds__(64*WPB)
void v27ter_bank_kernel(const V27Launch L)
{
    static_assert(WPB == 1  ||  CPW == 64, "several waves per workgroup: full waves only");
    static_assert(TILE%8 == 0  &&  TILE >= 8, "the PCM tile is staged in 16-byte pieces");
    __shared__ float t_rrc_re[kV27MaxSets*kRrcLen];     // [tap][set]
    __shared__ float t_rrc_im[kV27MaxSets*kRrcLen];
    __shared__ float t_sine[2048];
    __shared__ uint16_t t_sqrt[194];
    // per-lane RRC delay line (doubled) and PCM tile, index-major [word][CPW]; equaliser taps {re, im} [tap][lane]
    // RRC delay line as zero padded pairs, see v29_dev.hpp
    __shared__ float2 lanes[PK16  ?  1  :  WPB*CPW*2*kRrcLen];
    __shared__ uint32_t lanes16[PK16  ?  WPB*CPW*2*kRrcLen  :  1];
    __shared__ uint32_t pcm[WPB*CPW*(TILE/2)];
    __shared__ float2 taps[WPB*kV27EqLen*CPW];

    const int lane = threadIdx.x & 63;
    const int wv = (WPB == 1)  ?  0  :  (int) (threadIdx.x >> 6);
    const int ch = (blockIdx.x*WPB + wv)*CPW + lane;
    constexpr int kThreads = 64*WPB;
    const int tid = threadIdx.x;
    const V27Tables &TB = *L.tab;
    const bool fast = (L.bit_rate == 4800);
    const int sets = fast  ?  8  :  12;
    const int put_add = fast  ?  8*5/2  :  12*20/(3*2);

    {
        const float *sre = fast  ?  TB.re4800  :  TB.re2400;
        const float *sim = fast  ?  TB.im4800  :  TB.im2400;
        for (int i = tid;  i < sets*kRrcLen;  i += kThreads)
        {
            const int set = i/kRrcLen;
            const int tap = i - set*kRrcLen;
            t_rrc_re[tap*kV27MaxSets + set] = sre[i];
            t_rrc_im[tap*kV27MaxSets + set] = sim[i];
        }
    }
    for (int i = tid;  i < 2048;  i += kThreads)
        t_sine[i] = TB.sine[i];
    for (int i = tid;  i < 194;  i += kThreads)
        t_sqrt[i] = TB.sqrt_tab[i];
    __syncthreads();
    if (lane >= CPW  ||  ch >= L.n_ch)
        return;

    const size_t N = (size_t) L.n_ch;
    const int mylen = L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;
    auto ldf = [&](int w) { return __uint_as_float(L.state[(size_t) w*N + ch]); };
    auto ldi = [&](int w) { return (int32_t) L.state[(size_t) (kV27Floats + w)*N + ch]; };
    // A float word goes back as its bits -- except a NaN (a receiver whose equaliser has run away is full of them), which
    // goes back as x86's: there an invalid operation makes the negative quiet NaN and arithmetic hands an operand's NaN on
    // sign and all, while here the negated operand of a subtraction flips it.  Nothing ever depends on a NaN's sign.
    auto stf = [&](int w, float v) { L.state[(size_t) w*N + ch] = (v != v)  ?  0xFFC00000u  :  __float_as_uint(v); };
    auto sti = [&](int w, int32_t v) { L.state[(size_t) (kV27Floats + w)*N + ch] = (uint32_t) v; };

    float2 *rrc2 = &lanes[PK16  ?  0  :  (wv*CPW*2*kRrcLen + lane)];           // [2*27] pairs, stride CPW
    uint32_t *rrc16 = &lanes16[PK16  ?  (wv*CPW*2*kRrcLen + lane)  :  0];
    uint32_t *pcmw = &pcm[wv*CPW*(TILE/2)];
    auto rrc_put = [&](int k, float v)
    {
        if (PK16)
        {
            const uint32_t h = (uint32_t) (int) v & 0xFFFFu;
            rrc16[k*CPW] = h;
            rrc16[(kRrcLen + k)*CPW] = h << 16;
        }
        else
        {
            rrc2[k*CPW].x = v;
            rrc2[(kRrcLen + k)*CPW].y = v;
        }
    };
    auto rrc_at = [&](int k) -> float
    {
        if (PK16)
            return (float) (int) (short) (rrc16[k*CPW] & 0xFFFFu);
        return rrc2[k*CPW].x;
    };
    float2 *ctap = &taps[wv*kV27EqLen*CPW + lane];
#define TAP(i)      ctap[(i)*CPW]
    constexpr int EQN = kV27EqLen;

    float agc_scaling = ldf(WF_AGC);
    float agc_scaling_save = ldf(WF_AGC_SAVE);
    const float eq_delta = ldf(WF_EQ_DELTA);
    float training_error = ldf(WF_TRAIN_ERR);
    float carrier_track_p = ldf(WF_TRACK_P);
    float carrier_track_i = ldf(WF_TRACK_I);
    for (int i = 0;  i < kRrcLen;  i++)
    {
        const float v = ldf(WF_RRC + i);
        if (!PK16)
        {
            rrc2[i*CPW] = make_float2(v, 0.0f);
            rrc2[(kRrcLen + i)*CPW] = make_float2(0.0f, v);
        }
        rrc_put(i, v);
    }
    for (int i = 0;  i < EQN;  i++)
        TAP(i) = make_float2(ldf(WF_EQ_COEFF + 2*i), ldf(WF_EQ_COEFF + 2*i + 1));
    // equaliser delay line in age order: xre[i] = eq_buf[(eq_step + i) mod 32] (i = 0 oldest)
    float xre[EQN];
    float xim[EQN];
    bool eq_clear_pending = false;
    bool restart_pending = false;
    {
        const int es = ldi(WI_EQ_STEP);
#pragma unroll
        for (int i = 0;  i < EQN;  i++)
        {
            const int k = (es + i) & (EQN - 1);
            xre[i] = ldf(WF_EQ_BUF + 2*k);
            xim[i] = ldf(WF_EQ_BUF + 2*k + 1);
        }
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
        if (n_ev < L.ev_cap)
            evp[n_ev] = (int8_t) v;
        n_ev++;
    };

    // qam_report(user, constel, target, symbol) calls, for the kernel variant a caller's tap asks for: one record per
    // call = {events emitted before it in this launch, 1 if the pointers were NULL, symbol, constel re / im, target re / im}
    int n_q = 0;
    auto qam_report = [&](uint32_t null_ptrs, int symbol, float cre, float cim, float tre, float tim)
    {
        if constexpr (QAM)
        {
            if (n_q < L.qam_cap)
            {
                uint32_t *r = L.qam + ((size_t) ch*L.qam_cap + n_q)*7;
                r[0] = (uint32_t) n_ev;
                r[1] = null_ptrs;
                r[2] = (uint32_t) symbol;
                r[3] = __float_as_uint(cre);
                r[4] = __float_as_uint(cim);
                r[5] = __float_as_uint(tre);
                r[6] = __float_as_uint(tim);
            }
            n_q++;
        }
    };

    // v27ter_rx_restart() as the receive path reaches it (v27ter_rx.c:1091-1160; s->old_train is never set)
    auto restart = [&]()
    {
        for (int i = 0;  i < 2*kRrcLen;  i++)
        {
            if (PK16)
                rrc16[i*CPW] = 0;
            else
                rrc2[i*CPW] = make_float2(0.0f, 0.0f);
        }
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
            TAP(i) = make_float2((i == 17)  ?  1.414f  :  0.0f, 0.0f);      // V27TER_EQUALIZER_PRE_LEN + 1
        // (the equaliser delay line is register state: it is cleared where it is next looked at, see v29_dev.hpp)
        eq_clear_pending = true;
        eq_put_step = put_add;
        eq_step = 0;
        eq_skip = 0;
        last_sample = 0;
        gardner_integrate = 0;
        total_corr = 0;
        gardner_step = 512;
        baud_half = 0;
    };

    // vec_circular_dot_prodf() with its two partial sums kept in the halves of a packed pair (see v29_dev.hpp)
    auto rrc_dot = [&](const float *table, int row)
    {
        const float *y = table + row;
        const float2 *x = rrc2 + rrc_step*CPW;
        const uint32_t *xq = rrc16 + rrc_step*CPW;
        f32x2v xs[kRrcLen];
        float ys[kRrcLen];
#pragma unroll
        for (int i = 0;  i < kRrcLen;  i++)
        {
            if (PK16)
            {
                const uint32_t q = xq[i*CPW];
                xs[i] = (f32x2v) {(float) (int) (short) (q & 0xFFFFu), (float) ((int) q >> 16)};
            }
            else
            {
                const float2 w = x[i*CPW];
                xs[i] = (f32x2v) {w.x, w.y};
            }
            ys[i] = y[i*kV27MaxSets];
        }
        f32x2v a = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0;  i < kRrcLen;  i++)
            a += xs[i]*(f32x2v) {ys[i], ys[i]};
        return a.x + a.y;
    };
    // track_carrier() and tune_equalizer() are requested by the stage logic and carried out once, after it, with the
    // loop gains as they were when the reference would have called them (see v29_dev.hpp).
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
        if (pattern_count >= 33)
        {
            out_bit ^= 1;
            pattern_count = 0;
        }
        else if (training)
        {
            pattern_count = 0;
        }
        else
        {
            const uint32_t m = ((scramble_reg >> 7) ^ (uint32_t) in_bit) & ((scramble_reg >> 8) ^ (uint32_t) in_bit)
                             & ((scramble_reg >> 11) ^ (uint32_t) in_bit) & 1u;
            pattern_count = m  ?  0  :  (pattern_count + 1);
        }
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
        const int q = k >> 1;                               // 0: +re, 1: +im, 2: -re, 3: -im (even k); diagonals for odd k
        if (k & 1)
        {
            tre = (q == 0  ||  q == 3)  ?  1.0f  :  -1.0f;
            tim = (q == 0  ||  q == 1)  ?  1.0f  :  -1.0f;
        }
        else
        {
            tre = (q == 0)  ?  mag  :  (q == 2)  ?  -mag  :  0.0f;
            tim = (q == 1)  ?  mag  :  (q == 3)  ?  -mag  :  0.0f;
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

    const int16_t *src = L.amp + (size_t) ch*L.stride;
    for (int tile = 0;  tile < L.samples;  tile += TILE)
    {
    const int tn = max(0, min(TILE, mylen - tile));         // per lane when the call carries per-channel lengths
    // ---- stage this lane's stretch of PCM: pcm[k][lane] = samples 2k, 2k+1 of the tile ----------------------
    {
        const int16_t *row = src + tile;
        const bool wide = ((((uintptr_t) row) & 15) == 0)  &&  (tn == TILE);
        if (wide)
        {
#pragma unroll
            for (int k = 0;  k < TILE/8;  k++)
            {
                const int4 v = ((const int4 *) row)[k];
                pcmw[(4*k + 0)*CPW + lane] = (uint32_t) v.x;
                pcmw[(4*k + 1)*CPW + lane] = (uint32_t) v.y;
                pcmw[(4*k + 2)*CPW + lane] = (uint32_t) v.z;
                pcmw[(4*k + 3)*CPW + lane] = (uint32_t) v.w;
            }
        }
        else
        {
            for (int k = 0;  k < (tn + 1)/2;  k++)
            {
                const uint32_t lo = (uint16_t) row[2*k];
                const uint32_t hi = (2*k + 1 < tn)  ?  (uint16_t) row[2*k + 1]  :  0u;
                pcmw[k*CPW + lane] = lo | (hi << 16);
            }
        }
    }
    int pos = 0;
    for (;;)
    {
    // One round = one baud of every lane (see v29_dev.hpp): two T/2 instants, then the baud phase with all lanes in step.
    bool any_ready = false;
    bool restarted = false;
    bool baud_done = false;
    float zre = 0.0f;
    float zim = 0.0f;
    for (int half = 0;  half < 2;  half++)
    {
    const bool take = (half == 1)  ||  (baud_half == 0);
    // ---- phase A: every lane runs its own samples up to its next T/2 instant (cheap here: no per-sample filter) ----
    bool ready = false;
    int power = 0;
    while (__any(take  &&  !ready  &&  !restart_pending  &&  pos < tn))
    {
    if (take  &&  !ready  &&  !restart_pending  &&  pos < tn)
    {
        const uint32_t pw = pcmw[(pos >> 1)*CPW + lane];
        const int amp = (int) (short) ((pos & 1)  ?  (pw >> 16)  :  (pw & 0xFFFF));
        pos++;
        do
        {
        rrc_put(rrc_step, (float) amp);
        if (++rrc_step >= kRrcLen)
            rrc_step = 0;

        // signal_detect(), v27ter_rx.c:779-861 (IAXMODEM_STUFF is #defined at v27ter_rx.c:1)
        {
            const int x = amp >> 1;
            int diff = (int) (short) (x - last_sample);
            last_sample = x;
            power_reading += ((diff*diff - power_reading) >> 4);
            power = power_reading;
            diff = (int) (short) abs(diff);
            if (10*diff < high_sample)
            {
                if (++low_samples > 120)
                {
                    power_reading = 0;
                    high_sample = 0;
                    low_samples = 0;
                }
            }
            else
            {
                low_samples = 0;
                if (diff > high_sample)
                    high_sample = diff;
            }
            if (signal_present > 0)
            {
                if (drop_pending  ||  power < carrier_off_power)
                {
                    if (--signal_present <= 0)
                    {
                        // v27ter_rx_restart(): carried out right after this loop (see v29_dev.hpp)
                        restart_pending = true;
                        emit(-1);                           // SIG_STATUS_CARRIER_DOWN
                        power = 0;
                        break;
                    }
                    else
                    {
                        drop_pending = 1;
                    }
                }
            }
            else
            {
                if (power < carrier_on_power)
                {
                    power = 0;
                }
                else
                {
                    signal_present = 1;
                    drop_pending = 0;
                    emit(-2);                               // SIG_STATUS_CARRIER_UP
                }
            }
        }
        if (power == 0  ||  stage == V27_PARKED)
            break;

        eq_put_step -= sets;
        if (eq_put_step <= 0)
            ready = true;
        else
            carrier_phase += (uint32_t) carrier_phase_rate;
        }
        while (0);
    }
    }
    // ---- phase B: the T/2 instant, for all lanes that reached one ----------------------------------------------
    if (__any(restart_pending))
    {
        if (restart_pending)
        {
            restart();
            restart_pending = false;
            restarted = true;
        }
    }
    // (and the restart leaves the clearing of the equaliser delay line to here)
    if (__any(eq_clear_pending))
    {
        if (eq_clear_pending)
        {
#pragma unroll
            for (int i = 0;  i < EQN;  i++)
            {
                xre[i] = 0.0f;
                xim[i] = 0.0f;
            }
            eq_clear_pending = false;
        }
    }
    if (ready)
    {
        any_ready = true;
            if (stage == V27_SYMBOL_ACQUISITION)
            {
                int root_power;
                {
                    uint32_t xx = (uint32_t) power;
                    const int top = 31 - __builtin_clz(xx);
                    const int shift = 30 - (top & ~1);
                    xx <<= shift;
                    root_power = t_sqrt[((xx >> 24) & 0xFF) - 64] >> (shift >> 1);
                }
                if (root_power == 0)
                    root_power = 1;
                agc_scaling = (1.414f/1.000000f)/(float) root_power;
            }
            const int step = min(-eq_put_step, sets - 1);
            float v = rrc_dot(t_rrc_re, step);
            const float sre = v*agc_scaling;
            v = rrc_dot(t_rrc_im, step);
            const float sim = v*agc_scaling;
            const float dre = t_sine[(uint32_t) (carrier_phase + (1u << 30)) >> 21];
            const float dim = t_sine[carrier_phase >> 21];
            const float hre = sre*dre - sim*dim;
            const float him = -sre*dim - sim*dre;
            eq_put_step += put_add;

            // ---- process_half_baud(), v27ter_rx.c:531-777 ----
#pragma unroll
            for (int i = 0;  i < EQN - 1;  i++)
            {
                xre[i] = xre[i + 1];
                xim[i] = xim[i + 1];
            }
            xre[EQN - 1] = hre;
            xim[EQN - 1] = him;
            if (++eq_step >= EQN)
                eq_step = 0;
            baud_half ^= 1;
            if (baud_half == 0)
                baud_done = true;
        carrier_phase += (uint32_t) carrier_phase_rate;
    }
    }
    if (!__any(any_ready  ||  restarted))
        break;
    // ---- phase C: the baud, for every lane that completed one in this round ----------------------------------
    if (baud_done)
    {
        carrier_phase -= (uint32_t) carrier_phase_rate;
                {
                    // symbol_sync(), v27ter_rx.c:486-528
                    // eq_buf[(eq_step - 1, -2, -3) & 31] = the three newest entries of the delay line
                    float p = xre[EQN - 3] - xre[EQN - 1];
                    p *= xre[EQN - 2];
                    float q = xim[EQN - 3] - xim[EQN - 1];
                    q *= xim[EQN - 2];
                    gardner_integrate += (p + q > 0.0f)  ?  gardner_step  :  -gardner_step;
                    if (abs(gardner_integrate) >= 128)
                    {
                        eq_put_step += gardner_integrate/128;
                        total_corr += gardner_integrate/128;
                        qam_report(1, gardner_integrate, 0.0f, 0.0f, 0.0f, 0.0f);      // v27ter_rx.c:517-518
                        gardner_integrate = 0;
                    }
                }
                {
                    const int split = EQN - eq_step;
                    float2 cs[EQN];
#pragma unroll
                    for (int i = 0;  i < EQN;  i++)
                        cs[i] = TAP(i);
                    f32x2v acc = f32x2v{0.0f, 0.0f};
                    f32x2v fst = f32x2v{0.0f, 0.0f};
#pragma unroll
                    for (int i = 0;  i < EQN;  i++)
                    {
                        if (i == split)
                        {
                            fst = acc;
                            acc = f32x2v{0.0f, 0.0f};
                        }
                        const f32x2v t1 = f32x2v{xre[i], xre[i]}*f32x2v{cs[i].x, cs[i].y};
                        const f32x2v t2 = f32x2v{xim[i], xim[i]}*f32x2v{cs[i].y, cs[i].x};
                        acc += t1 + f32x2v{-t2.x, t2.y};
                    }
                    zre = fst.x + acc.x;
                    zim = fst.y + acc.y;
                }

                do_track = false;
                do_tune = false;
                do_save = false;
                float rep_re = 0.0f;                        // `target` of process_half_baud(), for the qam report
                float rep_im = 0.0f;
                if (stage == V27_NORMAL  ||  stage == V27_TEST_ONES)
                    decode_baud(zre, zim);
                switch (stage)
                {
                case V27_NORMAL:
                    if constexpr (QAM)
                        target_of(fast  ?  constellation_state  :  (constellation_state << 1), rep_re, rep_im);
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
#pragma unroll
                        for (int k = 0;  k < EQN;  k++)
                        {
                            const float xr = xre[k];
                            const float xi = xim[k];
                            xre[k] = xr*zc - xi*zs;
                            xim[k] = xr*zs + xi*zc;
                        }
                        carrier_phase += (uint32_t) angle;
                        gardner_step = 2;
                        training_bc = 1;
                        training_bc ^= descramble(1);
                        descramble(1);
                        descramble(1);
                        constellation_state = training_bc  ?  4  :  0;
                        rep_re = training_bc  ?  -1.414f  :  1.414f;
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
                    rep_re = tre;
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
                    rep_re = tre;
                    rep_im = tim;
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
                qam_report(0, constellation_state, zre, zim, rep_re, rep_im);      // v27ter_rx.c:765-777
                if (do_track)
                {
                    const float error = zim*tgt_re - zre*tgt_im;
                    carrier_phase_rate += v29_f2i(use_track_i*error);
                    carrier_phase += (uint32_t) v29_f2i(use_track_p*error);
                }
                if (do_tune)
                {
                    const float ere = (tgt_re - zre)*eq_delta;
                    const float eim = (tgt_im - zim)*eq_delta;
#pragma unroll
                    for (int i = 0;  i < EQN;  i++)
                    {
                        const float2 c0 = TAP(i);
                        const f32x2v u = f32x2v{xim[i], xre[i]}*f32x2v{eim, eim};
                        const f32x2v w = f32x2v{xre[i], xim[i]}*f32x2v{ere, ere};
                        const f32x2v c = f32x2v{c0.x, c0.y}*f32x2v{0.9999f, 0.9999f} + (u + f32x2v{w.x, -w.y});
                        TAP(i) = make_float2(c.x, c.y);
                    }
                }
                if (do_save)
                {
                    carrier_phase_rate_save = carrier_phase_rate;
                    for (int k = 0;  k < EQN;  k++)
                    {
                        const float2 c = TAP(k);
                        stf(WF_EQ_SAVE + 2*k, c.x);
                        stf(WF_EQ_SAVE + 2*k + 1, c.y);
                    }
                }
        carrier_phase += (uint32_t) carrier_phase_rate;     // dds_advancef() with the rate the baud left behind
    }
    }
    }

    stf(WF_AGC, agc_scaling);
    stf(WF_AGC_SAVE, agc_scaling_save);
    stf(WF_TRAIN_ERR, training_error);
    stf(WF_TRACK_P, carrier_track_p);
    stf(WF_TRACK_I, carrier_track_i);
    for (int i = 0;  i < kRrcLen;  i++)
        stf(WF_RRC + i, rrc_at(i));
    for (int i = 0;  i < EQN;  i++)
    {
        const float2 c = TAP(i);
        stf(WF_EQ_COEFF + 2*i, c.x);
        stf(WF_EQ_COEFF + 2*i + 1, c.y);
    }
    if (__any(eq_clear_pending))
    {
        if (eq_clear_pending)
        {
#pragma unroll
            for (int i = 0;  i < EQN;  i++)
            {
                xre[i] = 0.0f;
                xim[i] = 0.0f;
            }
            eq_clear_pending = false;
        }
    }
#pragma unroll
    for (int i = 0;  i < EQN;  i++)
    {
        const int k = (eq_step + i) & (EQN - 1);
        stf(WF_EQ_BUF + 2*k, xre[i]);
        stf(WF_EQ_BUF + 2*k + 1, xim[i]);
    }
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
    if constexpr (QAM)
        L.qam_count[ch] = n_q;
#undef RRC2
#undef TAP
}

}   // namespace spg
