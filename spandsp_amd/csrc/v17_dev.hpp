// v17_dev.hpp -- device side of the batched V.17 (and V.32bis 4800) receiver (reference: src/v17rx.c:214-1358;
// primitives as in v29_dev.hpp).  Same mapping and baud-aligned execution as the V.29 bank -- one channel per lane,
// RRC delay line / PCM tile / equaliser taps index-major in LDS, equaliser delay line in VGPRs in age order, reference
// summation order -- plus what V.17 adds:
//   * the 8-state trellis decoder (v17rx.c:396-589): accumulated path metrics in VGPRs, the 16-deep survivor memory
//     packed in LDS per lane (3 bits per predecessor state, 1 byte per surviving point), traceback as 15 dependent
//     LDS reads per baud;
//   * the soft-decision map of the bank's bit rate (36 x 36 x 8 bytes) and its constellation in LDS;
//   * the 192-phase pulse shaper (2 x 20 KB) stays in HBM/L2, transposed to [tap][phase] so that one wave's gather
//     for a tap falls in a 768 byte row;
//   * long and short training state machines, with the restart a carrier drop triggers choosing between them.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "v29_dev.hpp"

#include "v17_common.hpp"

namespace spg {

// WPB, TILE, PK16: several waves per workgroup sharing the tables, a short PCM tile and the RRC delay line as packed
// int16 pairs, for banks of full waves -- see v29_bank_kernel.  With the survivor memory a V.17 wave needs 45 KB of its
// own: three of them and the 20 KB of tables fill a CU (155 KB), where the one-wave workgroups of 81 KB put two on it.
template <int CPW, bool QAM = false, int WPB = 1, int TILE = 0, bool PK16 = false>
__global__ __launch_bounds__(64*WPB)
void v17_bank_kernel(const V17Launch L)
{
    static_assert(WPB == 1  ||  CPW == 64, "several waves per workgroup: full waves only");
    __shared__ float t_sine[2048];
    __shared__ float t_con[256];
    __shared__ uint32_t t_map[36*36*2];
    __shared__ uint16_t t_sqrt[194];
    // per-lane RRC delay line (doubled) + survivor memory, PCM tile, equaliser taps: all index-major [word][CPW]
    // (the RRC delay line as zero padded pairs, see v29_dev.hpp)
    __shared__ float2 lanes[PK16  ?  1  :  WPB*CPW*2*kRrcLen];
    __shared__ uint32_t lanes16[PK16  ?  WPB*CPW*2*kRrcLen  :  1];
    __shared__ uint32_t surv[WPB*CPW*(16 + 32)];
    // (a full wave's LDS must stay under half a CU's 160 KB so that two waves share a CU: shorter PCM tile there)
    constexpr int kTile = (TILE > 0)  ?  TILE  :  (CPW == 64)  ?  32  :  kPcmTile;
    static_assert(kTile%8 == 0, "the PCM tile is staged in 16-byte pieces");
    __shared__ uint32_t pcm[WPB*CPW*(kTile/2)];
    __shared__ float2 taps[WPB*kEqLen*CPW];

    const int lane = threadIdx.x & 63;
    const int wv = (WPB == 1)  ?  0  :  (int) (threadIdx.x >> 6);
    const int ch = (blockIdx.x*WPB + wv)*CPW + lane;
    constexpr int kThreads = 64*WPB;
    const int tid = threadIdx.x;
    const V17Tables &TB = *L.tab;
    const float *g_rrc_re = TB.rrc_re;
    const float *g_rrc_im = TB.rrc_im;

    for (int i = tid;  i < 2048;  i += kThreads)
        t_sine[i] = TB.sine[i];
    for (int i = tid;  i < 256;  i += kThreads)
        t_con[i] = TB.con[i];
    for (int i = tid;  i < 36*36*2;  i += kThreads)
        t_map[i] = TB.map[i];
    for (int i = tid;  i < 194;  i += kThreads)
        t_sqrt[i] = TB.sqrt_tab[i];
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
    __syncthreads();
    if (lane >= CPW  ||  ch >= L.n_ch)
        return;

    // bank-wide constants of the bit rate (v17rx.c:1403-1436)
    const int bit_rate = L.bit_rate;
    const int bits_per_symbol = (bit_rate == 14400)  ?  6  :  (bit_rate == 12000)  ?  5  :  (bit_rate == 9600)  ?  4  :  (bit_rate == 7200)  ?  3  :  2;
    const int space_map = (bit_rate == 12000)  ?  1  :  (bit_rate == 9600)  ?  2  :  (bit_rate == 7200)  ?  3  :  0;
    const float spacing = (space_map == 0)  ?  1.414f  :  (space_map == 1)  ?  2.0f  :  (space_map == 2)  ?  2.828f  :  4.0f;

    const size_t N = (size_t) L.n_ch;
    const int mylen = L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;
    auto ldf = [&](int w) { return __uint_as_float(L.state[(size_t) w*N + ch]); };
    auto ldi = [&](int w) { return (int32_t) L.state[(size_t) (kV17Floats + w)*N + ch]; };
    // A float word goes back as its bits -- except a NaN (a receiver whose equaliser has run away is full of them), which
    // goes back as x86's: there an invalid operation makes the negative quiet NaN and arithmetic hands an operand's NaN on
    // sign and all, while here the negated operand of a subtraction flips it.  Nothing ever depends on a NaN's sign.
    auto stf = [&](int w, float v) { L.state[(size_t) w*N + ch] = (v != v)  ?  0xFFC00000u  :  __float_as_uint(v); };
    auto sti = [&](int w, int32_t v) { L.state[(size_t) (kV17Floats + w)*N + ch] = (uint32_t) v; };

    float2 *rrc2 = &lanes[PK16  ?  0  :  (wv*CPW*2*kRrcLen + lane)];           // [2*27] pairs, stride CPW
    uint32_t *rrc16 = &lanes16[PK16  ?  (wv*CPW*2*kRrcLen + lane)  :  0];
    uint32_t *pcmw = &pcm[wv*CPW*(kTile/2)];
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
    uint32_t *past = &surv[wv*CPW*(16 + 32) + lane];    // [16]: 8 x 3 bit predecessor states per time step
    uint32_t *full = past + 16*CPW;                     // [16][2]: 8 x 1 byte surviving points per time step
    float2 *ctap = &taps[wv*kEqLen*CPW + lane];
#define TAP(i)      ctap[(i)*CPW]
#define PAST(t)     past[(t)*CPW]
#define FULL(t, h)  full[(2*(t) + (h))*CPW]

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
    for (int i = 0;  i < kRrcLen;  i++)
    {
        const float v = ldf(VF_RRC + i);
        if (!PK16)
        {
            rrc2[i*CPW] = make_float2(v, 0.0f);
            rrc2[(kRrcLen + i)*CPW] = make_float2(0.0f, v);
        }
        rrc_put(i, v);
    }
    for (int i = 0;  i < kEqLen;  i++)
        TAP(i) = make_float2(ldf(VF_EQ_COEFF + 2*i), ldf(VF_EQ_COEFF + 2*i + 1));
    // equaliser delay line in age order: xre[i] = eq_buf[(eq_step + i) mod 33] (i = 0 oldest)
    float xre[kEqLen];
    float xim[kEqLen];
    bool eq_clear_pending = false;
    bool restart_pending = false;
    {
        const int es = ldi(XI_EQ_STEP);
#pragma unroll
        for (int i = 0;  i < kEqLen;  i++)
        {
            int k = es + i;
            k = (k >= kEqLen)  ?  (k - kEqLen)  :  k;
            xre[i] = ldf(VF_EQ_BUF + 2*k);
            xim[i] = ldf(VF_EQ_BUF + 2*k + 1);
        }
    }
    float sd[8];
#pragma unroll
    for (int i = 0;  i < 8;  i++)
        sd[i] = ldf(XF_DIST + i);
    for (int t = 0;  t < 16;  t++)
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
    auto diff_ld = [&](int k) { return ldi(XI_DIFF_ANGLES + (k & 0xF)); };
    auto diff_st = [&](int k, int32_t v) { sti(XI_DIFF_ANGLES + (k & 0xF), v); };

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

    // v17_rx_restart(s, s->bit_rate, s->short_train), v17rx.c:1399-1500
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
#pragma unroll
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
        // (the equaliser delay line is register state: it is cleared where it is next looked at, see v29_dev.hpp)
        eq_clear_pending = true;
        eq_put_step = kV17Sets*10/(3*2) - 1;
        eq_step = 0;
        eq_skip = 0;
        if (short_train)
        {
            carrier_phase_rate = carrier_phase_rate_save;
            for (int i = 0;  i < kEqLen;  i++)
                TAP(i) = make_float2(ldf(VF_EQ_SAVE + 2*i), ldf(VF_EQ_SAVE + 2*i + 1));
            eq_delta = 0.1f*(0.21f/kEqLen);
            agc_scaling = agc_scaling_save;
            carrier_track_i = 0.0f;
            carrier_track_p = 40000.0f;
        }
        else
        {
            carrier_phase_rate = v29_f2i(1800.0f*65536.0f*65536.0f/8000);
            for (int i = 0;  i < kEqLen;  i++)
                TAP(i) = make_float2((i == 16)  ?  3.0f  :  0.0f, 0.0f);
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
            ys[i] = y[i*kV17Sets];
        }
        f32x2v a = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0;  i < kRrcLen;  i++)
            a += xs[i]*(f32x2v) {ys[i], ys[i]};
        return a.x + a.y;
    };
    // track_carrier() and tune_equalizer() are requested by the stage logic and carried out once, after it, with the
    // loop gains and step size as they were when the reference would have called them (see v29_dev.hpp).
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
    auto spin = [&](uint32_t phase_step)
    {
        const float p = phase_step*2.0f*3.1415926f/(65536.0f*65536.0f);
        const float zc = spg_sincosf(p, true);
        const float zs = -spg_sincosf(p, false);
#pragma unroll
        for (int k = 0;  k < kEqLen;  k++)
        {
            const float xr = xre[k];
            const float xi = xim[k];
            xre[k] = xr*zc - xi*zs;
            xim[k] = xr*zs + xi*zc;
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
            const int cs = (int) ((t_map[cell >> 2] >> (8*(cell & 3))) & 0xFF);
            // v32bis_4800_differential_decoder[diff][cs] = {{2,3,0,1},{0,2,1,3},{3,1,2,0},{1,0,3,2}}, 2 bits per entry
            constexpr uint32_t dec = (2u | 3u << 2 | 0u << 4 | 1u << 6) | (0u | 2u << 2 | 1u << 4 | 3u << 6) << 8
                                   | (3u | 1u << 2 | 2u << 4 | 0u << 6) << 16 | (1u | 0u << 2 | 3u << 4 | 2u << 6) << 24;
            const int raw = (int) ((dec >> (2*(diff*4 + cs))) & 3u);
            diff = cs;
            put_bit(raw);
            put_bit(raw >> 1);
            return cs;
        }
        const uint32_t c_lo = t_map[(re*36 + im)*2];
        const uint32_t c_hi = t_map[(re*36 + im)*2 + 1];
        int cell[8];
        float dist[8];
#pragma unroll
        for (int i = 0;  i < 8;  i++)
        {
            cell[i] = (int) (((i < 4  ?  c_lo  :  c_hi) >> (8*(i & 3))) & 0xFF);
            const float dx = t_con[2*cell[i]] - zre;
            const float dy = t_con[2*cell[i] + 1] - zim;
            dist[i] = dx*dx + dy*dy;
        }
        float mn = 9999999.0f;
        int cs = cell[0];
#pragma unroll
        for (int i = 0;  i < 8;  i++)
        {
            if (mn > dist[i])
            {
                mn = dist[i];
                cs = cell[i];
            }
        }
        track_carrier(t_con[2*cs], t_con[2*cs + 1]);

        if (++trellis_ptr >= 16)
            trellis_ptr = 0;
        constexpr int tp[8][4] =
        {
            {0, 6, 2, 4}, {6, 0, 4, 2}, {2, 4, 0, 6}, {4, 2, 6, 0}, {1, 3, 7, 5}, {5, 7, 3, 1}, {7, 5, 1, 3}, {3, 1, 5, 7}
        };
        float nd[8];
        uint32_t pw = 0;
        uint32_t f_lo = 0;
        uint32_t f_hi = 0;
#pragma unroll
        for (int i = 0;  i < 8;  i++)
        {
            const int set = i >> 2;
            float best = dist[tp[i][0]] + sd[set];
            float sel_sd = sd[set];
            float sel_d = dist[tp[i][0]];
            int sel_cell = cell[tp[i][0]];
            int kk = set;
#pragma unroll
            for (int j = 1;  j < 4;  j++)
            {
                const int k = (j << 1) + set;
                const float t = dist[tp[i][j]] + sd[k];
                if (best > t)
                {
                    best = t;
                    sel_sd = sd[k];
                    sel_d = dist[tp[i][j]];
                    sel_cell = cell[tp[i][j]];
                    kk = k;
                }
            }
            nd[i] = sel_sd*0.9f + sel_d*0.1f;
            pw |= (uint32_t) kk << (3*i);
            if (i < 4)
                f_lo |= (uint32_t) sel_cell << (8*i);
            else
                f_hi |= (uint32_t) sel_cell << (8*(i - 4));
        }
        PAST(trellis_ptr) = pw;
        FULL(trellis_ptr, 0) = f_lo;
        FULL(trellis_ptr, 1) = f_hi;
#pragma unroll
        for (int i = 0;  i < 8;  i++)
            sd[i] = nd[i];
        mn = sd[0];
        int k = 0;
#pragma unroll
        for (int i = 1;  i < 8;  i++)
        {
            if (mn > sd[i])
            {
                mn = sd[i];
                k = i;
            }
        }
        int j = trellis_ptr;
        for (int i = 0;  i < 15;  i++)
        {
            k = (int) ((PAST(j) >> (3*k)) & 7u);
            j = (j - 1) & 15;
        }
        const int nearest = (int) ((FULL(j, k >> 2) >> (8*(k & 3))) & 0xFF) >> 1;
        int raw = (nearest & 0x3C) | (((nearest & 3) - diff) & 3);      // v17_differential_decoder[diff][nearest & 3]
        diff = nearest & 3;
        for (int i = 0;  i < bits_per_symbol;  i++)
        {
            put_bit(raw);
            raw >>= 1;
        }
        return cs;
    };

    const int16_t *src = L.amp + (size_t) ch*L.stride;
    for (int tile = 0;  tile < L.samples;  tile += kTile)
    {
    const int tn = max(0, min(kTile, mylen - tile));         // per lane when the call carries per-channel lengths
    // ---- stage this lane's stretch of PCM: pcm[k][lane] = samples 2k, 2k+1 of the tile ----------------------
    {
        const int16_t *row = src + tile;
        const bool wide = ((((uintptr_t) row) & 15) == 0)  &&  (tn == kTile);
        if (wide)
        {
#pragma unroll
            for (int k = 0;  k < kTile/8;  k++)
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
    // ---- phase A: every lane runs its own samples up to its next T/2 instant -------------------------------
    bool ready = false;
    int power = 0;
    int step = 0;
    float sre = 0.0f;
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

        // signal_detect(), v17rx.c:1133-1210 (IAXMODEM_STUFF is #defined at v17rx.c:1)
        {
            const int x = amp >> 1;
            int d = (int) (short) (x - last_sample);
            last_sample = x;
            power_reading += ((d*d - power_reading) >> 4);
            power = power_reading;
            d = (int) (short) abs(d);
            if (10*d < high_sample)
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
                if (d > high_sample)
                    high_sample = d;
            }
            if (signal_present > 0)
            {
                if (drop_pending  ||  power < carrier_off_power)
                {
                    if (--signal_present <= 0)
                    {
                        // v17_rx_restart(): carried out right after this loop (see v29_dev.hpp)
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
        if (power == 0  ||  stage == V17_PARKED)
            break;

        eq_put_step -= kV17Sets;
        step = -eq_put_step;
        if (step < 0)
            step += kV17Sets;
        step = max(0, min(kV17Sets - 1, step));
        float v = rrc_dot(g_rrc_re, step);
        sre = v*agc_scaling;
        {
            float t = glow0*g0 + glow1*g1 + sre;
            glow1 = glow0;
            glow0 = t;
            t = ghigh0*g3 + ghigh1*g4 + sre;
            ghigh1 = ghigh0;
            ghigh0 = t;
        }
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
            for (int i = 0;  i < kEqLen;  i++)
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
        float v;
            if (agc_scaling_save == 0.0f)
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
                agc_scaling = (2.17f/1.000000f)/(float) root_power;
            }
            v = rrc_dot(g_rrc_im, step);
            const float sim = v*agc_scaling;
            const float dre = t_sine[(uint32_t) (carrier_phase + (1u << 30)) >> 21];
            const float dim = t_sine[carrier_phase >> 21];
            const float hre = sre*dre - sim*dim;
            const float him = -sre*dim - sim*dre;
            eq_put_step += kV17Sets*10/(3*2);

            // ---- process_half_baud(), v17rx.c:592-1130 ----
#pragma unroll
            for (int i = 0;  i < kEqLen - 1;  i++)
            {
                xre[i] = xre[i + 1];
                xim[i] = xim[i + 1];
            }
            xre[kEqLen - 1] = hre;
            xim[kEqLen - 1] = him;
            if (++eq_step >= kEqLen)
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
                {
                    const int split = kEqLen - eq_step;
                    float2 cs_[kEqLen];
#pragma unroll
                    for (int i = 0;  i < kEqLen;  i++)
                        cs_[i] = TAP(i);
                    f32x2v acc = f32x2v{0.0f, 0.0f};
                    f32x2v fst = f32x2v{0.0f, 0.0f};
#pragma unroll
                    for (int i = 0;  i < kEqLen;  i++)
                    {
                        if (i == split)
                        {
                            fst = acc;
                            acc = f32x2v{0.0f, 0.0f};
                        }
                        const f32x2v t1 = f32x2v{xre[i], xre[i]}*f32x2v{cs_[i].x, cs_[i].y};
                        const f32x2v t2 = f32x2v{xim[i], xim[i]}*f32x2v{cs_[i].y, cs_[i].x};
                        acc += t1 + f32x2v{-t2.x, t2.y};
                    }
                    zre = fst.x + acc.x;
                    zim = fst.y + acc.y;
                }

                float tre;
                float tim;
                do_track = false;
                do_tune = false;
                do_save = false;
                int cs = 0;
                float rep_re = 0.0f;                        // `target` of process_half_baud(), for the qam report
                float rep_im = 0.0f;
                if (stage == V17_NORMAL  ||  stage == V17_TCM_WINDUP  ||  stage == V17_TEST_ONES)
                {
                    cs = decode_baud(zre, zim);
                    if constexpr (QAM)
                    {
                        rep_re = t_con[2*cs];
                        rep_im = t_con[2*cs + 1];
                    }
                }
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
                        {
                            int skipped = descramble(1);
                            skipped = (skipped << 1) | descramble(1);
                            if constexpr (QAM)
                                cdba(skipped, rep_re, rep_im);
                        }
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
                    rep_re = tre;
                    rep_im = tim;
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
                    rep_re = tre;
                    rep_im = tim;
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
                    rep_re = tre;
                    rep_im = tim;
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
                    rep_re = zre;                           // target = &z (v17rx.c:914)
                    rep_im = zim;
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
                        {
                            int skipped = descramble(1);
                            skipped = (skipped << 1) | descramble(1);
                            if constexpr (QAM)
                                cdba(skipped, rep_re, rep_im);
                        }
                        training_error = 0.0f;
                        training_count = 1;
                        stage = V17_SHORT_TRAIN_ON_CDBA_AND_TEST;
                    }
                    else
                    {
                        cdba((training_count & 1) + 2, tre, tim);
                        rep_re = tre;
                        rep_im = tim;
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
                    rep_re = tre;
                    rep_im = tim;
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
                    const float ere = zre - t_con[2*cs];
                    const float eim = zim - t_con[2*cs + 1];
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
                    const float ere = zre - t_con[2*cs];
                    const float eim = zim - t_con[2*cs + 1];
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
                qam_report(0, cs, zre, zim, rep_re, rep_im);                        // v17rx.c:1117-1131
                if (do_track)
                {
                    const float error = zim*tgt_re - zre*tgt_im;
                    carrier_phase_rate += v29_f2i(use_track_i*error);
                    carrier_phase += (uint32_t) v29_f2i(use_track_p*error);
                }
                if (do_tune)
                {
                    const float ere = (tgt_re - zre)*use_delta;
                    const float eim = (tgt_im - zim)*use_delta;
#pragma unroll
                    for (int i = 0;  i < kEqLen;  i++)
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
                    for (int k = 0;  k < kEqLen;  k++)
                    {
                        const float2 c = TAP(k);
                        stf(VF_EQ_SAVE + 2*k, c.x);
                        stf(VF_EQ_SAVE + 2*k + 1, c.y);
                    }
                }
        carrier_phase += (uint32_t) carrier_phase_rate;     // dds_advancef() with the rate the baud left behind
    }
    }
    }

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
    for (int i = 0;  i < kRrcLen;  i++)
        stf(VF_RRC + i, rrc_at(i));
    for (int i = 0;  i < kEqLen;  i++)
    {
        const float2 c = TAP(i);
        stf(VF_EQ_COEFF + 2*i, c.x);
        stf(VF_EQ_COEFF + 2*i + 1, c.y);
    }
    if (__any(eq_clear_pending))
    {
        if (eq_clear_pending)
        {
#pragma unroll
            for (int i = 0;  i < kEqLen;  i++)
            {
                xre[i] = 0.0f;
                xim[i] = 0.0f;
            }
            eq_clear_pending = false;
        }
    }
#pragma unroll
    for (int i = 0;  i < kEqLen;  i++)
    {
        int k = eq_step + i;
        k = (k >= kEqLen)  ?  (k - kEqLen)  :  k;
        stf(VF_EQ_BUF + 2*k, xre[i]);
        stf(VF_EQ_BUF + 2*k + 1, xim[i]);
    }
#pragma unroll
    for (int i = 0;  i < 8;  i++)
        stf(XF_DIST + i, sd[i]);
    for (int t = 0;  t < 16;  t++)
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
    if constexpr (QAM)
        L.qam_count[ch] = n_q;
#undef RRC2
#undef TAP
#undef PAST
#undef FULL
}

}   // namespace spg
