// v29_dev.hpp -- device side of the batched V.29 receiver (reference: src/v29rx.c:400-965,
// src/godard.c:144-220, src/vector_float.c:890-939, src/complex_vector_float.c:137-219,
// src/power_meter.c:65, src/math_fixed.c:158, src/dds_float.c:2135-2177, spandsp/arctan2.h).
//
// Mapping: ONE CHANNEL PER LANE, one wavefront (64 channels) per workgroup.  The receiver is
// a per-channel state machine (carrier detect, AGC, polyphase RRC, Godard timing recovery,
// T/2 equaliser with LMS training, PI carrier loop, slicer, descrambler, training FSM); the
// only regular arithmetic is three short inner products per T/2 instant with per-channel
// operands and per-channel circular offsets -- nothing a matrix core can use.  So:
//   * scalars and the equaliser delay line live in VGPRs: the delay line is kept in age order
//     (a 66-register shift per T/2 instant), so every access has a compile-time index; the
//     reference's circular position survives only as the per-lane split point of the
//     summation and as the order in which state is stored.  The 33 complex taps sit in LDS,
//     [tap][lane], also always indexed by a compile-time tap number;
//   * the RRC delay line lives in a per-lane LDS column, stored twice back to back so
//     x[(pos + i) mod n] is the contiguous x2[pos + i], as zero padded pairs ({x, 0} then {0, x}) so that
//     one packed multiply-add per tap forms both partial sums of the circular inner product; the frame's
//     PCM is staged in LDS too;
//   * the polyphase RRC table (48 x 27 x {re, im}) and the sine table live once per
//     workgroup in LDS and are gathered by per-lane row;
//   * inner products keep the reference's exact summation tree: ascending coefficient
//     index, the circular split summed separately and added last (a per-lane split point,
//     handled by the zero padding above for the RRC filter and by snapshotting the accumulator,
//     instead of branching, for the equaliser).
// Execution is BAUD ALIGNED: a round of the main loop is one baud of every lane -- each lane
// consumes samples from its LDS tile until ITS next T/2 instant, all lanes run the half-baud
// phase together, twice, and then the baud phase (equaliser output, stage logic, carrier and
// tap updates) runs once with all lanes in step; a lane that enters a round in the middle of
// its baud sits out the first half.  Stepping all lanes sample by sample instead made the
// wave execute the half-baud and baud paths on nearly every sample with a fraction of its
// lanes (measured: 557 -> 450 us per 16 384 x 160 launch, V.27ter 890 -> 320, V.17 1480 -> 585).
// Expect VALU-issue bound behaviour and a low HBM figure for this kernel.
//
// Numerics: fp32, every op rounded separately (-ffp-contract=off), float->int conversions
// with the x86 out-of-range result the reference build has; the single libm dependency of
// the reference path (cosf/sinf of the training phase spin, v29rx.c:618-623) is computed
// with the C library's own algorithm (spg_sincosf below), so it has the same bits too.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "v29_common.hpp"

namespace spg {

// CPW = channels per workgroup (one wavefront): 64 when the bank is big enough to fill every SIMD of the chip with
// full waves, fewer (idle upper lanes) for small banks so that the channels still spread over all 1024 SIMDs.
// WPB waves per workgroup share the tables (WPB > 1 goes with CPW = 64); TILE = samples of PCM staged per lane at a
// time; PK16: the RRC delay line -- raw 16 bit samples -- is kept as packed pairs of int16 and converted on the way to
// the multipliers.  A full wave's per-lane LDS is then 13.8 KB (delay line) + 16.9 KB (equaliser taps) + the PCM tile
// instead of 27.6 + 16.9 + 10 KB: with four waves sharing 19.5 KB of tables a CU holds four waves, one per SIMD, where
// the one-wave workgroups of 74 KB left two SIMDs of every CU empty on banks of 64 K channels and more.
template <int CPW, bool QAM = false, int WPB = 1, int TILE = kPcmTile, bool PK16 = false>
__global__ __launch_bounds__(64*WPB)
void v29_bank_kernel(const V29Launch L)
{
    static_assert(WPB == 1  ||  CPW == 64, "several waves per workgroup: full waves only");
    static_assert(TILE%8 == 0  &&  TILE >= 8, "the PCM tile is staged in 16-byte pieces");
    // coefficient tables transposed to [tap][set]: lanes on different polyphase sets hit different banks
    __shared__ float t_rrc_re[kRrcSets*kRrcLen];
    __shared__ float t_rrc_im[kRrcSets*kRrcLen];
    __shared__ float t_sine[2048];
    __shared__ float t_const[32];
    __shared__ uint16_t t_sqrt[194];
    __shared__ uint8_t t_map[400];
    // per-lane RRC delay line and PCM tile, index-major [word][CPW]: lane l always uses the banks of (l mod 32)
    // whatever its position.  The delay line is stored as 2*27 pairs: pair k < 27 is {x[k], 0}, pair 27 + k is
    // {0, x[k]} -- the window of 27 pairs starting at the circular position then holds, in .x, the head part of
    // vec_circular_dot_prodf()'s sum padded with zeros and, in .y, zeros followed by its wrapped part.
    __shared__ float2 lanes[PK16  ?  1  :  WPB*CPW*2*kRrcLen];
    __shared__ uint32_t lanes16[PK16  ?  WPB*CPW*2*kRrcLen  :  1];     // PK16: pair k = x[k] | 0 << 16, pair 27 + k = 0 | x[k] << 16
    __shared__ uint32_t pcm[WPB*CPW*(TILE/2)];
    // equaliser taps {re, im}, [tap][lane]: always indexed by a compile-time tap number
    __shared__ float2 taps[WPB*kEqLen*CPW];

    const int lane = threadIdx.x & 63;
    const int wv = (WPB == 1)  ?  0  :  (int) (threadIdx.x >> 6);
    const int ch = (blockIdx.x*WPB + wv)*CPW + lane;
    const V29Tables &TB = *L.tab;
    constexpr int kThreads = 64*WPB;
    const int tid = threadIdx.x;

    // ---- tables -> LDS ----------------------------------------------------------------------
    for (int i = tid;  i < kRrcSets*kRrcLen;  i += kThreads)
    {
        const int set = i/kRrcLen;
        const int tap = i - set*kRrcLen;
        t_rrc_re[tap*kRrcSets + set] = TB.rrc_re[i];
        t_rrc_im[tap*kRrcSets + set] = TB.rrc_im[i];
    }
    for (int i = tid;  i < 2048;  i += kThreads)
        t_sine[i] = TB.sine[i];
    for (int i = tid;  i < 194;  i += kThreads)
        t_sqrt[i] = TB.sqrt_tab[i];
    for (int i = tid;  i < 400;  i += kThreads)
        t_map[i] = TB.space_map[i];
    if (tid < 16)
    {
        // v29tx_constellation_maps.h:58-77
        const float re[16] = {3, 1, 0, -1, -3, -1, 0, 1, 5, 3, 0, -3, -5, -3, 0, 3};
        const float im[16] = {0, 1, 3, 1, 0, -1, -3, -1, 0, 3, 5, 3, 0, -3, -5, -3};
        t_const[2*tid] = re[tid];
        t_const[2*tid + 1] = im[tid];
    }
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

    // ---- state -> registers / LDS ---------------------------------------------------------------
    const size_t N = (size_t) L.n_ch;
    const int mylen = L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;
    auto ldf = [&](int w) { return __uint_as_float(L.state[(size_t) w*N + ch]); };
    auto ldi = [&](int w) { return (int32_t) L.state[(size_t) (kV29Floats + w)*N + ch]; };
    // A float word goes back as its bits -- except a NaN (a receiver whose equaliser has run away is full of them), which
    // goes back as x86's: there an invalid operation makes the negative quiet NaN and arithmetic hands an operand's NaN on
    // sign and all, while here the negated operand of a subtraction flips it.  Nothing ever depends on a NaN's sign.
    auto stf = [&](int w, float v) { L.state[(size_t) w*N + ch] = (v != v)  ?  0xFFC00000u  :  __float_as_uint(v); };
    auto sti = [&](int w, int32_t v) { L.state[(size_t) (kV29Floats + w)*N + ch] = (uint32_t) v; };

    float2 *rrc2 = &lanes[PK16  ?  0  :  (wv*CPW*2*kRrcLen + lane)];           // [2*27] pairs, stride CPW
    uint32_t *rrc16 = &lanes16[PK16  ?  (wv*CPW*2*kRrcLen + lane)  :  0];
    uint32_t *pcmw = &pcm[wv*CPW*(TILE/2)];
    // delay line element k <- v (both of its pairs); the element back as a float
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
    float2 *ctap = &taps[wv*kEqLen*CPW + lane];
#define TAP(i)      ctap[(i)*CPW]
    for (int i = 0;  i < kEqLen;  i++)
        TAP(i) = make_float2(ldf(VF_EQ_COEFF + 2*i), ldf(VF_EQ_COEFF + 2*i + 1));
    // equaliser delay line in age order: xre[i] = eq_buf[(eq_step + i) mod 33] (i = 0 oldest)
    float xre[kEqLen];
    float xim[kEqLen];
    bool eq_clear_pending = false;
    bool restart_pending = false;
    {
        const int es = ldi(VI_EQ_STEP);
#pragma unroll
        for (int i = 0;  i < kEqLen;  i++)
        {
            int k = es + i;
            k = (k >= kEqLen)  ?  (k - kEqLen)  :  k;
            xre[i] = ldf(VF_EQ_BUF + 2*k);
            xim[i] = ldf(VF_EQ_BUF + 2*k + 1);
        }
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
    // diff_angles[16] is only touched during WAIT_FOR_CDCD: keep it in the state array in HBM
    auto diff_ld = [&](int k) { return ldi(VI_DIFF_ANGLES + (k & 0xF)); };
    auto diff_st = [&](int k, int32_t v) { sti(VI_DIFF_ANGLES + (k & 0xF), v); };

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

    // v29rx.c:1019-1098 (old_train == false, the only way the receive path calls it)
    auto restart = [&]()
    {
        for (int i = 0;  i < 2*kRrcLen;  i++)
        {
            if (PK16)
                rrc16[i*CPW] = 0;
            else
                rrc2[i*CPW] = make_float2(0.0f, 0.0f);
        }
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
            TAP(i) = make_float2((i == 16)  ?  3.0f  :  0.0f, 0.0f);       // V29_EQUALIZER_PRE_LEN
        // The equaliser delay line is register state, and this path is rare: clearing it here, inside the sample loop,
        // made every iteration of that loop pay ~450 register moves at the merge points.  It is cleared where it is
        // next looked at instead (start of the T/2 phase, write-back).
        eq_clear_pending = true;
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

    // vec_circular_dot_prodf(rrc_filter, coeffs[row], 27, rrc_step)   (vector_float.c:890-939)
    auto rrc_dot = [&](const float *table, int row)
    {
        const float *y = table + row;
        const float2 *x = rrc2 + rrc_step*CPW;
        const uint32_t *xq = rrc16 + rrc_step*CPW;
        // the LDS reads go out nine taps at a time, ahead of that group's part of the summation chain: all 54 at once
        // kept 81 registers live across a kernel that already overflows into AGPRs
        f32x2v a = {0.0f, 0.0f};
#pragma unroll
        for (int i0 = 0;  i0 < kRrcLen;  i0 += 9)
        {
            f32x2v xs[9];
            float ys[9];
#pragma unroll
            for (int i = 0;  i < 9;  i++)
            {
                if (PK16)
                {
                    const uint32_t q = xq[(i0 + i)*CPW];
                    xs[i] = (f32x2v) {(float) (int) (short) (q & 0xFFFFu), (float) ((int) q >> 16)};
                }
                else
                {
                    const float2 w = x[(i0 + i)*CPW];
                    xs[i] = (f32x2v) {w.x, w.y};
                }
                ys[i] = y[(i0 + i)*kRrcSets];
            }
            // .x: x[pos..n) . y[0..n-pos) then + 0*y (exact: a running sum that starts at +0 is never -0);
            // .y: 0*y then x[0..pos) . y[n-pos..n) -- the reference's two partial sums, each in its own order
#pragma unroll
            for (int i = 0;  i < 9;  i++)
                a += xs[i]*(f32x2v) {ys[i], ys[i]};
        }
        return a.x + a.y;
    };

    // track_carrier() and tune_equalizer() (v29rx.c:281-331) are requested by the stage logic and carried out once,
    // after it, with the loop gains as they were when the reference would have called them.
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
            int re = (int) ((zre + 5.0f)*2.0f);
            int im = (int) ((zim + 5.0f)*2.0f);
            re = max(0, min(19, re));
            im = max(0, min(19, im));
            nearest = t_map[re*20 + im];
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
        const float tre = t_const[2*nearest];
        const float tim = t_const[2*nearest + 1];
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
    // One round = one BAUD of every lane: two T/2 instants (a lane that enters the round in the middle of its baud
    // sits out the first), then the baud phase once, with all lanes in step.
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
        // ---- v29_rx(), v29rx.c:885-961 --------------------------------------------------------
        rrc_put(rrc_step, (float) amp);
        if (++rrc_step >= kRrcLen)
            rrc_step = 0;

        // signal_detect(), v29rx.c:788-865 (with the IAXMODEM_STUFF this snapshot #defines)
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
                        // v29_rx_restart() rewrites some forty state variables: done right after this loop (the lane
                        // takes no further sample until then), because with it inline every iteration of the loop
                        // paid for the register copies at its merge points
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
        if (power == 0  ||  stage == V29_PARKED)
            break;

        eq_put_step -= kRrcSets;
        step = -eq_put_step;
        if (step < 0)
            step += kRrcSets;
        step = max(0, min(kRrcSets - 1, step));
        float v = rrc_dot(t_rrc_re, step);
        sre = v*agc_scaling;
        {
            // godard_ted_rx(), godard.c:144-162
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
                // fixed_sqrt32(), math_fixed.c:158-169
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
                agc_scaling = (1.25f/1.0f)/(float) root_power;
            }
            v = rrc_dot(t_rrc_im, step);
            const float sim = v*agc_scaling;
            // dds_lookup_complexf(), dds_float.c:2135,2177
            const float dre = t_sine[(uint32_t) (carrier_phase + (1u << 30)) >> 21];
            const float dim = t_sine[carrier_phase >> 21];
            const float hre = sre*dre - sim*dim;
            const float him = -sre*dim - sim*dre;
            eq_put_step += kRrcSets*10/(3*2);

            // ---- process_half_baud(), v29rx.c:484-786 ----------------------------------------
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
    // (the reference advances the carrier phase after process_half_baud(), with the rate the baud processing may just
    // have changed; phase B already advanced it, so that advance is taken back here and redone at the end)
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
                // equalizer_get(): cvec_circular_dot_prodf (complex_vector_float.c:137-196)
                {
                    const int split = kEqLen - eq_step;
                    float2 cs[kEqLen];
#pragma unroll
                    for (int i = 0;  i < kEqLen;  i++)
                        cs[i] = TAP(i);
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
                        // {xr*c.x - xi*c.y, xr*c.y + xi*c.x}: two packed products, a packed add with one negated half
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
                if (stage == V29_NORMAL  ||  stage == V29_TEST_ONES)
                    decode_baud(zre, zim);
                switch (stage)
                {
                case V29_NORMAL:
                    if constexpr (QAM)
                    {
                        rep_re = t_const[2*constellation_state];
                        rep_im = t_const[2*constellation_state + 1];
                    }
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
                        // v29rx.c:618-624: spin the equaliser delay line and the carrier
                        const float p = ((uint32_t) angle)*2.0f*3.1415926f/(65536.0f*65536.0f);
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
                        carrier_phase += (uint32_t) angle;
                        const int bit = scrambled_training_bit();
                        constellation_state = (0x002030B0 >> (4*(training_cd + bit))) & 0xF;   // cdcd_pos = {0,11,0,3,0,2}
                        if constexpr (QAM)
                        {
                            rep_re = t_const[2*constellation_state];
                            rep_im = t_const[2*constellation_state + 1];
                        }
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
                    const float tre = t_const[2*constellation_state];
                    const float tim = t_const[2*constellation_state + 1];
                    rep_re = tre;
                    rep_im = tim;
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
                    const float tre = t_const[2*constellation_state];
                    const float tim = t_const[2*constellation_state + 1];
                    rep_re = tre;
                    rep_im = tim;
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
                    const float tre = t_const[2*constellation_state];
                    const float tim = t_const[2*constellation_state + 1];
                    rep_re = tre;
                    rep_im = tim;
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
                qam_report(0, constellation_state, zre, zim, rep_re, rep_im);      // v29rx.c:769-783
                if (do_track)
                {
                    const float error = zim*tgt_re - zre*tgt_im;
                    carrier_phase_rate += v29_f2i(use_track_i*error);
                    carrier_phase += (uint32_t) v29_f2i(use_track_p*error);
                }
                if (do_tune)
                {
                    // cvec_circular_lmsf (complex_vector_float.c:201-219)
                    const float ere = (tgt_re - zre)*eq_delta;
                    const float eim = (tgt_im - zim)*eq_delta;
#pragma unroll
                    for (int i = 0;  i < kEqLen;  i++)
                    {
                        const float2 c0 = TAP(i);
                        // {xi*eim + xr*ere, xr*eim - xi*ere}
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

    // ---- write back -----------------------------------------------------------------------------
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
        if constexpr (QAM)
            L.qam_count[ch] = n_q;
    }
}

#undef RRC2
#undef TAP

}   // namespace spg
