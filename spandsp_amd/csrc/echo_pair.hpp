// echo_pair.hpp -- the echo canceller kernel for big banks: TWO LANES PER CHANNEL, 32 channels per wavefront
// (reference: src/echo.c:120-661, src/spandsp/fir.h:121-183; the algorithm and its state words are those of
// echo_dev.hpp, whose header describes them).
//
// Why another mapping.  echo_can_update() is ~115 instructions of per-channel scalar control around the FIR and the LMS
// update, and the control is replicated in every lane of a channel: with G lanes per channel a wave pays it once per 64/G
// channels.  The kernels of echo_dev.hpp keep tap slices as one value per register (history shifts are register
// renaming inside a loop unrolled by the slice length), which at two lanes per channel needs 3 x 64 registers for the
// slices alone -- measured (tools/echo_ab.py, 131072 channels): 256 VGPRs + 89 spilled dwords, 682 us against 598 us at
// four lanes.  Here the 16-bit quantities are held PACKED, two to a register:
//   t32[TPL]   fir_taps32 slice                                  64 registers at 128 taps
//   tp[TPL/2]  fir_taps16[tap_set] slice, taps (2k, 2k+1) in register k          32
//   wp[TPL/2]  history slice in window order (sample 0 = newest), same packing    32
// and per sample and wave (32 channels):
//   history shift   v_alignbit_b32 x TPL/2 (+ one DPP move: lane 1 takes what leaves lane 0)
//   FIR             v_dot2_i32_i16 x TPL/2 into four accumulators, one DPP add (integer wrap-around: any order)
//   LMS update      v_mad_i32_i16 (op_sel picks the sample) x TPL, then the two 16-bit halves of tp[k] rewritten by
//                   SDWA shifts (v_lshrrev_b32 >> 15 into WORD_0 / WORD_1) x TPL -- when the step `factor` of some channel
//                   does not fit 16 bits (|clean_rx| >= 32768 with a quiet far end: legal, rare) the wave takes
//                   v_mad_i32_i24 on unpacked samples instead
//   control         once per wave
// No loop unrolling by phase is needed (the shift is arithmetic, not renaming), so the sample loop stays in the
// instruction cache.  As in echo_dev.hpp a sample on which any channel of the wave meets a set event is recognised before
// anything is committed, the shift is undone and the complete per-sample routine (`slow`) runs that sample.
#pragma once

#include "echo_dev.hpp"

namespace spg {

typedef short echo_s16x2 __attribute__((ext_vector_type(2)));

// c + a.lo*b.lo + a.hi*b.hi (v_dot2_i32_i16, wrap-around).  Through the builtin, not inline assembly: the dot
// instructions of this chip have read-after-write wait states of their own that the compiler only inserts for
// instructions it can see.
__device__ __forceinline__ int dot2_i16(int a, int b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(echo_s16x2, a), __builtin_bit_cast(echo_s16x2, b), c, false);
}

// One piece of lms_adapt() (echo.c:232-249) for four packed sample pairs w0..w3: the eight 32-bit taps a0..a7 take
// sample*f (16 x 16 bit products, wrap-around accumulate), then, if REPACK, the 16-bit coefficients p0..p3 are rewritten
// from bits 30..15 of the taps.  Only the lanes of `mask` take part: the piece narrows exec itself and restores it, so
// that to the compiler this is straight-line code.  (Written as a divergent branch around per-tap statements the update
// costs a register copy per tap and per coefficient at the merge -- 128 v_mov_b32 per sample, measured.)  The two halves
// of a coefficient register are written three instructions apart (sub-dword writes forward late on this chip).
template <bool REPACK>
__device__ __forceinline__ void lms_piece(int &a0, int &a1, int &a2, int &a3, int &a4, int &a5, int &a6, int &a7,
                                          int &p0, int &p1, int &p2, int &p3, int w0, int w1, int w2, int w3,
                                          int f, int fifteen, unsigned long long mask)
{
    unsigned long long saved;
    if (REPACK)
    {
        asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                     "v_mad_i32_i16 %[a0], %[w0], %[f], %[a0]\n\t"
                     "v_mad_i32_i16 %[a1], %[w0], %[f], %[a1] op_sel:[1,0,0,0]\n\t"
                     "v_mad_i32_i16 %[a2], %[w1], %[f], %[a2]\n\t"
                     "v_mad_i32_i16 %[a3], %[w1], %[f], %[a3] op_sel:[1,0,0,0]\n\t"
                     "v_mad_i32_i16 %[a4], %[w2], %[f], %[a4]\n\t"
                     "v_mad_i32_i16 %[a5], %[w2], %[f], %[a5] op_sel:[1,0,0,0]\n\t"
                     "v_mad_i32_i16 %[a6], %[w3], %[f], %[a6]\n\t"
                     "v_mad_i32_i16 %[a7], %[w3], %[f], %[a7] op_sel:[1,0,0,0]\n\t"
                     "v_lshrrev_b32_sdwa %[p0], %[sh], %[a0] dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
                     "v_lshrrev_b32_sdwa %[p1], %[sh], %[a2] dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
                     "v_lshrrev_b32_sdwa %[p2], %[sh], %[a4] dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
                     "v_lshrrev_b32_sdwa %[p3], %[sh], %[a6] dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
                     "v_lshrrev_b32_sdwa %[p0], %[sh], %[a1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
                     "v_lshrrev_b32_sdwa %[p1], %[sh], %[a3] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
                     "v_lshrrev_b32_sdwa %[p2], %[sh], %[a5] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
                     "v_lshrrev_b32_sdwa %[p3], %[sh], %[a7] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6),
                       [a7] "+v"(a7), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [sv] "=&s"(saved)
                     : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [f] "v"(f), [sh] "v"(fifteen), [m] "s"(mask)
                     : "scc");
    }
    else
    {
        asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                     "v_mad_i32_i16 %[a0], %[w0], %[f], %[a0]\n\t"
                     "v_mad_i32_i16 %[a1], %[w0], %[f], %[a1] op_sel:[1,0,0,0]\n\t"
                     "v_mad_i32_i16 %[a2], %[w1], %[f], %[a2]\n\t"
                     "v_mad_i32_i16 %[a3], %[w1], %[f], %[a3] op_sel:[1,0,0,0]\n\t"
                     "v_mad_i32_i16 %[a4], %[w2], %[f], %[a4]\n\t"
                     "v_mad_i32_i16 %[a5], %[w2], %[f], %[a5] op_sel:[1,0,0,0]\n\t"
                     "v_mad_i32_i16 %[a6], %[w3], %[f], %[a6]\n\t"
                     "v_mad_i32_i16 %[a7], %[w3], %[f], %[a7] op_sel:[1,0,0,0]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6),
                       [a7] "+v"(a7), [sv] "=&s"(saved)
                     : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [f] "v"(f), [m] "s"(mask)
                     : "scc");
    }
}

// half HI of p <- bits 30..15 of v (the 16-bit coefficient of a 32-bit tap, echo.c:246-247)
template <int HI>
__device__ __forceinline__ void put_tap16(int &p, int v, int fifteen)
{
    if (HI)
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(p) : "v"(fifteen), "v"(v));
    else
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(p) : "v"(fifteen), "v"(v));
}

__device__ __forceinline__ int half_lo(int p) { return (int) (short) (p & 0xFFFF); }
__device__ __forceinline__ int half_hi(int p) { return p >> 16; }

template <int TPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void echo_pair_kernel(const EchoLaunch L)
{
    static_assert(TPL >= 16  &&  (TPL & 3) == 0, "slices are whole packed registers, four FIR accumulators");
    constexpr int G = 2;
    constexpr int T = TPL*G;
    constexpr int NP = TPL/2;
    constexpr int kChPerWave = 32;
    constexpr int kMaxFrame = 40;                           // samples staged per pass
    constexpr int NL = 5;                                   // autocorrelation lags per lane: lag = j + 2m < 9
    __shared__ int io[4][kChPerWave][kMaxFrame + 1];        // tx | rx<<16 per sample, then the clean output (+1: read-ahead)
    __shared__ short bounce_all[4][kChPerWave][T];          // tap-set / history gathers at set events
    __shared__ float acf_all[4][kChPerWave][34];            // narrowband_detect scratch
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int g = lane >> 1;
    const int j = lane & 1;
    const int ch_raw = ((blockIdx.x*4 + wv)*kChPerWave) + g;
    const bool live = ch_raw < L.n_ch;
    const int ch = live  ?  ch_raw  :  (L.n_ch - 1);
    const bool leader = live  &&  (j == 0);
    short *bounce = &bounce_all[wv][g][0];
    float *acfbuf = &acf_all[wv][g][0];

    int32_t *sc = L.scal + (size_t) ch*kEchoScalars;
    int32_t *g32 = L.taps32 + (size_t) ch*T + j*TPL;
    int32_t *g16 = (int32_t *) (L.taps16 + (size_t) ch*4*T + j*TPL);       // + set*(T/2), packed pairs
    int32_t *gh = (int32_t *) (L.hist + (size_t) ch*T + j*TPL);

    // ---- scalars (replicated in the two lanes of a channel) ----------------------------------------
    int tx_power0 = sc[ES_TX_POWER0];
    int tx_power1 = sc[ES_TX_POWER1];
    int tx_power2 = sc[ES_TX_POWER2];
    int tx_power3 = sc[ES_TX_POWER3];
    int rx_power0 = sc[ES_RX_POWER0];
    int rx_power1 = sc[ES_RX_POWER1];
    int clean_rx_power = sc[ES_CLEAN_RX_POWER];
    int nonupdate_dwell = sc[ES_NONUPDATE_DWELL];
    int curr_pos = sc[ES_CURR_POS];
    const int mode = sc[ES_ADAPTION_MODE];
    int cng = sc[ES_CNG];
    int dtd_onset = sc[ES_DTD_ONSET];
    int tap_set = sc[ES_TAP_SET];
    int tap_rotate_counter = sc[ES_TAP_ROTATE_COUNTER];
    int narrowband_count = sc[ES_NARROWBAND_COUNT];
    int narrowband_score = sc[ES_NARROWBAND_SCORE];
    int32_t tx_hpf0 = sc[ES_TX_HPF0];
    int32_t tx_hpf1 = sc[ES_TX_HPF1];
    int32_t rx_hpf0 = sc[ES_RX_HPF0];
    int32_t rx_hpf1 = sc[ES_RX_HPF1];
    int cng_level = sc[ES_CNG_LEVEL];
    int cng_rndnum = sc[ES_CNG_RNDNUM];
    int cng_filter = sc[ES_CNG_FILTER];
    int fir_set = sc[ES_FIR_SET];
    int vad = sc[ES_VAD];
    int my_acf[NL];                                         // lane j holds last_acf[j + 2m]
#pragma unroll
    for (int m = 0;  m < NL;  m++)
        my_acf[m] = (j + m*G < 9)  ?  sc[ES_LAST_ACF + j + m*G]  :  0;

    // ---- per-lane tap slices ------------------------------------------------------------------------
    int t32[TPL];
    int tp[NP];
    int wp[NP];
#pragma unroll
    for (int k = 0;  k < TPL;  k++)
        t32[k] = g32[k];
#pragma unroll
    for (int k = 0;  k < NP;  k++)
    {
        tp[k] = g16[tap_set*(T/2) + k];
        wp[k] = gh[k];
    }
    int fifteen = 15;
    asm volatile("" : "+v"(fifteen));                       // the SDWA shifts take their count from a register

    auto load_set = [&](int set, int (&dst)[NP]) __attribute__((always_inline))
    {
#pragma unroll
        for (int k = 0;  k < NP;  k++)
            dst[k] = g16[set*(T/2) + k];
    };
    auto store_set = [&](int set, const int (&src)[NP]) __attribute__((always_inline))
    {
        if (live)
        {
#pragma unroll
            for (int k = 0;  k < NP;  k++)
                g16[set*(T/2) + k] = src[k];
        }
    };
    auto taps32_from_16 = [&]() __attribute__((always_inline))
    {
#pragma unroll
        for (int k = 0;  k < NP;  k++)
        {
            t32[2*k] = (int) ((uint32_t) half_lo(tp[k]) << 15);
            t32[2*k + 1] = (int) ((uint32_t) half_hi(tp[k]) << 15);
        }
    };
    // The history moves on by one sample (fir.h:168-183 writes history[curr_pos]): sample i of the slice becomes sample
    // i + 1, the newest place takes `incoming`; returns the sample that leaves the slice.
    auto shift_in = [&](int tx) __attribute__((always_inline)) -> int
    {
        const int out = half_hi(wp[NP - 1]);
        int in = dpp_mov<0xA0>(tx, out);                    // quad_perm [0,0,2,2]: the odd lane takes the even lane's
        in = (j == 0)  ?  tx  :  in;
#pragma unroll
        for (int k = NP - 1;  k > 0;  k--)
            wp[k] = __builtin_amdgcn_alignbit(wp[k], wp[k - 1], 16);
        wp[0] = (int) (((uint32_t) wp[0] << 16) | ((uint32_t) in & 0xFFFFu));
        return out;
    };
    auto fir_own = [&]() __attribute__((always_inline)) -> int
    {
        int ya[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0;  k < NP;  k += 4)
        {
#pragma unroll
            for (int q = 0;  q < 4;  q++)
                ya[q] = dot2_i16(tp[k + q], wp[k + q], ya[q]);
        }
        return row_sum2((ya[0] + ya[1]) + (ya[2] + ya[3]));
    };
    // lms_adapt(), echo.c:232-249, for the lanes of `mask` (a lane mask of the wave; the others keep taps and coefficients
    // exactly as they are -- a channel's coefficients need not equal its taps >> 15 between updates, e.g. after a tap set
    // was loaded).  The multiplier takes 16-bit operands; a step that does not fit (|clean_rx| >= 32768 with a quiet far
    // end: legal, rare) is applied as factor = h + h + r with h = factor >> 1 and r = 0 or 1, the same sum in wrap-around
    // arithmetic.
    auto lms_pass = [&](int f, unsigned long long mask, auto repack) __attribute__((always_inline))
    {
        constexpr bool REPACK = decltype(repack)::value;
#pragma unroll
        for (int k = 0;  k < NP;  k += 4)
            lms_piece<REPACK>(t32[2*k], t32[2*k + 1], t32[2*k + 2], t32[2*k + 3], t32[2*k + 4], t32[2*k + 5], t32[2*k + 6],
                              t32[2*k + 7], tp[k], tp[k + 1], tp[k + 2], tp[k + 3], wp[k], wp[k + 1], wp[k + 2], wp[k + 3],
                              f, fifteen, mask);
    };
    auto lms = [&](int factor, bool update) __attribute__((always_inline))
    {
        const unsigned long long mask = __ballot(update);
        if (mask == 0)
            return;
        // one copy of the update's code, run once or (wide step) three times: as an if / else of two copies the compiler
        // keeps both results of every tap apart until the merge
        const bool narrow = __all(!update  ||  factor == (int) (short) factor);
        const int h = factor >> 1;
        const int passes = narrow  ?  1  :  3;
#pragma unroll 1
        for (int p = 0;  p < passes;  p++)
        {
            const int f = narrow  ?  factor  :  (p < 2)  ?  h  :  (factor - 2*h);
            lms_pass(f, mask, std::true_type{});            // (the coefficients written by the first two of three passes are overwritten)
        }
    };
    auto lms_factor = [&](int tx, int clean_rx) __attribute__((always_inline)) -> int
    {
        // echo.c:530-553
        int factor = clean_rx;
        int sh;
        if (tx > 4*tx_power3)
            sh = top_bit_u32((uint32_t) tx) - 8;
        else
            sh = top_bit_u32((uint32_t) tx_power3) - 8;
        if (sh > 0)
            factor >>= sh;
        return factor;
    };
    // echo.c:613-651: the non-linear processor and comfort noise, then the position update and the output slot
    auto finish_sample = [&](int idx, int tx, int clean_rx) __attribute__((always_inline))
    {
        if (mode & kModeNlp)
        {
            if (rx_power1 < 30000000)
            {
                if (!cng)
                {
                    cng_level = clean_rx_power;
                    cng = 1;
                }
                if (mode & kModeCng)
                {
                    cng_rndnum = (int) (1664525U*(uint32_t) cng_rndnum + 1013904223U);
                    cng_filter = ((cng_rndnum & 0xFFFF) - 32768 + 5*cng_filter) >> 3;
                    clean_rx = (int) ((uint32_t) cng_filter*(uint32_t) cng_level) >> 17;
                }
                else
                {
                    clean_rx = 0;
                }
            }
            else
            {
                cng = 0;
            }
        }
        else
        {
            cng = 0;
        }
        // echo.c:655-658
        if (curr_pos <= 0)
            curr_pos = T;
        curr_pos--;
        if (j == 0)
            io[wv][g][idx] = ((int) (short) clean_rx & 0xFFFF) | (tx << 16);    // reuse the slot for the outputs
    };

    for (int base = 0;  base < L.samples;  base += kMaxFrame)
    {
        const int n = min(kMaxFrame, L.samples - base);
        // ---- stage tx/rx of this pass into LDS (each pair of lanes copies its own channel) -------------------
        unsigned long long st_part = 0;                     // this lane's share of the pass's received energy (L.stats)
        for (int i = j;  i < n;  i += G)
        {
            const int a = (uint16_t) L.tx[(size_t) ch*L.stride + base + i];
            const int b = (uint16_t) L.rx[(size_t) ch*L.stride + base + i];
            io[wv][g][i] = a | (b << 16);
            st_part += (unsigned long long) ((int) (short) b*(int) (short) b);
        }
        if (L.stats)
        {
            // the lanes of a channel each saw every G-th sample: add up, the first lane books the pass.  (Kept out of the
            // registers that live across the sample loop: four more of them there put spills into the common body.)
#pragma unroll
            for (int m = 1;  m < G;  m <<= 1)
                st_part += __shfl_xor(st_part, m);
            if (leader)
                L.stats[ch].sum_rx2 += st_part;
        }
        // (one wave per io / bounce / acf slice; LDS ops of a wave complete in order)
        echo_wave_sync();

        // ---- a common sample.  Returns false, with nothing changed but the history (which has taken the sample), when some
        // channel of the wave meets a set event.
        int ahead = io[wv][g][0];
        auto fast = [&](int idx) __attribute__((always_inline)) -> bool
        {
            const int word = ahead;
            ahead = io[wv][g][idx + 1];                         // the next sample's input, a whole sample early
            int tx = (int) (short) (word & 0xFFFF);
            int rx = (int) (short) (word >> 16);
            int32_t n_txh0 = tx_hpf0;
            int32_t n_txh1 = tx_hpf1;
            int32_t n_rxh0 = rx_hpf0;
            int32_t n_rxh1 = rx_hpf1;
            if (L.use_hpf_tx  &&  (mode & kModeTxHpf))
                tx = echo_hpf(n_txh0, n_txh1, tx);              // echo.c:663-669
            if (mode & kModeRxHpf)
                rx = echo_hpf(n_rxh0, n_rxh1, rx);              // echo.c:430
            (void) shift_in(tx);
            const int y = fir_own();
            const int echo_value = (int) (short) (y >> 15);
            const int clean_rx = rx - echo_value;                // echo.c:452
            const int n_dwell = nonupdate_dwell - ((nonupdate_dwell > 0)  ?  1  :  0);
            // echo.c:463-469
            const int n_tp3 = tx_power3 + ((abs(tx) - tx_power3) >> 5);
            const int n_tp2 = tx_power2 + ((tx*tx - tx_power2) >> 8);
            const int n_tp1 = tx_power1 + ((tx*tx - tx_power1) >> 5);
            const int n_tp0 = tx_power0 + ((tx*tx - tx_power0) >> 3);
            const int n_rp1 = rx_power1 + ((rx*rx - rx_power1) >> 6);
            const int n_rp0 = rx_power0 + ((rx*rx - rx_power0) >> 3);
            const int n_crp = clean_rx_power + (((int) ((uint32_t) clean_rx*(uint32_t) clean_rx) - clean_rx_power) >> 6);
            const bool loud = n_tp0 > 64*64;                     // MIN_TX_POWER_FOR_ADAPTION
            const bool single = n_tp1 > n_rp0;
            const bool adapting = loud & single & (n_dwell == 0);
            const bool doubletalk = loud & !single;
            const bool event = (adapting & ((narrowband_count >= 159) | (tap_rotate_counter <= 1)))
                               | (doubletalk & (dtd_onset == 0))
                               | ((n_rp1 > 2048*2048) & (n_crp > 4*n_rp1));
            if (__any(event))
                return false;                                    // (the history has moved on: slow() is told)
            tx_hpf0 = n_txh0;
            tx_hpf1 = n_txh1;
            rx_hpf0 = n_rxh0;
            rx_hpf1 = n_rxh1;
            nonupdate_dwell = n_dwell;
            tx_power3 = n_tp3;
            tx_power2 = n_tp2;
            tx_power1 = n_tp1;
            tx_power0 = n_tp0;
            rx_power1 = n_rp1;
            rx_power0 = n_rp0;
            clean_rx_power = n_crp;
            lms(lms_factor(tx, clean_rx), adapting  &&  (mode & kModeAdaption)  &&  narrowband_score == 0);
            narrowband_count += adapting  ?  1  :  0;
            dtd_onset = adapting  ?  0  :  dtd_onset;
            tap_rotate_counter -= adapting  ?  1  :  0;
            nonupdate_dwell = doubletalk  ?  600  :  nonupdate_dwell;      // NONUPDATE_DWELL_TIME
            finish_sample(idx, tx, clean_rx);
            return true;
        };

        // ---- any sample: the whole of echo_can_update() ---------------------------------------------------------
        auto slow = [&](int idx, bool shifted) __attribute__((always_inline))
        {
            const int word = io[wv][g][idx];
            int tx = (int) (short) (word & 0xFFFF);
            int rx = (int) (short) (word >> 16);
            if (L.use_hpf_tx  &&  (mode & kModeTxHpf))
                tx = echo_hpf(tx_hpf0, tx_hpf1, tx);            // echo.c:663-669
            if (mode & kModeRxHpf)
                rx = echo_hpf(rx_hpf0, rx_hpf1, rx);            // echo.c:430
            if (!shifted)
                (void) shift_in(tx);
            int y;
            if (__all(fir_set == tap_set))
            {
                y = fir_own();
            }
            else
            {
                // the inactive sets are current in HBM (every set event stores what it changes)
                const bool own = (fir_set == tap_set);
                y = 0;
#pragma unroll
                for (int k = 0;  k < NP;  k++)
                {
                    const int c = own  ?  tp[k]  :  g16[fir_set*(T/2) + k];
                    y = dot2_i16(c, wp[k], y);
                }
                y = row_sum2(y);
            }
            const int echo_value = (int) (short) (y >> 15);
            int clean_rx = rx - echo_value;                     // echo.c:452
            if (nonupdate_dwell > 0)
                nonupdate_dwell--;
            // echo.c:463-469
            tx_power3 += ((abs(tx) - tx_power3) >> 5);
            tx_power2 += ((tx*tx - tx_power2) >> 8);
            tx_power1 += ((tx*tx - tx_power1) >> 5);
            tx_power0 += ((tx*tx - tx_power0) >> 3);
            rx_power1 += ((rx*rx - rx_power1) >> 6);
            rx_power0 += ((rx*rx - rx_power0) >> 3);
            clean_rx_power += ((int) ((uint32_t) clean_rx*(uint32_t) clean_rx) - clean_rx_power) >> 6;

            // fir_taps16[-1] is the FIR history (echo_dev.hpp header): history[p] <- set[p]
            auto set_over_history = [&]() __attribute__((always_inline))
            {
#pragma unroll
                for (int k = 0;  k < NP;  k++)
                {
                    bounce[j*TPL + 2*k] = (short) half_lo(tp[k]);
                    bounce[j*TPL + 2*k + 1] = (short) half_hi(tp[k]);
                }
                echo_wave_sync();
#pragma unroll
                for (int k = 0;  k < NP;  k++)
                {
                    const int lo = (uint16_t) bounce[(j*TPL + 2*k + curr_pos)%T];
                    const int hi = (uint16_t) bounce[(j*TPL + 2*k + 1 + curr_pos)%T];
                    wp[k] = lo | (hi << 16);
                }
            };

            if (tx_power0 > 64*64)                              // MIN_TX_POWER_FOR_ADAPTION
            {
                if (tx_power1 > rx_power0)
                {
                    if (nonupdate_dwell == 0)
                    {
                        if (++narrowband_count >= 160)
                        {
                            narrowband_count = 0;
                            // ---- narrowband_detect(), echo.c:120-175 ---------------------------
                            // window samples 0..31 -> LDS, then every lag 0..8 has its lane (lane j: lags j, j + 2, ...)
#pragma unroll
                            for (int k = 0;  k < NP;  k++)
                            {
                                const int i = j*TPL + 2*k;
                                if (i < 32)
                                {
                                    acfbuf[i] = (curr_pos + i < T)  ?  (float) half_lo(wp[k])  :  0.0f;
                                    acfbuf[i + 1] = (curr_pos + i + 1 < T)  ?  (float) half_hi(wp[k])  :  0.0f;
                                }
                            }
                            echo_wave_sync();                       // (the samples are for both lanes of the pair)
                            float temp[NL];
#pragma unroll
                            for (int m = 0;  m < NL;  m++)
                            {
                                const int lag = j + m*G;
                                temp[m] = 0.0f;
                                if (lag < 9)
                                {
                                    for (int i = lag;  i < 32;  i++)
                                        temp[m] += acfbuf[i]*acfbuf[i - lag];
                                    if (lag == 0)
                                        acfbuf[32] = temp[m];
                                }
                            }
                            echo_wave_sync();                       // (... and so is the lag 0 sum)
                            const float scale = (float) 0x1FFFFFFF/acfbuf[32];
                            auto similar = [](int before, int now) -> bool
                            {
                                // echo.c:150-168: within a factor of two of the previous value, same sign
                                if (before >= 0  &&  now >= 0)
                                    return ((before >> 1) < now)  &&  (now < (int) ((uint32_t) before << 1));
                                if (before < 0  &&  now < 0)
                                    return ((before >> 1) > now)  &&  (now > (int) ((uint32_t) before << 1));
                                return false;
                            };
                            int score = 0;
#pragma unroll
                            for (int m = 0;  m < NL;  m++)
                            {
                                const bool mine = (j + m*G < 9);
                                const int acf = f2i_x86(temp[m]*scale);
                                const unsigned long long bal = __ballot(mine  &&  similar(my_acf[m], acf));
                                score += __popcll((bal >> (g*G)) & 3ull);
                                if (mine)
                                    my_acf[m] = acf;
                            }
                            if (score > 6)
                            {
                                if (narrowband_score == 0)
                                {
                                    // fir_taps16[3] <- fir_taps16[(tap_set + 1)%3]   (echo.c:494-496)
                                    int tmp[NP];
                                    load_set((tap_set + 1)%3, tmp);
                                    store_set(3, tmp);
                                }
                                narrowband_score += score;
                            }
                            else
                            {
                                if (narrowband_score > 200)
                                {
                                    // echo.c:504-510: revert to the set saved in [3]
                                    load_set(3, tp);
                                    const int d2 = (tap_set - 1)%3;
                                    if (d2 >= 0)
                                    {
                                        store_set(d2, tp);
                                        asm volatile("" ::: "memory");
                                    }
                                    else
                                    {
                                        set_over_history();
                                    }
                                    taps32_from_16();
                                    tap_rotate_counter = 1600;
                                }
                                narrowband_score = 0;
                            }
                        }
                        dtd_onset = 0;
                        if (--tap_rotate_counter <= 0)
                        {
                            // echo.c:518-527: rotate to the next tap set
                            tap_rotate_counter = 1600;
                            store_set(tap_set, tp);
                            tap_set++;
                            if (tap_set > 2)
                                tap_set = 0;
                            fir_set = tap_set;
                            load_set(tap_set, tp);
                        }
                        if ((mode & kModeAdaption)  &&  narrowband_score == 0)
                        {
                            lms(lms_factor(tx, clean_rx), true);
                        }
                    }
                }
                else
                {
                    if (!dtd_onset)
                    {
                        // echo.c:562-573: double talk -- fall back to the older tap set
                        const int src = (tap_set + 1)%3;
                        const int d2 = (tap_set - 1)%3;
                        load_set(src, tp);
                        if (d2 >= 0)
                        {
                            store_set(d2, tp);
                            asm volatile("" ::: "memory");
                        }
                        else
                        {
                            set_over_history();
                        }
                        taps32_from_16();
                        tap_rotate_counter = 1600;
                        dtd_onset = 1;
                    }
                    nonupdate_dwell = 600;                      // NONUPDATE_DWELL_TIME
                }
            }

            // echo.c:579-582 (vad) has no feedback into the canceller and is overwritten every
            // sample: it is evaluated once, from the final powers, at write-back.
            // echo.c:583-591
            if (rx_power1 > 2048*2048  &&  clean_rx_power > 4*rx_power1)
            {
                // The canceller is making things worse: zap every tap set
#pragma unroll
                for (int k = 0;  k < TPL;  k++)
                    t32[k] = 0;
#pragma unroll
                for (int k = 0;  k < NP;  k++)
                    tp[k] = 0;
                store_set(0, tp);
                store_set(1, tp);
                store_set(2, tp);
                store_set(3, tp);
            }
            finish_sample(idx, tx, clean_rx);
        };

        // ---- walk the pass ---------------------------------------------------------------------------------------
        for (int idx = 0;  idx < n;  idx++)
        {
            const bool tried = __all(fir_set == tap_set);
            if (tried  &&  fast(idx))
                continue;
            slow(idx, tried);
            ahead = io[wv][g][idx + 1];
        }

        // ---- clean samples out (each pair of lanes writes its own channel) ----------------------------------------
        unsigned long long cl_part = 0;
        if (live)
        {
            for (int i = j;  i < n;  i += G)
            {
                const int word = io[wv][g][i];
                L.clean[(size_t) ch*L.stride + base + i] = (int16_t) (word & 0xFFFF);
                cl_part += (unsigned long long) ((int) (short) (word & 0xFFFF)*(int) (short) (word & 0xFFFF));
                if (L.tx_out)
                    L.tx_out[(size_t) ch*L.stride + base + i] = (int16_t) (word >> 16);
            }
        }
        if (L.stats)
        {
#pragma unroll
            for (int m = 1;  m < G;  m <<= 1)
                cl_part += __shfl_xor(cl_part, m);
            if (leader)
                L.stats[ch].sum_clean2 += cl_part;
        }
    }

    // ---- write back -----------------------------------------------------------------------------------
    if (live)
    {
#pragma unroll
        for (int k = 0;  k < TPL;  k++)
            g32[k] = t32[k];
#pragma unroll
        for (int k = 0;  k < NP;  k++)
        {
            g16[tap_set*(T/2) + k] = tp[k];
            gh[k] = wp[k];
        }
#pragma unroll
        for (int m = 0;  m < NL;  m++)
        {
            if (j + m*G < 9)
                sc[ES_LAST_ACF + j + m*G] = my_acf[m];
        }
    }
    if (L.stats  &&  leader)
        L.stats[ch].samples += (uint32_t) L.samples;
    if (leader)
    {
        if (L.samples > 0)
            vad = (rx_power1)  ?  ((int) ((uint32_t) 8000*(uint32_t) clean_rx_power)/rx_power1)  :  0;
        sc[ES_TX_POWER0] = tx_power0;
        sc[ES_TX_POWER1] = tx_power1;
        sc[ES_TX_POWER2] = tx_power2;
        sc[ES_TX_POWER3] = tx_power3;
        sc[ES_RX_POWER0] = rx_power0;
        sc[ES_RX_POWER1] = rx_power1;
        sc[ES_CLEAN_RX_POWER] = clean_rx_power;
        sc[ES_NONUPDATE_DWELL] = nonupdate_dwell;
        sc[ES_CURR_POS] = curr_pos;
        sc[ES_FIR_CURR_POS] = curr_pos;
        sc[ES_CNG] = cng;
        sc[ES_DTD_ONSET] = dtd_onset;
        sc[ES_TAP_SET] = tap_set;
        sc[ES_TAP_ROTATE_COUNTER] = tap_rotate_counter;
        sc[ES_NARROWBAND_COUNT] = narrowband_count;
        sc[ES_NARROWBAND_SCORE] = narrowband_score;
        sc[ES_TX_HPF0] = tx_hpf0;
        sc[ES_TX_HPF1] = tx_hpf1;
        sc[ES_RX_HPF0] = rx_hpf0;
        sc[ES_RX_HPF1] = rx_hpf1;
        sc[ES_CNG_LEVEL] = cng_level;
        sc[ES_CNG_RNDNUM] = cng_rndnum;
        sc[ES_CNG_FILTER] = cng_filter;
        sc[ES_FIR_SET] = fir_set;
        sc[ES_VAD] = vad;
        sc[ES_LATEST_CORRECTION] = 0;
    }
}

}   // namespace spg
