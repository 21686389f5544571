// echo_dev.hpp -- device side of the batched G.168 line echo canceller
// (reference: src/echo.c:120-661, src/spandsp/fir.h:121-183).
//
// Mapping: G = SIXTEEN, EIGHT OR FOUR LANES PER CHANNEL, 64/G channels per wavefront (a 16-lane group is
// exactly one DPP row, an 8-lane group half of one, a 4-lane group a quad).  The scalar control below is
// replicated in the lanes of a group, so it costs one instruction stream per WAVE whatever G is: fewer
// lanes per channel put more channels behind every control instruction, at the price of longer tap slices
// per lane (see DESIGN.md 4.2 for the measured trade).  Lane j of a group owns taps [j*TPL, (j+1)*TPL) of its channel for
// the whole frame, in registers: the 32-bit LMS taps, the 16-bit FIR coefficients of the
// active tap set, and the matching slice of the FIR history.  The history is held in
// "window order" (w[0] = newest sample), so tap i always meets w[i]; advancing a sample
// shifts the window by one, which costs ONE row_shr:1 DPP move per lane because the
// sample loop is unrolled by TPL and the within-lane shift becomes register renaming.
// Per sample: TPL v_mad_i32_i24 (FIR) + a 4-step DPP row reduction + the scalar control
// (replicated in the 16 lanes, pure integer) + 2*TPL integer ops (LMS update).  No MFMA
// (per-channel operands, skinny integer dot products), no floating point except the
// 32x9 autocorrelation of narrowband_detect.
//
// Everything is 32-bit two's complement with wrap-around, as the reference's int
// arithmetic is on its build (products of 16-bit values through the 24-bit multiplier are
// exact in the low 32 bits).  The two accidents of the reference snapshot described in
// DESIGN.md (the fir_taps16[-1] alias onto the FIR history, and the 256-wrap of
// narrowband_detect) are part of the behaviour and are reproduced.
//
// Tap-set bookkeeping: the active set (tap_set) lives in registers and is written back at
// frame end and at the rare set events (rotation every 1600 adapted samples, double-talk
// revert, narrow-band revert, divergence zap); the other three sets stay in HBM.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace spg {

// Call f(idx + PH, integral_constant<PH>) for PH = 0..N-1 (compile-time phases).
template <int PH, int N, class F>
__device__ __forceinline__ void for_each_phase(F &f, int idx)
{
    if constexpr (PH < N)
    {
        f(idx + PH, std::integral_constant<int, PH>{});
        for_each_phase<PH + 1, N>(f, idx);
    }
}

constexpr int kEchoScalars = 48;        // int32 words per channel (layout below)

// Scalar word indices (the control fields of echo_can_state_t, src/spandsp/private/echo.h, in order)
enum
{
    ES_TX_POWER0 = 0, ES_TX_POWER1, ES_TX_POWER2, ES_TX_POWER3,
    ES_RX_POWER0, ES_RX_POWER1, ES_RX_POWER2,
    ES_CLEAN_RX_POWER, ES_RX_POWER_THRESHOLD, ES_NONUPDATE_DWELL, ES_CURR_POS, ES_TAPS, ES_TAP_MASK,
    ES_ADAPTION_MODE, ES_SUPP_TEST1, ES_SUPP_TEST2, ES_SUPP1, ES_SUPP2, ES_VAD, ES_CNG, ES_GEIGEL_MAX,
    ES_GEIGEL_LAG, ES_DTD_ONSET, ES_TAP_SET, ES_TAP_ROTATE_COUNTER, ES_LATEST_CORRECTION,
    ES_NARROWBAND_COUNT, ES_NARROWBAND_SCORE, ES_FIR_CURR_POS, ES_TX_HPF0, ES_TX_HPF1, ES_RX_HPF0,
    ES_RX_HPF1, ES_CNG_LEVEL, ES_CNG_RNDNUM, ES_CNG_FILTER, ES_FIR_SET,
    ES_LAST_ACF = 37        // 9 words
};

constexpr int kModeAdaption = 0x01;     // src/spandsp/echo.h:118-127
constexpr int kModeNlp = 0x02;
constexpr int kModeCng = 0x04;
constexpr int kModeTxHpf = 0x20;
constexpr int kModeRxHpf = 0x40;

// Per-channel line statistics (the result a multi-GPU echo run reports, see echo_stats_kernel below)
struct EchoStats
{
    unsigned long long sum_rx2;
    unsigned long long sum_clean2;
    uint32_t crc;               // running CRC-32 of the clean stream (pre- and post-conditioned as zlib's)
    uint32_t samples;           // samples in the sums
};

struct EchoLaunch
{
    const int16_t *tx;          // [n_ch][stride]
    const int16_t *rx;
    int16_t *clean;
    int16_t *tx_out;            // optional: the transmit samples after echo_can_hpf_tx() (what goes to the line)
    long long stride;
    int samples;
    int n_ch;
    int use_hpf_tx;             // apply echo_can_hpf_tx() to tx first (tests/echo_tests.c:577-594)
    int32_t *scal;              // [n_ch][kEchoScalars]
    int32_t *taps32;            // [n_ch][T]
    int16_t *taps16;            // [n_ch][4][T]
    int16_t *hist;              // [n_ch][T], window order: hist[i] = history[(i + curr_pos) mod T]
    EchoStats *stats;           // [n_ch] or nullptr: the update kernel itself adds the frame's energy sums (no second pass)
};

__device__ __forceinline__ int echo_hpf(int32_t &c0, int32_t &c1, int amp)
{
    // echo.c:382-419
    int32_t z = (int32_t) ((uint32_t) amp << 15);
    z -= (z >> 4);
    c0 += z - (c0 >> 3) - c1;
    c1 = z;
    z = c0 >> 15;
    z = max(-32768, min(32767, z));         // saturate16()
    return z;
}

__device__ __forceinline__ int top_bit_u32(uint32_t v)
{
    return (v == 0)  ?  -1  :  (31 - __builtin_clz(v));     // bit_operations.h:45-140
}

__device__ __forceinline__ int echo_pack_ncf(int narrowband_count, int dtd_onset, int narrowband_score)
{
    return (int) (((uint32_t) narrowband_count << 2) & 0x7FFFFFFCu) | (dtd_onset  ?  (int) 0x80000000u  :  0) | ((narrowband_score != 0)  ?  1  :  0);
}

// Between the lanes of ONE wave through LDS: what a lane has stored is for the others to read.  The hardware keeps a wave's LDS
// operations in order; this keeps the compiler from moving a lane's reads over stores that lane may not even execute.
__device__ __forceinline__ void echo_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// echo.c:534-543: max(top_bit(x) - 8, 0), the right shift lms_adapt()'s step gets
__device__ __forceinline__ int lms_shift(int x)
{
    int sh;
    asm("v_ffbh_u32_e32 %0, %1\n\tv_sub_u32_e64 %0, 23, %0 clamp" : "=&v"(sh) : "v"(x));
    return sh;
}

// x86-64 cvttss2si semantics of the reference build: NaN / out of range -> INT32_MIN
__device__ __forceinline__ int32_t f2i_x86(float v)
{
    if (!(v < 2147483648.0f)  ||  !(v >= -2147483648.0f))
        return (int32_t) 0x80000000u;
    return (int32_t) v;
}

// d = a*b + c through the 24 bit multiplier (operands sign-extended from bit 23, low 32 bits of the result:
// exact for the 16 bit x <= 24 bit products here).  Written as the instruction: given __mul24() on operands
// whose ranges it cannot see, hipcc emits v_bfe_i32 + the quarter-rate v_mul_lo_u32 instead.
__device__ __forceinline__ int mad24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// d = (the HIGH half of a, as a signed 16-bit number) * (the low half of b, likewise) + c: the FIR's multiply-add with the
// coefficient taken straight from the upper half of the doubled 32-bit tap (see "the taps, doubled" in echo_bank_kernel)
__device__ __forceinline__ int mad16hi(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false);
}

// Sum over the 8 lanes of half a DPP row, result in every lane: the two quad butterflies, then the mirror
// within the half row (after the butterflies a quad is uniform, so the mirror brings in the other quad's sum).
__device__ __forceinline__ int row_sum8(int v)
{
    v += dpp_mov<0xB1>(0, v);       // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(0, v);       // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(0, v);      // row_half_mirror
    return v;
}

// Sum over the 16 lanes of a DPP row, result in every lane (wrap-around add).
__device__ __forceinline__ int row_sum16(int v)
{
    v += dpp_mov<0x128>(0, v);      // row_ror:8
    v += dpp_mov<0x124>(0, v);      // row_ror:4
    v += dpp_mov<0x122>(0, v);      // row_ror:2
    v += dpp_mov<0x121>(0, v);      // row_ror:1
    return v;
}

// Bring the window registers from phase PH order (logical slot k in w[(k - PH) mod TPL]) back to phase 0 order.
template <int PH, int TPL>
__device__ __forceinline__ void echo_rotate_window(int (&w)[TPL])
{
    if constexpr ((PH%TPL) != 0)
    {
        int tmp[TPL];
#pragma unroll
        for (int k = 0;  k < TPL;  k++)
            tmp[k] = w[(k - PH + 8*TPL)%TPL];
#pragma unroll
        for (int k = 0;  k < TPL;  k++)
            w[k] = tmp[k];
    }
}

// Run the common-sample body for phases PH .. U-1 of a round of U samples.  Returns the number of samples completed; the
// window registers are left in the order of that phase (the caller rotates, through LDS: `rotate_window` in the kernel),
// which after a whole round of U == TPL samples is phase 0 order again.  A phase declines its sample when some channel of the wave meets a
// set event on it.
// `lim` (wave-uniform, 1 .. U) ends the round early: the samples a pass has left after its last whole round stay on the
// common body (a scalar compare and a branch per phase).
template <int PH, int U, int TPL, class F>
__device__ __forceinline__ int echo_fast_round(F &fast, int (&w)[TPL], int idx, int lim)
{
    if constexpr (PH == U)
    {
        return U;
    }
    else
    {
        if (PH == lim  ||  !fast(idx + PH, std::integral_constant<int, PH>{}))
            return PH;
        return echo_fast_round<PH + 1, U, TPL>(fast, w, idx, lim);
    }
}

// Sum over the 4 lanes of a quad, result in every lane.
__device__ __forceinline__ int row_sum4(int v)
{
    v += dpp_mov<0xB1>(0, v);       // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(0, v);       // quad_perm [2,3,0,1]
    return v;
}

// Sum over a pair of lanes, result in both (echo_pair.hpp).
__device__ __forceinline__ int row_sum2(int v)
{
    v += dpp_mov<0xB1>(0, v);       // quad_perm [1,0,3,2]
    return v;
}

template <int G>
__device__ __forceinline__ int group_sum(int v)
{
    return (G == 16)  ?  row_sum16(v)  :  (G == 8)  ?  row_sum8(v)  :  row_sum4(v);
}

// The history window moves one lane up inside a group; lane 0 of the group takes the new sample.
template <int G>
__device__ __forceinline__ int group_shift_in(int tx, int v, int j)
{
    if (G == 16)
        return dpp_mov<0x111>(tx, v);                   // row_shr:1, lane 0 of the row keeps `old` = tx
    if (G == 8)
    {
        v = dpp_mov<0x111>(tx, v);
        return (j == 0)  ?  tx  :  v;                   // the second channel of the row starts at lane 8
    }
    v = __builtin_amdgcn_mov_dpp(v, 0x90, 0xF, 0xF, true);     // quad_perm [0,0,1,2]: every lane has a source, no old value to set up
    return (j == 0)  ?  tx  :  v;
}

// Structure of a pass.  A sample is one of two kinds.  COMMON samples run the straight-line `fast` body: history shift,
// FIR, the power meters and, when the canceller is adapting, the LMS update; that body exists once per rotation phase of
// the history registers (a round is U unrolled samples, within which the shift of the window inside a lane is
// register renaming; a round ends with a physical rotation by U registers when U < TPL).
// A sample on which any channel of the wave meets a SET EVENT (narrow-band test every 160 adapted samples, tap set
// rotation every 1600, double-talk revert, divergence zap) -- all of which read or rewrite whole tap slices -- is
// recognised from values the fast body has computed but not yet committed; the fast round is abandoned with the
// registers brought back to phase 0 and the `slow` body, the complete per-sample algorithm, runs that one sample.
// Keeping the set events out of the unrolled bodies is what lets eight or four lanes per channel pay off: with them
// inline, a fifth of the instructions were register moves at their merge points.
// The FIR runs on the active tap set's registers; while fir_set != tap_set (after echo_can_flush(), until the next
// rotation) every sample of the wave takes the slow body, which then reads its FIR coefficients from HBM, where the
// inactive sets are always current.
// Register budget by tap slice length: 96 VGPRs (five waves per SIMD) up to 8 taps per lane, 168 (three) at 16,
// 256 (two) at 32 -- at 32 taps per lane three waves' worth of registers put spills into the common-sample body.
// A lane's slice of a 16-bit array (a tap set, the history), TPL consecutive shorts at p, as sign-extended registers and
// back -- moved 16 bytes at a time (8 for slices of four, 4 for slices of two).  As the plain loops this replaces the compiler
// made one global_load_sshort / global_store_short per element: 64 lanes x 32 two-byte accesses an array, which the memory
// side counts -- and serves -- as 32-byte partial writes (profiles/r5_hbm_calibration.json: WRITE_SIZE reads 16 x the bytes for
// that store shape, FETCH_SIZE 1.8 x for the load shape).
template <int TPL>
__device__ __forceinline__ void echo_load_shorts(const int16_t *p, int (&dst)[TPL])
{
    static_assert(TPL == 2  ||  TPL == 4  ||  (TPL%8) == 0, "slices of 2, 4 or a multiple of 8 taps");
    auto lo = [](int v) { return __builtin_amdgcn_sbfe(v, 0, 16); };
    auto hi = [](int v) { return v >> 16; };
    if constexpr (TPL == 2)
    {
        const int v = *(const int *) p;
        dst[0] = lo(v);
        dst[1] = hi(v);
    }
    else if constexpr (TPL == 4)
    {
        const int2 v = *(const int2 *) p;
        dst[0] = lo(v.x); dst[1] = hi(v.x); dst[2] = lo(v.y); dst[3] = hi(v.y);
    }
    else
    {
#pragma unroll
        for (int c = 0;  c < TPL/8;  c++)
        {
            const int4 v = ((const int4 *) p)[c];
            dst[8*c + 0] = lo(v.x); dst[8*c + 1] = hi(v.x); dst[8*c + 2] = lo(v.y); dst[8*c + 3] = hi(v.y);
            dst[8*c + 4] = lo(v.z); dst[8*c + 5] = hi(v.z); dst[8*c + 6] = lo(v.w); dst[8*c + 7] = hi(v.w);
        }
    }
}

template <int TPL>
__device__ __forceinline__ void echo_store_shorts(int16_t *p, const int (&src)[TPL])
{
    auto pk = [](int a, int b) { return (int) (((uint32_t) a & 0xFFFFu) | ((uint32_t) b << 16)); };
    if constexpr (TPL == 2)
    {
        *(int *) p = pk(src[0], src[1]);
    }
    else if constexpr (TPL == 4)
    {
        *(int2 *) p = make_int2(pk(src[0], src[1]), pk(src[2], src[3]));
    }
    else
    {
#pragma unroll
        for (int c = 0;  c < TPL/8;  c++)
            ((int4 *) p)[c] = make_int4(pk(src[8*c + 0], src[8*c + 1]), pk(src[8*c + 2], src[8*c + 3]),
                                        pk(src[8*c + 4], src[8*c + 5]), pk(src[8*c + 6], src[8*c + 7]));
    }
}

// The upper halves of TPL registers as TPL consecutive shorts at p (a tap set from the doubled taps)
template <int TPL>
__device__ __forceinline__ void echo_store_hi(int16_t *p, const int (&src)[TPL])
{
    auto pk = [](int a, int b) { return (int) (((uint32_t) a >> 16) | ((uint32_t) b & 0xFFFF0000u)); };
    if constexpr (TPL == 2)
    {
        *(int *) p = pk(src[0], src[1]);
    }
    else if constexpr (TPL == 4)
    {
        *(int2 *) p = make_int2(pk(src[0], src[1]), pk(src[2], src[3]));
    }
    else
    {
#pragma unroll
        for (int c = 0;  c < TPL/8;  c++)
            ((int4 *) p)[c] = make_int4(pk(src[8*c + 0], src[8*c + 1]), pk(src[8*c + 2], src[8*c + 3]),
                                        pk(src[8*c + 4], src[8*c + 5]), pk(src[8*c + 6], src[8*c + 7]));
    }
}

constexpr int echo_waves_per_simd(int tpl)
{
    return (tpl <= 8)  ?  5  :  (tpl <= 32)  ?  3  :  2;
}

// MODE >= 0: every channel of the bank has this adaption mode (the host knows: echo_api.hip, `uniform_mode`), so the tests
// of it are settled when the kernel is compiled and the code of the stages that are off is not there; MODE < 0: each
// channel's own mode word, tested per lane.
template <int TPL, int G, int MODE = -1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(echo_waves_per_simd(TPL), echo_waves_per_simd(TPL))))
void echo_bank_kernel(const EchoLaunch L)
{
    static_assert(G == 16  ||  G == 8  ||  G == 4, "a channel's lanes are a DPP row, half of one, or a quad");
    static_assert(((TPL*G) & (TPL*G - 1)) == 0, "positions wrap with & (T - 1)");
    static_assert(TPL*G <= 256  ||  G == 16, "histories longer than 256 samples: sixteen lanes per channel (narrowband_detect's walk)");
    constexpr int T = TPL*G;
    constexpr int kChPerWave = 64/G;
    constexpr int kMaxFrame = (G == 16)  ?  160  :  (G == 8)  ?  128  :  64;    // samples staged per pass
    // samples per unrolled round: all TPL phases (measured at TPL = 32: 636 us against 679 us with rounds of 8, although
    // the fully unrolled loop is larger than the instruction cache)
    constexpr int U = TPL;
    constexpr int NL = (9 + G - 1)/G;                       // autocorrelation lags per lane: lag = j + m*G < 9
    __shared__ int io[4][kChPerWave][kMaxFrame + 1];        // tx | rx<<16 per sample, then the clean output (+1: read-ahead)
    // Scratch of a wave, one use at a time (LDS operations of a wave complete in order):
    //   bounce   T shorts per channel          tap-set / history gathers at set events
    //   acfbuf   48 floats per channel         narrowband_detect
    //   rot      2*TPL rows of 64 shorts       the window registers of every lane, row-major (a row is one register of all 64
    //                                          lanes: no bank conflicts), written twice over so that a rotation is a row offset
    constexpr int kScratchBytes = (256*TPL > 192*kChPerWave)  ?  256*TPL  :  192*kChPerWave;
    __shared__ __attribute__((aligned(16))) char scratch[4][kScratchBytes];
    __shared__ int cold[4][kChPerWave];                     // narrowband_score: only the complete routine needs its value

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int g = lane/G;
    const int j = lane%G;
    const int ch_raw = ((blockIdx.x*4 + wv)*kChPerWave) + g;
    const bool live = ch_raw < L.n_ch;
    const int ch = live  ?  ch_raw  :  (L.n_ch - 1);
    const bool leader = live  &&  (j == 0);

    int32_t *sc = L.scal + (size_t) ch*kEchoScalars;
    int32_t *g32 = L.taps32 + (size_t) ch*T + j*TPL;
    int16_t *g16 = L.taps16 + (size_t) ch*4*T + j*TPL;     // + set*T
    int16_t *gh = L.hist + (size_t) ch*T + j*TPL;

    // ---- scalars (replicated in the lanes of the group) -----------------------------------
    // The seven power meters (echo.c:463-469, each p += (x - p) >> s) are DEALT over the lanes of a quad instead of being
    // repeated in every lane: lane q of a quad keeps tx_power[q] in meterA (x = tx*tx, or |tx| for [3]; s = 3, 5, 8, 5) and
    // rx_power[0], rx_power[1], clean_rx_power in meterB (x = rx*rx, rx*rx, clean*clean; s = 3, 6, 6; lane 3 idles): two updates
    // with a per-lane shift do the work of seven, and the decisions take the values they need from their lanes by DPP.
    const int q4 = j & 3;
    const int shiftA = (q4 == 0)  ?  3  :  (q4 == 2)  ?  8  :  5;
    const int shiftB = (q4 == 0)  ?  3  :  6;
    int meterA = sc[ES_TX_POWER0 + q4];
    int meterB = (q4 == 0)  ?  sc[ES_RX_POWER0]  :  (q4 == 1)  ?  sc[ES_RX_POWER1]  :  (q4 == 2)  ?  sc[ES_CLEAN_RX_POWER]  :  0;
    // quad_perm [k,k,k,k]: every lane has a source, so there is no old value to keep (and no move to set one up)
    auto meter_of = [&](auto lane_tag, int meter) { return __builtin_amdgcn_mov_dpp(meter, decltype(lane_tag)::value*0x55, 0xF, 0xF, true); };
    const unsigned long long is_lane3 = __builtin_amdgcn_ballot_w64(q4 == 3);      // (every lane of the wave is active here)
    using lane0 = std::integral_constant<int, 0>;
    using lane1 = std::integral_constant<int, 1>;
    using lane2 = std::integral_constant<int, 2>;
    using lane3 = std::integral_constant<int, 3>;
    int nonupdate_dwell = sc[ES_NONUPDATE_DWELL];
    // curr_pos steps T-1, T-2 .. 0, T-1 .. (echo.c:655-658): a sample's value follows from the first one's and the
    // number of samples since, and only the complete routine needs it
    const int curr_pos0 = sc[ES_CURR_POS];
    const int mode = (MODE >= 0)  ?  MODE  :  sc[ES_ADAPTION_MODE];
    int cng = sc[ES_CNG];
    int tap_set = sc[ES_TAP_SET];
    int tap_rotate_counter = sc[ES_TAP_ROTATE_COUNTER];
    // narrowband_count, dtd_onset and narrowband_score != 0 in one register, as the common body tests and steps them:
    //   bit 31 dtd_onset  |  bits 30..2 narrowband_count  |  bit 0 narrowband_score != 0
    // so that count >= 159 is an unsigned compare with 159*4 (true, too, while dtd_onset is set: that sample then takes the
    // complete routine, which is always right), dtd_onset == 0 a sign test, and count++, dtd_onset = 0 an add and an and.
    // (Three registers fewer in the common body, where there are none to spare: 168 at three waves per SIMD.)
    cold[wv][g] = sc[ES_NARROWBAND_SCORE];                  // (every lane of the channel, the same value)
    int ncf = echo_pack_ncf(sc[ES_NARROWBAND_COUNT], sc[ES_DTD_ONSET], sc[ES_NARROWBAND_SCORE]);
    // (a kernel compiled for a mode without a stage neither loads nor stores that stage's state: the registers are short)
    constexpr bool kTxHpf = (MODE < 0)  ||  (MODE & kModeTxHpf);
    constexpr bool kRxHpf = (MODE < 0)  ||  (MODE & kModeRxHpf);
    constexpr bool kNlp = (MODE < 0)  ||  (MODE & kModeNlp);
    int32_t tx_hpf0 = kTxHpf  ?  sc[ES_TX_HPF0]  :  0;
    int32_t tx_hpf1 = kTxHpf  ?  sc[ES_TX_HPF1]  :  0;
    int32_t rx_hpf0 = kRxHpf  ?  sc[ES_RX_HPF0]  :  0;
    int32_t rx_hpf1 = kRxHpf  ?  sc[ES_RX_HPF1]  :  0;
    int cng_level = kNlp  ?  sc[ES_CNG_LEVEL]  :  0;
    int cng_rndnum = kNlp  ?  sc[ES_CNG_RNDNUM]  :  0;
    int cng_filter = kNlp  ?  sc[ES_CNG_FILTER]  :  0;
    int fir_set = sc[ES_FIR_SET];
    int vad = sc[ES_VAD];
    // (last_acf[] stays in HBM: lane j reads and rewrites last_acf[j + m*G] when its channel runs the narrow-band test)

    // ---- per-lane tap slices ------------------------------------------------------------------
    // The taps, doubled (round 5).  The reference keeps fir_taps32[] and, for the FIR, fir_taps16[tap_set][] = (int16_t)
    // (fir_taps32[] >> 15), rewritten for every tap by every LMS update (echo.c:232-249): a multiply-add and a bit field
    // extract a tap and sample, and a third instruction for the FIR.  Here ts[k] holds (fir_taps32[k] << 1) mod 2^32: the
    // update is one multiply-add with the doubled step (wrap-around arithmetic is linear), and bits 15..30 of the tap -- the
    // 16-bit coefficient, with the reference's wrap -- ARE the upper half of ts[k], which v_mad_i32_i16 takes as its operand.
    // The extract is gone: 32 of the 166 vector instructions of a wave and sample.
    // What the doubling loses is bit 31 of the tap, which only the state in HBM needs.  The TRUE taps stay where they live, in
    // fir_taps32 in HBM, as of the last "rebase"; while the taps have moved by less than 2^30 since then (acc, the sum of the
    // |step|s since, times the largest sample magnitude 32768, bounds that) the true value is the one in HBM plus the 31-bit
    // difference of the low parts, sign extended -- rebase(): a load, four instructions and a store a tap, every few rounds.
    // (Kept in registers the true taps cost 32 of them, and the common body, which never looks at them, two spills a sample:
    // 0.51 ms where this is 0.4x.)  Rounds of the common body rebase between themselves when acc has passed half its budget; a
    // sample whose step would overrun it is a set event (the complete routine rebases first, and takes a step too large for one
    // piece in pieces); the write-back rebases.
    // "Stale": after a rotation of the tap sets without an update behind it (echo.c:518-527 with narrowband_score != 0) the FIR
    // runs on the rotated-in set's OLD contents, which are not the taps' upper bits.  Then ts[k] holds that set's coefficient
    // << 16 (the FIR does not know the difference), HBM the true taps (acc = 0), and bit 1 of ncf says so; the next update
    // starts from them.  A launch finds out which it is by comparing the active set with the taps.
    int ts[TPL];                // (fir_taps32 << 1) mod 2^32, or the active set's coefficients << 16 while stale
    int acc = 0;                // sum of |step| since the last rebase
    int w[TPL];                 // history window slice, physical register order (see `phase`)
    {
        int t0[TPL];
        int a16[TPL];
#pragma unroll
        for (int k = 0;  k < TPL;  k++)
            t0[k] = g32[k];
        echo_load_shorts<TPL>(g16 + tap_set*T, a16);
        bool differ = false;
#pragma unroll
        for (int k = 0;  k < TPL;  k++)
            differ |= (a16[k] != __builtin_amdgcn_sbfe(t0[k], 15, 16));
        const unsigned long long bal = __ballot(differ);
        const bool stale0 = ((bal >> ((lane/G)*G)) & ((G == 64)  ?  ~0ull  :  ((1ull << G) - 1ull))) != 0;
#pragma unroll
        for (int k = 0;  k < TPL;  k++)
            ts[k] = stale0  ?  (int) ((uint32_t) a16[k] << 16)  :  (int) ((uint32_t) t0[k] << 1);
        ncf |= stale0  ?  2  :  0;
    }
    echo_load_shorts<TPL>(gh, w);

    // the true taps in HBM from the doubled ones (lanes whose taps are stale, or that are told to sit it out, leave theirs)
    auto rebase = [&](bool skip)
    {
        if (!skip)
        {
            int t[TPL];
#pragma unroll
            for (int k = 0;  k < TPL;  k++)
                t[k] = g32[k];
#pragma unroll
            for (int k = 0;  k < TPL;  k++)
            {
                const int d = (int) (((uint32_t) ts[k] >> 1) - (uint32_t) t[k]);
                t[k] += __builtin_amdgcn_sbfe(d, 0, 31);
            }
            if (live)
            {
#pragma unroll
                for (int k = 0;  k < TPL;  k++)
                    g32[k] = t[k];
            }
            acc = 0;
        }
    };

    auto load_set = [&](int set, int (&dst)[TPL])
    {
        echo_load_shorts<TPL>(g16 + set*T, dst);
    };
    auto store_set = [&](int set, const int (&src)[TPL])
    {
        if (live)
            echo_store_shorts<TPL>(g16 + set*T, src);
    };
    // echo.c:613-651: the non-linear processor and comfort noise, then the position update and the output slot
    auto finish_sample = [&](int idx, int tx, int clean_rx, int rx_power1, int clean_rx_power)
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
        else if (MODE < 0)
        {
            cng = 0;
        }
        // (a kernel compiled for a mode without the NLP clears cng once, at write-back)
        // (echo.c:655-658, the position update: see curr_pos0)
        // (every lane of the channel stores the same word: no narrowing of exec for a leader)
        io[wv][g][idx] = ((int) (short) clean_rx & 0xFFFF) | (tx << 16);        // reuse the slot for the outputs
    };

    for (int base = 0;  base < L.samples;  base += kMaxFrame)
    {
        const int n = min(kMaxFrame, L.samples - base);
        // ---- stage tx/rx of this pass into LDS (each group copies its own channel) ----------
        unsigned long long st_part = 0;                     // this lane's share of the pass's received energy (L.stats)
        // Rows that allow it (16-byte aligned: every frame a caller's 160-sample rows lie in) move 16 bytes -- eight samples -- at
        // a time, in and out; a sample per lane and instruction, as this was, is a two-byte access a lane, which the memory side
        // serves as 32-byte partial accesses (profiles/r5_hbm_calibration.json).  Ragged ends and unaligned rows: a sample at a time.
        const int16_t *const txrow = L.tx + (size_t) ch*L.stride + base;
        const int16_t *const rxrow = L.rx + (size_t) ch*L.stride + base;
        int16_t *const cleanrow = L.clean + (size_t) ch*L.stride + base;
        int16_t *const txoutrow = L.tx_out  ?  (L.tx_out + (size_t) ch*L.stride + base)  :  nullptr;
        const bool vec = ((L.stride & 7) == 0)
                         &&  ((((uintptr_t) L.tx | (uintptr_t) L.rx | (uintptr_t) L.clean | (uintptr_t) L.tx_out) & 15) == 0);
        const int nvec = vec  ?  (n & ~7)  :  0;
        for (int c = j;  c < (nvec >> 3);  c += G)
        {
            const int4 a = *(const int4 *) (txrow + 8*c);
            const int4 b = *(const int4 *) (rxrow + 8*c);
            int *dst = &io[wv][g][8*c];
            const int av[4] = {a.x, a.y, a.z, a.w};
            const int bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int q = 0;  q < 4;  q++)
            {
                dst[2*q] = (int) (((uint32_t) av[q] & 0xFFFFu) | ((uint32_t) bv[q] << 16));
                dst[2*q + 1] = (int) (((uint32_t) av[q] >> 16) | ((uint32_t) bv[q] & 0xFFFF0000u));
                const int r0 = (int) (short) (bv[q] & 0xFFFF);
                const int r1 = bv[q] >> 16;
                st_part += (unsigned long long) (r0*r0) + (unsigned long long) (r1*r1);
            }
        }
        for (int i = nvec + j;  i < n;  i += G)
        {
            const int a = (uint16_t) txrow[i];
            const int b = (uint16_t) rxrow[i];
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
        // (one wave per io/scratch slice; LDS ops of a wave complete in order)
        echo_wave_sync();

        // ---- a common sample.  PH is the compile-time rotation phase of the history registers: logical window
        // slot k of this lane lives in w[(k - PH) mod TPL].  Returns false, with nothing changed, when some
        // channel of the wave meets a set event on this sample.
        int ahead = 0;
        auto fast = [&](int idx, auto ph_tag) -> bool
        {
            constexpr int PH = decltype(ph_tag)::value;
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
            // fir16(): history[curr_pos] = tx, i.e. the window shifts by one (fir.h:168-183).  The register that
            // held this lane's oldest slot receives the previous lane's oldest sample; lane 0 receives tx.
            constexpr int NEWP = (TPL - 1 - PH + 8*TPL)%TPL;
            const int w_old = w[NEWP];
            w[NEWP] = group_shift_in<G>(tx, w_old, j);
            // after the shift the phase is PH + 1: logical k -> w[(k - PH - 1) mod TPL]
            // four accumulators: integer addition wraps and associates, so the order is free, and no multiply-add
            // waits on the one issued just before it
            int y = 0;
            if constexpr (TPL >= 4)
            {
                int ya[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0;  k < TPL;  k += 4)
                {
#pragma unroll
                    for (int q = 0;  q < 4;  q++)
                        ya[q] = mad16hi(ts[k + q], w[(k + q - PH - 1 + 8*TPL)%TPL], ya[q]);
                }
                y = (ya[0] + ya[1]) + (ya[2] + ya[3]);
            }
            else
            {
#pragma unroll
                for (int k = 0;  k < TPL;  k++)
                    y = mad16hi(ts[k], w[(k - PH - 1 + 8*TPL)%TPL], y);
            }
            y = group_sum<G>(y);
            const int echo_value = (int) (short) (y >> 15);
            const int clean_rx = rx - echo_value;                // echo.c:452
            const int n_dwell = nonupdate_dwell - ((nonupdate_dwell > 0)  ?  1  :  0);
            // echo.c:463-469
            // (clean*clean is the low 32 bits of the product in the reference, an unsigned multiply: the 24-bit multiplier's, too)
            int xA;                                                 // (q4 == 3)  ?  abs(tx)  :  tx*tx, as a select: the compiler branches
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(xA) : "v"(tx*tx), "v"(abs(tx)), "s"(is_lane3));
            const int n_meterA = meterA + ((xA - meterA) >> shiftA);
            const int yB = (q4 == 2)  ?  clean_rx  :  rx;
            const int n_meterB = meterB + ((mad24(yB, yB, 0) - meterB) >> shiftB);
            const int n_tp0 = meter_of(lane0{}, n_meterA);
            const int n_tp1 = meter_of(lane1{}, n_meterA);
            const int n_rp0 = meter_of(lane0{}, n_meterB);
            const int n_rp1 = meter_of(lane1{}, n_meterB);
            const int n_crp = meter_of(lane2{}, n_meterB);
            // the set events (plain bit logic: with && and || hipcc builds these from exec-masked branches).  Testing the
            // ones that do not need the FIR's result before it runs was tried: the values kept alive across the FIR cost
            // more in spills than the abandoned work saves.
            const bool loud = n_tp0 > 64*64;                     // MIN_TX_POWER_FOR_ADAPTION
            const bool single = n_tp1 > n_rp0;
            const bool adapting = loud & single & (n_dwell == 0);
            const bool doubletalk = loud & !single;
            // does any channel of the wave meet one?  As arithmetic on the compares' lane masks (a vote on the combined
            // condition makes the compiler turn the mask into a register and compare that again)
            typedef unsigned long long mask_t;
            const mask_t m_loud = __builtin_amdgcn_ballot_w64(loud);
            const mask_t m_single = __builtin_amdgcn_ballot_w64(single);
            const mask_t m_adapting = m_loud & m_single & __builtin_amdgcn_ballot_w64(n_dwell == 0);
            // the update's step (echo.c:530-553; used only by the lanes that update): the shift is max(top_bit(x) - 8, 0) with
            // top_bit(0) = -1: 23 - ffbh(x), an unsigned subtraction that clamps at zero (ffbh(0) is all ones).  Formed ahead of
            // the event test, because two things about it are set events of this kernel's own: a step that would take the
            // taps further from their last rebase than the doubled representation can account for, and an update of taps
            // that are stale (see "the taps, doubled")
            const int n_tp3 = meter_of(lane3{}, n_meterA);
            const int factor = clean_rx >> lms_shift((tx > 4*n_tp3)  ?  tx  :  n_tp3);
            const int n_acc = acc + abs(factor);
            const mask_t m_lms = ((mode & kModeAdaption) != 0)  ?  (m_adapting & __builtin_amdgcn_ballot_w64((ncf & 1) == 0))  :  0;
            const mask_t m_event = (m_adapting & (__builtin_amdgcn_ballot_w64((uint32_t) ncf >= 159u*4u) | __builtin_amdgcn_ballot_w64(tap_rotate_counter <= 1)))
                                   | (m_loud & ~m_single & __builtin_amdgcn_ballot_w64(ncf >= 0))
                                   | (__builtin_amdgcn_ballot_w64(n_rp1 > 2048*2048) & __builtin_amdgcn_ballot_w64(n_crp > 4*n_rp1))
                                   | (m_lms & (__builtin_amdgcn_ballot_w64(n_acc > 32767) | __builtin_amdgcn_ballot_w64((ncf & 2) != 0)));
            if (__builtin_expect(m_event != 0, 0))
            {
                w[NEWP] = w_old;
                return false;
            }
            tx_hpf0 = n_txh0;
            tx_hpf1 = n_txh1;
            rx_hpf0 = n_rxh0;
            rx_hpf1 = n_rxh1;
            nonupdate_dwell = n_dwell;
            meterA = n_meterA;
            meterB = n_meterB;
            // (one conditional region, the update's: the two counters step by selects)
            ncf = adapting  ?  ((ncf + 4) & 0x7FFFFFFF)  :  ncf;            // narrowband_count++, dtd_onset = 0
            tap_rotate_counter -= adapting  ?  1  :  0;
            if (adapting  &  ((mode & kModeAdaption) != 0)  &  ((ncf & 1) == 0))     // ... and narrowband_score == 0
            {
                // lms_adapt(), echo.c:232-249, on the doubled taps: one multiply-add a tap (the coefficient of the FIR is the
                // upper half of the result)
                const int factor2 = factor << 1;
#pragma unroll
                for (int k = 0;  k < TPL;  k++)
                    ts[k] = mad24(w[(k - PH - 1 + 8*TPL)%TPL], factor2, ts[k]);
                acc = n_acc;
            }
            nonupdate_dwell = doubletalk  ?  600  :  nonupdate_dwell;      // NONUPDATE_DWELL_TIME
            finish_sample(idx, tx, clean_rx, n_rp1, n_crp);
            return true;
        };

        // ---- any sample: the whole of echo_can_update().  Takes the window registers in the order of phase TPL - 1 (logical
        // slot k in w[(k + 1) mod TPL]) and leaves them, one sample later, in phase 0 order. ----------------------------
        auto slow = [&](int idx)
        {
            int tx_power0 = meter_of(lane0{}, meterA);
            int tx_power1 = meter_of(lane1{}, meterA);
            int tx_power2 = meter_of(lane2{}, meterA);
            int tx_power3 = meter_of(lane3{}, meterA);
            int rx_power0 = meter_of(lane0{}, meterB);
            int rx_power1 = meter_of(lane1{}, meterB);
            int clean_rx_power = meter_of(lane2{}, meterB);
            int narrowband_count = (int) (((uint32_t) ncf >> 2) & 0x1FFFFFFFu);
            int dtd_onset = (ncf < 0)  ?  1  :  0;
            int narrowband_score = cold[wv][g];
            bool stale = (ncf & 2) != 0;                        // ts[] holds the active set's coefficients << 16, HBM the taps
            int curr_pos = (curr_pos0 - (base + idx)) & (T - 1);
            // (opaque: or every expression in curr_pos0 below is computed ahead of the sample loop, and then kept in scratch)
            asm volatile("" : "+v"(curr_pos));
            const int word = io[wv][g][idx];
            int tx = (int) (short) (word & 0xFFFF);
            int rx = (int) (short) (word >> 16);
            if (L.use_hpf_tx  &&  (mode & kModeTxHpf))
                tx = echo_hpf(tx_hpf0, tx_hpf1, tx);            // echo.c:663-669
            if (mode & kModeRxHpf)
                rx = echo_hpf(rx_hpf0, rx_hpf1, rx);            // echo.c:430
            w[0] = group_shift_in<G>(tx, w[0], j);
            // now logical k -> w[k]
            int y = 0;
            if (__all(fir_set == tap_set))
            {
#pragma unroll
                for (int k = 0;  k < TPL;  k++)
                    y = mad16hi(ts[k], w[k], y);
            }
            else
            {
                // the inactive sets are current in HBM (every set event stores what it changes)
                const bool own = (fir_set == tap_set);
#pragma unroll
                for (int k = 0;  k < TPL;  k++)
                {
                    const int c = own  ?  (ts[k] >> 16)  :  (int) g16[fir_set*T + k];
                    y = mad24(c, w[k], y);
                }
            }
            y = group_sum<G>(y);
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

            // fir_taps16[-1] is the FIR history (see the header): history[p] <- set[p]
            auto set_over_history = [&](const int (&c16)[TPL])
            {
                short *const bounce = (short *) scratch[wv] + g*T;
#pragma unroll
                for (int k = 0;  k < TPL;  k++)
                    bounce[j*TPL + k] = (short) c16[k];
                echo_wave_sync();
                const unsigned first = (unsigned) (j*TPL + curr_pos);
#pragma unroll
                for (int k = 0;  k < TPL;  k++)
                    w[k] = bounce[(first + k) & (T - 1)];
            };

            // echo.c:509-510,570-571: the taps become a coefficient set (fir_taps32[i] = fir_taps16[..][i] << 15) -- true and
            // doubled at once, nothing stale, nothing to account for
            auto taps_from_set = [&](const int (&c16)[TPL])
            {
#pragma unroll
                for (int k = 0;  k < TPL;  k++)
                    ts[k] = (int) ((uint32_t) c16[k] << 16);
                if (live)
                {
#pragma unroll
                    for (int k = 0;  k < TPL;  k++)
                        g32[k] = (int) ((uint32_t) c16[k] << 15);
                }
                acc = 0;
                stale = false;
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
                            // The window's first 32 samples go to LDS behind eight zeros: [0..7] zeros, [8..39] samples, [40] the
                            // lag 0 sum.  Every lag 0..8 has its lane (lane j: lags j, j + G, ...), and a lag's sum over
                            // i = lag .. 31 of x[i]*x[i - lag] runs over ALL i with x[i - lag] read from the zeros for i < lag:
                            // a sum that starts at +0 stays what it is when +-0 is added, so the order and the roundings are
                            // the reference's, and the operands can be fetched eight at a time with one wait instead of one wait per
                            // term (the loops this replaces were a rolled ds_read / wait / multiply / add per term: some ten
                            // thousand cycles of LDS latency per test, sixteen tests per wave and frame on lines out of step).
                            float *const acfbuf = (float *) scratch[wv] + g*48;
                            // last_acf[] of this lane's lags: asked for now, needed after the sums
                            int before[NL];
#pragma unroll
                            for (int m = 0;  m < NL;  m++)
                                before[m] = (j + m*G < 9)  ?  sc[ES_LAST_ACF + j + m*G]  :  0;
                            // (every k against a constant: written as i = j*TPL + k against T - curr_pos, the compiler computes the
                            // TPL values T - i ahead of the sample loop and keeps them in scratch)
                            const int jt = j*TPL;
                            const int lead = curr_pos + jt;
                            if (j < 2)
                            {
#pragma unroll
                                for (int k = 0;  k < 4;  k++)
                                    acfbuf[4*j + k] = 0.0f;
                            }
                            if constexpr (T <= 256)
                            {
#pragma unroll
                                for (int k = 0;  k < TPL;  k++)
                                {
                                    if (jt < 32 - k)
                                    {
                                        const bool inside = (T == 256)  ||  (lead < T - k);
                                        (acfbuf + 8 + jt)[k] = inside  ?  (float) w[k]  :  0.0f;
                                    }
                                }
                            }
                            else
                            {
                                // A history longer than the 256 the reference's walk wraps at (echo.c:133-139: k = curr_pos, then k++ and
                                // back to 0 at 256, whatever the length): sample 0 is history[curr_pos], wherever that lies, and the rest
                                // run on from there only while below 256 -- from curr_pos >= 256 they are history[0], [1], ...  The
                                // window goes to LDS in its own order (entry m = history[(curr_pos + m) mod T]) and each of the channel's
                                // lanes fetches sample j and sample j + 16 from where the walk finds them.
                                (void) lead;
                                short *const bounce = (short *) scratch[wv] + g*T;
#pragma unroll
                                for (int k = 0;  k < TPL;  k++)
                                    bounce[jt + k] = (short) w[k];
                                echo_wave_sync();
                                float mine[2];
#pragma unroll
                                for (int m = 0;  m < 2;  m++)
                                {
                                    const int i = j + m*G;
                                    const int q = (i == 0)  ?  curr_pos  :  (curr_pos >= 256)  ?  (i - 1)  :  ((curr_pos + i) & 255);
                                    mine[m] = (float) bounce[(q - curr_pos) & (T - 1)];
                                }
                                echo_wave_sync();
                                // (the zeros in front, again: the window stood where they are)
                                if (j < 2)
                                {
#pragma unroll
                                    for (int k = 0;  k < 4;  k++)
                                        acfbuf[4*j + k] = 0.0f;
                                }
                                acfbuf[8 + j] = mine[0];
                                acfbuf[8 + j + G] = mine[1];
                            }
                            echo_wave_sync();
                            float temp[NL];
                            const float *shifted[NL];
#pragma unroll
                            for (int m = 0;  m < NL;  m++)
                            {
                                temp[m] = 0.0f;
                                shifted[m] = acfbuf + 8 - min(j + m*G, 8);      // (a lag past 8 is nobody's: its sum is not used)
                            }
                            // (a rolled loop of four: unrolled, the scheduler asks for all 128 operands at once and spills a thousand registers)
#pragma nounroll
                            for (int c = 0;  c < 32;  c += 8)
                            {
                                float xv[8];
#pragma unroll
                                for (int i = 0;  i < 8;  i++)
                                    xv[i] = acfbuf[8 + c + i];
#pragma unroll
                                for (int m = 0;  m < NL;  m++)
                                {
                                    float yv[8];
#pragma unroll
                                    for (int i = 0;  i < 8;  i++)
                                        yv[i] = shifted[m][c + i];
#pragma unroll
                                    for (int i = 0;  i < 8;  i++)
                                        temp[m] += xv[i]*yv[i];
                                }
                            }
                            if (j == 0)
                                acfbuf[40] = temp[0];
                            echo_wave_sync();
                            const float scale = (float) 0x1FFFFFFF/acfbuf[40];
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
                                const unsigned long long bal = __ballot(mine  &&  similar(before[m], acf));
                                score += __popcll((bal >> (g*G)) & ((1ull << G) - 1ull));
                                if (mine  &&  live)
                                    sc[ES_LAST_ACF + j + m*G] = acf;
                            }
                            if (score > 6)
                            {
                                if (narrowband_score == 0)
                                {
                                    // fir_taps16[3] <- fir_taps16[(tap_set + 1)%3]   (echo.c:494-496)
                                    int tmp[TPL];
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
                                    int c16[TPL];
                                    load_set(3, c16);
                                    const int d2 = (tap_set - 1)%3;
                                    if (d2 >= 0)
                                        store_set(d2, c16);
                                    else
                                        set_over_history(c16);
                                    taps_from_set(c16);
                                    tap_rotate_counter = 1600;
                                }
                                narrowband_score = 0;
                            }
                        }
                        dtd_onset = 0;
                        if (--tap_rotate_counter <= 0)
                        {
                            // echo.c:518-527: rotate to the next tap set.  The set that was active goes to HBM; the FIR carries on
                            // with the OLD contents of the next one until an update rewrites them: stale (the update below,
                            // if it runs, ends that at once)
                            tap_rotate_counter = 1600;
                            {
                                int c16[TPL];
#pragma unroll
                                for (int k = 0;  k < TPL;  k++)
                                    c16[k] = ts[k] >> 16;
                                store_set(tap_set, c16);
                            }
                            tap_set++;
                            if (tap_set > 2)
                                tap_set = 0;
                            fir_set = tap_set;
                            rebase(stale);                      // the true taps, before ts[] changes its meaning
                            {
                                int c16[TPL];
                                load_set(tap_set, c16);
#pragma unroll
                                for (int k = 0;  k < TPL;  k++)
                                    ts[k] = (int) ((uint32_t) c16[k] << 16);
                            }
                            acc = 0;
                            stale = true;
                        }
                        if ((mode & kModeAdaption)  &&  narrowband_score == 0)
                        {
                            // echo.c:530-553 + lms_adapt(), echo.c:232-249
                            int factor = clean_rx;
                            int sh;
                            if (tx > 4*tx_power3)
                                sh = top_bit_u32((uint32_t) tx) - 8;
                            else
                                sh = top_bit_u32((uint32_t) tx_power3) - 8;
                            if (sh > 0)
                                factor >>= sh;
                            // the update on the doubled taps (see "the taps, doubled"): from the true taps if they were stale;
                            // rebased first if this step could take them 2^30 from the last rebase; a step too large for
                            // that on its own (|clean_rx| up to 65 535 unshifted) in pieces of 16 384 with a rebase after each
                            // (wrap-around arithmetic is linear: the pieces add up to the step)
                            if (stale)
                            {
#pragma unroll
                                for (int k = 0;  k < TPL;  k++)
                                    ts[k] = (int) ((uint32_t) g32[k] << 1);
                                acc = 0;
                                stale = false;
                            }
                            if (acc + abs(factor) > 32767)
                                rebase(false);
                            int rem = factor;
                            while (abs(rem) > 16384)
                            {
                                const int piece = (rem > 0)  ?  16384  :  -16384;
#pragma unroll
                                for (int k = 0;  k < TPL;  k++)
                                    ts[k] = mad24(w[k], piece << 1, ts[k]);
                                rebase(false);
                                rem -= piece;
                            }
#pragma unroll
                            for (int k = 0;  k < TPL;  k++)
                                ts[k] = mad24(w[k], rem << 1, ts[k]);
                            acc += abs(rem);
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
                        int c16[TPL];
                        load_set(src, c16);
                        if (d2 >= 0)
                            store_set(d2, c16);
                        else
                            set_over_history(c16);
                        taps_from_set(c16);
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
                int zero[TPL];
#pragma unroll
                for (int k = 0;  k < TPL;  k++)
                    zero[k] = 0;
                taps_from_set(zero);
                store_set(0, zero);
                store_set(1, zero);
                store_set(2, zero);
                store_set(3, zero);
            }
            finish_sample(idx, tx, clean_rx, rx_power1, clean_rx_power);
            meterA = (q4 == 0)  ?  tx_power0  :  (q4 == 1)  ?  tx_power1  :  (q4 == 2)  ?  tx_power2  :  tx_power3;
            meterB = (q4 == 0)  ?  rx_power0  :  (q4 == 1)  ?  rx_power1  :  (q4 == 2)  ?  clean_rx_power  :  0;
            cold[wv][g] = narrowband_score;                     // (the lanes of a channel agree)
            ncf = echo_pack_ncf(narrowband_count, dtd_onset, narrowband_score) | (stale  ?  2  :  0);
        };

        // ---- the window registers rotated by r (wave-uniform, 0 .. TPL-1) places: new w[k] = old w[(k - r) mod TPL], which
        // brings phase r order back to phase 0 order.  Through LDS: every register is written to rows i and i + TPL of the
        // wave's scratch, and row k + (TPL - r) is what register k takes.  (As register moves a rotation by a run-time count
        // is either a chain of rotations by one -- what the compiler made of thirty-two constant rotations at the exits of
        // a round, some 540 moves for an average set event, a fifth of the kernel's instructions on mixed lines -- or
        // log2(TPL) conditional rotations by powers of two, whose temporaries put spills into `slow`: 677 us against 560.)
        auto rotate_window = [&](int r)
        {
            if (r == 0)
                return;
            short *const rot = (short *) scratch[wv] + lane;
#pragma unroll
            for (int k = 0;  k < TPL;  k++)
            {
                rot[64*k] = (short) w[k];
                rot[64*(k + TPL)] = (short) w[k];
            }
            const short *from = rot + 64*(TPL - r);
#pragma unroll
            for (int k = 0;  k < TPL;  k++)
                w[k] = from[64*k];
        };

        // ---- walk the pass ---------------------------------------------------------------------------------------
        int idx = 0;
        while (idx < n)
        {
            int phase = 0;
            if (__all(fir_set == tap_set))
            {
                // (half the budget used: the true taps are brought up to date here, between rounds, so that no sample of the
                // common body needs to -- one that would overrun the budget after all is a set event)
                if (__any(acc > 16383))
                    rebase((ncf & 2) != 0);
                const int lim = __builtin_amdgcn_readfirstlane(min(U, n - idx));
                ahead = io[wv][g][idx];
                const int done = __builtin_amdgcn_readfirstlane(echo_fast_round<0, U, TPL>(fast, w, idx, lim));
                idx += done;
                if (done == lim)
                {
                    rotate_window(done%TPL);
                    continue;
                }
                phase = done;
            }
            // a set event on this sample (or the FIR on another tap set than the one that adapts)
            rotate_window((phase + 1)%TPL);
            slow(idx);
            idx++;
        }

        // ---- clean samples out (each group writes its own channel) ---------------------------
        unsigned long long cl_part = 0;
        if (live)
        {
            for (int c = j;  c < (nvec >> 3);  c += G)
            {
                const int *src = &io[wv][g][8*c];
                int lo[4];
                int hi[4];
#pragma unroll
                for (int q = 0;  q < 4;  q++)
                {
                    const int w0 = src[2*q];
                    const int w1 = src[2*q + 1];
                    lo[q] = (int) (((uint32_t) w0 & 0xFFFFu) | ((uint32_t) w1 << 16));
                    hi[q] = (int) (((uint32_t) w0 >> 16) | ((uint32_t) w1 & 0xFFFF0000u));
                    const int c0 = (int) (short) (w0 & 0xFFFF);
                    const int c1 = (int) (short) (w1 & 0xFFFF);
                    cl_part += (unsigned long long) (c0*c0) + (unsigned long long) (c1*c1);
                }
                *(int4 *) (cleanrow + 8*c) = make_int4(lo[0], lo[1], lo[2], lo[3]);
                if (txoutrow)
                    *(int4 *) (txoutrow + 8*c) = make_int4(hi[0], hi[1], hi[2], hi[3]);
            }
            for (int i = nvec + j;  i < n;  i += G)
            {
                const int word = io[wv][g][i];
                cleanrow[i] = (int16_t) (word & 0xFFFF);
                cl_part += (unsigned long long) ((int) (short) (word & 0xFFFF)*(int) (short) (word & 0xFFFF));
                if (txoutrow)
                    txoutrow[i] = (int16_t) (word >> 16);
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

    // ---- write back -----------------------------------------------------------------------------
    if (live)
    {
        rebase((ncf & 2) != 0);                             // the true taps, in place
        echo_store_hi<TPL>(g16 + tap_set*T, ts);            // (stale or not: the upper halves are the active set's coefficients)
        echo_store_shorts<TPL>(gh, w);
    }
    if (L.stats  &&  leader)
        L.stats[ch].samples += (uint32_t) L.samples;
    const int tx_power0 = meter_of(lane0{}, meterA);
    const int tx_power1 = meter_of(lane1{}, meterA);
    const int tx_power2 = meter_of(lane2{}, meterA);
    const int tx_power3 = meter_of(lane3{}, meterA);
    const int rx_power0 = meter_of(lane0{}, meterB);
    const int rx_power1 = meter_of(lane1{}, meterB);
    const int clean_rx_power = meter_of(lane2{}, meterB);
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
        const int curr_pos = (curr_pos0 - L.samples) & (T - 1);
        sc[ES_CURR_POS] = curr_pos;
        sc[ES_FIR_CURR_POS] = curr_pos;
        sc[ES_CNG] = (MODE >= 0  &&  !(MODE & kModeNlp)  &&  L.samples > 0)  ?  0  :  cng;
        sc[ES_DTD_ONSET] = (ncf < 0)  ?  1  :  0;
        sc[ES_TAP_SET] = tap_set;
        sc[ES_TAP_ROTATE_COUNTER] = tap_rotate_counter;
        sc[ES_NARROWBAND_COUNT] = (int) (((uint32_t) ncf >> 2) & 0x1FFFFFFFu);
        sc[ES_NARROWBAND_SCORE] = cold[wv][g];
        if (kTxHpf)
        {
            sc[ES_TX_HPF0] = tx_hpf0;
            sc[ES_TX_HPF1] = tx_hpf1;
        }
        if (kRxHpf)
        {
            sc[ES_RX_HPF0] = rx_hpf0;
            sc[ES_RX_HPF1] = rx_hpf1;
        }
        if (kNlp)
        {
            sc[ES_CNG_LEVEL] = cng_level;
            sc[ES_CNG_RNDNUM] = cng_rndnum;
            sc[ES_CNG_FILTER] = cng_filter;
        }
        sc[ES_FIR_SET] = fir_set;
        sc[ES_VAD] = vad;
        sc[ES_LATEST_CORRECTION] = 0;
    }
}

// echo_can_hpf_tx() on its own (src/echo.c:663-669): one thread per channel filters `samples` transmit samples in place
// of the fused path of echo_bank_kernel (callers that need the filtered sample before they have the matching rx sample).
__global__ __launch_bounds__(64)
void echo_hpf_tx_kernel(const int16_t *tx, int16_t *out, long long stride, int samples, int n_ch, int32_t *scal)
{
    const int ch = blockIdx.x*64 + threadIdx.x;
    if (ch >= n_ch)
        return;
    int32_t *sc = scal + (size_t) ch*kEchoScalars;
    const int mode = sc[ES_ADAPTION_MODE];
    int32_t c0 = sc[ES_TX_HPF0];
    int32_t c1 = sc[ES_TX_HPF1];
    for (int i = 0;  i < samples;  i++)
    {
        int v = tx[(size_t) ch*stride + i];
        if (mode & kModeTxHpf)
            v = echo_hpf(c0, c1, v);
        out[(size_t) ch*stride + i] = (int16_t) v;
    }
    sc[ES_TX_HPF0] = c0;
    sc[ES_TX_HPF1] = c1;
}

// ---- per-channel line statistics (the result a multi-GPU echo run reports) -----------------------------
// After an update, one lane per channel walks that channel's received (rx) and cleaned samples of the frame and
// accumulates sum rx^2 and sum clean^2 (exact, 64 bit) -- ERLE = 10 log10(sum rx^2 / sum clean^2) as the reference's
// level_measurements do (tests/echo_tests.c:577-594) -- and carries on the CRC-32 (zlib polynomial, the bytes of the
// int16 little-endian stream) of everything the canceller has put out, for bit-exactness checks against the CPU path.
// The frame was written a moment ago by echo_bank_kernel and is read from L2; the CRC table sits in LDS.
__global__ __launch_bounds__(256)
void echo_stats_kernel(const int16_t *rx, const int16_t *clean, long long stride, int samples, int n_ch, EchoStats *st)
{
    __shared__ uint32_t table[256];
    {
        uint32_t c = threadIdx.x;
        for (int k = 0;  k < 8;  k++)
            c = (c & 1u)  ?  (0xEDB88320u ^ (c >> 1))  :  (c >> 1);
        table[threadIdx.x] = c;
    }
    __syncthreads();
    const int ch = blockIdx.x*256 + threadIdx.x;
    if (ch >= n_ch)
        return;
    EchoStats s = st[ch];
    uint32_t crc = ~s.crc;
    const int16_t *r = rx + (size_t) ch*stride;
    const int16_t *c = clean + (size_t) ch*stride;
    auto one = [&](int a, int b)
    {
        s.sum_rx2 += (unsigned long long) (a*a);
        s.sum_clean2 += (unsigned long long) (b*b);
        crc = table[(crc ^ (uint32_t) b) & 0xFF] ^ (crc >> 8);
        crc = table[(crc ^ ((uint32_t) b >> 8)) & 0xFF] ^ (crc >> 8);
    };
    int i = 0;
    if ((((uintptr_t) r | (uintptr_t) c) & 15) == 0)
    {
        // rows on 16-byte boundaries: eight samples per load
        for (  ;  i + 8 <= samples;  i += 8)
        {
            const int4 ra = *(const int4 *) (r + i);
            const int4 ca = *(const int4 *) (c + i);
            const int rw[4] = {ra.x, ra.y, ra.z, ra.w};
            const int cw[4] = {ca.x, ca.y, ca.z, ca.w};
#pragma unroll
            for (int k = 0;  k < 4;  k++)
            {
                one((int) (short) (rw[k] & 0xFFFF), (int) (short) (cw[k] & 0xFFFF));
                one(rw[k] >> 16, cw[k] >> 16);
            }
        }
    }
    for (  ;  i < samples;  i++)
        one(r[i], c[i]);
    s.crc = ~crc;
    s.samples += (uint32_t) samples;
    st[ch] = s;
}

// ERLE in dB per channel from the sums (0 dB while nothing has been measured; capped at 120 dB when the residue is zero).
__global__ __launch_bounds__(256)
void echo_erle_kernel(const EchoStats *st, float *erle_db, int n_ch)
{
    const int ch = blockIdx.x*256 + threadIdx.x;
    if (ch >= n_ch)
        return;
    const EchoStats s = st[ch];
    float v = 0.0f;
    if (s.sum_rx2 != 0)
        v = (s.sum_clean2 == 0)  ?  120.0f  :  10.0f*log10f((float) ((double) s.sum_rx2/(double) s.sum_clean2));
    erle_db[ch] = v;
}

}   // namespace spg
