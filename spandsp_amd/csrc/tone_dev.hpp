// tone_dev.hpp -- device-side Goertzel bank: per-lane recurrences, block-end
// decisions and the LDS-staged frame walker shared by the DTMF / Bell MF / R2 MF /
// super-tone / generic Goertzel bank kernels (gfx950, wave64).
//
// Mapping.  LPC = lanes per channel.
//   LPC = 1: one channel per lane, a wavefront owns 64 consecutive channels and all
//            NB bins of each (the throughput mapping: least VALU work per sample).
//   LPC = 2: a wavefront owns 32 channels; lanes 0-31 run the first half of the bins
//            (DTMF: the four row tones), lanes 32-63 the second half (the column
//            tones).  Twice the wavefronts for the same bank, so a 65 536-channel bank
//            puts 2 waves on every SIMD instead of 1.  On CDNA4 a lone wave issues at
//            most one instruction per ~4 cycles whatever its type (measured: 50 % VALU
//            activity, every SALU/LDS instruction serialised behind the VALU stream);
//            a second resident wave overlaps them.  The halves exchange their block
//            energies with one cross-lane swap per bin at block end.
// The 2*bins recurrence registers stay in VGPRs for the whole frame.  PCM arrives
// channel-major (amp[ch][samples], what spandsp callers hold); it is staged HBM -> LDS
// by LDS-DMA in 80-sample segments, double buffered, and each lane walks its own row
// (see "Frame staging").  No MFMA: the work is independent 3-op recurrences, issued
// as v_pk_mul_f32 / v_pk_add_f32 pairs.
//
// Numerics: every multiply and add is rounded separately, in the reference's order
// (this translation unit is compiled with -ffp-contract=off; packed ops round each half
// independently), so all float state and all decisions are bit-identical to the
// reference's strict-IEEE build:
//   recurrence   v3' = (fac*v2 - v1) + x                 src/spandsp/tone_detect.h:172-192
//   block result 2*((v3*v3 + v2*v2) - (v2*v3)*fac)      src/tone_detect.c:160-205
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "cadence_dev.hpp"

namespace spg {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kMaxBins = 64;            // banks of more than 16 bins run two lanes per channel, of more than 32 four (16 bins per lane)

// Kernel argument block (passed by value; lives in SGPRs / kernarg segment).
struct ToneLaunch
{
    const int16_t *amp;
    long long stride;           // samples between channels (channel-major) or between samples (sample-major)
    int samples;                // samples per channel in this call
    int n_ch;
    int layout;                 // 0 channel-major, 1 sample-major
    int aligned16;              // channel-major rows are 16-byte aligned and padded to 8 samples (LDS-DMA loader)
    float *sf;                  // float state  [NSF][n_ch]
    int32_t *si;                // int state    [2][n_ch]
    uint32_t *rec;              // block records [maxb][n_ch]
    float *rec_energy;          // [maxb][n_ch] or nullptr
    int32_t *rec_dur;           // [maxb][n_ch] or nullptr
    float *trace;               // [maxb][NB+1][n_ch] or nullptr
    int maxb;
    int nbins;                  // run-time bin count (<= NB) for super-tone / generic banks
    int block_len;              // run-time block length for the generic bank
    int realtime;               // DTMF: realtime report mode (duration is zeroed on a report)
    int force_end;              // evaluate the block now, whatever its fill (goertzel_result() mid-block)
    float fac[kMaxBins];
    float threshold;
    float normal_twist;
    float reverse_twist;
    int fmt;                    // 0: int16 linear PCM, 1: G.711 A-law bytes, 2: G.711 u-law bytes (channel-major, LPC = 2 kernels)
    long long *probe_ts;        // tools/probe.hip only: per-wave timestamps (kernels built with ABL & 32)
    const int32_t *lens;        // nullptr: every channel has `samples`; else samples per channel in this call (0 = the
                                // channel sits this call out: its state and block phase are not touched)
    uint8_t *digits;            // nullptr, or [maxb][n_ch]: per block the digit it delivered (0 = none) -- the one-byte
                                // report a multi-GPU run gathers instead of the record words
    int functor;                // generic bank: 0 none, 1 v18.c's raw block decision, 2 ademco_contactid.c's (include/spangpu.h)
    float functor_threshold;
    int lens_ragged;            // host side only: lengths other than 0 and `samples` occur (the general kernel takes the call)
    const float *chan_parms;    // nullptr: threshold / twists / dial tone filter as set for the bank; else DTMF
                                // per channel, [4][n_ch]: threshold, normal twist, reverse twist, filter on (0 / 1)
    CadenceArgs cad;            // super-tone: cad.state != nullptr has the streaming kernel built with kToneCadence match
                                // the cadences in its epilogue (cadence_dev.hpp)
    int wg0;                    // streaming kernels: the launch covers workgroups wg0 .. wg0 + wgn - 1 of the bank (wgn = 0: all of
    int wgn;                    // them) -- a bank in queue mode is advanced by two launches on two streams (spangpu_api.hip)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------
// Per-lane filter bank state: NBL bins held as NBL/2 packed pairs.  One step is
// written in three phases (all products, all subtractions, all additions) separated
// by scheduling barriers: left alone, hipcc's scheduler serialises one bin pair across
// several samples, so every packed op waits on its predecessor.
// ---------------------------------------------------------------------------------
template <int NBL>
struct Bank
{
    static constexpr int NP = NBL/2;
    static_assert((NBL & 1) == 0, "bins are processed in packed pairs");
    f32x2 a[NP];            // v2 pairs
    f32x2 b[NP];            // v3 pairs

    __device__ __forceinline__ float v2(int i) const { return (i & 1)  ?  a[i >> 1].y  :  a[i >> 1].x; }
    __device__ __forceinline__ float v3(int i) const { return (i & 1)  ?  b[i >> 1].y  :  b[i >> 1].x; }
    __device__ __forceinline__ void set_v2(int i, float v) { if (i & 1) a[i >> 1].y = v; else a[i >> 1].x = v; }
    __device__ __forceinline__ void set_v3(int i, float v) { if (i & 1) b[i >> 1].y = v; else b[i >> 1].x = v; }

    // tone_detect.h:172-192: v3' = (fac*v2 - v1) + x, all bins
    __device__ __forceinline__ void step(const f32x2 (&fac)[NP], float x)
    {
        const f32x2 xx = {x, x};
        f32x2 t[NP];
#pragma unroll
        for (int i = 0;  i < NP;  i++)
            t[i] = fac[i]*b[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0;  i < NP;  i++)
            t[i] = t[i] - a[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0;  i < NP;  i++)
        {
            a[i] = b[i];
            b[i] = t[i] + xx;
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // Two consecutive samples held in one register pair (x01.x first, then x01.y): same arithmetic as two step()
    // calls, written as one asm block per group of up to four bin pairs, in place:
    //     sample 0:  a <- (fac*b - a) + x.lo      (a held v2 and becomes the new v3; b is now v2)
    //     sample 1:  b <- (fac*a - b) + x.hi      (b becomes the new v3; a is v2 again)
    // so (a, b) are (v2, v3) again after the pair and nothing is renamed.  Why asm: hipcc re-orders inside a phase and
    // then pads adjacent dependent packed ops with s_nop, and it copies the high sample into a fresh register pair
    // instead of naming it through op_sel; a lone wave per SIMD (a 65 536-channel bank) pays a full issue slot for
    // each of those.  FACS: the coefficients are wave-uniform (SGPR pairs).
    template <bool FACS>
    __device__ __forceinline__ void step2(const f32x2 (&fac)[NP], const f32x2 x01)
    {
        chain2<FACS, 0, (NP > 4)  ?  4  :  NP>(fac, x01);
        if constexpr (NP > 4)
            chain2<FACS, 4, NP - 4>(fac, x01);
    }

    // One sample, in place: a <- (fac*b - a) + x, then the roles are put right again by exchanging a and b.  For the odd
    // sample before or after a block end that does not fall on a pair boundary (same arithmetic as step()).
    template <bool FACS>
    __device__ __forceinline__ void step1(const f32x2 (&fac)[NP], const float x)
    {
        const f32x2 xx = {x, x};
#pragma unroll
        for (int i = 0;  i < NP;  i++)
        {
            f32x2 t;
#if defined(SPG_TONE_FMA)
#define SPG_STEP1 "v_pk_fma_f32 %0, %2, %3, %1 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\ts_nop 0\n\tv_pk_add_f32 %1, %0, %4 op_sel_hi:[1,0]"
#else
#define SPG_STEP1 "v_pk_mul_f32 %0, %2, %3\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0\n\t" \
                  "v_pk_add_f32 %1, %0, %4 op_sel_hi:[1,0]"
#endif
            if constexpr (FACS)
                asm(SPG_STEP1 : "=&v"(t), "+v"(a[i]) : "s"(fac[i]), "v"(b[i]), "v"(xx));
            else
                asm(SPG_STEP1 : "=&v"(t), "+v"(a[i]) : "v"(fac[i]), "v"(b[i]), "v"(xx));
#undef SPG_STEP1
            const f32x2 v3 = a[i];
            a[i] = b[i];
            b[i] = v3;
        }
    }

#define SPG_C_MUL(t, f, s)      "v_pk_mul_f32 %" #t ", %" #f ", %" #s "\n\t"
#define SPG_C_SUB(t, s)         "v_pk_add_f32 %" #t ", %" #t ", %" #s " neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define SPG_C_ADDL(d, t, x)     "v_pk_add_f32 %" #d ", %" #t ", %" #x " op_sel_hi:[1,0]\n\t"
#define SPG_C_ADDH(d, t, x)     "v_pk_add_f32 %" #d ", %" #t ", %" #x " op_sel:[0,1] op_sel_hi:[1,1]\n\t"
#define SPG_C_GAP               "s_nop 0\n\t"
#if !defined(SPG_TONE_FMA)
#define SPG_CHAIN_G1 SPG_C_MUL(2, 3, 1) SPG_C_GAP SPG_C_SUB(2, 0) SPG_C_GAP SPG_C_ADDL(0, 2, 4) SPG_C_GAP \
                     SPG_C_MUL(2, 3, 0) SPG_C_GAP SPG_C_SUB(2, 1) SPG_C_GAP SPG_C_ADDH(1, 2, 4)
#define SPG_CHAIN_G2 SPG_C_MUL(4, 6, 2) SPG_C_MUL(5, 7, 3) SPG_C_SUB(4, 0) SPG_C_SUB(5, 1) SPG_C_ADDL(0, 4, 8) SPG_C_ADDL(1, 5, 8) \
                     SPG_C_MUL(4, 6, 0) SPG_C_MUL(5, 7, 1) SPG_C_SUB(4, 2) SPG_C_SUB(5, 3) SPG_C_ADDH(2, 4, 8) SPG_C_ADDH(3, 5, 8)
#define SPG_CHAIN_G3 SPG_C_MUL(6, 9, 3) SPG_C_MUL(7, 10, 4) SPG_C_MUL(8, 11, 5) SPG_C_SUB(6, 0) SPG_C_SUB(7, 1) SPG_C_SUB(8, 2) \
                     SPG_C_ADDL(0, 6, 12) SPG_C_ADDL(1, 7, 12) SPG_C_ADDL(2, 8, 12) \
                     SPG_C_MUL(6, 9, 0) SPG_C_MUL(7, 10, 1) SPG_C_MUL(8, 11, 2) SPG_C_SUB(6, 3) SPG_C_SUB(7, 4) SPG_C_SUB(8, 5) \
                     SPG_C_ADDH(3, 6, 12) SPG_C_ADDH(4, 7, 12) SPG_C_ADDH(5, 8, 12)
#define SPG_CHAIN_G4 SPG_C_MUL(8, 12, 4) SPG_C_MUL(9, 13, 5) SPG_C_MUL(10, 14, 6) SPG_C_MUL(11, 15, 7) \
                     SPG_C_SUB(8, 0) SPG_C_SUB(9, 1) SPG_C_SUB(10, 2) SPG_C_SUB(11, 3) \
                     SPG_C_ADDL(0, 8, 16) SPG_C_ADDL(1, 9, 16) SPG_C_ADDL(2, 10, 16) SPG_C_ADDL(3, 11, 16) \
                     SPG_C_MUL(8, 12, 0) SPG_C_MUL(9, 13, 1) SPG_C_MUL(10, 14, 2) SPG_C_MUL(11, 15, 3) \
                     SPG_C_SUB(8, 4) SPG_C_SUB(9, 5) SPG_C_SUB(10, 6) SPG_C_SUB(11, 7) \
                     SPG_C_ADDH(4, 8, 16) SPG_C_ADDH(5, 9, 16) SPG_C_ADDH(6, 10, 16) SPG_C_ADDH(7, 11, 16)
#else
    // -DSPG_TONE_FMA: the experiment of round 6 (a BUILD variant, tools/build_variant.sh fma worktree
    // EXTRA=-DSPG_TONE_FMA): fac*v2 - v1 as one v_pk_fma_f32 -- one rounding where the reference has two, so the energies are
    // no longer the reference's to the bit (decisions and digits stay: tools/fma_check.py, profiles/r6_tone_fma.log)
#define SPG_C_FMA(t, f, s, a)   "v_pk_fma_f32 %" #t ", %" #f ", %" #s ", %" #a " neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
#define SPG_CHAIN_G1 SPG_C_FMA(2, 3, 1, 0) SPG_C_GAP SPG_C_ADDL(0, 2, 4) SPG_C_GAP SPG_C_FMA(2, 3, 0, 1) SPG_C_GAP SPG_C_ADDH(1, 2, 4)
#define SPG_CHAIN_G2 SPG_C_FMA(4, 6, 2, 0) SPG_C_FMA(5, 7, 3, 1) SPG_C_ADDL(0, 4, 8) SPG_C_ADDL(1, 5, 8) \
                     SPG_C_FMA(4, 6, 0, 2) SPG_C_FMA(5, 7, 1, 3) SPG_C_ADDH(2, 4, 8) SPG_C_ADDH(3, 5, 8)
#define SPG_CHAIN_G3 SPG_C_FMA(6, 9, 3, 0) SPG_C_FMA(7, 10, 4, 1) SPG_C_FMA(8, 11, 5, 2) \
                     SPG_C_ADDL(0, 6, 12) SPG_C_ADDL(1, 7, 12) SPG_C_ADDL(2, 8, 12) \
                     SPG_C_FMA(6, 9, 0, 3) SPG_C_FMA(7, 10, 1, 4) SPG_C_FMA(8, 11, 2, 5) \
                     SPG_C_ADDH(3, 6, 12) SPG_C_ADDH(4, 7, 12) SPG_C_ADDH(5, 8, 12)
#define SPG_CHAIN_G4 SPG_C_FMA(8, 12, 4, 0) SPG_C_FMA(9, 13, 5, 1) SPG_C_FMA(10, 14, 6, 2) SPG_C_FMA(11, 15, 7, 3) \
                     SPG_C_ADDL(0, 8, 16) SPG_C_ADDL(1, 9, 16) SPG_C_ADDL(2, 10, 16) SPG_C_ADDL(3, 11, 16) \
                     SPG_C_FMA(8, 12, 0, 4) SPG_C_FMA(9, 13, 1, 5) SPG_C_FMA(10, 14, 2, 6) SPG_C_FMA(11, 15, 3, 7) \
                     SPG_C_ADDH(4, 8, 16) SPG_C_ADDH(5, 9, 16) SPG_C_ADDH(6, 10, 16) SPG_C_ADDH(7, 11, 16)
#endif
    // G bin pairs starting at pair O.  Dependent packed ops are G instructions apart; for G = 1 they are adjacent
    // and need the one wait state hipcc would put there.
    template <bool FACS, int O, int G>
    __device__ __forceinline__ void chain2(const f32x2 (&fac)[NP], const f32x2 x)
    {
        static_assert(G >= 1  &&  G <= 4  &&  O + G <= NP, "groups of one to four pairs");
        f32x2 t0;
        f32x2 t1;
        f32x2 t2;
        f32x2 t3;
        if constexpr (G == 1)
        {
            if constexpr (FACS)
                asm(SPG_CHAIN_G1
                    : "+v"(a[O]), "+v"(b[O]), "=&v"(t0) : "s"(fac[O]), "v"(x));
            else
                asm(SPG_CHAIN_G1
                    : "+v"(a[O]), "+v"(b[O]), "=&v"(t0) : "v"(fac[O]), "v"(x));
        }
        else if constexpr (G == 2)
        {
            constexpr int P = O;
            if constexpr (FACS)
                asm(SPG_CHAIN_G2
                    : "+v"(a[P]), "+v"(a[P + 1]), "+v"(b[P]), "+v"(b[P + 1]), "=&v"(t0), "=&v"(t1)
                    : "s"(fac[P]), "s"(fac[P + 1]), "v"(x));
            else
                asm(SPG_CHAIN_G2
                    : "+v"(a[P]), "+v"(a[P + 1]), "+v"(b[P]), "+v"(b[P + 1]), "=&v"(t0), "=&v"(t1)
                    : "v"(fac[P]), "v"(fac[P + 1]), "v"(x));
        }
        else if constexpr (G == 3)
        {
            constexpr int P = O;
            if constexpr (FACS)
                asm(SPG_CHAIN_G3
                    : "+v"(a[P]), "+v"(a[P + 1]), "+v"(a[P + 2]), "+v"(b[P]), "+v"(b[P + 1]), "+v"(b[P + 2]),
                      "=&v"(t0), "=&v"(t1), "=&v"(t2)
                    : "s"(fac[P]), "s"(fac[P + 1]), "s"(fac[P + 2]), "v"(x));
            else
                asm(SPG_CHAIN_G3
                    : "+v"(a[P]), "+v"(a[P + 1]), "+v"(a[P + 2]), "+v"(b[P]), "+v"(b[P + 1]), "+v"(b[P + 2]),
                      "=&v"(t0), "=&v"(t1), "=&v"(t2)
                    : "v"(fac[P]), "v"(fac[P + 1]), "v"(fac[P + 2]), "v"(x));
        }
        else
        {
            constexpr int P = O;
            if constexpr (FACS)
                asm(SPG_CHAIN_G4
                    : "+v"(a[P]), "+v"(a[P + 1]), "+v"(a[P + 2]), "+v"(a[P + 3]),
                      "+v"(b[P]), "+v"(b[P + 1]), "+v"(b[P + 2]), "+v"(b[P + 3]),
                      "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                    : "s"(fac[P]), "s"(fac[P + 1]), "s"(fac[P + 2]), "s"(fac[P + 3]), "v"(x));
            else
                asm(SPG_CHAIN_G4
                    : "+v"(a[P]), "+v"(a[P + 1]), "+v"(a[P + 2]), "+v"(a[P + 3]),
                      "+v"(b[P]), "+v"(b[P + 1]), "+v"(b[P + 2]), "+v"(b[P + 3]),
                      "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                    : "v"(fac[P]), "v"(fac[P + 1]), "v"(fac[P + 2]), "v"(fac[P + 3]), "v"(x));
        }
    }
#undef SPG_C_MUL
#undef SPG_C_SUB
#undef SPG_C_ADDL
#undef SPG_C_ADDH
#undef SPG_C_GAP
#undef SPG_CHAIN_G1
#undef SPG_CHAIN_G2
#undef SPG_CHAIN_G3
#undef SPG_CHAIN_G4
#if defined(SPG_TONE_FMA)
#undef SPG_C_FMA
#endif

    // goertzel_result() for every bin: one zero sample, energy, reset (tone_detect.c:160-205)
    // Written phase by phase over all pairs (the same nine roundings per pair, in the same order:
    // q = fac*p - a;  r = ((q*q + p*p) - (p*q)*fac)*2): a pair's chain taken alone has every packed operation wait on
    // the one before it, which hipcc pads with s_nop -- a full issue slot each for a lone wave.
    // `sum`, if asked for: the energies added up pair-wise (any order does for what it is used for: is one of them a NaN)
    __device__ __forceinline__ void finish(const f32x2 (&fac)[NP], float (&e)[NBL], f32x2 *sum = nullptr)
    {
        f32x2 q[NP];
        f32x2 t[NP];
        f32x2 u[NP];
#pragma unroll
        for (int i = 0;  i < NP;  i++)
            q[i] = fac[i]*b[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0;  i < NP;  i++)
            q[i] = q[i] - a[i];                 // becomes v3 (b holds what becomes v2)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0;  i < NP;  i++)
            t[i] = q[i]*q[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0;  i < NP;  i++)
            u[i] = b[i]*b[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0;  i < NP;  i++)
            t[i] = t[i] + u[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0;  i < NP;  i++)
            u[i] = b[i]*q[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0;  i < NP;  i++)
            u[i] = u[i]*fac[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0;  i < NP;  i++)
            t[i] = t[i] - u[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0;  i < NP;  i++)
        {
            t[i] *= 2.0f;
            e[2*i] = t[i].x;
            e[2*i + 1] = t[i].y;
            a[i] = f32x2{0.0f, 0.0f};
            b[i] = f32x2{0.0f, 0.0f};
        }
        if (sum)
        {
            f32x2 acc = t[0];
#pragma unroll
            for (int i = 1;  i < NP;  i++)
                acc += t[i];
            *sum = acc;
        }
    }
};

// Record word written per completed block: hit | code<<8 | flags<<16
__device__ __forceinline__ uint32_t make_rec(int hit, int code, int flags)
{
    return (uint32_t) (hit & 0xFF) | ((uint32_t) (code & 0xFF) << 8) | ((uint32_t) flags << 16);
}

constexpr int kBlkValid = 0x01;
constexpr int kBlkChange = 0x02;
constexpr int kBlkReport = 0x04;
constexpr int kBlkToneOff = 0x08;

// 16 / 25 character key tables packed little-endian into 64-bit words so the lookup
// is two shifts instead of a divergent memory read.
__device__ __forceinline__ int key_from(const uint64_t w0, const uint64_t w1, const uint64_t w2, const uint64_t w3, int idx)
{
    const uint64_t w = (idx < 8)  ?  w0  :  (idx < 16)  ?  w1  :  (idx < 24)  ?  w2  :  w3;
    return (int) ((w >> ((idx & 7)*8)) & 0xFF);
}

constexpr uint64_t pack8(const char *s)
{
    uint64_t v = 0;
    for (int i = 7;  i >= 0;  i--)
        v = (v << 8) | (uint8_t) s[i];
    return v;
}

// ---------------------------------------------------------------------------------
// Detector policies.  Each supplies: NB, NSF (floats of state per channel), whether a
// block energy is accumulated, the optional input filter, and the block-end decision
// `decide()` over the NB Goertzel energies.  Integer state is two 32-bit words per
// channel: w0 = current_sample (16 bits) | detector bytes, w1 = detector word.
// `store` is false for lanes that must not write (shadow lanes; the second lane of a
// channel when LPC = 2) -- they still update their private copy of w0/w1.
// ---------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void write_trace(const ToneLaunch &L, const float (&e)[NB], float total, int ch, int nb)
{
#pragma unroll
    for (int i = 0;  i < NB;  i++)
        L.trace[((size_t) nb*(NB + 1) + i)*L.n_ch + ch] = e[i];
    L.trace[((size_t) nb*(NB + 1) + NB)*L.n_ch + ch] = total;
}

// The digit a block delivered, as one byte (0 = none): DTMF = the debouncer accepted a digit (dtmf.c:318-340: a change to
// a non-zero code), the MF detectors = the digit of a report (Bell MF: accepted, R2 MF: changed).
constexpr int kToneCadence = 8192;     // ABL bit: the super-tone cadence epilogue is compiled in (tone_fast.hpp)
constexpr int kToneDigits = 4096;       // ABL bit: the digit-byte stores are compiled in (a variant of its own: with the
                                        // stores merely switched by a pointer test the plain kernel ran 4 % slower)

template <bool DTMF>
__device__ __forceinline__ uint8_t tone_digit_byte(uint32_t recw)
{
    const uint32_t flags = (recw >> 16) & 0xFF;
    const uint32_t code = (recw >> 8) & 0xFF;
    const bool ev = DTMF  ?  ((flags & kBlkChange)  &&  code != 0)  :  ((flags & kBlkReport) != 0);
    return (uint8_t) (ev  ?  code  :  0u);
}

// v_max_f32 / v_min_f32 / v_max3_f32 on operands the caller knows to be ordinary numbers
__device__ __forceinline__ float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmed3(float a, float b, float c)
{
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---- DTMF (src/dtmf.c:132-361) ------------------------------------------------------
template <bool FILTER>
struct DtmfDet
{
    static constexpr int NB = 8;
    static constexpr bool kEnergy = true;
    static constexpr bool kDuration = true;
    static constexpr bool kFilter = FILTER;
    static constexpr bool kDigits = true;             // the kernels can also write one digit byte per block (ABL bit kToneDigits)
    static constexpr int NSF = 2*NB + 1 + 4;                // v2, v3, energy, z350[2], z440[2]
    __device__ static __forceinline__ int block_len(const ToneLaunch &) { return 102; }    // dtmf.c:71

    float z[FILTER  ?  4  :  1];
    bool filt;                  // this channel's dial tone filter is on (dtmf.c:421-445 sets it per detector)

    __device__ __forceinline__ void load_extra(const ToneLaunch &L, int ch)
    {
        filt = FILTER;
        if (FILTER)
        {
#pragma unroll
            for (int i = 0;  i < 4;  i++)
                z[i] = L.sf[(size_t) (2*NB + 1 + i)*L.n_ch + ch];
            if (L.chan_parms)
                filt = (L.chan_parms[(size_t) 3*L.n_ch + ch] != 0.0f);
        }
    }
    __device__ __forceinline__ void store_extra(const ToneLaunch &L, int ch)
    {
        if (FILTER)
        {
#pragma unroll
            for (int i = 0;  i < 4;  i++)
                L.sf[(size_t) (2*NB + 1 + i)*L.n_ch + ch] = z[i];
        }
    }

    // dtmf.c:167-183 -- two high-Q notches at 350 Hz and 440 Hz, float all the way
    __device__ __forceinline__ float prefilter(float x)
    {
        if (FILTER)
        {
            float v1 = 0.98356f*x + 1.8954426f*z[0] - 0.9691396f*z[1];
            float f = v1 - 1.9251480f*z[0] + z[1];
            const float v2 = 0.98456f*f + 1.8529543f*z[2] - 0.9691396f*z[3];
            const float g = v2 - 1.8819938f*z[2] + z[3];
            // a channel whose filter is off keeps its (zero) filter state and sees the raw sample
            z[1] = filt  ?  z[0]  :  z[1];
            z[0] = filt  ?  v1  :  z[0];
            z[3] = filt  ?  z[2]  :  z[3];
            z[2] = filt  ?  v2  :  z[2];
            return filt  ?  g  :  x;
        }
        return x;
    }

    // Decision (dtmf.c:209-258) and debounce (dtmf.c:304-347).
    // w0 = cs | last_hit<<16 | in_digit<<24 ; w1 = duration.
    __device__ __forceinline__ uint32_t decide(const ToneLaunch &L, const float (&e)[NB], float &energy,
                                               uint32_t &w0, int32_t &w1, int ch, int nb, bool store)
    {
        if (L.trace  &&  store)
            write_trace<NB>(L, e, energy, ch, nb);
        int br = 0;
        int bc = 0;
        float er = e[0];
        float ec = e[4];
#pragma unroll
        for (int i = 1;  i < 4;  i++)
        {
            if (e[i] > er)
            {
                er = e[i];
                br = i;
            }
            if (e[4 + i] > ec)
            {
                ec = e[4 + i];
                bc = i;
            }
        }
        // All tests are side-effect free: evaluate every one and combine with non-short-circuit logic (no divergent
        // branches at a block end, which every lane of the wave reaches together).
        float threshold = L.threshold;
        float normal_twist = L.normal_twist;
        float reverse_twist = L.reverse_twist;
        if (L.chan_parms)
        {
            threshold = L.chan_parms[ch];
            normal_twist = L.chan_parms[(size_t) L.n_ch + ch];
            reverse_twist = L.chan_parms[(size_t) 2*L.n_ch + ch];
        }
        bool ok = (er >= threshold)  &  (ec >= threshold);
        ok = ok  &  (ec < er*reverse_twist)  &  (ec*normal_twist > er);
        bool off_peak = false;
#pragma unroll
        for (int i = 0;  i < 4;  i++)
        {
            // dtmf.c:243-246; relative peak ratios are both 6.309f (dtmf.c:107-108)
            off_peak |= ((i != bc)  &  (e[4 + i]*6.309f > ec))  |  ((i != br)  &  (e[i]*6.309f > er));
        }
        ok = ok  &  !off_peak  &  ((er + ec) > 83.868f*energy);  // dtmf.c:109,250-252
        constexpr uint64_t k0 = pack8("123A456B");
        constexpr uint64_t k1 = pack8("789C*0#D");
        const int raw = ok  ?  key_from(k0, k1, 0, 0, (br << 2) + bc)  :  0;

        int last_hit = (w0 >> 16) & 0xFF;
        int in_digit = (w0 >> 24) & 0xFF;
        int hit = raw;
        int flags = kBlkValid;
        int code = in_digit;
        if (hit != in_digit  &&  last_hit != in_digit)
        {
            flags |= kBlkChange;
            hit = (hit  &&  hit == last_hit)  ?  hit  :  0;
            if (in_digit  ||  hit)
            {
                flags |= kBlkReport;
                if (in_digit  &&  !hit)
                    flags |= kBlkToneOff;
                if (L.realtime)
                {
                    if (L.rec_dur  &&  store)
                        L.rec_dur[(size_t) nb*L.n_ch + ch] = w1;
                    w1 = 0;
                }
            }
            in_digit = hit;
            code = hit;
        }
        last_hit = hit;
        if (store  &&  L.rec_energy)
            L.rec_energy[(size_t) nb*L.n_ch + ch] = energy;
        energy = 0.0f;
        w0 = ((uint32_t) last_hit << 16) | ((uint32_t) in_digit << 24);       // cs = 0
        return make_rec(raw, code, flags);
    }

    // The same decision and debounce for the production case -- bank-wide thresholds, no trace, no per-block energy
    // or duration reports (the streaming kernels test that once per launch) -- in a third of the instructions:
    //   * the strongest tone of a group is a max tree, the runner-up the second order statistic of the four
    //     (max(min(a,b), min(c,d), min(max(a,b), max(c,d)))), and "some other tone within 6.309 of the peak"
    //     (dtmf.c:243-246) is the one test runner_up*6.309f > peak: x -> fl(x*6.309f) is monotonic, so if any tone
    //     passes that test the largest of them does;
    //   * which tone the peak was is only looked up to form the key: the first one equal to it, which is what the
    //     reference's strict > scan from tone 0 keeps (dtmf.c:211-223).
    // v_max / v_min drop a NaN operand where the reference's compares carry it, so energies that are not all
    // ordinary numbers (a state imported with infinities in it) go to decide(); `plain_ok` is that test, made by the
    // caller for the whole wave.
    static constexpr bool kLean = true;
    __device__ static __forceinline__ bool lean_inputs_ok(const float (&e)[NB])
    {
        const float s = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
        return s == s;          // a NaN anywhere, or infinities of both signs, make the sum a NaN
    }
    __device__ __forceinline__ uint32_t decide_plain(const ToneLaunch &L, const float (&e)[NB], float &energy, uint32_t &w0, int32_t &)
    {
        // (asm: written with the builtins, hipcc canonicalises every operand first -- a v_max_f32 x, x each -- for the
        // signalling NaNs the caller has already excluded)
        const float rh01 = vmax(e[0], e[1]);
        const float rl01 = vmin(e[0], e[1]);
        const float rh23 = vmax(e[2], e[3]);
        const float rl23 = vmin(e[2], e[3]);
        const float ch01 = vmax(e[4], e[5]);
        const float cl01 = vmin(e[4], e[5]);
        const float ch23 = vmax(e[6], e[7]);
        const float cl23 = vmin(e[6], e[7]);
        const float er = vmax(rh01, rh23);
        const float ec = vmax(ch01, ch23);
        const float sr = vmax3(rl01, rl23, vmin(rh01, rh23));
        const float sc = vmax3(cl01, cl23, vmin(ch01, ch23));
        bool ok = (er >= L.threshold)  &  (ec >= L.threshold);
        ok = ok  &  (ec < er*L.reverse_twist)  &  (ec*L.normal_twist > er);
        ok = ok  &  !(sr*6.309f > er)  &  !(sc*6.309f > ec);
        ok = ok  &  ((er + ec) > 83.868f*energy);
        constexpr uint32_t kRow0 = '1' | ('2' << 8) | ('3' << 16) | ((uint32_t) 'A' << 24);
        constexpr uint32_t kRow1 = '4' | ('5' << 8) | ('6' << 16) | ((uint32_t) 'B' << 24);
        constexpr uint32_t kRow2 = '7' | ('8' << 8) | ('9' << 16) | ((uint32_t) 'C' << 24);
        constexpr uint32_t kRow3 = '*' | ('0' << 8) | ('#' << 16) | ((uint32_t) 'D' << 24);
        // (selects, not branches: all lanes are here)
        uint32_t keys = __builtin_unpredictable(e[2] == er)  ?  kRow2  :  kRow3;
        keys = __builtin_unpredictable(e[1] == er)  ?  kRow1  :  keys;
        keys = __builtin_unpredictable(e[0] == er)  ?  kRow0  :  keys;
        uint32_t shift = __builtin_unpredictable(e[6] == ec)  ?  16u  :  24u;
        shift = __builtin_unpredictable(e[5] == ec)  ?  8u  :  shift;
        shift = __builtin_unpredictable(e[4] == ec)  ?  0u  :  shift;
        const uint32_t raw = ok  ?  ((keys >> shift) & 0xFFu)  :  0u;

        // dtmf.c:304-347
        const uint32_t last_hit = (w0 >> 16) & 0xFFu;
        const uint32_t in_digit = w0 >> 24;
        const bool chg = (raw != in_digit)  &  (last_hit != in_digit);
        const uint32_t conf = (raw == last_hit)  ?  raw  :  0u;       // a hit counts when the block before had it too
        const uint32_t hit = chg  ?  conf  :  raw;
        const uint32_t now = chg  ?  conf  :  in_digit;
        uint32_t flags = kBlkValid;
        flags |= chg  ?  kBlkChange  :  0;
        flags |= (chg  &  ((in_digit | conf) != 0))  ?  kBlkReport  :  0;
        flags |= (chg  &  (in_digit != 0)  &  (conf == 0))  ?  kBlkToneOff  :  0;
        energy = 0.0f;
        w0 = (hit << 16) | (now << 24);                               // cs = 0
        return raw | (now << 8) | (flags << 16);
    }
};

// ---- Bell MF / R2 MF (src/bell_r2_mf.c:507-673, :750-880) -------------------------------
// Two strongest of six + level / twist / relative-peak tests (:556-622 == :793-858).
__device__ __forceinline__ int mf_pick_pair(const float (&e)[6], float threshold, float twist, float rel_peak)
{
    int best;
    int second;
    if (e[0] > e[1])
    {
        best = 0;
        second = 1;
    }
    else
    {
        best = 1;
        second = 0;
    }
    float eb = (e[0] > e[1])  ?  e[0]  :  e[1];
    float es = (e[0] > e[1])  ?  e[1]  :  e[0];
#pragma unroll
    for (int i = 2;  i < 6;  i++)
    {
        if (e[i] >= eb)
        {
            second = best;
            es = eb;
            best = i;
            eb = e[i];
        }
        else if (e[i] >= es)
        {
            second = i;
            es = e[i];
        }
    }
    bool ok = (eb >= threshold)  &  (es >= threshold)  &  (eb < es*twist)  &  (eb*twist > es);
#pragma unroll
    for (int i = 0;  i < 6;  i++)
        ok = ok  &  !((i != best)  &  (i != second)  &  (e[i]*rel_peak >= es));
    if (!ok)
        return -1;
    const int lo = (second < best)  ?  second  :  best;
    const int hi = (second < best)  ?  best  :  second;
    return lo*5 + hi - 1;
}

// The same pick for the production case, on energies the caller knows to be ordinary numbers (no NaN), as the key it
// leads to (0 = no digit), without the scan:
//   * the tests need three VALUES only -- the largest, the runner-up and the third of the six (two sorted triples merged:
//     v_max3 / v_med3 / v_min3).  "Some tone that is neither of the two within rel_peak of the runner-up" is one test on the
//     third: x -> fl(x*rel_peak) is monotonic, so if any tone passes it the largest of them does;
//   * WHICH tones the two are matters only when every test has passed, and then the third is strictly below the runner-up
//     (fl(e3*rel_peak) < es with es >= threshold > 0 and rel_peak > 1), so exactly two tones satisfy e[i] >= es -- the pair,
//     whatever the reference's >= tie-breaks (bell_r2_mf.c:567-581) did among equal values: the key depends on the
//     unordered pair only (:642-644 here, lo*5 + hi - 1).
// The key table is five words, one per lower tone of the pair, holding the keys of the upper tones 2..5 in bytes 0..3; the pair
// (0, 1) has a constant of its own.  `tab` is the 25-character table of the reference (bell_mf_positions / r2_mf_positions).
constexpr uint32_t mf_row_word(const char *tab, int lo)
{
    uint32_t w = 0;
    for (int hi = 2;  hi <= 5;  hi++)
    {
        if (hi > lo)
            w |= (uint32_t) (uint8_t) tab[lo*5 + hi - 1] << (8*(hi - 2));
    }
    return w;
}

template <class Tab>
__device__ __forceinline__ uint32_t mf_pick_key_lean(const float (&e)[6], float threshold, float twist, float rel_peak)
{
    const float ha = vmax3(e[0], e[1], e[2]);
    const float ma = vmed3(e[0], e[1], e[2]);
    const float la = vmin3(e[0], e[1], e[2]);
    const float hb = vmax3(e[3], e[4], e[5]);
    const float mb = vmed3(e[3], e[4], e[5]);
    const float lb = vmin3(e[3], e[4], e[5]);
    const float eb = vmax(ha, hb);
    const float es = vmax3(vmin(ha, hb), ma, mb);
    const float e3 = vmax(vmax3(la, lb, vmin(ma, hb)), vmin(ha, mb));
    bool ok = (eb >= threshold)  &  (es >= threshold)  &  (eb < es*twist)  &  (eb*twist > es);
    ok = ok  &  !(e3*rel_peak >= es);
    constexpr uint32_t w0 = mf_row_word(Tab::str(), 0);
    constexpr uint32_t w1 = mf_row_word(Tab::str(), 1);
    constexpr uint32_t w2 = mf_row_word(Tab::str(), 2);
    constexpr uint32_t w3 = mf_row_word(Tab::str(), 3);
    constexpr uint32_t w4 = mf_row_word(Tab::str(), 4);
    constexpr uint32_t k01 = (uint8_t) Tab::str()[0];
    const bool c0 = e[0] >= es;
    const bool c1 = e[1] >= es;
    const bool c2 = e[2] >= es;
    const bool c3 = e[3] >= es;
    const bool c4 = e[4] >= es;
    const bool c5 = e[5] >= es;
    // (selects, not branches: all lanes are here)
    uint32_t row = __builtin_unpredictable(c3)  ?  w3  :  w4;
    row = __builtin_unpredictable(c2)  ?  w2  :  row;
    row = __builtin_unpredictable(c1)  ?  w1  :  row;
    row = __builtin_unpredictable(c0)  ?  w0  :  row;
    uint32_t sh = __builtin_unpredictable(c3)  ?  8u  :  0u;
    sh = __builtin_unpredictable(c4)  ?  16u  :  sh;
    sh = __builtin_unpredictable(c5)  ?  24u  :  sh;
    uint32_t key = (row >> sh) & 0xFFu;
    key = __builtin_unpredictable(c0  &  c1)  ?  k01  :  key;
    return ok  ?  key  :  0u;
}

struct BellMfTab { __device__ __host__ static constexpr const char *str() { return "1247C-358A--69*---0B----#"; } };     // bell_r2_mf.c:262
struct R2MfTab { __device__ __host__ static constexpr const char *str() { return "1247B-358C--69D---0E----F"; } };       // bell_r2_mf.c:276

struct BellMfDet
{
    static constexpr bool kLean = true;
    static constexpr int NB = 6;
    static constexpr bool kEnergy = false;
    static constexpr bool kDuration = false;
    static constexpr bool kFilter = false;
    static constexpr bool kDigits = true;
    static constexpr int NSF = 2*NB;
    __device__ static __forceinline__ int block_len(const ToneLaunch &) { return 120; }    // bell_r2_mf.c:204
    __device__ __forceinline__ void load_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ void store_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ float prefilter(float x) { return x; }

    // w0 = cs | hits[0]<<16 | hits[1]<<24 ; w1 = hits[2] | hits[3]<<8 | hits[4]<<16
    __device__ __forceinline__ uint32_t decide(const ToneLaunch &L, const float (&e)[NB], float &,
                                               uint32_t &w0, int32_t &w1, int ch, int nb, bool store)
    {
        if (L.trace  &&  store)
            write_trace<NB>(L, e, 0.0f, ch, nb);
        // bell_r2_mf.c:236-238
        const int idx = mf_pick_pair(e, 3343803100.0f, 3.981f, 12.589f);
        constexpr uint64_t k0 = pack8("1247C-35");
        constexpr uint64_t k1 = pack8("8A--69*-");
        constexpr uint64_t k2 = pack8("--0B----");
        constexpr uint64_t k3 = pack8("#\0\0\0\0\0\0\0");
        const int hit = (idx >= 0)  ?  key_from(k0, k1, k2, k3, idx)  :  0;
        const int h0 = (w0 >> 16) & 0xFF;
        const int h1 = (w0 >> 24) & 0xFF;
        const uint32_t u1 = (uint32_t) w1;
        const int h2 = u1 & 0xFF;
        const int h3 = (u1 >> 8) & 0xFF;
        const int h4 = (u1 >> 16) & 0xFF;
        int flags = kBlkValid;
        int code = 0;
        // bell_r2_mf.c:629-635
        if (hit
            &&  hit == h4
            &&  hit == h3
            &&  ((hit != '*'  &&  hit != h2  &&  hit != h1)
                 ||
                 (hit == '*'  &&  hit == h2  &&  hit != h1  &&  hit != h0)))
        {
            flags |= kBlkReport;
            code = hit;
        }
        // bell_r2_mf.c:657-661: shift the hit history
        w0 = ((uint32_t) h1 << 16) | ((uint32_t) h2 << 24);
        w1 = (int32_t) ((uint32_t) h3 | ((uint32_t) h4 << 8) | ((uint32_t) hit << 16));
        return make_rec(hit, code, flags);
    }

    // The production case (no trace asked for; the streaming kernels test that once per launch): the pick without the scan
    // (mf_pick_key_lean) and the five-block rule of bell_r2_mf.c:629-635 on the packed history words as they are.
    __device__ __forceinline__ uint32_t decide_plain(const ToneLaunch &, const float (&e)[NB], float &, uint32_t &w0, int32_t &w1)
    {
        const uint32_t hit = mf_pick_key_lean<BellMfTab>(e, 3343803100.0f, 3.981f, 12.589f);
        const uint32_t u1 = (uint32_t) w1;
        const uint32_t h0 = (w0 >> 16) & 0xFFu;
        const uint32_t h1 = w0 >> 24;
        const uint32_t h2 = u1 & 0xFFu;
        const uint32_t h34 = (u1 >> 8) & 0xFFFFu;
        const bool star = (hit == (uint32_t) '*');
        bool rep = (hit != 0)  &  (h34 == hit*0x0101u)  &  (hit != h1);
        rep = rep  &  (star  ?  ((hit == h2)  &  (hit != h0))  :  (hit != h2));
        const uint32_t code = rep  ?  hit  :  0u;
        const uint32_t flags = rep  ?  (uint32_t) (kBlkValid | kBlkReport)  :  (uint32_t) kBlkValid;
        w0 = (h1 << 16) | (h2 << 24);                                   // bell_r2_mf.c:657-661; cs = 0
        w1 = (int32_t) (h34 | (hit << 16));
        return hit | (code << 8) | (flags << 16);
    }
};

struct R2MfDet
{
    static constexpr bool kLean = true;
    static constexpr int NB = 6;
    static constexpr bool kEnergy = false;
    static constexpr bool kDuration = false;
    static constexpr bool kFilter = false;
    static constexpr bool kDigits = true;
    static constexpr int NSF = 2*NB;
    __device__ static __forceinline__ int block_len(const ToneLaunch &) { return 133; }    // bell_r2_mf.c:206
    __device__ __forceinline__ void load_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ void store_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ float prefilter(float x) { return x; }

    // w0 = cs | current_digit<<16
    __device__ __forceinline__ uint32_t decide(const ToneLaunch &L, const float (&e)[NB], float &,
                                               uint32_t &w0, int32_t &, int ch, int nb, bool store)
    {
        if (L.trace  &&  store)
            write_trace<NB>(L, e, 0.0f, ch, nb);
        // bell_r2_mf.c:240-242
        const int idx = mf_pick_pair(e, 1031766650.0f, 5.012f, 12.589f);
        constexpr uint64_t k0 = pack8("1247B-35");
        constexpr uint64_t k1 = pack8("8C--69D-");
        constexpr uint64_t k2 = pack8("--0E----");
        constexpr uint64_t k3 = pack8("F\0\0\0\0\0\0\0");
        const int digit = (idx >= 0)  ?  key_from(k0, k1, k2, k3, idx)  :  0;
        const int current = (w0 >> 16) & 0xFF;
        int flags = kBlkValid;
        if (current != digit)
            flags |= kBlkReport;                                    // bell_r2_mf.c:869-875
        w0 = (uint32_t) digit << 16;
        return make_rec(digit, digit, flags);
    }

    __device__ __forceinline__ uint32_t decide_plain(const ToneLaunch &, const float (&e)[NB], float &, uint32_t &w0, int32_t &)
    {
        const uint32_t digit = mf_pick_key_lean<R2MfTab>(e, 1031766650.0f, 5.012f, 12.589f);
        const uint32_t flags = (((w0 >> 16) & 0xFFu) != digit)  ?  (uint32_t) (kBlkValid | kBlkReport)  :  (uint32_t) kBlkValid;      // bell_r2_mf.c:869-875
        w0 = digit << 16;
        return digit | (digit << 8) | (flags << 16);
    }
};

// ---- Super tone (src/super_tone_rx.c:289-362 on device; cadence FSM on the host) --------
// and the generic Goertzel bank (goertzel_update / goertzel_result, tone_detect.c:123-205).
template <int NBINS, bool SUPER>
struct MultiDet
{
    static constexpr bool kLean = false;
    static constexpr int NB = NBINS;
    // the generic bank keeps the block's total energy too: the Goertzel users outside tone_detect.c gate their
    // decisions on it (v18.c:1559,1597, ademco_contactid.c:903,920)
    static constexpr bool kEnergy = true;
    static constexpr bool kDuration = false;
    static constexpr bool kFilter = false;
    static constexpr bool kDigits = false;
    static constexpr int NSF = 2*NB + 1;
    __device__ static __forceinline__ int block_len(const ToneLaunch &L) { return SUPER  ?  128  :  L.block_len; }
    __device__ __forceinline__ void load_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ void store_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ float prefilter(float x) { return x; }

    __device__ __forceinline__ uint32_t decide(const ToneLaunch &L, const float (&ein)[NB], float &energy,
                                               uint32_t &w0, int32_t &, int ch, int nb, bool store)
    {
        const int m = L.nbins;
        uint32_t recw;
        if (SUPER)
        {
            int k1 = -1;
            int k2 = -1;
            // super_tone_rx.c:301-309: below the total-energy gate the bins are reset unread
            const bool loud = !(energy < 2104205.6f);               // super_tone_rx.c:75
            float e[NB];
#pragma unroll
            for (int i = 0;  i < NB;  i++)
                e[i] = loud  ?  ein[i]  :  0.0f;
            if (loud)
            {
                // super_tone_rx.c:320-347 (requires m >= 2)
                float e1;
                float e2;
                if (e[0] > e[1])
                {
                    k1 = 0;
                    k2 = 1;
                    e1 = e[0];
                    e2 = e[1];
                }
                else
                {
                    k1 = 1;
                    k2 = 0;
                    e1 = e[1];
                    e2 = e[0];
                }
#pragma unroll
                for (int j = 2;  j < NB;  j++)
                {
                    if (j < m)
                    {
                        if (e[j] >= e1)
                        {
                            k2 = k1;
                            e2 = e1;
                            k1 = j;
                            e1 = e[j];
                        }
                        else if (e[j] >= e2)
                        {
                            k2 = j;
                            e2 = e[j];
                        }
                    }
                }
                // super_tone_rx.c:348-362 (constants :76-77)
                if ((e1 + e2) < 1.995f*energy)
                {
                    k1 = -1;
                    k2 = -1;
                }
                else if (e1 > 3.981f*e2)
                {
                    k2 = -1;
                }
                else if (k2 < k1)
                {
                    const int t = k1;
                    k1 = k2;
                    k2 = t;
                }
            }
            if (store)
            {
                if (L.rec_energy)
                    L.rec_energy[(size_t) nb*L.n_ch + ch] = energy;
                if (L.trace)
                    write_trace<NB>(L, e, energy, ch, nb);
            }
            recw = make_rec(k1 + 1, k2 + 1, kBlkValid);
            energy = 0.0f;
        }
        else
        {
            if (store)
            {
                if (L.rec_energy)
                    L.rec_energy[(size_t) nb*L.n_ch + ch] = energy;
                if (L.trace)
                    write_trace<NB>(L, ein, energy, ch, nb);
            }
            int hit = 0;
            if (L.functor == 1)
            {
                // v18.c:1580-1600: strongest bin by a strict > scan from zero, level test, fraction-of-total test
                float best = 0.0f;
#pragma unroll
                for (int i = 0;  i < NB;  i++)
                {
                    if (i < m  &&  ein[i] > best)
                    {
                        best = ein[i];
                        hit = i;
                    }
                }
                if (best < L.functor_threshold  ||  best <= 83.868f*energy)     // v18.c:192
                    hit = 0;
            }
            else if (L.functor == 2)
            {
                // ademco_contactid.c:915-935 (constants :461-462)
                const float e1400 = ein[0];
                const float e2300 = ein[(NB > 1)  ?  1  :  0];
                if (e1400 > 49728296.6f  ||  e2300 > 49728296.6f)
                {
                    if (e1400 > e2300)
                        hit = (e1400 > 45.2233f*energy)  ?  1  :  0;
                    else
                        hit = (e2300 > 45.2233f*energy)  ?  2  :  0;
                }
            }
            recw = make_rec(hit, 0, kBlkValid);
            energy = 0.0f;
        }
        w0 = 0;
        return recw;
    }
};

// ---------------------------------------------------------------------------------
// Frame staging.  The rows of one 80-sample segment (160 B per channel) of a wave's
// channels are copied HBM -> LDS by the LDS-DMA path (global_load_lds_dwordx4: 16 B per
// lane, 1 KiB per instruction, no VGPR round trip) into one of two per-wave LDS
// buffers.  The LDS image is lane-linear (dest = base + lane*16), i.e. row-major
// [rows][80] int16 with a 160-byte pitch; each lane then reads its own row 8 samples
// (ds_read_b128) at a time.  (A 160-byte pitch gives 2-way bank conflicts on those
// reads; LDS traffic is a few % of the kernel's cycles, so linear-and-coalesced beats
// padded-and-scattered here.)  The DMA of segment s+1 is in flight while segment s is
// consumed.  The DMA is issued from inline asm (hipcc neither counts it in its s_waitcnt
// bookkeeping nor drains it early); completion is awaited with an explicit
// s_waitcnt vmcnt(0) immediately before the first read of the buffer.
// ---------------------------------------------------------------------------------
constexpr int kSeg = 80;                            // samples per LDS segment
constexpr int kRowBytes = kSeg*2;                   // 160 B per channel row
constexpr int kChunksPerRow = kRowBytes/16;         // 10 x 16 B

__device__ __forceinline__ float s16_lo(int w) { return (float) (short) (w & 0xFFFF); }
__device__ __forceinline__ float s16_hi(int w) { return (float) (short) (w >> 16); }

// NDMA LDS-DMA instructions: lane copies 16 B from its own global address g[j] to
// lds_base + j*1024 + lane*16.  M0 carries the wave-uniform LDS base; it is
// compiler-reserved, so it is saved and restored inside the same statement (and each
// write of M0 is followed by the one wait state the LDS-DMA read of it needs).
#define SPG_DMA_FIRST   "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
#define SPG_DMA_NEXT(n) "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %" #n ", off\n\t"
#define SPG_DMA_LAST    "s_mov_b32 m0, %0"

template <int NDMA>
__device__ __forceinline__ void dma_issue(const void *const (&g)[NDMA], uint32_t lds_base);

template <>
__device__ __forceinline__ void dma_issue<5>(const void *const (&g)[5], uint32_t lds_base)
{
    uint32_t keep;
    asm volatile(SPG_DMA_FIRST SPG_DMA_NEXT(3) SPG_DMA_NEXT(4) SPG_DMA_NEXT(5) SPG_DMA_NEXT(6) SPG_DMA_LAST
                 : "=&s"(keep)
                 : "s"(lds_base), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4])
                 : "memory", "scc");
}

template <>
__device__ __forceinline__ void dma_issue<10>(const void *const (&g)[10], uint32_t lds_base)
{
    uint32_t keep;
    asm volatile(SPG_DMA_FIRST SPG_DMA_NEXT(3) SPG_DMA_NEXT(4) SPG_DMA_NEXT(5) SPG_DMA_NEXT(6) SPG_DMA_NEXT(7)
                 SPG_DMA_NEXT(8) SPG_DMA_NEXT(9) SPG_DMA_NEXT(10) SPG_DMA_NEXT(11) SPG_DMA_LAST
                 : "=&s"(keep)
                 : "s"(lds_base), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]),
                   "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(g[8]), "v"(g[9])
                 : "memory", "scc");
}

// (the four-lane mapping never issues one: its rows do not divide over the lanes)
template <>
__device__ __forceinline__ void dma_issue<1>(const void *const (&)[1], uint32_t)
{
}

__device__ __forceinline__ void dma_wait_all()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------
// The bank kernel
// ---------------------------------------------------------------------------------
// ABL is a tuning-probe knob (tools/probe.hip) that removes one cost at a time; the
// library only instantiates ABL = 0.
// LDS bytes one workgroup of the bank kernel needs: two segment buffers per wave
template <int LPC>
struct ToneLds
{
    static constexpr int kBufBytes = (kWave/LPC)*kRowBytes;
    // the LPC = 2 / 4 kernels also hold the 256-entry G.711 decode table (the LPC = 1 kernels use every byte of the CU's
    // LDS for their two workgroups, and take linear PCM only)
    static constexpr int kLutBytes = (LPC >= 2)  ?  1024  :  0;
    static constexpr int kBytes = kWavesPerBlock*2*kBufBytes + kLutBytes;
};

// The body of the bank kernel for workgroup `wg` of the bank described by L.  It is a device function so that one
// launch can carry several banks (tone_multi_kernel below): a workgroup belongs to exactly one bank.
template <class Det, int LPC, int ABL = 0>
__device__ __forceinline__ void tone_bank_body(const ToneLaunch &L, const int wg, char *lds_raw)
{
    constexpr int NB = Det::NB;
    constexpr int NBH = (NB + LPC - 1)/LPC;                     // real bins per lane
    constexpr int NBL = (NBH + 1) & ~1;                         // padded to packed pairs
    constexpr int CPW = kWave/LPC;                              // channels per wave
    constexpr int NDMA = (LPC <= 2)  ?  CPW*kChunksPerRow/kWave  :  1;      // LDS-DMA instructions per segment (LPC = 4: no DMA)
    constexpr int kBufBytes = CPW*kRowBytes;
    char (*lds)[2][kBufBytes] = (char (*)[2][kBufBytes]) lds_raw;
    const float *lut = (const float *) (lds_raw + kWavesPerBlock*2*kBufBytes);

    // G.711 input: the decode table, spandsp/g711.h:165-175 (u-law) and :239-252 (A-law), as floats
    const int bps = (LPC >= 2  &&  L.fmt != 0)  ?  1  :  2;     // bytes per sample
    if (LPC >= 2  &&  L.fmt != 0)
    {
        float *wl = (float *) (lds_raw + kWavesPerBlock*2*kBufBytes);
        for (int code = threadIdx.x;  code < 256;  code += kWave*kWavesPerBlock)
        {
            int v;
            if (L.fmt == 2)
            {
                const int u = ~code & 0xFF;
                const int t = (((u & 0x0F) << 3) + 0x84) << ((u & 0x70) >> 4);
                v = (u & 0x80)  ?  (0x84 - t)  :  (t - 0x84);
            }
            else
            {
                const int a = code ^ 0x55;
                int i = (a & 0x0F) << 4;
                const int sg = (a & 0x70) >> 4;
                i = sg  ?  ((i + 0x108) << (sg - 1))  :  (i + 8);
                v = (a & 0x80)  ?  i  :  -i;
            }
            wl[code] = (float) (short) v;
        }
        __syncthreads();
    }

    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ch0 = (wg*kWavesPerBlock + wv)*CPW;
    if (ch0 >= L.n_ch)
        return;                                     // whole wave idle (wave-uniform exit)
    const int cl = (LPC == 1)  ?  lane  :  (lane & (CPW - 1));  // channel within the wave
    const int sub = (LPC == 1)  ?  0  :  (lane/CPW);            // which half / quarter of the bins
    const bool in_bank = (ch0 + cl) < L.n_ch;
    const int ch = in_bank  ?  (ch0 + cl)  :  (L.n_ch - 1);     // shadow lanes follow the last channel, never store
    // A call with per-channel lengths: a channel with no samples in it rides along as a shadow lane too (nothing of
    // it is stored but empty record slots), and a wave with no channel taking part leaves at once.
    int mylen = L.samples;
    if (L.lens)
    {
        mylen = min(max(L.lens[ch], 0), L.samples);
        if (in_bank  &&  sub == 0  &&  mylen == 0)
        {
            for (int b = 0;  b < L.maxb;  b++)
            {
                L.rec[(size_t) b*L.n_ch + ch] = 0;
                if ((ABL & kToneDigits)  &&  L.digits)
                    L.digits[(size_t) b*L.n_ch + ch] = 0;
            }
        }
        if (!__any(in_bank  &&  mylen > 0))
            return;
    }
    const bool live = in_bank  &&  (!L.lens  ||  mylen > 0);    // (a forced block end is a call of no samples)
    const bool store = live  &&  (sub == 0);

    // probe-only (ABL & 32): per-wave timestamps into L.probe_ts, 16 slots per wave
    long long *ts = nullptr;
    if (ABL & 32)
        ts = L.probe_ts + (size_t) (wg*kWavesPerBlock + wv)*16;
    auto stamp = [&](int k)
    {
        if ((ABL & 32)  &&  lane == 0  &&  k < 16)
            ts[k] = (long long) __builtin_readcyclecounter();
    };
    stamp(0);
    // (sixteen rows of ten chunks do not divide over 64 lanes: the four-lane mapping has its rows fetched by their lanes)
    const bool fast_loader = (LPC <= 2)  &&  (L.layout == 0)  &&  L.aligned16  &&  (L.samples > 0);
    const int seg_samples = kRowBytes/bps;                      // a 160-byte row is 80 linear samples or 160 G.711 codes
    const int spc = 16/bps;                                     // samples per 16-byte chunk
    const int nseg = (L.samples + seg_samples - 1)/seg_samples;
    const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) char *) &lds[wv][0][0];

    // Per-lane DMA geometry, fixed for the launch: chunk c = j*64 + lane covers row c/10,
    // 16-byte column c%10 of a segment.  Rows outside the bank follow the last channel and
    // sample offsets past the (8-sample padded) frame re-read its last chunk; neither is
    // ever consumed.
    const int spad = (L.samples + spc - 1) & ~(spc - 1);
    size_t dma_row[NDMA];
    int dma_col8[NDMA];                                          // first sample of the lane's chunk within a segment
#pragma unroll
    for (int j = 0;  j < NDMA;  j++)
    {
        const int c = j*kWave + lane;
        const int r = c/kChunksPerRow;
        dma_col8[j] = (c - r*kChunksPerRow)*spc;
        dma_row[j] = (size_t) min(ch0 + r, L.n_ch - 1)*(size_t) L.stride*bps;
    }
    auto issue_dma = [&](int seg, int buf)
    {
        const void *g[NDMA];
#pragma unroll
        for (int j = 0;  j < NDMA;  j++)
            g[j] = (const char *) L.amp + dma_row[j] + (size_t) (bps*min(seg*seg_samples + dma_col8[j], spad - spc));
        dma_issue<NDMA>(g, __builtin_amdgcn_readfirstlane(lds0 + buf*kBufBytes));
    };
    // The first segment's DMA goes out ahead of the state loads.  hipcc's s_waitcnt bookkeeping does not see the
    // DMA instructions: the wait it places before the first use of the state then covers exactly what the first sample
    // needs (the older DMA and the state), and nothing younger is in flight to be waited for by accident.  The
    // second segment is requested from inside the consume loop of the first.
    if (fast_loader  &&  !(ABL & 16))
        issue_dma(0, 0);

    // ---- load per-channel state (coalesced: SoA, lane == channel) -------------------
    Bank<NBL> bk;
    Det det;
    f32x2 fac[NBL/2];
#pragma unroll
    for (int i = 0;  i < NBL;  i++)
    {
        float f = 0.0f;
        float s2 = 0.0f;
        float s3 = 0.0f;
        if (LPC == 1)
        {
            f = L.fac[i];
            s2 = L.sf[(size_t) i*L.n_ch + ch];
            s3 = L.sf[(size_t) (NB + i)*L.n_ch + ch];
        }
        else if (LPC == 4)
        {
            // lane quarter `sub` owns global bins sub*NBH + i, i < NBH (selects, not an index: L.fac[] sits in scalar registers)
            float fq = 0.0f;
#pragma unroll
            for (int q = 0;  q < 4;  q++)
            {
                const int gq = q*NBH + i;
                const float fv = (i < NBH  &&  gq < NB)  ?  L.fac[(gq < kMaxBins)  ?  gq  :  0]  :  0.0f;
                fq = (sub == q)  ?  fv  :  fq;
            }
            f = fq;
            const int gi = sub*NBH + i;
            if (i < NBH  &&  gi < NB)
            {
                s2 = L.sf[(size_t) gi*L.n_ch + ch];
                s3 = L.sf[(size_t) (NB + gi)*L.n_ch + ch];
            }
        }
        else
        {
            // lane half `sub` owns global bins sub*NBH + i, i < NBH
            const bool real0 = (i < NBH);
            const bool real1 = (i < NBH)  &&  (NBH + i < NB);
            const float f0 = real0  ?  L.fac[(i < kMaxBins)  ?  i  :  0]  :  0.0f;
            const float f1 = real1  ?  L.fac[(NBH + i < kMaxBins)  ?  (NBH + i)  :  0]  :  0.0f;
            f = sub  ?  f1  :  f0;
            const bool real = sub  ?  real1  :  real0;
            if (real)
            {
                const int gi = sub*NBH + i;
                s2 = L.sf[(size_t) gi*L.n_ch + ch];
                s3 = L.sf[(size_t) (NB + gi)*L.n_ch + ch];
            }
        }
        if (i & 1)
            fac[i >> 1].y = f;
        else
            fac[i >> 1].x = f;
        bk.set_v2(i, s2);
        bk.set_v3(i, s3);
    }
    float energy = 0.0f;
    if (Det::kEnergy)
        energy = L.sf[(size_t) (2*NB)*L.n_ch + ch];
    det.load_extra(L, ch);
    uint32_t w0 = (uint32_t) L.si[ch];
    int32_t w1 = L.si[(size_t) L.n_ch + ch];

    int cs = (int) (w0 & 0xFFFF);
    w0 &= 0xFFFF0000u;
    const int block = Det::block_len(L);

    // Wave-uniform block phase?  (true whenever the wave's channels were started
    // together, which the host slot allocator arranges.)
    int cs_first = __builtin_amdgcn_readfirstlane(cs);
    bool uniform = __all(cs == cs_first);
    if (L.lens)
    {
        // the phase the channels taking part share, if they do; the others adopt it for the ride
        const unsigned long long act = __ballot(live);
        cs_first = __builtin_amdgcn_readlane(cs, (int) __ffsll(act) - 1);
        uniform = __all(!live  ||  (cs == cs_first  &&  mylen == L.samples));
        if (uniform)
            cs = cs_first;
    }
    stamp(1);

    int nb = 0;                 // blocks completed by this lane in this call
    int take_acc = 0;           // samples since the last duration update (dtmf.c:202-204)

    auto one_sample = [&](float xin)
    {
        const float x = det.prefilter(xin);
        if (Det::kEnergy  &&  !(ABL & 1))
            energy += x*x;
        if (!(ABL & 8))
            bk.step(fac, x);
        else
            energy += x;
    };
    auto end_block = [&]()
    {
        if (Det::kDuration)
        {
            if (w1 < INT_MAX - take_acc)
                w1 += take_acc;
        }
        take_acc = 0;
        if (ABL & 4)
        {
            nb++;
            return;
        }
        float el[NBL];
        bk.finish(fac, el);
        float e[NB];
        if (LPC == 1)
        {
#pragma unroll
            for (int i = 0;  i < NB;  i++)
                e[i] = el[i];
        }
        else if (LPC == 4)
        {
            // every lane of a channel collects all four quarters
#pragma unroll
            for (int q = 0;  q < 4;  q++)
            {
#pragma unroll
                for (int i = 0;  i < NBH;  i++)
                {
                    if (q*NBH + i < NB)
                        e[q*NBH + i] = __shfl(el[i], cl + q*CPW);
                }
            }
        }
        else
        {
            // exchange with the lane that holds the other half of this channel's bins
#pragma unroll
            for (int i = 0;  i < NBH;  i++)
            {
                const float other = __shfl_xor(el[i], 32);
                e[i] = sub  ?  other  :  el[i];
                if (NBH + i < NB)
                    e[NBH + i] = sub  ?  el[i]  :  other;
            }
        }
        const uint32_t recw = det.decide(L, e, energy, w0, w1, ch, nb, store);
        if (store)
            L.rec[(size_t) nb*L.n_ch + ch] = recw;
        if ((ABL & kToneDigits)  &&  L.digits  &&  store)
            L.digits[(size_t) nb*L.n_ch + ch] = tone_digit_byte<Det::kDuration>(recw);
        nb++;
    };

    for (int seg = 0;  seg < nseg;  seg++)
    {
        const int seg_base = seg*seg_samples;
        const int seglen = min(seg_samples, L.samples - seg_base);
        const int buf = seg & 1;
        char *mybuf = &lds[wv][buf][0];
        // ---- make this segment resident in LDS -------------------------------------------
        if (fast_loader)
        {
            dma_wait_all();
            if (seg == 0)
                stamp(2);
        }
        else if (sub == 0)
        {
            short *wrow = (short *) (mybuf + cl*kRowBytes);
            if (bps == 1)
            {
                const uint8_t *src = (const uint8_t *) L.amp + (size_t) ch*L.stride + seg_base;
                for (int j = 0;  j < seglen;  j++)
                    ((uint8_t *) wrow)[j] = src[j];
            }
            else if (L.layout == 0)
            {
                const int16_t *src = L.amp + (size_t) ch*L.stride + seg_base;
                for (int j = 0;  j < seglen;  j++)
                    wrow[j] = src[j];
            }
            else
            {
                const int16_t *src = L.amp + (size_t) seg_base*L.stride + ch;
                for (int j = 0;  j < seglen;  j++)
                    wrow[j] = src[(size_t) j*L.stride];
            }
        }
        // (single wave per buffer: LDS operations of one wave complete in order, no barrier)
        const int4 *rowv = (const int4 *) (mybuf + cl*kRowBytes);
        const short *row = (const short *) rowv;

        // ---- consume it ------------------------------------------------------------------
        if (uniform  &&  bps == 1)
        {
            // G.711 codes: 16 samples per 16-byte chunk, each through the decode table
            int cs_s = __builtin_amdgcn_readfirstlane(cs);
            const int nchunks = (seglen + 15) >> 4;
            const int nfull = seglen >> 4;
            int4 cur = rowv[0];
            int q = 0;
            bool dma_due = fast_loader  &&  seg + 1 < nseg  &&  !(ABL & 16);
            while (q < nchunks)
            {
                const int nrun = min(nfull - q, (block - cs_s) >> 4);
                for (int k = 0;  k < nrun;  k++)
                {
                    const int4 nxt = rowv[min(q + k + 1, nchunks - 1)];
                    if (dma_due)
                    {
                        issue_dma(seg + 1, buf ^ 1);
                        dma_due = false;
                    }
                    float xs[16];
#pragma unroll
                    for (int b = 0;  b < 16;  b++)
                    {
                        const int w = (b < 4)  ?  cur.x  :  (b < 8)  ?  cur.y  :  (b < 12)  ?  cur.z  :  cur.w;
                        xs[b] = lut[(w >> (8*(b & 3))) & 0xFF];
                    }
#pragma unroll
                    for (int b = 0;  b < 16;  b++)
                        one_sample(xs[b]);
                    cur = nxt;
                }
                if (nrun > 0)
                {
                    q += nrun;
                    cs_s += 16*nrun;
                    take_acc += 16*nrun;
                    if (cs_s == block)
                    {
                        end_block();
                        cs_s = 0;
                    }
                    continue;
                }
                {
                    const int4 nxt = rowv[min(q + 1, nchunks - 1)];
                    if (dma_due)
                    {
                        issue_dma(seg + 1, buf ^ 1);
                        dma_due = false;
                    }
                    const int n = min(16, seglen - q*16);
                    for (int j = 0;  j < n;  j++)
                    {
                        const int w = (j < 4)  ?  cur.x  :  (j < 8)  ?  cur.y  :  (j < 12)  ?  cur.z  :  cur.w;
                        one_sample(lut[(w >> (8*(j & 3))) & 0xFF]);
                        cs_s++;
                        take_acc++;
                        if (cs_s == block)
                        {
                            end_block();
                            cs_s = 0;
                        }
                    }
                    cur = nxt;
                    q++;
                }
            }
            cs = cs_s;
        }
        else if (uniform)
        {
            int cs_s = __builtin_amdgcn_readfirstlane(cs);
            const int nchunks = (seglen + 7) >> 3;
            const int nfull = seglen >> 3;                  // chunks holding 8 samples
            int4 cur = rowv[0];
            int q = 0;
            // Start the next segment's DMA from INSIDE the consume loop: hipcc drains vmcnt(0) in loop preheaders,
            // which would serialise a DMA issued before the loop.
            bool dma_due = fast_loader  &&  seg + 1 < nseg  &&  !(ABL & 16);
            while (q < nchunks)
            {
                // (1) a run of whole chunks that end before the block does: one tight loop, nothing else in it
                const int nrun = min(nfull - q, (block - cs_s) >> 3);
                for (int k = 0;  k < nrun;  k++)
                {
                    const int4 nxt = rowv[min(q + k + 1, nchunks - 1)];     // one chunk ahead
                    if (dma_due)
                    {
                        issue_dma(seg + 1, buf ^ 1);
                        dma_due = false;
                    }
                    if (ABL & 2)
                        cur = make_int4(0x00010002, 0x00030004, 0x00050006, 0x00070008);
                    one_sample(s16_lo(cur.x));
                    one_sample(s16_hi(cur.x));
                    one_sample(s16_lo(cur.y));
                    one_sample(s16_hi(cur.y));
                    one_sample(s16_lo(cur.z));
                    one_sample(s16_hi(cur.z));
                    one_sample(s16_lo(cur.w));
                    one_sample(s16_hi(cur.w));
                    cur = nxt;
                }
                if (nrun > 0)
                {
                    q += nrun;
                    cs_s += 8*nrun;
                    take_acc += 8*nrun;
                    if (cs_s == block)
                    {
                        end_block();
                        cs_s = 0;
                    }
                    continue;
                }
                // (2) the chunk a block boundary (or the end of the call) falls in: sample by sample
                {
                    const int4 nxt = rowv[min(q + 1, nchunks - 1)];
                    if (dma_due)
                    {
                        issue_dma(seg + 1, buf ^ 1);
                        dma_due = false;
                    }
                    const int n = min(8, seglen - q*8);
                    for (int j = 0;  j < n;  j++)
                    {
                        const int w = (j < 2)  ?  cur.x  :  (j < 4)  ?  cur.y  :  (j < 6)  ?  cur.z  :  cur.w;
                        one_sample((float) (short) (w >> ((j & 1)*16)));
                        cs_s++;
                        take_acc++;
                        if (cs_s == block)
                        {
                            end_block();
                            cs_s = 0;
                        }
                    }
                    cur = nxt;
                    q++;
                }
            }
            cs = cs_s;
            stamp(3 + seg);
        }
        else
        {
            // Divergent block phases inside the wave: correct, slower.  (With LPC = 2 / 4 the
            // lanes of a channel share its phase, so they reach end_block() together.)
            for (int pos = 0;  pos < seglen;  pos++)
            {
                if (pos == 0  &&  fast_loader  &&  seg + 1 < nseg)
                    issue_dma(seg + 1, buf ^ 1);
                if (seg_base + pos < mylen)
                {
                    one_sample((bps == 1)  ?  lut[((const uint8_t *) row)[pos]]  :  (float) row[pos]);
                    cs++;
                    take_acc++;
                    if (cs >= block)
                    {
                        end_block();
                        cs = 0;
                    }
                }
            }
        }
    }
    if (L.force_end)
    {
        end_block();
        cs = 0;
    }
    if (Det::kDuration)
    {
        if (take_acc > 0  &&  w1 < INT_MAX - take_acc)
            w1 += take_acc;
    }

    // ---- write back ---------------------------------------------------------------------------
    if (live)
    {
#pragma unroll
        for (int i = 0;  i < NBH;  i++)
        {
            const int gi = sub*NBH + i;
            if (gi < NB)
            {
                L.sf[(size_t) gi*L.n_ch + ch] = bk.v2(i);
                L.sf[(size_t) (NB + gi)*L.n_ch + ch] = bk.v3(i);
            }
        }
    }
    if (store)
    {
        if (Det::kEnergy)
            L.sf[(size_t) (2*NB)*L.n_ch + ch] = energy;
        det.store_extra(L, ch);
        L.si[ch] = (int32_t) (w0 | (uint32_t) cs);
        L.si[(size_t) L.n_ch + ch] = w1;
        for (int b = nb;  b < L.maxb;  b++)
        {
            L.rec[(size_t) b*L.n_ch + ch] = 0;         // slots without a completed block
            if ((ABL & kToneDigits)  &&  L.digits)
                L.digits[(size_t) b*L.n_ch + ch] = 0;
        }
    }
    stamp(15);
}

template <class Det, int LPC, int ABL = 0>
__global__ __launch_bounds__(kWave*kWavesPerBlock)
void tone_bank_kernel(const ToneLaunch L)
{
    __shared__ __attribute__((aligned(16))) char lds_raw[ToneLds<LPC>::kBytes];
    tone_bank_body<Det, LPC, ABL>(L, (int) blockIdx.x, lds_raw);
}

// Several banks in ONE launch: workgroups [first[k], first[k + 1]) belong to bank k.  A 20 ms tick of a mixed
// population (e.g. Bell MF + R2 MF + call-progress banks of ~43 k channels each) then pays launch overhead and the
// ramp-up / drain of the chip once instead of once per bank.  kind[k] selects the detector policy per workgroup
// (wave-uniform branch).
constexpr int kMaxMulti = 4;
enum
{
    TONE_K_DTMF = 0, TONE_K_BELL, TONE_K_R2, TONE_K_ST4, TONE_K_ST8, TONE_K_ST12, TONE_K_ST16
};

struct ToneMultiLaunch
{
    ToneLaunch bank[kMaxMulti];
    int first[kMaxMulti + 1];
    int kind[kMaxMulti];
    int n;
};

template <int LPC>
__global__ __launch_bounds__(kWave*kWavesPerBlock)
void tone_multi_kernel(const ToneMultiLaunch M)
{
    __shared__ __attribute__((aligned(16))) char lds_raw[ToneLds<LPC>::kBytes];
    int k = 0;
    while (k + 1 < M.n  &&  (int) blockIdx.x >= M.first[k + 1])
        k++;
    const int block = (int) blockIdx.x - M.first[k];
    const ToneLaunch &L = M.bank[k];
    switch (M.kind[k])
    {
    case TONE_K_DTMF: tone_bank_body<DtmfDet<false>, LPC>(L, block, lds_raw); break;
    case TONE_K_BELL: tone_bank_body<BellMfDet, LPC>(L, block, lds_raw); break;
    case TONE_K_R2:   tone_bank_body<R2MfDet, LPC>(L, block, lds_raw); break;
    case TONE_K_ST4:  tone_bank_body<MultiDet<4, true>, LPC>(L, block, lds_raw); break;
    case TONE_K_ST8:  tone_bank_body<MultiDet<8, true>, LPC>(L, block, lds_raw); break;
    case TONE_K_ST12: tone_bank_body<MultiDet<12, true>, LPC>(L, block, lds_raw); break;
    default:          tone_bank_body<MultiDet<16, true>, LPC>(L, block, lds_raw); break;
    }
}

}   // namespace spg
