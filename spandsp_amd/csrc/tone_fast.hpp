// tone_fast.hpp -- the streaming Goertzel bank kernel (gfx950, wave64): the production path of
// spangpu_bank_rx() for channel-major frames whose rows are 16-byte aligned (what a slot allocator
// hands out).  Detector policies, the per-lane recurrence and the block-end decisions are the ones of
// tone_dev.hpp; what differs from tone_bank_body there is how a frame reaches the lanes:
//
//   * A frame is cut into 64-byte row pieces ("segments": 32 linear samples or 64 G.711 codes of every
//     channel of the wave).  64 bytes is what 320-byte rows (160-sample frames) are made of -- every piece
//     is exactly one half of a 128-byte memory line, never a straddle.
//   * Each wave owns a ring of R segment slots in LDS (R x 4 KiB at one channel per lane, R x 2 KiB at two
//     lanes per channel).  Segments are copied HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, SGPR base +
//     32-bit VGPR offset: one VGPR of addressing per 1-KiB instruction) and up to R-1 of them are in flight
//     while one is consumed.  The first request of a wave is small (one segment + the channel state),
//     so that when every wave of a bank starts at once the chip has ~150 B per channel to deliver before
//     the first sample is processed, not the half frame the 80-sample stage needed.
//   * The ring is a fraction of the old 2 x 10 KiB stage, so LDS no longer caps a CU at eight waves, and the
//     kernel is written to stay under 128 VGPRs: four (or more) waves per SIMD cover each other's waits.
//   * LDS image of a slot: the four 16-byte chunks of a row piece are stored XOR-swizzled by (row/4)%4, so
//     that the 16 lanes ds_read_b128 serves per cycle hit 16 different bank groups (row-major 64-byte rows
//     would be a 4-way conflict).  The swizzle is applied on the global side of the DMA (which chunk a lane
//     fetches); the LDS side of LDS-DMA is lane-linear by construction.
//
// Reference semantics are unchanged (src/spandsp/tone_detect.h:172-192, src/dtmf.c:164-258 ...): same
// per-sample operation order, so every state word and decision is bit-identical to tone_bank_body's.
#pragma once

#include <type_traits>

#include "tone_dev.hpp"

namespace spg {

constexpr int kPiece = 64;                          // bytes of one row per segment

template <int LPC, int R, bool G711, int WPB>
struct FastLds
{
    static constexpr int kSlot = (kWave/LPC)*kPiece;
    static constexpr int kRing = R*kSlot;
    static constexpr int kLutBytes = G711  ?  1024  :  0;
    static constexpr int kBytes = WPB*kRing + kLutBytes;
};

#define SPG_FDMA_FIRST(nt)   "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %2" nt "\n\t"
#define SPG_FDMA_NEXT(n, nt) "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %" #n ", %2" nt "\n\t"
#define SPG_FDMA_LAST        "s_mov_b32 m0, %0"

// N LDS-DMA instructions: lane copies 16 B from sbase + voff[j] to lds_dst + j*1024 + lane*16.
template <int N, bool NT>
__device__ __forceinline__ void fdma_issue(const uint32_t (&voff)[N], const void *sbase, uint32_t lds_dst);

template <>
__device__ __forceinline__ void fdma_issue<2, false>(const uint32_t (&voff)[2], const void *sbase, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile(SPG_FDMA_FIRST("") SPG_FDMA_NEXT(4, "") SPG_FDMA_LAST
                 : "=&s"(keep) : "s"(lds_dst), "s"(sbase), "v"(voff[0]), "v"(voff[1]) : "memory", "scc");
}

template <>
__device__ __forceinline__ void fdma_issue<2, true>(const uint32_t (&voff)[2], const void *sbase, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile(SPG_FDMA_FIRST(" nt") SPG_FDMA_NEXT(4, " nt") SPG_FDMA_LAST
                 : "=&s"(keep) : "s"(lds_dst), "s"(sbase), "v"(voff[0]), "v"(voff[1]) : "memory", "scc");
}

template <>
__device__ __forceinline__ void fdma_issue<4, false>(const uint32_t (&voff)[4], const void *sbase, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile(SPG_FDMA_FIRST("") SPG_FDMA_NEXT(4, "") SPG_FDMA_NEXT(5, "") SPG_FDMA_NEXT(6, "") SPG_FDMA_LAST
                 : "=&s"(keep) : "s"(lds_dst), "s"(sbase), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3])
                 : "memory", "scc");
}

template <>
__device__ __forceinline__ void fdma_issue<4, true>(const uint32_t (&voff)[4], const void *sbase, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile(SPG_FDMA_FIRST(" nt") SPG_FDMA_NEXT(4, " nt") SPG_FDMA_NEXT(5, " nt") SPG_FDMA_NEXT(6, " nt") SPG_FDMA_LAST
                 : "=&s"(keep) : "s"(lds_dst), "s"(sbase), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3])
                 : "memory", "scc");
}

// Wait until at most K vector-memory loads are outstanding.  Loads return in order, so with K = the number of
// DMA instructions issued after the awaited segment's, its data has landed (stores in between only make the wait
// longer, never shorter).
template <int K>
__device__ __forceinline__ void fdma_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K) : "memory");
}

// Workgroup barrier that LDS reads and LDS-DMA issues are not moved across.
__device__ __forceinline__ void seg_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

#if defined(SPG_TONE_FMA)
#include "tone_pairs_asm_fma.inc"
#else
#include "tone_pairs_asm.inc"
#endif

// Sample pairs [k0, k1) of the sixteen of one row piece through the recurrence, as one asm body that is entered at pair k0
// and left after pair k1 - 1 (tone_pairs_asm.inc, written by tools/gen_pairs_asm.py): the piece that holds a block end is
// taken as [0, p), the block end, [p, 16) -- the same straight-line pair blocks either side, two scalar instructions per pair
// for the exit test, where the rolled loop this replaces cost twice the common segment's time (stamps in
// profiles/r4_probe.log: 6 000 ticks against 2 860).  ad: LDS byte address of the lane's row piece (chunk j at ad ^ 16 j).
template <int NP, bool ENERGY>
__device__ __forceinline__ void pairs_asm(f32x2 (&a)[NP], f32x2 (&b)[NP], float &energy, const f32x2 (&fac)[NP], uint32_t ad, int k0, int k1)
{
    const uint32_t ad1 = ad ^ 16u;
    const uint32_t ad2 = ad ^ 32u;
    const uint32_t ad3 = ad ^ 48u;
#define SPG_PA_ADDR [ad0] "v"(ad), [ad1] "v"(ad1), [ad2] "v"(ad2), [ad3] "v"(ad3), [k0] "s"(k0), [k1] "s"(k1)
    if constexpr (NP == 2)
    {
        if constexpr (ENERGY)
            asm volatile(SPG_PAIRS_ASM_NP2_E1 : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [b0] "+v"(b[0]), [b1] "+v"(b[1]), [en] "+v"(energy)
                         : [f0] "s"(fac[0]), [f1] "s"(fac[1]), SPG_PA_ADDR : SPG_PAIRS_ASM_CLOBBERS);
        else
            asm volatile(SPG_PAIRS_ASM_NP2_E0 : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [b0] "+v"(b[0]), [b1] "+v"(b[1]), [en] "+v"(energy)
                         : [f0] "s"(fac[0]), [f1] "s"(fac[1]), SPG_PA_ADDR : SPG_PAIRS_ASM_CLOBBERS);
    }
    else if constexpr (NP == 3)
    {
        if constexpr (ENERGY)
            asm volatile(SPG_PAIRS_ASM_NP3_E1 : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]), [en] "+v"(energy)
                         : [f0] "s"(fac[0]), [f1] "s"(fac[1]), [f2] "s"(fac[2]), SPG_PA_ADDR : SPG_PAIRS_ASM_CLOBBERS);
        else
            asm volatile(SPG_PAIRS_ASM_NP3_E0 : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]), [en] "+v"(energy)
                         : [f0] "s"(fac[0]), [f1] "s"(fac[1]), [f2] "s"(fac[2]), SPG_PA_ADDR : SPG_PAIRS_ASM_CLOBBERS);
    }
    else if constexpr (NP == 4)
    {
        if constexpr (ENERGY)
            asm volatile(SPG_PAIRS_ASM_NP4_E1 : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]),
                           [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]), [b3] "+v"(b[3]), [en] "+v"(energy)
                         : [f0] "s"(fac[0]), [f1] "s"(fac[1]), [f2] "s"(fac[2]), [f3] "s"(fac[3]), SPG_PA_ADDR : SPG_PAIRS_ASM_CLOBBERS);
        else
            asm volatile(SPG_PAIRS_ASM_NP4_E0 : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]),
                           [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]), [b3] "+v"(b[3]), [en] "+v"(energy)
                         : [f0] "s"(fac[0]), [f1] "s"(fac[1]), [f2] "s"(fac[2]), [f3] "s"(fac[3]), SPG_PA_ADDR : SPG_PAIRS_ASM_CLOBBERS);
    }
    else
    {
        static_assert(NP == 2  ||  NP == 3  ||  NP == 4, "pairs_asm: banks of 4, 6 or 8 bins per lane");
    }
#undef SPG_PA_ADDR
}

// The loader wave of a workgroup (LDR kernels): it issues every LDS-DMA of the workgroup's WPB consumer waves and tells
// them, one barrier per segment, that a segment has landed.  Why a wave of its own: while the chip streams a frame the
// memory pipeline is saturated and a global_load_lds is not accepted until there is room for it -- the issuing wave
// stands still for hundreds of cycles per instruction (measured: the DMA time of a lone wave per SIMD simply adds to
// its compute time).  Here the wave that stands still has nothing else to do.
//   prologue: segments 0 .. R-2 requested;   round s: wait until segment s has landed, barrier s (the consumers have
//   finished segment s-1 when they arrive), request segment s+R-1 into the slot segment s-1 occupied.
// Every round issues NDMA*WPB instructions whatever the number of live consumer waves (rows past the bank re-read its
// last row into slots nobody reads), so that the s_waitcnt counts are compile-time constants.
// (Round 4, measured and dropped: "touches" -- one dword per row requested by the loader two segments ahead, so that the
// DMA of every segment finds its line in the L2; the loader's stamps show every other segment's DMA taking 2 900 - 3 100
// ticks from issue to landed against 800 - 900 for the ones in between.  With the touches all of them landed in 800, and
// the launch took 12.16 us instead of 11.66 (profiles/r4_probe_ab.log).)
template <int LPC, int R, bool G711, bool NT, int WPB, int ABL>
__device__ __forceinline__ void tone_loader(const ToneLaunch &L, const int wg, char *lds_raw)
{
    constexpr int CPW = kWave/LPC;
    constexpr int NDMA = CPW/16;
    constexpr int kSlot = CPW*kPiece;
    constexpr int kRing = R*kSlot;
    constexpr int BPS = G711  ?  1  :  2;
    constexpr int PER_SEG = NDMA*WPB;               // DMA instructions per segment of the workgroup
    static_assert((R - 2)*PER_SEG <= 63, "s_waitcnt vmcnt range");
    const unsigned lane = threadIdx.x & (kWave - 1);
    const unsigned rowbytes = (unsigned) L.samples*BPS;
    const unsigned lim = ((rowbytes + 15u) & ~15u) - 16u;
    const int nseg = (ABL & 256)  ?  0  :  (int) ((rowbytes + kPiece - 1)/kPiece);
    const unsigned stride_b = (unsigned) L.stride*BPS;
    const unsigned wg_ch0 = (unsigned) wg*WPB*CPW;
    const unsigned last = (unsigned) L.n_ch - 1u - wg_ch0;     // last live row of the workgroup (grid is sized by channels: >= 0)
    const char *wbase = (const char *) L.amp + (size_t) wg_ch0*(size_t) stride_b;
    const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) char *) lds_raw;
    uint32_t rowoff[WPB][NDMA];
#pragma unroll
    for (int cw = 0;  cw < WPB;  cw++)
    {
#pragma unroll
        for (int j = 0;  j < NDMA;  j++)
            rowoff[cw][j] = min((unsigned) cw*CPW + 16u*j + (lane >> 2), last)*stride_b;
    }
    const uint32_t g16 = (ABL & 64)  ?  ((lane & 3u) << 4)  :  (((lane & 3u) ^ ((lane >> 4) & 3u)) << 4);
    auto issue = [&](int seg, int slot)
    {
        if (ABL & 16)
            return;
        const uint32_t in_row = min((uint32_t) seg*kPiece + g16, lim);
#pragma unroll
        for (int cw = 0;  cw < WPB;  cw++)
        {
            uint32_t voff[NDMA];
#pragma unroll
            for (int j = 0;  j < NDMA;  j++)
                voff[j] = rowoff[cw][j] + in_row;
            fdma_issue<NDMA, NT>(voff, wbase, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t) cw*kRing + (uint32_t) slot*kSlot));
        }
    };
    // probe-only (ABL & 32): the loader's own stamps, after the consumers' (slot = waves of the bank + workgroup)
    long long *ts = nullptr;
    if (ABL & 32)
        ts = L.probe_ts + ((size_t) ((L.n_ch + CPW - 1)/CPW) + 8 + wg)*16;
    auto stamp = [&](int k)
    {
        if ((ABL & 32)  &&  lane == 0  &&  k < 16)
            ts[k] = (long long) __builtin_readcyclecounter();
    };
    stamp(0);
#pragma unroll
    for (int s = 0;  s < R - 1;  s++)
    {
        if (s < nseg)
            issue(s, s);
    }
    stamp(1);
    int slot_free = R - 1;                          // the slot the next request goes to
    for (int seg = 0;  seg < nseg;  seg++)
    {
        const int later = min(R - 2, nseg - 1 - seg);
        if (R >= 4  &&  later >= 2)
            fdma_wait<2*PER_SEG>();
        else if (R >= 3  &&  later >= 1)
            fdma_wait<PER_SEG>();
        else
            fdma_wait<0>();
        stamp(2 + 2*seg);
        seg_barrier();
        stamp(3 + 2*seg);
        if (seg + R - 1 < nseg)
            issue(seg + R - 1, slot_free);
        slot_free = (slot_free == R - 1)  ?  0  :  (slot_free + 1);
    }
}

// The body of the streaming kernel for workgroup `wg` of the bank described by L.  Preconditions (the host checks
// them and otherwise launches tone_bank_kernel): L.layout == 0, L.aligned16, L.samples > 0, and linear PCM unless
// G711.  ABL is the tuning-probe knob of tools/probe.hip (bit 3: no recurrence, bit 4: no DMA, bit 5: timestamps, bit 10: no state traffic, bit 11: the general block-end decision only); the
// library instantiates ABL = 0 only.
template <class Det, int LPC, int R, bool G711, bool NT, int WPB, int ABL = 0, bool LDR = false>
__device__ __forceinline__ void tone_fast_body(const ToneLaunch &L, const int wg, char *lds_raw)
{
    constexpr int NB = Det::NB;
    constexpr int NBH = (LPC == 1)  ?  NB  :  (NB + 1)/2;      // real bins per lane
    constexpr int NBL = (NBH + 1) & ~1;                         // padded to packed pairs
    constexpr int CPW = kWave/LPC;                              // channels per wave
    constexpr int NDMA = CPW/16;                                // 1-KiB LDS-DMA instructions per segment
    constexpr int kSlot = CPW*kPiece;
    constexpr int kRing = R*kSlot;
    constexpr int BPS = G711  ?  1  :  2;                       // bytes per sample
    constexpr int SPC = 16/BPS;                                 // samples per 16-byte chunk
    static_assert(R >= 2  &&  R <= 4, "ring of 2..4 segment slots");

    const float *lut = (const float *) (lds_raw + WPB*kRing);
    if (G711)
    {
        // the decode table, spandsp/g711.h:165-175 (u-law) and :239-252 (A-law), as floats
        float *wl = (float *) (lds_raw + WPB*kRing);
        for (int code = threadIdx.x;  code < 256;  code += kWave*(WPB + (LDR  ?  1  :  0)))
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

    // Super-tone cadences: the tables the walk at the end of the kernel looks things up in live in LDS (a look-up in global
    // memory there is a full memory latency with nothing to overlap it, and the walk makes two or three in a row).  Round 6: a copy
    // per consumer wave, written by the wave itself from loads it issues WITH its state loads (below) -- the workgroup's one copy
    // was staged at the top of the kernel behind a barrier, and every wave's state loads and first DMA waited a memory round
    // trip for it.
    constexpr int kCadCopies = ((ABL & kToneCadence) != 0)  ?  WPB  :  1;
    __shared__ int32_t cad_first_all[kCadCopies][((ABL & kToneCadence) != 0)  ?  (kCadLdsTones + 1)  :  1];
    __shared__ int4 cad_elem_all[kCadCopies][((ABL & kToneCadence) != 0)  ?  kCadLdsElems  :  1];
    if (ABL & 128)
        return;                                     // probe: launch + dispatch cost alone
    const unsigned lane = threadIdx.x & (kWave - 1);
    const int wv = (WPB == 1  &&  !LDR)  ?  0  :  __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (LDR  &&  wv == WPB)
    {
        tone_loader<LPC, R, G711, NT, WPB, ABL>(L, wg, lds_raw);
        return;
    }
    const int ch0 = (wg*WPB + wv)*CPW;
    if (ch0 >= L.n_ch)
    {
        // whole wave idle (wave-uniform): with a loader wave in the workgroup it still keeps the segment barriers
        if (LDR)
        {
            const int nseg_idle = (ABL & 256)  ?  0  :  (int) (((unsigned) L.samples*BPS + kPiece - 1)/kPiece);
            for (int seg = 0;  seg < nseg_idle;  seg++)
                seg_barrier();
        }
        return;
    }
    const unsigned cl = (LPC == 1)  ?  lane  :  (lane & (CPW - 1));     // channel within the wave
    const int sub = (LPC == 1)  ?  0  :  (int) (lane >> 5);             // which half of the bins
    const bool in_bank = (ch0 + (int) cl) < L.n_ch;
    const unsigned ch = in_bank  ?  (unsigned) ch0 + cl  :  (unsigned) (L.n_ch - 1);   // shadow lanes follow the last channel, never store
    const unsigned nch = (unsigned) L.n_ch;
    // A call with an active mask (L.lens holds 0 or `samples` per channel; other lengths go to the general kernel): a
    // channel sitting the call out rides along as a shadow lane -- nothing of it is stored but empty record slots --
    // and a wave with no channel taking part only keeps the workgroup's barriers.
    bool live = in_bank;
    if (L.lens)
    {
        live = in_bank  &&  (L.lens[ch] > 0);
        if (in_bank  &&  !live  &&  sub == 0)
        {
            for (int b = 0;  b < L.maxb;  b++)
            {
                L.rec[(size_t) b*nch + ch] = 0;
                if ((ABL & kToneDigits)  &&  L.digits)
                    L.digits[(size_t) b*nch + ch] = 0;
            }
        }
        if (!__any(live))
        {
            if (LDR)
            {
                const int nseg_idle = (int) (((unsigned) L.samples*BPS + kPiece - 1)/kPiece);
                for (int seg = 0;  seg < nseg_idle;  seg++)
                    seg_barrier();
            }
            return;
        }
    }
    const bool store = live  &&  (sub == 0);

    long long *ts = nullptr;
    if (ABL & 32)
        ts = L.probe_ts + (size_t) (wg*WPB + wv)*16;
    auto stamp = [&](int k)
    {
        if ((ABL & 32)  &&  lane == 0  &&  k < 16)
            ts[k] = (long long) __builtin_readcyclecounter();
    };
    stamp(0);

    // ---- frame geometry (wave-uniform) -------------------------------------------------------------
    const unsigned rowbytes = (unsigned) L.samples*BPS;
    const unsigned lim = ((rowbytes + 15u) & ~15u) - 16u;       // last 16-byte chunk of a (padded) row
    const int nseg = (ABL & 256)  ?  0  :  (int) ((rowbytes + kPiece - 1)/kPiece);      // probe bit 8: state in, state out, nothing else
    const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) char *) lds_raw + (uint32_t) wv*kRing;
    const unsigned stride_b = (unsigned) L.stride*BPS;
    const char *wbase = (const char *) L.amp + (size_t) ch0*(size_t) stride_b;      // this wave's first row (SGPR pair)

    // Per-lane DMA geometry.  Instruction j covers rows 16j .. 16j+15 of the wave, four lanes per row; lane i
    // fetches chunk (i & 3) ^ swz of its row, swz = (row/4) & 3 = (i >> 4) & 3, and the data lands lane-linearly.
    uint32_t rowoff[NDMA];
#pragma unroll
    for (int j = 0;  j < NDMA;  j++)
    {
        const unsigned r = min((unsigned) ch0 + 16u*j + (lane >> 2), nch - 1u) - (unsigned) ch0;   // rows past the bank re-read the last one
        rowoff[j] = r*stride_b;
    }
    const uint32_t g16 = (ABL & 64)  ?  ((lane & 3u) << 4)  :  (((lane & 3u) ^ ((lane >> 4) & 3u)) << 4);
    auto issue_dma = [&](int seg, int slot)
    {
        if ((ABL & 16)  ||  LDR)                    // with a loader wave, the consumers issue nothing
            return;
        uint32_t voff[NDMA];
        const uint32_t in_row = min((uint32_t) seg*kPiece + g16, lim);      // chunks past the frame re-read its last one
#pragma unroll
        for (int j = 0;  j < NDMA;  j++)
            voff[j] = rowoff[j] + in_row;
        fdma_issue<NDMA, NT>(voff, wbase, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t) slot*kSlot));
    };
    // Reader side: chunk k of this lane's row piece sits at (rd0 + slot*kSlot) ^ (k << 4).
    const uint32_t rd0 = (uint32_t) wv*kRing + (cl >> 4)*1024u + (cl & 15u)*64u + ((ABL & 64)  ?  0u  :  (((cl >> 2) & 3u) << 4));

    // The first segment's request goes out ahead of the state loads (hipcc's s_waitcnt bookkeeping does not see
    // the DMA: issued after them, the compiler's wait for the state would also cover whatever DMA is younger).
    issue_dma(0, 0);

    // ---- load per-channel state (coalesced: SoA, lane == channel; SGPR base + 32-bit lane offset, so that no
    //      per-lane 64-bit address is formed, let alone kept for the write-back) -----------------------------
    const unsigned ch4 = ch*4u;
    auto ldf = [&](const float *base) -> float { return *(const float *) ((const char *) base + ch4); };
    auto ldi = [&](const int32_t *base) -> int32_t { return *(const int32_t *) ((const char *) base + ch4); };
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
            if (!(ABL & 1024))
            {
                s2 = ldf(L.sf + (size_t) i*nch);
                s3 = ldf(L.sf + (size_t) (NB + i)*nch);
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
                s2 = ldf(L.sf + (size_t) gi*nch);
                s3 = ldf(L.sf + (size_t) (NB + gi)*nch);
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
    if (Det::kEnergy  &&  !(ABL & 1024))
        energy = ldf(L.sf + (size_t) (2*NB)*nch);
    det.load_extra(L, (int) ch);
    CadenceRegs cad_regs;
    int32_t cad_tf = 0;
    int4 cad_te0 = make_int4(0, 0, 0, 0);
    int4 cad_te1 = make_int4(0, 0, 0, 0);
    if constexpr ((ABL & kToneCadence) != 0)
    {
        if (L.cad.state)
        {
            cadence_state_load(L.cad, (int) ch, (int) nch, cad_regs);     // needed a frame from now: the latency costs nothing here
            // ... and this wave's copy of the tables: lane l asks for first[l] and elements l and l + 64 (the host only builds this
            // variant into a launch when the tables fit: kCadLdsTones, kCadLdsElems)
            static_assert(kCadLdsTones + 1 <= kWave  &&  kCadLdsElems <= 2*kWave, "a wave's lanes cover the cadence tables");
            if ((int) lane <= L.cad.n_tones)
                cad_tf = L.cad.first[lane];
            if ((int) lane < L.cad.n_elems)
                cad_te0 = L.cad.elem[lane];
            if ((int) lane + kWave < L.cad.n_elems)
                cad_te1 = L.cad.elem[lane + kWave];
        }
    }
    uint32_t w0 = (ABL & 1024)  ?  0u  :  (uint32_t) ldi(L.si);
    int32_t w1 = (ABL & 1024)  ?  0  :  ldi(L.si + (size_t) nch);

    int cs = (int) (w0 & 0xFFFF);
    w0 &= 0xFFFF0000u;
    const int block = Det::block_len(L);
    int cs_first = __builtin_amdgcn_readfirstlane(cs);
    bool uniform = __all(cs == cs_first);           // true whenever the wave's channels were started together
    if (L.lens)
    {
        // the phase the channels taking part share, if they do; the others adopt it for the ride
        const unsigned long long act = __ballot(live);
        cs_first = __builtin_amdgcn_readlane(cs, (int) __ffsll(act) - 1);
        uniform = __all(!live  ||  cs == cs_first);
        if (uniform)
            cs = cs_first;
    }
    int32_t (&cad_first)[sizeof(cad_first_all[0])/sizeof(int32_t)] = cad_first_all[kCadCopies > 1  ?  wv  :  0];
    int4 (&cad_elem)[sizeof(cad_elem_all[0])/sizeof(int4)] = cad_elem_all[kCadCopies > 1  ?  wv  :  0];
    if constexpr ((ABL & kToneCadence) != 0)
    {
        if (L.cad.state)
        {
            // (the loads came back with the state's; a wave's own LDS operations are performed in order: no barrier)
            if ((int) lane <= kCadLdsTones)
                cad_first[lane] = cad_tf;
            cad_elem[lane] = cad_te0;
            if ((int) lane + kWave < kCadLdsElems)
                cad_elem[lane + kWave] = cad_te1;
        }
    }
    stamp(1);

    int nb = 0;                 // blocks completed by this lane in this call
    int take_acc = 0;           // samples since the last duration update (dtmf.c:202-204)
    uint32_t rec0 = 0;          // record words of this call's first two blocks (0 = no block completed in the slot)
    uint32_t rec1 = 0;

    auto one_sample = [&](float xin)
    {
        const float x = det.prefilter(xin);
        if (Det::kEnergy)
            energy += x*x;
        if (!(ABL & 8))
            bk.template step1<LPC == 1>(fac, x);
        else
            energy += x;
    };
    // NPAIR consecutive sample pairs; get(k) yields pair k as floats (k is a compile-time constant after unrolling).
    // The few instructions of a pair that are not the recurrence are dealt around the recurrence block so that no
    // dependent two of them are neighbours: before block k the converts of pair k + 1 and the first energy add of
    // pair k, after it the second add and the squares of pair k + 1 (energy += x*x per sample, dtmf.c:199,
    // super_tone_rx.c:467: the adds stay in sample order).
    auto run_pairs = [&](auto npair_tag, auto get)
    {
        constexpr int NPAIR = decltype(npair_tag)::value;
        f32x2 x = get(0);
        if (Det::kFilter)
        {
            x.x = det.prefilter(x.x);
            x.y = det.prefilter(x.y);
        }
        f32x2 sq = x*x;
#pragma unroll
        for (int k = 0;  k < NPAIR;  k++)
        {
            f32x2 xn = x;
            if (k + 1 < NPAIR)
            {
                xn = get(k + 1);
                if (Det::kFilter)
                {
                    xn.x = det.prefilter(xn.x);
                    xn.y = det.prefilter(xn.y);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (Det::kEnergy)
                energy += sq.x;
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 8))
                bk.template step2<LPC == 1>(fac, x);
            else
                energy += x.x + x.y;
            __builtin_amdgcn_sched_barrier(0);
            if (Det::kEnergy)
            {
                energy += sq.y;
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < NPAIR)
                    sq = xn*xn;
                __builtin_amdgcn_sched_barrier(0);
            }
            x = xn;
        }
    };
    // pair k (samples 2k, 2k + 1) of a run of 16-byte chunks
    auto pair_from = [&](const int4 &c, int k) -> f32x2
    {
        f32x2 x01;
        if (G711)
        {
            const int j = 2*k;
            const int w = (j < 4)  ?  c.x  :  (j < 8)  ?  c.y  :  (j < 12)  ?  c.z  :  c.w;
            x01.x = lut[(w >> (8*(j & 3))) & 0xFF];
            x01.y = lut[(w >> (8*((j + 1) & 3))) & 0xFF];
        }
        else
        {
            const int w = (k == 0)  ?  c.x  :  (k == 1)  ?  c.y  :  (k == 2)  ?  c.z  :  c.w;
            x01.x = s16_lo(w);
            x01.y = s16_hi(w);
        }
        return x01;
    };
    constexpr int PPC = SPC/2;                      // sample pairs per chunk
    // (probe bit 17: the rolled loop for every segment with a block end, as before round 4)
    constexpr bool kPairsAsm = !G711  &&  !Det::kFilter  &&  LPC == 1  &&  NBL/2 >= 2  &&  NBL/2 <= 4  &&  !(ABL & 131072);
    // the production case of a block end (wave-uniform): nothing but the decision itself is asked for, with the
    // bank's own thresholds -- Det::decide_plain() then does what Det::decide() does in a third of the instructions
    const bool plain = Det::kLean  &&  !L.trace  &&  !L.chan_parms  &&  !L.rec_energy  &&  !L.realtime;
    auto end_block = [&](auto uni_tag)
    {
        constexpr bool UNI = decltype(uni_tag)::value;      // every lane of the wave is here, with the same nb
        if (Det::kDuration)
        {
            if (w1 < INT_MAX - take_acc)
                w1 += take_acc;
        }
        take_acc = 0;
        float el[NBL];
        f32x2 esum;
        bk.finish(fac, el, &esum);
        float e[NB];
        if (LPC == 1)
        {
#pragma unroll
            for (int i = 0;  i < NB;  i++)
                e[i] = el[i];
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
        uint32_t recw;
        bool lean = false;
        if constexpr (Det::kLean  &&  LPC == 1  &&  !(ABL & 2048))
            lean = plain  &&  __all((esum.x + esum.y) == (esum.x + esum.y));     // a NaN anywhere, or infinities of both signs, make the sum a NaN
        if (lean)
        {
            if constexpr (Det::kLean)
                recw = det.decide_plain(L, e, energy, w0, w1);
        }
        else
        {
            recw = det.decide(L, e, energy, w0, w1, (int) ch, nb, store);
        }
        // The record words of the first two blocks of a call (all there are in a 160-sample frame) wait in registers
        // for the write-back: a store here would stand in the memory pipeline's queue with the sample loop behind it.
        if constexpr (UNI)
        {
            const int nbu = __builtin_amdgcn_readfirstlane(nb);
            if (nbu == 0)
                rec0 = recw;
            else if (nbu == 1)
                rec1 = recw;
            else if (store)
                *(uint32_t *) ((char *) (L.rec + (size_t) nbu*nch) + ch4) = recw;
        }
        else
        {
            if (nb == 0)
                rec0 = recw;
            else if (nb == 1)
                rec1 = recw;
            else if (store)
                *(uint32_t *) ((char *) (L.rec + (size_t) nb*nch) + ch4) = recw;
        }
        if ((ABL & kToneDigits)  &&  L.digits  &&  store)
            L.digits[(size_t) nb*nch + ch] = tone_digit_byte<Det::kDuration>(recw);
        nb++;
    };
    auto chunk_at = [&](uint32_t a) -> int4
    {
        return *(const int4 *) (lds_raw + a);
    };
    int pos = 0;                                    // samples of the frame consumed so far (wave-uniform)
    int cs_s = cs_first;
    int slot = 0;                                   // ring slot of the current segment
    for (int seg = 0;  seg < nseg;  seg++)
    {
        // ---- make this segment resident; keep the ring full ------------------------------------------
        if (LDR)
        {
            seg_barrier();                          // the loader has seen this segment land; everyone is done with the last one
            if (seg == 0)
                stamp(2);
        }
        else if (seg == 0)
        {
            fdma_wait<0>();
            stamp(2);
#pragma unroll
            for (int s = 1;  s < R;  s++)
            {
                if (s < nseg)
                    issue_dma(s, s);
            }
        }
        else
        {
            const int later = min(R - 2, nseg - 1 - seg);       // segments in flight behind this one
            if (R >= 4  &&  later >= 2)
                fdma_wait<2*NDMA>();
            else if (R >= 3  &&  later >= 1)
                fdma_wait<NDMA>();
            else
                fdma_wait<0>();
            if (seg + R - 1 < nseg)
                issue_dma(seg + R - 1, (slot == 0)  ?  (R - 1)  :  (slot - 1));    // into the slot consumed last
        }
        const uint32_t rd = rd0 + (uint32_t) slot*kSlot;
        const int seglen = min(4*SPC, L.samples - pos);

        if (uniform  &&  seglen == 4*SPC  &&  block - cs_s >= 4*SPC)
        {
            // the common segment: whole, and no block ends inside it -- straight-line code, all four chunk reads up front
            const int4 c0 = chunk_at(rd);
            const int4 c1 = chunk_at(rd ^ 16u);
            const int4 c2 = chunk_at(rd ^ 32u);
            const int4 c3 = chunk_at(rd ^ 48u);
            run_pairs(std::integral_constant<int, 4*PPC>(), [&](int k)
            {
                return pair_from((k < PPC)  ?  c0  :  (k < 2*PPC)  ?  c1  :  (k < 3*PPC)  ?  c2  :  c3, k%PPC);
            });
            cs_s += 4*SPC;
            take_acc += 4*SPC;
            if (cs_s == block)
            {
                end_block(std::true_type());
                cs_s = 0;
            }
            cs = cs_s;
        }
        else if (uniform  &&  kPairsAsm  &&  seglen == 4*SPC  &&  block >= 4*SPC  &&  ((block - cs_s) & 1) == 0)
        {
            // A whole segment with a block end inside it, on a pair boundary: pairs [0, p), the block end, pairs [p, 16) --
            // both through the one asm body (pairs_asm above)
            if constexpr (kPairsAsm)
            {
                const int m = block - cs_s;
                const uint32_t ad = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) char *) lds_raw + rd;
                float no_energy = 0.0f;
#pragma nounroll
                for (int part = 0;  part < 2;  part++)
                {
                    const int k0 = __builtin_amdgcn_readfirstlane(part  ?  (m >> 1)  :  0);
                    const int k1 = __builtin_amdgcn_readfirstlane(part  ?  4*PPC  :  (m >> 1));
                    pairs_asm<NBL/2, Det::kEnergy>(bk.a, bk.b, Det::kEnergy  ?  energy  :  no_energy, fac, ad, k0, k1);
                    if (part == 0)
                    {
                        take_acc += m;
                        end_block(std::true_type());
                    }
                }
                cs_s = 4*SPC - m;
                take_acc += 4*SPC - m;
                cs = cs_s;
            }
        }
        else if (uniform)
        {
            // A segment with a block end inside, or the short last one: taken in pieces that end where the block or the
            // segment does.  A piece is a tight rolled loop over its sample pairs (nothing but the in-place recurrence
            // block, the converts and the energy adds in it; dwords fetched from the slot two rounds ahead), with a
            // single sample before and after it where a block end splits a pair.  Block ends are handled in one
            // place, after a piece.  Pair p of the lane's row piece is the dword (linear PCM) or half dword (G.711)
            // at rd ^ (d << 2): d << 2 = 16*(chunk) + 4*(dword in chunk), and rd is 16-byte aligned below the swizzle.
            constexpr int PPD = 2/BPS;                  // sample pairs per dword
            auto pair_dword = [&](int pidx) -> int
            {
                return *(const int *) (lds_raw + (rd ^ ((uint32_t) (pidx/PPD) << 2)));
            };
            auto pair_floats = [&](int wd, int pidx) -> f32x2
            {
                f32x2 x01;
                if (G711)
                {
                    const int h = wd >> (16*(pidx & 1));
                    x01.x = lut[h & 0xFF];
                    x01.y = lut[(h >> 8) & 0xFF];
                }
                else
                {
                    x01.x = s16_lo(wd);
                    x01.y = s16_hi(wd);
                }
                if (Det::kFilter)
                {
                    x01.x = det.prefilter(x01.x);
                    x01.y = det.prefilter(x01.y);
                }
                return x01;
            };
            auto lone_sample = [&](int sidx) -> float
            {
                const uint32_t a = (rd ^ ((uint32_t) (sidx*BPS/4) << 2)) + (uint32_t) ((sidx*BPS) & 3);
                if (G711)
                    return lut[*(const uint8_t *) (lds_raw + a)];
                return (float) *(const short *) (lds_raw + a);
            };
            const int lastp = (seglen - 1) >> 1;        // last pair index there is data for
            int k = 0;                                  // samples of the segment taken so far
            while (k < seglen)
            {
                const int m = min(block - cs_s, seglen - k);        // samples of this piece (>= 1)
                int kk = k;
                int left = m;
                if (kk & 1)
                {
                    one_sample(lone_sample(kk));
                    kk++;
                    left--;
                }
                const int np = left >> 1;
                if (np > 0)
                {
                    const int p0 = kk >> 1;
                    f32x2 x = pair_floats(pair_dword(p0), p0);
                    f32x2 sq = x*x;
                    int wn = pair_dword(min(p0 + 1, lastp));
                    for (int i = 0;  i < np;  i++)
                    {
                        const int wnn = pair_dword(min(p0 + i + 2, lastp));
                        if (Det::kEnergy)
                            energy += sq.x;
                        __builtin_amdgcn_sched_barrier(0);
                        if (!(ABL & 8))
                            bk.template step2<LPC == 1>(fac, x);
                        else
                            energy += x.x + x.y;
                        __builtin_amdgcn_sched_barrier(0);
                        if (Det::kEnergy)
                            energy += sq.y;
                        if (!Det::kFilter  ||  i + 1 < np)      // (the input filter has state: no look-ahead past the piece)
                            x = pair_floats(wn, p0 + i + 1);
                        sq = x*x;
                        wn = wnn;
                    }
                    kk += 2*np;
                    left -= 2*np;
                }
                if (left > 0)
                    one_sample(lone_sample(kk));
                k += m;
                cs_s += m;
                take_acc += m;
                if (cs_s == block)
                {
                    end_block(std::true_type());
                    cs_s = 0;
                }
            }
            cs = cs_s;
        }
        else
        {
            // Divergent block phases inside the wave: correct, slower.  (With LPC = 2 the two lanes of a channel
            // share its phase, so they reach end_block() together.)
            for (int p = 0;  p < seglen;  p++)
            {
                const uint32_t a = (rd ^ ((uint32_t) (p/SPC) << 4)) + (uint32_t) (p%SPC)*BPS;
                float x;
                if (G711)
                    x = lut[*(const uint8_t *) (lds_raw + a)];
                else
                    x = (float) *(const short *) (lds_raw + a);
                one_sample(x);
                cs++;
                take_acc++;
                if (cs >= block)
                {
                    end_block(std::false_type());
                    cs = 0;
                }
            }
        }
        pos += seglen;
        slot = (slot == R - 1)  ?  0  :  (slot + 1);
        stamp(3 + seg);
    }
    if (L.force_end)
    {
        end_block(std::false_type());
        cs = 0;
    }
    if (Det::kDuration)
    {
        if (take_acc > 0  &&  w1 < INT_MAX - take_acc)
            w1 += take_acc;
    }

    // ---- write back -----------------------------------------------------------------------------------
    unsigned st4 = ch4;
    asm volatile("" : "+v"(st4));                   // a fresh value: the offsets of the loads are not kept alive for this
    // The state goes back with write-through stores (sc0 sc1): nothing of it is left dirty in the L2 for the end of the kernel
    // to flush, and the next launch reads it from the memory side either way.  (Not inline asm: a scalar base that hipcc has
    // spilled comes back through v_readlane_b32, and a vector memory instruction it cannot see gets none of the five wait
    // states that needs -- found by the 12-bin banks, whose state rows went to the wrong addresses.)  Measured against plain stores
    // (profiles/r4_probe_ab.log): 11.34 -> 11.23 us at 65 536 channels, 19.3 -> 18.5 at 131 072, 119 -> 117 at 1 M.
    // (Probe bits 14 / 15 / 16: non-temporal stores / no stores at all / plain stores.)
    auto stf = [&](float *base, float v)
    {
        if (ABL & 32768)
            return;
        if (ABL & 16384)
            __builtin_nontemporal_store(v, (float *) ((char *) base + st4));
        else if (ABL & 65536)
            *(float *) ((char *) base + st4) = v;
        else        // (a relaxed atomic store of system scope is what hipcc writes as global_store_dword ... sc0 sc1)
            __hip_atomic_store((uint32_t *) ((char *) base + st4), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto sti = [&](int32_t *base, int32_t v)
    {
        if (ABL & 32768)
            return;
        if (ABL & 16384)
            __builtin_nontemporal_store(v, (int32_t *) ((char *) base + st4));
        else if (ABL & 65536)
            *(int32_t *) ((char *) base + st4) = v;
        else
            __hip_atomic_store((int32_t *) ((char *) base + st4), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    bool keep = live;
    if (ABL & 1024)
    {
        // probe bit 10: no state traffic -- zero state in, and the results kept alive by a store that never happens
        float acc = energy;
#pragma unroll
        for (int i = 0;  i < NBH;  i++)
            acc += bk.v2(i) + bk.v3(i);
        keep = live  &&  (acc == 1.2345e-30f);
    }
    if (keep)
    {
#pragma unroll
        for (int i = 0;  i < NBH;  i++)
        {
            const int gi = sub*NBH + i;
            if (gi < NB)
            {
                stf(L.sf + (size_t) gi*nch, bk.v2(i));
                stf(L.sf + (size_t) (NB + gi)*nch, bk.v3(i));
            }
        }
    }
    if (store  &&  keep)
    {
        if (Det::kEnergy)
            stf(L.sf + (size_t) (2*NB)*nch, energy);
        det.store_extra(L, (int) ch);
        sti(L.si, (int32_t) (w0 | (uint32_t) cs));
        sti(L.si + (size_t) nch, w1);
        // (probe bit 18: the two record rows not stored -- what they cost the end of a launch, tools/probe2 rec)
        if (L.maxb > 0  &&  !(ABL & 262144))
            sti((int32_t *) L.rec, (int32_t) rec0);
        if (L.maxb > 1  &&  !(ABL & 262144))
            sti((int32_t *) L.rec + (size_t) nch, (int32_t) rec1);
        if ((ABL & 262144)  &&  (rec0 ^ rec1) == 0x12345678u)
            sti((int32_t *) L.rec, (int32_t) rec0);        // (keeps the values alive)
        for (int b = max(nb, 2);  b < L.maxb;  b++)
            *(int32_t *) ((char *) ((int32_t *) L.rec + (size_t) b*nch) + st4) = 0;    // slots without a completed block
        if ((ABL & kToneDigits)  &&  L.digits)
        {
            for (int b = nb;  b < L.maxb;  b++)
                L.digits[(size_t) b*nch + ch] = 0;
        }
    }
    if constexpr ((ABL & kToneCadence) != 0)
    {
        // Super-tone cadences: the lane that made a channel's records walks them here, before the wave ends -- a launch of
        // its own for this costs 10 us at 65 536 channels, nearly all of it latency that overlaps nothing.
        if (L.cad.state)
        {
            if (L.maxb > 2)
                __threadfence_block();          // records past the second were stored by end_block(): read them back whole
            // (the compact list is made on demand by cadence_list_kernel: an atomic per wave on its counter, a thousand of
            // them on one address, took longer here than the whole walk)
            (void) cadence_walk_loaded(L.cad, cad_first, cad_elem, cad_regs, (int) ch, (int) nch, L.maxb, rec0, rec1, L.rec, store  &&  keep);
        }
    }
    stamp(15);
}

// The first sixteen dwords of the argument block (frame pointer and geometry, state and record pointers) are also
// passed as leading scalar arguments: built with -mllvm -amdgpu-kernarg-preload-count=16 they arrive in SGPRs with the
// wave, and the state loads and the first DMA can be addressed without waiting for a scalar load of the kernarg
// segment first.
template <class Det, int LPC, int R, bool G711, bool NT, int WPB, int ABL = 0, bool LDR = false>
__global__ __launch_bounds__(kWave*(WPB + (LDR  ?  1  :  0))) __attribute__((amdgpu_waves_per_eu(LDR  ?  2  :  3)))
void tone_fast_kernel(const int16_t *amp, long long stride, int samples, int n_ch, int wg0, int aligned16,
                      float *sf, int32_t *si, uint32_t *rec, const ToneLaunch L0)
{
    __shared__ __attribute__((aligned(1024))) char lds_raw[FastLds<LPC, R, G711, WPB>::kBytes];
    ToneLaunch L = L0;
    L.amp = amp;
    L.stride = stride;
    L.samples = samples;
    L.n_ch = n_ch;
    L.layout = 0;               // (a precondition of this kernel; its slot among the preloaded arguments carries wg0)
    L.aligned16 = aligned16;
    L.sf = sf;
    L.si = si;
    L.rec = rec;
    // (wg0: the first workgroup of the bank this launch covers -- a bank in queue mode is advanced by two launches)
    tone_fast_body<Det, LPC, R, G711, NT, WPB, ABL, LDR>(L, (int) blockIdx.x + wg0, lds_raw);
}

// Host-side launch of tone_fast_kernel (argument order above).
template <class Det, int LPC, int R, bool G711, bool NT, int WPB, int ABL = 0, bool LDR = false>
static inline void launch_tone_fast(const ToneLaunch &L, int blocks, hipStream_t st)
{
    hipLaunchKernelGGL((tone_fast_kernel<Det, LPC, R, G711, NT, WPB, ABL, LDR>), dim3(blocks), dim3(kWave*(WPB + (LDR  ?  1  :  0))), 0, st,
                       L.amp, L.stride, L.samples, L.n_ch, L.wg0, L.aligned16, L.sf, L.si, L.rec, L);
}

// Several banks in ONE launch (see tone_multi_kernel in tone_dev.hpp): workgroups [first[k], first[k + 1]) belong to
// bank k, the detector policy is chosen per workgroup.
template <int LPC, int R, bool LDR>
__global__ __launch_bounds__(kWave*(4 + (LDR  ?  1  :  0))) __attribute__((amdgpu_waves_per_eu(LDR  ?  2  :  3)))
void tone_multi_fast_kernel(const ToneMultiLaunch M)
{
    __shared__ __attribute__((aligned(1024))) char lds_raw[FastLds<LPC, R, false, 4>::kBytes];
    int k = 0;
    while (k + 1 < M.n  &&  (int) blockIdx.x >= M.first[k + 1])
        k++;
    const int block = (int) blockIdx.x - M.first[k];
    const ToneLaunch &L = M.bank[k];
    switch (M.kind[k])
    {
    case TONE_K_DTMF: tone_fast_body<DtmfDet<false>, LPC, R, false, false, 4, 0, LDR>(L, block, lds_raw); break;
    case TONE_K_BELL: tone_fast_body<BellMfDet, LPC, R, false, false, 4, 0, LDR>(L, block, lds_raw); break;
    case TONE_K_R2:   tone_fast_body<R2MfDet, LPC, R, false, false, 4, 0, LDR>(L, block, lds_raw); break;
    case TONE_K_ST4:  tone_fast_body<MultiDet<4, true>, LPC, R, false, false, 4, 0, LDR>(L, block, lds_raw); break;
    case TONE_K_ST8:  tone_fast_body<MultiDet<8, true>, LPC, R, false, false, 4, 0, LDR>(L, block, lds_raw); break;
    case TONE_K_ST12: tone_fast_body<MultiDet<12, true>, LPC, R, false, false, 4, 0, LDR>(L, block, lds_raw); break;
    default:          tone_fast_body<MultiDet<16, true>, LPC, R, false, false, 4, 0, LDR>(L, block, lds_raw); break;
    }
}

}   // namespace spg
