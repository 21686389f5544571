// mct_dev.hpp -- device side of the modem connect tone detector banks:
// N detectors of one tone type, one channel per lane.
//
// What is restated (paths relative to the reference tree; float build, x86-64, <tgmath.h> in use):
//   modem_connect_tones_rx()   src/modem_connect_tones.c:521-785   every tone type
//   v21_put_bit()              src/modem_connect_tones.c:437-518   HDLC flag hunt on the V.21 bit stream
//   report_tone_state()        src/modem_connect_tones.c:416-435
//   fsk_rx()                   through fsk_dev.hpp (V.21 channel 2, synchronous, cutoff -45.5 dBm0)
//
// The notch / band-pass recurrences are binary32 evaluated left to right (the library is built with
// -ffp-contract=off); lfastrintf() is a truncating cast on x86-64.  A report's level needs log10f(): the
// kernel records the integer it is computed from and the host applies libm's log10f (mct_api.hip), as
// the DTMF shim does for its levels.
//
// For MODEM_CONNECT_TONES_FAX_CED_OR_PREAMBLE the reference runs fsk_rx() over the whole buffer and then
// the 2100 Hz detector over the whole buffer, both reporting through the same tone_present; the kernel
// keeps that order (so, like the reference, the outcome can depend on where the caller cuts its frames).

#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fsk_dev.hpp"

namespace spg
{

enum
{
    MC_TONE_TYPE = 0,
    MC_ZNOTCH_1 = 1,
    MC_ZNOTCH_2 = 2,
    MC_Z15HZ_1 = 3,
    MC_Z15HZ_2 = 4,
    MC_NOTCH_LEVEL = 5,
    MC_CHANNEL_LEVEL = 6,
    MC_AM_LEVEL = 7,
    MC_TONE_PRESENT = 8,
    MC_TONE_ON = 9,
    MC_CYCLE_DURATION = 10,
    MC_GOOD_CYCLES = 11,
    MC_HIT = 12,
    MC_RAW_BITS = 13,
    MC_NUM_BITS = 14,
    MC_FLAGS_SEEN = 15,
    MC_FRAMING_OK = 16,
    MC_PAD = 17,
    kMctWords = 18
};

enum
{
    MCT_NONE = 0, MCT_FAX_CNG = 1, MCT_ANS = 2, MCT_ANS_PR = 3, MCT_ANSAM = 4, MCT_ANSAM_PR = 5, MCT_FAX_PREAMBLE = 6,
    MCT_FAX_CED_OR_PREAMBLE = 7, MCT_BELL_ANS = 8, MCT_CALLING_TONE = 9
};

constexpr int kMctV21Span = 26;     // 8000*100/30000

struct MctLaunch
{
    int32_t *st;                // [kMctWords (+ kFskScalars + 4*26)][n_ch]
    const int16_t *pcm;
    const int16_t *quarter;
    int32_t *events;            // [n_ch][ev_cap][2]: tone, the integer the level is computed from
    int32_t *ev_count;
    long long stride;
    int n_ch;
    int samples;
    const int32_t *lens;        // nullptr, or samples per channel in this call (<= samples; 0 = the channel sits it out)
    int ev_cap;
    int vec;
    int latch;                  // no callback installed: reports also set `hit` (modem_connect_tones.c:426-429)
};

struct MctRegs
{
    float znotch_1, znotch_2, z15hz_1, z15hz_2;
    int32_t notch_level, channel_level, am_level;
    int32_t tone_present, tone_on, cycle, good_cycles, hit;
    uint32_t raw_bits;
    int32_t num_bits, flags_seen, framing_ok;
    int32_t n_ev;
};

__device__ __forceinline__ void mct_load(MctRegs &m, const int32_t *st, size_t n)
{
    m.znotch_1 = __int_as_float(st[MC_ZNOTCH_1*n]);
    m.znotch_2 = __int_as_float(st[MC_ZNOTCH_2*n]);
    m.z15hz_1 = __int_as_float(st[MC_Z15HZ_1*n]);
    m.z15hz_2 = __int_as_float(st[MC_Z15HZ_2*n]);
    m.notch_level = st[MC_NOTCH_LEVEL*n];
    m.channel_level = st[MC_CHANNEL_LEVEL*n];
    m.am_level = st[MC_AM_LEVEL*n];
    m.tone_present = st[MC_TONE_PRESENT*n];
    m.tone_on = st[MC_TONE_ON*n];
    m.cycle = st[MC_CYCLE_DURATION*n];
    m.good_cycles = st[MC_GOOD_CYCLES*n];
    m.hit = st[MC_HIT*n];
    m.raw_bits = (uint32_t) st[MC_RAW_BITS*n];
    m.num_bits = st[MC_NUM_BITS*n];
    m.flags_seen = st[MC_FLAGS_SEEN*n];
    m.framing_ok = st[MC_FRAMING_OK*n];
    m.n_ev = 0;
}

// The words the 2100 Hz / CNG / ... detector owns, the ones the V.21 flag hunt owns, and the two both report through
__device__ __forceinline__ void mct_store_tone(const MctRegs &m, int32_t *st, size_t n)
{
    st[MC_ZNOTCH_1*n] = __float_as_int(m.znotch_1);
    st[MC_ZNOTCH_2*n] = __float_as_int(m.znotch_2);
    st[MC_Z15HZ_1*n] = __float_as_int(m.z15hz_1);
    st[MC_Z15HZ_2*n] = __float_as_int(m.z15hz_2);
    st[MC_NOTCH_LEVEL*n] = m.notch_level;
    st[MC_CHANNEL_LEVEL*n] = m.channel_level;
    st[MC_AM_LEVEL*n] = m.am_level;
    st[MC_TONE_ON*n] = m.tone_on;
    st[MC_CYCLE_DURATION*n] = m.cycle;
    st[MC_GOOD_CYCLES*n] = m.good_cycles;
}

__device__ __forceinline__ void mct_store_hdlc(const MctRegs &m, int32_t *st, size_t n)
{
    st[MC_RAW_BITS*n] = (int32_t) m.raw_bits;
    st[MC_NUM_BITS*n] = m.num_bits;
    st[MC_FLAGS_SEEN*n] = m.flags_seen;
    st[MC_FRAMING_OK*n] = m.framing_ok;
}

__device__ __forceinline__ void mct_store_present(const MctRegs &m, int32_t *st, size_t n)
{
    st[MC_TONE_PRESENT*n] = m.tone_present;
    st[MC_HIT*n] = m.hit;
}

// v21_put_bit(), modem_connect_tones.c:437-518; report(tone, from) stands for report_tone_state()
template <class Report>
__device__ __forceinline__ void mct_put_bit(MctRegs &m, const FskRegs &r, int bit, Report &&report)
{
    if (bit < 0)
    {
        if (bit == -1  &&  m.tone_present == MCT_FAX_PREAMBLE)
            report(MCT_NONE, 0);
        m.raw_bits = 0;
        m.num_bits = 0;
        m.flags_seen = 0;
        m.framing_ok = 0;
        return;
    }
    m.raw_bits = (m.raw_bits << 1) | ((uint32_t) (bit << 8) & 0x100u);
    m.num_bits++;
    if ((m.raw_bits & 0x7F00u) == 0x7E00u)
    {
        if (m.raw_bits & 0x8000u)
        {
            m.flags_seen = 0;           // HDLC abort
        }
        else if (m.flags_seen < 5)
        {
            if (m.num_bits != 8)
                m.flags_seen = 0;
            if (++m.flags_seen >= 5  &&  !m.framing_ok)
            {
                report(MCT_FAX_PREAMBLE, r.power);      // lfastrintf(fsk_rx_signal_power()) on the host
                m.framing_ok = 1;
            }
        }
        m.num_bits = 0;
    }
    else if (m.flags_seen >= 5  &&  m.num_bits == 8)
    {
        m.framing_ok = 0;
        m.flags_seen = 0;
    }
}

// The V.21 receiver over a channel's frame with v21_put_bit() as its bit sink
template <bool VEC, class Report>
__device__ __forceinline__ void mct_v21_frame(MctRegs &m, FskRegs &r, int32_t *win, const uint32_t *wave, int lane, const int16_t *row,
                                              int mylen, int samples, Report &&report)
{
    auto put_bit = [&](int bit) __attribute__((always_inline))
    {
        mct_put_bit(m, r, bit, report);
    };
    FskRow<VEC> rw;
    fsk_row_begin(rw, row, mylen);
    for (int base = 0;  base < samples;  base += 8)
    {
        const int todo = max(0, min(8, mylen - base));
        int32_t a[8];
        int32_t c0[8];
        int32_t q0[8];
        int32_t c1[8];
        int32_t q1[8];
        fsk_row_block(rw, base, todo, a);
        fsk_block_lookups(r, wave, todo, c0, q0, c1, q1);
        if (__builtin_expect(__all(todo == 8), 1))
        {
#pragma unroll
            for (int k = 0;  k < 8;  k++)
                fsk_step<false>(r, win, lane, kMctV21Span, a[k], c0[k], q0[k], c1[k], q1[k], put_bit);
        }
        else
        {
#pragma unroll
            for (int k = 0;  k < 8;  k++)
            {
                if (k < todo)
                    fsk_step<false>(r, win, lane, kMctV21Span, a[k], c0[k], q0[k], c1[k], q1[k], put_bit);
            }
        }
    }
}

// The tone detector of type TYPE over a channel's frame, modem_connect_tones.c:531-785
template <int TYPE, bool VEC, class Report>
__device__ __forceinline__ void mct_tone_frame(MctRegs &m, const int16_t *row, int mylen, int samples, Report &&report)
{
    // notch section: v1 = g*x + a1*z1 - a2*z2;  y = v1 + b1*z1 + z2
    float g = 0.0f;
    float a1 = 0.0f;
    float a2 = 0.0f;
    float b1 = 0.0f;
    if (TYPE == MCT_FAX_CNG)
    {
        // 1100 Hz, modem_connect_tones.c:536-540
        g = 0.792928f;  a1 = 1.0018744927985f;  a2 = 0.54196833412465f;  b1 = -1.2994747954630f;
    }
    else if (TYPE == MCT_BELL_ANS)
    {
        // 2225 Hz, modem_connect_tones.c:700-704
        g = 0.739651f;  a1 = -0.257384f;  a2 = 0.510404f;  b1 = 0.351437f;
    }
    else if (TYPE == MCT_CALLING_TONE)
    {
        // 1300 Hz, modem_connect_tones.c:754-761
        g = 0.755582f;  a1 = 0.820887174515f;  a2 = 0.541968324778f;  b1 = -1.0456667108f;
    }
    else
    {
        // 2100 Hz, modem_connect_tones.c:607-611
        g = 0.7552f;  a1 = -0.1183852f;  a2 = 0.5104039f;  b1 = 0.1567596f;
    }
    FskRow<VEC> rw;
    fsk_row_begin(rw, row, mylen);
    for (int base = 0;  base < samples;  base += 8)
    {
        const int todo = max(0, min(8, mylen - base));
        int32_t a[8];
        fsk_row_block(rw, base, todo, a);
#pragma unroll
        for (int k = 0;  k < 8;  k++)
        {
            if (k >= todo)
                continue;
            const int32_t s = a[k];
            const float famp = (float) s;
            const int32_t mag = (s < 0)  ?  -s  :  s;
            if (TYPE == MCT_ANS  ||  TYPE == MCT_FAX_CED_OR_PREAMBLE)
            {
                // the 15 Hz AM detector, modem_connect_tones.c:593-601
                const float v15 = fabsf(famp) + 1.996667f*m.z15hz_1 - 0.9968004f*m.z15hz_2;
                const float filtered = 0.001599787f*(v15 - m.z15hz_2);
                m.z15hz_2 = m.z15hz_1;
                m.z15hz_1 = v15;
                const int32_t fi = (int32_t) filtered;
                m.am_level += ((fi < 0)  ?  -fi  :  fi) - (m.am_level >> 8);
            }
            const float v1 = g*famp + a1*m.znotch_1 - a2*m.znotch_2;
            const float y = v1 + b1*m.znotch_1 + m.znotch_2;
            m.znotch_2 = m.znotch_1;
            m.znotch_1 = v1;
            const int32_t notched = (int32_t) (int16_t) (int32_t) y;
            const int32_t nmag = (notched < 0)  ?  -notched  :  notched;
            m.channel_level += (mag - m.channel_level) >> 5;
            if (TYPE == MCT_ANS  ||  TYPE == MCT_FAX_CED_OR_PREAMBLE)
            {
                // modem_connect_tones.c:620-690
                m.notch_level += (nmag - m.notch_level) >> 4;
                if (m.channel_level <= 70)
                {
                    if (m.tone_present != MCT_NONE)
                        report(MCT_NONE, 0);
                    m.cycle = 0;
                    m.good_cycles = 0;
                    m.tone_on = 0;
                    continue;
                }
                m.cycle++;
                const bool am = (m.am_level*15/256 > m.channel_level);
                if (m.notch_level*6 < m.channel_level)
                {
                    if (!m.tone_on)
                    {
                        if (m.cycle >= 8*(450 - 25))
                        {
                            if (++m.good_cycles == 3)
                                report(am  ?  MCT_ANSAM_PR  :  MCT_ANS_PR, m.channel_level);
                        }
                        else
                        {
                            m.good_cycles = 0;
                        }
                        m.cycle = 0;
                    }
                    else if (m.cycle >= 8*(450 + 100))
                    {
                        if (m.tone_present == MCT_NONE)
                            report(am  ?  MCT_ANSAM  :  MCT_ANS, m.channel_level);
                        m.good_cycles = 0;
                        m.cycle = 8*(450 + 100);
                    }
                    m.tone_on = 1;
                }
                else if (m.notch_level*5 > m.channel_level)
                {
                    if (m.tone_present == MCT_ANS)
                    {
                        report(MCT_NONE, 0);
                        m.good_cycles = 0;
                    }
                    else if (m.cycle >= 8*(450 + 25))
                    {
                        if (m.tone_present == MCT_ANS_PR  ||  m.tone_present == MCT_ANSAM_PR)
                            report(MCT_NONE, 0);
                        m.good_cycles = 0;
                    }
                    m.tone_on = 0;
                }
            }
            else
            {
                // CNG / Bell answer / calling tone, modem_connect_tones.c:545-577,711-739,765-781
                m.notch_level += (nmag - m.notch_level) >> 5;
                if (m.channel_level > 70  &&  m.notch_level*6 < m.channel_level)
                {
                    if (m.tone_present != TYPE)
                    {
                        if (++m.cycle >= 8*415)
                            report(TYPE, m.channel_level);
                    }
                }
                else
                {
                    if (m.tone_present == TYPE)
                        report(MCT_NONE, 0);
                    m.cycle = 0;
                }
            }
        }
    }
}

template <int TYPE>
__global__ __launch_bounds__(64) void mct_bank_kernel(const MctLaunch L)
{
    constexpr bool kFsk = (TYPE == MCT_FAX_PREAMBLE  ||  TYPE == MCT_FAX_CED_OR_PREAMBLE);
    extern __shared__ int32_t win[];        // [4*26][64] when the V.21 receiver runs
    __shared__ uint32_t wave[kFsk  ?  kFskWave  :  1];
    const int lane = threadIdx.x;
    const int ch = blockIdx.x*64 + lane;
    const bool live = ch < L.n_ch;
    const size_t n = (size_t) L.n_ch;
    int32_t *st = L.st + (live  ?  ch  :  0);
    int32_t *fst = st + (size_t) kMctWords*n;

    if (kFsk)
    {
        fsk_fill_wave(wave, L.quarter, lane, 64);
        fsk_load_window(win, fst + (size_t) kFskScalars*n, n, kMctV21Span, lane);
        __syncthreads();
    }
    if (!live)
        return;
    const int mylen = L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;       // per lane with per-channel lengths

    MctRegs m;
    mct_load(m, st, n);
    int32_t *ev = L.events + (size_t) ch*L.ev_cap*2;
    const int ev_cap = L.ev_cap;
    const bool latch = (L.latch != 0);
    // report_tone_state(), modem_connect_tones.c:416-435; `from` is what the host turns into the level
    auto report = [&](int tone, int from) __attribute__((always_inline))
    {
        if (tone == m.tone_present)
            return;
        if (m.n_ev < ev_cap)
        {
            ev[2*m.n_ev] = tone;
            ev[2*m.n_ev + 1] = from;
        }
        m.n_ev++;
        if (latch  &&  tone != MCT_NONE)
            m.hit = tone;
        m.tone_present = tone;
    };
    const int16_t *row = L.pcm + (size_t) ch*L.stride;

    if (kFsk)
    {
        FskRegs r;
        fsk_load_regs(r, fst, n);
        if (L.vec)
            mct_v21_frame<true>(m, r, win, wave, lane, row, mylen, L.samples, report);
        else
            mct_v21_frame<false>(m, r, win, wave, lane, row, mylen, L.samples, report);
        fsk_store_regs(r, fst, n);
        fsk_store_window(win, fst + (size_t) kFskScalars*n, n, kMctV21Span, lane);
        mct_store_hdlc(m, st, n);
    }
    if (TYPE != MCT_FAX_PREAMBLE)
    {
        if (L.vec)
            mct_tone_frame<TYPE, true>(m, row, mylen, L.samples, report);
        else
            mct_tone_frame<TYPE, false>(m, row, mylen, L.samples, report);
        mct_store_tone(m, st, n);
    }
    mct_store_present(m, st, n);
    L.ev_count[ch] = m.n_ev;
}

// MODEM_CONNECT_TONES_FAX_CED_OR_PREAMBLE with its two machines on two waves of a workgroup, on the same 64 channels:
// wave 0 runs the V.21 receiver and its flag hunt, wave 1 the 2100 Hz detector.  The reference runs them one after the
// other and they meet in two words only: both report through tone_present (and latch `hit`), the V.21 side first.  So
// wave 1 works on the assumption that the V.21 side reports nothing in this frame -- true for all but a few frames of
// a call -- and keeps what it would report itself in registers.  When the two have met at the barrier, a lane whose
// V.21 side did report, or whose detector had more to report than the registers hold, has its detector put back and
// run again from the V.21 side's outcome with its reports written directly (the whole wave walks along: rare).
constexpr int kMctHeld = 2;             // reports wave 1 holds per lane and frame

__global__ __launch_bounds__(128) void mct_ced_pair_kernel(const MctLaunch L)
{
    constexpr int TYPE = MCT_FAX_CED_OR_PREAMBLE;
    extern __shared__ int32_t win[];        // [4*26][64], then the meeting place [3][64]
    __shared__ uint32_t wave[kFskWave];
    const int lane = threadIdx.x & 63;
    const int side = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int ch = blockIdx.x*64 + lane;
    const bool live = ch < L.n_ch;
    const size_t n = (size_t) L.n_ch;
    int32_t *st = L.st + (live  ?  ch  :  0);
    int32_t *fst = st + (size_t) kMctWords*n;
    int32_t *meet = win + 4*kMctV21Span*64;

    fsk_fill_wave(wave, L.quarter, threadIdx.x, 128);
    if (side == 0)
        fsk_load_window(win, fst + (size_t) kFskScalars*n, n, kMctV21Span, lane);
    __syncthreads();
    const int mylen = !live  ?  0  :  L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;
    MctRegs m;
    mct_load(m, st, n);
    int32_t *ev = L.events + (size_t) (live  ?  ch  :  0)*L.ev_cap*2;
    const int ev_cap = L.ev_cap;
    const bool latch = (L.latch != 0);
    const int16_t *row = L.pcm + (size_t) (live  ?  ch  :  0)*L.stride;
    auto report = [&](int tone, int from) __attribute__((always_inline))
    {
        if (tone == m.tone_present)
            return;
        if (m.n_ev < ev_cap)
        {
            ev[2*m.n_ev] = tone;
            ev[2*m.n_ev + 1] = from;
        }
        m.n_ev++;
        if (latch  &&  tone != MCT_NONE)
            m.hit = tone;
        m.tone_present = tone;
    };

    if (side == 0)
    {
        FskRegs r;
        fsk_load_regs(r, fst, n);
        if (L.vec)
            mct_v21_frame<true>(m, r, win, wave, lane, row, mylen, L.samples, report);
        else
            mct_v21_frame<false>(m, r, win, wave, lane, row, mylen, L.samples, report);
        meet[lane] = m.n_ev;
        meet[64 + lane] = m.tone_present;
        meet[128 + lane] = m.hit;
        __syncthreads();
        if (live)
        {
            fsk_store_regs(r, fst, n);
            fsk_store_window(win, fst + (size_t) kFskScalars*n, n, kMctV21Span, lane);
            mct_store_hdlc(m, st, n);
        }
    }
    else
    {
        const MctRegs m0 = m;
        int32_t held[2*kMctHeld];
#pragma unroll
        for (int i = 0;  i < 2*kMctHeld;  i++)
            held[i] = 0;
        auto hold = [&](int tone, int from) __attribute__((always_inline))
        {
            if (tone == m.tone_present)
                return;
#pragma unroll
            for (int i = 0;  i < kMctHeld;  i++)
            {
                if (m.n_ev == i)
                {
                    held[2*i] = tone;
                    held[2*i + 1] = from;
                }
            }
            m.n_ev++;
            if (latch  &&  tone != MCT_NONE)
                m.hit = tone;
            m.tone_present = tone;
        };
        if (L.vec)
            mct_tone_frame<TYPE, true>(m, row, mylen, L.samples, hold);
        else
            mct_tone_frame<TYPE, false>(m, row, mylen, L.samples, hold);
        __syncthreads();
        const int v21_ev = meet[lane];
        const bool again = live  &&  (v21_ev > 0  ||  m.n_ev > kMctHeld);
        if (__builtin_expect(__any(again), 0))
        {
            if (again)
            {
                m = m0;
                m.n_ev = v21_ev;
                m.tone_present = meet[64 + lane];
                m.hit = meet[128 + lane];
                if (L.vec)
                    mct_tone_frame<TYPE, true>(m, row, mylen, L.samples, report);
                else
                    mct_tone_frame<TYPE, false>(m, row, mylen, L.samples, report);
            }
        }
        if (live)
        {
            if (!again)
            {
#pragma unroll
                for (int i = 0;  i < kMctHeld;  i++)
                {
                    if (i < m.n_ev  &&  i < ev_cap)
                    {
                        ev[2*i] = held[2*i];
                        ev[2*i + 1] = held[2*i + 1];
                    }
                }
            }
            mct_store_tone(m, st, n);
            mct_store_present(m, st, n);
            L.ev_count[ch] = m.n_ev;
        }
    }
}

}   // namespace spg
