// sigtone_dev.hpp -- device side of the in-band signalling tone banks (SURVEY.md section 8(f)-4 names sig_tone.c beside
// v18.c and ademco_contactid.c): N receivers, or N senders, of one tone type, one channel per lane.
//
// What is restated (paths relative to the reference tree; float build, x86-64):
//   sig_tone_rx()              src/sig_tone.c:402-663   notch filters (two cascaded bi-quads per tone), the flat mode
//                                                       bi-quad, four leaky power meters, the sharp / flat detectors with
//                                                       their persistence checks, the notch insertion logic, and the media
//                                                       path: the frame is rewritten in place (muted, passed, or notched)
//   sig_tone_tx()              src/sig_tone.c:246-323   silence or pass-through plus one or two DDS tones, high level first
//   sig_tone_tx_set_mode()     src/sig_tone.c:326-345
//   the descriptors            src/sig_tone.c:77-244
//   power_meter_update()       src/power_meter.c:65-69;  dds_mod() / dds_lookup()  src/dds_int.c:340-387
//
// The bi-quads are binary32 sums in the order written (the library is built with -ffp-contract=off, and so is this
// file): v = (x*a0 + z0*b1) + z1*b2;  v += (z0*a1 + z1*a2).  A float handed to power_meter_update() is truncated to int
// and then to 16 bits.  A receiver's report is the tone callback's (signalling_state, 0, duration): the kernel records
// it with the sample it happened at, and the host replays the callbacks in order.
//
// The sender's update request is a callback from inside sig_tone_tx() in which the caller sets the next mode.  A kernel
// cannot call out: a channel whose timeout runs out stops where the reference would call back, says so, and is taken up
// again from that sample by the next launch, after the host has had its callback (sigtone_api.hip).

#pragma once

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <type_traits>

namespace spg
{

enum
{
    // receiver: tone j at 5j .. 5j+4
    SG_Z10 = 0, SG_Z11 = 1, SG_Z20 = 2, SG_Z21 = 3, SG_POWER = 4,
    SG_FLAT_Z0 = 15,
    SG_FLAT_Z1 = 16,
    SG_FLAT_POWER = 17,
    SG_PERSISTENCE = 18,
    SG_LAST_PRESENT = 19,
    SG_FLAT_MODE = 20,
    SG_FLAT_TIMEOUT = 21,
    SG_NOTCH_TIMEOUT = 22,
    SG_STATE = 23,
    SG_DURATION = 24,
    SG_NOTCH_FILTER = 25,
    SG_RX_TONE = 26,
    kSigRxWords = 27
};

enum
{
    SX_PHASE0 = 0,
    SX_PHASE1 = 1,
    SX_HIGH_LOW = 2,
    SX_TONE = 3,
    SX_TIMEOUT = 4,
    kSigTxWords = 5
};

enum
{
    SIG_1_PRESENT = 0x001, SIG_1_CHANGE = 0x002, SIG_2_PRESENT = 0x004, SIG_2_CHANGE = 0x008,
    SIG_TX_PASSTHROUGH = 0x010, SIG_RX_PASSTHROUGH = 0x040, SIG_RX_FILTER_TONE = 0x080
};

struct SigRxLaunch
{
    int32_t *st;                // [kSigRxWords][n_ch]
    int16_t *pcm;               // rewritten in place
    int32_t *events;            // [n_ch][ev_cap][3]: sample of the call, signalling_state, duration
    int32_t *ev_count;
    long long stride;
    int n_ch;
    int samples;
    const int32_t *lens;        // nullptr, or samples per channel in this call (<= samples; 0 = the channel sits it out)
    int ev_cap;
    int vec;                    // rows are 16-byte aligned
    int32_t flat_threshold;     // power_meter_level_dbm0() of the descriptor's thresholds, and the detection ratio
    int32_t sharp_threshold;
    int32_t detection_ratio;
};

struct SigNotch
{
    float a10, a11, a12, b11, b12, a21, a22, b21, b22;
};

// sig_tone.c:77-121 (float branch); which: 0 = 2280 Hz, 1 = 2400 Hz, 2 = 2600 Hz
template <int WHICH>
__device__ __forceinline__ SigNotch sig_notch()
{
    if (WHICH == 0)
        return {0.878906f, 0.439362f, 1.0f, -0.287627f, -0.883605f, 0.433228f, 1.0f, -0.530792f, -0.883605f};
    if (WHICH == 1)
        return {0.862000f, 0.612055f, 1.0f, -0.456264f, -0.864899f, 0.621021f, 1.0f, -0.690738f, -0.864899f};
    return {0.862000f, 0.902374f, 1.0f, -0.732727f, -0.864899f, 0.910766f, 1.0f, -0.952393f, -0.864899f};
}

struct SigTone
{
    float z10, z11, z20, z21;
    int32_t power;
};

// one sample through the two cascaded bi-quads of a notch, sig_tone.c:459-474
__device__ __forceinline__ float sig_notch_step(const SigNotch &c, SigTone &t, float signal)
{
    float v = signal*c.a10 + t.z10*c.b11 + t.z11*c.b12;
    float x = v;
    v += t.z10*c.a11 + t.z11*c.a12;
    t.z11 = t.z10;
    t.z10 = x;
    v += t.z20*c.b21 + t.z21*c.b22;
    x = v;
    v += t.z20*c.a21 + t.z21*c.a22;
    t.z21 = t.z20;
    t.z20 = x;
    return v;
}

// power_meter_update() with damping 5 on a value handed over as int16_t
__device__ __forceinline__ int32_t sig_meter(int32_t &reading, int32_t amp16)
{
    reading += ((amp16*amp16 - reading) >> 5);
    return reading;
}

__device__ __forceinline__ int32_t sig_to_i16(float v)
{
    return (int32_t) (int16_t) (int32_t) v;
}

// fsaturatef(), saturated.h:142-149 (lrintf: to nearest, ties to even)
__device__ __forceinline__ int32_t sig_fsat(float v)
{
    if (v > 32767.0f)
        return 32767;
    if (v < -32768.0f)
        return -32768;
    return (int32_t) rintf(v);
}

// TYPE: 1 = 2280 Hz, 2 = 2600 Hz, 3 = 2400 Hz / 2600 Hz (sig_tone.h:57-64)
template <int TYPE>
__global__ __launch_bounds__(64) void sigtone_rx_kernel(const SigRxLaunch L)
{
    constexpr int NT = (TYPE == 3)  ?  3  :  1;                // notch filters run per sample (sig_tone.c:427-431)
    constexpr int K0 = (TYPE == 1)  ?  0  :  (TYPE == 2)  ?  2  :  1;      // desc->notch[0]
    constexpr int K1 = 2;                                       // desc->notch[1] of the two-tone type
    constexpr bool kFlat = (TYPE == 1);                         // the type with a flat mode filter and a sharp -> flat timeout
    constexpr int kSharpFlat = (TYPE == 1)  ?  225*8  :  0;
    constexpr int kNotchLag = 225*8;
    constexpr int kOnCheck = 3*8;
    constexpr int kOffCheck = 8*8;
    const int lane = threadIdx.x;
    const int ch = blockIdx.x*64 + lane;
    if (ch >= L.n_ch)
        return;
    const size_t n = (size_t) L.n_ch;
    int32_t *st = L.st + ch;
    const int mylen = L.lens  ?  min(max(L.lens[ch], 0), L.samples)  :  L.samples;
    if (mylen == 0)
    {
        L.ev_count[ch] = 0;
        return;
    }

    SigTone t[NT];
#pragma unroll
    for (int j = 0;  j < NT;  j++)
    {
        t[j].z10 = __int_as_float(st[(size_t) (5*j + SG_Z10)*n]);
        t[j].z11 = __int_as_float(st[(size_t) (5*j + SG_Z11)*n]);
        t[j].z20 = __int_as_float(st[(size_t) (5*j + SG_Z20)*n]);
        t[j].z21 = __int_as_float(st[(size_t) (5*j + SG_Z21)*n]);
        t[j].power = st[(size_t) (5*j + SG_POWER)*n];
    }
    float flat_z0 = 0.0f;
    float flat_z1 = 0.0f;
    if (kFlat)
    {
        flat_z0 = __int_as_float(st[(size_t) SG_FLAT_Z0*n]);
        flat_z1 = __int_as_float(st[(size_t) SG_FLAT_Z1*n]);
    }
    int32_t flat_power = st[(size_t) SG_FLAT_POWER*n];
    int32_t persistence = st[(size_t) SG_PERSISTENCE*n];
    int32_t last_present = st[(size_t) SG_LAST_PRESENT*n];
    int32_t flat_mode = st[(size_t) SG_FLAT_MODE*n];
    int32_t flat_timeout = st[(size_t) SG_FLAT_TIMEOUT*n];
    int32_t notch_timeout = st[(size_t) SG_NOTCH_TIMEOUT*n];
    int32_t state = st[(size_t) SG_STATE*n];
    int32_t duration = st[(size_t) SG_DURATION*n];
    int32_t notch_filter = st[(size_t) SG_NOTCH_FILTER*n];
    const int32_t rx_tone = st[(size_t) SG_RX_TONE*n];

    const SigNotch c0 = sig_notch<K0>();
    const SigNotch c1 = sig_notch<K1>();
    int32_t *ev = L.events + (size_t) ch*L.ev_cap*3;
    int n_ev = 0;
    int16_t *row = L.pcm + (size_t) ch*L.stride;

    // The frame goes through the lane in chunks of eight samples (one 16-byte access each way); the next chunk is requested
    // before the current one is worked on.  The per-sample logic is written as selects, not branches: the lanes of a wave
    // are different lines in different signalling states, and a branch taken by any of them is paid by all.
    int32_t a[8];
    auto sample = [&](int k, int at) __attribute__((always_inline))
    {
        const float famp = (float) a[k];
        duration += (duration < INT_MAX)  ?  1  :  0;
        // ---- the notch filters and their power meters, sig_tone.c:437-487 ----
        float notched1 = 0.0f;
        float notched2 = 0.0f;
        int32_t np1 = INT_MAX;
        int32_t np2 = INT_MAX;
        const float notched0 = sig_notch_step(c0, t[0], famp);
        const int32_t np0 = sig_meter(t[0].power, sig_to_i16(notched0));
        if constexpr (NT == 3)
        {
            notched1 = sig_notch_step(c1, t[1], famp);
            np1 = sig_meter(t[1].power, sig_to_i16(notched1));
            notched2 = sig_notch_step(c0, t[2], notched1);
            np2 = sig_meter(t[2].power, sig_to_i16(notched2));
        }
        // ---- sharp or flat, sig_tone.c:488-499 ----
        const bool present = (state & (SIG_1_PRESENT | SIG_2_PRESENT)) != 0;
        bool flat = false;
        float band = famp;
        if constexpr (kFlat)
        {
            const bool tick = present  &&  (flat_timeout != 0);
            const int32_t ft = flat_timeout - (tick  ?  1  :  0);
            flat_mode = present  ?  ((tick  &&  ft == 0)  ?  1  :  flat_mode)  :  0;
            flat_timeout = present  ?  ft  :  kSharpFlat;
            flat = (flat_mode != 0);
            // the flat mode bi-quad, sig_tone.c:507-528: it only runs (its state only moves) in flat mode
            float v = famp*0.393676f + flat_z0*-0.261778f + flat_z1*-0.359985f;
            const float x = v;
            v += flat_z0*-0.5f + flat_z1*-0.5f;
            band = v;
            flat_z1 = flat  ?  flat_z0  :  flat_z1;
            flat_z0 = flat  ?  x  :  flat_z0;
        }
        const int32_t fp = sig_meter(flat_power, flat  ?  sig_to_i16(band)  :  a[k]);
        // ---- flat mode, sig_tone.c:530-561: a plain power threshold ----
        int32_t st_flat = state;
        int32_t nt_flat = notch_timeout;
        if constexpr (kFlat)
        {
            st_flat = present  ?  ((fp < L.flat_threshold)  ?  ((state & ~SIG_1_PRESENT) | SIG_1_CHANGE)  :  state)
                               :  ((fp > L.flat_threshold)  ?  (state | SIG_1_PRESENT | SIG_1_CHANGE)  :  state);
            nt_flat = (st_flat & (SIG_1_PRESENT | SIG_2_PRESENT))  ?  kNotchLag  :  (notch_timeout - ((notch_timeout != 0)  ?  1  :  0));
        }
        // ---- sharp mode, sig_tone.c:563-625: notched against total power, then the persistence checks ----
        const int m = (np0 < np1)  ?  0  :  1;
        const int32_t npm = m  ?  np1  :  np0;
        const bool t1 = (npm >> 6)*L.detection_ratio < (fp >> 6);
        const bool t2 = (np2 >> 6)*L.detection_ratio < (fp >> 7);
        const int imm = (fp >= L.sharp_threshold)  ?  (t1  ?  m  :  (t2  ?  2  :  -1))  :  -1;
        const int32_t p_dec = persistence - 1;
        const bool miss = (imm != notch_filter);
        const bool hit = (imm >= 0)  &&  (imm == last_present);
        const bool off_confirmed = present  &&  miss  &&  (p_dec == 0);
        const bool on_confirmed = !present  &&  hit  &&  (p_dec == 0);
        const int32_t pers_sharp = present  ?  (miss  ?  ((p_dec == 0)  ?  kOnCheck  :  p_dec)  :  kOffCheck)
                                            :  (hit  ?  ((p_dec == 0)  ?  kOffCheck  :  p_dec)  :  kOnCheck);
        const int bits = (imm == 0)  ?  SIG_1_PRESENT  :  (imm == 1)  ?  SIG_2_PRESENT  :  (SIG_1_PRESENT | SIG_2_PRESENT);
        int32_t st_sharp = off_confirmed  ?  ((state | ((state & (SIG_1_PRESENT | SIG_2_PRESENT)) << 1)) & ~(SIG_1_PRESENT | SIG_2_PRESENT))  :  state;
        st_sharp = on_confirmed  ?  (state | bits | (bits << 1))  :  st_sharp;
        int32_t nt_sharp = present  ?  notch_timeout  :  (notch_timeout - ((notch_timeout != 0)  ?  1  :  0));
        nt_sharp = on_confirmed  ?  kNotchLag  :  nt_sharp;
        state = flat  ?  st_flat  :  st_sharp;
        notch_timeout = flat  ?  nt_flat  :  nt_sharp;
        persistence = flat  ?  persistence  :  pers_sharp;
        notch_filter = (!flat  &&  on_confirmed)  ?  imm  :  notch_filter;
        const int immediate = flat  ?  -1  :  imm;
        // ---- the report, sig_tone.c:627-635 ----
        if (__builtin_expect((state & (SIG_1_CHANGE | SIG_2_CHANGE)) != 0, 0))
        {
            if (n_ev < L.ev_cap)
            {
                ev[3*n_ev] = at;
                ev[3*n_ev + 1] = state;
                ev[3*n_ev + 2] = duration;
            }
            n_ev++;
            state &= ~(SIG_1_CHANGE | SIG_2_CHANGE);
            duration = 0;
        }
        // ---- the media path, sig_tone.c:637-653 ----
        const float pick = (NT == 1)  ?  ((notch_filter == 0)  ?  notched0  :  0.0f)
                                      :  ((notch_filter == 0)  ?  notched0  :  (notch_filter == 1)  ?  notched1  :  notched2);
        const bool pass = (rx_tone & SIG_RX_PASSTHROUGH) != 0;
        const bool filter = (rx_tone & SIG_RX_FILTER_TONE)  ||  notch_timeout;
        a[k] = pass  ?  (filter  ?  sig_fsat(pick)  :  a[k])  :  0;
        last_present = immediate;
    };
    // Two copies of the chunk loop.  With aligned rows every chunk is one aligned 16-byte load, also a frame's last, shorter
    // one (it starts inside the row, so it cannot leave the page the row's last sample is on), there is no lane-dependent
    // branch around the loads and no other kind of load in the loop -- where such paths join, the compiler waits for
    // everything in flight, the chunk just requested included -- and a chunk that is whole in every lane (the common case)
    // runs its eight samples without guards, whose bodies the compiler would move out of line.
    auto frame = [&](auto aligned) __attribute__((always_inline))
    {
        constexpr bool VEC = decltype(aligned)::value;
        int4 q_next = {0, 0, 0, 0};
        __builtin_amdgcn_s_waitcnt(0x0F70);         // the state words home before the first chunk is requested (vmcnt is in order)
        if (VEC)
            q_next = *(const int4 *) row;
        for (int base = 0;  base < mylen;  base += 8)
        {
            const int todo = min(8, mylen - base);
            if (VEC)
            {
                const int4 q = q_next;
                a[0] = (int16_t) q.x;  a[1] = q.x >> 16;
                a[2] = (int16_t) q.y;  a[3] = q.y >> 16;
                a[4] = (int16_t) q.z;  a[5] = q.z >> 16;
                a[6] = (int16_t) q.w;  a[7] = q.w >> 16;
                q_next = *(const int4 *) (row + ((base + 8 < mylen)  ?  (base + 8)  :  0));
            }
            else
            {
#pragma unroll
                for (int k = 0;  k < 8;  k++)
                    a[k] = (k < todo)  ?  row[base + k]  :  0;
            }
            if (VEC  &&  __builtin_expect(__all(todo == 8), 1))
            {
#pragma unroll
                for (int k = 0;  k < 8;  k++)
                    sample(k, base + k);
                int4 q;
                q.x = (a[0] & 0xFFFF) | (a[1] << 16);
                q.y = (a[2] & 0xFFFF) | (a[3] << 16);
                q.z = (a[4] & 0xFFFF) | (a[5] << 16);
                q.w = (a[6] & 0xFFFF) | (a[7] << 16);
                *(int4 *) (row + base) = q;
            }
            else
            {
#pragma unroll
                for (int k = 0;  k < 8;  k++)
                {
                    if (k < todo)
                    {
                        sample(k, base + k);
                        row[base + k] = (int16_t) a[k];
                    }
                }
            }
        }
    };
    if (L.vec)
        frame(std::true_type{});
    else
        frame(std::false_type{});

#pragma unroll
    for (int j = 0;  j < NT;  j++)
    {
        st[(size_t) (5*j + SG_Z10)*n] = __float_as_int(t[j].z10);
        st[(size_t) (5*j + SG_Z11)*n] = __float_as_int(t[j].z11);
        st[(size_t) (5*j + SG_Z20)*n] = __float_as_int(t[j].z20);
        st[(size_t) (5*j + SG_Z21)*n] = __float_as_int(t[j].z21);
        st[(size_t) (5*j + SG_POWER)*n] = t[j].power;
    }
    if (kFlat)
    {
        st[(size_t) SG_FLAT_Z0*n] = __float_as_int(flat_z0);
        st[(size_t) SG_FLAT_Z1*n] = __float_as_int(flat_z1);
    }
    st[(size_t) SG_FLAT_POWER*n] = flat_power;
    st[(size_t) SG_PERSISTENCE*n] = persistence;
    st[(size_t) SG_LAST_PRESENT*n] = last_present;
    st[(size_t) SG_FLAT_MODE*n] = flat_mode;
    st[(size_t) SG_FLAT_TIMEOUT*n] = flat_timeout;
    st[(size_t) SG_NOTCH_TIMEOUT*n] = notch_timeout;
    st[(size_t) SG_STATE*n] = state;
    st[(size_t) SG_DURATION*n] = duration;
    st[(size_t) SG_NOTCH_FILTER*n] = notch_filter;
    L.ev_count[ch] = n_ev;
}

// ---- sender ----------------------------------------------------------------------------------------------------

struct SigTxLaunch
{
    int32_t *st;                // [kSigTxWords][n_ch]
    int16_t *pcm;               // rewritten in place
    const int16_t *quarter;     // the quarter sine of dds_int.c, 257 entries
    int32_t *start;             // [n_ch]: the sample a channel is taken up at; on return, where it stopped
    int32_t *request;           // [n_ch]: 1 = the channel stopped for its update request
    long long stride;
    int n_ch;
    int samples;
    int tones;
    int32_t phase_rate[2];
    int32_t scaling[2][2];
};

__device__ __forceinline__ int32_t sig_dds_lookup(const int16_t *quarter, uint32_t phase)
{
    // dds_lookup(), dds_int.c:340-355
    phase >>= 22;
    uint32_t step = phase & 255u;
    if (phase & 256u)
        step = 256u - step;
    const int32_t amp = quarter[step];
    return (phase & 512u)  ?  -amp  :  amp;
}

__global__ __launch_bounds__(64) void sigtone_tx_kernel(const SigTxLaunch L)
{
    __shared__ int16_t quarter[260];
    const int lane = threadIdx.x;
    for (int i = lane;  i < 257;  i += 64)
        quarter[i] = L.quarter[i];
    __syncthreads();
    const int ch = blockIdx.x*64 + lane;
    if (ch >= L.n_ch)
        return;
    const size_t n = (size_t) L.n_ch;
    int32_t *st = L.st + ch;
    int i = L.start[ch];
    const int len = L.samples;
    if (i >= len)
    {
        L.request[ch] = 0;
        return;
    }
    uint32_t phase[2] = {(uint32_t) st[(size_t) SX_PHASE0*n], (uint32_t) st[(size_t) SX_PHASE1*n]};
    int32_t high_low_timer = st[(size_t) SX_HIGH_LOW*n];
    const int32_t tone = st[(size_t) SX_TONE*n];
    int32_t timeout = st[(size_t) SX_TIMEOUT*n];
    int16_t *row = L.pcm + (size_t) ch*L.stride;
    int request = 0;
    // sig_tone_tx(), sig_tone.c:256-321: one pass of its loop per segment, up to the one that ends in the callback
    while (i < len  &&  !request)
    {
        int seg;
        if (timeout)
        {
            if (timeout <= len - i)
            {
                seg = timeout;
                request = 1;
            }
            else
            {
                seg = len - i;
            }
            timeout -= seg;
        }
        else
        {
            seg = len - i;
        }
        if (!(tone & SIG_TX_PASSTHROUGH))
        {
            for (int j = i;  j < i + seg;  j++)
                row[j] = 0;
        }
        if ((tone & (SIG_1_PRESENT | SIG_2_PRESENT)))
        {
            int high_low;
            if (high_low_timer > 0)
            {
                if (seg > high_low_timer)
                    seg = high_low_timer;
                high_low_timer -= seg;
                high_low = 0;
            }
            else
            {
                high_low = 1;
            }
            for (int k = 0;  k < L.tones;  k++)
            {
                const int bit = (k == 0)  ?  SIG_1_PRESENT  :  SIG_2_PRESENT;
                if ((tone & bit)  &&  L.phase_rate[k])
                {
                    const int32_t scale = L.scaling[k][high_low];
                    for (int j = i;  j < i + seg;  j++)
                    {
                        // dds_mod() and sat_add16()
                        const int32_t v = (int32_t) (int16_t) ((sig_dds_lookup(quarter, phase[k])*scale) >> 15);
                        phase[k] += (uint32_t) L.phase_rate[k];
                        const int32_t z = (int32_t) row[j] + v;
                        row[j] = (int16_t) min(max(z, -32768), 32767);
                    }
                }
            }
        }
        i += seg;
    }
    st[(size_t) SX_PHASE0*n] = (int32_t) phase[0];
    st[(size_t) SX_PHASE1*n] = (int32_t) phase[1];
    st[(size_t) SX_HIGH_LOW*n] = high_low_timer;
    st[(size_t) SX_TIMEOUT*n] = timeout;
    L.start[ch] = i;
    L.request[ch] = request;
}

// sig_tone_tx_set_mode(), sig_tone.c:326-345, on the channels whose mode is not negative
__global__ void sigtone_tx_set_mode_kernel(int32_t *st, int n_ch, const int32_t *modes, const int32_t *durations, int high_low_timeout)
{
    const int ch = blockIdx.x*blockDim.x + threadIdx.x;
    if (ch >= n_ch)
        return;
    const int mode = modes[ch];
    if (mode < 0)
        return;
    const size_t n = (size_t) n_ch;
    int32_t *s = st + ch;
    const int cur = s[(size_t) SX_TONE*n];
    const int old_tones = cur & (SIG_1_PRESENT | SIG_2_PRESENT);
    const int new_tones = mode & (SIG_1_PRESENT | SIG_2_PRESENT);
    if (new_tones  &&  old_tones != new_tones)
        s[(size_t) SX_HIGH_LOW*n] = high_low_timeout;
    if ((mode & SIG_1_PRESENT)  &&  !(cur & SIG_1_PRESENT))
        s[(size_t) SX_PHASE0*n] = 0;
    if ((mode & SIG_2_PRESENT)  &&  !(cur & SIG_2_PRESENT))
        s[(size_t) SX_PHASE1*n] = 0;
    s[(size_t) SX_TONE*n] = mode;
    s[(size_t) SX_TIMEOUT*n] = durations[ch];
}

}   // namespace spg
