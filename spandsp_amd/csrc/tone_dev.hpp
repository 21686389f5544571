// tone_dev.hpp -- device-side Goertzel bank: per-lane recurrences, block-end
// decisions and the LDS-staged frame walker shared by the DTMF / Bell MF / R2 MF /
// super-tone / generic Goertzel bank kernels (gfx950, wave64).
//
// Mapping: ONE CHANNEL PER LANE.  A wavefront owns 64 consecutive channels; their
// 2*NB recurrence registers stay in VGPRs for the whole frame.  PCM arrives
// channel-major (amp[ch][samples], what spandsp callers hold), so a wave's tile is
// one contiguous HBM region; it is fetched with 16-byte coalesced loads, 32 samples
// (64 B per channel) at a time, and transposed through a per-wave LDS tile with a
// 17-dword row pitch so the per-lane row reads are bank-conflict free.  The taps
// (2cos(w)) are wave-uniform and live in SGPRs.  No MFMA: the work is 8 (or 6, or
// M) independent 3-op recurrences per sample per channel.
//
// Numerics: every multiply and add is rounded separately, in the reference's order
// (this translation unit is compiled with -ffp-contract=off), so all float state and
// all decisions are bit-identical to the reference's strict-IEEE build:
//   recurrence   v3' = (fac*v2 - v1) + x                 src/spandsp/tone_detect.h:172-192
//   block result 2*((v3*v3 + v2*v2) - (v2*v3)*fac)      src/tone_detect.c:160-205
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

namespace spg {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kSegSamples = 32;             // samples staged per LDS tile (64 B per channel)
constexpr int kRowPitch = 17;               // dwords per LDS row: 16 data + 1 pad (odd => conflict-free)
constexpr int kMaxBins = 16;

// Kernel argument block (passed by value; lives in SGPRs / kernarg segment).
struct ToneLaunch
{
    const int16_t *amp;
    long long stride;           // samples between channels (channel-major) or between samples (sample-major)
    int samples;                // samples per channel in this call
    int n_ch;
    int layout;                 // 0 channel-major, 1 sample-major
    int aligned16;              // channel-major rows are 16-byte aligned (fast coalesced loader)
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
    float fac[kMaxBins];
    float threshold;
    float normal_twist;
    float reverse_twist;
};

// ---------------------------------------------------------------------------------
// Per-lane filter bank state
// ---------------------------------------------------------------------------------
template <int NB>
struct Bank
{
    float v2[NB];
    float v3[NB];

    __device__ __forceinline__ void step(const float (&fac)[NB], float x)
    {
#pragma unroll
        for (int i = 0;  i < NB;  i++)
        {
            const float v1 = v2[i];
            v2[i] = v3[i];
            v3[i] = fac[i]*v2[i] - v1 + x;
        }
    }

    // goertzel_result(): one zero sample, energy, reset (tone_detect.c:160-205)
    __device__ __forceinline__ float finish(int i, float f)
    {
        const float v1 = v2[i];
        const float a = v3[i];              // becomes v2
        const float b = f*a - v1;           // becomes v3
        float r = b*b + a*a - a*b*f;
        r *= 2.0f;
        v2[i] = 0.0f;
        v3[i] = 0.0f;
        return r;
    }

    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int i = 0;  i < NB;  i++)
        {
            v2[i] = 0.0f;
            v3[i] = 0.0f;
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
// block energy is accumulated, the optional input filter, and the block-end decision.
// Integer state is two 32-bit words per channel: w0 = current_sample (16 bits) | ...
// ---------------------------------------------------------------------------------

// ---- DTMF (src/dtmf.c:132-361) ------------------------------------------------------
template <bool FILTER>
struct DtmfDet
{
    static constexpr int NB = 8;
    static constexpr bool kEnergy = true;
    static constexpr int kExtra = FILTER  ?  4  :  0;       // z350[2], z440[2]
    static constexpr int NSF = 2*NB + 1 + kExtra;
    static constexpr bool kRuntimeBlock = false;
    __device__ static __forceinline__ int block_len(const ToneLaunch &) { return 102; }    // dtmf.c:71

    float z[FILTER  ?  4  :  1];

    __device__ __forceinline__ void load_extra(const ToneLaunch &L, int ch)
    {
        if (FILTER)
        {
#pragma unroll
            for (int i = 0;  i < 4;  i++)
                z[i] = L.sf[(size_t) (2*NB + 1 + i)*L.n_ch + ch];
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
            z[1] = z[0];
            z[0] = v1;
            v1 = 0.98456f*f + 1.8529543f*z[2] - 0.9691396f*z[3];
            f = v1 - 1.8819938f*z[2] + z[3];
            z[3] = z[2];
            z[2] = v1;
            return f;
        }
        return x;
    }

    // Block end: energies, decision (dtmf.c:209-258), debounce (dtmf.c:304-347).
    // w0 = cs | last_hit<<16 | in_digit<<24 ; w1 = duration.
    __device__ __forceinline__ void block_end(const ToneLaunch &L, Bank<NB> &bk, float &energy,
                                              uint32_t &w0, int32_t &w1, int ch, int nb, bool live)
    {
        float e[NB];
#pragma unroll
        for (int i = 0;  i < NB;  i++)
            e[i] = bk.finish(i, L.fac[i]);
        if (L.trace  &&  live)
        {
#pragma unroll
            for (int i = 0;  i < NB;  i++)
                L.trace[((size_t) nb*(NB + 1) + i)*L.n_ch + ch] = e[i];
            L.trace[((size_t) nb*(NB + 1) + NB)*L.n_ch + ch] = energy;
        }
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
        bool ok = (er >= L.threshold)  &&  (ec >= L.threshold);
        ok = ok  &&  (ec < er*L.reverse_twist)  &&  (ec*L.normal_twist > er);
        bool peaky = true;
#pragma unroll
        for (int i = 0;  i < 4;  i++)
        {
            // dtmf.c:243-246; relative peak ratios are both 6.309f (dtmf.c:107-108)
            if ((i != bc  &&  e[4 + i]*6.309f > ec)  ||  (i != br  &&  e[i]*6.309f > er))
                peaky = false;
        }
        ok = ok  &&  peaky  &&  ((er + ec) > 83.868f*energy);     // dtmf.c:109,250-252
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
                    if (L.rec_dur  &&  live)
                        L.rec_dur[(size_t) nb*L.n_ch + ch] = w1;
                    w1 = 0;
                }
            }
            in_digit = hit;
            code = hit;
        }
        last_hit = hit;
        if (L.rec_energy  &&  live)
            L.rec_energy[(size_t) nb*L.n_ch + ch] = energy;
        if (live)
            L.rec[(size_t) nb*L.n_ch + ch] = make_rec(raw, code, flags);
        energy = 0.0f;
        w0 = ((uint32_t) last_hit << 16) | ((uint32_t) in_digit << 24);       // cs = 0
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
    float eb = e[best];
    float es = e[second];
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
    bool ok = (eb >= threshold)  &&  (es >= threshold)  &&  (eb < es*twist)  &&  (eb*twist > es);
#pragma unroll
    for (int i = 0;  i < 6;  i++)
    {
        if (i != best  &&  i != second  &&  e[i]*rel_peak >= es)
            ok = false;
    }
    if (!ok)
        return -1;
    const int lo = (second < best)  ?  second  :  best;
    const int hi = (second < best)  ?  best  :  second;
    return lo*5 + hi - 1;
}

struct BellMfDet
{
    static constexpr int NB = 6;
    static constexpr bool kEnergy = false;
    static constexpr int NSF = 2*NB;
    static constexpr bool kRuntimeBlock = false;
    __device__ static __forceinline__ int block_len(const ToneLaunch &) { return 120; }    // bell_r2_mf.c:204
    __device__ __forceinline__ void load_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ void store_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ float prefilter(float x) { return x; }

    // w0 = cs | hits[0]<<16 | hits[1]<<24 ; w1 = hits[2] | hits[3]<<8 | hits[4]<<16
    __device__ __forceinline__ void block_end(const ToneLaunch &L, Bank<NB> &bk, float &,
                                              uint32_t &w0, int32_t &w1, int ch, int nb, bool live)
    {
        float e[NB];
#pragma unroll
        for (int i = 0;  i < NB;  i++)
            e[i] = bk.finish(i, L.fac[i]);
        if (L.trace  &&  live)
        {
#pragma unroll
            for (int i = 0;  i < NB;  i++)
                L.trace[((size_t) nb*(NB + 1) + i)*L.n_ch + ch] = e[i];
            L.trace[((size_t) nb*(NB + 1) + NB)*L.n_ch + ch] = 0.0f;
        }
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
        if (live)
            L.rec[(size_t) nb*L.n_ch + ch] = make_rec(hit, code, flags);
        // bell_r2_mf.c:657-661: shift the hit history
        w0 = ((uint32_t) h1 << 16) | ((uint32_t) h2 << 24);
        w1 = (int32_t) ((uint32_t) h3 | ((uint32_t) h4 << 8) | ((uint32_t) hit << 16));
    }
};

struct R2MfDet
{
    static constexpr int NB = 6;
    static constexpr bool kEnergy = false;
    static constexpr int NSF = 2*NB;
    static constexpr bool kRuntimeBlock = false;
    __device__ static __forceinline__ int block_len(const ToneLaunch &) { return 133; }    // bell_r2_mf.c:206
    __device__ __forceinline__ void load_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ void store_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ float prefilter(float x) { return x; }

    // w0 = cs | current_digit<<16
    __device__ __forceinline__ void block_end(const ToneLaunch &L, Bank<NB> &bk, float &,
                                              uint32_t &w0, int32_t &, int ch, int nb, bool live)
    {
        float e[NB];
#pragma unroll
        for (int i = 0;  i < NB;  i++)
            e[i] = bk.finish(i, L.fac[i]);
        if (L.trace  &&  live)
        {
#pragma unroll
            for (int i = 0;  i < NB;  i++)
                L.trace[((size_t) nb*(NB + 1) + i)*L.n_ch + ch] = e[i];
            L.trace[((size_t) nb*(NB + 1) + NB)*L.n_ch + ch] = 0.0f;
        }
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
        if (live)
            L.rec[(size_t) nb*L.n_ch + ch] = make_rec(digit, digit, flags);
        w0 = (uint32_t) digit << 16;
    }
};

// ---- Super tone (src/super_tone_rx.c:289-362 on device; cadence FSM on the host) --------
// and the generic Goertzel bank (goertzel_update / goertzel_result, tone_detect.c:123-205).
template <int NBINS, bool SUPER>
struct MultiDet
{
    static constexpr int NB = NBINS;
    static constexpr bool kEnergy = SUPER;
    static constexpr int NSF = 2*NB + (SUPER  ?  1  :  0);
    static constexpr bool kRuntimeBlock = !SUPER;
    __device__ static __forceinline__ int block_len(const ToneLaunch &L) { return SUPER  ?  128  :  L.block_len; }
    __device__ __forceinline__ void load_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ void store_extra(const ToneLaunch &, int) {}
    __device__ __forceinline__ float prefilter(float x) { return x; }

    __device__ __forceinline__ void block_end(const ToneLaunch &L, Bank<NB> &bk, float &energy,
                                              uint32_t &w0, int32_t &, int ch, int nb, bool live)
    {
        float e[NB];
        const int m = L.nbins;
        if (SUPER)
        {
            int k1 = -1;
            int k2 = -1;
            // super_tone_rx.c:301-309: below the total-energy gate the bins are reset unread
            const bool loud = !(energy < 2104205.6f);               // super_tone_rx.c:75
#pragma unroll
            for (int i = 0;  i < NB;  i++)
            {
                const float r = bk.finish(i, L.fac[i]);
                e[i] = loud  ?  r  :  0.0f;
            }
            if (loud)
            {
                // super_tone_rx.c:320-347 (requires m >= 2)
                float e1;
                float e2;
                if (e[0] > e[1])
                {
                    k1 = 0;
                    k2 = 1;
                }
                else
                {
                    k1 = 1;
                    k2 = 0;
                }
                e1 = e[k1 == 0  ?  0  :  1];
                e2 = e[k2 == 0  ?  0  :  1];
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
            if (L.rec_energy  &&  live)
                L.rec_energy[(size_t) nb*L.n_ch + ch] = energy;
            if (live)
                L.rec[(size_t) nb*L.n_ch + ch] = make_rec(k1 + 1, k2 + 1, kBlkValid);
            if (L.trace  &&  live)
                L.trace[((size_t) nb*(NB + 1) + NB)*L.n_ch + ch] = energy;
            energy = 0.0f;
        }
        else
        {
#pragma unroll
            for (int i = 0;  i < NB;  i++)
                e[i] = bk.finish(i, L.fac[i]);
            if (live)
                L.rec[(size_t) nb*L.n_ch + ch] = make_rec(0, 0, kBlkValid);
            if (L.trace  &&  live)
                L.trace[((size_t) nb*(NB + 1) + NB)*L.n_ch + ch] = 0.0f;
        }
        if (L.trace  &&  live)
        {
#pragma unroll
            for (int i = 0;  i < NB;  i++)
                L.trace[((size_t) nb*(NB + 1) + i)*L.n_ch + ch] = e[i];
        }
        w0 = 0;
    }
};

// ---------------------------------------------------------------------------------
// The bank kernel
// ---------------------------------------------------------------------------------
template <class Det>
__global__ __launch_bounds__(kWave*kWavesPerBlock)
void tone_bank_kernel(const ToneLaunch L)
{
    constexpr int NB = Det::NB;
    __shared__ int tile[kWavesPerBlock][kWave*kRowPitch];

    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x >> 6;
    const int ch0 = (blockIdx.x*kWavesPerBlock + wv)*kWave;
    if (ch0 >= L.n_ch)
        return;                                     // whole wave idle (wave-uniform exit)
    const bool live = (ch0 + lane) < L.n_ch;
    const int ch = live  ?  (ch0 + lane)  :  (L.n_ch - 1);      // dead lanes shadow the last channel, never store

    // ---- load per-channel state (coalesced: SoA, lane == channel) -------------------
    Bank<NB> bk;
    Det det;
    float fac[NB];
#pragma unroll
    for (int i = 0;  i < NB;  i++)
    {
        fac[i] = L.fac[i];
        bk.v2[i] = L.sf[(size_t) i*L.n_ch + ch];
        bk.v3[i] = L.sf[(size_t) (NB + i)*L.n_ch + ch];
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
    const int cs_first = __builtin_amdgcn_readfirstlane(cs);
    const bool uniform = __all(cs == cs_first);

    int *mytile = &tile[wv][0];
    const short *row = (const short *) &mytile[lane*kRowPitch];

    int nb = 0;                 // blocks completed by this lane in this call
    int take_acc = 0;           // samples since the last duration update (dtmf.c:202-204)

    // ---- prefetch registers for the coalesced channel-major loader -------------------
    // Tile = 64 rows x 64 B; chunk c = j*64 + lane -> row c>>2, 16-byte column c&3.
    int4 g[4];
    auto fetch = [&](int seg_base)
    {
#pragma unroll
        for (int j = 0;  j < 4;  j++)
        {
            const int c = j*kWave + lane;
            const int r = c >> 2;
            const int col = c & 3;
            const int s0 = seg_base + col*8;
            g[j] = make_int4(0, 0, 0, 0);
            if ((ch0 + r) < L.n_ch  &&  s0 < L.samples)
                g[j] = *(const int4 *) (L.amp + (size_t) (ch0 + r)*L.stride + s0);
        }
    };
    auto commit = [&]()
    {
#pragma unroll
        for (int j = 0;  j < 4;  j++)
        {
            const int c = j*kWave + lane;
            int *p = &mytile[(c >> 2)*kRowPitch + (c & 3)*4];
            p[0] = g[j].x;
            p[1] = g[j].y;
            p[2] = g[j].z;
            p[3] = g[j].w;
        }
    };
    const bool fast_loader = (L.layout == 0)  &&  L.aligned16;
    if (fast_loader)
        fetch(0);

    for (int seg_base = 0;  seg_base < L.samples;  seg_base += kSegSamples)
    {
        const int seglen = min(kSegSamples, L.samples - seg_base);
        // ---- stage this segment into the wave's LDS tile ------------------------------
        if (fast_loader)
        {
            commit();
            if (seg_base + kSegSamples < L.samples)
                fetch(seg_base + kSegSamples);
        }
        else
        {
            short *wrow = (short *) &mytile[lane*kRowPitch];
            if (L.layout == 0)
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
        // (single wave per tile: LDS operations of one wave complete in order, no barrier)

        // ---- consume it ------------------------------------------------------------------
        if (uniform)
        {
            int pos = 0;
            int cs_s = __builtin_amdgcn_readfirstlane(cs);
            while (pos < seglen)
            {
                int run = block - cs_s;
                if (run > seglen - pos)
                    run = seglen - pos;
                int k = 0;
                for (  ;  k + 4 <= run;  k += 4)
                {
                    const float x0 = det.prefilter((float) row[pos + k]);
                    const float x1 = det.prefilter((float) row[pos + k + 1]);
                    const float x2 = det.prefilter((float) row[pos + k + 2]);
                    const float x3 = det.prefilter((float) row[pos + k + 3]);
                    if (Det::kEnergy)
                        energy += x0*x0;
                    bk.step(fac, x0);
                    if (Det::kEnergy)
                        energy += x1*x1;
                    bk.step(fac, x1);
                    if (Det::kEnergy)
                        energy += x2*x2;
                    bk.step(fac, x2);
                    if (Det::kEnergy)
                        energy += x3*x3;
                    bk.step(fac, x3);
                }
                for (  ;  k < run;  k++)
                {
                    const float x = det.prefilter((float) row[pos + k]);
                    if (Det::kEnergy)
                        energy += x*x;
                    bk.step(fac, x);
                }
                pos += run;
                cs_s += run;
                take_acc += run;
                if (cs_s >= block)
                {
                    if (w1 < INT_MAX - take_acc)
                        w1 += take_acc;
                    take_acc = 0;
                    det.block_end(L, bk, energy, w0, w1, ch, nb, live);
                    nb++;
                    cs_s = 0;
                }
            }
            cs = cs_s;
        }
        else
        {
            // Divergent block phases inside the wave: correct, slower.
            for (int pos = 0;  pos < seglen;  pos++)
            {
                const float x = det.prefilter((float) row[pos]);
                if (Det::kEnergy)
                    energy += x*x;
                bk.step(fac, x);
                cs++;
                take_acc++;
                if (cs >= block)
                {
                    if (w1 < INT_MAX - take_acc)
                        w1 += take_acc;
                    take_acc = 0;
                    det.block_end(L, bk, energy, w0, w1, ch, nb, live);
                    nb++;
                    cs = 0;
                }
            }
        }
    }
    if (take_acc > 0  &&  w1 < INT_MAX - take_acc)
        w1 += take_acc;

    // ---- write back ---------------------------------------------------------------------------
    if (live)
    {
#pragma unroll
        for (int i = 0;  i < NB;  i++)
        {
            L.sf[(size_t) i*L.n_ch + ch] = bk.v2[i];
            L.sf[(size_t) (NB + i)*L.n_ch + ch] = bk.v3[i];
        }
        if (Det::kEnergy)
            L.sf[(size_t) (2*NB)*L.n_ch + ch] = energy;
        det.store_extra(L, ch);
        L.si[ch] = (int32_t) (w0 | (uint32_t) cs);
        L.si[(size_t) L.n_ch + ch] = w1;
        for (int b = nb;  b < L.maxb;  b++)
            L.rec[(size_t) b*L.n_ch + ch] = 0;         // slots without a completed block
    }
}

}   // namespace spg
