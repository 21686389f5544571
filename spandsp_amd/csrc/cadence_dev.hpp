// Super-tone cadence matching for one channel (one lane): what super_tone_rx.c:164-228 (test_cadence) and :364-448 (the
// tail of super_tone_chunk) decide from the stream of per-block bin pairs (k1, k2).  Used by cadence_kernel (a launch of
// its own behind any detector launch) and by the streaming super-tone kernel's epilogue (tone_fast.hpp, ABL bit
// kToneCadence), where the lane that just produced a channel's records walks them before the wave ends.
//
// The ten newest runs are kept newest first, so that "the run j places back" is register j: the 24 state words are read in
// one go, shifted down when a run ends (a few times a second) and only what changed is written back.  Window limits are
// kept in blocks (lo <= 128 b  <=>  b >= ceil(lo/128)), which makes every test a 32-bit compare.
//   state words ([word][channel]): 0 seen f1, 1 seen f2, 2 tone followed (-1 none), 3 turn, 4..13 run pair
//   (f1 & 0xFFFF | f2 << 16), newest first, 14..23 run length in blocks.
// Events, in the order the reference calls back, two words each at ev[(slot*n_ch + ch)*2]:
//   word 0 = kind | (f1 + 1) << 8 | (f2 + 1) << 16 | block << 24, word 1 = tone number (kind 1) or milliseconds (kind 3);
//   kind 1 = tone recognised (tone_callback(user, tone, -10, 0)), 2 = tone lost (tone_callback(user, -1, -10, 0)),
//   3 = a segment ended (segment_callback(user, f1, f2, ms)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spg {

constexpr int kCadHistory = 10;
constexpr int kCadWords = 4 + 2*kCadHistory;
constexpr int kCadSlotsPerBlock = 3;
constexpr uint32_t kCadBlkValid = 0x01;        // SPANGPU_BLK_VALID in the record's flag byte

struct CadenceArgs
{
    const int32_t *first;       // [n_tones + 1]: where each tone's elements start
    const int4 *elem;           // (pair, least blocks, most blocks, 0)
    int32_t *state;             // [kCadWords][n_ch]; nullptr = no cadences to match in this launch
    uint32_t *ev;               // [slots][n_ch][2]
    int32_t *count;             // [n_ch]
    uint32_t *list;             // two counters used in turn, then (channel, word 0, word 1) per event
    uint32_t list_cap;
    int n_tones;
    int segments;
    int which;                  // the counter this launch adds to (it clears the other for the next launch)
};

__host__ __device__ static inline int32_t cad_pair(int f1, int f2)
{
    return (int32_t) (((uint32_t) f1 & 0xFFFFu) | ((uint32_t) f2 << 16));
}

// Walks the records of one launch for channel ch (rec0 / rec1 = the first two, already in registers; further ones are read
// from rec).  Returns the number of events left in the slot arrays.  A lane that is not `active` walks along and stores
// nothing.
__device__ static inline int cadence_walk(const CadenceArgs &A, int ch, int n_ch, int maxb, uint32_t rec0, uint32_t rec1,
                                          const uint32_t *rec, bool active)
{
    const int32_t *__restrict__ first = A.first;
    const int4 *__restrict__ elem = A.elem;
    int32_t *st = A.state;
    int32_t pf[kCadHistory];
    int32_t bl[kCadHistory];
    int32_t w0[4];
#pragma unroll
    for (int i = 0;  i < 4;  i++)
        w0[i] = st[(size_t) i*n_ch + ch];
#pragma unroll
    for (int i = 0;  i < kCadHistory;  i++)
    {
        pf[i] = st[(size_t) (4 + i)*n_ch + ch];
        bl[i] = st[(size_t) (4 + kCadHistory + i)*n_ch + ch];
    }
    int seen1 = w0[0];
    int seen2 = w0[1];
    int tone = w0[2];
    int turn = w0[3];
    bool shifted = false;
    bool touched = false;
    int n_ev = 0;
    int blk = 0;
    auto emit = [&](uint32_t kind, int32_t pair, int32_t v)
    {
        if (!active)
            return;
        uint32_t *e = A.ev + ((size_t) n_ev*n_ch + ch)*2;
        const int f1 = (int) (int16_t) (pair & 0xFFFF);
        const int f2 = pair >> 16;
        e[0] = kind | ((uint32_t) ((f1 + 1) & 0xFF) << 8) | ((uint32_t) ((f2 + 1) & 0xFF) << 16) | ((uint32_t) blk << 24);
        e[1] = (uint32_t) v;
        n_ev++;
    };
    auto fits = [&](const int4 &e, int32_t pair, int32_t blocks) { return e.x == pair  &&  e.y <= blocks  &&  blocks <= e.z; };
    // Is the cadence followed still alive?  `turn` elements of it have gone by since it was recognised (on its last element),
    // so the current run must be element (turn - 1) mod n and not yet too long; when a run has just ended, the one before
    // it must in addition have been a proper element (turn - 2) mod n.
    auto alive = [&](int t, int turn_now, bool run_ended)
    {
        const int e0 = first[t];
        const int n = first[t + 1] - e0;
        if (n <= 0)
            return false;
        if (run_ended  &&  !fits(elem[e0 + (turn_now + n - 2)%n], pf[1], bl[1]))
            return false;
        const int4 e = elem[e0 + (turn_now + n - 1)%n];
        return e.x == pf[0]  &&  bl[0] <= e.z;
    };
    for (blk = 0;  blk < maxb;  blk++)
    {
        const uint32_t w = (blk == 0)  ?  rec0  :  (blk == 1)  ?  rec1  :  rec[(size_t) blk*n_ch + ch];
        if (!((w >> 16) & kCadBlkValid))
            continue;
        touched = true;
        const int k1 = (int) (w & 0xFF) - 1;
        const int k2 = (int) ((w >> 8) & 0xFF) - 1;
        const int32_t pair = cad_pair(k1, k2);
        const bool repeat = (k1 == seen1  &&  k2 == seen2);
        seen1 = k1;
        seen2 = k2;
        if (!repeat)
        {
            // a pair seen for the first time may be a glitch: the block still counts towards the current run
            bl[0]++;
        }
        else if (pair != pf[0])
        {
            // seen twice in a row and not what the current run is made of: that run is over
            if (tone >= 0)
            {
                const int t_now = turn++;
                if (!alive(tone, t_now, true))
                {
                    tone = -1;
                    emit(2, -1, -1);
                }
            }
            if (A.segments)
                emit(3, pf[0], (int32_t) ((uint32_t) bl[0]*16u));
#pragma unroll
            for (int i = kCadHistory - 1;  i > 0;  i--)
            {
                pf[i] = pf[i - 1];
                bl[i] = bl[i - 1];
            }
            pf[0] = pair;
            bl[0] = 1;
            shifted = true;
        }
        else
        {
            // more of the same (tested before this block is counted, as the reference does)
            if (tone >= 0  &&  !alive(tone, turn, false))
            {
                tone = -1;
                emit(2, -1, -1);
            }
            bl[0]++;
        }
        if (tone >= 0)
            continue;
        // do the newest runs spell out a whole cadence, the current run being its last element?
        for (int t = 0;  t < A.n_tones;  t++)
        {
            const int e0 = first[t];
            const int n = first[t + 1] - e0;
            if (n > kCadHistory)
                continue;
            bool ok = true;
#pragma unroll
            for (int j = 0;  j < kCadHistory;  j++)
            {
                if (j < n)
                    ok = ok  &&  fits(elem[e0 + n - 1 - j], pf[j], bl[j]);
            }
            if (ok)
            {
                tone = t;
                turn = 0;
                emit(1, -1, t);
                break;
            }
        }
    }
    if (active)
        A.count[ch] = n_ev;
    if (touched  &&  active)
    {
        st[(size_t) 0*n_ch + ch] = seen1;
        st[(size_t) 1*n_ch + ch] = seen2;
        st[(size_t) 2*n_ch + ch] = tone;
        st[(size_t) 3*n_ch + ch] = turn;
        st[(size_t) 4*n_ch + ch] = pf[0];
        st[(size_t) (4 + kCadHistory)*n_ch + ch] = bl[0];
        if (shifted)
        {
#pragma unroll
            for (int i = 1;  i < kCadHistory;  i++)
            {
                st[(size_t) (4 + i)*n_ch + ch] = pf[i];
                st[(size_t) (4 + kCadHistory + i)*n_ch + ch] = bl[i];
            }
        }
    }
    return n_ev;
}

// The lane's n_ev events (just written to the slot arrays) onto the compact list at position `at`.
__device__ static inline void cadence_list_copy(const CadenceArgs &A, int ch, int n_ch, int n_ev, uint32_t at)
{
    for (int k = 0;  k < n_ev;  k++)
    {
        if (at + k < A.list_cap)
        {
            const uint32_t *e = A.ev + ((size_t) k*n_ch + ch)*2;
            uint32_t *o = A.list + 2 + (size_t) (at + k)*3;
            o[0] = (uint32_t) ch;
            o[1] = e[0];
            o[2] = e[1];
        }
    }
}

// Room on the list for a whole wave's events with one atomic (all 64 lanes must be here); a channel's events stay together.
__device__ static inline void cadence_list_wave(const CadenceArgs &A, int ch, int n_ch, int n_ev, unsigned lane)
{
    int x = n_ev;
#pragma unroll
    for (int d = 1;  d < 64;  d <<= 1)
    {
        const int y = __shfl_up(x, d);
        if ((int) lane >= d)
            x += y;
    }
    const int total = __shfl(x, 63);
    if (total == 0)
        return;
    uint32_t base = 0;
    if (lane == 0)
        base = atomicAdd(A.list + A.which, (uint32_t) total);
    base = (uint32_t) __shfl((int) base, 0);
    cadence_list_copy(A, ch, n_ch, n_ev, base + (uint32_t) (x - n_ev));
}

}   // namespace spg
