// Super-tone cadence matching for one channel (one lane): what super_tone_rx.c:164-228 (test_cadence) and :364-448 (the
// tail of super_tone_chunk) decide from the stream of per-block bin pairs (k1, k2).  Used by cadence_kernel (a launch of
// its own behind any detector launch) and by the streaming super-tone kernel's epilogue (tone_fast.hpp, ABL bit
// kToneCadence), where the lane that just produced a channel's records walks them before the wave ends.
//
// The ten newest runs are kept newest first, so that "the run j places back" is register j: the 24 state words are read in
// one go, shifted down when a run ends (a few times a second) and only what changed is written back.  Window limits are
// kept in blocks (lo <= 128 b  <=>  b >= ceil(lo/128)), which makes every test a 32-bit compare.
//   state words ([word][channel]) as spangpu_bank_cadence_get_state() shows them: 0 seen f1, 1 seen f2, 2 tone followed
//   (-1 none), 3 turn, 4..13 run pair (f1 & 0xFFFF | f2 << 16), newest first, 14..23 run length in blocks.
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
constexpr int kCadLdsTones = 32;              // the streaming kernel keeps the cadence tables in LDS: up to this many tones
constexpr int kCadLdsElems = 96;              // ... and elements (larger sets are matched by cadence_kernel)
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
    int n_elems;                // elements in all (first[n_tones])
};

// A lane's cadence state, asked for early (the streaming kernel requests it with the detector state, a whole frame before
// it is needed: read at the end of the kernel its latency overlapped nothing)
struct CadenceRegs
{
    int32_t w[kCadWords];
};

__device__ static inline void cadence_state_load(const CadenceArgs &A, int ch, int n_ch, CadenceRegs &r)
{
#pragma unroll
    for (int i = 0;  i < kCadWords;  i++)
        r.w[i] = A.state[(size_t) i*n_ch + ch];
}

__host__ __device__ static inline int32_t cad_pair(int f1, int f2)
{
    return (int32_t) (((uint32_t) f1 & 0xFFFFu) | ((uint32_t) f2 << 16));
}

// Walks the records of one launch for channel ch (rec0 / rec1 = the first two, already in registers; further ones are read
// from rec).  Returns the number of events left in the slot arrays.  A lane that is not `active` walks along and stores
// nothing.
// `first` / `elem`: the cadence tables, wherever the caller keeps them (global memory, or its copy in LDS); `r`: the lane's
// state as cadence_state_load() fetched it.
//
// `turn` (word 3) counts the elements of the followed cadence that have gone by, modulo its length.  (A ring for the ten
// runs in memory, so that a run end rewrites two words instead of twenty, was tried: putting the ring into age order
// in registers cost more than the stores it saved.)
template <class FirstT, class ElemT>
__device__ static inline int cadence_walk_loaded(const CadenceArgs &A, const FirstT &first, const ElemT &elem, const CadenceRegs &r,
                                                 int ch, int n_ch, int maxb, uint32_t rec0, uint32_t rec1, const uint32_t *rec,
                                                 bool active)
{
    int32_t *st = A.state;
    int32_t pf[kCadHistory];
    int32_t bl[kCadHistory];
    int seen1 = r.w[0];
    int seen2 = r.w[1];
    int tone = r.w[2];
    int turn = r.w[3];
#pragma unroll
    for (int j = 0;  j < kCadHistory;  j++)
    {
        pf[j] = r.w[4 + j];
        bl[j] = r.w[4 + kCadHistory + j];
    }
    const int32_t pf0_in = pf[0];
    const int32_t bl0_in = bl[0];
    int shifts = 0;
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
    // Is the cadence followed still alive?  `turn_now` elements of it have gone by since it was recognised (on its last
    // element), counted modulo its length n: the current run must be element turn_now - 1 (mod n) and not yet too long; when
    // a run has just ended, the one before it must in addition have been a proper element turn_now - 2 (mod n).  `n_out`:
    // the cadence's length.
    auto alive = [&](int t, int turn_now, bool run_ended, int &n_out)
    {
        const int e0 = first[t];
        const int n = first[t + 1] - e0;
        n_out = n;
        if (n <= 0)
            return false;
        // (a count set from outside -- spangpu_bank_cadence_set_state(), for a tone whose length the host does not keep -- may
        // not be reduced yet: never index past the cadence's own elements)
        while (turn_now >= n)
            turn_now -= n;
        int i1 = turn_now - 1;
        i1 += (i1 < 0)  ?  n  :  0;
        int i2 = i1 - 1;
        i2 += (i2 < 0)  ?  n  :  0;
        if (run_ended  &&  !fits(elem[e0 + i2], pf[1], bl[1]))
            return false;
        const int4 e = elem[e0 + i1];
        return e.x == pf[0]  &&  bl[0] <= e.z;
    };
    for (blk = 0;  blk < maxb;  blk++)
    {
        const uint32_t w = (blk == 0)  ?  rec0  :  (blk == 1)  ?  rec1  :  rec[(size_t) blk*n_ch + ch];
        if (!((w >> 16) & kCadBlkValid))
            continue;
        touched = true;
        bool lost_now = false;
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
                // (the count of elements gone by moves on by one, modulo the cadence's length)
                int n_t = 0;
                const bool ok = alive(tone, turn, true, n_t);
                turn = (turn + 1 >= n_t)  ?  0  :  (turn + 1);
                if (!ok)
                {
                    tone = -1;
                    lost_now = true;
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
            shifts++;
        }
        else
        {
            // more of the same (tested before this block is counted, as the reference does)
            if (tone >= 0)
            {
                int n_t = 0;
                if (!alive(tone, turn, false, n_t))
                {
                    tone = -1;
                    lost_now = true;
                    emit(2, -1, -1);
                }
            }
            bl[0]++;
        }
        if (tone >= 0)
            continue;
        // Do the newest runs spell out a whole cadence, the current run being its last element?  While a run goes on only its
        // length changes, by one per block, so a cadence can first fit at the block where the run reaches its last element's
        // least length (or at once, if that is one block): only there are its other elements looked at, when any lane of
        // the wave is at that point.  The exception: a channel that has just stopped following a tone (or was told to forget
        // the one it followed: tone == -2) may find itself in the middle of such a window, and looks at everything once.
        // The tables are the same for every lane: every cadence's last element is requested first, all together, then the
        // elements of the cadences that are due, a cadence at a time and without short cuts.
        const bool look_all = lost_now  ||  (tone == -2);
        tone = -1;
        int found = -1;
        int e_next = first[0];
        for (int t = 0;  t < A.n_tones;  t++)
        {
            const int e0 = e_next;
            e_next = first[t + 1];
            const int n = e_next - e0;
            if (n > kCadHistory)
                continue;
            if (n > 0)
            {
                const int4 last = elem[e0 + n - 1];
                const bool due = look_all  ||  (last.x == pf[0]  &&  bl[0] == ((last.y > 1)  ?  last.y  :  1));
                if (!__any(due))
                    continue;
            }
            bool ok = true;
            if (n <= 4)
            {
                // (what call progress cadences are made of: a batch of four, not of ten)
                int4 els[4];
#pragma unroll
                for (int j = 0;  j < 4;  j++)
                    els[j] = elem[(j < n)  ?  (e0 + n - 1 - j)  :  0];
#pragma unroll
                for (int j = 0;  j < 4;  j++)
                    ok = ok  &  ((j >= n)  |  fits(els[j], pf[j], bl[j]));
            }
            else
            {
                int4 els[kCadHistory];
#pragma unroll
                for (int j = 0;  j < kCadHistory;  j++)
                    els[j] = elem[(j < n)  ?  (e0 + n - 1 - j)  :  0];
#pragma unroll
                for (int j = 0;  j < kCadHistory;  j++)
                    ok = ok  &  ((j >= n)  |  fits(els[j], pf[j], bl[j]));
            }
            found = (ok  &&  found < 0)  ?  t  :  found;
        }
        if (found >= 0)
        {
            tone = found;
            turn = 0;
            emit(1, -1, found);
        }
    }
    if (active)
        A.count[ch] = n_ev;
    if (touched  &&  active)
    {
        // only what changed goes back: on a line that stays as it is one word per block (the length of its current run);
        // after a run end all of the runs
        if (seen1 != r.w[0])
            st[(size_t) 0*n_ch + ch] = seen1;
        if (seen2 != r.w[1])
            st[(size_t) 1*n_ch + ch] = seen2;
        if (tone != r.w[2])
            st[(size_t) 2*n_ch + ch] = tone;
        if (turn != r.w[3])
            st[(size_t) 3*n_ch + ch] = turn;
        if (shifts == 0)
        {
            if (pf[0] != pf0_in)
                st[(size_t) 4*n_ch + ch] = pf[0];
            if (bl[0] != bl0_in)
                st[(size_t) (4 + kCadHistory)*n_ch + ch] = bl[0];
        }
        else
        {
#pragma unroll
            for (int i = 0;  i < kCadHistory;  i++)
            {
                st[(size_t) (4 + i)*n_ch + ch] = pf[i];
                st[(size_t) (4 + kCadHistory + i)*n_ch + ch] = bl[i];
            }
        }
    }
    return n_ev;
}

__device__ static inline int cadence_walk(const CadenceArgs &A, int ch, int n_ch, int maxb, uint32_t rec0, uint32_t rec1,
                                          const uint32_t *rec, bool active)
{
    CadenceRegs r;
    cadence_state_load(A, ch, n_ch, r);
    return cadence_walk_loaded(A, A.first, A.elem, r, ch, n_ch, maxb, rec0, rec1, rec, active);
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
