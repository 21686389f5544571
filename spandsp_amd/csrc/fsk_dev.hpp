// fsk_dev.hpp -- device side of the FSK receiver banks: N non-coherent FSK
// demodulators (V.21, V.23, Bell 103/202, Weitbrecht), one channel per lane, integer arithmetic
// throughout, so the results are bit-exact with the reference by construction.
//
// What is restated (paths relative to the reference tree):
//   fsk_rx()                   src/fsk.c:393-622     the whole per-sample loop, all three framing modes
//   put_frame()                src/fsk.c:352-391
//   dds_complexi()/dds_lookup()  src/dds_int.c       quarter-wave table of 257 int16 entries
//   power_meter_update()       src/power_meter.c:65-69
//
// Layout: state is structure-of-arrays int32 words [28 + 4*span][n_channels] in HBM: 28 scalars, then the
// sliding correlation window as [slot][tone][re, im].  A wave keeps the scalars of its 64 channels in
// registers and the windows in LDS, index-major ([word][lane]: a wave's access to one word of 64 windows
// touches 64 different banks), for the length of the frame.  PCM is read eight samples (16 B) per lane at
// a time from the caller's channel-major buffer.  Events (put_bit() values: bits, SIG_STATUS_* codes,
// framed characters) are appended to a per-channel int16 list.

#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace spg
{

enum
{
    FS_BAUD_RATE = 0,
    FS_FRAMING = 1,         // 0 async, 1 sync, 2 framed (fsk.h:124-129)
    FS_DATA_BITS = 2,
    FS_PARITY = 3,          // 0 none, 1 even, 2 odd, 3 mark, 4 space (async.h:151-157)
    FS_STOP_BITS = 4,
    FS_TOTAL_BITS = 5,
    FS_ON_POWER = 6,
    FS_OFF_POWER = 7,
    FS_POWER = 8,
    FS_LAST_SAMPLE = 9,
    FS_SIGNAL_PRESENT = 10,
    FS_RATE0 = 11,
    FS_RATE1 = 12,
    FS_ACC0 = 13,
    FS_ACC1 = 14,
    FS_SPAN = 15,
    FS_DOT0RE = 16,
    FS_DOT0IM = 17,
    FS_DOT1RE = 18,
    FS_DOT1IM = 19,
    FS_BUF_PTR = 20,
    FS_FRAME_POS = 21,
    FS_FRAME = 22,
    FS_BAUD_PHASE = 23,
    FS_LAST_BIT = 24,
    FS_SHIFT = 25,
    FS_PARITY_ERR = 26,
    FS_FRAMING_ERR = 27,
    kFskScalars = 28
};

constexpr int kFskMaxWindow = 128;
constexpr int kFskRateX100 = 8000*100;

struct FskLaunch
{
    int32_t *st;                // [kFskScalars + 4*span][n_ch]
    const int16_t *pcm;         // [n_ch][stride]
    const int16_t *quarter;     // [257] in HBM
    int16_t *events;            // [n_ch][ev_cap]
    int32_t *ev_count;          // [n_ch]
    long long stride;
    int n_ch;
    int samples;
    const int32_t *lens;        // nullptr, or samples per channel in this call (<= samples; 0 = the channel sits it out)
    int span;
    int ev_cap;
    int vec;                    // rows are 16 B aligned: eight samples per load
};

struct FskRegs
{
    int32_t on_power, off_power, power, last_sample, signal_present;
    int32_t rate0, rate1;
    uint32_t acc0, acc1;
    int32_t dot0re, dot0im, dot1re, dot1im;
    int32_t ptr, frame_pos, frame, baud_phase, last_bit, shift;
    int32_t baud_rate, framing, parity, total_bits, parity_err, framing_err;
    int32_t n_ev;
};

// dds_lookup(), dds_int.c: one quadrant of a sine, mirrored and negated
__device__ __forceinline__ int32_t fsk_lookup(const int16_t *quarter, uint32_t phase)
{
    const uint32_t p = phase >> 22;
    uint32_t step = p & 255u;
    step = (p & 256u)  ?  (256u - step)  :  step;
    const int32_t amp = quarter[step];
    return (p & 512u)  ?  -amp  :  amp;
}

// The kernels look the oscillators up in the same sine unfolded over the whole circle, with the cosine beside it:
// wave[p] = sin(p) in the low half, sin(p + 256) = cos(p) in the high half, p = phase >> 22 (dds_complexi() takes
// the cosine at phase + 2^30, which is p + 256 with no carry from below).  One LDS read and a shift per oscillator
// and sample where the quadrant logic was eight instructions a value; 4 KB per workgroup, filled from the quarter
// wave in HBM at the start of a launch.
constexpr int kFskWave = 1024;

__device__ __forceinline__ void fsk_fill_wave(uint32_t *wave, const int16_t *quarter_hbm, int tid, int n_threads)
{
    for (int p = tid;  p < kFskWave;  p += n_threads)
    {
        const uint32_t sn = (uint32_t) fsk_lookup(quarter_hbm, (uint32_t) p << 22) & 0xFFFFu;
        const uint32_t cs = (uint32_t) fsk_lookup(quarter_hbm, (uint32_t) ((p + 256) & 1023) << 22) & 0xFFFFu;
        wave[p] = sn | (cs << 16);
    }
}

__device__ __forceinline__ void fsk_cos_sin(const uint32_t *wave, uint32_t phase, int32_t &c, int32_t &q)
{
    const uint32_t w = wave[phase >> 22];
    c = (int32_t) w >> 16;
    q = (int32_t) (int16_t) (w & 0xFFFFu);
}

__device__ __forceinline__ void fsk_load_regs(FskRegs &r, const int32_t *st, size_t n)
{
    r.baud_rate = st[FS_BAUD_RATE*n];
    r.framing = st[FS_FRAMING*n];
    r.parity = st[FS_PARITY*n];
    r.total_bits = st[FS_TOTAL_BITS*n];
    r.on_power = st[FS_ON_POWER*n];
    r.off_power = st[FS_OFF_POWER*n];
    r.power = st[FS_POWER*n];
    r.last_sample = st[FS_LAST_SAMPLE*n];
    r.signal_present = st[FS_SIGNAL_PRESENT*n];
    r.rate0 = st[FS_RATE0*n];
    r.rate1 = st[FS_RATE1*n];
    r.acc0 = (uint32_t) st[FS_ACC0*n];
    r.acc1 = (uint32_t) st[FS_ACC1*n];
    r.dot0re = st[FS_DOT0RE*n];
    r.dot0im = st[FS_DOT0IM*n];
    r.dot1re = st[FS_DOT1RE*n];
    r.dot1im = st[FS_DOT1IM*n];
    r.ptr = st[FS_BUF_PTR*n];
    r.frame_pos = st[FS_FRAME_POS*n];
    r.frame = st[FS_FRAME*n];
    r.baud_phase = st[FS_BAUD_PHASE*n];
    r.last_bit = st[FS_LAST_BIT*n];
    r.shift = st[FS_SHIFT*n];
    r.parity_err = st[FS_PARITY_ERR*n];
    r.framing_err = st[FS_FRAMING_ERR*n];
    r.n_ev = 0;
}

__device__ __forceinline__ void fsk_store_regs(const FskRegs &r, int32_t *st, size_t n)
{
    st[FS_POWER*n] = r.power;
    st[FS_LAST_SAMPLE*n] = r.last_sample;
    st[FS_SIGNAL_PRESENT*n] = r.signal_present;
    st[FS_ACC0*n] = (int32_t) r.acc0;
    st[FS_ACC1*n] = (int32_t) r.acc1;
    st[FS_DOT0RE*n] = r.dot0re;
    st[FS_DOT0IM*n] = r.dot0im;
    st[FS_DOT1RE*n] = r.dot1re;
    st[FS_DOT1IM*n] = r.dot1im;
    st[FS_BUF_PTR*n] = r.ptr;
    st[FS_FRAME_POS*n] = r.frame_pos;
    st[FS_FRAME*n] = r.frame;
    st[FS_BAUD_PHASE*n] = r.baud_phase;
    st[FS_LAST_BIT*n] = r.last_bit;
    st[FS_PARITY_ERR*n] = r.parity_err;
    st[FS_FRAMING_ERR*n] = r.framing_err;
}

// The correlation windows of a wave's 64 channels, HBM <-> LDS ([word][lane]); sixteen words at a time, all
// sixteen loads in flight before the first LDS write.
__device__ __forceinline__ void fsk_load_window(int32_t *win, const int32_t *st_win, size_t n, int span, int lane)
{
    for (int w0 = 0;  w0 < 4*span;  w0 += 16)
    {
        int32_t t[16];
#pragma unroll
        for (int k = 0;  k < 16;  k++)
            t[k] = st_win[(size_t) ((w0 + k < 4*span)  ?  (w0 + k)  :  0)*n];
#pragma unroll
        for (int k = 0;  k < 16;  k++)
        {
            if (w0 + k < 4*span)
                win[(w0 + k)*64 + lane] = t[k];
        }
    }
}

__device__ __forceinline__ void fsk_store_window(const int32_t *win, int32_t *st_win, size_t n, int span, int lane)
{
    for (int w = 0;  w < 4*span;  w++)
        st_win[(size_t) w*n] = win[w*64 + lane];
}

// One tone's correlator for one sample, fsk.c:411-423: the product enters the sliding window, the one it
// replaces leaves the running sum; returns |dot|^2 on the scale the comparison uses.  `slot` points at the
// window word of this lane's current slot for this tone (re; im is 64 words on).
__device__ __forceinline__ int32_t fsk_correlate(int32_t &dotre, int32_t &dotim, int32_t *slot, int32_t a, int32_t c, int32_t q, int32_t shift)
{
    const int32_t nre = __mul24(c, a) >> shift;         // 16 bit x 16 bit
    const int32_t nim = __mul24(q, a) >> shift;
    dotre += nre - slot[0];
    dotim += nim - slot[64];
    slot[0] = nre;
    slot[64] = nim;
    const int32_t dr = dotre >> 15;
    const int32_t di = dotim >> 15;
    return __mul24(dr, dr) + __mul24(di, di);           // |dot| < 2^30 (span values of < 2^30/2^shift), so 24 bit multiplies are exact
}

// What the carrier detector made of a sample (fsk.c:425-475).  DROP, QUIET and COUNTING are the three places
// the reference `continue`s (the window slot stays where it is); RISE goes on to the bit clock.
enum
{
    FSK_KIND_NORMAL = 0,
    FSK_KIND_DROP = 1,          // SIG_STATUS_CARRIER_DOWN
    FSK_KIND_QUIET = 2,
    FSK_KIND_COUNTING = 3,
    FSK_KIND_RISE = 4,          // SIG_STATUS_CARRIER_UP
    FSK_KIND_NONE = 5           // no sample (past the end of this channel's frame)
};

// Power behind a one-tap DC blocker and the carrier detector, as selects (the branches are data dependent per
// lane and nearly all of them just move a counter).  `count_phase` is baud_phase in its role as the counter of
// the samples a returning carrier has lasted (it is only read while no carrier is present).
__device__ __forceinline__ int fsk_carrier(int32_t &power, int32_t &last_sample, int32_t &signal_present, int32_t &count_phase,
                                           int32_t on_power, int32_t off_power, int span, int32_t a)
{
    const int32_t x = a >> 1;
    const int32_t diff = (int32_t) (int16_t) (x - last_sample);
    power += (__mul24(diff, diff) - power) >> 4;
    last_sample = x;
    const bool present = (signal_present != 0);
    const bool low_off = (power < off_power);
    const bool low_on = (power < on_power);
    const bool dec = present  &&  low_off;
    const int32_t sp1 = signal_present - 1;
    const bool drop = dec  &&  (sp1 <= 0);
    const bool quiet = !present  &&  low_on;
    const bool counting = !present  &&  !low_on  &&  (count_phase < (span >> 1) - 30);
    const bool rise = !present  &&  !low_on  &&  !counting;
    signal_present = dec  ?  sp1  :  (rise  ?  1  :  signal_present);
    count_phase = (drop  ||  quiet  ||  rise)  ?  0  :  (counting  ?  (count_phase + 1)  :  count_phase);
    int kind = FSK_KIND_NORMAL;
    kind = drop  ?  FSK_KIND_DROP  :  kind;
    kind = quiet  ?  FSK_KIND_QUIET  :  kind;
    kind = counting  ?  FSK_KIND_COUNTING  :  kind;
    kind = rise  ?  FSK_KIND_RISE  :  kind;
    return kind;
}

struct FskBits
{
    int32_t frame_pos, frame, baud_phase, last_bit;
    int32_t baud_rate, framing, parity, total_bits, parity_err, framing_err;
};

// The carrier detector's effect on the bit clock's words, then -- for the samples that get that far -- the bit
// clock and the framing, fsk.c:476-614.  Returns false where the reference `continue`s.
//
// FRAMED = false is the copy for a wave none of whose channels frames characters (bit-synchronous and asynchronous
// receivers): a sample is straight-line code there but for the rare events, which sit out of line.  A lone wave pays
// every taken branch with an empty instruction buffer, and the framing tree is a dozen of them per sample even when
// no lane enters it.
template <bool FRAMED, class Emit>
__device__ __forceinline__ bool fsk_bits(FskBits &r, int kind, int state, Emit &&emit)
{
    const bool drop = (kind == FSK_KIND_DROP);
    const bool quiet = (kind == FSK_KIND_QUIET);
    const bool counting = (kind == FSK_KIND_COUNTING);
    const bool rise = (kind == FSK_KIND_RISE);
    r.baud_phase = (drop  ||  quiet  ||  rise)  ?  0  :  (counting  ?  (r.baud_phase + 1)  :  r.baud_phase);
    r.frame_pos = rise  ?  -2  :  r.frame_pos;
    r.frame = rise  ?  0  :  r.frame;
    r.last_bit = rise  ?  0  :  r.last_bit;
    if (__builtin_expect(drop, 0))
        emit(-1);                               // SIG_STATUS_CARRIER_DOWN
    if (__builtin_expect(rise, 0))
        emit(-2);                               // SIG_STATUS_CARRIER_UP
    if (!FRAMED)
    {
        // (the same arithmetic as the branch below, with a sample that stops here leaving every word as it is)
        const bool go = !(drop  ||  quiet  ||  counting);
        const bool change = (r.last_bit != state);
        const int32_t eighth = r.baud_rate >> 3;
        const int32_t nudged = r.baud_phase + ((r.baud_phase < kFskRateX100/2)  ?  eighth  :  -eighth);
        const int32_t moved = (r.framing == 1)  ?  nudged  :  (kFskRateX100/2);
        const int32_t bp = (change  ?  moved  :  r.baud_phase) + r.baud_rate;
        const bool fire = (bp >= kFskRateX100);
        r.last_bit = go  ?  state  :  r.last_bit;
        r.baud_phase = go  ?  (fire  ?  (bp - kFskRateX100)  :  bp)  :  r.baud_phase;
        if (__builtin_expect(go  &&  fire, 0))
            emit(state);
        return go;
    }
    if (drop  ||  quiet  ||  counting)
        return false;
    if (r.framing != 2)
    {
        // synchronous (fsk.c:489-512): a transition nudges the baud phase towards the middle of the baud;
        // asynchronous (fsk.c:513-537): a transition sets it there.  Then one bit per baud.
        const bool change = (r.last_bit != state);
        const int32_t eighth = r.baud_rate >> 3;
        const int32_t nudged = r.baud_phase + ((r.baud_phase < kFskRateX100/2)  ?  eighth  :  -eighth);
        const int32_t moved = (r.framing == 1)  ?  nudged  :  (kFskRateX100/2);
        r.last_bit = state;
        const int32_t bp = (change  ?  moved  :  r.baud_phase) + r.baud_rate;
        const bool fire = (bp >= kFskRateX100);
        r.baud_phase = fire  ?  (bp - kFskRateX100)  :  bp;
        if (__builtin_expect(fire, 0))
            emit(state);
    }
    else if (r.frame_pos == -2)
    {
        // framed, fsk.c:538-614: hunting for a start bit
        if (state == 0)
        {
            r.baud_phase = 8000*(100 - 40)/2;
            r.frame_pos = -1;
            r.frame = 0;
            r.last_bit = -1;
        }
    }
    else if (r.frame_pos == -1)
    {
        if (state != 0)
        {
            r.frame_pos = -2;
        }
        else
        {
            r.baud_phase += r.baud_rate;
            if (r.baud_phase >= kFskRateX100)
            {
                r.frame_pos = 0;
                r.last_bit = state;
            }
        }
    }
    else
    {
        r.baud_phase += r.baud_rate;
        if (r.baud_phase >= 8000*(100 - 40))
        {
            if (r.last_bit < 0)
                r.last_bit = state;
            if (r.last_bit != state)
            {
                r.frame_pos = -2;
                r.framing_err++;
            }
            else if (r.baud_phase >= kFskRateX100)
            {
                if (r.frame_pos++ > r.total_bits)
                {
                    if (state == 1)
                    {
                        // put_frame(), fsk.c:352-391
                        uint32_t frame = (uint32_t) r.frame & 0xFFFFu;
                        if (r.parity != 0)
                        {
                            const uint32_t sent = (frame >> 15) & 1u;
                            frame = (frame & 0x7FFFu) >> (16 - r.total_bits);
                            uint32_t x8 = frame & 0xFFu;
                            x8 = (x8 ^ (x8 >> 4)) & 0x0Fu;
                            x8 = (0x6996u >> x8) & 1u;
                            uint32_t want = 0u;                 // ASYNC_PARITY_SPACE
                            want = (r.parity == 2)  ?  (x8 ^ 1u)  :  want;
                            want = (r.parity == 1)  ?  x8  :  want;
                            want = (r.parity == 3)  ?  1u  :  want;
                            if (sent == want)
                                emit((int) frame);
                            else
                                r.parity_err++;
                        }
                        else
                        {
                            emit((int) (frame >> (16 - r.total_bits)));
                        }
                    }
                    else
                    {
                        r.framing_err++;
                    }
                    r.frame_pos = -2;
                }
                else
                {
                    r.frame = ((r.frame >> 1) | (state << 15)) & 0xFFFF;
                }
                r.baud_phase -= kFskRateX100;
                r.last_bit = -1;
            }
        }
    }
    return true;
}

// One sample of fsk_rx(), fsk.c:408-618, for the kernels that run a whole receiver in one lane; (c0, q0) and
// (c1, q1) are the two oscillators' cos / sin for this sample.  emit(v) stands for put_bit(v).  Returns without
// advancing the window slot where the reference `continue`s.
template <bool FRAMED, class Emit>
__device__ __forceinline__ void fsk_step(FskRegs &r, int32_t *win, int lane, int span, int32_t a, int32_t c0, int32_t q0,
                                         int32_t c1, int32_t q1, Emit &&emit)
{
    int32_t *slot = win + (r.ptr*4)*64 + lane;
    const int32_t sum0 = fsk_correlate(r.dot0re, r.dot0im, slot, a, c0, q0, r.shift);
    const int32_t sum1 = fsk_correlate(r.dot1re, r.dot1im, slot + 128, a, c1, q1, r.shift);
    int32_t count_phase = r.baud_phase;
    const int kind = fsk_carrier(r.power, r.last_sample, r.signal_present, count_phase, r.on_power, r.off_power, span, a);
    FskBits b = {r.frame_pos, r.frame, r.baud_phase, r.last_bit, r.baud_rate, r.framing, r.parity, r.total_bits, r.parity_err, r.framing_err};
    const bool on = fsk_bits<FRAMED>(b, kind, (sum0 < sum1)  ?  1  :  0, emit);
    r.frame_pos = b.frame_pos;
    r.frame = b.frame;
    r.baud_phase = b.baud_phase;
    r.last_bit = b.last_bit;
    r.parity_err = b.parity_err;
    r.framing_err = b.framing_err;
    const int32_t next = (r.ptr + 1 >= span)  ?  0  :  (r.ptr + 1);
    r.ptr = on  ?  next  :  r.ptr;
}

// The look-ups of a block of up to eight samples, issued together ahead of the sample-serial part: the two
// oscillators run on whatever the receiver does with a sample (dds_complexi() comes before any early out).
__device__ __forceinline__ void fsk_block_lookups(FskRegs &r, const uint32_t *wave, int todo, int32_t (&c0)[8], int32_t (&q0)[8],
                                                  int32_t (&c1)[8], int32_t (&q1)[8])
{
#pragma unroll
    for (int k = 0;  k < 8;  k++)
    {
        const uint32_t p0 = r.acc0 + (uint32_t) k*(uint32_t) r.rate0;
        const uint32_t p1 = r.acc1 + (uint32_t) k*(uint32_t) r.rate1;
        fsk_cos_sin(wave, p0, c0[k], q0[k]);
        fsk_cos_sin(wave, p1, c1[k], q1[k]);
    }
    r.acc0 += (uint32_t) todo*(uint32_t) r.rate0;
    r.acc1 += (uint32_t) todo*(uint32_t) r.rate1;
}

// A lane's row of the frame, eight samples (one 16-byte load) at a time, the next eight requested before the current
// ones are worked on: a wave has nothing else to do while a load is on its way (a bank of 65 536 channels is one or two
// waves per SIMD), and twenty loads from HBM one after the other are a third of a launch.
template <bool VEC>        // rows are 16 B aligned (a launch's property: the two cases are separate copies of the sample loop)
struct FskRow
{
    const int16_t *row;
    int mylen;
    uint4 q_next;
};

template <bool VEC>
__device__ __forceinline__ void fsk_row_begin(FskRow<VEC> &w, const int16_t *row, int mylen)
{
    w.row = row;
    w.mylen = mylen;
    w.q_next = make_uint4(0u, 0u, 0u, 0u);
    // every load of the prologue home first (vmcnt(0)): the counter is in order, so a state word still on its way at the
    // top of the sample loop would have the compiler wait for *everything* there on every pass -- the next block included
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (VEC)
        w.q_next = *reinterpret_cast<const uint4 *>(row);
}

// Samples base .. base + todo - 1 (todo <= 8; blocks are taken in order; what lies past todo is not used).  With aligned
// rows every block is one aligned 16-byte load, also a frame's last, shorter one: it starts inside the row, so it cannot
// leave the page the row's last sample is on.  No lane-dependent branch around the loads, and no other kind of load in
// the same loop: the compiler waits for everything in flight where such paths join, which would put the wait for the
// next block right behind its request.
template <bool VEC>
__device__ __forceinline__ void fsk_row_block(FskRow<VEC> &w, int base, int todo, int32_t (&a)[8])
{
    if (VEC)
    {
        const uint4 q = w.q_next;
        a[0] = (int32_t) (int16_t) (q.x & 0xFFFFu);
        a[1] = (int32_t) q.x >> 16;
        a[2] = (int32_t) (int16_t) (q.y & 0xFFFFu);
        a[3] = (int32_t) q.y >> 16;
        a[4] = (int32_t) (int16_t) (q.z & 0xFFFFu);
        a[5] = (int32_t) q.z >> 16;
        a[6] = (int32_t) (int16_t) (q.w & 0xFFFFu);
        a[7] = (int32_t) q.w >> 16;
        const int next = (base + 8 < w.mylen)  ?  (base + 8)  :  0;
        w.q_next = *reinterpret_cast<const uint4 *>(w.row + next);
    }
    else
    {
#pragma unroll
        for (int k = 0;  k < 8;  k++)
            a[k] = (k < todo)  ?  (int32_t) w.row[base + k]  :  0;
    }
}

// ---- A receiver over two waves -----------------------------------------------------------------------------
// A lone wave issues this kind of code (short dependent chains of selects, LDS reads, 24-bit products) at about
// half the rate two waves reach together, and a bank of 65 536 channels is one wave per SIMD.  So the receiver
// is cut into two instruction streams of about equal length that run as two waves of one workgroup on the same
// 64 channels:
//   the signal side   tone 0's oscillator and correlator, the power meter and the carrier detector;
//   the bit side      tone 1's oscillator and correlator, the comparison, the bit clock, the framing, put_bit().
// The carrier detector depends on the samples only, so the signal side never waits for the bit side.  It hands
// over, per block of eight samples, what the detector made of each sample (a FSK_KIND_* nibble) and tone 0's
// |dot|^2; the bit side follows one block behind (two message buffers in LDS, one barrier per block).  Each
// side owns its half of every window slot and of the state words, in LDS and in HBM.

constexpr int kFskMsgWords = 9;             // per lane and block: the kinds, then eight sums

struct FskSigSide
{
    int32_t on_power, off_power, power, last_sample, signal_present, count_phase;
    int32_t rate, dotre, dotim, ptr, shift;
    uint32_t acc;
};

struct FskBitSide
{
    int32_t rate, dotre, dotim, ptr, shift;
    uint32_t acc;
    FskBits b;
    int32_t n_ev;
};

// One side's half of the windows (tone = 0 or 1): words 4*slot + 2*tone + {0, 1}
__device__ __forceinline__ void fsk_load_window_half(int32_t *win, const int32_t *st_win, size_t n, int span, int lane, int tone)
{
    for (int s0 = 0;  s0 < span;  s0 += 8)
    {
        int32_t t[16];
#pragma unroll
        for (int k = 0;  k < 16;  k++)
        {
            const int w = 4*(s0 + (k >> 1)) + 2*tone + (k & 1);
            t[k] = st_win[(size_t) ((s0 + (k >> 1) < span)  ?  w  :  0)*n];
        }
#pragma unroll
        for (int k = 0;  k < 16;  k++)
        {
            const int w = 4*(s0 + (k >> 1)) + 2*tone + (k & 1);
            if (s0 + (k >> 1) < span)
                win[w*64 + lane] = t[k];
        }
    }
}

__device__ __forceinline__ void fsk_store_window_half(const int32_t *win, int32_t *st_win, size_t n, int span, int lane, int tone)
{
    for (int s = 0;  s < span;  s++)
    {
        const int w = 4*s + 2*tone;
        st_win[(size_t) w*n] = win[w*64 + lane];
        st_win[(size_t) (w + 1)*n] = win[(w + 1)*64 + lane];
    }
}

__device__ __forceinline__ void fsk_side_lookups(uint32_t &acc, int32_t rate, const uint32_t *wave, int todo, int32_t (&c)[8], int32_t (&q)[8])
{
#pragma unroll
    for (int k = 0;  k < 8;  k++)
    {
        const uint32_t p = acc + (uint32_t) k*(uint32_t) rate;
        fsk_cos_sin(wave, p, c[k], q[k]);
    }
    acc += (uint32_t) todo*(uint32_t) rate;
}

__device__ __forceinline__ void fsk_sig_load(FskSigSide &s, const int32_t *st, size_t n)
{
    s.on_power = st[FS_ON_POWER*n];
    s.off_power = st[FS_OFF_POWER*n];
    s.power = st[FS_POWER*n];
    s.last_sample = st[FS_LAST_SAMPLE*n];
    s.signal_present = st[FS_SIGNAL_PRESENT*n];
    s.count_phase = st[FS_BAUD_PHASE*n];
    s.rate = st[FS_RATE0*n];
    s.acc = (uint32_t) st[FS_ACC0*n];
    s.dotre = st[FS_DOT0RE*n];
    s.dotim = st[FS_DOT0IM*n];
    s.ptr = st[FS_BUF_PTR*n];
    s.shift = st[FS_SHIFT*n];
}

__device__ __forceinline__ void fsk_sig_store(const FskSigSide &s, int32_t *st, size_t n)
{
    st[FS_POWER*n] = s.power;
    st[FS_LAST_SAMPLE*n] = s.last_sample;
    st[FS_SIGNAL_PRESENT*n] = s.signal_present;
    st[FS_ACC0*n] = (int32_t) s.acc;
    st[FS_DOT0RE*n] = s.dotre;
    st[FS_DOT0IM*n] = s.dotim;
    st[FS_BUF_PTR*n] = s.ptr;
}

__device__ __forceinline__ void fsk_bit_load(FskBitSide &t, const int32_t *st, size_t n)
{
    t.rate = st[FS_RATE1*n];
    t.acc = (uint32_t) st[FS_ACC1*n];
    t.dotre = st[FS_DOT1RE*n];
    t.dotim = st[FS_DOT1IM*n];
    t.ptr = st[FS_BUF_PTR*n];
    t.shift = st[FS_SHIFT*n];
    t.b.frame_pos = st[FS_FRAME_POS*n];
    t.b.frame = st[FS_FRAME*n];
    t.b.baud_phase = st[FS_BAUD_PHASE*n];
    t.b.last_bit = st[FS_LAST_BIT*n];
    t.b.baud_rate = st[FS_BAUD_RATE*n];
    t.b.framing = st[FS_FRAMING*n];
    t.b.parity = st[FS_PARITY*n];
    t.b.total_bits = st[FS_TOTAL_BITS*n];
    t.b.parity_err = st[FS_PARITY_ERR*n];
    t.b.framing_err = st[FS_FRAMING_ERR*n];
    t.n_ev = 0;
}

__device__ __forceinline__ void fsk_bit_store(const FskBitSide &t, int32_t *st, size_t n)
{
    st[FS_ACC1*n] = (int32_t) t.acc;
    st[FS_DOT1RE*n] = t.dotre;
    st[FS_DOT1IM*n] = t.dotim;
    st[FS_FRAME_POS*n] = t.b.frame_pos;
    st[FS_FRAME*n] = t.b.frame;
    st[FS_BAUD_PHASE*n] = t.b.baud_phase;
    st[FS_LAST_BIT*n] = t.b.last_bit;
    st[FS_PARITY_ERR*n] = t.b.parity_err;
    st[FS_FRAMING_ERR*n] = t.b.framing_err;
}

// The signal side's block: up to eight samples of the lane's row from `base`; msg = this block's message words
template <bool VEC>
__device__ __forceinline__ void fsk_sig_block(FskSigSide &s, int32_t *win, const uint32_t *wave, int32_t *msg, int lane, int span,
                                              FskRow<VEC> &row, int base, int todo)
{
    int32_t a[8];
    int32_t c[8];
    int32_t q[8];
    fsk_row_block(row, base, todo, a);
    fsk_side_lookups(s.acc, s.rate, wave, todo, c, q);
    uint32_t kinds = 0;
    auto sample = [&](int k)
    {
        const int32_t sum = fsk_correlate(s.dotre, s.dotim, win + (s.ptr*4)*64 + lane, a[k], c[k], q[k], s.shift);
        const int kind = fsk_carrier(s.power, s.last_sample, s.signal_present, s.count_phase, s.on_power, s.off_power, span, a[k]);
        msg[(1 + k)*64 + lane] = sum;
        const bool on = (kind == FSK_KIND_NORMAL  ||  kind == FSK_KIND_RISE);
        const int32_t next = (s.ptr + 1 >= span)  ?  0  :  (s.ptr + 1);
        s.ptr = on  ?  next  :  s.ptr;
        kinds |= (uint32_t) kind << (4*k);
    };
    // (a whole block in every lane is the common case: no per-sample guard, whose bodies the compiler moves out of line)
    if (__builtin_expect(__all(todo == 8), 1))
    {
#pragma unroll
        for (int k = 0;  k < 8;  k++)
            sample(k);
    }
    else
    {
#pragma unroll
        for (int k = 0;  k < 8;  k++)
        {
            if (k < todo)
                sample(k);
            else
                kinds |= (uint32_t) FSK_KIND_NONE << (4*k);
        }
    }
    msg[lane] = (int32_t) kinds;
}

// The bit side's block, from the signal side's message for the same eight samples
template <bool VEC, bool FRAMED, class Emit>
__device__ __forceinline__ void fsk_bit_block(FskBitSide &t, int32_t *win, const uint32_t *wave, const int32_t *msg, int lane, int span,
                                              FskRow<VEC> &row, int base, int todo, Emit &&emit)
{
    int32_t a[8];
    int32_t c[8];
    int32_t q[8];
    fsk_row_block(row, base, todo, a);
    fsk_side_lookups(t.acc, t.rate, wave, todo, c, q);
    const uint32_t kinds = (uint32_t) msg[lane];
    if (__builtin_expect(__all(todo == 8), 1))
    {
        // (the message taken whole: its reads would not move across the window writes by themselves)
        int32_t sums[8];
#pragma unroll
        for (int k = 0;  k < 8;  k++)
            sums[k] = msg[(1 + k)*64 + lane];
#pragma unroll
        for (int k = 0;  k < 8;  k++)
        {
            const int32_t sum1 = fsk_correlate(t.dotre, t.dotim, win + (t.ptr*4 + 2)*64 + lane, a[k], c[k], q[k], t.shift);
            const int32_t sum0 = sums[k];
            const bool on = fsk_bits<FRAMED>(t.b, (int) ((kinds >> (4*k)) & 7u), (sum0 < sum1)  ?  1  :  0, emit);
            const int32_t next = (t.ptr + 1 >= span)  ?  0  :  (t.ptr + 1);
            t.ptr = on  ?  next  :  t.ptr;
        }
    }
    else
    {
#pragma unroll
        for (int k = 0;  k < 8;  k++)
        {
            if (k < todo)
            {
                const int32_t sum1 = fsk_correlate(t.dotre, t.dotim, win + (t.ptr*4 + 2)*64 + lane, a[k], c[k], q[k], t.shift);
                const int32_t sum0 = msg[(1 + k)*64 + lane];
                const bool on = fsk_bits<true>(t.b, (int) ((kinds >> (4*k)) & 7u), (sum0 < sum1)  ?  1  :  0, emit);
                const int32_t next = (t.ptr + 1 >= span)  ?  0  :  (t.ptr + 1);
                t.ptr = on  ?  next  :  t.ptr;
            }
        }
    }
}

}   // namespace spg
