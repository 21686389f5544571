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

// One sample of fsk_rx(), fsk.c:408-618; (c0, q0) and (c1, q1) are the two oscillators' cos / sin for this
// sample.  emit(v) stands for put_bit(v).  Returns without advancing the window slot where the reference
// `continue`s.
template <class Emit>
__device__ __forceinline__ void fsk_step(FskRegs &r, int32_t *win, int lane, int span, int32_t a, int32_t c0, int32_t q0,
                                         int32_t c1, int32_t q1, Emit &&emit)
{
    int32_t *slot = win + (r.ptr*4)*64 + lane;
    int32_t sum0;
    int32_t sum1;
    {
        const int32_t c = c0;
        const int32_t q = q0;
        const int32_t nre = __mul24(c, a) >> r.shift;        // 16 bit x 16 bit
        const int32_t nim = __mul24(q, a) >> r.shift;
        r.dot0re += nre - slot[0];
        r.dot0im += nim - slot[64];
        slot[0] = nre;
        slot[64] = nim;
        const int32_t dr = r.dot0re >> 15;
        const int32_t di = r.dot0im >> 15;
        sum0 = __mul24(dr, dr) + __mul24(di, di);       // |dot| < 2^30 (span values of < 2^30/2^shift), so 24 bit multiplies are exact
    }
    {
        const int32_t c = c1;
        const int32_t q = q1;
        const int32_t nre = __mul24(c, a) >> r.shift;        // 16 bit x 16 bit
        const int32_t nim = __mul24(q, a) >> r.shift;
        r.dot1re += nre - slot[128];
        r.dot1im += nim - slot[192];
        slot[128] = nre;
        slot[192] = nim;
        const int32_t dr = r.dot1re >> 15;
        const int32_t di = r.dot1im >> 15;
        sum1 = __mul24(dr, dr) + __mul24(di, di);
    }
    // power behind a one-tap DC blocker, fsk.c:425-431
    const int32_t x = a >> 1;
    const int32_t diff = (int32_t) (int16_t) (x - r.last_sample);
    r.power += (__mul24(diff, diff) - r.power) >> 4;
    r.last_sample = x;
    // Carrier detect, fsk.c:433-475, as selects (the branches are data dependent per lane and nearly all of
    // them just move a counter).  drop / quiet / counting are the three places the reference `continue`s.
    const bool present = (r.signal_present != 0);
    const bool low_off = (r.power < r.off_power);
    const bool low_on = (r.power < r.on_power);
    const bool dec = present  &&  low_off;
    const int32_t sp1 = r.signal_present - 1;
    const bool drop = dec  &&  (sp1 <= 0);
    const bool quiet = !present  &&  low_on;
    const bool counting = !present  &&  !low_on  &&  (r.baud_phase < (span >> 1) - 30);
    const bool rise = !present  &&  !low_on  &&  !counting;
    r.signal_present = dec  ?  sp1  :  (rise  ?  1  :  r.signal_present);
    r.baud_phase = (drop  ||  quiet  ||  rise)  ?  0  :  (counting  ?  (r.baud_phase + 1)  :  r.baud_phase);
    r.frame_pos = rise  ?  -2  :  r.frame_pos;
    r.frame = rise  ?  0  :  r.frame;
    r.last_bit = rise  ?  0  :  r.last_bit;
    if (drop)
        emit(-1);                               // SIG_STATUS_CARRIER_DOWN
    if (rise)
        emit(-2);                               // SIG_STATUS_CARRIER_UP
    if (drop  ||  quiet  ||  counting)
        return;
    const int state = (sum0 < sum1)  ?  1  :  0;
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
        if (fire)
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
    r.ptr = (r.ptr + 1 >= span)  ?  0  :  (r.ptr + 1);
}

// The look-ups of a block of up to eight samples, issued together ahead of the sample-serial part: the two
// oscillators run on whatever the receiver does with a sample (dds_complexi() comes before any early out).
__device__ __forceinline__ void fsk_block_lookups(FskRegs &r, const int16_t *quarter, int todo, int32_t (&c0)[8], int32_t (&q0)[8],
                                                  int32_t (&c1)[8], int32_t (&q1)[8])
{
#pragma unroll
    for (int k = 0;  k < 8;  k++)
    {
        const uint32_t p0 = r.acc0 + (uint32_t) k*(uint32_t) r.rate0;
        const uint32_t p1 = r.acc1 + (uint32_t) k*(uint32_t) r.rate1;
        c0[k] = fsk_lookup(quarter, p0 + (1u << 30));
        q0[k] = fsk_lookup(quarter, p0);
        c1[k] = fsk_lookup(quarter, p1 + (1u << 30));
        q1[k] = fsk_lookup(quarter, p1);
    }
    r.acc0 += (uint32_t) todo*(uint32_t) r.rate0;
    r.acc1 += (uint32_t) todo*(uint32_t) r.rate1;
}

// Eight samples of a lane's row (fewer at the end of a frame; the rest read as 0 and are not used)
__device__ __forceinline__ void fsk_block_samples(const int16_t *row, int base, int todo, bool vec, int32_t (&a)[8])
{
    if (vec  &&  todo == 8)
    {
        const uint4 q = *reinterpret_cast<const uint4 *>(row + base);
        a[0] = (int32_t) (int16_t) (q.x & 0xFFFFu);
        a[1] = (int32_t) q.x >> 16;
        a[2] = (int32_t) (int16_t) (q.y & 0xFFFFu);
        a[3] = (int32_t) q.y >> 16;
        a[4] = (int32_t) (int16_t) (q.z & 0xFFFFu);
        a[5] = (int32_t) q.z >> 16;
        a[6] = (int32_t) (int16_t) (q.w & 0xFFFFu);
        a[7] = (int32_t) q.w >> 16;
    }
    else
    {
#pragma unroll
        for (int k = 0;  k < 8;  k++)
            a[k] = (k < todo)  ?  (int32_t) row[base + k]  :  0;
    }
}

}   // namespace spg
