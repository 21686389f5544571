// awgn_dev.hpp -- device side of the noise source banks: batched awgn() (reference src/awgn.c:82-195).
//
// One lane per channel, one wave per workgroup.  A channel's generator is three small LCGs feeding a 97 entry
// shuffle table of doubles and a polar Box-Muller transform in binary64; the table is indexed by the generator's own
// output, so it lives in LDS for the launch ([entry][lane], 97*64*8 = 49 664 bytes per wave) and goes back to HBM at
// the end.  Everything is IEEE binary64 arithmetic rounded operation by operation exactly as the reference's compiled
// code rounds it (-ffp-contract=off), so the accept / reject sequence and the table are identical by construction;
// log(), the one library call of the path, is GNU libc's routine restated bit for bit (glibc_log_dev.hpp), its table in
// LDS beside the shuffle table.  Every int16 and the carried amp2 equal the reference's.
//
// State words of a channel (word-major, [word][channel]), the layout of the reference's awgn_state_t fields in use:
//   0,1 rms   2,3 amp2 (doubles, low word first)   4 odd   5 ix1   6 ix2   7 ix3   8.. r[97] (doubles)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "glibc_log_dev.hpp"

namespace spg
{

enum
{
    AW_RMS = 0,
    AW_AMP2 = 2,
    AW_ODD = 4,
    AW_IX1,
    AW_IX2,
    AW_IX3,
    AW_R,
    kAwgnWords = AW_R + 2*97
};

#ifndef SPG_AWGN_WAITING
#define SPG_AWGN_WAITING 4
#endif
constexpr int kAwgnWaiting = SPG_AWGN_WAITING;     // accepted candidates a lane may hold ahead of the wave (awgn_bank_kernel)

struct AwgnLaunch
{
    int32_t *st;
    int16_t *amp;               // [channel][stride]
    long long stride;
    int n_ch;
    int samples;
    int mix;                    // 0: amp[i] = awgn();  1: amp[i] = saturate16(amp[i] + awgn())
};

struct AwgnRegs
{
    double rms;
    int32_t ix1;
    int32_t ix2;
    int32_t ix3;
    const double *logtab;       // {invc, logc} pairs in LDS
};

// ran1(): awgn.c:107-131
__device__ __forceinline__ double awgn_uniform(AwgnRegs &g, double *col)
{
    g.ix1 = (7141*g.ix1 + 54773)%259200;
    g.ix2 = (8121*g.ix2 + 28411)%134456;
    g.ix3 = (4561*g.ix3 + 51349)%243000;
    const int j = (97*g.ix3)/243000;
    const double t = col[j*64];
    col[j*64] = ((double) g.ix1 + (double) g.ix2*(1.0/134456.0))*(1.0/259200.0);
    return t;
}

// One Box-Muller pair: awgn.c:178-191.  first = v2*r (returned by the generating call), second = v1*r (amp2).
__device__ __forceinline__ void awgn_pair(AwgnRegs &g, double *col, double &first, double &second)
{
    double v1;
    double v2;
    double r;
    do
    {
        v1 = 2.0*awgn_uniform(g, col) - 1.0;
        v2 = 2.0*awgn_uniform(g, col) - 1.0;
        r = v1*v1 + v2*v2;
    }
    while (r >= 1.0);
    r = sqrt(-2.0*glibc_log(r, g.logtab)/r);
    second = v1*r;
    first = v2*r;
}

// fsaturate(amp*rms): saturated.h:152-159
__device__ __forceinline__ int awgn_sample(AwgnRegs &g, double amp)
{
    amp *= g.rms;
    if (amp > 32767.0)
        return 32767;
    if (amp < -32768.0)
        return -32768;
    if (amp != amp)
        return 0;               // (int16_t) lrint(NaN) on the reference's host
    return (int) rint(amp);
}

__device__ __forceinline__ void awgn_put(const AwgnLaunch &L, int16_t *row, int i, int v)
{
    if (L.mix)
    {
        v += row[i];
        v = (v > 32767)  ?  32767  :  (v < -32768)  ?  -32768  :  v;
    }
    row[i] = (int16_t) v;
}

__global__ __launch_bounds__(64) void awgn_bank_kernel(AwgnLaunch L)
{
    extern __shared__ double awgn_tab[];        // [97][64], then the 128 {invc, logc} pairs of log()

    const int lane = threadIdx.x;
    const int c = blockIdx.x*64 + lane;
    double *logtab = awgn_tab + 97*64;
    for (int j = lane;  j < 2*kLogTab;  j += 64)
        logtab[j] = g_log_tab[j];
    __syncthreads();
    if (c >= L.n_ch)
        return;
    const size_t N = (size_t) L.n_ch;
    int32_t *st = L.st + c;
    double *col = awgn_tab + lane;
    for (int j = 0;  j < 97;  j++)
        col[j*64] = __hiloint2double(st[(AW_R + 2*j + 1)*N], st[(AW_R + 2*j)*N]);
    AwgnRegs g;
    g.rms = __hiloint2double(st[(AW_RMS + 1)*N], st[AW_RMS*N]);
    double amp2 = __hiloint2double(st[(AW_AMP2 + 1)*N], st[AW_AMP2*N]);
    int odd = st[AW_ODD*N];
    g.ix1 = st[AW_IX1*N];
    g.ix2 = st[AW_IX2*N];
    g.ix3 = st[AW_IX3*N];
    g.logtab = logtab;

    int16_t *row = L.amp + (size_t) c*L.stride;
    int i = 0;
    if (odd == 0  &&  L.samples > 0)
    {
        // the second half of a pair made by the previous call
        awgn_put(L, row, i++, awgn_sample(g, amp2));
        odd = 1;
    }
    // The pairs of this call.  The polar method rejects 21 % of its candidates, and a wave that makes every pair in step
    // repeats each until its slowest lane accepts (3.7 rounds a pair against 1.27 on average).  So the two halves of a pair
    // are uncoupled: a lane draws candidates -- its generator's calls in the reference's order, whatever the other lanes do
    // -- and keeps up to kAwgnWaiting accepted ones waiting; the expensive half (log, divide, square root, scaling, rounding,
    // the stores) runs when every lane has one waiting, for all lanes at once.
    const int pairs = (L.samples - i + 1) >> 1;         // pairs this lane still makes (the last one may be used by half)
    int made = 0;
    int drawn = 0;
    int waiting = 0;
    double w1[kAwgnWaiting];
    double w2[kAwgnWaiting];
    double wr[kAwgnWaiting];
#pragma unroll
    for (int k = 0;  k < kAwgnWaiting;  k++)
    {
        w1[k] = 0.0;
        w2[k] = 0.0;
        wr[k] = 1.0;
    }
    while (__any(made < pairs))
    {
        if (drawn < pairs  &&  waiting < kAwgnWaiting)
        {
            const double v1 = 2.0*awgn_uniform(g, col) - 1.0;
            const double v2 = 2.0*awgn_uniform(g, col) - 1.0;
            const double r = v1*v1 + v2*v2;
            if (!(r >= 1.0))
            {
#pragma unroll
                for (int k = 0;  k < kAwgnWaiting;  k++)
                {
                    const bool here = (waiting == k);
                    w1[k] = here  ?  v1  :  w1[k];
                    w2[k] = here  ?  v2  :  w2[k];
                    wr[k] = here  ?  r  :  wr[k];
                }
                waiting++;
                drawn++;
            }
        }
        if (__all(made >= pairs  ||  waiting >= 1))
        {
            if (made < pairs)
            {
                const double r = sqrt(-2.0*glibc_log(wr[0], g.logtab)/wr[0]);
                const double first = w2[0]*r;
                amp2 = w1[0]*r;
                awgn_put(L, row, i, awgn_sample(g, first));
                if (i + 1 < L.samples)
                    awgn_put(L, row, i + 1, awgn_sample(g, amp2));
                else
                    odd = 0;                            // the pair's second half waits for the next call
                i += 2;
                made++;
#pragma unroll
                for (int k = 0;  k + 1 < kAwgnWaiting;  k++)
                {
                    w1[k] = w1[k + 1];
                    w2[k] = w2[k + 1];
                    wr[k] = wr[k + 1];
                }
                waiting--;
            }
        }
    }

    for (int j = 0;  j < 97;  j++)
    {
        const double v = col[j*64];
        st[(AW_R + 2*j)*N] = __double2loint(v);
        st[(AW_R + 2*j + 1)*N] = __double2hiint(v);
    }
    st[AW_AMP2*N] = __double2loint(amp2);
    st[(AW_AMP2 + 1)*N] = __double2hiint(amp2);
    st[AW_ODD*N] = odd;
    st[AW_IX1*N] = g.ix1;
    st[AW_IX2*N] = g.ix2;
    st[AW_IX3*N] = g.ix3;
}

}   // namespace spg
