// txgen_dev.hpp -- device side of the signal-source banks: N cadenced
// multi-tone generators, optionally fed by a per-channel digit queue, one channel per lane.
//
// What is restated (reference paths relative to the reference tree, float build, x86-64):
//   tone_gen()                 src/tone_generate.c:128-229
//   dds_modf()                 src/dds_float.c:2167-2174   (2048 entry sine table, phase >> 21)
//   lfastrintf()               src/spandsp/fast_convert.h:184-197: a plain (long) cast on x86-64,
//                              i.e. truncation toward zero
//   dtmf_tx()                  src/dtmf.c:551-590
//   bell_mf_tx()               src/bell_r2_mf.c:306-329
//   r2_mf_tx()                 src/bell_r2_mf.c:399-414
//   queue_read_byte()          src/queue.c:197-220
//
// Layout: state is structure-of-arrays int32 words [kTxWords][n_channels] in HBM, so a wave's
// loads and stores of one word are one coalesced access.  The kernel (see tx_bank_kernel) turns each
// channel's frame into a few run descriptors with one lane, then lets the whole wave fill the
// channel's row of the caller's channel-major PCM buffer with contiguous stores.  The sine table
// sits in LDS.
//
// The reference leaves amp[len..max) untouched when a sender runs out of digits; this bank
// zero-fills that tail and reports len per channel.

#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spg
{

enum
{
    TX_RATE0 = 0,       // 4 words: phase rate of each tone (0 ends the list; tone 0 < 0: AM pair)
    TX_GAIN0 = 4,       // 4 words: float gain
    TX_PHASE0 = 8,      // 4 words: DDS phase accumulators
    TX_DUR0 = 12,       // 4 words: cadence section lengths in samples
    TX_REPEAT = 16,
    TX_SECTION = 17,    // -1: idle
    TX_POS = 18,
    TX_LOW = 19,        // dtmf_tx: float level of the row / column tone, on and off time (samples)
    TX_HIGH = 20,
    TX_ON = 21,
    TX_OFF = 22,
    TX_QRD = 23,        // digit queue: read index and fill of a 128 byte ring
    TX_QCOUNT = 24,
    TX_R2DIGIT = 25,    // r2_mf_tx: 0 = send silence
    TX_Q0 = 26,         // 32 words: the ring, four digits per word, little endian
    kTxWords = 58
};

constexpr int kTxQueue = 128;

enum { TXK_TONE_GEN = 1, TXK_DTMF = 2, TXK_BELL_MF = 3, TXK_R2_FWD = 4, TXK_R2_BACK = 5 };

// Per-digit descriptors of the queued senders (built on the host from the reference's frequency
// tables with dds_phase_ratef()/dds_scaling_dbm0f()).
struct TxDigitTable
{
    int32_t rate[16][2];
    float gain[16][2];
    int32_t on[16];
    int32_t off[16];
    char keys[17];
    int n;
};

struct TxLaunch
{
    int32_t *st;
    const float *sine;      // [2048] in HBM
    int16_t *pcm;           // [n_ch][stride]
    int32_t *lens;          // [n_ch] or null
    long long stride;
    int n_ch;
    int samples;
    int kind;
    TxDigitTable dig;
};

struct TxGen
{
    int32_t rate[4];
    float gain[4];
    uint32_t phase[4];
    int32_t dur[4];
    int32_t repeat;
    int32_t section;
    int32_t pos;
};

// The length of cadence section `section` (values, not references: a conditional on array elements
// would otherwise become a run-time indexed access and push the generator out of registers).
__device__ __forceinline__ int tx_dur_of(const TxGen &g, int section)
{
    const int d0 = g.dur[0];
    const int d1 = g.dur[1];
    const int d2 = g.dur[2];
    const int d3 = g.dur[3];
    int d = d0;
    d = (section == 1)  ?  d1  :  d;
    d = (section == 2)  ?  d2  :  d;
    d = (section == 3)  ?  d3  :  d;
    return d;
}

// End-of-section bookkeeping, tone_generate.c:211-227.
__device__ __forceinline__ void tx_section_end(TxGen &g)
{
    g.pos = 0;
    g.section++;
    const int nd = tx_dur_of(g, g.section);
    if (g.section > 3  ||  nd == 0)
        g.section = g.repeat  ?  0  :  -1;
}

__device__ __forceinline__ int tx_cur_dur(const TxGen &g)
{
    return tx_dur_of(g, g.section);
}

__device__ __forceinline__ int tx_key_index(const TxDigitTable &T, int digit)
{
    int idx = -1;
    for (int k = 0;  k < 16;  k++)
        idx = (k < T.n  &&  digit == (int) T.keys[k]  &&  idx < 0)  ?  k  :  idx;
    return idx;
}

// A frame of one channel is a handful of RUNS: stretches of samples inside one cadence section of one
// digit.  Inside a run sample k is a pure function of k (DDS phases are phase0 + k*rate mod 2^32), so the
// kernel splits the work in two:
//   phase 1 (one lane per channel): walk the cadence / digit-queue state machine run by run -- no
//            per-sample work -- and leave up to kTxRuns run descriptors per channel in LDS;
//   phase 2 (all 64 lanes over the wave's channels x samples): a lane makes two adjacent samples from the
//            descriptor(s) of the run(s) they fall in, and the wave stores 256 contiguous bytes per pass.
// A channel with more runs than fit (cadences of a few ms) simply takes another round.
constexpr int kTxRuns = 3;
constexpr int kTxRunWords = 16;
// run descriptor words: 0 start, 1 mode (0 silence, 1 sum of tones, 2 AM pair), 2-3 rate[0..1], 4-5 gain[0..1],
// 6-7 phase[0..1], 8-9 rate[2..3], 10-11 gain[2..3], 12-13 phase[2..3]; mode bit 2: tones 2..3 are in use

constexpr int kTxWaves = 4;         // waves per workgroup; they share one copy of the sine table
constexpr int kTxChannelsPerWave = 16;

template <int CPW>
__global__ __launch_bounds__(64*kTxWaves) void tx_bank_kernel(const TxLaunch L)
{
    __shared__ float sine[2048];
    __shared__ int32_t dig_rate[16][2];
    __shared__ float dig_gain[16][2];
    __shared__ int32_t dig_on[16];
    __shared__ int32_t dig_off[16];
    __shared__ __attribute__((aligned(16))) int32_t all_runs[kTxWaves][CPW][kTxRuns][kTxRunWords];
    __shared__ __attribute__((aligned(16))) int32_t all_hdr[kTxWaves][CPW][4];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int32_t (*runs)[kTxRuns][kTxRunWords] = all_runs[wave];
    int32_t (*r_hdr)[4] = all_hdr[wave];
    const int ch0 = (blockIdx.x*kTxWaves + wave)*CPW;
    const int ch = ch0 + lane;
    const bool owner = (lane < CPW)  &&  (ch < L.n_ch);

    for (int i = threadIdx.x;  i < 2048;  i += 64*kTxWaves)
        sine[i] = L.sine[i];
    if (threadIdx.x < 16)
    {
        dig_rate[lane][0] = L.dig.rate[lane][0];
        dig_rate[lane][1] = L.dig.rate[lane][1];
        dig_gain[lane][0] = L.dig.gain[lane][0];
        dig_gain[lane][1] = L.dig.gain[lane][1];
        dig_on[lane] = L.dig.on[lane];
        dig_off[lane] = L.dig.off[lane];
    }
    __syncthreads();

    int32_t *st = L.st + (owner  ?  ch  :  ch0);
    const size_t n = (size_t) L.n_ch;
    const bool queued = (L.kind == TXK_DTMF  ||  L.kind == TXK_BELL_MF);
    const bool is_r2 = (L.kind == TXK_R2_FWD  ||  L.kind == TXK_R2_BACK);
    TxGen g;
    float low = 0.0f;
    float high = 0.0f;
    int on_time = 0;
    int off_time = 0;
    int qrd = 0;
    int qcount = 0;
    bool r2_silent = false;
#pragma unroll
    for (int i = 0;  i < 4;  i++)
    {
        g.rate[i] = 0;
        g.gain[i] = 0.0f;
        g.phase[i] = 0u;
        g.dur[i] = 0;
    }
    g.repeat = 0;
    g.section = -1;
    g.pos = 0;
    if (owner)
    {
#pragma unroll
        for (int i = 0;  i < 4;  i++)
        {
            g.rate[i] = st[(TX_RATE0 + i)*n];
            g.gain[i] = __int_as_float(st[(TX_GAIN0 + i)*n]);
            g.phase[i] = (uint32_t) st[(TX_PHASE0 + i)*n];
            g.dur[i] = st[(TX_DUR0 + i)*n];
        }
        g.repeat = st[TX_REPEAT*n];
        g.section = st[TX_SECTION*n];
        g.pos = st[TX_POS*n];
        if (queued)
        {
            qrd = st[TX_QRD*n];
            qcount = st[TX_QCOUNT*n];
            if (L.kind == TXK_DTMF)
            {
                low = __int_as_float(st[TX_LOW*n]);
                high = __int_as_float(st[TX_HIGH*n]);
                on_time = st[TX_ON*n];
                off_time = st[TX_OFF*n];
            }
        }
        // r2_mf_tx() with no digit keyed writes silence and leaves the generator alone (bell_r2_mf.c:403-407)
        r2_silent = is_r2  &&  st[TX_R2DIGIT*n] == 0;
    }

    const int samples = L.samples;
    int done = owner  ?  0  :  samples;
    int len = 0;
    bool stopped = false;

    for (;;)
    {
        // ---- phase 1: the next runs of my channel ----
        const int lo = done;
        int nr = 0;
        if (owner)
        {
            int32_t (*R)[kTxRunWords] = runs[lane];
            bool can_merge = false;
            int prev_mode = -1;
            int guard = 0;
            while (done < samples  &&  nr < kTxRuns)
            {
                if (r2_silent  ||  stopped  ||  guard > 8)
                {
                    // silence to the end of the frame; only r2_mf_tx() counts it as produced
#pragma unroll
                    for (int w = 0;  w < kTxRunWords;  w++)
                        R[nr][w] = 0;
                    R[nr][0] = done;
                    nr++;
                    len += r2_silent  ?  (samples - done)  :  0;
                    done = samples;
                    break;
                }
                if (g.section < 0)
                {
                    // tone_gen() is finished: the next queued digit, if any (dtmf.c:562-582, bell_r2_mf.c:316-327)
                    if (!queued  ||  qcount == 0)
                    {
                        stopped = true;
                        continue;
                    }
                    const int word = st[(size_t) (TX_Q0 + (qrd >> 2))*n];
                    const int digit = (word >> ((qrd & 3)*8)) & 0xFF;
                    qrd = (qrd + 1) & (kTxQueue - 1);
                    qcount--;
                    const int k = (digit == 0)  ?  -1  :  tx_key_index(L.dig, digit);
                    if (k < 0)
                        continue;
                    // tone_gen_init() on the digit's descriptor (tone_generate.c:232-262)
                    g.rate[0] = dig_rate[k][0];
                    g.rate[1] = dig_rate[k][1];
                    g.rate[2] = g.rate[3] = 0;
                    g.gain[0] = dig_gain[k][0];
                    g.gain[1] = dig_gain[k][1];
                    g.gain[2] = g.gain[3] = 0.0f;
                    g.phase[0] = g.phase[1] = g.phase[2] = g.phase[3] = 0u;
                    g.dur[0] = dig_on[k];
                    g.dur[1] = dig_off[k];
                    g.dur[2] = g.dur[3] = 0;
                    g.repeat = 0;
                    if (L.kind == TXK_DTMF)
                    {
                        // dtmf.c:577-580
                        g.gain[0] = low;
                        g.gain[1] = high;
                        g.dur[0] = on_time;
                        g.dur[1] = off_time;
                    }
                    g.section = 0;
                    g.pos = 0;
                    can_merge = false;
                    guard = 0;
                    continue;
                }
                // tone_generate.c:141-147: to the end of the section or of the buffer
                const int cur = tx_cur_dur(g);
                int run = cur - g.pos;
                run = (run > samples - done)  ?  (samples - done)  :  run;
                if (run > 0)
                {
                    const int mode = (g.section & 1)  ?  0  :  ((g.rate[0] < 0)  ?  2  :  1);
                    // how many tones sound: the list ends at the first zero rate (tone_generate.c:195-198)
                    int nt = 0;
                    nt = (g.rate[0] != 0)  ?  1  :  0;
                    nt = (nt == 1  &&  g.rate[1] != 0)  ?  2  :  nt;
                    nt = (nt == 2  &&  g.rate[2] != 0)  ?  3  :  nt;
                    nt = (nt == 3  &&  g.rate[3] != 0)  ?  4  :  nt;
                    nt = (mode == 2)  ?  2  :  nt;
                    if (!(can_merge  &&  prev_mode == mode))
                    {
                        // a silent tone slot adds sine*0 = +-0, which leaves the sum as it is
                        R[nr][0] = done;
                        R[nr][1] = mode | ((nt > 2)  ?  4  :  0);
                        nt = (mode == 0)  ?  0  :  nt;     // a silent section: every gain 0, so the sum is 0
                        R[nr][2] = (mode == 2)  ?  -g.rate[0]  :  g.rate[0];
                        R[nr][3] = g.rate[1];
                        R[nr][4] = __float_as_int((nt > 0)  ?  g.gain[0]  :  0.0f);
                        R[nr][5] = __float_as_int((nt > 1)  ?  g.gain[1]  :  0.0f);
                        R[nr][6] = (int32_t) g.phase[0];
                        R[nr][7] = (int32_t) g.phase[1];
                        R[nr][8] = g.rate[2];
                        R[nr][9] = g.rate[3];
                        R[nr][10] = __float_as_int((nt > 2)  ?  g.gain[2]  :  0.0f);
                        R[nr][11] = __float_as_int((nt > 3)  ?  g.gain[3]  :  0.0f);
                        R[nr][12] = (int32_t) g.phase[2];
                        R[nr][13] = (int32_t) g.phase[3];
                        nr++;
                    }
                    prev_mode = mode;
                    can_merge = true;
                    if (mode == 2)
                    {
                        g.phase[0] += (uint32_t) run*(uint32_t) (-g.rate[0]);
                        g.phase[1] += (uint32_t) run*(uint32_t) g.rate[1];
                    }
                    else if (mode == 1)
                    {
#pragma unroll
                        for (int i = 0;  i < 4;  i++)
                            g.phase[i] += (i < nt)  ?  (uint32_t) run*(uint32_t) g.rate[i]  :  0u;
                    }
                    g.pos += run;
                    done += run;
                    len += run;
                    guard = 0;
                }
                else
                {
                    // only a cadence of nothing but empty sections gets here twice in a row (the reference
                    // would spin for ever in tone_gen(); the host API refuses such descriptors)
                    guard++;
                }
                if (g.pos >= cur)
                    tx_section_end(g);
            }
        }
        if (lane < CPW)
        {
            // this round's stretch [lo, hi) and where its second and third run begin
            r_hdr[lane][0] = lo;
            r_hdr[lane][1] = done;
            r_hdr[lane][2] = (owner  &&  nr > 1)  ?  runs[lane][1][0]  :  0x7FFFFFFF;
            r_hdr[lane][3] = (owner  &&  nr > 2)  ?  runs[lane][2][0]  :  0x7FFFFFFF;
        }
        __syncthreads();

        // ---- phase 2: the samples of those runs.  The wave's channels x sample pairs form one flat index
        // space, so every pass keeps all 64 lanes busy and stores 256 contiguous bytes per row touched.
        {
            const int nchan = (L.n_ch - ch0 < CPW)  ?  (L.n_ch - ch0)  :  CPW;
            const int ppr = (samples + 1) >> 1;
            const uint32_t magic = 0xFFFFFFFFu/(uint32_t) ppr + 1u;     // floor(idx/ppr) = umulhi(idx, magic) while idx*ppr < 2^32
            const bool exact = (ppr > 1  &&  ppr < 16384);
            const bool pair_store = ((L.stride & 1) == 0)  &&  ((reinterpret_cast<uintptr_t>(L.pcm) & 3) == 0);
            const int total = nchan*ppr;
#pragma unroll 2
            for (int idx = lane;  idx < total;  idx += 64)
            {
                const int c = exact  ?  (int) __umulhi((uint32_t) idx, magic)  :  (idx/ppr);
                const int i0 = (idx - c*ppr)*2;
                const int4 hdr = *reinterpret_cast<const int4 *>(&r_hdr[c][0]);     // lo, hi, start of run 1, of run 2
                const bool in0 = (i0 >= hdr.x)  &&  (i0 < hdr.y);
                const bool in1 = (i0 + 1 >= hdr.x)  &&  (i0 + 1 < hdr.y);
                if (!(in0  ||  in1))
                    continue;
                const int r0 = ((i0 >= hdr.z)  ?  1  :  0) + ((i0 >= hdr.w)  ?  1  :  0);
                const int r1 = ((i0 + 1 >= hdr.z)  ?  1  :  0) + ((i0 + 1 >= hdr.w)  ?  1  :  0);
                int v[2];
                int4 a = *reinterpret_cast<const int4 *>(&runs[c][r0][0]);     // start, mode, rate0, rate1
                int4 b = *reinterpret_cast<const int4 *>(&runs[c][r0][4]);     // gain0, gain1, phase0, phase1
#pragma unroll
                for (int h = 0;  h < 2;  h++)
                {
                    const int i = i0 + h;
                    const int r = h  ?  r1  :  r0;
                    if (h == 1  &&  r1 != r0)
                    {
                        a = *reinterpret_cast<const int4 *>(&runs[c][r1][0]);
                        b = *reinterpret_cast<const int4 *>(&runs[c][r1][4]);
                    }
                    const uint32_t k = (uint32_t) (i - a.x);
                    const float t0 = __fmul_rn(sine[((uint32_t) b.z + k*(uint32_t) a.z) >> 21], __int_as_float(b.x));
                    const float t1 = __fmul_rn(sine[((uint32_t) b.w + k*(uint32_t) a.w) >> 21], __int_as_float(b.y));
                    // sum of tones, tone_generate.c:190-207 (the leading 0.0f + t0 cannot change the integer result)
                    float x = __fadd_rn(t0, t1);
                    if (a.y & 6)
                    {
                        if (a.y & 4)
                        {
                            const int32_t *D = runs[c][r];
                            const int4 d = *reinterpret_cast<const int4 *>(&D[8]);  // rate2, rate3, gain2, gain3
                            x = __fadd_rn(x, __fmul_rn(sine[((uint32_t) D[12] + k*(uint32_t) d.x) >> 21], __int_as_float(d.z)));
                            x = __fadd_rn(x, __fmul_rn(sine[((uint32_t) D[13] + k*(uint32_t) d.y) >> 21], __int_as_float(d.w)));
                        }
                        else
                        {
                            // amplitude modulated pair, tone_generate.c:166-183
                            x = __fmul_rn(t0, __fadd_rn(1.0f, t1));
                        }
                    }
                    // lfastrintf() on x86-64: truncation
                    v[h] = (int) x;
                }
                int16_t *at = L.pcm + (size_t) (ch0 + c)*L.stride + i0;
                if (in0  &&  in1  &&  pair_store)
                {
                    *reinterpret_cast<uint32_t *>(at) = ((uint32_t) v[0] & 0xFFFFu) | ((uint32_t) v[1] << 16);
                }
                else
                {
                    if (in0)
                        at[0] = (int16_t) v[0];
                    if (in1)
                        at[1] = (int16_t) v[1];
                }
            }
        }
        if (!__syncthreads_or(done < samples))
            break;
    }

    if (owner)
    {
        if (!r2_silent)
        {
#pragma unroll
            for (int i = 0;  i < 4;  i++)
            {
                st[(TX_RATE0 + i)*n] = g.rate[i];
                st[(TX_GAIN0 + i)*n] = __float_as_int(g.gain[i]);
                st[(TX_PHASE0 + i)*n] = (int32_t) g.phase[i];
                st[(TX_DUR0 + i)*n] = g.dur[i];
            }
            st[TX_REPEAT*n] = g.repeat;
            st[TX_SECTION*n] = g.section;
            st[TX_POS*n] = g.pos;
            if (queued)
            {
                st[TX_QRD*n] = qrd;
                st[TX_QCOUNT*n] = qcount;
            }
        }
        if (L.lens)
            L.lens[ch] = len;
    }
}

// tone_gen_init() of one descriptor on channels [lo, hi): words 0..18 (and the R2 digit).
struct TxDescriptor
{
    int32_t w[13];      // rate[4], gain[4], duration[4], repeat
    int32_t r2digit;    // < 0: leave TX_R2DIGIT alone
    int32_t load;       // 0: only set TX_R2DIGIT
};

__global__ void tx_load_descriptor_kernel(int32_t *st, int n_ch, int lo, int hi, const TxDescriptor d)
{
    const int ch = lo + blockIdx.x*blockDim.x + threadIdx.x;
    if (ch >= hi)
        return;
    const size_t n = (size_t) n_ch;
    int32_t *s = st + ch;
    if (d.load)
    {
#pragma unroll
        for (int i = 0;  i < 4;  i++)
        {
            s[(TX_RATE0 + i)*n] = d.w[i];
            s[(TX_GAIN0 + i)*n] = d.w[4 + i];
            s[(TX_PHASE0 + i)*n] = 0;
            s[(TX_DUR0 + i)*n] = d.w[8 + i];
        }
        s[TX_REPEAT*n] = d.w[12];
        s[TX_SECTION*n] = 0;
        s[TX_POS*n] = 0;
    }
    if (d.r2digit >= 0)
        s[TX_R2DIGIT*n] = d.r2digit;
}

__global__ void tx_set_words_kernel(int32_t *st, int n_ch, int lo, int hi, int idx0, int32_t v0, int idx1, int32_t v1)
{
    const int ch = lo + blockIdx.x*blockDim.x + threadIdx.x;
    if (ch >= hi)
        return;
    st[(size_t) idx0*n_ch + ch] = v0;
    if (idx1 >= 0)
        st[(size_t) idx1*n_ch + ch] = v1;
}

// xxx_tx_put() on channels [lo, hi): digits of channel c are at digits[(c - lo)*dstride ...] with length
// lens[c - lo] (or `len` for every channel when lens is null).  All or nothing per channel, like
// queue_write() with QUEUE_WRITE_ATOMIC (queue.c); result[c - lo] = characters that did not fit.
__global__ void tx_put_kernel(int32_t *st, int n_ch, int lo, int hi, const uint8_t *digits, int dstride, const int32_t *lens,
                              int len, int32_t *result)
{
    const int ch = lo + blockIdx.x*blockDim.x + threadIdx.x;
    if (ch >= hi)
        return;
    const size_t n = (size_t) n_ch;
    int32_t *s = st + ch;
    const int mine = lens  ?  lens[ch - lo]  :  len;
    const uint8_t *src = digits + (size_t) (ch - lo)*dstride;
    const int rd = s[TX_QRD*n];
    const int count = s[TX_QCOUNT*n];
    const int space = kTxQueue - count;
    if (mine > space)
    {
        result[ch - lo] = mine - space;
        return;
    }
    for (int i = 0;  i < mine;  i++)
    {
        const int at = (rd + count + i) & (kTxQueue - 1);
        int32_t w = s[(size_t) (TX_Q0 + (at >> 2))*n];
        w = (w & ~(0xFF << ((at & 3)*8))) | ((int32_t) src[i] << ((at & 3)*8));
        s[(size_t) (TX_Q0 + (at >> 2))*n] = w;
    }
    s[TX_QCOUNT*n] = count + mine;
    result[ch - lo] = 0;
}

}   // namespace spg
