// feed_api.hip -- the pipelined host-buffer path of a tone bank (include/spangpu.h: spangpu_feed_*).
//
// spangpu_bank_rx(.., SPANGPU_MEM_HOST, ..) is a synchronous convenience: copy the frame in, launch, and the caller then
// asks for records.  A caller that holds its frames in host memory tick after tick -- every caller that is not itself
// on the GPU -- wants the three legs to overlap: while tick t's kernel runs and its digits travel back, tick t + 1's
// frame is already crossing PCIe.  A feed owns, per slot of a small ring: a pinned host buffer the caller's receive
// path writes the frame into (no staging copy by the library), a device frame buffer, a device digit list and its
// pinned host copy.  commit() queues  H2D (copy stream) -> event -> kernel + digit list + D2H (bank stream) -> event;
// collect() waits for the oldest tick's last event and hands out its digits: channel | digit << 20 | block << 28, the
// entries of spangpu_bank_digit_events().  What crosses PCIe per tick is the frame down and four bytes per digit up,
// not the 32-bit record of every block of every channel.
#include <hip/hip_runtime.h>
#include <time.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"

extern "C" int spangpu_set_error(int code, const char *msg);

#define FEED_TRY(x) do { if ((x) != hipSuccess) return spangpu_set_error(SPANGPU_ERR_HIP, #x " failed"); } while (0)

enum { kFeedMaxDepth = 8 };

struct spangpu_feed_s
{
    spangpu_bank_t *bank;
    int device;
    int n_ch;
    int max_samples;
    int law;                    // 0 = 16 bit linear, SPANGPU_G711_ALAW / _ULAW = one byte per sample
    int depth;
    int cap;                    // digit list entries per tick (no tick can make more: blocks per frame x channels)
    int quick;                  // entries that travel back with every tick; a livelier tick has the rest fetched when it is collected
    long long stride;           // samples per row of the staging buffers (rows 16-byte aligned)
    size_t frame_bytes;
    void *h_stage[kFeedMaxDepth];
    void *d_frame[kFeedMaxDepth];
    uint32_t *d_list[kFeedMaxDepth];
    uint32_t *h_list[kFeedMaxDepth];
    hipEvent_t ev_h2d[kFeedMaxDepth];
    hipEvent_t ev_done[kFeedMaxDepth];
    bool busy[kFeedMaxDepth];
    hipStream_t copy_stream;
    long long n_commit;
    long long n_collect;
};

extern "C" {

int spangpu_feed_destroy(spangpu_feed_t *f)
{
    if (f == nullptr)
        return SPANGPU_OK;
    (void) hipSetDevice(f->device);
    if (f->copy_stream)
    {
        (void) hipStreamSynchronize(f->copy_stream);
        (void) spangpu_bank_sync(f->bank);
        (void) hipStreamDestroy(f->copy_stream);
    }
    for (int k = 0;  k < f->depth;  k++)
    {
        if (f->h_stage[k]) (void) hipHostFree(f->h_stage[k]);
        if (f->d_frame[k]) (void) hipFree(f->d_frame[k]);
        if (f->d_list[k]) (void) hipFree(f->d_list[k]);
        if (f->h_list[k]) (void) hipHostFree(f->h_list[k]);
        if (f->ev_h2d[k]) (void) hipEventDestroy(f->ev_h2d[k]);
        if (f->ev_done[k]) (void) hipEventDestroy(f->ev_done[k]);
    }
    free(f);
    return SPANGPU_OK;
}

int spangpu_feed_create(spangpu_feed_t **out, spangpu_bank_t *bank, int device, int max_samples, int law, int depth)
{
    if (out == nullptr  ||  bank == nullptr  ||  max_samples <= 0  ||  depth < 1  ||  depth > kFeedMaxDepth)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments (1 .. 8 slots)");
    if (law != 0  &&  law != SPANGPU_G711_ALAW  &&  law != SPANGPU_G711_ULAW)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "law must be 0 (16 bit linear), SPANGPU_G711_ALAW or SPANGPU_G711_ULAW");
    *out = nullptr;
    const int n_ch = spangpu_bank_channels(bank);
    if (n_ch <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad bank");
    // the frame buffers and the copy stream live where the kernels read them: on the bank's device, no other
    if (device != spangpu_bank_device(bank))
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "the feed's device must be the bank's");
    // spangpu_bank_digit_events() packs the block index in four bits: a tick of more than 16 blocks per channel (the shortest
    // block is 64 samples) could queue its copy and kernel and then fail to report -- refused here instead of half way
    if (max_samples > 16*64)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "a feed's ticks hold at most 1024 samples per channel (16 blocks)");
    spangpu_feed_t *f = (spangpu_feed_t *) calloc(1, sizeof(*f));
    if (f == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of memory");
    f->bank = bank;
    f->device = device;
    f->n_ch = n_ch;
    f->max_samples = max_samples;
    f->law = law;
    f->depth = depth;
    const int bps = law  ?  1  :  2;
    f->stride = ((long long) max_samples*bps + 15)/16*16/bps;
    f->frame_bytes = (size_t) f->stride*bps*n_ch;
    // a block is at least 64 samples long on every detector kind: blocks per frame (+1 for one in progress)
    f->cap = n_ch*(max_samples/64 + 2);
    f->quick = (n_ch/8 > 4096)  ?  n_ch/8  :  4096;
    if (f->quick > f->cap)
        f->quick = f->cap;
    bool ok = (hipSetDevice(device) == hipSuccess)  &&  (hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking) == hipSuccess);
    for (int k = 0;  ok  &&  k < depth;  k++)
    {
        ok = hipHostMalloc(&f->h_stage[k], f->frame_bytes) == hipSuccess
             &&  hipMalloc(&f->d_frame[k], f->frame_bytes + 64) == hipSuccess
             &&  hipMalloc((void **) &f->d_list[k], (size_t) (1 + f->cap)*sizeof(uint32_t)) == hipSuccess
             &&  hipHostMalloc((void **) &f->h_list[k], (size_t) (1 + f->cap)*sizeof(uint32_t)) == hipSuccess
             &&  hipEventCreateWithFlags(&f->ev_h2d[k], hipEventDisableTiming) == hipSuccess
             &&  hipEventCreateWithFlags(&f->ev_done[k], hipEventDisableTiming) == hipSuccess;
        if (ok)
            memset(f->h_stage[k], 0, f->frame_bytes);
    }
    if (!ok)
    {
        spangpu_feed_destroy(f);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of (pinned) memory for the feed");
    }
    *out = f;
    return SPANGPU_OK;
}

// The layout of a staging buffer: channel c's samples start at c*stride samples (16 bit linear) or bytes (G.711).
long long spangpu_feed_stride(const spangpu_feed_t *f)
{
    return f  ?  f->stride  :  (long long) SPANGPU_ERR_BAD_ARG;
}

// The pinned host buffer of the next tick, for the caller to write the frame into; NULL while every slot holds a tick
// that has not been collected.
void *spangpu_feed_acquire(spangpu_feed_t *f)
{
    if (f == nullptr)
        return nullptr;
    const int slot = (int) (f->n_commit % f->depth);
    if (f->busy[slot])
    {
        spangpu_set_error(SPANGPU_ERR_STATE, "every slot of the feed holds a tick: collect one first");
        return nullptr;
    }
    return f->h_stage[slot];
}

// Queue the tick whose frame the caller has written into the acquired buffer.  Returns at once.
int spangpu_feed_commit(spangpu_feed_t *f, int samples)
{
    if (f == nullptr  ||  samples <= 0  ||  samples > f->max_samples)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int slot = (int) (f->n_commit % f->depth);
    if (f->busy[slot])
        return spangpu_set_error(SPANGPU_ERR_STATE, "every slot of the feed holds a tick: collect one first");
    FEED_TRY(hipSetDevice(f->device));
    hipStream_t bs = (hipStream_t) spangpu_bank_get_stream(f->bank);
    // (the slot's device buffers are free: its last tick was collected, i.e. its kernel and copies are done)
    FEED_TRY(hipMemcpyAsync(f->d_frame[slot], f->h_stage[slot], f->frame_bytes, hipMemcpyHostToDevice, f->copy_stream));
    FEED_TRY(hipEventRecord(f->ev_h2d[slot], f->copy_stream));
    FEED_TRY(hipStreamWaitEvent(bs, f->ev_h2d[slot], 0));
    int rc;
    if (f->law)
        rc = spangpu_bank_rx_g711(f->bank, (const uint8_t *) f->d_frame[slot], SPANGPU_MEM_DEVICE, f->law, samples, f->stride);
    else
        rc = spangpu_bank_rx(f->bank, (const int16_t *) f->d_frame[slot], SPANGPU_MEM_DEVICE, SPANGPU_LAYOUT_CHANNEL_MAJOR, samples, f->stride);
    if (rc < 0)
        return rc;
    if ((rc = spangpu_bank_digit_events(f->bank, f->d_list[slot], f->cap)) < 0)
        return rc;
    FEED_TRY(hipMemcpyAsync(f->h_list[slot], f->d_list[slot], (size_t) (1 + f->quick)*sizeof(uint32_t), hipMemcpyDeviceToHost, bs));
    FEED_TRY(hipEventRecord(f->ev_done[slot], bs));
    f->busy[slot] = true;
    f->n_commit++;
    return SPANGPU_OK;
}

// The digits of the oldest tick not yet collected (waits for it): *entries points at `return value` words
// channel | digit << 20 | block << 28, good until the slot is committed again.  Returns 0 with *entries = NULL when no
// tick is outstanding.
int spangpu_feed_collect(spangpu_feed_t *f, const uint32_t **entries)
{
    if (f == nullptr  ||  entries == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *entries = nullptr;
    if (f->n_collect >= f->n_commit)
        return 0;
    const int slot = (int) (f->n_collect % f->depth);
    FEED_TRY(hipSetDevice(f->device));
    FEED_TRY(hipEventSynchronize(f->ev_done[slot]));
    f->busy[slot] = false;
    f->n_collect++;
    const uint32_t n = f->h_list[slot][0];
    if (n > (uint32_t) f->cap)
        return spangpu_set_error(SPANGPU_ERR_STATE, "digit list overflow");
    if (n > (uint32_t) f->quick)
    {
        // more digits than travel with every tick (an eighth of the channels delivering one in the same 20 ms): the rest now
        FEED_TRY(hipMemcpy(f->h_list[slot] + 1 + f->quick, f->d_list[slot] + 1 + f->quick, (size_t) (n - (uint32_t) f->quick)*sizeof(uint32_t),
                           hipMemcpyDeviceToHost));
    }
    *entries = f->h_list[slot] + 1;
    return (int) n;
}

int spangpu_feed_outstanding(const spangpu_feed_t *f)
{
    return f  ?  (int) (f->n_commit - f->n_collect)  :  SPANGPU_ERR_BAD_ARG;
}

// A caller's tick loop in C, for measurements (bench.py's `e2e`): `ticks` times acquire (the slots keep the frames they
// hold: the caller's receive path is not part of this path) / commit / collect with `lag` ticks between a commit and the
// collect of its digits (1 .. depth - 1), then the rest collected.  *elapsed_ms = wall time of the whole loop.
int spangpu_feed_run(spangpu_feed_t *f, int samples, int ticks, int lag, double *elapsed_ms, long long *digits)
{
    if (f == nullptr  ||  ticks <= 0  ||  lag < 1  ||  lag >= f->depth  ||  spangpu_feed_outstanding(f) != 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments (lag in 1 .. depth - 1, nothing outstanding)");
    struct timespec t0, t1;
    long long seen = 0;
    const uint32_t *entries;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0;  i < ticks;  i++)
    {
        if (spangpu_feed_acquire(f) == nullptr)
            return SPANGPU_ERR_STATE;
        int rc = spangpu_feed_commit(f, samples);
        if (rc < 0)
            return rc;
        if (spangpu_feed_outstanding(f) > lag)
        {
            if ((rc = spangpu_feed_collect(f, &entries)) < 0)
                return rc;
            seen += rc;
        }
    }
    while (spangpu_feed_outstanding(f) > 0)
    {
        const int rc = spangpu_feed_collect(f, &entries);
        if (rc < 0)
            return rc;
        seen += rc;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (elapsed_ms)
        *elapsed_ms = (t1.tv_sec - t0.tv_sec)*1e3 + (t1.tv_nsec - t0.tv_nsec)*1e-6;
    if (digits)
        *digits = seen;
    return SPANGPU_OK;
}

}   // extern "C"

// =====================================================================================================================
// The pipelined host path of the echo canceller and modem receiver banks (round 4; include/spangpu.h: spangpu_echo_feed_*,
// spangpu_modem_feed_*).
//
// echo_can_update() hands the cleaned sample back to its caller (src/echo.c:421-661) and v29_rx() delivers bits through
// put_bit (src/v29rx.c:867-965): for these paths the way back over PCIe is as much the product as the way down.  A slot of
// these feeds therefore has a pinned buffer each way, and a tick's three legs run on three streams --
//     copy-in stream:   H2D of the tick's input                       -> event
//     the bank's stream: (waits) the kernel(s)                         -> event
//     copy-out stream:  (waits) D2H of the tick's output              -> event (the tick is done)
// so that tick t + 1's input crosses PCIe downwards while tick t's kernel runs and tick t - 1's output crosses it upwards:
// the link is full duplex, and a tick costs max(H2D, D2H, kernel), not their sum.
//
// Echo: input = tx rows then rx rows ([2][n_ch][stride] int16, or G.711 bytes decoded on the device), output = the clean
// rows ([n_ch][stride] int16, or G.711 bytes encoded on the device: linear_to_alaw / linear_to_ulaw, spandsp/g711.h).
// Modem: input = PCM rows, output = the put_bit stream packed by spangpu_modem_pack_events() (a header word and the data
// bits per channel, one sparse list of status reports per bank): 28 bytes per channel and tick for V.29 9600 instead of
// one byte per put_bit call.
// =====================================================================================================================
extern "C" {
void *spangpu_echo_get_stream(spangpu_echo_t *e);
int spangpu_echo_device(const spangpu_echo_t *e);
void *spangpu_modem_get_stream(spangpu_modem_t *m);
int spangpu_modem_device(const spangpu_modem_t *m);
}

namespace {

// spandsp/g711.h:165-175 (ulaw_to_linear) and :239-252 (alaw_to_linear)
__device__ __forceinline__ int g711_decode(int code, int law)
{
    if (law == SPANGPU_G711_ULAW)
    {
        const int u = ~code & 0xFF;
        const int t = (((u & 0x0F) << 3) + 0x84) << ((u & 0x70) >> 4);
        return (u & 0x80)  ?  (0x84 - t)  :  (t - 0x84);
    }
    const int a = code ^ 0x55;
    int i = (a & 0x0F) << 4;
    const int sg = (a & 0x70) >> 4;
    i = sg  ?  ((i + 0x108) << (sg - 1))  :  (i + 8);
    return (a & 0x80)  ?  i  :  -i;
}

// spandsp/g711.h:128-163 (linear_to_ulaw) and :198-237 (linear_to_alaw); top_bit() = 31 - clz
__device__ __forceinline__ int g711_encode(int linear, int law)
{
    if (law == SPANGPU_G711_ULAW)
    {
        int mask;
        if (linear >= 0)
        {
            linear = 0x84 + linear;             // G711_ULAW_BIAS
            mask = 0xFF;
        }
        else
        {
            linear = 0x84 - linear;
            mask = 0x7F;
        }
        const int seg = (31 - __clz(linear | 0xFF)) - 7;
        if (seg >= 8)
            return 0x7F ^ mask;
        return ((seg << 4) | ((linear >> (seg + 3)) & 0xF)) ^ mask;
    }
    int mask;
    if (linear >= 0)
    {
        mask = 0x55 | 0x80;                     // G711_ALAW_AMI_MASK | 0x80
    }
    else
    {
        mask = 0x55;
        linear = -linear - 1;
    }
    const int seg = (31 - __clz(linear | 0xFF)) - 7;
    if (seg >= 8)
        return 0x7F ^ mask;
    return ((seg << 4) | ((linear >> (seg  ?  (seg + 3)  :  4)) & 0x0F)) ^ mask;
}

// rows of `n` samples: 16 codes per thread-step (one 16-byte load, two 16-byte stores)
__global__ void g711_to_linear_kernel(const uint8_t *src, int16_t *dst, long long total16, int law)
{
    const long long i = (long long) blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= total16)
        return;
    const uint4 c = ((const uint4 *) src)[i];
    const uint32_t w[4] = {c.x, c.y, c.z, c.w};
    uint32_t o[8];
#pragma unroll
    for (int k = 0;  k < 4;  k++)
    {
        const int a = g711_decode(w[k] & 0xFF, law);
        const int b = g711_decode((w[k] >> 8) & 0xFF, law);
        const int cc = g711_decode((w[k] >> 16) & 0xFF, law);
        const int d = g711_decode(w[k] >> 24, law);
        o[2*k] = ((uint32_t) a & 0xFFFFu) | ((uint32_t) b << 16);
        o[2*k + 1] = ((uint32_t) cc & 0xFFFFu) | ((uint32_t) d << 16);
    }
    ((uint4 *) dst)[2*i] = make_uint4(o[0], o[1], o[2], o[3]);
    ((uint4 *) dst)[2*i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
}

__global__ void linear_to_g711_kernel(const int16_t *src, uint8_t *dst, long long total16, int law)
{
    const long long i = (long long) blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= total16)
        return;
    const uint4 a = ((const uint4 *) src)[2*i];
    const uint4 b = ((const uint4 *) src)[2*i + 1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0;  k < 4;  k++)
    {
        const uint32_t c0 = (uint32_t) g711_encode((int) (int16_t) (w[2*k] & 0xFFFF), law) & 0xFF;
        const uint32_t c1 = (uint32_t) g711_encode((int) (int16_t) (w[2*k] >> 16), law) & 0xFF;
        const uint32_t c2 = (uint32_t) g711_encode((int) (int16_t) (w[2*k + 1] & 0xFFFF), law) & 0xFF;
        const uint32_t c3 = (uint32_t) g711_encode((int) (int16_t) (w[2*k + 1] >> 16), law) & 0xFF;
        o[k] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    }
    ((uint4 *) dst)[i] = make_uint4(o[0], o[1], o[2], o[3]);
}

}   // namespace

// D2H by a kernel: rows read from HBM and stored into the slot's pinned buffer through its device mapping.  The runtime's
// two copy directions share what they share (measured: an H2D and a D2H hipMemcpyAsync on two streams take nearly the sum of
// their times); stores from a kernel run beside the H2D copy engine.
typedef uint32_t feed_u32x4 __attribute__((ext_vector_type(4)));

__global__ void copy_out_kernel(feed_u32x4 *dst, const feed_u32x4 *src, size_t n16)
{
    for (size_t i = (size_t) blockIdx.x*blockDim.x + threadIdx.x;  i < n16;  i += (size_t) gridDim.x*blockDim.x)
        __builtin_nontemporal_store(src[i], &dst[i]);
}

struct spangpu_xfeed_s
{
    int what;                   // 1 echo, 2 modem
    int d2h_kernel;             // the way up by copy_out_kernel instead of hipMemcpyAsync
    void *h_out_dev[kFeedMaxDepth];
    spangpu_echo_t *echo;
    spangpu_modem_t *modem;
    int device;
    int n_ch;
    int max_samples;
    int law;
    int depth;
    long long stride;           // samples per row (rows 16-byte aligned in either format)
    size_t in_bytes;            // what travels down per tick
    size_t out_bytes;           // what travels up per tick
    int wpc;                    // modem: packed words per channel
    int status_cap;             // modem: status list entries
    int use_hpf_tx;
    void *h_in[kFeedMaxDepth];
    void *d_in[kFeedMaxDepth];
    void *h_out[kFeedMaxDepth];
    void *d_out[kFeedMaxDepth];
    int16_t *d_pcm_in;          // echo, G.711: the decoded tx and rx rows (one buffer: a tick's kernel has read it before the next decode, both on the bank's stream)
    int16_t *d_pcm_out;         // echo, G.711: the clean rows before encoding
    hipEvent_t ev_in[kFeedMaxDepth];
    hipEvent_t ev_k[kFeedMaxDepth];
    hipEvent_t ev_done[kFeedMaxDepth];
    bool busy[kFeedMaxDepth];
    int samples_of[kFeedMaxDepth];
    hipStream_t in_stream;
    hipStream_t out_stream;
    long long n_commit;
    long long n_collect;
};

typedef struct spangpu_xfeed_s spangpu_xfeed_t;

static int xfeed_destroy(spangpu_xfeed_t *f)
{
    if (f == nullptr)
        return SPANGPU_OK;
    (void) hipSetDevice(f->device);
    if (f->in_stream) (void) hipStreamSynchronize(f->in_stream);
    if (f->what == 1  &&  f->echo) (void) spangpu_echo_sync(f->echo);
    if (f->what == 2  &&  f->modem) (void) spangpu_modem_sync(f->modem);
    if (f->out_stream) (void) hipStreamSynchronize(f->out_stream);
    if (f->in_stream) (void) hipStreamDestroy(f->in_stream);
    if (f->out_stream) (void) hipStreamDestroy(f->out_stream);
    for (int k = 0;  k < f->depth;  k++)
    {
        if (f->h_in[k]) (void) hipHostFree(f->h_in[k]);
        if (f->d_in[k]) (void) hipFree(f->d_in[k]);
        if (f->h_out[k]) (void) hipHostFree(f->h_out[k]);
        if (f->d_out[k]) (void) hipFree(f->d_out[k]);
        if (f->ev_in[k]) (void) hipEventDestroy(f->ev_in[k]);
        if (f->ev_k[k]) (void) hipEventDestroy(f->ev_k[k]);
        if (f->ev_done[k]) (void) hipEventDestroy(f->ev_done[k]);
    }
    if (f->d_pcm_in) (void) hipFree(f->d_pcm_in);
    if (f->d_pcm_out) (void) hipFree(f->d_pcm_out);
    free(f);
    return SPANGPU_OK;
}

static int xfeed_copy_out(spangpu_xfeed_t *f, int slot)
{
    if (f->d2h_kernel)
    {
        const size_t n16 = (f->out_bytes + 15)/16;
        hipLaunchKernelGGL(copy_out_kernel, dim3(256), dim3(256), 0, f->out_stream, (feed_u32x4 *) f->h_out_dev[slot], (const feed_u32x4 *) f->d_out[slot], n16);
        FEED_TRY(hipGetLastError());
        return SPANGPU_OK;
    }
    FEED_TRY(hipMemcpyAsync(f->h_out[slot], f->d_out[slot], f->out_bytes, hipMemcpyDeviceToHost, f->out_stream));
    return SPANGPU_OK;
}

static int xfeed_alloc(spangpu_xfeed_t *f)
{
    // Measured (profiles/r4_feed_ab.log, 131 072 echo lines / 16 384 V.29 receivers per tick): the echo feed's 21 - 42 MB way
    // up is fastest as a DMA copy on a stream of its own priority (u-law 0.82 ms a tick against 1.08; int16 1.66 against 1.73),
    // the modem feed's 0.6 MB as a kernel's stores on a plain stream (0.20 ms against 0.35 - 0.49).
    const char *how = getenv("SPANGPU_FEED_D2H");           // "dma" / "kernel" for A-B runs
    f->d2h_kernel = (how == nullptr)  ?  (f->what == 2)  :  (strcmp(how, "kernel") == 0);
    f->out_bytes = (f->out_bytes + 15)/16*16;
    // The copy streams get a priority of their own: the runtime deals streams of one priority round over a handful of
    // hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and a copy stream that lands on the bank stream's queue runs in
    // turn with its kernels instead of beside them (measured: the way up of tick t then sits between the kernels of t and t + 1).
    int lo = 0;
    int hi = 0;
    (void) hipDeviceGetStreamPriorityRange(&lo, &hi);
    const char *pr = getenv("SPANGPU_FEED_PRIORITY");       // "0": plain streams (A-B runs)
    const bool prio = (pr == nullptr)  ?  (f->what == 1)  :  (strcmp(pr, "0") != 0);
    bool ok = (hipSetDevice(f->device) == hipSuccess)
              &&  ((prio  ?  hipStreamCreateWithPriority(&f->in_stream, hipStreamNonBlocking, hi)  :  hipStreamCreateWithFlags(&f->in_stream, hipStreamNonBlocking)) == hipSuccess)
              &&  ((prio  ?  hipStreamCreateWithPriority(&f->out_stream, hipStreamNonBlocking, hi)  :  hipStreamCreateWithFlags(&f->out_stream, hipStreamNonBlocking)) == hipSuccess);
    for (int k = 0;  ok  &&  k < f->depth;  k++)
    {
        ok = hipHostMalloc(&f->h_in[k], f->in_bytes) == hipSuccess
             &&  hipMalloc(&f->d_in[k], f->in_bytes + 64) == hipSuccess
             &&  hipHostMalloc(&f->h_out[k], f->out_bytes) == hipSuccess
             &&  hipMalloc(&f->d_out[k], f->out_bytes + 64) == hipSuccess
             &&  hipEventCreateWithFlags(&f->ev_in[k], hipEventDisableTiming) == hipSuccess
             &&  hipEventCreateWithFlags(&f->ev_k[k], hipEventDisableTiming) == hipSuccess
             &&  hipEventCreateWithFlags(&f->ev_done[k], hipEventDisableTiming) == hipSuccess;
        if (ok)
        {
            memset(f->h_in[k], 0, f->in_bytes);
            memset(f->h_out[k], 0, f->out_bytes);
            if (f->d2h_kernel  &&  hipHostGetDevicePointer(&f->h_out_dev[k], f->h_out[k], 0) != hipSuccess)
                f->d2h_kernel = 0;
        }
    }
    if (ok  &&  f->what == 1  &&  f->law)
    {
        const size_t rows = (size_t) f->stride*f->n_ch*sizeof(int16_t);
        ok = hipMalloc((void **) &f->d_pcm_in, 2*rows + 64) == hipSuccess  &&  hipMalloc((void **) &f->d_pcm_out, rows + 64) == hipSuccess;
    }
    return ok  ?  SPANGPU_OK  :  spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of (pinned) memory for the feed");
}

extern "C" {

// ---- echo ------------------------------------------------------------------------------------------------------------------
int spangpu_echo_feed_create(spangpu_echo_feed_t **out, spangpu_echo_t *ec, int max_samples, int law, int depth, int use_hpf_tx)
{
    if (out == nullptr  ||  ec == nullptr  ||  max_samples <= 0  ||  depth < 1  ||  depth > kFeedMaxDepth)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments (1 .. 8 slots)");
    if (law != 0  &&  law != SPANGPU_G711_ALAW  &&  law != SPANGPU_G711_ULAW)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "law must be 0 (16 bit linear), SPANGPU_G711_ALAW or SPANGPU_G711_ULAW");
    *out = nullptr;
    spangpu_xfeed_t *f = (spangpu_xfeed_t *) calloc(1, sizeof(*f));
    if (f == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of memory");
    f->what = 1;
    f->echo = ec;
    f->device = spangpu_echo_device(ec);
    f->n_ch = spangpu_echo_channels(ec);
    f->max_samples = max_samples;
    f->law = law;
    f->depth = depth;
    f->use_hpf_tx = use_hpf_tx;
    const int bps = law  ?  1  :  2;
    f->stride = ((long long) max_samples + 15)/16*16;            // rows of whole 16-sample groups in either format
    f->in_bytes = (size_t) 2*f->stride*bps*f->n_ch;
    f->out_bytes = (size_t) f->stride*bps*f->n_ch;
    const int rc = xfeed_alloc(f);
    if (rc != SPANGPU_OK)
    {
        xfeed_destroy(f);
        return rc;
    }
    *out = (spangpu_echo_feed_t *) f;
    return SPANGPU_OK;
}

int spangpu_echo_feed_destroy(spangpu_echo_feed_t *feed) { return xfeed_destroy((spangpu_xfeed_t *) feed); }

long long spangpu_echo_feed_stride(const spangpu_echo_feed_t *feed)
{
    return feed  ?  ((const spangpu_xfeed_t *) feed)->stride  :  (long long) SPANGPU_ERR_BAD_ARG;
}

// The pinned buffers of the next tick: *tx and *rx for the caller to write (channel c's samples at c*stride samples -- int16,
// or bytes with a G.711 law); NULL / SPANGPU_ERR_STATE while every slot holds a tick that has not been collected.
int spangpu_echo_feed_acquire(spangpu_echo_feed_t *feed, void **tx, void **rx)
{
    spangpu_xfeed_t *f = (spangpu_xfeed_t *) feed;
    if (f == nullptr  ||  f->what != 1  ||  tx == nullptr  ||  rx == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int slot = (int) (f->n_commit % f->depth);
    if (f->busy[slot])
        return spangpu_set_error(SPANGPU_ERR_STATE, "every slot of the feed holds a tick: collect one first");
    *tx = f->h_in[slot];
    *rx = (char *) f->h_in[slot] + f->in_bytes/2;
    return SPANGPU_OK;
}

int spangpu_echo_feed_commit(spangpu_echo_feed_t *feed, int samples)
{
    spangpu_xfeed_t *f = (spangpu_xfeed_t *) feed;
    if (f == nullptr  ||  f->what != 1  ||  samples <= 0  ||  samples > f->max_samples)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int slot = (int) (f->n_commit % f->depth);
    if (f->busy[slot])
        return spangpu_set_error(SPANGPU_ERR_STATE, "every slot of the feed holds a tick: collect one first");
    FEED_TRY(hipSetDevice(f->device));
    hipStream_t bs = (hipStream_t) spangpu_echo_get_stream(f->echo);
    FEED_TRY(hipMemcpyAsync(f->d_in[slot], f->h_in[slot], f->in_bytes, hipMemcpyHostToDevice, f->in_stream));
    FEED_TRY(hipEventRecord(f->ev_in[slot], f->in_stream));
    FEED_TRY(hipStreamWaitEvent(bs, f->ev_in[slot], 0));
    const size_t rows = (size_t) f->stride*f->n_ch;
    int rc;
    if (f->law)
    {
        const long long n16 = (long long) (2*rows/16);
        hipLaunchKernelGGL(g711_to_linear_kernel, dim3((unsigned) ((n16 + 255)/256)), dim3(256), 0, bs, (const uint8_t *) f->d_in[slot], f->d_pcm_in, n16, f->law);
        FEED_TRY(hipGetLastError());
        rc = spangpu_echo_update(f->echo, f->d_pcm_in, f->d_pcm_in + rows, f->d_pcm_out, SPANGPU_MEM_DEVICE, samples, f->stride, f->use_hpf_tx);
        if (rc < 0)
            return rc;
        const long long m16 = (long long) (rows/16);
        hipLaunchKernelGGL(linear_to_g711_kernel, dim3((unsigned) ((m16 + 255)/256)), dim3(256), 0, bs, (const int16_t *) f->d_pcm_out, (uint8_t *) f->d_out[slot], m16, f->law);
        FEED_TRY(hipGetLastError());
    }
    else
    {
        rc = spangpu_echo_update(f->echo, (const int16_t *) f->d_in[slot], (const int16_t *) f->d_in[slot] + rows, (int16_t *) f->d_out[slot],
                                 SPANGPU_MEM_DEVICE, samples, f->stride, f->use_hpf_tx);
        if (rc < 0)
            return rc;
    }
    FEED_TRY(hipEventRecord(f->ev_k[slot], bs));
    FEED_TRY(hipStreamWaitEvent(f->out_stream, f->ev_k[slot], 0));
    if ((rc = xfeed_copy_out(f, slot)) < 0)
        return rc;
    FEED_TRY(hipEventRecord(f->ev_done[slot], f->out_stream));
    f->busy[slot] = true;
    f->samples_of[slot] = samples;
    f->n_commit++;
    return SPANGPU_OK;
}

// The clean rows of the oldest tick not yet collected (waits for it): *clean is good until the slot is committed again.
// Returns the tick's samples per channel, 0 with *clean = NULL when no tick is outstanding.
int spangpu_echo_feed_collect(spangpu_echo_feed_t *feed, const void **clean)
{
    spangpu_xfeed_t *f = (spangpu_xfeed_t *) feed;
    if (f == nullptr  ||  f->what != 1  ||  clean == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *clean = nullptr;
    if (f->n_collect >= f->n_commit)
        return 0;
    const int slot = (int) (f->n_collect % f->depth);
    FEED_TRY(hipSetDevice(f->device));
    FEED_TRY(hipEventSynchronize(f->ev_done[slot]));
    f->busy[slot] = false;
    f->n_collect++;
    *clean = f->h_out[slot];
    return f->samples_of[slot];
}

int spangpu_echo_feed_outstanding(const spangpu_echo_feed_t *feed)
{
    const spangpu_xfeed_t *f = (const spangpu_xfeed_t *) feed;
    return f  ?  (int) (f->n_commit - f->n_collect)  :  SPANGPU_ERR_BAD_ARG;
}

// A caller's tick loop in C, for measurements: `ticks` x { acquire (the slots keep what they hold), commit, collect with
// `lag` ticks between a commit and the collect of its clean rows }, then the rest collected.
int spangpu_echo_feed_run(spangpu_echo_feed_t *feed, int samples, int ticks, int lag, double *elapsed_ms)
{
    spangpu_xfeed_t *f = (spangpu_xfeed_t *) feed;
    if (f == nullptr  ||  ticks <= 0  ||  lag < 1  ||  lag >= f->depth  ||  spangpu_echo_feed_outstanding(feed) != 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments (lag in 1 .. depth - 1, nothing outstanding)");
    struct timespec t0, t1;
    const void *clean;
    void *tx;
    void *rx;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0;  i < ticks;  i++)
    {
        int rc = spangpu_echo_feed_acquire(feed, &tx, &rx);
        if (rc < 0)
            return rc;
        if ((rc = spangpu_echo_feed_commit(feed, samples)) < 0)
            return rc;
        if (spangpu_echo_feed_outstanding(feed) > lag  &&  (rc = spangpu_echo_feed_collect(feed, &clean)) < 0)
            return rc;
    }
    while (spangpu_echo_feed_outstanding(feed) > 0)
    {
        const int rc = spangpu_echo_feed_collect(feed, &clean);
        if (rc < 0)
            return rc;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (elapsed_ms)
        *elapsed_ms = (t1.tv_sec - t0.tv_sec)*1e3 + (t1.tv_nsec - t0.tv_nsec)*1e-6;
    return SPANGPU_OK;
}

// ---- modem receivers ---------------------------------------------------------------------------------------------------------
int spangpu_modem_feed_create(spangpu_modem_feed_t **out, spangpu_modem_t *modem, int max_samples, int max_bit_rate, int depth)
{
    if (out == nullptr  ||  modem == nullptr  ||  max_samples <= 0  ||  max_bit_rate <= 0  ||  depth < 1  ||  depth > kFeedMaxDepth)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments (1 .. 8 slots)");
    *out = nullptr;
    spangpu_xfeed_t *f = (spangpu_xfeed_t *) calloc(1, sizeof(*f));
    if (f == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of memory");
    f->what = 2;
    f->modem = modem;
    f->device = spangpu_modem_device(modem);
    f->n_ch = spangpu_modem_channels(modem);
    f->max_samples = max_samples;
    f->depth = depth;
    f->stride = ((long long) max_samples + 7)/8*8;
    f->wpc = spangpu_modem_packed_words(max_bit_rate, max_samples);
    // What a tick can produce: a receiver reports at most a carrier drop and a new carrier with its training in progress in
    // 160 samples (SIG_STATUS_CARRIER_DOWN, _CARRIER_UP, _TRAINING_IN_PROGRESS; or _TRAINING_FAILED and _CARRIER_DOWN), and a bank
    // fed in step does so on every channel in the same tick -- two entries a channel, and one more for every further 160 samples
    // of the tick.  (One entry a channel, as this was, lost a tick's reports above 4 096 channels: the raw events of a tick are
    // gone once the next one runs.)
    const long long per_channel = 2 + (max_samples + 159)/160;
    const long long cap = (long long) f->n_ch*per_channel;
    f->status_cap = (int) ((cap > 4096)  ?  ((cap < 0x3FFFFFFF)  ?  cap  :  0x3FFFFFFF)  :  4096);
    f->in_bytes = (size_t) f->stride*sizeof(int16_t)*f->n_ch;
    f->out_bytes = ((size_t) f->n_ch*f->wpc + 1 + 2*(size_t) f->status_cap)*sizeof(uint32_t);
    const int rc = xfeed_alloc(f);
    if (rc != SPANGPU_OK)
    {
        xfeed_destroy(f);
        return rc;
    }
    *out = (spangpu_modem_feed_t *) f;
    return SPANGPU_OK;
}

int spangpu_modem_feed_destroy(spangpu_modem_feed_t *feed) { return xfeed_destroy((spangpu_xfeed_t *) feed); }

long long spangpu_modem_feed_stride(const spangpu_modem_feed_t *feed)
{
    return feed  ?  ((const spangpu_xfeed_t *) feed)->stride  :  (long long) SPANGPU_ERR_BAD_ARG;
}

int spangpu_modem_feed_words_per_channel(const spangpu_modem_feed_t *feed)
{
    return feed  ?  ((const spangpu_xfeed_t *) feed)->wpc  :  SPANGPU_ERR_BAD_ARG;
}

int spangpu_modem_feed_status_cap(const spangpu_modem_feed_t *feed)
{
    return feed  ?  ((const spangpu_xfeed_t *) feed)->status_cap  :  SPANGPU_ERR_BAD_ARG;
}

void *spangpu_modem_feed_acquire(spangpu_modem_feed_t *feed)
{
    spangpu_xfeed_t *f = (spangpu_xfeed_t *) feed;
    if (f == nullptr  ||  f->what != 2)
        return nullptr;
    const int slot = (int) (f->n_commit % f->depth);
    if (f->busy[slot])
    {
        spangpu_set_error(SPANGPU_ERR_STATE, "every slot of the feed holds a tick: collect one first");
        return nullptr;
    }
    return f->h_in[slot];
}

int spangpu_modem_feed_commit(spangpu_modem_feed_t *feed, int samples)
{
    spangpu_xfeed_t *f = (spangpu_xfeed_t *) feed;
    if (f == nullptr  ||  f->what != 2  ||  samples <= 0  ||  samples > f->max_samples)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int slot = (int) (f->n_commit % f->depth);
    if (f->busy[slot])
        return spangpu_set_error(SPANGPU_ERR_STATE, "every slot of the feed holds a tick: collect one first");
    FEED_TRY(hipSetDevice(f->device));
    hipStream_t bs = (hipStream_t) spangpu_modem_get_stream(f->modem);
    FEED_TRY(hipMemcpyAsync(f->d_in[slot], f->h_in[slot], f->in_bytes, hipMemcpyHostToDevice, f->in_stream));
    FEED_TRY(hipEventRecord(f->ev_in[slot], f->in_stream));
    FEED_TRY(hipStreamWaitEvent(bs, f->ev_in[slot], 0));
    int rc = spangpu_modem_rx(f->modem, (const int16_t *) f->d_in[slot], SPANGPU_MEM_DEVICE, samples, f->stride);
    if (rc < 0)
        return rc;
    uint32_t *packed = (uint32_t *) f->d_out[slot];
    uint32_t *status = packed + (size_t) f->n_ch*f->wpc;
    if ((rc = spangpu_modem_pack_events(f->modem, packed, f->wpc, status, f->status_cap)) < 0)
        return rc;
    FEED_TRY(hipEventRecord(f->ev_k[slot], bs));
    FEED_TRY(hipStreamWaitEvent(f->out_stream, f->ev_k[slot], 0));
    if ((rc = xfeed_copy_out(f, slot)) < 0)
        return rc;
    FEED_TRY(hipEventRecord(f->ev_done[slot], f->out_stream));
    f->busy[slot] = true;
    f->samples_of[slot] = samples;
    f->n_commit++;
    return SPANGPU_OK;
}

// The oldest tick's put_bit stream in packed form (waits for it): *packed = [n_ch][words_per_channel] rows, *status = the
// bank's status list (word 0 = entries), both good until the slot is committed again; spangpu_modem_unpack_events() turns
// them into the calls.  Returns the tick's samples per channel, 0 when no tick is outstanding.
int spangpu_modem_feed_collect(spangpu_modem_feed_t *feed, const uint32_t **packed, const uint32_t **status)
{
    spangpu_xfeed_t *f = (spangpu_xfeed_t *) feed;
    if (f == nullptr  ||  f->what != 2  ||  packed == nullptr  ||  status == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *packed = nullptr;
    *status = nullptr;
    if (f->n_collect >= f->n_commit)
        return 0;
    const int slot = (int) (f->n_collect % f->depth);
    FEED_TRY(hipSetDevice(f->device));
    FEED_TRY(hipEventSynchronize(f->ev_done[slot]));
    f->busy[slot] = false;
    f->n_collect++;
    *packed = (const uint32_t *) f->h_out[slot];
    *status = *packed + (size_t) f->n_ch*f->wpc;
    return f->samples_of[slot];
}

int spangpu_modem_feed_outstanding(const spangpu_modem_feed_t *feed)
{
    const spangpu_xfeed_t *f = (const spangpu_xfeed_t *) feed;
    return f  ?  (int) (f->n_commit - f->n_collect)  :  SPANGPU_ERR_BAD_ARG;
}

int spangpu_modem_feed_run(spangpu_modem_feed_t *feed, int samples, int ticks, int lag, double *elapsed_ms, long long *bits)
{
    spangpu_xfeed_t *f = (spangpu_xfeed_t *) feed;
    if (f == nullptr  ||  ticks <= 0  ||  lag < 1  ||  lag >= f->depth  ||  spangpu_modem_feed_outstanding(feed) != 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments (lag in 1 .. depth - 1, nothing outstanding)");
    struct timespec t0, t1;
    const uint32_t *packed;
    const uint32_t *status;
    long long seen = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0;  i <= ticks;  i++)
    {
        int rc;
        if (i < ticks)
        {
            if (spangpu_modem_feed_acquire(feed) == nullptr)
                return SPANGPU_ERR_STATE;
            if ((rc = spangpu_modem_feed_commit(feed, samples)) < 0)
                return rc;
        }
        while (spangpu_modem_feed_outstanding(feed) > ((i < ticks)  ?  lag  :  0))
        {
            if ((rc = spangpu_modem_feed_collect(feed, &packed, &status)) < 0)
                return rc;
            for (int c = 0;  c < f->n_ch;  c += 64)             // (a look at the headers of some rows: the data is there)
                seen += packed[(size_t) c*f->wpc] & 0x7FFFu;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (elapsed_ms)
        *elapsed_ms = (t1.tv_sec - t0.tv_sec)*1e3 + (t1.tv_nsec - t0.tv_nsec)*1e-6;
    if (bits)
        *bits = seen;
    return SPANGPU_OK;
}

}   // extern "C"
