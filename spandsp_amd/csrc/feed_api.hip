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
