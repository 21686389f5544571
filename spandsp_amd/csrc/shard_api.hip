// shard_api.hip -- one logical tone bank over several devices, behind the C ABI (include/spangpu.h: spangpu_shard_*).
//
// channels are independent, so a bank of N channels shards as contiguous channel ranges, n/G per device; every
// device owns its channels' state for their lifetime; inputs are delivered per device; nothing is exchanged between compute
// steps.  The one exchange is the gather of the per-channel results of a reporting interval to one device: here the digit
// byte of every block and channel (what bench.py's multi-GPU runs gather through RCCL from Python, spandsp_amd/parallel.py),
// written by each shard's detector kernel itself into a buffer on its own device and brought to the collecting device with
// hipMemcpyPeerAsync behind the kernel, on the shard's own stream -- device-to-device over xGMI where the devices are
// peers, no host in the path.  A C caller needs no torch.distributed for it.
//
// One host thread drives all shards: every call below only queues work (a launch and a copy per shard) and returns; the
// devices run side by side because each shard has a stream of its own on its own device.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"

extern "C" int spangpu_set_error(int code, const char *msg);

#define SH_TRY(x) do { if ((x) != hipSuccess) return spangpu_set_error(SPANGPU_ERR_HIP, #x " failed"); } while (0)

enum { kMaxShards = 64 };

// How shard i's results reach the collecting device: SPANGPU_LINK_SAME (it is the collecting device), SPANGPU_LINK_PEER (peer
// access is on: hipMemcpyPeerAsync goes device to device over xGMI) or SPANGPU_LINK_STAGED (the devices cannot reach each
// other, or enabling failed: the runtime stages the copy through host memory -- correct, and slow; spangpu_*_shard_info() says so).
static int link_to(int from_device, int to_device)
{
    if (from_device == to_device)
        return SPANGPU_LINK_SAME;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, from_device, to_device) != hipSuccess  ||  !can)
    {
        (void) hipGetLastError();
        return SPANGPU_LINK_STAGED;
    }
    // (the current device is from_device: it is given access to to_device's memory)
    const hipError_t e = hipDeviceEnablePeerAccess(to_device, 0);
    (void) hipGetLastError();
    return (e == hipSuccess  ||  e == hipErrorPeerAccessAlreadyEnabled)  ?  SPANGPU_LINK_PEER  :  SPANGPU_LINK_STAGED;
}

// Debug knob (spangpu_tune_force_peer_copy): a shard on the collecting device itself sends its results with hipMemcpyPeerAsync
// too (source and destination device equal: HIP allows it), so that the multi-device code path runs on a one-GPU box.
static std::atomic<int> g_force_peer{0};

static hipError_t gather_copy(void *dst, int dst_device, const void *src, int src_device, size_t bytes, hipStream_t st)
{
    if (src_device == dst_device  &&  !g_force_peer.load())
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    return hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, st);
}

static int shard_info(int n, const int *device, const int *first, const int *link, int collect_device, int i, spangpu_shard_info_t *info)
{
    if (i < 0  ||  i >= n  ||  info == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad shard");
    info->device = device[i];
    info->first_channel = first[i];
    info->n_channels = first[i + 1] - first[i];
    info->collect_device = collect_device;
    info->link = link[i];
    info->forced_peer_copy = g_force_peer.load();
    return SPANGPU_OK;
}

struct spangpu_shard_s
{
    int link[kMaxShards];               // SPANGPU_LINK_*: how shard i's bytes reach the collecting device
    int n;                              // shards
    int n_ch;                           // channels of the whole bank
    int kind;
    int collect_device;                 // where the gathered bytes go: the first shard's device
    int device[kMaxShards];
    int first[kMaxShards + 1];          // first channel of shard i; first[n] = n_ch
    spangpu_bank_t *bank[kMaxShards];
    uint8_t *digits[kMaxShards];        // [max_blocks][channels of the shard], on the shard's device
    hipEvent_t done[2][kMaxShards];     // the shard's bytes of a step have arrived on the collecting device (per slot)
    uint8_t *gathered[2];               // on collect_device: shard-major, shard i's [max_blocks][n_i] at max_blocks*first[i].  Two
                                        // slots used in turn: a step's bytes stay whole while the next step is queued and runs
                                        // (a reader that takes a step's bytes before the step after next is queued never sees a mix)
    uint8_t *h_gathered;                // pinned host copy (spangpu_shard_digits_host)
    int max_blocks;
    int last_blocks;
    unsigned steps;                     // spangpu_shard_rx() calls so far; the last one wrote slot (steps - 1) & 1
};

extern "C" {

int spangpu_shard_destroy(spangpu_shard_t *s)
{
    if (s == nullptr)
        return SPANGPU_OK;
    for (int i = 0;  i < s->n;  i++)
    {
        (void) hipSetDevice(s->device[i]);
        if (s->bank[i])
        {
            (void) spangpu_bank_sync(s->bank[i]);
            (void) spangpu_bank_destroy(s->bank[i]);
        }
        if (s->digits[i]) (void) hipFree(s->digits[i]);
        for (int k = 0;  k < 2;  k++)
        {
            if (s->done[k][i]) (void) hipEventDestroy(s->done[k][i]);
        }
    }
    (void) hipSetDevice(s->collect_device);
    for (int k = 0;  k < 2;  k++)
    {
        if (s->gathered[k]) (void) hipFree(s->gathered[k]);
    }
    if (s->h_gathered) (void) hipHostFree(s->h_gathered);
    free(s);
    return SPANGPU_OK;
}

// devices[i] is the HIP device of shard i (a device may appear more than once: two shards on one GPU, each with its own
// stream -- how a one-GPU box exercises this path).  Channels are dealt in contiguous ranges, as evenly as they go, in
// multiples of 64 (a wavefront's worth) except for the last shard.  max_samples sizes the digit buffers.
int spangpu_shard_create(spangpu_shard_t **out, const int *devices, int n_devices, int kind, int n_channels, int max_samples,
                         const void *params, size_t params_size)
{
    if (out == nullptr  ||  devices == nullptr  ||  n_devices < 1  ||  n_devices > kMaxShards  ||  n_channels < n_devices  ||  max_samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments (1 .. 64 shards, at least a channel each)");
    if (kind != SPANGPU_DTMF  &&  kind != SPANGPU_BELL_MF  &&  kind != SPANGPU_R2_MF)
        return spangpu_set_error(SPANGPU_ERR_UNSUPPORTED, "sharded banks: DTMF, Bell MF, R2 MF (the kinds that report digit bytes)");
    *out = nullptr;
    spangpu_shard_t *s = (spangpu_shard_t *) calloc(1, sizeof(*s));
    if (s == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of memory");
    s->n = n_devices;
    s->n_ch = n_channels;
    s->kind = kind;
    s->collect_device = devices[0];
    // shortest block of the kinds above: DTMF 102 samples
    s->max_blocks = max_samples/102 + 2;
    const int per = ((n_channels + n_devices - 1)/n_devices + 63)/64*64;
    int at = 0;
    for (int i = 0;  i < n_devices;  i++)
    {
        s->device[i] = devices[i];
        s->first[i] = at;
        int left = n_channels - at;
        int mine = (i == n_devices - 1)  ?  left  :  ((per < left - (n_devices - 1 - i))  ?  per  :  (left - (n_devices - 1 - i)));
        if (mine < 1)
            mine = 1;
        at += mine;
    }
    s->first[n_devices] = n_channels;
    if (at != n_channels)
    {
        free(s);
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "channels do not deal out over the shards");
    }
    int rc = SPANGPU_OK;
    for (int i = 0;  i < n_devices  &&  rc == SPANGPU_OK;  i++)
    {
        const int mine = s->first[i + 1] - s->first[i];
        if (hipSetDevice(devices[i]) != hipSuccess)
            rc = spangpu_set_error(SPANGPU_ERR_HIP, "hipSetDevice failed");
        else if ((rc = spangpu_bank_create(&s->bank[i], devices[i], kind, mine, params, params_size)) == SPANGPU_OK)
        {
            if (hipMalloc((void **) &s->digits[i], (size_t) s->max_blocks*mine) != hipSuccess
                ||  hipEventCreateWithFlags(&s->done[0][i], hipEventDisableTiming) != hipSuccess
                ||  hipEventCreateWithFlags(&s->done[1][i], hipEventDisableTiming) != hipSuccess)
                rc = spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of device memory");
            else
                rc = spangpu_bank_set_digits_buffer(s->bank[i], s->digits[i], (size_t) s->max_blocks*mine);
        }
        if (rc == SPANGPU_OK)
            s->link[i] = link_to(devices[i], s->collect_device);
    }
    if (rc == SPANGPU_OK)
    {
        if (hipSetDevice(s->collect_device) != hipSuccess
            ||  hipMalloc((void **) &s->gathered[0], (size_t) s->max_blocks*n_channels) != hipSuccess
            ||  hipMalloc((void **) &s->gathered[1], (size_t) s->max_blocks*n_channels) != hipSuccess
            ||  hipHostMalloc((void **) &s->h_gathered, (size_t) s->max_blocks*n_channels) != hipSuccess)
            rc = spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of memory for the gathered digits");
    }
    if (rc != SPANGPU_OK)
    {
        spangpu_shard_destroy(s);
        return rc;
    }
    *out = s;
    return SPANGPU_OK;
}

int spangpu_shard_count(const spangpu_shard_t *s) { return s  ?  s->n  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_shard_channels(const spangpu_shard_t *s) { return s  ?  s->n_ch  :  SPANGPU_ERR_BAD_ARG; }

// Shard i: its device, its first channel and how many it has; its bank (for everything a bank can do: parameters, state,
// records, a stream of the caller's choice ...).
int spangpu_shard_range(const spangpu_shard_t *s, int i, int *device, int *first_channel, int *n_channels)
{
    if (s == nullptr  ||  i < 0  ||  i >= s->n)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad shard");
    if (device) *device = s->device[i];
    if (first_channel) *first_channel = s->first[i];
    if (n_channels) *n_channels = s->first[i + 1] - s->first[i];
    return SPANGPU_OK;
}

int spangpu_shard_info(const spangpu_shard_t *s, int i, spangpu_shard_info_t *info)
{
    if (s == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null shard set");
    return shard_info(s->n, s->device, s->first, s->link, s->collect_device, i, info);
}

int spangpu_tune_force_peer_copy(int on)
{
    return g_force_peer.exchange(on  ?  1  :  0);
}

spangpu_bank_t *spangpu_shard_bank(spangpu_shard_t *s, int i)
{
    return (s  &&  i >= 0  &&  i < s->n)  ?  s->bank[i]  :  nullptr;
}

// One step of the whole bank: amp[i] = shard i's frame on ITS device (channel-major rows of `stride` samples, its own
// channels only), `samples` samples per channel.  Queues, per shard, the detector launch and the copy of its digit bytes to
// the collecting device behind it; returns without waiting.  Returns the blocks per channel the step can complete.
int spangpu_shard_rx(spangpu_shard_t *s, const int16_t *const *amp, int samples, long long stride)
{
    if (s == nullptr  ||  amp == nullptr  ||  samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int block = (s->kind == SPANGPU_DTMF)  ?  102  :  (s->kind == SPANGPU_BELL_MF)  ?  120  :  133;
    const int maxb = (samples + block - 1)/block;
    if (maxb > s->max_blocks)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "more samples than the shard was made for");
    const int slot = (int) (s->steps & 1u);
    for (int i = 0;  i < s->n;  i++)
    {
        const int mine = s->first[i + 1] - s->first[i];
        SH_TRY(hipSetDevice(s->device[i]));
        const int rc = spangpu_bank_rx(s->bank[i], amp[i], SPANGPU_MEM_DEVICE, SPANGPU_LAYOUT_CHANNEL_MAJOR, samples, stride);
        if (rc < 0)
            return rc;
        hipStream_t st = (hipStream_t) spangpu_bank_get_stream(s->bank[i]);
        uint8_t *dst = s->gathered[slot] + (size_t) s->max_blocks*s->first[i];
        SH_TRY(gather_copy(dst, s->collect_device, s->digits[i], s->device[i], (size_t) maxb*mine, st));
        SH_TRY(hipEventRecord(s->done[slot][i], st));
    }
    s->steps++;
    s->last_blocks = maxb;
    return maxb;
}

// Makes `hip_stream` (a stream of the collecting device; NULL: the calling host thread) wait until every shard's digit bytes
// of the last spangpu_shard_rx() have arrived, and hands out where they are: on the collecting device (the first shard's),
// shard-major -- shard i's bytes as [blocks][its channels] at offset max_blocks*first_channel(i); 0 = no digit in that block.
int spangpu_shard_digits_device(spangpu_shard_t *s, void *hip_stream, const uint8_t **digits, int *collect_device, int *max_blocks)
{
    if (s == nullptr  ||  digits == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (s->steps == 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no step has been queued yet");
    const int slot = (int) ((s->steps - 1u) & 1u);
    for (int i = 0;  i < s->n;  i++)
    {
        if (hip_stream)
        {
            SH_TRY(hipSetDevice(s->collect_device));
            SH_TRY(hipStreamWaitEvent((hipStream_t) hip_stream, s->done[slot][i], 0));
        }
        else
        {
            SH_TRY(hipSetDevice(s->device[i]));
            SH_TRY(hipEventSynchronize(s->done[slot][i]));
        }
    }
    *digits = s->gathered[slot];
    if (collect_device) *collect_device = s->collect_device;
    if (max_blocks) *max_blocks = s->max_blocks;
    return s->last_blocks;
}

// The same on the host, in the whole bank's channel order: out[b*n_channels + c] = the digit block b of the last step
// delivered on channel c (0 = none), b < the return value.
int spangpu_shard_digits_host(spangpu_shard_t *s, uint8_t *out, size_t out_bytes)
{
    const uint8_t *dev;
    if (s == nullptr  ||  out == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int nb = spangpu_shard_digits_device(s, nullptr, &dev, nullptr, nullptr);
    if (nb < 0)
        return nb;
    if (out_bytes < (size_t) nb*s->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "destination too small");
    SH_TRY(hipSetDevice(s->collect_device));
    SH_TRY(hipMemcpy(s->h_gathered, dev, (size_t) s->max_blocks*s->n_ch, hipMemcpyDeviceToHost));
    for (int i = 0;  i < s->n;  i++)
    {
        const int mine = s->first[i + 1] - s->first[i];
        const uint8_t *src = s->h_gathered + (size_t) s->max_blocks*s->first[i];
        for (int b = 0;  b < nb;  b++)
            memcpy(out + (size_t) b*s->n_ch + s->first[i], src + (size_t) b*mine, (size_t) mine);
    }
    return nb;
}

int spangpu_shard_sync(spangpu_shard_t *s)
{
    if (s == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null shard");
    for (int i = 0;  i < s->n;  i++)
    {
        SH_TRY(hipSetDevice(s->device[i]));
        const int rc = spangpu_bank_sync(s->bank[i]);
        if (rc < 0)
            return rc;
    }
    return SPANGPU_OK;
}


// ---- BASELINE configs[4]'s own object: the echo cancellers of N lines over several devices --------------------------------
// Channels in contiguous ranges (multiples of 64), every device owning its lines' taps, history and control words for their
// lifetime; per step one update launch per device on the shard's own stream (tx / rx rows in, clean rows out, each on its
// device); per reporting interval every shard turns its lines' energy sums into ERLE on its own device (echo_erle_kernel)
// and the floats travel to the collecting device (the first shard's) behind that, on the shard's stream: hipMemcpyPeerAsync,
// xGMI where the devices are peers.  Two result slots used in turn, as above.  (Reference: echo.c:421-661 per line; the
// ERLE is tests/echo_tests.c:577-594's level measurement, 10 log10(sum rx^2 / sum clean^2).)
struct spangpu_echo_shard_s
{
    int link[kMaxShards];               // SPANGPU_LINK_*
    int n;
    int n_ch;
    int collect_device;
    int device[kMaxShards];
    int first[kMaxShards + 1];
    spangpu_echo_t *bank[kMaxShards];
    float *erle[kMaxShards];            // [channels of the shard] on the shard's device
    hipEvent_t done[2][kMaxShards];
    float *gathered[2];                 // [n_ch] on collect_device, the whole bank's channel order
    unsigned reports;
};

// (the dealing of spangpu_shard_create(): contiguous ranges, multiples of 64 but for the last)
static int deal_channels(int n_channels, int n_devices, int *first)
{
    const int per = ((n_channels + n_devices - 1)/n_devices + 63)/64*64;
    int at = 0;
    for (int i = 0;  i < n_devices;  i++)
    {
        first[i] = at;
        const int left = n_channels - at;
        int mine = (i == n_devices - 1)  ?  left  :  ((per < left - (n_devices - 1 - i))  ?  per  :  (left - (n_devices - 1 - i)));
        if (mine < 1)
            mine = 1;
        at += mine;
    }
    first[n_devices] = n_channels;
    return (at == n_channels)  ?  SPANGPU_OK  :  SPANGPU_ERR_BAD_ARG;
}


int spangpu_echo_shard_destroy(spangpu_echo_shard_t *s)
{
    if (s == nullptr)
        return SPANGPU_OK;
    for (int i = 0;  i < s->n;  i++)
    {
        (void) hipSetDevice(s->device[i]);
        if (s->bank[i])
        {
            (void) spangpu_echo_sync(s->bank[i]);
            (void) spangpu_echo_destroy(s->bank[i]);
        }
        if (s->erle[i]) (void) hipFree(s->erle[i]);
        for (int k = 0;  k < 2;  k++)
        {
            if (s->done[k][i]) (void) hipEventDestroy(s->done[k][i]);
        }
    }
    (void) hipSetDevice(s->collect_device);
    for (int k = 0;  k < 2;  k++)
    {
        if (s->gathered[k]) (void) hipFree(s->gathered[k]);
    }
    free(s);
    return SPANGPU_OK;
}

int spangpu_echo_shard_create(spangpu_echo_shard_t **out, const int *devices, int n_devices, int n_channels, int taps, int adaption_mode)
{
    if (out == nullptr  ||  devices == nullptr  ||  n_devices < 1  ||  n_devices > kMaxShards  ||  n_channels < n_devices)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments (1 .. 64 shards, at least a channel each)");
    *out = nullptr;
    spangpu_echo_shard_t *s = (spangpu_echo_shard_t *) calloc(1, sizeof(*s));
    if (s == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of memory");
    s->n = n_devices;
    s->n_ch = n_channels;
    s->collect_device = devices[0];
    if (deal_channels(n_channels, n_devices, s->first) != SPANGPU_OK)
    {
        free(s);
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "channels do not deal out over the shards");
    }
    int rc = SPANGPU_OK;
    for (int i = 0;  i < n_devices  &&  rc == SPANGPU_OK;  i++)
    {
        const int mine = s->first[i + 1] - s->first[i];
        s->device[i] = devices[i];
        if (hipSetDevice(devices[i]) != hipSuccess)
            rc = spangpu_set_error(SPANGPU_ERR_HIP, "hipSetDevice failed");
        else if ((rc = spangpu_echo_create(&s->bank[i], devices[i], mine, taps, adaption_mode)) == SPANGPU_OK)
        {
            if (hipMalloc((void **) &s->erle[i], (size_t) mine*sizeof(float)) != hipSuccess
                ||  hipEventCreateWithFlags(&s->done[0][i], hipEventDisableTiming) != hipSuccess
                ||  hipEventCreateWithFlags(&s->done[1][i], hipEventDisableTiming) != hipSuccess)
                rc = spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of device memory");
            else
                rc = spangpu_echo_stats(s->bank[i], 2);         // the update kernel itself keeps the energy sums
        }
        if (rc == SPANGPU_OK)
            s->link[i] = link_to(devices[i], s->collect_device);
    }
    if (rc == SPANGPU_OK)
    {
        if (hipSetDevice(s->collect_device) != hipSuccess
            ||  hipMalloc((void **) &s->gathered[0], (size_t) n_channels*sizeof(float)) != hipSuccess
            ||  hipMalloc((void **) &s->gathered[1], (size_t) n_channels*sizeof(float)) != hipSuccess)
            rc = spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of memory for the gathered ERLE");
    }
    if (rc != SPANGPU_OK)
    {
        spangpu_echo_shard_destroy(s);
        return rc;
    }
    *out = s;
    return SPANGPU_OK;
}

int spangpu_echo_shard_count(const spangpu_echo_shard_t *s) { return s  ?  s->n  :  SPANGPU_ERR_BAD_ARG; }

int spangpu_echo_shard_range(const spangpu_echo_shard_t *s, int i, int *device, int *first_channel, int *n_channels)
{
    if (s == nullptr  ||  i < 0  ||  i >= s->n)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad shard");
    if (device) *device = s->device[i];
    if (first_channel) *first_channel = s->first[i];
    if (n_channels) *n_channels = s->first[i + 1] - s->first[i];
    return SPANGPU_OK;
}

int spangpu_echo_shard_info(const spangpu_echo_shard_t *s, int i, spangpu_shard_info_t *info)
{
    if (s == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null shard set");
    return shard_info(s->n, s->device, s->first, s->link, s->collect_device, i, info);
}

spangpu_echo_t *spangpu_echo_shard_bank(spangpu_echo_shard_t *s, int i)
{
    return (s  &&  i >= 0  &&  i < s->n)  ?  s->bank[i]  :  nullptr;
}

// One step: tx[i], rx[i], clean[i] = shard i's rows on ITS device (its own lines only), `samples` per line.  Queues one
// update launch per shard and returns.
int spangpu_echo_shard_update(spangpu_echo_shard_t *s, const int16_t *const *tx, const int16_t *const *rx, int16_t *const *clean,
                              int samples, long long stride)
{
    if (s == nullptr  ||  tx == nullptr  ||  rx == nullptr  ||  clean == nullptr  ||  samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    for (int i = 0;  i < s->n;  i++)
    {
        SH_TRY(hipSetDevice(s->device[i]));
        const int rc = spangpu_echo_update(s->bank[i], tx[i], rx[i], clean[i], SPANGPU_MEM_DEVICE, samples, stride, 0);
        if (rc < 0)
            return rc;
    }
    return SPANGPU_OK;
}

// A report: every shard's ERLE over the samples since its sums were last cleared, gathered to the collecting device (the
// whole bank's channel order); with `reset` the sums start again behind it.  Queues and returns.
int spangpu_echo_shard_report(spangpu_echo_shard_t *s, int reset)
{
    if (s == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null shard");
    const int slot = (int) (s->reports & 1u);
    for (int i = 0;  i < s->n;  i++)
    {
        const int mine = s->first[i + 1] - s->first[i];
        SH_TRY(hipSetDevice(s->device[i]));
        int rc = spangpu_echo_erle(s->bank[i], s->erle[i], SPANGPU_MEM_DEVICE);
        if (rc < 0)
            return rc;
        hipStream_t st = (hipStream_t) spangpu_echo_get_stream(s->bank[i]);
        float *dst = s->gathered[slot] + s->first[i];
        SH_TRY(gather_copy(dst, s->collect_device, s->erle[i], s->device[i], (size_t) mine*sizeof(float), st));
        SH_TRY(hipEventRecord(s->done[slot][i], st));
        if (reset  &&  (rc = spangpu_echo_stats_reset(s->bank[i], SPANGPU_ECHO_STATS_SUMS)) < 0)
            return rc;
    }
    s->reports++;
    return SPANGPU_OK;
}

// The last report: `hip_stream` (of the collecting device; NULL: the calling thread) waits for every shard's floats.
int spangpu_echo_shard_erle_device(spangpu_echo_shard_t *s, void *hip_stream, const float **erle_db, int *collect_device)
{
    if (s == nullptr  ||  erle_db == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (s->reports == 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no report has been queued yet");
    const int slot = (int) ((s->reports - 1u) & 1u);
    for (int i = 0;  i < s->n;  i++)
    {
        if (hip_stream)
        {
            SH_TRY(hipSetDevice(s->collect_device));
            SH_TRY(hipStreamWaitEvent((hipStream_t) hip_stream, s->done[slot][i], 0));
        }
        else
        {
            SH_TRY(hipSetDevice(s->device[i]));
            SH_TRY(hipEventSynchronize(s->done[slot][i]));
        }
    }
    *erle_db = s->gathered[slot];
    if (collect_device) *collect_device = s->collect_device;
    return s->n_ch;
}

int spangpu_echo_shard_erle_host(spangpu_echo_shard_t *s, float *out, size_t out_floats)
{
    const float *dev;
    if (s == nullptr  ||  out == nullptr  ||  out_floats < (size_t) s->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int n = spangpu_echo_shard_erle_device(s, nullptr, &dev, nullptr);
    if (n < 0)
        return n;
    SH_TRY(hipSetDevice(s->collect_device));
    SH_TRY(hipMemcpy(out, dev, (size_t) n*sizeof(float), hipMemcpyDeviceToHost));
    return n;
}

int spangpu_echo_shard_sync(spangpu_echo_shard_t *s)
{
    if (s == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null shard");
    for (int i = 0;  i < s->n;  i++)
    {
        SH_TRY(hipSetDevice(s->device[i]));
        const int rc = spangpu_echo_sync(s->bank[i]);
        if (rc < 0)
            return rc;
    }
    return SPANGPU_OK;
}

// ---- modem receivers over several devices: the put_bit streams of a step gathered to one device ---------------------------
// What travels per shard and step is what spangpu_modem_copy_events() lays out: int32 counts[n_i], then int8 events[n_i][per]
// (24 bytes per channel and 160-sample frame for V.29 at 9600 bit/s); shard i's block sits at byte offset
// (4 + per)*first_channel(i) of the collecting buffer.
struct spangpu_modem_shard_s
{
    int link[kMaxShards];               // SPANGPU_LINK_*
    int n;
    int n_ch;
    int per;
    int collect_device;
    int device[kMaxShards];
    int first[kMaxShards + 1];
    spangpu_modem_t *bank[kMaxShards];
    uint8_t *ev[kMaxShards];
    hipEvent_t done[2][kMaxShards];
    uint8_t *gathered[2];
    uint8_t *h_gathered;
    unsigned steps;
};

int spangpu_modem_shard_destroy(spangpu_modem_shard_t *s)
{
    if (s == nullptr)
        return SPANGPU_OK;
    for (int i = 0;  i < s->n;  i++)
    {
        (void) hipSetDevice(s->device[i]);
        if (s->bank[i])
        {
            (void) spangpu_modem_sync(s->bank[i]);
            (void) spangpu_modem_destroy(s->bank[i]);
        }
        if (s->ev[i]) (void) hipFree(s->ev[i]);
        for (int k = 0;  k < 2;  k++)
        {
            if (s->done[k][i]) (void) hipEventDestroy(s->done[k][i]);
        }
    }
    (void) hipSetDevice(s->collect_device);
    for (int k = 0;  k < 2;  k++)
    {
        if (s->gathered[k]) (void) hipFree(s->gathered[k]);
    }
    if (s->h_gathered) (void) hipHostFree(s->h_gathered);
    free(s);
    return SPANGPU_OK;
}

int spangpu_modem_shard_create(spangpu_modem_shard_t **out, const int *devices, int n_devices, int kind, int n_channels, int bit_rate,
                               int events_per_channel)
{
    if (out == nullptr  ||  devices == nullptr  ||  n_devices < 1  ||  n_devices > kMaxShards  ||  n_channels < n_devices
        ||  events_per_channel < 1  ||  events_per_channel > 4096)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments (1 .. 64 shards, at least a channel each, 1 .. 4096 events a channel and step)");
    *out = nullptr;
    spangpu_modem_shard_t *s = (spangpu_modem_shard_t *) calloc(1, sizeof(*s));
    if (s == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of memory");
    s->n = n_devices;
    s->n_ch = n_channels;
    s->per = events_per_channel;
    s->collect_device = devices[0];
    if (deal_channels(n_channels, n_devices, s->first) != SPANGPU_OK)
    {
        free(s);
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "channels do not deal out over the shards");
    }
    const size_t per_ch = 4u + (size_t) events_per_channel;
    int rc = SPANGPU_OK;
    for (int i = 0;  i < n_devices  &&  rc == SPANGPU_OK;  i++)
    {
        const int mine = s->first[i + 1] - s->first[i];
        s->device[i] = devices[i];
        if (hipSetDevice(devices[i]) != hipSuccess)
            rc = spangpu_set_error(SPANGPU_ERR_HIP, "hipSetDevice failed");
        else if ((rc = spangpu_modem_create(&s->bank[i], devices[i], kind, mine, bit_rate)) == SPANGPU_OK)
        {
            if (hipMalloc((void **) &s->ev[i], per_ch*mine) != hipSuccess
                ||  hipEventCreateWithFlags(&s->done[0][i], hipEventDisableTiming) != hipSuccess
                ||  hipEventCreateWithFlags(&s->done[1][i], hipEventDisableTiming) != hipSuccess)
                rc = spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of device memory");
        }
        if (rc == SPANGPU_OK)
            s->link[i] = link_to(devices[i], s->collect_device);
    }
    if (rc == SPANGPU_OK)
    {
        if (hipSetDevice(s->collect_device) != hipSuccess
            ||  hipMalloc((void **) &s->gathered[0], per_ch*n_channels) != hipSuccess
            ||  hipMalloc((void **) &s->gathered[1], per_ch*n_channels) != hipSuccess
            ||  hipHostMalloc((void **) &s->h_gathered, per_ch*n_channels) != hipSuccess)
            rc = spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of memory for the gathered events");
    }
    if (rc != SPANGPU_OK)
    {
        spangpu_modem_shard_destroy(s);
        return rc;
    }
    *out = s;
    return SPANGPU_OK;
}

int spangpu_modem_shard_range(const spangpu_modem_shard_t *s, int i, int *device, int *first_channel, int *n_channels)
{
    if (s == nullptr  ||  i < 0  ||  i >= s->n)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad shard");
    if (device) *device = s->device[i];
    if (first_channel) *first_channel = s->first[i];
    if (n_channels) *n_channels = s->first[i + 1] - s->first[i];
    return SPANGPU_OK;
}

int spangpu_modem_shard_info(const spangpu_modem_shard_t *s, int i, spangpu_shard_info_t *info)
{
    if (s == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null shard set");
    return shard_info(s->n, s->device, s->first, s->link, s->collect_device, i, info);
}

spangpu_modem_t *spangpu_modem_shard_bank(spangpu_modem_shard_t *s, int i)
{
    return (s  &&  i >= 0  &&  i < s->n)  ?  s->bank[i]  :  nullptr;
}

// One step: amp[i] = shard i's rows on its device; queues per shard the receiver launch, the copy of its event block and
// the block's trip to the collecting device, and returns.
int spangpu_modem_shard_rx(spangpu_modem_shard_t *s, const int16_t *const *amp, int samples, long long stride)
{
    if (s == nullptr  ||  amp == nullptr  ||  samples <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int slot = (int) (s->steps & 1u);
    const size_t per_ch = 4u + (size_t) s->per;
    for (int i = 0;  i < s->n;  i++)
    {
        const int mine = s->first[i + 1] - s->first[i];
        SH_TRY(hipSetDevice(s->device[i]));
        int rc = spangpu_modem_rx(s->bank[i], amp[i], SPANGPU_MEM_DEVICE, samples, stride);
        if (rc < 0)
            return rc;
        if ((rc = spangpu_modem_copy_events(s->bank[i], s->ev[i], per_ch*mine, s->per)) < 0)
            return rc;
        hipStream_t st = (hipStream_t) spangpu_modem_get_stream(s->bank[i]);
        uint8_t *dst = s->gathered[slot] + per_ch*s->first[i];
        SH_TRY(gather_copy(dst, s->collect_device, s->ev[i], s->device[i], per_ch*mine, st));
        SH_TRY(hipEventRecord(s->done[slot][i], st));
    }
    s->steps++;
    return SPANGPU_OK;
}

// The last step's events on the host, in the whole bank's channel order: counts[c] = put_bit calls of channel c in the step
// (bits and negative SIG_STATUS_* codes), events[c*per + k] = the k-th of them (k < min(counts[c], per)).
int spangpu_modem_shard_events_host(spangpu_modem_shard_t *s, int32_t *counts, int8_t *events)
{
    if (s == nullptr  ||  counts == nullptr  ||  events == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (s->steps == 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no step has been queued yet");
    const int slot = (int) ((s->steps - 1u) & 1u);
    const size_t per_ch = 4u + (size_t) s->per;
    for (int i = 0;  i < s->n;  i++)
    {
        SH_TRY(hipSetDevice(s->device[i]));
        SH_TRY(hipEventSynchronize(s->done[slot][i]));
    }
    SH_TRY(hipSetDevice(s->collect_device));
    SH_TRY(hipMemcpy(s->h_gathered, s->gathered[slot], per_ch*s->n_ch, hipMemcpyDeviceToHost));
    for (int i = 0;  i < s->n;  i++)
    {
        const int mine = s->first[i + 1] - s->first[i];
        const uint8_t *blk = s->h_gathered + per_ch*s->first[i];
        memcpy(counts + s->first[i], blk, (size_t) mine*4u);
        memcpy(events + (size_t) s->first[i]*s->per, blk + (size_t) mine*4u, (size_t) mine*s->per);
    }
    return s->n_ch;
}

int spangpu_modem_shard_sync(spangpu_modem_shard_t *s)
{
    if (s == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null shard");
    for (int i = 0;  i < s->n;  i++)
    {
        SH_TRY(hipSetDevice(s->device[i]));
        const int rc = spangpu_modem_sync(s->bank[i]);
        if (rc < 0)
            return rc;
    }
    return SPANGPU_OK;
}

}   // extern "C"
