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
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"

extern "C" int spangpu_set_error(int code, const char *msg);

#define SH_TRY(x) do { if ((x) != hipSuccess) return spangpu_set_error(SPANGPU_ERR_HIP, #x " failed"); } while (0)

enum { kMaxShards = 64 };

struct spangpu_shard_s
{
    int n;                              // shards
    int n_ch;                           // channels of the whole bank
    int kind;
    int collect_device;                 // where the gathered bytes go: the first shard's device
    int device[kMaxShards];
    int first[kMaxShards + 1];          // first channel of shard i; first[n] = n_ch
    spangpu_bank_t *bank[kMaxShards];
    uint8_t *digits[kMaxShards];        // [max_blocks][channels of the shard], on the shard's device
    hipEvent_t done[kMaxShards];        // the shard's bytes of the last step have arrived on the collecting device
    uint8_t *gathered;                  // on collect_device: shard-major, shard i's [max_blocks][n_i] at max_blocks*first[i]
    uint8_t *h_gathered;                // pinned host copy (spangpu_shard_digits_host)
    int max_blocks;
    int last_blocks;
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
        if (s->done[i]) (void) hipEventDestroy(s->done[i]);
    }
    (void) hipSetDevice(s->collect_device);
    if (s->gathered) (void) hipFree(s->gathered);
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
                ||  hipEventCreateWithFlags(&s->done[i], hipEventDisableTiming) != hipSuccess)
                rc = spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of device memory");
            else
                rc = spangpu_bank_set_digits_buffer(s->bank[i], s->digits[i], (size_t) s->max_blocks*mine);
        }
        if (rc == SPANGPU_OK  &&  devices[i] != s->collect_device)
        {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], s->collect_device) == hipSuccess  &&  can)
                (void) hipDeviceEnablePeerAccess(s->collect_device, 0);      // (already enabled is fine; without it the copy is staged)
            (void) hipGetLastError();
        }
    }
    if (rc == SPANGPU_OK)
    {
        if (hipSetDevice(s->collect_device) != hipSuccess
            ||  hipMalloc((void **) &s->gathered, (size_t) s->max_blocks*n_channels) != hipSuccess
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
    for (int i = 0;  i < s->n;  i++)
    {
        const int mine = s->first[i + 1] - s->first[i];
        SH_TRY(hipSetDevice(s->device[i]));
        const int rc = spangpu_bank_rx(s->bank[i], amp[i], SPANGPU_MEM_DEVICE, SPANGPU_LAYOUT_CHANNEL_MAJOR, samples, stride);
        if (rc < 0)
            return rc;
        hipStream_t st = (hipStream_t) spangpu_bank_get_stream(s->bank[i]);
        uint8_t *dst = s->gathered + (size_t) s->max_blocks*s->first[i];
        if (s->device[i] == s->collect_device)
            SH_TRY(hipMemcpyAsync(dst, s->digits[i], (size_t) maxb*mine, hipMemcpyDeviceToDevice, st));
        else
            SH_TRY(hipMemcpyPeerAsync(dst, s->collect_device, s->digits[i], s->device[i], (size_t) maxb*mine, st));
        SH_TRY(hipEventRecord(s->done[i], st));
    }
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
    for (int i = 0;  i < s->n;  i++)
    {
        if (hip_stream)
        {
            SH_TRY(hipSetDevice(s->collect_device));
            SH_TRY(hipStreamWaitEvent((hipStream_t) hip_stream, s->done[i], 0));
        }
        else
        {
            SH_TRY(hipSetDevice(s->device[i]));
            SH_TRY(hipEventSynchronize(s->done[i]));
        }
    }
    *digits = s->gathered;
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

}   // extern "C"
