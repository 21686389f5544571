// echo_api.hip -- C ABI of the batched G.168 line echo canceller (include/spangpu.h,
// "echo canceller banks").  Device code: echo_dev.hpp.  No CPU implementation exists
// behind these entry points.

#include <hip/hip_runtime.h>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"
#include "../../include/spangpu_refstate.h"
#include "echo_dev.hpp"
#include "echo_pair.hpp"

using namespace spg;

extern "C" int spangpu_set_error(int code, const char *msg);

#define ECHO_TRY(expr)                                                                      \
    do                                                                                      \
    {                                                                                       \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
        {                                                                                   \
            char m_[256];                                                                   \
            snprintf(m_, sizeof(m_), "%s failed: %s", #expr, hipGetErrorString(e_));        \
            return spangpu_set_error(SPANGPU_ERR_HIP, m_);                                  \
        }                                                                                   \
    }                                                                                       \
    while (0)

struct spangpu_echo_s
{
    int device;
    int n_ch;
    int taps;
    int tpl;
    int group;              // lanes per channel: 16, 8 or 4
    hipStream_t stream;
    bool own_stream;
    int32_t *scal;
    int32_t *taps32;
    int16_t *taps16;
    int16_t *hist;
    int16_t *d_io;          // staging for host-resident tx / rx / clean: [3][n_ch][cap]
    size_t io_cap;          // samples per channel
    EchoStats *stats;       // per-channel line statistics, allocated by spangpu_echo_stats(ec, 1)
    int stats_on;           // 0 off, 1 energy sums and CRC by a pass of their own after the update, 2 energy sums only, by the update kernel itself
    float *d_erle;          // scratch for spangpu_echo_erle() with a host destination
    int uniform_mode;       // the adaption mode every channel has, or -1 when they differ: picks the kernel compiled for that mode
    bool mode_dirty;        // a single channel's mode was written since the channels were last compared: they may all agree again
    int *d_span;            // two ints of device scratch for that comparison (made with the bank: nothing is allocated or freed in the update path)
};

__global__ void echo_set_scalar_kernel(int32_t *scal, int lo, int hi, int idx, int value)
{
    const int c = lo + blockIdx.x*blockDim.x + threadIdx.x;
    if (c < hi)
        scal[(size_t) c*kEchoScalars + idx] = value;
}

// Do the channels' adaption modes agree?  out[0] = the smallest, out[1] = the largest (out preset to INT_MAX, INT_MIN).
__global__ void echo_mode_span_kernel(const int32_t *scal, int n_ch, int *out)
{
    const int c = blockIdx.x*blockDim.x + threadIdx.x;
    if (c < n_ch)
    {
        const int m = scal[(size_t) c*kEchoScalars + ES_ADAPTION_MODE];
        atomicMin(&out[0], m);
        atomicMax(&out[1], m);
    }
}

// After single channels' modes were written (spangpu_echo_adaption_mode(ch), spangpu_echo_set_state) the bank runs the
// general kernel; once the channels all agree again, the kernel compiled for that mode is 10 - 18 % faster.  Looked at once
// per such edit, at the next update.
static void refresh_uniform_mode(spangpu_echo_t *e)
{
    e->mode_dirty = false;
    if (e->uniform_mode >= 0  ||  e->d_span == nullptr)
        return;
    int h[2] = {0x7FFFFFFF, (int) 0x80000000};
    if (hipMemcpyAsync(e->d_span, h, sizeof(h), hipMemcpyHostToDevice, e->stream) == hipSuccess)
    {
        hipLaunchKernelGGL(echo_mode_span_kernel, dim3((e->n_ch + 255)/256), dim3(256), 0, e->stream, (const int32_t *) e->scal, e->n_ch, e->d_span);
        if (hipMemcpyAsync(h, e->d_span, sizeof(h), hipMemcpyDeviceToHost, e->stream) == hipSuccess
            &&  hipStreamSynchronize(e->stream) == hipSuccess  &&  h[0] == h[1])
            e->uniform_mode = h[0];
        else if (h[0] > h[1])
            e->mode_dirty = true;       // the comparison did not run (a HIP error): the general kernel stays, and the next update looks again
    }
    else
    {
        e->mode_dirty = true;
    }
    (void) hipGetLastError();
}

static void init_scalars(int32_t *s, int taps, int mode)
{
    // echo_can_init(), echo.c:254-301
    memset(s, 0, sizeof(int32_t)*kEchoScalars);
    s[ES_TAPS] = taps;
    s[ES_CURR_POS] = taps - 1;
    s[ES_FIR_CURR_POS] = taps - 1;
    s[ES_TAP_MASK] = taps - 1;
    s[ES_RX_POWER_THRESHOLD] = 10000000;
    s[ES_TAP_ROTATE_COUNTER] = 1600;
    s[ES_CNG_LEVEL] = 1000;
    s[ES_ADAPTION_MODE] = mode;
}

static std::atomic<int> g_echo_group{0};      // spangpu_tune_echo_lanes_per_channel(): process-wide, read once at bank creation

extern "C" {

// Tuning / A-B testing: lanes per channel of banks created from now on (0 = choose by length, 8, 16).
int spangpu_tune_echo_lanes_per_channel(int lanes)
{
    if (lanes != 0  &&  lanes != 2  &&  lanes != 4  &&  lanes != 8  &&  lanes != 16)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "lanes per channel must be 0 (auto), 2, 4, 8 or 16");
    g_echo_group = lanes;
    return SPANGPU_OK;
}

int spangpu_echo_create(spangpu_echo_t **out, int device, int n_channels, int taps, int adaption_mode)
{
    if (out == nullptr  ||  n_channels <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *out = nullptr;
    if (taps != 32  &&  taps != 64  &&  taps != 128  &&  taps != 256  &&  taps != 512  &&  taps != 1024)
        return spangpu_set_error(SPANGPU_ERR_UNSUPPORTED, "echo canceller length must be 32, 64, 128, 256, 512 or 1024 taps");
    if (spangpu_device_count() <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    if (device < 0  ||  device >= spangpu_device_count())
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "device out of range");
    ECHO_TRY(hipSetDevice(device));
    spangpu_echo_t *e = (spangpu_echo_t *) calloc(1, sizeof(*e));
    if (e == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    e->device = device;
    e->n_ch = n_channels;
    e->taps = taps;
    // Lanes per channel.  The scalar control of echo_can_update() is replicated in a channel's lanes, so the fewer lanes a
    // channel has the more channels share each control instruction; a small bank wants the opposite -- more,
    // narrower-sliced waves, so that every SIMD has some.  Measured, 128 taps, kernel time in us per 160-sample frame for
    // 16 / 8 / 4 / 2 lanes (two lanes = echo_pair.hpp, the 16-bit quantities packed in pairs):
    //   every channel adapting in step (tools/echo_ab.py): 4096 channels 100 / 118 / 104 / -, 8192: 131 / 122 / 122 / -,
    //     16384: 182 / 153 / 183 / -, 32768: - / 215 / 193 / 207, 65536: - / 384 / 347 / 293, 131072: 977 / 723 / 597 / 594
    //   mixed lines -- single talk, double talk, silence, DC offsets, so that set events fall on different samples in
    //     different channels (tools/bench_paths.py --workload echo --echo-lanes G): 32768: - / 232 / 218 / 265,
    //     65536: - / 400 / 370 / 361, 131072: - / 737 / 624 / 660
    //   round 4 (rounds leave through LDS, dealt power meters, one conditional region a sample; profiles/r4_echo_lanes.log), mixed
    //     G.168 lines, 4 / 8 / 16 lanes: 4096: 113 / 88 / 76, 8192: 126 / 90 / 91, 16384: 133 / 114 / 142, 32768: 170 / 194 / 244,
    //     131072: 489 at four lanes -- the crossovers stand (with the kernels compiled for mode 0x01 at 8 and 16 lanes as well,
    //     profiles/r4_echo_lanes_mode_kernels.log: 4096: - / 76 / 62, 8192: - / 79 / 78, 16384: - / 103 / 119)
    // The two-lane kernel executes 10.7 VALU instructions per channel and sample against 15.6 at four lanes and 23.1 at
    // eight (profiles/r2_echo_pmc.txt), but a sample on which ANY of a wave's channels meets a set event takes the whole
    // wave through the complete routine, and its waves hold 32 channels: on mixed lines that eats the gain.  So four
    // lanes from 24576 channels, eight from 8192, sixteen below; two lanes on request
    // (spangpu_tune_echo_lanes_per_channel()), for banks whose channels run in step.
    // Slices are at most 32 taps (four lanes) or 16 taps per lane; two lanes take 32, 64 or 128 taps.
    const int tuned = g_echo_group.load(std::memory_order_relaxed);
    e->group = (tuned != 0)  ?  tuned  :  (n_channels >= 24576)  ?  4  :  (n_channels >= 8192)  ?  8  :  16;
    if (taps > 256)
        e->group = 16;          // 64 and 128 ms tails: a DPP row of lanes per channel, slices of 32 / 64 taps (echo.c:254: any power of two)
    if (e->group == 2  &&  taps != 128  &&  taps != 64  &&  taps != 32)
        e->group = (tuned != 0  ||  n_channels >= 131072)  ?  4  :  8;
    if (e->group == 4  &&  (taps/4 < 2  ||  taps/4 > 32))
        e->group = 8;
    if (e->group == 8  &&  (taps/8 < 2  ||  taps/8 > 16))
        e->group = 16;
    e->tpl = taps/e->group;
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
    {
        free(e);
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipStreamCreate failed");
    }
    e->own_stream = true;
    const size_t n = (size_t) n_channels;
    if (hipMalloc(&e->scal, n*kEchoScalars*sizeof(int32_t)) != hipSuccess
        ||  hipMalloc(&e->taps32, n*taps*sizeof(int32_t)) != hipSuccess
        ||  hipMalloc(&e->taps16, n*4*taps*sizeof(int16_t)) != hipSuccess
        ||  hipMalloc(&e->hist, n*taps*sizeof(int16_t)) != hipSuccess
        ||  hipMalloc((void **) &e->d_span, 2*sizeof(int)) != hipSuccess)
    {
        spangpu_echo_destroy(e);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "hipMalloc of echo state failed");
    }
    (void) hipMemsetAsync(e->taps32, 0, n*taps*sizeof(int32_t), e->stream);
    (void) hipMemsetAsync(e->taps16, 0, n*4*taps*sizeof(int16_t), e->stream);
    (void) hipMemsetAsync(e->hist, 0, n*taps*sizeof(int16_t), e->stream);
    int32_t *h = (int32_t *) malloc(n*kEchoScalars*sizeof(int32_t));
    if (h == nullptr)
    {
        spangpu_echo_destroy(e);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "malloc");
    }
    for (size_t c = 0;  c < n;  c++)
        init_scalars(h + c*kEchoScalars, taps, adaption_mode);
    e->uniform_mode = adaption_mode;
    hipError_t rc = hipMemcpyAsync(e->scal, h, n*kEchoScalars*sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
    (void) hipStreamSynchronize(e->stream);
    free(h);
    if (rc != hipSuccess)
    {
        spangpu_echo_destroy(e);
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    }
    *out = e;
    return SPANGPU_OK;
}

int spangpu_echo_destroy(spangpu_echo_t *e)
{
    if (e == nullptr)
        return SPANGPU_OK;
    (void) hipSetDevice(e->device);
    if (e->stream)
        (void) hipStreamSynchronize(e->stream);
    if (e->scal) (void) hipFree(e->scal);
    if (e->taps32) (void) hipFree(e->taps32);
    if (e->taps16) (void) hipFree(e->taps16);
    if (e->hist) (void) hipFree(e->hist);
    if (e->d_io) (void) hipFree(e->d_io);
    if (e->stats) (void) hipFree(e->stats);
    if (e->d_erle) (void) hipFree(e->d_erle);
    if (e->d_span) (void) hipFree(e->d_span);
    if (e->own_stream  &&  e->stream)
        (void) hipStreamDestroy(e->stream);
    free(e);
    return SPANGPU_OK;
}

int spangpu_echo_channels(const spangpu_echo_t *e) { return e  ?  e->n_ch  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_echo_taps(const spangpu_echo_t *e) { return e  ?  e->taps  :  SPANGPU_ERR_BAD_ARG; }
int spangpu_echo_lanes_per_channel(const spangpu_echo_t *e) { return e  ?  e->group  :  SPANGPU_ERR_BAD_ARG; }

void *spangpu_echo_get_stream(spangpu_echo_t *e) { return e  ?  (void *) e->stream  :  nullptr; }
int spangpu_echo_device(const spangpu_echo_t *e) { return e  ?  e->device  :  SPANGPU_ERR_BAD_ARG; }

int spangpu_echo_set_stream(spangpu_echo_t *e, void *hip_stream)
{
    if (e == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    (void) hipStreamSynchronize(e->stream);
    if (e->own_stream)
        (void) hipStreamDestroy(e->stream);
    if (hip_stream)
    {
        e->stream = (hipStream_t) hip_stream;
        e->own_stream = false;
    }
    else
    {
        ECHO_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
    }
    return SPANGPU_OK;
}

int spangpu_echo_sync(spangpu_echo_t *e)
{
    if (e == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    ECHO_TRY(hipStreamSynchronize(e->stream));
    return SPANGPU_OK;
}

int spangpu_echo_update(spangpu_echo_t *e, const int16_t *tx, const int16_t *rx, int16_t *clean,
                        int mem, int samples, long long stride, int use_hpf_tx)
{
    return spangpu_echo_update_tx(e, tx, rx, clean, nullptr, mem, samples, stride, use_hpf_tx);
}

int spangpu_echo_update_tx(spangpu_echo_t *e, const int16_t *tx, const int16_t *rx, int16_t *clean, int16_t *tx_out,
                           int mem, int samples, long long stride, int use_hpf_tx)
{
    if (e == nullptr  ||  tx == nullptr  ||  rx == nullptr  ||  clean == nullptr  ||  samples < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (samples == 0)
        return 0;
    if (stride <= 0)
        stride = samples;
    ECHO_TRY(hipSetDevice(e->device));
    EchoLaunch L;
    memset(&L, 0, sizeof(L));
    if (mem == SPANGPU_MEM_HOST)
    {
        if ((size_t) samples > e->io_cap)
        {
            if (e->d_io) (void) hipFree(e->d_io);
            e->d_io = nullptr;
            e->io_cap = 0;
            ECHO_TRY(hipMalloc(&e->d_io, (size_t) 4*e->n_ch*samples*sizeof(int16_t)));
            e->io_cap = samples;
        }
        int16_t *dtx = e->d_io;
        int16_t *drx = e->d_io + (size_t) e->n_ch*e->io_cap;
        int16_t *dcl = e->d_io + (size_t) 2*e->n_ch*e->io_cap;
        ECHO_TRY(hipMemcpy2DAsync(dtx, e->io_cap*sizeof(int16_t), tx, stride*sizeof(int16_t), samples*sizeof(int16_t),
                                  e->n_ch, hipMemcpyHostToDevice, e->stream));
        ECHO_TRY(hipMemcpy2DAsync(drx, e->io_cap*sizeof(int16_t), rx, stride*sizeof(int16_t), samples*sizeof(int16_t),
                                  e->n_ch, hipMemcpyHostToDevice, e->stream));
        L.tx = dtx;
        L.rx = drx;
        L.clean = dcl;
        L.tx_out = tx_out  ?  (e->d_io + (size_t) 3*e->n_ch*e->io_cap)  :  nullptr;
        L.stride = (long long) e->io_cap;
    }
    else if (mem == SPANGPU_MEM_DEVICE)
    {
        L.tx = tx;
        L.rx = rx;
        L.clean = clean;
        L.tx_out = tx_out;
        L.stride = stride;
    }
    else
    {
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    }
    if (e->mode_dirty)
        refresh_uniform_mode(e);
    L.samples = samples;
    L.n_ch = e->n_ch;
    L.use_hpf_tx = use_hpf_tx;
    L.scal = e->scal;
    L.taps32 = e->taps32;
    L.taps16 = e->taps16;
    L.hist = e->hist;
    L.stats = (e->stats_on == 2)  ?  e->stats  :  nullptr;
    const int per_wave = 64/e->group;
    const int waves = (e->n_ch + per_wave - 1)/per_wave;
    const int blocks = (waves + 3)/4;
    if (e->group == 2)
    {
        switch (e->tpl)
        {
        case 16: hipLaunchKernelGGL(echo_pair_kernel<16>, dim3(blocks), dim3(256), 0, e->stream, L); break;
        case 32: hipLaunchKernelGGL(echo_pair_kernel<32>, dim3(blocks), dim3(256), 0, e->stream, L); break;
        default: hipLaunchKernelGGL(echo_pair_kernel<64>, dim3(blocks), dim3(256), 0, e->stream, L); break;
        }
    }
    else if (e->group == 4)
    {
        switch (e->tpl)
        {
        case 8:  hipLaunchKernelGGL((echo_bank_kernel<8, 4>), dim3(blocks), dim3(256), 0, e->stream, L);  break;
        case 16: hipLaunchKernelGGL((echo_bank_kernel<16, 4>), dim3(blocks), dim3(256), 0, e->stream, L); break;
        default:
            // the kernels compiled for one mode: the three echo_tests.c runs its lines in (adaption alone is SURVEY 8(d)-5's,
            // and what a bank has until somebody changes it); any other mode, or lines of different modes: the general one
            if (e->uniform_mode == kModeAdaption)
                hipLaunchKernelGGL((echo_bank_kernel<32, 4, kModeAdaption>), dim3(blocks), dim3(256), 0, e->stream, L);
            else if (e->uniform_mode == (kModeAdaption | kModeNlp))
                hipLaunchKernelGGL((echo_bank_kernel<32, 4, kModeAdaption | kModeNlp>), dim3(blocks), dim3(256), 0, e->stream, L);
            else if (e->uniform_mode == (kModeAdaption | kModeNlp | kModeCng))
                hipLaunchKernelGGL((echo_bank_kernel<32, 4, kModeAdaption | kModeNlp | kModeCng>), dim3(blocks), dim3(256), 0, e->stream, L);
            else
                hipLaunchKernelGGL((echo_bank_kernel<32, 4>), dim3(blocks), dim3(256), 0, e->stream, L);
            break;
        }
    }
    else if (e->group == 8)
    {
        switch (e->tpl)
        {
        case 4:  hipLaunchKernelGGL((echo_bank_kernel<4, 8>), dim3(blocks), dim3(256), 0, e->stream, L);  break;
        case 8:  hipLaunchKernelGGL((echo_bank_kernel<8, 8>), dim3(blocks), dim3(256), 0, e->stream, L);  break;
        default:
            if (e->uniform_mode == kModeAdaption)
                hipLaunchKernelGGL((echo_bank_kernel<16, 8, kModeAdaption>), dim3(blocks), dim3(256), 0, e->stream, L);
            else
                hipLaunchKernelGGL((echo_bank_kernel<16, 8>), dim3(blocks), dim3(256), 0, e->stream, L);
            break;
        }
    }
    else
    {
        switch (e->tpl)
        {
        case 2:  hipLaunchKernelGGL((echo_bank_kernel<2, 16>), dim3(blocks), dim3(256), 0, e->stream, L);  break;
        case 4:  hipLaunchKernelGGL((echo_bank_kernel<4, 16>), dim3(blocks), dim3(256), 0, e->stream, L);  break;
        case 8:
            if (e->uniform_mode == kModeAdaption)
                hipLaunchKernelGGL((echo_bank_kernel<8, 16, kModeAdaption>), dim3(blocks), dim3(256), 0, e->stream, L);
            else
                hipLaunchKernelGGL((echo_bank_kernel<8, 16>), dim3(blocks), dim3(256), 0, e->stream, L);
            break;
        case 32: hipLaunchKernelGGL((echo_bank_kernel<32, 16>), dim3(blocks), dim3(256), 0, e->stream, L); break;
        case 64: hipLaunchKernelGGL((echo_bank_kernel<64, 16>), dim3(blocks), dim3(256), 0, e->stream, L); break;
        default: hipLaunchKernelGGL((echo_bank_kernel<16, 16>), dim3(blocks), dim3(256), 0, e->stream, L); break;
        }
    }
    ECHO_TRY(hipGetLastError());
    if (e->stats_on == 1)
    {
        hipLaunchKernelGGL(echo_stats_kernel, dim3((e->n_ch + 255)/256), dim3(256), 0, e->stream,
                           L.rx, (const int16_t *) L.clean, L.stride, samples, e->n_ch, e->stats);
        ECHO_TRY(hipGetLastError());
    }
    if (mem == SPANGPU_MEM_HOST)
    {
        ECHO_TRY(hipMemcpy2DAsync(clean, stride*sizeof(int16_t), L.clean, e->io_cap*sizeof(int16_t), samples*sizeof(int16_t),
                                  e->n_ch, hipMemcpyDeviceToHost, e->stream));
        if (tx_out)
            ECHO_TRY(hipMemcpy2DAsync(tx_out, stride*sizeof(int16_t), L.tx_out, e->io_cap*sizeof(int16_t), samples*sizeof(int16_t),
                                      e->n_ch, hipMemcpyDeviceToHost, e->stream));
        ECHO_TRY(hipStreamSynchronize(e->stream));
    }
    return 0;
}

// echo_can_hpf_tx() for every channel, separately from the update (the spandsp calling sequence: tx' = hpf_tx(tx), send
// tx' to the line, later clean = update(tx', rx)).  Host buffers only; out may alias tx.
int spangpu_echo_hpf_tx(spangpu_echo_t *e, const int16_t *tx, int16_t *out, int samples, long long stride)
{
    if (e == nullptr  ||  tx == nullptr  ||  out == nullptr  ||  samples < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (samples == 0)
        return 0;
    if (stride <= 0)
        stride = samples;
    ECHO_TRY(hipSetDevice(e->device));
    if ((size_t) samples > e->io_cap)
    {
        if (e->d_io) (void) hipFree(e->d_io);
        e->d_io = nullptr;
        e->io_cap = 0;
        ECHO_TRY(hipMalloc(&e->d_io, (size_t) 4*e->n_ch*samples*sizeof(int16_t)));
        e->io_cap = samples;
    }
    int16_t *dtx = e->d_io;
    int16_t *dout = e->d_io + (size_t) 3*e->n_ch*e->io_cap;
    ECHO_TRY(hipMemcpy2DAsync(dtx, e->io_cap*sizeof(int16_t), tx, stride*sizeof(int16_t), samples*sizeof(int16_t),
                              e->n_ch, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(echo_hpf_tx_kernel, dim3((e->n_ch + 63)/64), dim3(64), 0, e->stream, dtx, dout, (long long) e->io_cap,
                       samples, e->n_ch, e->scal);
    ECHO_TRY(hipGetLastError());
    ECHO_TRY(hipMemcpy2DAsync(out, stride*sizeof(int16_t), dout, e->io_cap*sizeof(int16_t), samples*sizeof(int16_t),
                              e->n_ch, hipMemcpyDeviceToHost, e->stream));
    ECHO_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

// State export in the REFERENCE's layout: scal[48] as enumerated in echo_dev.hpp (same order
// as the fields of echo_can_state_t that matter), taps32[T], taps16[4][T], and the FIR
// history in its physical (circular) order.
int spangpu_echo_get_state(spangpu_echo_t *e, int channel, int32_t *scal, int32_t *taps32, int16_t *taps16, int16_t *history)
{
    if (e == nullptr  ||  channel < 0  ||  channel >= e->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    ECHO_TRY(hipSetDevice(e->device));
    ECHO_TRY(hipStreamSynchronize(e->stream));
    const int T = e->taps;
    int32_t s[kEchoScalars];
    ECHO_TRY(hipMemcpy(s, e->scal + (size_t) channel*kEchoScalars, sizeof(s), hipMemcpyDeviceToHost));
    if (scal)
        memcpy(scal, s, sizeof(s));
    if (taps32)
        ECHO_TRY(hipMemcpy(taps32, e->taps32 + (size_t) channel*T, T*sizeof(int32_t), hipMemcpyDeviceToHost));
    if (taps16)
        ECHO_TRY(hipMemcpy(taps16, e->taps16 + (size_t) channel*4*T, 4*T*sizeof(int16_t), hipMemcpyDeviceToHost));
    if (history)
    {
        int16_t w[1024];
        ECHO_TRY(hipMemcpy(w, e->hist + (size_t) channel*T, T*sizeof(int16_t), hipMemcpyDeviceToHost));
        // window order -> physical order: w[i] = history[(i + curr_pos + 1) mod T]
        for (int i = 0;  i < T;  i++)
            history[(i + s[ES_CURR_POS] + 1)%T] = w[i];
    }
    return SPANGPU_OK;
}

int spangpu_echo_set_state(spangpu_echo_t *e, int channel, const int32_t *scal, const int32_t *taps32, const int16_t *taps16, const int16_t *history)
{
    if (e == nullptr  ||  channel < 0  ||  channel >= e->n_ch  ||  scal == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    ECHO_TRY(hipSetDevice(e->device));
    ECHO_TRY(hipStreamSynchronize(e->stream));
    const int T = e->taps;
    // (the kernels derive every sample's position from this word with & (T - 1): echo_dev.hpp)
    if (scal[ES_CURR_POS] < 0  ||  scal[ES_CURR_POS] >= T)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "curr_pos outside 0 .. taps - 1");
    e->mode_dirty = true;
    if (scal[ES_ADAPTION_MODE] != e->uniform_mode  &&  e->n_ch > 1)
        e->uniform_mode = -1;
    else
        e->uniform_mode = scal[ES_ADAPTION_MODE];
    ECHO_TRY(hipMemcpy(e->scal + (size_t) channel*kEchoScalars, scal, kEchoScalars*sizeof(int32_t), hipMemcpyHostToDevice));
    if (taps32)
        ECHO_TRY(hipMemcpy(e->taps32 + (size_t) channel*T, taps32, T*sizeof(int32_t), hipMemcpyHostToDevice));
    if (taps16)
        ECHO_TRY(hipMemcpy(e->taps16 + (size_t) channel*4*T, taps16, 4*T*sizeof(int16_t), hipMemcpyHostToDevice));
    if (history)
    {
        int16_t w[1024];
        for (int i = 0;  i < T;  i++)
            w[i] = history[(i + scal[ES_CURR_POS] + 1)%T];
        ECHO_TRY(hipMemcpy(e->hist + (size_t) channel*T, w, T*sizeof(int16_t), hipMemcpyHostToDevice));
    }
    return SPANGPU_OK;
}

// ---- a channel's state in the reference's own struct layout (include/spangpu_refstate.h) ----------------------------
static void echo_words_from_ref(const spangpu_ref_echo_can_t *ec, int32_t *s)
{
    memset(s, 0, kEchoScalars*sizeof(int32_t));
    for (int i = 0;  i < 4;  i++)
        s[ES_TX_POWER0 + i] = ec->tx_power[i];
    for (int i = 0;  i < 3;  i++)
        s[ES_RX_POWER0 + i] = ec->rx_power[i];
    s[ES_CLEAN_RX_POWER] = ec->clean_rx_power;
    s[ES_RX_POWER_THRESHOLD] = ec->rx_power_threshold;
    s[ES_NONUPDATE_DWELL] = ec->nonupdate_dwell;
    s[ES_CURR_POS] = ec->curr_pos;
    s[ES_TAPS] = ec->taps;
    s[ES_TAP_MASK] = ec->tap_mask;
    s[ES_ADAPTION_MODE] = ec->adaption_mode;
    s[ES_SUPP_TEST1] = ec->supp_test1;
    s[ES_SUPP_TEST2] = ec->supp_test2;
    s[ES_SUPP1] = ec->supp1;
    s[ES_SUPP2] = ec->supp2;
    s[ES_VAD] = ec->vad;
    s[ES_CNG] = ec->cng;
    s[ES_GEIGEL_MAX] = ec->geigel_max;
    s[ES_GEIGEL_LAG] = ec->geigel_lag;
    s[ES_DTD_ONSET] = ec->dtd_onset;
    s[ES_TAP_SET] = ec->tap_set;
    s[ES_TAP_ROTATE_COUNTER] = ec->tap_rotate_counter;
    s[ES_LATEST_CORRECTION] = ec->latest_correction;
    s[ES_NARROWBAND_COUNT] = ec->narrowband_count;
    s[ES_NARROWBAND_SCORE] = ec->narrowband_score;
    s[ES_FIR_CURR_POS] = ec->fir_state.curr_pos;
    s[ES_TX_HPF0] = ec->tx_hpf[0];
    s[ES_TX_HPF1] = ec->tx_hpf[1];
    s[ES_RX_HPF0] = ec->rx_hpf[0];
    s[ES_RX_HPF1] = ec->rx_hpf[1];
    s[ES_CNG_LEVEL] = ec->cng_level;
    s[ES_CNG_RNDNUM] = ec->cng_rndnum;
    s[ES_CNG_FILTER] = ec->cng_filter;
    s[ES_FIR_SET] = ec->tap_set;
    for (int i = 0;  i < 4;  i++)
    {
        if (ec->fir_state.coeffs == ec->fir_taps16[i])
            s[ES_FIR_SET] = i;
    }
    for (int i = 0;  i < 9;  i++)
        s[ES_LAST_ACF + i] = ec->last_acf[i];
}

int spangpu_echo_import_state(spangpu_echo_t *e, int channel, const spangpu_ref_echo_can_t *ec)
{
    if (e == nullptr  ||  ec == nullptr  ||  channel < 0  ||  channel >= e->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (ec->taps != e->taps  ||  ec->fir_taps32 == nullptr  ||  ec->fir_state.history == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "the canceller is of another length than the bank, or has no arrays");
    const int T = e->taps;
    int32_t s[kEchoScalars];
    echo_words_from_ref(ec, s);
    int16_t *sets = (int16_t *) malloc((size_t) 4*T*sizeof(int16_t));
    if (sets == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "malloc");
    for (int i = 0;  i < 4;  i++)
    {
        if (ec->fir_taps16[i])
            memcpy(sets + (size_t) i*T, ec->fir_taps16[i], T*sizeof(int16_t));
        else
            memset(sets + (size_t) i*T, 0, T*sizeof(int16_t));
    }
    const int rc = spangpu_echo_set_state(e, channel, s, ec->fir_taps32, sets, ec->fir_state.history);
    free(sets);
    return rc;
}

int spangpu_echo_export_state(spangpu_echo_t *e, int channel, spangpu_ref_echo_can_t *ec)
{
    if (e == nullptr  ||  ec == nullptr  ||  channel < 0  ||  channel >= e->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (ec->taps != e->taps  ||  ec->fir_taps32 == nullptr  ||  ec->fir_state.history == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "the canceller is of another length than the bank, or has no arrays");
    const int T = e->taps;
    int32_t s[kEchoScalars];
    int16_t *sets = (int16_t *) malloc((size_t) 4*T*sizeof(int16_t));
    if (sets == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "malloc");
    const int rc = spangpu_echo_get_state(e, channel, s, ec->fir_taps32, sets, ec->fir_state.history);
    if (rc != SPANGPU_OK)
    {
        free(sets);
        return rc;
    }
    for (int i = 0;  i < 4;  i++)
    {
        if (ec->fir_taps16[i])
            memcpy(ec->fir_taps16[i], sets + (size_t) i*T, T*sizeof(int16_t));
    }
    free(sets);
    for (int i = 0;  i < 4;  i++)
        ec->tx_power[i] = s[ES_TX_POWER0 + i];
    for (int i = 0;  i < 3;  i++)
        ec->rx_power[i] = s[ES_RX_POWER0 + i];
    ec->clean_rx_power = s[ES_CLEAN_RX_POWER];
    ec->rx_power_threshold = s[ES_RX_POWER_THRESHOLD];
    ec->nonupdate_dwell = s[ES_NONUPDATE_DWELL];
    ec->curr_pos = s[ES_CURR_POS];
    ec->tap_mask = s[ES_TAP_MASK];
    ec->adaption_mode = s[ES_ADAPTION_MODE];
    ec->supp_test1 = s[ES_SUPP_TEST1];
    ec->supp_test2 = s[ES_SUPP_TEST2];
    ec->supp1 = s[ES_SUPP1];
    ec->supp2 = s[ES_SUPP2];
    ec->vad = s[ES_VAD];
    ec->cng = s[ES_CNG];
    ec->geigel_max = (int16_t) s[ES_GEIGEL_MAX];
    ec->geigel_lag = s[ES_GEIGEL_LAG];
    ec->dtd_onset = s[ES_DTD_ONSET];
    ec->tap_set = s[ES_TAP_SET];
    ec->tap_rotate_counter = s[ES_TAP_ROTATE_COUNTER];
    ec->latest_correction = s[ES_LATEST_CORRECTION];
    ec->narrowband_count = s[ES_NARROWBAND_COUNT];
    ec->narrowband_score = s[ES_NARROWBAND_SCORE];
    ec->fir_state.curr_pos = s[ES_FIR_CURR_POS];
    ec->fir_state.taps = T;
    if (s[ES_FIR_SET] >= 0  &&  s[ES_FIR_SET] < 4  &&  ec->fir_taps16[s[ES_FIR_SET]])
        ec->fir_state.coeffs = ec->fir_taps16[s[ES_FIR_SET]];
    ec->tx_hpf[0] = s[ES_TX_HPF0];
    ec->tx_hpf[1] = s[ES_TX_HPF1];
    ec->rx_hpf[0] = s[ES_RX_HPF0];
    ec->rx_hpf[1] = s[ES_RX_HPF1];
    ec->cng_level = s[ES_CNG_LEVEL];
    ec->cng_rndnum = s[ES_CNG_RNDNUM];
    ec->cng_filter = s[ES_CNG_FILTER];
    for (int i = 0;  i < 9;  i++)
        ec->last_acf[i] = s[ES_LAST_ACF + i];
    return SPANGPU_OK;
}

// echo_can_adaption_mode(), echo.c:324-328 (channel < 0: every channel)
// ---- per-channel line statistics ----------------------------------------------------------------------------------
int spangpu_echo_stats(spangpu_echo_t *e, int enable)
{
    if (e == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    ECHO_TRY(hipSetDevice(e->device));
    if (enable  &&  e->stats == nullptr)
    {
        ECHO_TRY(hipMalloc(&e->stats, (size_t) e->n_ch*sizeof(EchoStats)));
        ECHO_TRY(hipMemsetAsync(e->stats, 0, (size_t) e->n_ch*sizeof(EchoStats), e->stream));
    }
    e->stats_on = (enable == 2)  ?  2  :  (enable != 0)  ?  1  :  0;
    return SPANGPU_OK;
}

__global__ void echo_stats_reset_kernel(EchoStats *st, int n_ch, int what)
{
    const int ch = blockIdx.x*256 + threadIdx.x;
    if (ch >= n_ch)
        return;
    if (what & SPANGPU_ECHO_STATS_SUMS)
    {
        st[ch].sum_rx2 = 0;
        st[ch].sum_clean2 = 0;
        st[ch].samples = 0;
    }
    if (what & SPANGPU_ECHO_STATS_CRC)
        st[ch].crc = 0;
}

int spangpu_echo_stats_reset(spangpu_echo_t *e, int what)
{
    if (e == nullptr  ||  e->stats == nullptr)
        return spangpu_set_error(SPANGPU_ERR_STATE, "statistics are not enabled on this bank");
    ECHO_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(echo_stats_reset_kernel, dim3((e->n_ch + 255)/256), dim3(256), 0, e->stream, e->stats, e->n_ch, what);
    ECHO_TRY(hipGetLastError());
    return SPANGPU_OK;
}

int spangpu_echo_stats_get(spangpu_echo_t *e, int first, int n, spangpu_echo_stats_t *out)
{
    if (e == nullptr  ||  e->stats == nullptr)
        return spangpu_set_error(SPANGPU_ERR_STATE, "statistics are not enabled on this bank");
    if (out == nullptr  ||  first < 0  ||  n < 0  ||  first + n > e->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    static_assert(sizeof(spangpu_echo_stats_t) == sizeof(EchoStats), "ABI struct and device struct are the same record");
    ECHO_TRY(hipSetDevice(e->device));
    ECHO_TRY(hipMemcpyAsync(out, e->stats + first, (size_t) n*sizeof(EchoStats), hipMemcpyDeviceToHost, e->stream));
    ECHO_TRY(hipStreamSynchronize(e->stream));
    return SPANGPU_OK;
}

int spangpu_echo_erle(spangpu_echo_t *e, float *erle_db, int mem)
{
    if (e == nullptr  ||  e->stats == nullptr)
        return spangpu_set_error(SPANGPU_ERR_STATE, "statistics are not enabled on this bank");
    if (erle_db == nullptr  ||  (mem != SPANGPU_MEM_HOST  &&  mem != SPANGPU_MEM_DEVICE))
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    ECHO_TRY(hipSetDevice(e->device));
    float *dst = erle_db;
    if (mem == SPANGPU_MEM_HOST)
    {
        if (e->d_erle == nullptr)
            ECHO_TRY(hipMalloc(&e->d_erle, (size_t) e->n_ch*sizeof(float)));
        dst = e->d_erle;
    }
    hipLaunchKernelGGL(echo_erle_kernel, dim3((e->n_ch + 255)/256), dim3(256), 0, e->stream, (const EchoStats *) e->stats, dst, e->n_ch);
    ECHO_TRY(hipGetLastError());
    if (mem == SPANGPU_MEM_HOST)
    {
        ECHO_TRY(hipMemcpyAsync(erle_db, dst, (size_t) e->n_ch*sizeof(float), hipMemcpyDeviceToHost, e->stream));
        ECHO_TRY(hipStreamSynchronize(e->stream));
    }
    return SPANGPU_OK;
}

int spangpu_echo_adaption_mode(spangpu_echo_t *e, int channel, int adaption_mode)
{
    if (e == nullptr  ||  channel >= e->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    ECHO_TRY(hipSetDevice(e->device));
    ECHO_TRY(hipStreamSynchronize(e->stream));
    const int lo = (channel < 0)  ?  0  :  channel;
    const int hi = (channel < 0)  ?  e->n_ch  :  (channel + 1);
    if (hi - lo == e->n_ch)
        e->uniform_mode = adaption_mode;
    else if (adaption_mode != e->uniform_mode)
        e->uniform_mode = -1;
    e->mode_dirty = (hi - lo != e->n_ch);
    hipLaunchKernelGGL(echo_set_scalar_kernel, dim3((hi - lo + 255)/256), dim3(256), 0, e->stream,
                       e->scal, lo, hi, (int) ES_ADAPTION_MODE, adaption_mode);
    ECHO_TRY(hipGetLastError());
    ECHO_TRY(hipStreamSynchronize(e->stream));
    return SPANGPU_OK;
}

// echo_can_flush(), echo.c:331-372
int spangpu_echo_flush(spangpu_echo_t *e, int channel)
{
    if (e == nullptr  ||  channel < 0  ||  channel >= e->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    ECHO_TRY(hipSetDevice(e->device));
    ECHO_TRY(hipStreamSynchronize(e->stream));
    const int T = e->taps;
    int32_t s[kEchoScalars];
    ECHO_TRY(hipMemcpy(s, e->scal + (size_t) channel*kEchoScalars, sizeof(s), hipMemcpyDeviceToHost));
    s[ES_TX_POWER0] = s[ES_TX_POWER1] = s[ES_TX_POWER2] = s[ES_TX_POWER3] = 0;
    s[ES_RX_POWER0] = s[ES_RX_POWER1] = s[ES_RX_POWER2] = 0;
    s[ES_CLEAN_RX_POWER] = 0;
    s[ES_NONUPDATE_DWELL] = 0;
    s[ES_FIR_CURR_POS] = T - 1;
    s[ES_CURR_POS] = T - 1;
    s[ES_SUPP_TEST1] = s[ES_SUPP_TEST2] = s[ES_SUPP1] = s[ES_SUPP2] = 0;
    s[ES_VAD] = 0;
    s[ES_CNG_LEVEL] = 1000;
    s[ES_CNG_FILTER] = 0;
    s[ES_GEIGEL_MAX] = s[ES_GEIGEL_LAG] = 0;
    s[ES_DTD_ONSET] = 0;
    s[ES_TAP_SET] = 0;                      // fir_state.coeffs (ES_FIR_SET) is deliberately NOT reset
    s[ES_TAP_ROTATE_COUNTER] = 1600;
    s[ES_LATEST_CORRECTION] = 0;
    for (int i = 0;  i < 9;  i++)
        s[ES_LAST_ACF + i] = 0;
    s[ES_NARROWBAND_COUNT] = 0;
    s[ES_NARROWBAND_SCORE] = 0;
    ECHO_TRY(hipMemcpy(e->scal + (size_t) channel*kEchoScalars, s, sizeof(s), hipMemcpyHostToDevice));
    ECHO_TRY(hipMemset(e->taps32 + (size_t) channel*T, 0, T*sizeof(int32_t)));
    ECHO_TRY(hipMemset(e->taps16 + (size_t) channel*4*T, 0, 4*T*sizeof(int16_t)));
    ECHO_TRY(hipMemset(e->hist + (size_t) channel*T, 0, T*sizeof(int16_t)));
    return SPANGPU_OK;
}

}   // extern "C"
