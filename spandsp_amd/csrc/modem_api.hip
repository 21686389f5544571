// modem_api.hip -- C ABI of the batched modem receivers (include/spangpu.h, "Modem receiver banks").
// Device code: v29_dev.hpp, v27ter_dev.hpp, v17_dev.hpp; constant tables: modem_tables.c.  No CPU implementation
// exists behind these entry points.

#include <hip/hip_runtime.h>
#include <atomic>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spangpu.h"
#include "modem_tables.h"
#include "v29_dev.hpp"
#include "v29_quad.hpp"
#include "v27ter_dev.hpp"
#include "v17_dev.hpp"
#include "v17_quad.hpp"
#include "v27ter_quad.hpp"

// the four-lane kernels live in translation units of their own, each compiled with the instruction scheduler that suits it
// (Makefile; modem_v27q.hip says what was measured)
namespace spg {
void launch_v29_quad(const V29Launch &L, hipStream_t stream);
void launch_v17_quad(const V17Launch &L, hipStream_t stream);
void launch_v27ter_quad(const V27Launch &L, hipStream_t stream);
}

using namespace spg;

extern "C" int spangpu_set_error(int code, const char *msg);

static std::atomic<int> g_modem_mapping{0};     // spangpu_tune_modem_mapping(): a process-wide knob another thread may turn while a bank launches

#define V29_TRY(expr)                                                                       \
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

struct spangpu_modem_s
{
    const int32_t *next_lens;   // per-channel lengths of the call being prepared (device), or nullptr
    int32_t *d_lens;            // [n_ch], device
    int32_t *h_lens;            // [n_ch], pinned
    int kind;
    int n_words;
    int n_floats;
    int device;
    int n_ch;
    int bit_rate;
    hipStream_t stream;
    bool own_stream;
    uint32_t *state;            // [kV29Words][n_ch]
    void *tab;
    int16_t *d_amp;
    size_t amp_cap;
    int8_t *events;
    int32_t *ev_count;
    int ev_cap;
    int8_t *h_events;
    int32_t *h_count;
    int last_cap;
    bool qam_tap;               // spangpu_modem_qam_tap(): run the kernel variant that records the qam_report calls
    uint32_t *qam;              // [n_ch][qam_cap][7]
    int32_t *qam_count;
    int qam_cap;
    uint32_t *h_qam;
    int32_t *h_qam_count;
    int last_qam_cap;
    uint32_t *d_packed;         // spangpu_modem_events_packed(): [n_ch][packed_wpc] rows, then the status list
    uint32_t *h_packed;
    int packed_wpc;
    int packed_status_cap;
    int last_samples;           // frame length of the last spangpu_modem_rx()
};

// power_meter_level_dbm0(), power_meter.c:82-92
static int32_t level_dbm0(float level)
{
    level -= (3.14f + 3.02f);
    if (level > 0.0)
        level = 0.0;
    const float l = powf(10.0f, level/10.0f)*(32767.0f*32767.0f);
    return (int32_t) l;
}

// v29_rx_init() + v29_rx_restart(.., false), v29rx.c:1019-1131, as one channel's state words
static int v29_initial_words(uint32_t *w, int bit_rate, float cutoff_dbm0)
{
    float f[kV29Floats];
    int32_t i[kV29Ints];
    memset(f, 0, sizeof(f));
    memset(i, 0, sizeof(i));
    switch (bit_rate)
    {
    case 9600: i[VI_TRAINING_CD] = 0; break;
    case 7200: i[VI_TRAINING_CD] = 2; break;
    case 4800: i[VI_TRAINING_CD] = 4; break;
    default: return -1;
    }
    i[VI_BIT_RATE] = bit_rate;
    i[VI_TRAIN_SCRAMBLE] = 0x2A;
    i[VI_STAGE] = V29_SYMBOL_ACQUISITION;
    i[VI_PHASE_RATE] = (int32_t) (1700.0f*65536.0f*65536.0f/8000);
    i[VI_ON_POWER] = (int32_t) (level_dbm0(cutoff_dbm0 + 2.5f)*0.4f);       // v29rx.c:163-169
    i[VI_OFF_POWER] = (int32_t) (level_dbm0(cutoff_dbm0 - 2.5f)*0.4f);
    i[VI_EQ_PUT_STEP] = kRrcSets*10/(3*2) - 1;
    f[VF_EQ_COEFF + 2*16] = 3.0f;                                           // equalizer_reset()
    f[VF_EQ_DELTA] = 0.21f/kEqLen;
    f[VF_AGC] = (1.25f/1.0f)/735.0f;
    f[VF_TRACK_I] = 8000.0f;
    f[VF_TRACK_P] = 8000000.0f;
    memcpy(w, f, sizeof(f));
    memcpy(w + kV29Floats, i, sizeof(i));
    return 0;
}

// v27ter_rx_init() + v27ter_rx_restart(), v27ter_rx.c:1091-1190
static int v27_initial_words(uint32_t *w, int bit_rate, float cutoff_dbm0)
{
    float f[kV27Floats];
    int32_t i[kV27Ints];
    memset(f, 0, sizeof(f));
    memset(i, 0, sizeof(i));
    if (bit_rate != 4800  &&  bit_rate != 2400)
        return -1;
    i[WI_BIT_RATE] = bit_rate;
    i[WI_SCRAMBLE] = 0x3C;
    i[WI_STAGE] = V27_SYMBOL_ACQUISITION;
    i[WI_PHASE_RATE] = (int32_t) (1800.0f*65536.0f*65536.0f/8000);
    i[WI_ON_POWER] = (int32_t) (level_dbm0(cutoff_dbm0 + 2.5f)*0.4f);
    i[WI_OFF_POWER] = (int32_t) (level_dbm0(cutoff_dbm0 - 2.5f)*0.4f);
    i[WI_EQ_PUT_STEP] = (bit_rate == 4800)  ?  8*5/2  :  12*20/(3*2);
    i[WI_GARDNER_STEP] = 512;
    f[WF_EQ_COEFF + 2*17] = 1.414f;
    f[WF_EQ_DELTA] = 0.25f/kV27EqLen;
    f[WF_AGC] = (1.414f/1.000000f)/283.0f;
    f[WF_TRACK_I] = 200000.0f;
    f[WF_TRACK_P] = 10000000.0f;
    memcpy(w, f, sizeof(f));
    memcpy(w + kV27Floats, i, sizeof(i));
    return 0;
}

// v17_rx_init() + v17_rx_restart(.., false), v17rx.c:1399-1535
static int v17_initial_words(uint32_t *w, int bit_rate, float cutoff_dbm0)
{
    float f[kV17Floats];
    int32_t i[kV17Ints];
    memset(f, 0, sizeof(f));
    memset(i, 0, sizeof(i));
    switch (bit_rate)
    {
    case 14400: i[XI_SPACE_MAP] = 0; i[XI_BITS_PER_SYMBOL] = 6; break;
    case 12000: i[XI_SPACE_MAP] = 1; i[XI_BITS_PER_SYMBOL] = 5; break;
    case 9600: i[XI_SPACE_MAP] = 2; i[XI_BITS_PER_SYMBOL] = 4; break;
    case 7200: i[XI_SPACE_MAP] = 3; i[XI_BITS_PER_SYMBOL] = 3; break;
    case 4800: i[XI_SPACE_MAP] = 0; i[XI_BITS_PER_SYMBOL] = 2; break;
    default: return -1;
    }
    i[XI_BIT_RATE] = bit_rate;
    i[XI_DIFF] = 1;
    i[XI_SCRAMBLE] = 0x2ECDD5;
    i[XI_SCRAMBLER_TAP] = 18 - 1;
    i[XI_STAGE] = V17_SYMBOL_ACQUISITION;
    i[XI_TRELLIS_PTR] = 14;
    i[XI_PHASE_RATE] = (int32_t) (1800.0f*65536.0f*65536.0f/8000);
    i[XI_PHASE_RATE_SAVE] = i[XI_PHASE_RATE];
    i[XI_ON_POWER] = (int32_t) (level_dbm0(cutoff_dbm0 + 2.5f)*0.4f);
    i[XI_OFF_POWER] = (int32_t) (level_dbm0(cutoff_dbm0 - 2.5f)*0.4f);
    i[XI_EQ_PUT_STEP] = kV17Sets*10/(3*2) - 1;
    f[VF_EQ_COEFF + 2*16] = 3.0f;
    f[VF_EQ_DELTA] = 0.21f/kEqLen;
    f[VF_AGC] = (2.17f/1.000000f)/735.0f;
    f[VF_TRACK_I] = 5000.0f;
    f[VF_TRACK_P] = 40000.0f;
    for (int k = 1;  k < 8;  k++)
        f[XF_DIST + k] = 99.0f*1.0f;
    memcpy(w, f, sizeof(f));
    memcpy(w + kV17Floats, i, sizeof(i));
    return 0;
}

static int initial_words(int kind, uint32_t *w, int bit_rate)
{
    switch (kind)
    {
    case SPANGPU_V17:
        return v17_initial_words(w, bit_rate, -45.5f);
    case SPANGPU_V29:
        return v29_initial_words(w, bit_rate, -28.5f);
    case SPANGPU_V27TER:
        return v27_initial_words(w, bit_rate, -45.5f);
    }
    return -1;
}

static constexpr int kMaxWords = 1024;

extern "C" {

int spangpu_modem_state_words(int kind, int *n_floats, int *n_ints)
{
    int nf;
    int ni;
    switch (kind)
    {
    case SPANGPU_V29: nf = kV29Floats; ni = kV29Ints; break;
    case SPANGPU_V27TER: nf = kV27Floats; ni = kV27Ints; break;
    case SPANGPU_V17: nf = kV17Floats; ni = kV17Ints; break;
    default: return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "not a modem kind");
    }
    if (n_floats) *n_floats = nf;
    if (n_ints) *n_ints = ni;
    return nf + ni;
}

#if defined(SPG_QUAD_PROF)
// builder's instrument (tools/quad_prof.py): cycles per phase of the quad kernels, summed over waves; reading clears
extern "C" __attribute__((visibility("default"))) int spangpu_debug_quad_prof(unsigned long long *out)
{
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(spg::spg_quad_prof), sizeof(zero)) != hipSuccess)
        return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(spg::spg_quad_prof), zero, sizeof(zero)) != hipSuccess)
        return -1;
    return 0;
}
#endif

// Tuning / A-B testing: how the receiver kernels map channels to lanes from now on (0 = by bank size: four lanes per channel
// below 65 536 channels, one above; 1 = one channel per lane at every size; 4 = four lanes per channel, 16 channels per
// wave, at every size, all three receivers; 8 is accepted and means 4).  Results are identical.
int spangpu_tune_modem_mapping(int mapping)
{
    if (mapping != 0  &&  mapping != 1  &&  mapping != 4  &&  mapping != 8)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "modem mapping must be 0 (auto), 1, 4 or 8");
    g_modem_mapping.store(mapping, std::memory_order_relaxed);
    return SPANGPU_OK;
}

int spangpu_modem_create(spangpu_modem_t **out, int device, int kind, int n_channels, int bit_rate)
{
    if (out == nullptr  ||  n_channels <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    *out = nullptr;
    uint32_t w[kMaxWords];
    const int n_words = spangpu_modem_state_words(kind, nullptr, nullptr);
    if (n_words < 0)
        return n_words;
    if (initial_words(kind, w, bit_rate) < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bit rate not valid for this modem (V.29: 9600/7200/4800, V.27ter: 4800/2400, V.17: 14400/12000/9600/7200/4800)");
    if (spangpu_device_count() <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    if (device < 0  ||  device >= spangpu_device_count())
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "device out of range");
    V29_TRY(hipSetDevice(device));
    spangpu_modem_t *m = (spangpu_modem_t *) calloc(1, sizeof(*m));
    if (m == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "calloc");
    m->kind = kind;
    m->n_words = n_words;
    spangpu_modem_state_words(kind, &m->n_floats, nullptr);
    m->device = device;
    m->n_ch = n_channels;
    m->bit_rate = bit_rate;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess)
    {
        free(m);
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipStreamCreate failed");
    }
    m->own_stream = true;
    const size_t n = (size_t) n_channels;
    const size_t tab_bytes = (kind == SPANGPU_V29)  ?  sizeof(V29Tables)  :  (kind == SPANGPU_V17)  ?  sizeof(V17Tables)  :  sizeof(V27Tables);
    void *ht = calloc(1, tab_bytes);
    uint32_t *hs = (uint32_t *) malloc(n*n_words*sizeof(uint32_t));
    if (ht == nullptr  ||  hs == nullptr
        ||  hipMalloc(&m->state, n*n_words*sizeof(uint32_t)) != hipSuccess
        ||  hipMalloc(&m->tab, tab_bytes) != hipSuccess
        ||  hipMalloc(&m->ev_count, n*sizeof(int32_t)) != hipSuccess
        ||  hipHostMalloc(&m->h_count, n*sizeof(int32_t)) != hipSuccess)
    {
        free(ht);
        free(hs);
        spangpu_modem_destroy(m);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "allocation of V.29 bank failed");
    }
    // constant tables (modem_tables.c): V.29 rx pulse shaper = 48 x 27, 1700 Hz, 2400 baud, 50 % excess
    // bandwidth (make_modem_filter.c:403-411); Godard = 1700 Hz, 2400 baud, alpha 0.99, triggers 1000 / 30,
    // steps 5 / 1 (src/Makefile.am:559-560)
    if (kind == SPANGPU_V29)
    {
        V29Tables *t = (V29Tables *) ht;
        spg_make_rx_pulseshaper(kRrcSets, kRrcLen, 1700.0, 2400.0, 0.5, t->rrc_re, t->rrc_im);
        spg_make_sine_table(t->sine);
        spg_make_sqrt_table(t->sqrt_tab);
        spg_make_godard(1700.0, 2400.0, 0.99, t->godard);
        t->coarse_trigger = 1000.0f;
        t->fine_trigger = 30.0f;
        t->coarse_step = 5;
        t->fine_step = 1;
        spg_make_v29_space_map(t->space_map);
    }
    else if (kind == SPANGPU_V17)
    {
        // V.17 rx pulse shaper: 192 phases x 27 taps, 1800 Hz, 2400 baud, 50 % excess bandwidth (make_modem_filter.c:327-339),
        // Godard: 1800 Hz, 2400 baud, alpha 0.99, triggers 1000 / 100, steps 15 / 1 (src/Makefile.am:490-491)
        V17Tables *t = (V17Tables *) ht;
        float *re = (float *) malloc(2*kV17Sets*kRrcLen*sizeof(float));
        uint8_t *maps = (uint8_t *) malloc(4*36*36*8 + 36*36);
        int8_t pts[128][2];
        if (re == nullptr  ||  maps == nullptr)
        {
            free(re);
            free(maps);
            free(ht);
            free(hs);
            spangpu_modem_destroy(m);
            return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "table scratch");
        }
        float *im = re + kV17Sets*kRrcLen;
        spg_make_rx_pulseshaper(kV17Sets, kRrcLen, 1800.0, 2400.0, 0.5, re, im);
        for (int set = 0;  set < kV17Sets;  set++)
        {
            for (int tap = 0;  tap < kRrcLen;  tap++)
            {
                t->rrc_re[tap*kV17Sets + set] = re[set*kRrcLen + tap];
                t->rrc_im[tap*kV17Sets + set] = im[set*kRrcLen + tap];
                t->rrc_q[2*(tap*kV17Sets + set)] = re[set*kRrcLen + tap];
                t->rrc_q[2*(tap*kV17Sets + set) + 1] = im[set*kRrcLen + tap];
            }
        }
        spg_make_sine_table(t->sine);
        spg_make_sqrt_table(t->sqrt_tab);
        spg_make_godard(1800.0, 2400.0, 0.99, t->godard);
        t->coarse_trigger = 1000.0f;
        t->fine_trigger = 100.0f;
        t->coarse_step = 15;
        t->fine_step = 1;
        const int np = spg_make_v17_constellation(bit_rate, pts);
        for (int k = 0;  k < np;  k++)
        {
            t->con[2*k] = (float) pts[k][0];
            t->con[2*k + 1] = (float) pts[k][1];
        }
        spg_make_v17_rx_maps(maps, maps + 4*36*36*8);
        if (bit_rate == 4800)
            memcpy(t->map, maps + 4*36*36*8, 36*36);
        else
            memcpy(t->map, maps + (size_t) w[kV17Floats + XI_SPACE_MAP]*36*36*8, 36*36*8);
        free(re);
        free(maps);
    }
    else
    {
        // V.27ter rx pulse shapers: 1800 Hz carrier, 50 % excess bandwidth; 8 sets at 1600 baud, 12 sets at 1200 baud
        // (make_modem_filter.c:379-402)
        V27Tables *t = (V27Tables *) ht;
        spg_make_rx_pulseshaper(8, kRrcLen, 1800.0, 1600.0, 0.5, t->re4800, t->im4800);
        spg_make_rx_pulseshaper(12, kRrcLen, 1800.0, 1200.0, 0.5, t->re2400, t->im2400);
        spg_make_sine_table(t->sine);
        spg_make_sqrt_table(t->sqrt_tab);
    }
    for (int k = 0;  k < n_words;  k++)
    {
        for (size_t c = 0;  c < n;  c++)
            hs[(size_t) k*n + c] = w[k];
    }
    hipError_t rc1 = hipMemcpy(m->tab, ht, tab_bytes, hipMemcpyHostToDevice);
    hipError_t rc2 = hipMemcpy(m->state, hs, n*n_words*sizeof(uint32_t), hipMemcpyHostToDevice);
    free(ht);
    free(hs);
    if (rc1 != hipSuccess  ||  rc2 != hipSuccess)
    {
        spangpu_modem_destroy(m);
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    }
    *out = m;
    return SPANGPU_OK;
}

int spangpu_modem_destroy(spangpu_modem_t *m)
{
    if (m == nullptr)
        return SPANGPU_OK;
    (void) hipSetDevice(m->device);
    if (m->stream)
        (void) hipStreamSynchronize(m->stream);
    if (m->state) (void) hipFree(m->state);
    if (m->tab) (void) hipFree(m->tab);
    if (m->d_amp) (void) hipFree(m->d_amp);
    if (m->d_lens) (void) hipFree(m->d_lens);
    if (m->h_lens) (void) hipHostFree(m->h_lens);
    if (m->events) (void) hipFree(m->events);
    if (m->ev_count) (void) hipFree(m->ev_count);
    if (m->h_events) (void) hipHostFree(m->h_events);
    if (m->h_count) (void) hipHostFree(m->h_count);
    if (m->qam) (void) hipFree(m->qam);
    if (m->qam_count) (void) hipFree(m->qam_count);
    if (m->h_qam) (void) hipHostFree(m->h_qam);
    if (m->h_qam_count) (void) hipHostFree(m->h_qam_count);
    if (m->d_packed) (void) hipFree(m->d_packed);
    if (m->h_packed) (void) hipHostFree(m->h_packed);
    if (m->own_stream  &&  m->stream)
        (void) hipStreamDestroy(m->stream);
    free(m);
    return SPANGPU_OK;
}

int spangpu_modem_channels(const spangpu_modem_t *m) { return m  ?  m->n_ch  :  SPANGPU_ERR_BAD_ARG; }

int spangpu_modem_set_stream(spangpu_modem_t *m, void *hip_stream)
{
    if (m == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    (void) hipStreamSynchronize(m->stream);
    if (m->own_stream)
        (void) hipStreamDestroy(m->stream);
    if (hip_stream)
    {
        m->stream = (hipStream_t) hip_stream;
        m->own_stream = false;
    }
    else
    {
        V29_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
        m->own_stream = true;
    }
    return SPANGPU_OK;
}

int spangpu_modem_sync(spangpu_modem_t *m)
{
    if (m == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null bank");
    V29_TRY(hipStreamSynchronize(m->stream));
    return SPANGPU_OK;
}

int spangpu_modem_rx(spangpu_modem_t *m, const int16_t *amp, int mem, int samples, long long stride)
{
    if (m == nullptr  ||  amp == nullptr  ||  samples < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (samples == 0)
        return 0;
    if (stride <= 0)
        stride = samples;
    V29_TRY(hipSetDevice(m->device));
    // at most 4 (V.17: 6) bits per baud and nominally a baud every 8000/2400 samples; symbol timing recovery can run the
    // baud clock fast by up to 5/160 of a baud per baud, so 1/16 more bauds than nominal are provided for, plus room
    // for every status report a call can make.  spangpu_modem_events() refuses to hand out a stream that did not fit.
    const int bauds = (samples*3 + 9)/10;
    const int cap = ((bauds + bauds/16 + 2)*((m->kind == SPANGPU_V17)  ?  6  :  4) + 16 + 15) & ~15;
    if (cap > m->ev_cap)
    {
        if (m->events) (void) hipFree(m->events);
        if (m->h_events) (void) hipHostFree(m->h_events);
        m->events = nullptr;
        m->h_events = nullptr;
        m->ev_cap = 0;
        V29_TRY(hipMalloc(&m->events, (size_t) m->n_ch*cap));
        V29_TRY(hipHostMalloc(&m->h_events, (size_t) m->n_ch*cap));
        m->ev_cap = cap;
    }
    if (m->qam_tap)
    {
        // one report per baud (at most 2400 per second) and, V.27ter, one per timing hop (at most one per baud)
        const int qcap = 2*((samples*3 + 9)/10 + 2);
        if (qcap > m->qam_cap)
        {
            if (m->qam) (void) hipFree(m->qam);
            if (m->h_qam) (void) hipHostFree(m->h_qam);
            m->qam = nullptr;
            m->h_qam = nullptr;
            m->qam_cap = 0;
            V29_TRY(hipMalloc(&m->qam, (size_t) m->n_ch*qcap*7*sizeof(uint32_t)));
            V29_TRY(hipHostMalloc(&m->h_qam, (size_t) m->n_ch*qcap*7*sizeof(uint32_t)));
            m->qam_cap = qcap;
        }
        if (m->qam_count == nullptr)
        {
            V29_TRY(hipMalloc(&m->qam_count, (size_t) m->n_ch*sizeof(int32_t)));
            V29_TRY(hipHostMalloc(&m->h_qam_count, (size_t) m->n_ch*sizeof(int32_t)));
        }
    }
    const int16_t *d_amp = amp;
    long long d_stride = stride;
    if (mem == SPANGPU_MEM_HOST)
    {
        if ((size_t) samples > m->amp_cap)
        {
            if (m->d_amp) (void) hipFree(m->d_amp);
            m->d_amp = nullptr;
            m->amp_cap = 0;
            V29_TRY(hipMalloc(&m->d_amp, (size_t) m->n_ch*samples*sizeof(int16_t)));
            m->amp_cap = samples;
        }
        V29_TRY(hipMemcpy2DAsync(m->d_amp, m->amp_cap*sizeof(int16_t), amp, stride*sizeof(int16_t),
                                 samples*sizeof(int16_t), m->n_ch, hipMemcpyHostToDevice, m->stream));
        V29_TRY(hipStreamSynchronize(m->stream));         // amp[] is only borrowed for the duration of the call
        d_amp = m->d_amp;
        d_stride = (long long) m->amp_cap;
    }
    else if (mem != SPANGPU_MEM_DEVICE)
    {
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad mem kind");
    }
    // enough workgroups to put a wave on every SIMD (256 CUs x 4) before filling the waves
    const int cpw = (m->n_ch >= 64*1024)  ?  64  :  (m->n_ch >= 32*1024)  ?  32  :  16;
    // lanes per channel: 1 = the one-channel-per-lane kernels, 4 = a quad per channel with 16 channels per wave (8 is taken as
    // 4: the variant with 8 channels per wave, two waves per SIMD, was measured -- slower -- and removed);
    // spangpu_tune_modem_mapping() overrides
    // (measured, 16 384-channel rounds of the quad kernels against the one-lane kernels, V.29 / V.17 / V.27ter: 32 768 channels
    // 0.32 / 0.40 / 0.24 ms against 0.75 / 0.86 / 0.26; 49 152 channels 0.47 / 0.59 / 0.34 against 0.75 / 0.84 / 0.48; from
    // 65 536 channels the full-wave one-lane kernels win: 0.45 / 0.92 / 0.29 ms against four rounds of 0.156 / 0.215 / 0.111)
    const int mapping = g_modem_mapping.load(std::memory_order_relaxed);        // read once per launch
    const int quad = (mapping != 0)  ?  mapping  :  (m->n_ch < 64*1024)  ?  4  :  1;
    const bool forced_quad = (mapping == 4  ||  mapping == 8);       // an explicit four lanes per channel holds at every bank size (A-B runs)
    const dim3 grid((m->n_ch + cpw - 1)/cpw);
    if (m->kind == SPANGPU_V29)
    {
        V29Launch L;
        memset(&L, 0, sizeof(L));
        L.amp = d_amp;
        L.stride = d_stride;
        L.samples = samples;
        L.lens = m->next_lens;
        L.n_ch = m->n_ch;
        L.state = m->state;
        L.events = m->events;
        L.ev_count = m->ev_count;
        L.ev_cap = m->ev_cap;
        L.tab = (const V29Tables *) m->tab;
        L.qam = m->qam;
        L.qam_count = m->qam_count;
        L.qam_cap = m->qam_cap;
        if (m->qam_tap)
            hipLaunchKernelGGL((v29_bank_kernel<16, true>), dim3((m->n_ch + 15)/16), dim3(64), 0, m->stream, L);
        else if (cpw == 64  &&  !forced_quad)
        {
            // full waves: four to a workgroup, sharing the tables, the RRC delay line as packed int16 pairs -- 150 KB of
            // LDS per workgroup, one workgroup per CU, a wave on every SIMD (v29_dev.hpp)
            const int waves = (m->n_ch + 63)/64;
            hipLaunchKernelGGL((v29_bank_kernel<64, false, 4, 16, true>), dim3((waves + 3)/4), dim3(256), 0, m->stream, L);
        }
        else if (quad == 4  ||  quad == 8)
        {
            // banks that cannot fill the chip's 1 024 SIMDs with full waves of one channel per lane: four lanes per
            // channel, 16 channels per wave, four waves per workgroup sharing the tables (v29_quad.hpp)
            launch_v29_quad(L, m->stream);                          // modem_v29q.hip (a scheduler of its own)
        }
        else if (cpw == 32)
            hipLaunchKernelGGL(v29_bank_kernel<32>, grid, dim3(64), 0, m->stream, L);
        else
            hipLaunchKernelGGL(v29_bank_kernel<16>, grid, dim3(64), 0, m->stream, L);
    }
    else if (m->kind == SPANGPU_V17)
    {
        V17Launch L;
        memset(&L, 0, sizeof(L));
        L.amp = d_amp;
        L.stride = d_stride;
        L.samples = samples;
        L.lens = m->next_lens;
        L.n_ch = m->n_ch;
        L.bit_rate = m->bit_rate;
        L.state = m->state;
        L.events = m->events;
        L.ev_count = m->ev_count;
        L.ev_cap = m->ev_cap;
        L.tab = (const V17Tables *) m->tab;
        L.qam = m->qam;
        L.qam_count = m->qam_count;
        L.qam_cap = m->qam_cap;
        if (m->qam_tap)
            hipLaunchKernelGGL((v17_bank_kernel<16, true>), dim3((m->n_ch + 15)/16), dim3(64), 0, m->stream, L);
        else if (cpw == 64  &&  !forced_quad)
        {
            const int waves = (m->n_ch + 63)/64;
            hipLaunchKernelGGL((v17_bank_kernel<64, false, 3, 16, true>), dim3((waves + 2)/3), dim3(192), 0, m->stream, L);
        }
        else if (quad == 4  ||  quad == 8)
            launch_v17_quad(L, m->stream);                          // modem_v17q.hip (a scheduler of its own)
        else if (cpw == 32)
            hipLaunchKernelGGL(v17_bank_kernel<32>, grid, dim3(64), 0, m->stream, L);
        else
            hipLaunchKernelGGL(v17_bank_kernel<16>, grid, dim3(64), 0, m->stream, L);
    }
    else
    {
        V27Launch L;
        memset(&L, 0, sizeof(L));
        L.amp = d_amp;
        L.stride = d_stride;
        L.samples = samples;
        L.lens = m->next_lens;
        L.n_ch = m->n_ch;
        L.bit_rate = m->bit_rate;
        L.state = m->state;
        L.events = m->events;
        L.ev_count = m->ev_count;
        L.ev_cap = m->ev_cap;
        L.tab = (const V27Tables *) m->tab;
        L.qam = m->qam;
        L.qam_count = m->qam_count;
        L.qam_cap = m->qam_cap;
        if (m->qam_tap)
            hipLaunchKernelGGL((v27ter_bank_kernel<16, true>), dim3((m->n_ch + 15)/16), dim3(64), 0, m->stream, L);
        else if (cpw == 64  &&  !forced_quad)
        {
            const int waves = (m->n_ch + 63)/64;
            hipLaunchKernelGGL((v27ter_bank_kernel<64, false, 4, 16, true>), dim3((waves + 3)/4), dim3(256), 0, m->stream, L);
        }
        else if (quad == 4  ||  quad == 8)
            launch_v27ter_quad(L, m->stream);                       // modem_v27q.hip (a scheduler of its own)
        else if (cpw == 32)
            hipLaunchKernelGGL(v27ter_bank_kernel<32>, grid, dim3(64), 0, m->stream, L);
        else
            hipLaunchKernelGGL(v27ter_bank_kernel<16>, grid, dim3(64), 0, m->stream, L);
    }
    V29_TRY(hipGetLastError());
    m->last_cap = m->ev_cap;
    m->last_qam_cap = m->qam_tap  ?  m->qam_cap  :  0;
    if (mem == SPANGPU_MEM_HOST)
        V29_TRY(hipStreamSynchronize(m->stream));
    return 0;
}

// spangpu_modem_rx() for a tick in which not every receiver has a frame, or frames differ in length: channel c takes
// lens[c] samples of its row (0: it sits the call out, its state as it was, no events).  lens[] is host memory.
int spangpu_modem_rx_var(spangpu_modem_t *m, const int16_t *amp, int mem, const int32_t *lens, int max_samples, long long stride)
{
    if (m == nullptr  ||  amp == nullptr  ||  lens == nullptr  ||  max_samples < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    int longest = 0;
    bool all = true;
    for (int c = 0;  c < m->n_ch;  c++)
    {
        if (lens[c] < 0  ||  lens[c] > max_samples)
            return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "a channel's length is outside 0..max_samples");
        if (lens[c] > longest)
            longest = lens[c];
    }
    if (longest == 0)
        return 0;
    for (int c = 0;  c < m->n_ch;  c++)
        all &= (lens[c] == longest);
    if (stride <= 0)
        stride = max_samples;
    if (all)
        return spangpu_modem_rx(m, amp, mem, longest, stride);
    V29_TRY(hipSetDevice(m->device));
    if (m->d_lens == nullptr)
    {
        V29_TRY(hipMalloc(&m->d_lens, (size_t) m->n_ch*sizeof(int32_t)));
        V29_TRY(hipHostMalloc(&m->h_lens, (size_t) m->n_ch*sizeof(int32_t)));
    }
    V29_TRY(hipStreamSynchronize(m->stream));
    memcpy(m->h_lens, lens, (size_t) m->n_ch*sizeof(int32_t));
    V29_TRY(hipMemcpyAsync(m->d_lens, m->h_lens, (size_t) m->n_ch*sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
    m->next_lens = m->d_lens;
    const int rc = spangpu_modem_rx(m, amp, mem, longest, stride);
    m->next_lens = nullptr;
    return rc;
}

// The tap behind xxx_rx_set_qam_report_handler(): from the next spangpu_modem_rx() on, every channel's
// qam_report(user, constel, target, symbol) calls are recorded beside its put_bit stream.
int spangpu_modem_qam_tap(spangpu_modem_t *m, int enable)
{
    if (m == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null modem bank");
    m->qam_tap = (enable != 0);
    return 0;
}

// The qam_report calls of the last spangpu_modem_rx(): for channel c, counts[c] records of seven words at
// records + c*cap*7 -- {put_bit / status calls that came before it in this rx call, 1 if constel and target were NULL
// (V.27ter's timing hop report, v27ter_rx.c:517), symbol, constel re, im, target re, im (binary32 bits)}.  Returns cap.
int spangpu_modem_qam_reports(spangpu_modem_t *m, const uint32_t **records, const int32_t **counts)
{
    if (m == nullptr  ||  records == nullptr  ||  counts == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (m->last_qam_cap <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no spangpu_modem_rx() with the tap on yet");
    V29_TRY(hipSetDevice(m->device));
    V29_TRY(hipMemcpyAsync(m->h_qam, m->qam, (size_t) m->n_ch*m->last_qam_cap*7*sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream));
    V29_TRY(hipMemcpyAsync(m->h_qam_count, m->qam_count, (size_t) m->n_ch*sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    V29_TRY(hipStreamSynchronize(m->stream));
    *records = m->h_qam;
    *counts = m->h_qam_count;
    return m->last_qam_cap;
}

// The put_bit stream of the last spangpu_modem_rx() call: for channel c, counts[c] entries at
// events + c*cap, each 0/1 (a descrambled data bit) or a negative SIG_STATUS_* code
// (spandsp/async.h:66-103), in the order v29_rx() would have called put_bit().  Returns cap.
int spangpu_modem_events(spangpu_modem_t *m, const int8_t **events, const int32_t **counts)
{
    if (m == nullptr  ||  events == nullptr  ||  counts == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (m->last_cap <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no spangpu_modem_rx() yet");
    V29_TRY(hipSetDevice(m->device));
    V29_TRY(hipMemcpyAsync(m->h_events, m->events, (size_t) m->n_ch*m->last_cap, hipMemcpyDeviceToHost, m->stream));
    V29_TRY(hipMemcpyAsync(m->h_count, m->ev_count, (size_t) m->n_ch*sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    V29_TRY(hipStreamSynchronize(m->stream));
    for (int c = 0;  c < m->n_ch;  c++)
    {
        if (m->h_count[c] > m->last_cap)
            return spangpu_set_error(SPANGPU_ERR_STATE, "modem event buffer overflow: a channel produced more events than the call's frame length allows");
    }
    *events = m->h_events;
    *counts = m->h_count;
    return m->last_cap;
}

// The last call's events, device to device, for a gather across GPUs (the bit stream words of a frame):
// dst = int32 counts[n_ch], then int8 events[n_ch][per_channel].  Asynchronous on the bank's stream; a channel that made
// more than per_channel events shows it by its count (the caller sizes per_channel for its frame length: a V.29 9600
// receiver makes at most 4 per baud, 192 + status reports per 160 samples).
int spangpu_modem_copy_events(spangpu_modem_t *m, void *dev_dst, size_t dst_bytes, int per_channel)
{
    if (m == nullptr  ||  dev_dst == nullptr  ||  per_channel <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (m->last_cap <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no spangpu_modem_rx() yet");
    const size_t need = (size_t) m->n_ch*(sizeof(int32_t) + (size_t) per_channel);
    if (dst_bytes < need)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "destination too small");
    V29_TRY(hipSetDevice(m->device));
    V29_TRY(hipMemcpyAsync(dev_dst, m->ev_count, (size_t) m->n_ch*sizeof(int32_t), hipMemcpyDeviceToDevice, m->stream));
    const int w = (per_channel < m->last_cap)  ?  per_channel  :  m->last_cap;
    V29_TRY(hipMemcpy2DAsync((char *) dev_dst + (size_t) m->n_ch*sizeof(int32_t), (size_t) per_channel, m->events, (size_t) m->last_cap,
                             (size_t) w, (size_t) m->n_ch, hipMemcpyDeviceToDevice, m->stream));
    return SPANGPU_OK;
}

// ---- the put_bit stream in packed form (24 bytes per channel and frame for V.29 9600) -----------------------
// spangpu_modem_events() brings one byte per put_bit() call to the host (n_ch x cap bytes a tick).  What a caller needs to
// replay the calls is less: the data bits, eight to a byte, and -- rarely -- a status report with its place in the bit
// stream.  Row c of `packed` (words_per_channel words): word 0 = data bits | status reports << 16 of the call, then the
// bits LSB first.  The status reports of the whole bank go to one list: list[0] = how many, then pairs
// {channel, data bits that came before it | (code & 0xFFFF) << 16}; the reports of one channel stand in call order.
__global__ void modem_pack_kernel(const int8_t *events, const int32_t *counts, int n_ch, int cap, uint32_t *packed, int wpc,
                                  uint32_t *status, int status_cap)
{
    const int c = blockIdx.x*blockDim.x + threadIdx.x;
    if (c >= n_ch)
        return;
    const int8_t *ev = events + (size_t) c*cap;
    const int n = min(counts[c], cap);
    uint32_t *row = packed + (size_t) c*wpc;
    const int room = (wpc - 1)*32;
    uint32_t acc = 0;
    int nbits = 0;
    int nstat = 0;
    // sixteen events a load where the rows allow it (cap a multiple of 16: what spangpu_modem_rx() sizes them to)
    const bool wide = ((cap & 15) == 0)  &&  (((uintptr_t) events & 15) == 0);
    uint4 chunk = make_uint4(0, 0, 0, 0);
    for (int i = 0;  i < n;  i++)
    {
        int v;
        if (wide)
        {
            if ((i & 15) == 0)
                chunk = *(const uint4 *) (ev + i);
            const int k = i & 15;
            const uint32_t w = (k < 4)  ?  chunk.x  :  (k < 8)  ?  chunk.y  :  (k < 12)  ?  chunk.z  :  chunk.w;
            v = (int) (int8_t) (w >> (8*(k & 3)));
        }
        else
        {
            v = ev[i];
        }
        if (v >= 0)
        {
            if (nbits < room)
            {
                acc |= (uint32_t) (v & 1) << (nbits & 31);
                if ((nbits & 31) == 31)
                {
                    row[1 + (nbits >> 5)] = acc;
                    acc = 0;
                }
            }
            nbits++;
        }
        else
        {
            const uint32_t at = atomicAdd(&status[0], 1u);
            if (at < (uint32_t) status_cap)
            {
                status[1 + 2*at] = (uint32_t) c;
                status[2 + 2*at] = ((uint32_t) nbits & 0xFFFFu) | (((uint32_t) v & 0xFFFFu) << 16);
            }
            nstat++;
        }
    }
    if ((nbits & 31) != 0  &&  nbits < room)
        row[1 + (nbits >> 5)] = acc;
    // (a count of data bits above what the row holds says the row was too short; counts[c] > cap shows in bit 31)
    row[0] = ((uint32_t) nbits & 0x7FFFu) | (((uint32_t) nstat & 0x7FFFu) << 16) | ((counts[c] > cap)  ?  0x80000000u  :  0u);
}

// Words per channel that hold every data bit a call of `samples` samples can deliver at `bit_rate` (one header word + the
// bits; the receivers deliver a whole number of bauds, so a call can run a baud ahead of its share).
int spangpu_modem_packed_words(int bit_rate, int samples)
{
    if (bit_rate <= 0  ||  samples < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const long long bits = ((long long) samples*bit_rate + 7999)/8000 + 16;
    return 1 + (int) ((bits + 31)/32);
}

// The last spangpu_modem_rx()'s put_bit stream, packed (see modem_pack_kernel above), device to device on the bank's stream:
// packed_device [n_ch][words_per_channel], status_device [1 + 2*status_cap] (its first word is cleared here).
int spangpu_modem_pack_events(spangpu_modem_t *m, uint32_t *packed_device, int words_per_channel, uint32_t *status_device, int status_cap)
{
    if (m == nullptr  ||  packed_device == nullptr  ||  status_device == nullptr  ||  words_per_channel < 2  ||  status_cap < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (m->last_cap <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no spangpu_modem_rx() yet");
    V29_TRY(hipSetDevice(m->device));
    V29_TRY(hipMemsetAsync(status_device, 0, sizeof(uint32_t), m->stream));
    hipLaunchKernelGGL(modem_pack_kernel, dim3((m->n_ch + 255)/256), dim3(256), 0, m->stream, (const int8_t *) m->events,
                       (const int32_t *) m->ev_count, m->n_ch, m->last_cap, packed_device, words_per_channel, status_device, status_cap);
    V29_TRY(hipGetLastError());
    return SPANGPU_OK;
}

// Host side: the packed form back into the form spangpu_modem_events() delivers (events [n_ch][cap] int8, counts[n_ch]),
// i.e. the put_bit calls of every channel in order -- what a shim replays callbacks from.  n_status = status[0] as
// delivered (entries beyond status_cap were not recorded: returns SPANGPU_ERR_STATE then, as for a row that was too short).
int spangpu_modem_unpack_events(const uint32_t *packed, int words_per_channel, const uint32_t *status, int status_cap, int n_ch,
                                int8_t *events, int cap, int32_t *counts)
{
    if (packed == nullptr  ||  status == nullptr  ||  events == nullptr  ||  counts == nullptr  ||  words_per_channel < 2  ||  cap <= 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const uint32_t n_status = status[0];
    if (n_status > (uint32_t) status_cap)
        return spangpu_set_error(SPANGPU_ERR_STATE, "more status reports than the list holds");
    // a channel's reports stand in order in the list: walk it once, keeping a cursor per channel in counts[] (reused below)
    int32_t *next = (int32_t *) calloc((size_t) n_ch + 1, sizeof(int32_t));
    int32_t *order = (int32_t *) malloc((size_t) (n_status + 1)*sizeof(int32_t));
    if (next == nullptr  ||  order == nullptr)
    {
        free(next);
        free(order);
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "out of memory");
    }
    for (uint32_t k = 0;  k < n_status;  k++)
    {
        const uint32_t c = status[1 + 2*k];
        if (c < (uint32_t) n_ch)
            next[c + 1]++;
    }
    for (int c = 0;  c < n_ch;  c++)
        next[c + 1] += next[c];                         // next[c] = first slot of channel c in `order`
    for (uint32_t k = 0;  k < n_status;  k++)
    {
        const uint32_t c = status[1 + 2*k];
        if (c < (uint32_t) n_ch)
            order[next[c]++] = (int32_t) k;             // (stable: list order is call order within a channel)
    }
    int rc = SPANGPU_OK;
    int32_t first = 0;
    for (int c = 0;  c < n_ch;  c++)
    {
        const uint32_t *row = packed + (size_t) c*words_per_channel;
        const int nbits = (int) (row[0] & 0x7FFFu);
        const int nstat = (int) ((row[0] >> 16) & 0x7FFFu);
        const int32_t last = next[c];                   // one past this channel's reports in `order`
        if ((row[0] & 0x80000000u)  ||  nbits > (words_per_channel - 1)*32  ||  nstat != last - first  ||  nbits + nstat > cap)
            rc = SPANGPU_ERR_STATE;
        int8_t *ev = events + (size_t) c*cap;
        int n = 0;
        int32_t s = first;
        for (int b = 0;  b <= nbits  &&  n < cap;  b++)
        {
            while (s < last  &&  (int) (status[2 + 2*order[s]] & 0xFFFFu) == b  &&  n < cap)
                ev[n++] = (int8_t) (int16_t) (status[2 + 2*order[s++]] >> 16);
            if (b < nbits  &&  b < (words_per_channel - 1)*32  &&  n < cap)
                ev[n++] = (int8_t) ((row[1 + (b >> 5)] >> (b & 31)) & 1u);
        }
        counts[c] = n;
        first = last;
    }
    free(next);
    free(order);
    if (rc != SPANGPU_OK)
        return spangpu_set_error(rc, "packed events: a row or the status list was too short for what the call produced");
    return SPANGPU_OK;
}

// spangpu_modem_events() by way of the packed form: the same answer (events, counts, return value), a twentieth of the bytes
// over PCIe -- the device packs, one copy brings rows and status list up, the host spreads them out again.  What a shim with
// many channels on one bank calls.
int spangpu_modem_events_packed(spangpu_modem_t *m, const int8_t **events, const int32_t **counts)
{
    if (m == nullptr  ||  events == nullptr  ||  counts == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (m->last_cap <= 0)
        return spangpu_set_error(SPANGPU_ERR_STATE, "no spangpu_modem_rx() yet");
    V29_TRY(hipSetDevice(m->device));
    const int wpc = 1 + (m->last_cap + 31)/32;              // (no call delivers more data bits than its event buffer has room for)
    const int scap = (m->n_ch > 1024)  ?  m->n_ch  :  1024;
    if (wpc > m->packed_wpc  ||  scap > m->packed_status_cap)
    {
        if (m->d_packed) (void) hipFree(m->d_packed);
        if (m->h_packed) (void) hipHostFree(m->h_packed);
        m->d_packed = nullptr;
        m->h_packed = nullptr;
        m->packed_wpc = 0;
        const size_t words = (size_t) m->n_ch*wpc + 1 + 2*(size_t) scap;
        V29_TRY(hipMalloc(&m->d_packed, words*sizeof(uint32_t)));
        V29_TRY(hipHostMalloc(&m->h_packed, words*sizeof(uint32_t)));
        m->packed_wpc = wpc;
        m->packed_status_cap = scap;
    }
    uint32_t *d_status = m->d_packed + (size_t) m->n_ch*m->packed_wpc;
    int rc = spangpu_modem_pack_events(m, m->d_packed, m->packed_wpc, d_status, m->packed_status_cap);
    if (rc < 0)
        return rc;
    const size_t words = (size_t) m->n_ch*m->packed_wpc + 1 + 2*(size_t) m->packed_status_cap;
    V29_TRY(hipMemcpyAsync(m->h_packed, m->d_packed, words*sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream));
    V29_TRY(hipStreamSynchronize(m->stream));
    const uint32_t *h_status = m->h_packed + (size_t) m->n_ch*m->packed_wpc;
    if (h_status[0] > (uint32_t) m->packed_status_cap)
        return spangpu_modem_events(m, events, counts);     // (more status reports in one call than channels: the plain way)
    rc = spangpu_modem_unpack_events(m->h_packed, m->packed_wpc, h_status, m->packed_status_cap, m->n_ch, m->h_events, m->last_cap, m->h_count);
    if (rc < 0)
        return spangpu_modem_events(m, events, counts);     // (a row too short -- an event buffer overflow shows there: same error path)
    *events = m->h_events;
    *counts = m->h_count;
    return m->last_cap;
}

void *spangpu_modem_get_stream(spangpu_modem_t *m)
{
    return m  ?  (void *) m->stream  :  nullptr;
}

int spangpu_modem_bit_rate(const spangpu_modem_t *m)
{
    return m  ?  m->bit_rate  :  SPANGPU_ERR_BAD_ARG;
}

int spangpu_modem_device(const spangpu_modem_t *m)
{
    return m  ?  m->device  :  SPANGPU_ERR_BAD_ARG;
}

}   // extern "C"

// ---- host-side state edits: restart, fill-in, cutoff ---------------------------------------------
// These are the control-plane calls of the reference API; they run on the host against one channel's words.

static void zero_f(uint32_t *w, int first, int count)
{
    memset(w + first, 0, count*sizeof(uint32_t));
}

static void put_f(uint32_t *w, int idx, float v)
{
    memcpy(w + idx, &v, sizeof(float));
}

// v29_rx_restart(), v29rx.c:1019-1098
static int v29_restart_words(uint32_t *w, int bit_rate, int old_train)
{
    int32_t *iw = (int32_t *) (w + kV29Floats);
    switch (bit_rate)
    {
    case 9600: iw[VI_TRAINING_CD] = 0; break;
    case 7200: iw[VI_TRAINING_CD] = 2; break;
    case 4800: iw[VI_TRAINING_CD] = 4; break;
    default: return -1;
    }
    iw[VI_BIT_RATE] = bit_rate;
    zero_f(w, VF_RRC, kRrcLen);
    iw[VI_RRC_STEP] = 0;
    iw[VI_SCRAMBLE] = 0;
    iw[VI_TRAIN_SCRAMBLE] = 0x2A;
    iw[VI_STAGE] = V29_SYMBOL_ACQUISITION;
    iw[VI_TRAIN_COUNT] = 0;
    iw[VI_SIGNAL_PRESENT] = 0;
    iw[VI_HIGH_SAMPLE] = 0;
    iw[VI_LOW_SAMPLES] = 0;
    iw[VI_DROP_PENDING] = 0;
    iw[VI_OLD_TRAIN] = old_train  ?  1  :  0;
    memset(&iw[VI_DIFF_ANGLES], 0, 16*sizeof(int32_t));
    iw[VI_CARRIER_PHASE] = 0;
    iw[VI_POWER] = 0;
    iw[VI_CONSTEL] = 0;
    if (old_train)
    {
        iw[VI_PHASE_RATE] = iw[VI_PHASE_RATE_SAVE];
        memcpy(w + VF_EQ_COEFF, w + VF_EQ_SAVE, 2*kEqLen*sizeof(uint32_t));
        w[VF_AGC] = w[VF_AGC_SAVE];
    }
    else
    {
        iw[VI_PHASE_RATE] = (int32_t) (1700.0f*65536.0f*65536.0f/8000);
        zero_f(w, VF_EQ_COEFF, 2*kEqLen);
        put_f(w, VF_EQ_COEFF + 2*16, 3.0f);
        put_f(w, VF_AGC_SAVE, 0.0f);
        put_f(w, VF_AGC, (1.25f/1.0f)/735.0f);
    }
    zero_f(w, VF_EQ_BUF, 2*kEqLen);
    put_f(w, VF_EQ_DELTA, 0.21f/kEqLen);
    iw[VI_EQ_PUT_STEP] = kRrcSets*10/(3*2) - 1;
    iw[VI_EQ_STEP] = 0;
    put_f(w, VF_TRACK_I, 8000.0f);
    put_f(w, VF_TRACK_P, 8000000.0f);
    iw[VI_LAST_SAMPLE] = 0;
    iw[VI_EQ_SKIP] = 0;
    zero_f(w, VF_GLOW, 7);
    iw[VI_TOTAL_CORR] = 0;
    iw[VI_BAUD_HALF] = 0;
    return 0;
}

// v27ter_rx_restart(), v27ter_rx.c:1091-1160 (old_train is accepted and, as in the reference, has no effect)
static int v27_restart_words(uint32_t *w, int bit_rate)
{
    int32_t *iw = (int32_t *) (w + kV27Floats);
    if (bit_rate != 4800  &&  bit_rate != 2400)
        return -1;
    iw[WI_BIT_RATE] = bit_rate;
    zero_f(w, WF_RRC, kRrcLen);
    put_f(w, WF_TRAIN_ERR, 0.0f);
    iw[WI_RRC_STEP] = 0;
    iw[WI_SCRAMBLE] = 0x3C;
    iw[WI_PATTERN_COUNT] = 0;
    iw[WI_STAGE] = V27_SYMBOL_ACQUISITION;
    iw[WI_TRAINING_BC] = 0;
    iw[WI_TRAIN_COUNT] = 0;
    iw[WI_SIGNAL_PRESENT] = 0;
    iw[WI_HIGH_SAMPLE] = 0;
    iw[WI_LOW_SAMPLES] = 0;
    iw[WI_DROP_PENDING] = 0;
    memset(&iw[WI_DIFF_ANGLES], 0, 16*sizeof(int32_t));
    iw[WI_CARRIER_PHASE] = 0;
    put_f(w, WF_TRACK_I, 200000.0f);
    put_f(w, WF_TRACK_P, 10000000.0f);
    iw[WI_POWER] = 0;
    iw[WI_CONSTEL] = 0;
    iw[WI_PHASE_RATE] = (int32_t) (1800.0f*65536.0f*65536.0f/8000);
    put_f(w, WF_AGC, (1.414f/1.000000f)/283.0f);
    zero_f(w, WF_EQ_COEFF, 2*kV27EqLen);
    put_f(w, WF_EQ_COEFF + 2*17, 1.414f);
    zero_f(w, WF_EQ_BUF, 2*kV27EqLen);
    put_f(w, WF_EQ_DELTA, 0.25f/kV27EqLen);
    iw[WI_EQ_PUT_STEP] = (bit_rate == 4800)  ?  8*5/2  :  12*20/(3*2);
    iw[WI_EQ_STEP] = 0;
    iw[WI_EQ_SKIP] = 0;
    iw[WI_LAST_SAMPLE] = 0;
    iw[WI_GARDNER_INT] = 0;
    iw[WI_TOTAL_CORR] = 0;
    iw[WI_GARDNER_STEP] = 512;
    iw[WI_BAUD_HALF] = 0;
    return 0;
}

// v17_rx_restart(), v17rx.c:1399-1500
static int v17_restart_words(uint32_t *w, int bit_rate, int short_train)
{
    int32_t *iw = (int32_t *) (w + kV17Floats);
    switch (bit_rate)
    {
    case 14400: iw[XI_SPACE_MAP] = 0; iw[XI_BITS_PER_SYMBOL] = 6; break;
    case 12000: iw[XI_SPACE_MAP] = 1; iw[XI_BITS_PER_SYMBOL] = 5; break;
    case 9600: iw[XI_SPACE_MAP] = 2; iw[XI_BITS_PER_SYMBOL] = 4; break;
    case 7200: iw[XI_SPACE_MAP] = 3; iw[XI_BITS_PER_SYMBOL] = 3; break;
    case 4800: iw[XI_SPACE_MAP] = 0; iw[XI_BITS_PER_SYMBOL] = 2; break;
    default: return -1;
    }
    iw[XI_BIT_RATE] = bit_rate;
    zero_f(w, VF_RRC, kRrcLen);
    put_f(w, VF_TRAIN_ERR, 0.0f);
    iw[XI_RRC_STEP] = 0;
    iw[XI_DIFF] = 1;
    iw[XI_SCRAMBLE] = 0x2ECDD5;
    iw[XI_STAGE] = V17_SYMBOL_ACQUISITION;
    iw[XI_TRAIN_COUNT] = 0;
    iw[XI_SIGNAL_PRESENT] = 0;
    iw[XI_HIGH_SAMPLE] = 0;
    iw[XI_LOW_SAMPLES] = 0;
    iw[XI_DROP_PENDING] = 0;
    if (short_train != 2)
        iw[XI_SHORT_TRAIN] = short_train  ?  1  :  0;
    iw[XI_LAST_ANGLES] = 0;
    iw[XI_LAST_ANGLES + 1] = 0;
    memset(&iw[XI_DIFF_ANGLES], 0, 16*sizeof(int32_t));
    for (int k = 0;  k < 8;  k++)
        put_f(w, XF_DIST + k, k  ?  99.0f*1.0f  :  0.0f);
    memset(&iw[XI_FULL_PATH], 0, 256*sizeof(int32_t));
    iw[XI_TRELLIS_PTR] = 14;
    iw[XI_CARRIER_PHASE] = 0;
    iw[XI_POWER] = 0;
    zero_f(w, VF_EQ_BUF, 2*kEqLen);
    iw[XI_EQ_PUT_STEP] = kV17Sets*10/(3*2) - 1;
    iw[XI_EQ_STEP] = 0;
    iw[XI_EQ_SKIP] = 0;
    if (iw[XI_SHORT_TRAIN])
    {
        iw[XI_PHASE_RATE] = iw[XI_PHASE_RATE_SAVE];
        memcpy(w + VF_EQ_COEFF, w + VF_EQ_SAVE, 2*kEqLen*sizeof(uint32_t));
        put_f(w, VF_EQ_DELTA, 0.1f*(0.21f/kEqLen));
        w[VF_AGC] = w[VF_AGC_SAVE];
        put_f(w, VF_TRACK_I, 0.0f);
        put_f(w, VF_TRACK_P, 40000.0f);
    }
    else
    {
        iw[XI_PHASE_RATE] = (int32_t) (1800.0f*65536.0f*65536.0f/8000);
        zero_f(w, VF_EQ_COEFF, 2*kEqLen);
        put_f(w, VF_EQ_COEFF + 2*16, 3.0f);
        put_f(w, VF_EQ_DELTA, 0.21f/kEqLen);
        put_f(w, VF_AGC_SAVE, 0.0f);
        put_f(w, VF_AGC, (2.17f/1.000000f)/735.0f);
        put_f(w, VF_TRACK_I, 5000.0f);
        put_f(w, VF_TRACK_P, 40000.0f);
    }
    iw[XI_LAST_SAMPLE] = 0;
    zero_f(w, VF_GLOW, 7);
    iw[XI_TOTAL_CORR] = 0;
    iw[XI_BAUD_HALF] = 0;
    return 0;
}

static int fetch_words(spangpu_modem_t *m, int channel, uint32_t *w)
{
    V29_TRY(hipSetDevice(m->device));
    V29_TRY(hipStreamSynchronize(m->stream));
    V29_TRY(hipMemcpy2D(w, sizeof(uint32_t), m->state + channel, (size_t) m->n_ch*sizeof(uint32_t),
                        sizeof(uint32_t), m->n_words, hipMemcpyDeviceToHost));
    return SPANGPU_OK;
}

static int store_words(spangpu_modem_t *m, int channel, const uint32_t *w)
{
    V29_TRY(hipMemcpy2D(m->state + channel, (size_t) m->n_ch*sizeof(uint32_t), w, sizeof(uint32_t),
                        sizeof(uint32_t), m->n_words, hipMemcpyHostToDevice));
    return SPANGPU_OK;
}

extern "C" {

// One channel's state as spangpu_modem_state_words() 32 bit words: the float words, then the int words
// (order: "State word map" in v29_dev.hpp / v27ter_dev.hpp / v17_dev.hpp).
int spangpu_modem_get_state(spangpu_modem_t *m, int channel, uint32_t *words)
{
    if (m == nullptr  ||  channel < 0  ||  channel >= m->n_ch  ||  words == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int rc = fetch_words(m, channel, words);
    return (rc < 0)  ?  rc  :  m->n_words;
}

int spangpu_modem_set_state(spangpu_modem_t *m, int channel, const uint32_t *words)
{
    if (m == nullptr  ||  channel < 0  ||  channel >= m->n_ch  ||  words == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    V29_TRY(hipSetDevice(m->device));
    V29_TRY(hipStreamSynchronize(m->stream));
    return store_words(m, channel, words);
}

// v29_rx_restart(s, bit_rate, old_train) / v27ter_rx_restart(s, bit_rate, old_train) / v17_rx_restart(s, bit_rate,
// short_train) for one channel.  A V.29 channel may change rate; V.27ter and V.17 banks run one rate (their tables
// are per rate), so for them bit_rate must be the bank's.  Returns -1 for a rate the modem does not have, like the
// reference.
int spangpu_modem_restart_ex(spangpu_modem_t *m, int channel, int bit_rate, int train_flag)
{
    if (m == nullptr  ||  channel < 0  ||  channel >= m->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    uint32_t w[kMaxWords];
    int rc = fetch_words(m, channel, w);
    if (rc < 0)
        return rc;
    switch (m->kind)
    {
    case SPANGPU_V29:
        rc = v29_restart_words(w, bit_rate, train_flag);
        break;
    case SPANGPU_V27TER:
        rc = (bit_rate == m->bit_rate  ||  (bit_rate != 4800  &&  bit_rate != 2400))  ?  v27_restart_words(w, bit_rate)  :  -2;
        break;
    default:
        rc = (bit_rate == m->bit_rate  ||  spg_v17_constellation_size(bit_rate) < 0)  ?  v17_restart_words(w, bit_rate, train_flag)  :  -2;
        break;
    }
    if (rc == -2)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "this bank runs one bit rate; create another bank for a different rate");
    if (rc < 0)
        return -1;
    return store_words(m, channel, w);
}

// xxx_rx_restart(s, current rate, false)
int spangpu_modem_restart(spangpu_modem_t *m, int channel)
{
    if (m == nullptr  ||  channel < 0  ||  channel >= m->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    int rate = m->bit_rate;
    if (m->kind == SPANGPU_V29)
    {
        uint32_t w[kMaxWords];
        const int rc = fetch_words(m, channel, w);
        if (rc < 0)
            return rc;
        rate = (int) w[kV29Floats + VI_BIT_RATE];
    }
    return spangpu_modem_restart_ex(m, channel, rate, 0);
}

// xxx_rx_fillin(s, len): keep the carrier and symbol phase running over a gap (v29rx.c:967-996, v27ter_rx.c:1030-1067,
// v17rx.c:1320-1358)
int spangpu_modem_fillin(spangpu_modem_t *m, int channel, int len)
{
    if (m == nullptr  ||  channel < 0  ||  channel >= m->n_ch  ||  len < 0)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    uint32_t w[kMaxWords];
    const int rc = fetch_words(m, channel, w);
    if (rc < 0)
        return rc;
    int32_t *iw = (int32_t *) (w + m->n_floats);
    int i_present, i_stage, i_phase, i_rate, i_put, sets, add, parked;
    if (m->kind == SPANGPU_V29)
    {
        i_present = VI_SIGNAL_PRESENT; i_stage = VI_STAGE; i_phase = VI_CARRIER_PHASE; i_rate = VI_PHASE_RATE;
        i_put = VI_EQ_PUT_STEP; sets = kRrcSets; add = kRrcSets*10/(3*2); parked = V29_PARKED;
    }
    else if (m->kind == SPANGPU_V17)
    {
        i_present = XI_SIGNAL_PRESENT; i_stage = XI_STAGE; i_phase = XI_CARRIER_PHASE; i_rate = XI_PHASE_RATE;
        i_put = XI_EQ_PUT_STEP; sets = kV17Sets; add = kV17Sets*10/(3*2); parked = V17_PARKED;
    }
    else
    {
        i_present = WI_SIGNAL_PRESENT; i_stage = WI_STAGE; i_phase = WI_CARRIER_PHASE; i_rate = WI_PHASE_RATE;
        i_put = WI_EQ_PUT_STEP; parked = V27_PARKED;
        sets = (m->bit_rate == 4800)  ?  8  :  12;
        add = (m->bit_rate == 4800)  ?  8*5/2  :  12*20/(3*2);
    }
    if (iw[i_present] <= 0  ||  iw[i_stage] == parked)
        return 0;
    for (int i = 0;  i < len;  i++)
    {
        iw[i_phase] = (int32_t) ((uint32_t) iw[i_phase] + (uint32_t) iw[i_rate]);
        if ((iw[i_put] -= sets) <= 0)
            iw[i_put] += add;
    }
    return store_words(m, channel, w);
}

// xxx_rx_set_signal_cutoff(s, cutoff) (v29rx.c:163-169)
int spangpu_modem_set_signal_cutoff(spangpu_modem_t *m, int channel, float cutoff_dbm0)
{
    if (m == nullptr  ||  channel < -1  ||  channel >= m->n_ch)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int i_on = (m->kind == SPANGPU_V29)  ?  VI_ON_POWER  :  (m->kind == SPANGPU_V17)  ?  XI_ON_POWER  :  WI_ON_POWER;
    if (channel == -1)
    {
        // every channel of the bank (what fax_modems.c:416 does for each of its receivers): the two words are rows of the state
        V29_TRY(hipSetDevice(m->device));
        V29_TRY(hipStreamSynchronize(m->stream));
        uint32_t *row = m->state + (size_t) (m->n_floats + i_on)*m->n_ch;
        V29_TRY(hipMemsetD32((hipDeviceptr_t) row, (int32_t) (level_dbm0(cutoff_dbm0 + 2.5f)*0.4f), (size_t) m->n_ch));
        V29_TRY(hipMemsetD32((hipDeviceptr_t) (row + m->n_ch), (int32_t) (level_dbm0(cutoff_dbm0 - 2.5f)*0.4f), (size_t) m->n_ch));
        return SPANGPU_OK;
    }
    uint32_t w[kMaxWords];
    const int rc = fetch_words(m, channel, w);
    if (rc < 0)
        return rc;
    int32_t *iw = (int32_t *) (w + m->n_floats);
    iw[i_on] = (int32_t) (level_dbm0(cutoff_dbm0 + 2.5f)*0.4f);
    iw[i_on + 1] = (int32_t) (level_dbm0(cutoff_dbm0 - 2.5f)*0.4f);
    return store_words(m, channel, w);
}

// xxx_rx_set_signal_cutoff() of every channel with a cutoff of its own (an installation sets the carrier detector for the
// level of its lines): two rows of the state
int spangpu_modem_set_signal_cutoffs(spangpu_modem_t *m, const float *cutoff_dbm0)
{
    if (m == nullptr  ||  cutoff_dbm0 == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    const int i_on = (m->kind == SPANGPU_V29)  ?  VI_ON_POWER  :  (m->kind == SPANGPU_V17)  ?  XI_ON_POWER  :  WI_ON_POWER;
    int32_t *rows = (int32_t *) malloc((size_t) 2*m->n_ch*sizeof(int32_t));
    if (rows == nullptr)
        return spangpu_set_error(SPANGPU_ERR_NO_MEMORY, "malloc");
    for (int c = 0;  c < m->n_ch;  c++)
    {
        rows[c] = (int32_t) (level_dbm0(cutoff_dbm0[c] + 2.5f)*0.4f);
        rows[m->n_ch + c] = (int32_t) (level_dbm0(cutoff_dbm0[c] - 2.5f)*0.4f);
    }
    hipError_t e = hipSetDevice(m->device);
    if (e == hipSuccess)
        e = hipStreamSynchronize(m->stream);
    if (e == hipSuccess)
        e = hipMemcpy(m->state + (size_t) (m->n_floats + i_on)*m->n_ch, rows, (size_t) 2*m->n_ch*sizeof(int32_t), hipMemcpyHostToDevice);
    free(rows);
    if (e != hipSuccess)
        return spangpu_set_error(SPANGPU_ERR_HIP, "state upload failed");
    return SPANGPU_OK;
}

// The constant tables this library builds (for tests).  which: 0 sine [2048], 1 sqrt (as float) [193],
// 10/11 V.29 rx pulse shaper re/im [48*27], 12 V.29 Godard [7], 20/21 V.27ter 4800 re/im [8*27],
// 22/23 V.27ter 2400 re/im [12*27], 30/31 V.17 re/im [192*27], 32 V.17 Godard [7], 33 V.17 constellations [244*2].
// Returns the number of values written.
int spangpu_modem_table(int which, float *out, int max)
{
    static float re[192*27];
    static float im[192*27];
    uint16_t sq[193];
    int n = 0;
    const float *src = re;
    if (out == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null output");
    switch (which)
    {
    case 0:
        n = 2048;
        if (n <= max) spg_make_sine_table(out);
        return (n <= max)  ?  n  :  spangpu_set_error(SPANGPU_ERR_BAD_ARG, "buffer too small");
    case 1:
        n = 193;
        spg_make_sqrt_table(sq);
        for (int i = 0;  i < n  &&  i < max;  i++)
            out[i] = (float) sq[i];
        return n;
    case 10: case 11:
        n = 48*27;
        spg_make_rx_pulseshaper(48, 27, 1700.0, 2400.0, 0.5, re, im);
        src = (which & 1)  ?  im  :  re;
        break;
    case 12:
        n = 7;
        spg_make_godard(1700.0, 2400.0, 0.99, re);
        break;
    case 20: case 21:
        n = 8*27;
        spg_make_rx_pulseshaper(8, 27, 1800.0, 1600.0, 0.5, re, im);
        src = (which & 1)  ?  im  :  re;
        break;
    case 22: case 23:
        n = 12*27;
        spg_make_rx_pulseshaper(12, 27, 1800.0, 1200.0, 0.5, re, im);
        src = (which & 1)  ?  im  :  re;
        break;
    case 30: case 31:
        n = 192*27;
        spg_make_rx_pulseshaper(192, 27, 1800.0, 2400.0, 0.5, re, im);
        src = (which & 1)  ?  im  :  re;
        break;
    case 32:
        n = 7;
        spg_make_godard(1800.0, 2400.0, 0.99, re);
        break;
    case 33:
    {
        // the five V.17 / V.32bis constellations back to back: 14400, 12000, 9600, 7200, 4800 bps, {re, im} each
        static const int rates[5] = {14400, 12000, 9600, 7200, 4800};
        int8_t pts[128][2];
        n = 0;
        for (int r = 0;  r < 5;  r++)
        {
            const int m = spg_make_v17_constellation(rates[r], pts);
            for (int i = 0;  i < m;  i++)
            {
                re[n++] = pts[i][0];
                re[n++] = pts[i][1];
            }
        }
        break;
    }
    default:
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "unknown table");
    }
    if (n > max)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "buffer too small");
    memcpy(out, src, n*sizeof(float));
    return n;
}

// The V.17 receiver's soft-decision maps as built by this library: maps [4*36*36*8], map_4800 [36*36].
int spangpu_v17_rx_maps(uint8_t *maps, uint8_t *map_4800)
{
    if (maps == nullptr  ||  map_4800 == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null output");
    spg_make_v17_rx_maps(maps, map_4800);
    return SPANGPU_OK;
}

}   // extern "C"
