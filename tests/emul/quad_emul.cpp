// quad_emul.cpp -- TEST INFRASTRUCTURE: runs the four-lanes-per-channel receiver kernels of spandsp_amd/csrc (the very
// source the GPU runs: v29_quad.hpp ...) on the host, one channel at a time, its four lanes as four fibers that hand over
// at every exchange point of quad_ctx.hpp.  tests/test_quad_emul.py compares the result with the oracle.  Nothing in the
// product uses this.
#define SPG_HOST_EMUL 1
#include <stdio.h>
#include <ucontext.h>

#include "../../spandsp_amd/csrc/v29_quad.hpp"
#include "../../spandsp_amd/csrc/v17_quad.hpp"
#include "../../spandsp_amd/csrc/v27ter_quad.hpp"
#include "../../spandsp_amd/csrc/modem_tables.h"

namespace spg {

struct Sched
{
    ucontext_t main_ctx;
    ucontext_t ctx[4];
    char *stack[4];
    bool done[4];
    int order[4];               // the sequence in which the lanes get their turn
    int where[4];               // lane -> index in order
    int site[4];
    int gen_seen[4];
    int errors;
    void (*body)(int lane, void *arg);
    void *arg;
    int cur;
};

static Sched *g_sched;

static void next_from(Sched *s, int lane, ucontext_t *from)
{
    for (int k = 1;  k <= 4;  k++)
    {
        const int nl = s->order[(s->where[lane] + k) & 3];
        if (nl == lane)
            break;
        if (!s->done[nl])
        {
            s->cur = nl;
            swapcontext(from, &s->ctx[nl]);
            return;
        }
    }
    // nobody else can run
    bool all = true;
    for (int k = 0;  k < 4;  k++)
        all = all  &&  (s->done[k]  ||  k == lane);
    if (!s->done[lane])
    {
        // this lane waits at an exchange, the others have finished: the lanes did not take the same path
        s->errors++;
        s->done[lane] = true;
    }
    (void) all;
    swapcontext(from, &s->main_ctx);
}

void quad_host_yield(QuadHostState *st, int lane, int site)
{
    Sched *s = (Sched *) st->impl;
    // every lane must come by the same call sites in the same order: the first lane of the turn order records the site
    // of a generation, the others compare (the first lane is never more than one generation ahead)
    const int g = st->gen[lane];
    if (lane == s->order[0])
    {
        s->site[g & 1] = site;
        s->gen_seen[g & 1] = g;
    }
    else if (s->gen_seen[g & 1] != g  ||  s->site[g & 1] != site)
    {
        if (s->errors < 5)
            fprintf(stderr, "quad_emul: lane %d at site %d gen %d, lane %d was at site %d gen %d\n", lane, site, g, s->order[0], s->site[g & 1], s->gen_seen[g & 1]);
        s->errors++;
    }
    next_from(s, lane, &s->ctx[lane]);
}

static void trampoline(int lane)
{
    Sched *s = g_sched;
    s->body(lane, s->arg);
    s->done[lane] = true;
    next_from(s, lane, &s->ctx[lane]);
}

static int run_quad(void (*body)(int, void *), void *arg, QuadHostState *st, const int order[4])
{
    static Sched s;
    memset(&s, 0, sizeof(s));
    g_sched = &s;
    s.body = body;
    s.arg = arg;
    st->impl = &s;
    for (int k = 0;  k < 4;  k++)
    {
        s.order[k] = order[k];
        s.where[order[k]] = k;
    }
    const size_t stack_bytes = 1 << 20;
    for (int k = 0;  k < 4;  k++)
    {
        s.stack[k] = (char *) malloc(stack_bytes);
        getcontext(&s.ctx[k]);
        s.ctx[k].uc_stack.ss_sp = s.stack[k];
        s.ctx[k].uc_stack.ss_size = stack_bytes;
        s.ctx[k].uc_link = &s.main_ctx;
        makecontext(&s.ctx[k], (void (*)()) trampoline, 1, k);
    }
    s.cur = s.order[0];
    swapcontext(&s.main_ctx, &s.ctx[s.order[0]]);
    for (int k = 0;  k < 4;  k++)
    {
        if (!s.done[k])
            s.errors++;
        free(s.stack[k]);
    }
    return s.errors;
}

struct V29Job
{
    V29Launch L;
    V29QuadTables T;
    V29QuadChan C;
    uint32_t pcm[kQuadPcmStride];
    float2 rrc[kQuadRrcStride];
    float2 u[kQuadEqStride];
    float taps[kQuadTapStride];
    QuadHostState st;
};

static void v29_body(int lane, void *arg)
{
    V29Job *j = (V29Job *) arg;
    QuadHost q;
    q.st = &j->st;
    q.lane = lane;
    v29_quad_run(q, j->L, 0, j->T, j->C);
}

struct V17Job
{
    V17Launch L;
    V17QuadTables T;
    V17QuadChan C;
    uint32_t pcm[kQuadPcmStride];
    float2 rrc[kQuadRrcStride];
    float2 u[kQuad17EqStride];
    float taps[kQuadTapStride];
    uint32_t trellis[kQuad17TrellisStride];
    QuadHostState st;
};

static void v17_body(int lane, void *arg)
{
    V17Job *j = (V17Job *) arg;
    QuadHost q;
    q.st = &j->st;
    q.lane = lane;
    v17_quad_run(q, j->L, 0, j->T, j->C);
}

struct V27Job
{
    V27Launch L;
    V27QuadTables T;
    V29QuadChan C;
    uint32_t pcm[kQuadPcmStride];
    float2 rrc[kQuadRrcStride];
    float2 u[kQuad27EqStride];
    float taps[kQuadTapStride];
    QuadHostState st;
};

static void v27_body(int lane, void *arg)
{
    V27Job *j = (V27Job *) arg;
    QuadHost q;
    q.st = &j->st;
    q.lane = lane;
    v27_quad_run(q, j->L, 0, j->T, j->C);
}

}   // namespace spg

using namespace spg;

static V29Tables g_v29_tab;
static bool g_v29_tab_ready;

// One channel's v29_rx() call: state = the 281 state words (in and out), returns the number of events or < 0.
extern "C" int emul_v29_rx(uint32_t *state, const int16_t *amp, int n, int8_t *events, int ev_cap, const int *order)
{
    if (!g_v29_tab_ready)
    {
        memset(&g_v29_tab, 0, sizeof(g_v29_tab));
        spg_make_rx_pulseshaper(kRrcSets, kRrcLen, 1700.0, 2400.0, 0.5, g_v29_tab.rrc_re, g_v29_tab.rrc_im);
        spg_make_sine_table(g_v29_tab.sine);
        spg_make_sqrt_table(g_v29_tab.sqrt_tab);
        spg_make_godard(1700.0, 2400.0, 0.99, g_v29_tab.godard);
        g_v29_tab.coarse_trigger = 1000.0f;
        g_v29_tab.fine_trigger = 30.0f;
        g_v29_tab.coarse_step = 5;
        g_v29_tab.fine_step = 1;
        spg_make_v29_space_map(g_v29_tab.space_map);
        g_v29_tab_ready = true;
    }
    static V29Job job;
    memset(&job, 0, sizeof(job));
    int32_t count = 0;
    job.L.amp = amp;
    job.L.stride = n;
    job.L.samples = n;
    job.L.lens = nullptr;
    job.L.n_ch = 1;
    job.L.state = state;
    job.L.events = events;
    job.L.ev_count = &count;
    job.L.ev_cap = ev_cap;
    job.L.tab = &g_v29_tab;
    v29_quad_tables(job.T, g_v29_tab, 0, 1);
    // LDS starts out as rubbish on the device
    memset(job.pcm, 0xA5, sizeof(job.pcm));
    memset(job.rrc, 0xA5, sizeof(job.rrc));
    memset(job.u, 0xA5, sizeof(job.u));
    memset(job.taps, 0xA5, sizeof(job.taps));
    job.C.pcm = job.pcm;
    job.C.rrc = job.rrc;
    job.C.u = job.u;
    job.C.taps = job.taps;
    const int errs = run_quad(v29_body, &job, &job.st, order);
    if (errs)
        return -errs;
    return count;
}

static V17Tables g_v17_tab;
static int g_v17_rate;

// One channel's v17_rx() call: state = the 547 state words (in and out), returns the number of events or < 0.
extern "C" int emul_v17_rx(int bit_rate, uint32_t *state, const int16_t *amp, int n, int8_t *events, int ev_cap, const int *order)
{
    if (g_v17_rate != bit_rate)
    {
        // as spangpu_modem_create() builds them (modem_api.hip)
        V17Tables *t = &g_v17_tab;
        memset(t, 0, sizeof(*t));
        float *re = (float *) malloc(2*kV17Sets*kRrcLen*sizeof(float));
        uint8_t *maps = (uint8_t *) malloc(4*36*36*8 + 36*36);
        int8_t pts[128][2];
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
        const int space_map = (bit_rate == 12000)  ?  1  :  (bit_rate == 9600)  ?  2  :  (bit_rate == 7200)  ?  3  :  0;
        if (bit_rate == 4800)
            memcpy(t->map, maps + 4*36*36*8, 36*36);
        else
            memcpy(t->map, maps + (size_t) space_map*36*36*8, 36*36*8);
        free(re);
        free(maps);
        g_v17_rate = bit_rate;
    }
    static V17Job job;
    memset(&job, 0, sizeof(job));
    int32_t count = 0;
    job.L.amp = amp;
    job.L.stride = n;
    job.L.samples = n;
    job.L.lens = nullptr;
    job.L.n_ch = 1;
    job.L.bit_rate = bit_rate;
    job.L.state = state;
    job.L.events = events;
    job.L.ev_count = &count;
    job.L.ev_cap = ev_cap;
    job.L.tab = &g_v17_tab;
    v17_quad_tables(job.T, g_v17_tab, 0, 1);
    memset(job.pcm, 0xA5, sizeof(job.pcm));
    memset(job.rrc, 0xA5, sizeof(job.rrc));
    memset(job.u, 0xA5, sizeof(job.u));
    memset(job.taps, 0xA5, sizeof(job.taps));
    memset(job.trellis, 0xA5, sizeof(job.trellis));
    job.C.pcm = job.pcm;
    job.C.rrc = job.rrc;
    job.C.u = job.u;
    job.C.taps = job.taps;
    job.C.trellis = job.trellis;
    const int errs = run_quad(v17_body, &job, &job.st, order);
    if (errs)
        return -errs;
    return count;
}

static V27Tables g_v27_tab;
static bool g_v27_ready;

// One channel's v27ter_rx() call: state = the 270 state words (in and out), returns the number of events or < 0.
extern "C" int emul_v27ter_rx(int bit_rate, uint32_t *state, const int16_t *amp, int n, int8_t *events, int ev_cap, const int *order)
{
    if (!g_v27_ready)
    {
        V27Tables *t = &g_v27_tab;
        memset(t, 0, sizeof(*t));
        spg_make_rx_pulseshaper(8, kRrcLen, 1800.0, 1600.0, 0.5, t->re4800, t->im4800);
        spg_make_rx_pulseshaper(12, kRrcLen, 1800.0, 1200.0, 0.5, t->re2400, t->im2400);
        spg_make_sine_table(t->sine);
        spg_make_sqrt_table(t->sqrt_tab);
        g_v27_ready = true;
    }
    static V27Job job;
    memset(&job, 0, sizeof(job));
    int32_t count = 0;
    job.L.amp = amp;
    job.L.stride = n;
    job.L.samples = n;
    job.L.lens = nullptr;
    job.L.n_ch = 1;
    job.L.bit_rate = bit_rate;
    job.L.state = state;
    job.L.events = events;
    job.L.ev_count = &count;
    job.L.ev_cap = ev_cap;
    job.L.tab = &g_v27_tab;
    v27_quad_tables(job.T, g_v27_tab, bit_rate == 4800, 0, 1);
    memset(job.pcm, 0xA5, sizeof(job.pcm));
    memset(job.rrc, 0xA5, sizeof(job.rrc));
    memset(job.u, 0xA5, sizeof(job.u));
    memset(job.taps, 0xA5, sizeof(job.taps));
    job.C.pcm = job.pcm;
    job.C.rrc = job.rrc;
    job.C.u = job.u;
    job.C.taps = job.taps;
    const int errs = run_quad(v27_body, &job, &job.st, order);
    if (errs)
        return -errs;
    return count;
}
