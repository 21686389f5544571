/* ref_glue_mt.c -- TEST / BENCH INFRASTRUCTURE: a pthread driver that times the REFERENCE's receive functions on
 * the host cores for the cpu_baseline leg of bench.py and tools/bench_paths.py.  One call runs the whole bounded
 * sample: every thread takes a static slice of the channel objects and loops over frames and repetitions inside C
 * (the round-1 baseline drove the reference from one Python thread per core, one ctypes call per 64-channel frame:
 * it measured the interpreter's hand-off, not the reference).  The elapsed time is taken between two barriers, so
 * thread start-up is outside it.  Nothing here is linked into the product. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <inttypes.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

#include "spandsp/telephony.h"
#include "spandsp/alloc.h"
#include "spandsp/logging.h"
#include "spandsp/fast_convert.h"
#include "spandsp/queue.h"
#include "spandsp/complex.h"
#include "spandsp/dds.h"
#include "spandsp/tone_detect.h"
#include "spandsp/tone_generate.h"
#include "spandsp/super_tone_rx.h"
#include "spandsp/dtmf.h"
#include "spandsp/bell_r2_mf.h"
#include "spandsp/saturated.h"
#include "spandsp/dc_restore.h"
#include "spandsp/bit_operations.h"
#include "spandsp/echo.h"
#include "spandsp/async.h"
#include "spandsp/power_meter.h"
#include "spandsp/vector_float.h"
#include "spandsp/complex_vector_float.h"
#include "spandsp/godard.h"
#include "spandsp/fsk.h"
#include "spandsp/modem_connect_tones.h"
#include "spandsp/sig_tone.h"
#include "spandsp/v29rx.h"
#include "spandsp/v27ter_rx.h"
#include "spandsp/v17rx.h"

#define GLUE __attribute__((visibility("default")))

enum
{
    GLUE_MT_DTMF = 0, GLUE_MT_BELL_MF, GLUE_MT_R2_MF, GLUE_MT_SUPER_TONE, GLUE_MT_V29, GLUE_MT_V27TER, GLUE_MT_V17,
    GLUE_MT_FSK, GLUE_MT_MCT, GLUE_MT_SIGTONE
};

typedef struct
{
    int kind;
    void **states;
    const int16_t *frames;          /* [n_frames][...] : frame f of channel c at frames + f*frame_stride + c*ch_stride */
    const int16_t *frames2;         /* echo: the received signal (frames = transmitted) */
    int16_t *out;                   /* echo: clean samples of the last frame, [n_ch][samples] */
    long long ch_stride;
    long long frame_stride;
    int n_frames;
    int samples;
    int loops;
    int lo;
    int hi;
    pthread_barrier_t *gate;
} glue_mt_job_t;

static void rx_one(int kind, void *s, const int16_t *amp, int n)
{
    switch (kind)
    {
    case GLUE_MT_DTMF:          dtmf_rx((dtmf_rx_state_t *) s, amp, n); break;
    case GLUE_MT_BELL_MF:       bell_mf_rx((bell_mf_rx_state_t *) s, amp, n); break;
    case GLUE_MT_R2_MF:         r2_mf_rx((r2_mf_rx_state_t *) s, amp, n); break;
    case GLUE_MT_SUPER_TONE:    super_tone_rx((super_tone_rx_state_t *) s, amp, n); break;
    case GLUE_MT_V29:           v29_rx((v29_rx_state_t *) s, amp, n); break;
    case GLUE_MT_V27TER:        v27ter_rx((v27ter_rx_state_t *) s, amp, n); break;
    case GLUE_MT_V17:           v17_rx((v17_rx_state_t *) s, amp, n); break;
    case GLUE_MT_FSK:           fsk_rx((fsk_rx_state_t *) s, amp, n); break;
    case GLUE_MT_MCT:           modem_connect_tones_rx((modem_connect_tones_rx_state_t *) s, amp, n); break;
    case GLUE_MT_SIGTONE:
        {
            /* the receiver rewrites its frame: it gets a copy, so that every pass sees the same signal */
            int16_t scratch[1024];
            const int m = (n > 1024)  ?  1024  :  n;

            memcpy(scratch, amp, sizeof(int16_t)*m);
            sig_tone_rx((sig_tone_rx_state_t *) s, scratch, m);
        }
        break;
    }
}

static void *rx_worker(void *arg)
{
    glue_mt_job_t *j = (glue_mt_job_t *) arg;
    int l;
    int f;
    int c;

    pthread_barrier_wait(j->gate);
    for (l = 0;  l < j->loops;  l++)
    {
        for (f = 0;  f < j->n_frames;  f++)
        {
            const int16_t *base = j->frames + (long long) f*j->frame_stride;

            for (c = j->lo;  c < j->hi;  c++)
                rx_one(j->kind, j->states[c], base + (long long) c*j->ch_stride, j->samples);
        }
    }
    pthread_barrier_wait(j->gate);
    return NULL;
}

static void *echo_worker(void *arg)
{
    glue_mt_job_t *j = (glue_mt_job_t *) arg;
    int l;
    int f;
    int c;
    int i;

    pthread_barrier_wait(j->gate);
    for (l = 0;  l < j->loops;  l++)
    {
        for (f = 0;  f < j->n_frames;  f++)
        {
            const int16_t *tx = j->frames + (long long) f*j->frame_stride;
            const int16_t *rx = j->frames2 + (long long) f*j->frame_stride;

            for (c = j->lo;  c < j->hi;  c++)
            {
                echo_can_state_t *ec = (echo_can_state_t *) j->states[c];
                const int16_t *t = tx + (long long) c*j->ch_stride;
                const int16_t *r = rx + (long long) c*j->ch_stride;
                int16_t *o = j->out + (long long) c*j->samples;

                for (i = 0;  i < j->samples;  i++)
                    o[i] = echo_can_update(ec, t[i], r[i]);
            }
        }
    }
    pthread_barrier_wait(j->gate);
    return NULL;
}

static double run_jobs(glue_mt_job_t *proto, int n_ch, int n_threads, void *(*worker)(void *))
{
    pthread_t *th;
    glue_mt_job_t *jobs;
    pthread_barrier_t gate;
    struct timespec t0;
    struct timespec t1;
    int i;

    if (n_threads < 1)
        n_threads = 1;
    if (n_threads > n_ch)
        n_threads = n_ch;
    th = (pthread_t *) malloc(sizeof(pthread_t)*n_threads);
    jobs = (glue_mt_job_t *) malloc(sizeof(glue_mt_job_t)*n_threads);
    pthread_barrier_init(&gate, NULL, n_threads + 1);
    for (i = 0;  i < n_threads;  i++)
    {
        jobs[i] = *proto;
        jobs[i].lo = (int) ((long long) n_ch*i/n_threads);
        jobs[i].hi = (int) ((long long) n_ch*(i + 1)/n_threads);
        jobs[i].gate = &gate;
        pthread_create(&th[i], NULL, worker, &jobs[i]);
    }
    pthread_barrier_wait(&gate);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_barrier_wait(&gate);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (i = 0;  i < n_threads;  i++)
        pthread_join(th[i], NULL);
    pthread_barrier_destroy(&gate);
    free(th);
    free(jobs);
    return (double) (t1.tv_sec - t0.tv_sec) + 1.0e-9*(double) (t1.tv_nsec - t0.tv_nsec);
}

/* Runs `loops` passes over `n_frames` frames of `samples` samples through the n_ch receiver objects of `kind` on
   n_threads threads; returns the elapsed seconds. */
GLUE double glue_mt_rx(int kind, void **states, const int16_t *frames, int n_ch, int n_frames, int samples,
                       long long ch_stride, long long frame_stride, int loops, int n_threads)
{
    glue_mt_job_t p;

    memset(&p, 0, sizeof(p));
    p.kind = kind;
    p.states = states;
    p.frames = frames;
    p.ch_stride = ch_stride;
    p.frame_stride = frame_stride;
    p.n_frames = n_frames;
    p.samples = samples;
    p.loops = loops;
    return run_jobs(&p, n_ch, n_threads, rx_worker);
}

/* The same for echo_can_update(): tx / rx frames laid out alike, clean samples of the last frame into out[n_ch][samples]. */
GLUE double glue_mt_echo(void **states, const int16_t *tx, const int16_t *rx, int16_t *out, int n_ch, int n_frames,
                         int samples, long long ch_stride, long long frame_stride, int loops, int n_threads)
{
    glue_mt_job_t p;

    memset(&p, 0, sizeof(p));
    p.states = states;
    p.frames = tx;
    p.frames2 = rx;
    p.out = out;
    p.ch_stride = ch_stride;
    p.frame_stride = frame_stride;
    p.n_frames = n_frames;
    p.samples = samples;
    p.loops = loops;
    return run_jobs(&p, n_ch, n_threads, echo_worker);
}
