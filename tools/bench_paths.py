#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json (bench.py carries the headline DTMF line):

    python tools/bench_paths.py --workload mixed  [--channels 131072]   # configs[2]: Bell MF + R2 MF + super-tone banks
    python tools/bench_paths.py --workload v29    [--channels 16384]    # configs[3]
    python tools/bench_paths.py --workload echo   [--channels 131072]   # configs[4], one GPU's shard of 1 048 576
    python tools/bench_paths.py --workload v17 | v27ter                 # the other receivers of SURVEY 8(a)

One JSON line per run, same fields as bench.py.  A step = one 160-sample frame of every channel, inputs resident
in HBM.  Synthetic inputs: V.29 = the committed reference transmission (tests/golden/v29_9600.npz: training + PRBS data
from the reference's own modulator) repeated, with per-channel delay, gain and AWGN applied on the GPU."""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FRAME = 160
HBM_PEAK_GBPS = 8000.0


MODEMS = {"v29": ("v29_9600.npz", 9600, 281), "v17": ("v17_14400.npz", 14400, 547), "v27ter": ("v27ter_4800.npz", 4800, 270)}


def synth_v29(n_ch, n_frames, dev, seed, fixture="v29_9600.npz"):
    g = np.load(os.path.join(ROOT, "tests", "golden", fixture))
    base = torch.tensor(g["amp"].astype(np.float32), device=dev)
    period = base.numel() + 270
    base = torch.cat([base, torch.zeros(270, device=dev)])
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    delay = torch.randint(0, FRAME, (n_ch, 1), device=dev, generator=gen)
    gain = torch.pow(10.0, torch.empty(n_ch, 1, device=dev).uniform_(-14.0, 3.0, generator=gen)/20.0)
    sigma = torch.empty(n_ch, 1, device=dev).uniform_(1.0, 12.0 if "v27" in fixture else 40.0, generator=gen)
    out = torch.empty(n_frames, n_ch, FRAME, dtype=torch.int16, device=dev)
    for f in range(n_frames):
        t = torch.arange(f*FRAME, (f + 1)*FRAME, device=dev).unsqueeze(0) - delay
        x = torch.where(t >= 0, base[torch.remainder(t, period)], torch.zeros((), device=dev))
        x = x*gain + sigma*torch.randn(n_ch, FRAME, device=dev, generator=gen)
        out[f] = torch.clamp(torch.round(x), -32768, 32767).to(torch.int16)
    return out


def synth_v29_on_device(n_ch, n_frames, dev, stream, seed, modem="v29", line="contract"):
    """V.29 9600 bps / V.27ter 4800 bps / V.17 14400 bps input made where it is consumed: a transmitter bank (the reference's
    modulator, bit-exact, on the device) writes every channel's own transmission (its own data bits) frame by frame into HBM,
    and a noise source bank (the reference's awgn()) mixes line noise into it.  Returns int16 [n_frames, n_ch, FRAME] and a
    description of the lines.
    line = "contract": SURVEY 8(d)-4 as written -- every channel's carrier uniform in nominal +- 7 Hz, its level uniform in
    -30 .. -10 dBm0, AWGN at an SNR uniform in 25 .. 40 dB (V.27ter: 38 .. 50 dB, its receiver in the reference gives up training
    below some 35 dB); the random start delay is applied by the caller.  line = "in_step": the rounds-1-to-5 workload (carrier
    nominal, every 4th channel off -14 dBm0, noise of 1 .. 30 LSB rms)."""
    from spandsp_amd import engine
    rng = np.random.default_rng(seed)
    seeds = rng.integers(1, 0x7FFF, n_ch).astype(np.uint32)
    tx = {"v29": lambda: engine.V29TxBank(n_ch, 9600, False, seeds), "v27ter": lambda: engine.V27terTxBank(n_ch, 4800, False, seeds),
          "v17": lambda: engine.V17TxBank(n_ch, 14400, False, seeds)}[modem]()
    tx.set_stream(ctypes.c_void_p(stream.cuda_stream))
    nominal = 1700.0 if modem == "v29" else 1800.0
    if line == "contract":
        level = rng.uniform(-30.0, -10.0, n_ch)
        hz = nominal + rng.uniform(-7.0, 7.0, n_ch)
        snr = rng.uniform(38.0, 50.0, n_ch) if modem == "v27ter" else rng.uniform(25.0, 40.0, n_ch)
        # (A-B runs of what each condition costs: LINE_PARTS=carrier,level,snr leaves out what is not named)
        parts = os.environ.get("LINE_PARTS", "carrier,level,snr").split(",")
        if "carrier" not in parts:
            hz[:] = nominal
        if "level" not in parts:
            level[:] = -14.0
        if "snr" not in parts:
            snr[:] = 45.0
        tx.line(level, hz)
        noise_dbm0 = level - snr
        what = "carrier %.0f Hz +- 7 Hz, level -30 .. -10 dBm0, AWGN at SNR %s dB, per channel" % (nominal, "38 .. 50" if modem == "v27ter" else "25 .. 40")
    else:
        for c in range(0, n_ch, 4):                             # a spread of levels (every 4th channel moved off -14 dBm0)
            tx.power(c, float(rng.uniform(-26.0, -10.0)))
        top = {"v29": 30.0, "v27ter": 12.0, "v17": 10.0}[modem]
        sigma = rng.uniform(1.0, top, n_ch)
        noise_dbm0 = 20.0*np.log10(sigma/32768.0) + 3.14 + 3.02
        what = "carrier nominal, -14 dBm0 (every 4th channel -26 .. -10), AWGN of 1 .. %d LSB rms" % int(top)
    out = torch.empty(n_frames, n_ch, FRAME, dtype=torch.int16, device=dev)
    for f in range(n_frames):
        tx.tx_device(ctypes.c_void_p(out[f].data_ptr()), FRAME, FRAME)
    tx.sync()
    # line noise from the noise source bank (the reference's awgn(), on the device)
    noise = engine.AwgnBank(rng.integers(1, 2_000_000, n_ch), noise_dbm0)
    noise.set_stream(ctypes.c_void_p(stream.cuda_stream))
    for f in range(n_frames):
        noise.tx_device(ctypes.c_void_p(out[f].data_ptr()), FRAME, FRAME, mix=True)
    noise.sync()
    return out, what, (level if line == "contract" else None)


def bench_v29_tx(args, dev, stream):
    """the V.29 transmitter bank writing 160-sample frames into HBM."""
    from spandsp_amd import engine
    n_ch = args.channels or 65536
    tx = engine.V29TxBank(n_ch, 9600)
    tx.set_stream(ctypes.c_void_p(stream.cuda_stream))
    out = torch.zeros(4, n_ch, FRAME, dtype=torch.int16, device=dev)

    def step(i):
        tx.tx_device(ctypes.c_void_p(out[i % 4].data_ptr()), FRAME, FRAME)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record(stream)
        step(args.warmup + i)
        evs[i][1].record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [a.elapsed_time(b) for a, b in evs]
    avg_ms = sum(per)/len(per)
    rms = float(out.float().pow(2).mean().sqrt())
    alg = n_ch*(FRAME*2 + 2*32*4)
    value = args.steps*n_ch*FRAME/dt/1e6
    return {
        "metric": "Msamples/s of batched V.29 9600 bps transmit (signal source bank)", "value": value, "unit": "Msamples/s",
        "realtime_channels": value*1e6/8000.0, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt*1e3/args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "config": {"workload": "v29_tx bank, %d channels x %d-sample frames" % (n_ch, FRAME),
                                        "channels_per_gpu": n_ch, "rms_of_last_frames": rms},
        "roofline": {"bound": "hbm", "kernel": "modemtx_bank_kernel<V.29>", "achieved": alg/(avg_ms*1e-3)/1e9, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": alg/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS, "traffic": None, "alg_bytes_per_launch": alg,
                     "avg_launch_us": avg_ms*1e3, "min_launch_us": min(per)*1e3,
                     "note": "sample-serial modulator, one channel per lane: latency bound at one wave per SIMD"},
        "cpu_baseline": cpu_v29_tx(args)}


def cpu_v29_tx(args):
    """The reference's own v29_tx() (oracle/_ref) carrying a PRBS, on one host core (a modulator is a serial loop)."""
    if args.no_cpu_baseline:
        return None
    import oracle
    from oracle import ref
    if not oracle.have_ref():
        return None
    with ref.quiet_stdout():
        ref.v29_tx(9600, 200000, seed=3)
        t1 = time.perf_counter()
        n = 0
        while time.perf_counter() - t1 < 3.0:
            n += len(ref.v29_tx(9600, 400000, seed=5 + n))
        dt = time.perf_counter() - t1
    return {"value": n/dt/1e6, "unit": "Msamples/s", "cores": 1, "kind": "reference",
            "sample": "oracle/_ref v29_tx() 9600 bps, 400000-sample runs for %.1f s on one core" % dt}


def bench_awgn(args, dev, stream):
    """the noise source bank (awgn x N) writing 160-sample frames into HBM."""
    from spandsp_amd import engine
    n_ch = args.channels or 65536
    rng = np.random.default_rng(3)
    bank = engine.AwgnBank(rng.integers(1, 2_000_000, n_ch), rng.uniform(-50.0, -10.0, n_ch))
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    out = torch.zeros(4, n_ch, FRAME, dtype=torch.int16, device=dev)

    def step(i):
        bank.tx_device(ctypes.c_void_p(out[i % 4].data_ptr()), FRAME, FRAME)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record(stream)
        step(args.warmup + i)
        evs[i][1].record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [a.elapsed_time(b) for a, b in evs]
    avg_ms = sum(per)/len(per)
    rms = float(out.float().pow(2).mean().sqrt())
    alg = n_ch*(FRAME*2 + 2*202*4)
    value = args.steps*n_ch*FRAME/dt/1e6
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import ref
        t1 = time.perf_counter()
        n_cpu = 0
        while time.perf_counter() - t1 < 5.0:
            ref.awgn(12345 + n_cpu, -30.0, 400000)
            n_cpu += 400000
        cpu = {"value": n_cpu/(time.perf_counter() - t1)/1e6, "unit": "Msamples/s", "cores": 1, "kind": "reference",
               "sample": "oracle/_ref awgn(), 400000-sample runs for 5 s on one thread"}
    return {
        "metric": "Msamples/s of batched awgn (noise source bank)", "value": value, "unit": "Msamples/s",
        "realtime_channels": value*1e6/8000.0, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt*1e3/args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "config": {"workload": "awgn bank, %d channels x %d-sample frames" % (n_ch, FRAME),
                                        "channels_per_gpu": n_ch, "rms_of_last_frames": rms},
        "roofline": {"bound": "hbm", "kernel": "awgn_bank_kernel", "achieved": alg/(avg_ms*1e-3)/1e9, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": alg/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS, "traffic": None, "alg_bytes_per_launch": alg,
                     "avg_launch_us": avg_ms*1e3, "min_launch_us": min(per)*1e3,
                     "note": "binary64 Box-Muller (log, sqrt, divide per pair) behind a data-dependent LDS shuffle table"},
        "cpu_baseline": cpu}


def run_threads(n_ch, work):
    """Python threads over a C batch function that releases the GIL: as many as the container may really run."""
    try:
        from oracle import ref
        usable = ref.usable_cores()[0]
    except Exception:
        usable = os.cpu_count() or 1
    cores = max(1, min(usable, n_ch))
    bounds = np.linspace(0, n_ch, cores + 1).astype(int)
    th = [threading.Thread(target=work, args=(int(bounds[i]), int(bounds[i + 1]))) for i in range(cores)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    return cores, time.perf_counter() - t0


def as_shipped(measure_both, threads):
    """The same measurement on the reference as the library ships it (oracle/Makefile's fast flavour: -O2 -ffast-math -msse2,
    SPANDSP_USE_SSE2 -- configure.ac:276,346,374-375,509-519), beside the strict build the parity tests use.  Timing only."""
    import oracle
    from oracle import ref
    if not oracle.have_ref_fast():
        return None
    with ref.flavour("fast"):
        all_rate, one_rate = measure_both()
    return {"kind": "reference-fastmath", "value": all_rate/1e6, "single_core": one_rate/1e6, "unit": "Msamples/s", "cores": threads,
            "build": "gcc -std=gnu99 -O2 -ffast-math -msse2 -DSPANDSP_USE_SSE2; never used for parity"}


def ref_baseline(what, kind_id, new_state, free_state, frames_host, seconds=1.0):
    """The reference receiver on the host over a bounded sample of the same frames, driven by the pthread driver of
    oracle/ref_glue/ref_glue_mt.c (every thread loops over its slice of channel objects, frames and passes inside C):
    all usable cores (the headline value) and one core."""
    import oracle
    from oracle import ref
    assert oracle.have_ref(), "the cpu baseline of this path needs oracle/_ref"
    n_frames, n_ch, _ = frames_host.shape
    usable, note = ref.usable_cores()
    threads = max(1, min(usable, n_ch))

    def measure(ch, th, secs):
        states = [new_state(c) for c in range(ch)]
        sub = np.ascontiguousarray(frames_host[:, :ch])
        rate, loops, dt = ref.timed_baseline(lambda l: ref.mt_rx(kind_id, states, sub, l, th), float(sub.size), secs)
        if free_state is not None:
            for st in states:
                free_state(st)
        return rate, loops, dt
    all_rate, all_loops, all_dt = measure(n_ch, threads, seconds)
    one_ch = min(n_ch, 64)
    one_rate, one_loops, one_dt = measure(one_ch, 1, seconds)
    shipped = as_shipped(lambda: (measure(n_ch, threads, 0.6*seconds)[0], measure(one_ch, 1, 0.6*seconds)[0]), threads)
    return {"value": all_rate/1e6, "unit": "Msamples/s", "cores": threads, "kind": "reference", "single_core": one_rate/1e6,
            "host_cores": note, "as_shipped": shipped,
            "sample": "reference (oracle/_ref) %s, pthread driver: %d channels x %d frames x %d passes on %d threads in %.2f s; "
                      "one core: %d channels x %d passes in %.2f s" % (what, n_ch, n_frames, all_loops, threads, all_dt,
                                                                        one_ch, one_loops, one_dt)}


def cpu_modem(kind, bit_rate, frames_host, cutoffs=None):
    """The reference receiver (oracle/_ref) on the host cores over a bounded sample of the same frames."""
    from oracle import ref
    L = ref.lib()
    new = getattr(L, "glue_%s_rx_new_quiet" % kind)
    new.restype = ctypes.c_void_p
    new.argtypes = [ctypes.c_int]
    kind_id = {"v29": ref.MT_V29, "v27ter": ref.MT_V27TER, "v17": ref.MT_V17}[kind]

    def make(c):
        p = new(bit_rate)
        if cutoffs is not None:
            getattr(L, "%s_rx_set_signal_cutoff" % kind)(ctypes.c_void_p(p), ctypes.c_float(float(cutoffs[c])))
        return p
    return ref_baseline("%s_rx()" % kind, kind_id, make, None, frames_host)


# ---- echo canceller -----------------------------------------------------------------------------------
ECHO_TAPS = 128
ECHO_MODE = 0x01                        # ECHO_CAN_USE_ADAPTION (spandsp/echo.h:120-131), as SURVEY 8(d) config 5


def g168_models():
    """The eight echo path models of ITU-T G.168 as the reference's test program uses them (src/spandsp/g168models.h,
    tests/echo_tests.c:396-446): taps and gain constants, from the committed fixture (tests/golden/g168_models.npz)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "g168_models.npz"))
    return [(g["taps_%d" % m].astype(np.float32), float(g["ki"][i])) for i, m in enumerate(g["models"])]


def synth_echo(n_ch, n_frames, dev, seed):
    """SURVEY 8(d)-5's lines, made on the device: tx = white noise at -15 dBm0; rx = tx through the G.168 echo path model
    D(2 + c mod 8) at an ERL drawn from 6 ... 24 dB, the way the reference's test program simulates a line
    (tests/echo_tests.c:443,487: echo = fir32(model, tx*gain), gain = 32768*10^(-ERL/20)*ki); near end silent, except that
    every tenth line has talk bursts (0.5 s in every 1.5 s).  The frames are one continuous signal of n_frames*160
    samples -- no loop, no seam.  (The FIR runs in binary32 here: its sums are exact to ~1e-7 of an LSB of the echo.)"""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    n = n_frames*FRAME
    tx = torch.empty(n_frames, n_ch, FRAME, dtype=torch.int16, device=dev)
    rx = torch.empty_like(tx)
    models = g168_models()
    sigma = 32768.0*10.0**((-15.0 - 3.14)/20.0)
    erl_db = torch.empty(n_ch, device=dev).uniform_(6.0, 24.0, generator=gen)
    ki = torch.tensor([m[1] for m in models], device=dev)
    chunk = 4096                                            # lines per synthesis chunk
    for c0 in range(0, n_ch, chunk):
        c1 = min(n_ch, c0 + chunk)
        idx = torch.arange(c0, c1, device=dev) % 8
        t = torch.clamp(torch.round(sigma*torch.randn(c1 - c0, n, device=dev, generator=gen)), -32768, 32767)
        gain = (32768.0*torch.pow(10.0, -erl_db[c0:c1]/20.0)*ki[idx]).unsqueeze(1)
        s = torch.trunc(t*gain)
        r = torch.empty(c1 - c0, n, device=dev)
        for k in range(8):
            m = (idx == k).nonzero().squeeze(1)
            if m.numel():
                w = torch.tensor(models[k][0], device=dev).flip(0).view(1, 1, -1)
                x = torch.nn.functional.pad(s[m].unsqueeze(1), (w.shape[2] - 1, 0))
                r[m] = torch.floor(torch.nn.functional.conv1d(x, w).squeeze(1)/32768.0)
        talk = (torch.arange(c0, c1, device=dev) % 10 == 0).unsqueeze(1)
        burst = ((torch.arange(n, device=dev)//4000) % 3 == 1).unsqueeze(0)
        r = r + torch.where(talk & burst, torch.round(3000.0*torch.randn(c1 - c0, n, device=dev, generator=gen)), torch.zeros((), device=dev))
        tx[:, c0:c1] = t.to(torch.int16).view(c1 - c0, n_frames, FRAME).permute(1, 0, 2)
        rx[:, c0:c1] = torch.clamp(r, -32768, 32767).to(torch.int16).view(c1 - c0, n_frames, FRAME).permute(1, 0, 2)
        del t, s, r
    return tx, rx


def cpu_echo(tx_host, rx_host, seconds=1.0):
    """echo_can_update() of the reference (oracle/_ref) on the host: all usable cores and one core (pthread driver)."""
    import oracle
    from oracle import ref
    assert oracle.have_ref(), "cpu baseline for the echo path needs oracle/_ref"
    n_frames, n_ch, _ = tx_host.shape
    usable, note = ref.usable_cores()
    threads = max(1, min(usable, n_ch))

    def measure(ch, th, secs):
        # every canceller is made and run through the sample once before the clock starts (its four heap blocks are touched, the
        # taps have begun to adapt: the first pass over freshly allocated cancellers measures page faults), then at least three
        # timed passes
        cans = [ref.EchoCan(ECHO_TAPS, ECHO_MODE) for _ in range(ch)]
        tx = np.ascontiguousarray(tx_host[:, :ch])
        rx = np.ascontiguousarray(rx_host[:, :ch])
        ptrs = [c.p for c in cans]
        warm = ref.mt_echo(ptrs, tx, rx, 1, th)[0]
        loops = max(3, int(np.ceil(secs/max(warm, 1e-6))))
        dt = ref.mt_echo(ptrs, tx, rx, loops, th)[0]
        return float(tx.size)*loops/dt, loops, dt
    def measure_processes(ch, procs, secs):
        """The same on `procs` PROCESSES (forked here; each makes, warms and runs its slice of the cancellers on one thread; a
        barrier starts the timed passes together; the slowest one's time counts).  echo.c counts samples in a global
        (`int sample_no`, echo.c:374, incremented by every echo_can_update(), :429): the threads of one process all write that
        one cache line on every sample and do not scale (16 threads: 3.8 x one core on the GPU box, 1.04 x on 8 threads of the
        build container) -- a process per core is how the reference reaches the host's cores."""
        import multiprocessing as mp
        ctx = mp.get_context("fork")
        gate = ctx.Barrier(procs + 1)
        q = ctx.SimpleQueue()           # (put() writes the pipe itself: nothing is left in a feeder thread when the child _exits)
        bounds = np.linspace(0, ch, procs + 1).astype(int)

        def work(k):
            try:
                lo, hi = int(bounds[k]), int(bounds[k + 1])
                cans = [ref.EchoCan(ECHO_TAPS, ECHO_MODE) for _ in range(hi - lo)]
                tx = np.ascontiguousarray(tx_host[:, lo:hi])
                rx = np.ascontiguousarray(rx_host[:, lo:hi])
                ptrs = [c.p for c in cans]
                warm = ref.mt_echo(ptrs, tx, rx, 1, 1)[0]
                loops = max(3, int(np.ceil(secs/max(warm, 1e-6))))
                gate.wait()
                dt = ref.mt_echo(ptrs, tx, rx, loops, 1)[0]
                q.put((float(tx.size)*loops, dt, loops))
            except Exception as e:                  # pragma: no cover
                q.put((0.0, 1.0, repr(e)))
            finally:
                os._exit(0)                         # (a forked copy of a process that holds a HIP context: leave without its atexit work)
        ps = [ctx.Process(target=work, args=(k,)) for k in range(procs)]
        [p_.start() for p_ in ps]
        gate.wait()
        res = [q.get() for _ in ps]
        [p_.join() for p_ in ps]
        return sum(r[0] for r in res)/max(r[1] for r in res), min(r[2] for r in res if isinstance(r[2], int)), max(r[1] for r in res)
    # the one-core figure on the slice one thread of the all-core run has (the same working set per core)
    one_ch = max(1, n_ch//threads)
    thr_rate, thr_loops, thr_dt = measure(n_ch, threads, seconds)
    one_rate, one_loops, one_dt = measure(one_ch, 1, seconds)
    all_rate, all_loops, all_dt = measure_processes(n_ch, threads, seconds)
    shipped = as_shipped(lambda: (measure_processes(n_ch, threads, 0.6*seconds)[0], measure(one_ch, 1, 0.6*seconds)[0]), threads)
    eff = all_rate/(threads*one_rate)
    out = {"value": all_rate/1e6, "unit": "Msamples/s", "cores": threads, "kind": "reference", "single_core": one_rate/1e6,
           "scaling_efficiency": eff, "parallelism": "%d processes, one thread each" % threads,
           "threads_of_one_process": {"value": thr_rate/1e6, "threads": threads, "scaling_efficiency": thr_rate/(threads*one_rate),
                                      "why": "echo.c:374,429: every echo_can_update() increments the global sample_no -- one cache line written by all threads on every sample"},
           "host_cores": note, "as_shipped": shipped,
           "sample": "reference (oracle/_ref) echo_can_update() 128 taps, every canceller warmed by one untimed pass: %d channels x %d frames "
                     "x %d passes on %d processes in %.2f s; %d threads of one process: %d passes in %.2f s; one core: %d channels x %d "
                     "passes in %.2f s" % (n_ch, n_frames, all_loops, threads, all_dt, threads, thr_loops, thr_dt, one_ch, one_loops, one_dt)}
    if eff < 0.5:
        # the processes did not get a core each: say what the run was worth
        out["processes"] = threads
        out["cores"] = max(1, int(round(all_rate/one_rate)))
        out["note"] = ("%d processes reached %.1f x the one-core rate: `cores` is that figure, not the process count" % (threads, all_rate/one_rate))
    return out


def copy_alone_ms(n_bytes, direction, dev):
    """The floor of a host path leg on this box: the same bytes between pinned host memory and HBM, nothing else."""
    pinned = torch.empty(max(n_bytes, 16), dtype=torch.uint8).pin_memory()
    onboard = torch.empty(max(n_bytes, 16), dtype=torch.uint8, device=dev)
    a, b = (onboard, pinned) if direction == "h2d" else (pinned, onboard)
    a.copy_(b, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        a.copy_(b, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0)/10*1e3


def e2e_echo(n_ch, tx, rx, dev, ticks=40):
    """SURVEY 8(d)'s second number for the echo path: a tick as a caller with HOST buffers sees it, through the pipelined
    feed (spangpu_echo_feed_*): tx and rx rows down, the clean rows up, three streams, the tick loop in C.  int16 and G.711."""
    from spandsp_amd import engine
    out = {}
    enc = np.load(os.path.join(ROOT, "tests", "golden", "g711_encode.npz"))["ulaw"]
    for law, name in ((0, "int16"), (2, "ulaw")):
        bank = engine.EchoBank(n_ch, ECHO_TAPS, ECHO_MODE)
        feed = engine.EchoFeed(bank, FRAME, law=law, depth=3)
        for k in range(3):
            a, b = feed.slots()
            t = tx[k].cpu().numpy()
            r = rx[k].cpu().numpy()
            a[:, :FRAME] = enc[t.astype(np.int32) + 32768] if law else t
            b[:, :FRAME] = enc[r.astype(np.int32) + 32768] if law else r
            feed.commit(FRAME)
        while feed.outstanding():
            feed.collect()
        ms = feed.run(FRAME, ticks, 1)
        ms2 = feed.run(FRAME, ticks, 2)
        bps = 1 if law else 2
        down, up = 2*n_ch*feed.stride*bps, n_ch*feed.stride*bps
        h2d, d2h = copy_alone_ms(down, "h2d", dev), copy_alone_ms(up, "d2h", dev)
        ms, ms1 = ms2, ms                           # two ticks of latency keep all three legs busy: the headline of this path
        out[name] = {"ms_per_step": ms/ticks, "ms_per_step_one_tick_of_latency": ms1/ticks, "h2d_copy_alone_ms": h2d, "d2h_copy_alone_ms": d2h, "bytes_down": down, "bytes_up": up,
                     "over_the_larger_copy": ms/ticks/max(h2d, d2h), "value": n_ch*FRAME/(ms/ticks*1e-3)/1e6, "unit": "Msamples/s"}
        feed.close()
        bank.close()
    out["includes"] = ("per step (tick loop in C, spangpu_echo_feed_run()), pipelined over three pinned slots and three streams: H2D of the tx and rx rows, "
                       "the canceller kernel (with G.711: decode and encode kernels around it), D2H of the clean rows; one tick of latency")
    return out


def e2e_modem(kind, bit_rate, n_ch, frames, dev, ticks=40):
    """The same for a modem receiver bank (spangpu_modem_feed_*): PCM rows down, the put_bit stream up in packed form (a
    header word and the data bits per channel + one status list per bank) instead of a byte per put_bit call."""
    from spandsp_amd import engine
    bank = engine.ModemBank(kind, n_ch, bit_rate)
    feed = engine.ModemFeed(bank, FRAME, bit_rate, depth=3)
    nf = frames.shape[0]
    host = [frames[k].cpu().numpy() for k in range(min(nf, 3))]
    for k in range(3):
        feed.slot()[:, :FRAME] = host[k % len(host)]
        feed.commit(FRAME)
    while feed.outstanding():
        feed.collect(raw=True)
    ms, bits = feed.run(FRAME, ticks, 1)
    down = n_ch*feed.stride*2
    up = (n_ch*feed.wpc + 1 + 2*feed.status_cap)*4
    h2d, d2h = copy_alone_ms(down, "h2d", dev), copy_alone_ms(up, "d2h", dev)
    out = {"ms_per_step": ms/ticks, "h2d_copy_alone_ms": h2d, "d2h_copy_alone_ms": d2h, "bytes_down": down, "bytes_up": up,
           "bytes_up_per_channel": feed.wpc*4, "bytes_up_as_one_byte_per_call": n_ch*(FRAME*4 + 64),
           "over_the_larger_copy": ms/ticks/max(h2d, d2h), "value": n_ch*FRAME/(ms/ticks*1e-3)/1e6, "unit": "Msamples/s",
           "includes": "per step (tick loop in C, spangpu_modem_feed_run()), pipelined over three pinned slots and three streams: H2D of the PCM rows, the "
                       "receiver kernel, the pack kernel, D2H of the packed put_bit stream; one tick of latency"}
    feed.close()
    bank.close()
    return out


def echo_spot_check(tx64, rx64, clean64):
    """The oracle's echo_can_update() (oracle/echo_oracle.c, pinned to the reference) over the same first frames of 64 lines:
    every clean sample equal?  tx64 / rx64 / clean64: int16 [frames, 64, FRAME] on the host."""
    from oracle import restated as orc
    nfr, v, _ = tx64.shape
    ok = True
    for c in range(v):
        d = orc.EchoCan(ECHO_TAPS, ECHO_MODE)
        want = d.run(np.ascontiguousarray(tx64[:, c]).reshape(-1), np.ascontiguousarray(rx64[:, c]).reshape(-1), False)
        ok = ok and np.array_equal(want, np.ascontiguousarray(clean64[:, c]).reshape(-1))
    return {"channels": v, "frames": nfr, "checked": "every clean sample of the first %d frames of lines 0..%d against the oracle's echo_can_update()" % (nfr, v - 1),
            "bit_exact": bool(ok)}


def bench_echo(args, dev, stream):
    """BASELINE configs[4], one GPU's shard, on SURVEY 8(d)-5's workload: `seconds` of continuous signal on G.168 lines
    (synth_echo); the first second warms up (and is checked against the oracle on 64 lines), the rest is timed; the ERLE of
    every line over the last second comes from the bank's own sums."""
    from spandsp_amd import engine
    n_ch = args.channels or 131072
    seconds = getattr(args, "echo_seconds", 10)
    nf = seconds*50
    tx, rx = synth_echo(n_ch, nf, dev, seed=0xEC40)
    clean = torch.empty(n_ch, FRAME, dtype=torch.int16, device=dev)
    if args.echo_lanes:
        engine.lib().spangpu_tune_echo_lanes_per_channel(args.echo_lanes)
    bank = engine.EchoBank(n_ch, ECHO_TAPS, ECHO_MODE)
    engine.lib().spangpu_tune_echo_lanes_per_channel(0)
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    bank.stats(2)                       # energy sums by the update kernel itself
    fb = n_ch*FRAME*2
    warm = 50
    steps = nf - warm
    v = min(64, n_ch)
    kept = []

    def step(k):
        bank.update_device(ctypes.c_void_p(tx.data_ptr() + k*fb), ctypes.c_void_p(rx.data_ptr() + k*fb),
                           ctypes.c_void_p(clean.data_ptr()), FRAME, FRAME)
    for k in range(warm):
        step(k)
        if k < 25:
            kept.append(clean[:v].clone())
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for k in range(warm, nf):
        if k == nf - 50:
            bank.stats_reset(sums=True, crc=False)
        step(k)
    ev1.record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    avg_ms = ev0.elapsed_time(ev1)/steps
    erle = torch.zeros(n_ch, dtype=torch.float32, device=dev)
    bank.erle_device(ctypes.c_void_p(erle.data_ptr()))
    torch.cuda.synchronize()
    quiet = torch.arange(n_ch, device=dev) % 10 != 0
    es = erle[quiet]
    # per channel and frame read 640 B (tx, rx) + 1 300 B of state (taps32 512, the ACTIVE taps16 set 256, history
    # 256, scalars 276), write 320 B (clean) + 1 300 B -- the contract's algorithmic bytes.  What this implementation keeps per
    # channel: 48 control words (192 B) + taps32 + history + FOUR taps16 sets, of which a launch reads and writes the active one
    # (echo_dev.hpp: the prologue / write-back; another set only at a set event).
    alg_read = n_ch*(2*FRAME*2 + 1300)
    alg_write = n_ch*(FRAME*2 + 1300)
    actual_read = n_ch*(2*FRAME*2 + 48*4 + ECHO_TAPS*4 + ECHO_TAPS*2 + ECHO_TAPS*2)
    state_resident = n_ch*(48*4 + ECHO_TAPS*4 + 4*ECHO_TAPS*2 + ECHO_TAPS*2)
    cpu = None
    if not args.no_cpu_baseline:
        nc = min(args.cpu_channels, n_ch, 4096)
        cpu = cpu_echo(tx[:60, :nc].contiguous().cpu().numpy(), rx[:60, :nc].contiguous().cpu().numpy(), getattr(args, "cpu_seconds", 1.0))
        cpu["spot_check"] = echo_spot_check(tx[:25, :v].cpu().numpy(), rx[:25, :v].cpu().numpy(), torch.stack(kept).cpu().numpy())
    value = steps*n_ch*FRAME/dt/1e6
    lanes = engine.lib().spangpu_echo_lanes_per_channel(bank.h)
    e2e = None if getattr(args, "no_e2e", False) else e2e_echo(n_ch, tx, rx, dev)
    return {
        "e2e": e2e,
        "metric": "Msamples/s of batched G.168 echo cancellation, 128 taps (8 kHz channels at real-time = value*1e6/8000)",
        "value": value, "unit": "Msamples/s", "realtime_channels": value*1e6/8000.0, "n_gpus": 1,
        "steps": steps, "warmup": warm, "ms_per_step": dt*1e3/steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4] (one GPU's shard), SURVEY 8(d)-5's lines: echo_can_update 128 taps, mode "
                               "ECHO_CAN_USE_ADAPTION, %d channels x %d-sample frames, %d s of continuous signal: white noise at "
                               "-15 dBm0 through G.168 echo path models D2..D9 (by channel mod 8), ERL 6..24 dB, every tenth line "
                               "with near end talk bursts" % (n_ch, FRAME, seconds), "channels_per_gpu": n_ch,
                   "erle_db_last_second_single_talk_lines": {"median": float(es.median()), "p10": float(es.quantile(0.1)),
                                                             "p90": float(es.quantile(0.9))}},
        "roofline": {"bound": "hbm", "kernel": "echo canceller kernel, %d lanes per channel (echo_%s), 128 taps" % (lanes, "pair_kernel" if lanes == 2 else "bank_kernel"),
                     "achieved": alg_read/(avg_ms*1e-3)/1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_read/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS, "traffic": None,
                     "alg_read_bytes_per_launch": alg_read, "alg_write_bytes_per_launch": alg_write,
                     "state_bytes_actual_read_per_launch": actual_read, "state_bytes_resident": state_resident,
                     "avg_launch_us": avg_ms*1e3,
                     "note": "integer-VALU bound (2 x 128 MACs per sample per channel); the HBM figure is reported, not targeted"},
        "cpu_baseline": cpu}


MIXED_ST_PLAN = ([(400, 0, 700, 0)], [(1100, 0, 400, 600), (0, 0, 2800, 3200)], [(350, 440, 400, 0)],
                 [(480, 620, 450, 550), (0, 0, 450, 550)], [(950, 0, 300, 0)], [(1400, 0, 300, 0)])


# the lines the super-tone third of configs[2] listens to: cycles of (f1, f2, ms) that follow MIXED_ST_PLAN's cadences (and one
# that is in nobody's plan), tests/synth.py cadence_plan_channels
MIXED_ST_LINES = ([(400, 0, 1500)], [(1100, 0, 500), (0, 0, 3000)], [(350, 440, 1200), (0, 0, 300)], [(480, 620, 500), (0, 0, 500)],
                  [(950, 0, 330), (1400, 0, 330), (0, 0, 1000)], [(620, 0, 300), (0, 0, 200)])
MIXED_ST_FREQS = [400, 1100, 350, 440, 480, 620, 950, 1400]          # in the order MIXED_ST_PLAN's descriptor monitors them


def mixed_st_cadences():
    """MIXED_ST_PLAN as spangpu_bank_set_cadences() takes it: bins in place of frequencies (-1 = silence)"""
    bins = {0: -1}
    bins.update({f: i for i, f in enumerate(MIXED_ST_FREQS)})
    return [[(bins[f1], bins[f2], lo, hi) for f1, f2, lo, hi in t] for t in MIXED_ST_PLAN]


def mixed_spot_check(srcs, kept, kept_cad):
    """The oracle's bell_mf_rx() / r2_mf_rx() / super_tone_rx() block decisions (oracle/tone_oracle.c, pinned to the
    reference) over the first frames of 64 channels of each bank against the records the launch wrote, and -- the
    super-tone third -- every tone report and segment report super_tone_rx() makes (its cadence matcher,
    super_tone_rx.c:164-228, :369-445) against the events the launch's epilogue wrote."""
    from oracle import restated as orc
    desc = orc.SuperToneDesc()
    for tone in MIXED_ST_PLAN:
        t = desc.add_tone()
        for f1, f2, lo, hi in tone:
            desc.add_element(t, f1, f2, lo, hi)
    ok = True
    blocks = 0
    reports = segments = 0
    for kind in range(3):
        v, n = srcs[kind].shape
        for c in range(v):
            o = orc.BellMf(0) if kind == 0 else orc.R2Mf(True, True) if kind == 1 else orc.SuperTone(desc, True)
            want = []
            for k in range(n//FRAME):
                want.extend(o.rx(srcs[kind][c, k*FRAME:(k + 1)*FRAME]))
                if kind == 2:
                    ev = [tuple(int(x) for x in e) for e in o.sink.events()]
                    o.sink.clear()
                    ok = ok and ev == kept_cad[k][c]
                    reports += sum(1 for e in ev if e[0] == 1)
                    segments += sum(1 for e in ev if e[0] == 4)
            got = [(int(r["hit"]), int(r["code"])) for b in kept[kind] for r in b[b["channel"] == c]]
            ok = ok and got == [(int(x["hit"]), int(x["aux"])) for x in want]
            blocks += len(want)
    return {"channels": 3*srcs[0].shape[0], "frames": len(kept[0]), "checked": "hit and code of every block (%d) of the first frames of 64 channels of each bank "
            "against the oracle's bell_mf_rx / r2_mf_rx / super_tone_rx, and the super-tone channels' %d tone reports and %d segment reports against "
            "super_tone_rx()'s callbacks" % (blocks, reports, segments), "tone_reports": reports, "segment_reports": segments,
            "bit_exact": bool(ok and reports > 0 and segments > 0)}


# ---- mixed Goertzel banks ------------------------------------------------------------------------------
def bench_mixed(args, dev, stream):
    """BASELINE configs[2]: one third Bell MF, one third R2 MF (forward), one third super_tone_rx() -- 8 monitored frequencies
    AND its cadence matcher (MIXED_ST_PLAN: the descriptor of tests/super_tone_rx_tests.c:361-374 plus four more tones), tone and
    segment reports on, matched on the device in the detector launch; a step = the three launches of one 20 ms tick."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    from spandsp_amd import engine
    n_ch = args.channels or 131072
    n_each = [n_ch//3, n_ch//3, n_ch - 2*(n_ch//3)]
    nf = 200                                                # 4 s of line: long enough for the ring-back cadence (0.5 s on, 3 s off)
    n_src = 512                                             # distinct source channels per kind, tiled with a frame rotation
    srcs = [synth.bell_mf_channels(n_src, nf*FRAME, 21)[0], synth.r2_mf_channels(n_src, nf*FRAME, 22, True)[0],
            synth.cadence_plan_channels(n_src, nf*FRAME, 23, MIXED_ST_LINES)]
    frames = []
    for kind in range(3):
        src = torch.tensor(srcs[kind], device=dev).view(n_src, nf, FRAME)
        idx = torch.arange(n_each[kind], device=dev)
        rot = (idx//n_src) % nf
        fsel = (torch.arange(nf, device=dev).unsqueeze(0) + rot.unsqueeze(1)) % nf          # [ch, frame]
        f = src[(idx % n_src).unsqueeze(1), fsel]                                            # [ch, frame, FRAME]
        frames.append(f.permute(1, 0, 2).contiguous())
    st_freqs = [float(f) for f in MIXED_ST_FREQS]
    fac = [engine.goertzel_fac(f) for f in st_freqs]
    banks = [engine.ToneBank(engine.BELL_MF, n_each[0]), engine.ToneBank(engine.R2_MF, n_each[1], r2_fwd=True),
             engine.ToneBank(engine.SUPER_TONE, n_each[2], bin_fac=fac)]
    with_cadences = not getattr(args, "no_cadences", False)
    if with_cadences:
        banks[2].set_cadences(mixed_st_cadences(), want_segments=True)
    # Three ways to run a tick (results identical: tests/test_tone_gpu.py): ONE launch for the three banks on one stream
    # (tone_multi_fast_kernel); a launch per bank, every bank free-running on a stream (= hardware queue) of its own -- one
    # bank's launch boundary, start burst and write-back under the other banks' steady state; the banks share nothing and the
    # host joins the streams when it reads records -- which is how the step is timed unless --one-launch is given; and, for
    # the record, a launch per bank on one stream (--separate-launches).  spangpu_banks_rx() does the first or the second
    # according to the streams the banks were given.
    mode = "separate" if args.separate_launches else "one_launch" if getattr(args, "one_launch", False) else "bank_streams" if getattr(args, "three_queues", False) else "two_queues"
    own = None                                              # the banks' own streams as torch sees them, once they have them
    plan = engine.BanksPlan(banks)
    handles = [plan.frame([frames[kind].data_ptr() + f*n_each[kind]*FRAME*2 for kind in range(3)]) for f in range(nf)]
    addr = [[ctypes.c_void_p(frames[kind].data_ptr() + f*n_each[kind]*FRAME*2) for kind in range(3)] for f in range(nf)]

    def set_mode(m):
        nonlocal own
        torch.cuda.synchronize()
        if m == "bank_streams":
            # a stream per bank on hardware queues proven to be different ones (spangpu_banks_own_queues; three torch streams
            # may or may not share a queue -- profiles/r6_mixed_trace_overlap_collision.txt)
            set_mode.proven = engine.banks_own_queues(banks)
            own = [torch.cuda.ExternalStream(engine.lib().spangpu_bank_get_stream(b.h), device=dev) for b in banks]
        elif m == "two_queues":
            # the super-tone bank, whose launch carries the cadence matcher and is the longest, alone on one queue; Bell MF and
            # R2 MF share the other and with it a launch (spangpu_banks_rx groups banks by stream)
            set_mode.proven = engine.banks_own_queues([banks[0], banks[2]])
            banks[1].share_stream(banks[0])
            own = [torch.cuda.ExternalStream(engine.lib().spangpu_bank_get_stream(b.h), device=dev) for b in (banks[0], banks[2])]
        else:
            for b in banks:
                b.set_stream(ctypes.c_void_p(stream.cuda_stream))

    set_mode.proven = None

    def step(i, m):
        if m == "separate":
            for kind in range(3):
                banks[kind].rx_device(addr[i % nf][kind], FRAME, FRAME)
        else:
            plan.rx(handles[i % nf], FRAME)

    def timed(m, n_steps):
        """(wall seconds, milliseconds by events on `stream` with the banks' own streams forked from and joined into it)"""
        set_mode(m)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        if m in ("bank_streams", "two_queues"):
            for s in own:
                s.wait_event(ev0)
        for i in range(n_steps):
            step(timed.pos + i, m)
        if m in ("bank_streams", "two_queues"):
            for s in own:
                e = torch.cuda.Event()
                e.record(s)
                stream.wait_event(e)
        ev1.record(stream)
        torch.cuda.synchronize()
        timed.pos += n_steps
        return time.perf_counter() - t0, ev0.elapsed_time(ev1)
    timed.pos = 0

    set_mode(mode)
    kept = [[] for _ in range(3)]
    kept_cad = []
    v = 64
    n_check = 190                                           # frames of the spot check: the whole source, bar the frames where the tiling wraps
    for i in range(max(args.warmup, 0 if args.no_cpu_baseline else n_check)):
        step(i, mode)
        if i < n_check and not args.no_cpu_baseline:
            for kind in range(3):
                b = banks[kind].blocks()
                kept[kind].append(b[b["channel"] < v].copy())
            if with_cadences:
                kept_cad.append(banks[2].cadence_events()[:v])
    timed.pos = max(args.warmup, 0 if args.no_cpu_baseline else n_check)
    torch.cuda.synchronize()
    # the step: HIP events around the whole timed region / steps in it
    reps = max(1, int(np.ceil(2000/args.steps)))
    one_launch_us = None
    other_us = None
    if mode in ("bank_streams", "two_queues") and not getattr(args, "single_mode", False):
        other = "bank_streams" if mode == "two_queues" else "two_queues"
        timed(other, 200)
        other_us = timed(other, args.steps*reps)[1]/(args.steps*reps)*1e3
        timed("one_launch", 200)
        one_launch_us = timed("one_launch", args.steps*reps)[1]/(args.steps*reps)*1e3     # the per-launch figure of the one-launch form
        timed(mode, 200)
    dt, ms = timed(mode, args.steps*reps)
    dt /= reps
    avg_ms = ms/(args.steps*reps)
    fused = mode == "one_launch"
    hits = [int((b.blocks()["hit"] != 0).sum()) for b in banks]
    # Bell MF / R2 MF read 320 + 64 B per channel and frame; super-tone with M bins 320 + (8M + 160) B, the 160 B
    # being the cadence matcher's state -- owed only when the matcher runs in the timed region
    alg_read = n_each[0]*(320 + 64) + n_each[1]*(320 + 64) + n_each[2]*(320 + 8*8 + (160 if with_cadences else 0))
    value = args.steps*n_ch*FRAME/dt/1e6
    cpu = None
    if not args.no_cpu_baseline:
        # the reference on the host, the three detector kinds one after the other on a third of the sample each
        from oracle import ref
        L = ref.lib()
        n_cpu = min(args.cpu_channels, n_ch)//3
        desc = ref.SuperToneDesc()
        for tone in MIXED_ST_PLAN:
            t = desc.add_tone()
            for f1, f2, lo, hi in tone:
                desc.add_element(t, f1, f2, lo, hi)
        parts = [
            ref_baseline("bell_mf_rx()", ref.MT_BELL_MF, lambda c: L.glue_bell_mf_rx_new(None, 0), None,
                         frames[0][:50, :n_cpu].contiguous().cpu().numpy(), 0.7),
            ref_baseline("r2_mf_rx()", ref.MT_R2_MF, lambda c: L.glue_r2_mf_rx_new(None, 1, 0), None,
                         frames[1][:50, :n_cpu].contiguous().cpu().numpy(), 0.7),
            ref_baseline("super_tone_rx() with segment reports", ref.MT_SUPER_TONE, lambda c: L.glue_super_tone_rx_new(desc.p, L.glue_sink_new(), 1), None,
                         frames[2][:50, :n_cpu].contiguous().cpu().numpy(), 0.7),
        ]
        w = [n_each[k]/float(n_ch) for k in range(3)]
        cpu = {"value": 1.0/sum(w[k]/parts[k]["value"] for k in range(3)), "unit": "Msamples/s", "cores": parts[0]["cores"],
               "kind": "reference", "single_core": 1.0/sum(w[k]/parts[k]["single_core"] for k in range(3)),
               "host_cores": parts[0]["host_cores"],
               "sample": "channel-weighted harmonic mean of: " + " | ".join(p["sample"] for p in parts)}
        if all(p.get("as_shipped") for p in parts):
            cpu["as_shipped"] = dict(parts[0]["as_shipped"],
                                     value=1.0/sum(w[k]/parts[k]["as_shipped"]["value"] for k in range(3)),
                                     single_core=1.0/sum(w[k]/parts[k]["as_shipped"]["single_core"] for k in range(3)))
        if kept[0]:
            cpu["spot_check"] = mixed_spot_check([srcs[k][:v, :len(kept[k])*FRAME] for k in range(3)], kept, kept_cad)
    return {
        "metric": "Msamples/s of mixed Bell MF + R2 MF + super-tone Goertzel banks (8 kHz channels at real-time = value*1e6/8000)",
        "value": value, "unit": "Msamples/s", "realtime_channels": value*1e6/8000.0, "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt*1e3/args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: %d Bell MF + %d R2 MF + %d super_tone_rx (8 monitored frequencies, %s) channels x %d-sample "
                               "frames, %s" % (n_each[0], n_each[1], n_each[2],
                                                "its 6-tone cadence plan matched in the launch, tone and segment reports on" if with_cadences else "block decisions only: NO cadence matcher",
                                                FRAME,
                                                {"one_launch": "one launch per step (spangpu_banks_rx, the banks on one stream)",
                                                 "bank_streams": "a launch per bank and step, every bank on a stream (hardware queue) of its own (spangpu_banks_own_queues, spangpu_banks_rx)",
                                                 "two_queues": "two launches per step on two hardware queues: the super-tone bank with its cadence matcher on one, Bell MF + R2 MF sharing a launch on the other (spangpu_banks_own_queues, spangpu_banks_rx)",
                                                 "separate": "three launches per step on one stream"}[mode]),
                   "channels_per_gpu": n_ch, "blocks_with_a_hit_in_last_step": hits, "cadence_matcher_in_timed_region": with_cadences,
                   "hardware_queues_proven_distinct": set_mode.proven},
        "roofline": {"bound": "hbm", "kernel": "tone_multi_fast_kernel (Bell MF + R2 MF + super-tone workgroups in one launch)" if fused
                               else "tone_fast_kernel<MultiDet<8, true> + cadence epilogue> beside tone_multi_fast_kernel (Bell MF + R2 MF), two queues" if mode == "two_queues"
                               else "tone_fast_kernel<BellMfDet | R2MfDet | MultiDet<8, true> + cadence epilogue> (3 launches%s)" % (" on 3 streams" if mode == "bank_streams" else ""),
                     "one_launch_us": one_launch_us, ("three_queues_us" if mode == "two_queues" else "two_queues_us"): other_us,
                     "achieved": alg_read/(avg_ms*1e-3)/1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg_read/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS, "traffic": None,
                     "alg_read_bytes_per_launch": alg_read, "avg_launch_us": avg_ms*1e3,
                     "note": "avg_launch_us is one whole step (all three banks): events around the timed region / steps in it"},
        "cpu_baseline": cpu}


def bench_supertone(args, dev, stream):
    """A bank of super-tone detectors (a nine-frequency call-progress plan of six tones) with the cadences matched on the
    device: a step = the launch of one 20 ms tick (detector + cadence epilogue); the detector alone and the tick with the
    event list brought to the host are timed beside it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    import test_cadence_gpu as tc
    from spandsp_amd import engine
    from oracle import ref
    n_ch = args.channels or 65536
    nf = 100
    n_src = 512
    src = torch.tensor(synth.cadence_plan_channels(n_src, nf*FRAME, 81, tc.PLANS), device=dev).view(n_src, nf, FRAME)
    idx = torch.arange(n_ch, device=dev)
    frames = src[idx % n_src].permute(1, 0, 2).contiguous()                 # [frame][channel][sample]
    quiet = float(os.environ.get("SUPERTONE_QUIET", "0"))                   # fraction of the lines that stay silent
    if quiet > 0.0:
        frames[:, (idx % 100) < int(100*quiet), :] = 0
    desc = ref.SuperToneDesc()
    tc.build(desc)
    hz = [400, 1100, 350, 440, 480, 620, 950, 1400, 1800]
    bins = {0: -1}
    bins.update({f: i for i, f in enumerate(hz)})
    tones = [[(bins[f1], bins[f2], lo, hi) for f1, f2, lo, hi in t] for t in tc.TONES]
    bank = engine.ToneBank(engine.SUPER_TONE, n_ch, bin_fac=[engine.goertzel_fac(float(f)) for f in hz])
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    addr = [ctypes.c_void_p(frames.data_ptr() + f*n_ch*FRAME*2) for f in range(nf)]

    def timed(step, steps):
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for i in range(steps):
            step(args.warmup + i)
        ev1.record(stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0)/steps, ev0.elapsed_time(ev1)/steps

    def detect(i):
        bank.rx_device(addr[i % nf], FRAME, FRAME)

    def tick(i):
        bank.rx_device(addr[i % nf], FRAME, FRAME)
        bank.cadence_run()

    n_events = [0]

    def tick_host(i):
        bank.rx_device(addr[i % nf], FRAME, FRAME)
        lst = ctypes.c_void_p()
        n_events[0] += engine.lib().spangpu_bank_cadence_list(bank.h, ctypes.byref(lst))

    reps = max(1, int(np.ceil(2000/args.steps)))
    _, det_ms = timed(detect, args.steps*reps)              # the bank as a plain detector: no cadences given yet
    bank.set_cadences(tones, want_segments=True)
    dt, tick_ms = timed(tick, args.steps*reps)
    dt_host, _ = timed(tick_host, args.steps)
    value = n_ch*FRAME/dt/1e6
    alg_read = n_ch*(320 + 9*8 + 160)
    cpu = None
    if not args.no_cpu_baseline:
        L = ref.lib()
        n_cpu = min(args.cpu_channels, n_ch)
        cpu = ref_baseline("super_tone_rx()", ref.MT_SUPER_TONE, lambda c: L.glue_super_tone_rx_new(desc.p, L.glue_sink_new(), 1), None,
                           frames[:, :n_cpu].contiguous().cpu().numpy(), 1.0)
    return {
        "metric": "Msamples/s of a super-tone bank with its cadences matched on the device (8 kHz channels at real-time = value*1e6/8000)",
        "value": value, "unit": "Msamples/s", "realtime_channels": value*1e6/8000.0, "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt*1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d super-tone channels (9 monitored frequencies, 6 tones of 1-4 elements, segment reports on) x "
                               "%d-sample frames: one launch per step, the cadences matched in the detector kernel's epilogue%s" % (n_ch, FRAME, (", %d %% of the lines silent" % int(100*quiet)) if quiet > 0.0 else ""),
                   "channels_per_gpu": n_ch, "detector_only_us": det_ms*1e3, "detector_and_matcher_us": tick_ms*1e3,
                   "with_event_list_on_the_host_ms": dt_host*1e3, "events_per_tick": n_events[0]/float(args.steps + args.warmup)},
        "roofline": {"bound": "hbm", "kernel": "tone_fast_kernel<MultiDet<12, true>, ..., kToneCadence> (detector + cadence epilogue)", "achieved": alg_read/(tick_ms*1e-3)/1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_read/(tick_ms*1e-3)/1e9/HBM_PEAK_GBPS, "traffic": None,
                     "alg_read_bytes_per_launch": alg_read, "avg_launch_us": tick_ms*1e3,
                     "note": "events around the timed region / steps in it"},
        "cpu_baseline": cpu}


def bench_fsk(args, dev, stream):
    """a V.21 channel 2 receiver bank in synchronous mode (the FAX control channel), inputs in HBM."""
    import synth
    from spandsp_amd import engine
    n_ch = args.channels or 65536
    which = engine.FSK_V21CH2
    sp = engine.fsk_preset(which)
    nf = 50
    n_src = 256
    src_host = synth.fsk_channels(n_src, nf*FRAME, 77, sp.freq_zero, sp.freq_one, sp.baud_rate)
    src = torch.tensor(src_host, device=dev).view(n_src, nf, FRAME)
    idx = torch.arange(n_ch, device=dev)
    fsel = (torch.arange(nf, device=dev).unsqueeze(0) + ((idx//n_src) % nf).unsqueeze(1)) % nf
    frames = src[(idx % n_src).unsqueeze(1), fsel].permute(1, 0, 2).contiguous()         # [frame, ch, FRAME]
    bank = engine.FskBank(which, n_ch, engine.FSK_FRAME_MODE_SYNC)
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    frame_bytes = n_ch*FRAME*2

    def step(i):
        bank.rx_device(ctypes.c_void_p(frames.data_ptr() + (i % nf)*frame_bytes), FRAME, FRAME)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record(stream)
        step(args.warmup + i)
        evs[i][1].record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [a.elapsed_time(b) for a, b in evs]
    avg_ms = sum(per)/len(per)
    ev_last = int(sum(len(e) for e in bank.events()))
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import ref
        L = ref.lib()
        n_cpu = min(args.cpu_channels, n_ch)
        counters = np.zeros(n_cpu*8, np.int64)              # one per receiver, a cache line apart
        cpu = ref_baseline("fsk_rx() V.21 ch 2 sync", ref.MT_FSK,
                           lambda c: L.glue_fsk_rx_new_quiet(which, 1, counters.ctypes.data + c*64), L.fsk_rx_free,
                           frames[:, :n_cpu].contiguous().cpu().numpy())
    words = bank.words
    alg_read = n_ch*(FRAME*2 + words*4)
    alg_write = n_ch*((words - 12)*4 + 4)
    value = args.steps*n_ch*FRAME/dt/1e6
    return {
        "metric": "Msamples/s of batched V.21 FSK receive (8 kHz channels at real-time = value*1e6/8000)", "value": value,
        "unit": "Msamples/s", "realtime_channels": value*1e6/8000.0, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt*1e3/args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": "fsk_rx V.21 ch 2, synchronous, %d channels x %d-sample frames" % (n_ch, FRAME),
                   "channels_per_gpu": n_ch, "events_in_last_frame": ev_last},
        "roofline": {"bound": "hbm", "kernel": "fsk_bank_kernel" if args.fsk_waves == 1 else "fsk_pair_kernel (two waves per 64 receivers)", "achieved": (alg_read + alg_write)/(avg_ms*1e-3)/1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (alg_read + alg_write)/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS,
                     "traffic": None, "alg_read_bytes_per_launch": alg_read, "alg_write_bytes_per_launch": alg_write,
                     "avg_launch_us": avg_ms*1e3, "min_launch_us": min(per)*1e3,
                     "note": "integer state machine, VALU/LDS-issue bound; the HBM figure is reported, not targeted"},
        "cpu_baseline": cpu}


def bench_mct(args, dev, stream):
    """modem connect tone detectors, type FAX_CED_OR_PREAMBLE (the V.21 preamble hunter + the 2100 Hz
    detector), the one a FAX terminal runs at the head of a call."""
    import synth
    from spandsp_amd import engine
    n_ch = args.channels or 65536
    nf = 50
    n_src = 256
    src = torch.tensor(synth.connect_tone_channels(n_src, nf*FRAME, 78, "mix"), device=dev).view(n_src, nf, FRAME)
    idx = torch.arange(n_ch, device=dev)
    fsel = (torch.arange(nf, device=dev).unsqueeze(0) + ((idx//n_src) % nf).unsqueeze(1)) % nf
    frames = src[(idx % n_src).unsqueeze(1), fsel].permute(1, 0, 2).contiguous()
    bank = engine.MctBank(int(os.environ.get("MCT_TYPE", engine.MCT_FAX_CED_OR_PREAMBLE)), n_ch)     # (MCT_TYPE: A-B runs of the other detectors)
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    frame_bytes = n_ch*FRAME*2

    def step(i):
        bank.rx_device(ctypes.c_void_p(frames.data_ptr() + (i % nf)*frame_bytes), FRAME, FRAME)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record(stream)
        step(args.warmup + i)
        evs[i][1].record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [a.elapsed_time(b) for a, b in evs]
    avg_ms = sum(per)/len(per)
    reports = 0
    for i in range(nf):                                  # one more pass over the signal, untimed, counting reports
        step(args.warmup + args.steps + i)
        reports += int(sum(len(e) for e in bank.events()))
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import ref
        L = ref.lib()
        n_cpu = min(args.cpu_channels, n_ch)
        cpu = ref_baseline("modem_connect_tones_rx() FAX_CED_OR_PREAMBLE", ref.MT_MCT,
                           lambda c: L.modem_connect_tones_rx_init(None, 7, None, None), L.modem_connect_tones_rx_free,
                           frames[:, :n_cpu].contiguous().cpu().numpy())
    words = bank.words
    alg_read = n_ch*(FRAME*2 + words*4)
    alg_write = n_ch*((words - 12)*4 + 4)
    value = args.steps*n_ch*FRAME/dt/1e6
    return {
        "metric": "Msamples/s of batched modem connect tone detection (8 kHz channels at real-time = value*1e6/8000)",
        "value": value, "unit": "Msamples/s", "realtime_channels": value*1e6/8000.0, "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt*1e3/args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32+int32", "data": "synthetic",
        "config": {"workload": "modem_connect_tones_rx FAX_CED_OR_PREAMBLE, %d channels x %d-sample frames" % (n_ch, FRAME),
                   "channels_per_gpu": n_ch, "tone_reports_in_one_more_second_of_signal": reports},
        "roofline": {"bound": "hbm", "kernel": "mct_bank_kernel<7>" if args.fsk_waves == 1 else "mct_ced_pair_kernel (V.21 receiver and 2100 Hz detector on two waves)", "achieved": (alg_read + alg_write)/(avg_ms*1e-3)/1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (alg_read + alg_write)/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS,
                     "traffic": None, "alg_read_bytes_per_launch": alg_read, "alg_write_bytes_per_launch": alg_write,
                     "avg_launch_us": avg_ms*1e3, "min_launch_us": min(per)*1e3,
                     "note": "sample-serial state machines (V.21 receiver + biquads), latency bound at one wave per SIMD"},
        "cpu_baseline": cpu}


def bench_sigtone(args, dev, stream):
    """a bank of 2280 Hz signalling tone receivers (sig_tone_rx) in pass-through mode with the notch switched
    in while a tone is about -- every frame is read, filtered and written back."""
    import synth
    from spandsp_amd import engine
    n_ch = args.channels or 65536
    nf = 50
    n_src = 256
    src = torch.tensor(synth.sig_tone_channels(n_src, nf*FRAME, 79, 1), device=dev).view(n_src, nf, FRAME)
    idx = torch.arange(n_ch, device=dev)
    fsel = (torch.arange(nf, device=dev).unsqueeze(0) + ((idx//n_src) % nf).unsqueeze(1)) % nf
    frames = src[(idx % n_src).unsqueeze(1), fsel].permute(1, 0, 2).contiguous()
    work = frames.clone()
    bank = engine.SigToneRxBank(engine.SIG_TONE_2280HZ, n_ch)
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    bank.set_mode(engine.SIG_TONE_RX_PASSTHROUGH)
    frame_bytes = n_ch*FRAME*2

    def step(i):
        bank.rx_device(ctypes.c_void_p(work.data_ptr() + (i % nf)*frame_bytes), FRAME, FRAME)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    work.copy_(frames)                                   # the receivers rewrite their frames: a fresh copy for the timed pass
    torch.cuda.synchronize()
    steps = min(args.steps, nf)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        evs[i][0].record(stream)
        step(i)
        evs[i][1].record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [a.elapsed_time(b) for a, b in evs]
    avg_ms = sum(per)/len(per)
    work.copy_(frames)
    reports = 0
    for i in range(nf):                                  # one more pass over the signal, untimed, counting reports
        step(i)
        reports += int(sum(len(e) for e in bank.events()))
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import ref
        L = ref.lib()
        n_cpu = min(args.cpu_channels, n_ch)
        counters = np.zeros(n_cpu*8, np.int64)
        cpu = ref_baseline("sig_tone_rx() 2280 Hz, pass-through", ref.MT_SIGTONE,
                           lambda c: L.glue_sigtone_rx_new_quiet(1, 0x40, counters.ctypes.data + c*64), L.sig_tone_rx_free,
                           frames[:, :n_cpu].contiguous().cpu().numpy())
    words = 27
    used = 5 + 3 + 9                                     # one notch, the flat filter and its meter, the counters and flags
    alg_read = n_ch*(FRAME*2 + used*4)
    alg_write = n_ch*(FRAME*2 + (used - 1)*4)
    value = steps*n_ch*FRAME/dt/1e6
    return {
        "metric": "Msamples/s of batched in-band signalling tone receive (8 kHz channels at real-time = value*1e6/8000)",
        "value": value, "unit": "Msamples/s", "realtime_channels": value*1e6/8000.0, "n_gpus": 1, "steps": steps,
        "warmup": args.warmup, "ms_per_step": dt*1e3/steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32+int32", "data": "synthetic",
        "config": {"workload": "sig_tone_rx 2280 Hz, pass-through with notch insertion, %d channels x %d-sample frames" % (n_ch, FRAME),
                   "channels_per_gpu": n_ch, "state_words": words, "reports_in_one_more_second_of_signal": reports},
        "roofline": {"bound": "hbm", "kernel": "sigtone_rx_kernel<1>", "achieved": (alg_read + alg_write)/(avg_ms*1e-3)/1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (alg_read + alg_write)/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS,
                     "traffic": None, "alg_read_bytes_per_launch": alg_read, "alg_write_bytes_per_launch": alg_write,
                     "avg_launch_us": avg_ms*1e3, "min_launch_us": min(per)*1e3,
                     "note": "sample-serial bi-quads and integer logic per channel; frames read and written in place"},
        "cpu_baseline": cpu}


def bench_fax_rx(args, dev, stream):
    """The receive front end of N FAX terminals: a V.29 receiver bank and a V.21 receiver bank fed the same frames, each on
    its own stream (what fax_modems_v29_v21_rx() does per channel, src/fax_modems.c:290-308): the step time of the pair
    against each bank on its own."""
    from spandsp_amd import engine
    n_ch = args.channels or 16384
    nf = args.steps + args.warmup
    frames, _, _ = synth_v29_on_device(n_ch, nf, dev, stream, seed=0x2929, modem="v29", line="in_step")
    frame_bytes = n_ch*FRAME*2
    s2 = torch.cuda.Stream(device=dev)

    def run(use_fast, use_slow):
        fast = engine.ModemBank(engine.V29, n_ch, 9600) if use_fast else None
        slow = engine.FskBank(engine.FSK_V21CH2, n_ch, engine.FSK_FRAME_MODE_SYNC) if use_slow else None
        if fast:
            fast.set_stream(ctypes.c_void_p(stream.cuda_stream))
        if slow:
            slow.set_stream(ctypes.c_void_p(s2.cuda_stream))

        def step(i):
            ptr = ctypes.c_void_p(frames.data_ptr() + i*frame_bytes)
            if fast:
                fast.rx_device(ptr, FRAME, FRAME)
            if slow:
                slow.rx_device(ptr, FRAME, FRAME)
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0)/args.steps
    t_fast = run(True, False)
    t_slow = run(False, True)
    t_pair = run(True, True)
    value = n_ch*FRAME/t_pair/1e6
    return {
        "metric": "Msamples/s of a batched FAX receive front end, V.29 + V.21 on the same frames (8 kHz channels at real-time = value*1e6/8000)",
        "value": value, "unit": "Msamples/s", "realtime_channels": value*1e6/8000.0, "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": t_pair*1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32+int32", "data": "synthetic",
        "config": {"workload": "v29_rx 9600 bps + fsk_rx V.21 ch 2 on the same %d channels x %d-sample frames, one stream per bank" % (n_ch, FRAME),
                   "channels_per_gpu": n_ch, "ms_per_step_v29_alone": t_fast*1e3, "ms_per_step_v21_alone": t_slow*1e3,
                   "ms_per_step_pair": t_pair*1e3},
        "roofline": None, "cpu_baseline": None}


def bench_dtmf_tx(args, dev, stream):
    """a DTMF sender bank (dtmf_tx x N) writing 160-sample frames into HBM, digits queued up front."""
    from spandsp_amd import engine
    n_ch = args.channels or 65536
    rng = np.random.default_rng(5)
    keys = np.frombuffer(b"0123456789ABCD*#", np.uint8)
    digs = keys[rng.integers(0, 16, (n_ch, 128))]
    bank = engine.TxBank(engine.TX_DTMF, n_ch)
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    lens = np.full(n_ch, 128, np.int32)
    res = np.zeros(n_ch, np.int32)
    assert engine.lib().spangpu_txbank_put_each(bank.h, 0, n_ch, digs.ctypes.data, 128, lens.ctypes.data, res.ctypes.data) == 0
    out = torch.zeros(4, n_ch, FRAME, dtype=torch.int16, device=dev)
    d_lens = torch.zeros(n_ch, dtype=torch.int32, device=dev)

    def step(i):
        bank.tx_device(ctypes.c_void_p(out[i % 4].data_ptr()), FRAME, FRAME, ctypes.c_void_p(d_lens.data_ptr()))
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record(stream)
        step(args.warmup + i)
        evs[i][1].record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [a.elapsed_time(b) for a, b in evs]
    avg_ms = sum(per)/len(per)
    assert int(d_lens.min()) == FRAME                     # every sender still had digits queued
    tone_frac = float((out != 0).float().mean())         # all senders share one cadence, so look at four frames
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import restated as orc
        from test_oracle_pin import use_golden_modem_tables
        use_golden_modem_tables()
        n_cpu = min(args.cpu_channels, n_ch)
        arr = (orc._DtmfTxState*n_cpu)()
        for c in range(n_cpu):
            orc.lib().orc_dtmf_tx_init(ctypes.byref(arr[c]))
            orc.lib().orc_dtmf_tx_put(ctypes.byref(arr[c]), digs[c].tobytes(), 128)
        buf = np.zeros((n_cpu, FRAME), np.int16)
        fn = orc.lib().orc_dtmf_tx_run_batch
        fn.restype = ctypes.c_longlong
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
        frames_cpu = 400
        item = ctypes.sizeof(orc._DtmfTxState)

        def work(lo, hi):
            fn(ctypes.addressof(arr) + lo*item, hi - lo, buf[lo:].ctypes.data, FRAME, FRAME, frames_cpu)
        cores, t = run_threads(n_cpu, work)
        cpu = {"value": n_cpu*frames_cpu*FRAME/t/1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
               "sample": "oracle/tonegen_oracle.c orc_dtmf_tx on %d channels x %d frames, %d threads" % (n_cpu, frames_cpu, cores)}
    alg_write = n_ch*(FRAME*2 + 4)
    alg_read = n_ch*26*4
    value = args.steps*n_ch*FRAME/dt/1e6
    return {
        "metric": "Msamples/s of batched dtmf_tx (signal source bank)", "value": value, "unit": "Msamples/s",
        "realtime_channels": value*1e6/8000.0, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt*1e3/args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "dtmf_tx bank, %d channels x %d-sample frames, 128 random digits queued per channel" % (n_ch, FRAME),
                   "channels_per_gpu": n_ch, "non_zero_sample_fraction_in_last_4_frames": tone_frac},
        "roofline": {"bound": "hbm", "kernel": "tx_bank_kernel", "achieved": (alg_write + alg_read)/(avg_ms*1e-3)/1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (alg_write + alg_read)/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS,
                     "traffic": None, "alg_write_bytes_per_launch": alg_write, "alg_read_bytes_per_launch": alg_read,
                     "avg_launch_us": avg_ms*1e3, "min_launch_us": min(per)*1e3},
        "cpu_baseline": cpu}


def modem_spot_check(kind, bit_rate, frames64, events64, cutoffs=None):
    """The oracle's receiver (oracle/v29_oracle.c ..., pinned to the reference) over the same first frames of 64 channels:
    the put_bit stream of every frame equal?  frames64: int16 [frames, 64, FRAME]; events64[frame][channel]: int8 arrays."""
    from oracle import restated as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()
    O = {"v29": orc.V29, "v27ter": orc.V27ter, "v17": orc.V17}[kind]
    nfr, v, _ = frames64.shape
    ok = True
    total = 0
    for c in range(v):
        o = O(bit_rate)
        if cutoffs is not None:
            o.set_signal_cutoff(float(cutoffs[c]))
        for k in range(nfr):
            o.sink.clear()
            o.rx(np.ascontiguousarray(frames64[k, c]))
            want = o.sink.events()["a"].astype(np.int8)
            ok = ok and np.array_equal(want, events64[k][c])
            total += len(want)
    return {"channels": v, "frames": nfr, "checked": "every put_bit / status call (%d) of the first %d frames of channels 0..%d against the oracle's %s_rx()"
            % (total, nfr, v - 1, kind), "bit_exact": bool(ok)}


def bench_modem(args, dev, stream):
    from spandsp_amd import engine
    fixture, bit_rate, n_words = MODEMS[args.workload]
    kind = {"v29": engine.V29, "v17": engine.V17, "v27ter": engine.V27TER}[args.workload]
    n_ch = args.channels or 16384
    nf = args.steps + args.warmup
    line = getattr(args, "line", "contract") or "contract"
    if args.workload in ("v29", "v27ter", "v17") and not args.replay_fixture:
        frames, line_what, line_levels = synth_v29_on_device(n_ch, nf, dev, stream, seed=0x2929, modem=args.workload, line=line)
    else:
        frames, line_what, line_levels = synth_v29(n_ch, nf, dev, seed=0x29290000, fixture=fixture), "the committed reference transmission replayed with per-channel delay, gain and noise", None
    stagger = getattr(args, "stagger", None)
    stagger = (FRAME if line == "contract" else 0) if stagger is None else int(stagger)
    if stagger:
        # every channel's transmission that many samples later than the bank's first (drawn per channel; SURVEY 8(d)-4: "random
        # start delay 0-159 samples"): the receivers then leave training at different bauds, and what they do every n-th baud
        # of data (V.29: the equaliser update, every 10th) no longer falls on the same round for all channels of a wavefront
        rng = np.random.default_rng(0x57A6)
        delay = torch.as_tensor(rng.integers(0, stagger, n_ch), device=dev)
        t = torch.arange(nf*FRAME, device=dev)
        for c0 in range(0, n_ch, 1024):
            c1 = min(n_ch, c0 + 1024)
            x = frames[:, c0:c1].permute(1, 0, 2).reshape(c1 - c0, nf*FRAME)
            idx = t[None, :] - delay[c0:c1, None]
            y = torch.where(idx >= 0, torch.gather(x, 1, idx.clamp(min=0)), torch.zeros((), dtype=x.dtype, device=dev))
            frames[:, c0:c1] = y.reshape(c1 - c0, nf, FRAME).permute(1, 0, 2)
        del t, delay
    engine.tune_modem_mapping(args.modem_mapping)
    bank = engine.ModemBank(kind, n_ch, bit_rate)
    # v29_rx_init() leaves the carrier detector at -28.5 dBm0 (v29rx.c:1129): it would never see the contract's lines below some
    # -26 dBm0; at the -45.5 dBm0 of the reference's FAX front end (fax_modems.c:416) the noise of its louder lines (-10 dBm0 at
    # 25 dB SNR = -35 dBm0) holds the detector up for ever and a receiver that once failed stays parked.  An installation sets
    # the detector for the level of its lines: v29_rx_set_signal_cutoff(level - 12 dB) per line -- on the bank, in the host
    # baseline and in the spot check alike.
    cutoffs = None
    if args.workload == "v29" and line == "contract" and not args.replay_fixture:
        cutoffs = (line_levels - 12.0).astype(np.float32)
        bank.set_signal_cutoffs(cutoffs)
        line_what += ", v29_rx_set_signal_cutoff(level - 12 dB) per line"
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    frame_bytes = n_ch*FRAME*2
    torch.cuda.synchronize()
    for i in range(args.warmup):
        bank.rx_device(ctypes.c_void_p(frames.data_ptr() + i*frame_bytes), FRAME, FRAME)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record(stream)
        bank.rx_device(ctypes.c_void_p(frames.data_ptr() + (args.warmup + i)*frame_bytes), FRAME, FRAME)
        evs[i][1].record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [a.elapsed_time(b) for a, b in evs]
    avg_ms = sum(per)/len(per)
    ev = bank.events()
    bits_last = int(sum(len(e) for e in ev))
    trained = 0
    for c in range(0, n_ch, max(1, n_ch//256)):
        _, w = bank.get_state(c)
        trained += int(w[6] == 0)
    # V.29 reads 320 B of PCM + 768 B of state per channel and frame and writes 768 + 24 B -- the contract's
    # algorithmic bytes (the hot subset of v29_rx_state_t); this implementation's state is every word the reference's struct
    # holds for the receiver (n_words: 281 / 270 / 547 for V.29 / V.27ter / V.17), reported beside it
    contract_state = {"v29": 768}.get(args.workload, n_words*4)
    alg_read = n_ch*(FRAME*2 + contract_state)
    alg_write = n_ch*(contract_state + 24)
    actual_read = n_ch*(FRAME*2 + n_words*4)
    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_modem(args.workload, bit_rate, frames[:, :min(args.cpu_channels, n_ch)].contiguous().cpu().numpy(), cutoffs)
        # a 64-channel bank on the first rows of the same frames (the same kernel family as every bank below 65 536 channels)
        v, nchk = min(64, n_ch), min(40, nf)
        small = engine.ModemBank(kind, v, bit_rate)
        if cutoffs is not None:
            small.set_signal_cutoffs(cutoffs[:v])
        ev64 = []
        for k in range(nchk):
            small.rx_device(ctypes.c_void_p(frames.data_ptr() + k*frame_bytes), FRAME, FRAME)
            ev64.append(small.events())
        cpu["spot_check"] = modem_spot_check(args.workload, bit_rate, frames[:nchk, :v].cpu().numpy(), ev64, cutoffs)
        small.close()
    value = args.steps*n_ch*FRAME/dt/1e6
    e2e = None if getattr(args, "no_e2e", False) else e2e_modem(kind, bit_rate, n_ch, frames[:3], dev)
    return {
        "e2e": e2e,
        "metric": "Msamples/s of batched %s %d bps receive (8 kHz channels at real-time = value*1e6/8000)" % (args.workload, bit_rate),
        "value": value, "unit": "Msamples/s", "realtime_channels": value*1e6/8000.0, "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt*1e3/args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s %d bps RX, %d channels x %d-sample frames%s: %s, %s"
                               % (args.workload, bit_rate, n_ch, FRAME, " (BASELINE configs[3])" if args.workload == "v29" else "", line_what,
                                  ("random start delay 0 .. %d samples" % (stagger - 1)) if stagger else "all channels starting in step"),
                   "channels_per_gpu": n_ch, "modem_mapping": args.modem_mapping,
                   "sampled_channels_in_data_mode_at_end": "%d of %d" % (trained, len(range(0, n_ch, max(1, n_ch//256)))),
                   "events_in_last_frame": bits_last},
        "roofline": {"bound": "hbm", "kernel": ("%s_quad_kernel<16, 4>" if (n_ch < 32768 and args.modem_mapping in (0, 4, 8) and args.workload in ("v29", "v17", "v27ter")) else "%s_bank_kernel") % args.workload, "achieved": alg_read/(avg_ms*1e-3)/1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_read/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS,
                     "traffic": None, "alg_read_bytes_per_launch": alg_read, "alg_write_bytes_per_launch": alg_write,
                     "state_bytes_actual_read_per_launch": actual_read,
                     "avg_launch_us": avg_ms*1e3, "min_launch_us": min(per)*1e3, "max_launch_us": max(per)*1e3,
                     "note": "VALU/LDS-issue bound state machine (SURVEY 8(d)); the HBM figure is reported, not targeted"},
        "cpu_baseline": cpu}




def compact_path(line, key, channels, stream_peak=None):
    """One BASELINE configuration's line boiled down for bench.py's `paths` object."""
    from spandsp_amd import roofline as rl
    rl.add_traffic(line.get("roofline"), key, channels)
    roof = dict(line["roofline"])
    if stream_peak:
        roof["measured_stream_peak"] = stream_peak
        roof["frac_of_measured_stream"] = roof["achieved"]/stream_peak
    out = {"workload": line["config"]["workload"], "channels": channels, "steps": line["steps"], "ms_per_step": line["ms_per_step"],
           "value": line["value"], "unit": line["unit"], "realtime_channels": line["realtime_channels"], "dtype": line["dtype"],
           "roofline": roof, "roofline_valu": rl.valu_roof(key, roof.get("avg_launch_us"), channels=channels),
           "cpu_baseline": line.get("cpu_baseline")}
    if line.get("e2e"):
        out["e2e"] = line["e2e"]
    for k in ("erle_db_last_second_single_talk_lines", "sampled_channels_in_data_mode_at_end", "events_in_last_frame", "blocks_with_a_hit_in_last_step"):
        if k in line["config"]:
            out[k] = line["config"][k]
    return out


def paths_for_bench(dev, stream, no_cpu_baseline=False, stream_peak=None, echo_seconds=10):
    """BASELINE configs[2], [3] and [4] at full size for the same JSON line as the headline (bench.py `paths`): each with its
    step time, the roofline of its kernel, a short cpu_baseline of the reference on the host cores and -- inside that leg --
    a spot check of 64 channels against the oracle.  Bounded: a few tens of seconds in all."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    base = dict(channels=0, warmup=0, no_cpu_baseline=no_cpu_baseline, echo_lanes=0, separate_launches=False, one_launch=False, cpu_channels=4096,
                fsk_waves=0, modem_mapping=0, replay_fixture=False, echo_seconds=echo_seconds, cpu_seconds=0.7)
    out = {}
    t0 = time.perf_counter()
    try:
        a = types.SimpleNamespace(workload="mixed", steps=200, **base)
        a.warmup = 20
        out["mixed"] = compact_path(bench_mixed(a, dev, stream), "mixed", 131072, stream_peak)
    except Exception as e:                                    # a path that fails must not take the headline line with it
        out["mixed"] = {"error": repr(e)}
    try:
        # the contract's workload: carrier 1700 +- 7 Hz, -30 .. -10 dBm0, SNR 25 .. 40 dB, random start delay 0 .. 159
        a = types.SimpleNamespace(workload="v29", steps=150, line="contract", stagger=None, **base)
        out["v29"] = compact_path(bench_modem(a, dev, stream), "v29", 16384, stream_peak)
        # beside it, the favourable case the earlier rounds quoted: every channel on the nominal carrier and starting in the same
        # frame, so that every receiver of a wavefront updates its equaliser on the same baud
        a = types.SimpleNamespace(workload="v29", steps=150, line="in_step", stagger=0, **dict(base, no_cpu_baseline=True, no_e2e=True))
        st = bench_modem(a, dev, stream)
        out["v29"]["in_step_nominal_carrier"] = {"workload": st["config"]["workload"], "ms_per_step": st["ms_per_step"], "avg_launch_us": st["roofline"]["avg_launch_us"],
                                                 "sampled_channels_in_data_mode_at_end": st["config"]["sampled_channels_in_data_mode_at_end"]}
    except Exception as e:
        out["v29"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    try:
        a = types.SimpleNamespace(workload="echo", steps=0, **base)
        out["echo"] = compact_path(bench_echo(a, dev, stream), "echo", 131072, stream_peak)
    except Exception as e:
        out["echo"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    out["seconds"] = time.perf_counter() - t0
    return out


def emit(line, key, channels=None):
    """Print a bench line with the bounds beside the HBM roofline: the measured stream ceiling of this device in this run, and
    the VALU issue floor of the workload's kernel (spandsp_amd/roofline.py)."""
    from spandsp_amd import roofline as rl
    roof = line.get("roofline")
    if roof:
        rl.add_measured(roof, 0)
        rl.add_traffic(roof, key, channels)
        line["roofline_valu"] = rl.valu_roof(key, roof.get("avg_launch_us") or roof.get("avg_tick_us"), channels=channels)
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["v29", "v17", "v27ter", "echo", "mixed", "dtmf_tx", "fsk", "mct", "sigtone", "supertone", "fax_rx", "v29_tx", "awgn"], default="v29")
    ap.add_argument("--channels", type=int, default=0)
    ap.add_argument("--steps", type=int, default=0, help="default: 150 (190 for v27ter, whose training alone is 0.7 s)")
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--echo-lanes", type=int, default=0, help="echo: lanes per channel (0 = the library's choice; 2, 4, 8, 16 for A-B runs)")
    ap.add_argument("--separate-launches", action="store_true", help="mixed: one launch per bank instead of one per step")
    ap.add_argument("--one-launch", action="store_true", help="mixed: time the one-launch form (the banks on one stream) instead of a launch per bank on streams of their own")
    ap.add_argument("--cpu-channels", type=int, default=16384)
    ap.add_argument("--stagger", type=int, default=None, help="v29 / v17 / v27ter: every channel's transmission starts a random number of samples (below this) late (default: 160 with --line contract, 0 with --line in_step)")
    ap.add_argument("--single-mode", action="store_true", help="mixed: time the chosen way of running a tick only (counter passes: every kernel then has one launch a tick)")
    ap.add_argument("--three-queues", action="store_true", help="mixed: a launch per bank on three hardware queues (the default is two queues: super-tone | Bell MF + R2 MF)")
    ap.add_argument("--no-cadences", action="store_true", help="mixed: the super-tone third without its cadence matcher (the rounds-1-to-5 workload; not configs[2])")
    ap.add_argument("--line", choices=["contract", "in_step"], default="contract", help="v29 / v17 / v27ter: SURVEY 8(d)-4's lines (carrier +- 7 Hz, -30 .. -10 dBm0, SNR 25 .. 40 dB, random start) or the nominal-carrier, all-in-step workload of the earlier rounds")
    ap.add_argument("--fsk-waves", type=int, default=0,
                    help="fsk / mct / sigtone: 0 = the library's choice, 1 = one wavefront per 64 receivers, 2 = two (A-B runs)")
    ap.add_argument("--modem-mapping", type=int, default=0,
                    help="v29 / v17 / v27ter: 0 = the library's choice, 1 = one channel per lane, 4 / 8 = four lanes per channel (A-B runs)")
    ap.add_argument("--echo-seconds", type=int, default=10, help="echo: seconds of continuous signal (the first one warms up, the rest is timed)")
    ap.add_argument("--replay-fixture", action="store_true",
                    help="v29 / v27ter / v17: replay the committed reference transmission instead of running the transmitter bank")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device; the engine has no CPU fallback")
    if args.steps <= 0:
        args.steps = 190 if args.workload == "v27ter" else 150
    dev = torch.device("cuda", 0)
    from spandsp_amd import engine
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    engine.tune_fsk_waves(args.fsk_waves)
    if args.workload == "echo":
        emit(bench_echo(args, dev, stream), "echo", args.channels or None)
        return
    if args.workload == "mixed":
        emit(bench_mixed(args, dev, stream), "mixed", args.channels or None)
        return
    if args.workload == "v29_tx":
        emit(bench_v29_tx(args, dev, stream), "v29_tx", args.channels or None)
        return
    if args.workload == "awgn":
        emit(bench_awgn(args, dev, stream), "awgn", args.channels or None)
        return
    if args.workload == "mct":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        emit(bench_mct(args, dev, stream), "mct", args.channels or None)
        return
    if args.workload == "fax_rx":
        emit(bench_fax_rx(args, dev, stream), "fax_rx", args.channels or None)
        return
    if args.workload == "supertone":
        emit(bench_supertone(args, dev, stream), "supertone", args.channels or None)
        return
    if args.workload == "sigtone":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        emit(bench_sigtone(args, dev, stream), "sigtone", args.channels or None)
        return
    if args.workload == "fsk":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        emit(bench_fsk(args, dev, stream), "fsk", args.channels or None)
        return
    if args.workload == "dtmf_tx":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        emit(bench_dtmf_tx(args, dev, stream), "dtmf_tx", args.channels or None)
        return
    emit(bench_modem(args, dev, stream), args.workload, args.channels or 16384)


if __name__ == "__main__":
    main()
