#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X.

Workload (N=1): BASELINE.json configs[1] -- "Batched DTMF Goertzel bank: 65 536
channels x 160-sample frames, 1 MI355X".  One STEP = one 20 ms tick: one pass of
the hot path (spangpu_bank_rx -> tone_bank_kernel<DtmfDet>) over the next
160-sample frame of all channels, inputs already resident in HBM, channel-major
[frame][channel][160] int16 exactly as N spandsp callers would hand them over.
Successive steps walk successive frames of a continuous synthetic signal (cadenced
DTMF digits + noise per SURVEY.md 8(d)), so the detectors do real work.

Multi-GPU (--gpus N via torch.distributed.run): channels shard across ranks
(65 536 per GPU, weak scaling, no data-path collective); after each step the
per-channel block records are gathered to rank 0 with RCCL (the only exchange the
path has).  value = samples processed by all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FRAME = 160                     # samples per channel per step (20 ms at 8 kHz)
ALG_READ_BYTES = 400            # 320 B PCM + 80 B state per channel-frame
ALG_WRITE_BYTES = 84
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def synth_dtmf_frames(n_ch, n_frames, device, seed):
    """Cadenced DTMF + noise for n_ch channels, returned as int16 [n_frames, n_ch, FRAME]
    on `device` (synthetic data; recipe of SURVEY.md 8(d) config 2)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    row = torch.tensor([697.0, 770.0, 852.0, 941.0], device=device)
    col = torch.tensor([1209.0, 1336.0, 1477.0, 1633.0], device=device)
    on, off = 400, 440                                      # 50 ms / 55 ms (dtmf.c:67-69)
    period = on + off
    n_samples = n_frames*FRAME
    n_sym = n_samples//period + 2
    keys = torch.randint(0, 16, (n_ch, n_sym), device=device, generator=g)
    start = torch.randint(0, period, (n_ch, 1), device=device, generator=g)
    lvl = torch.empty(n_ch, 1, device=device).uniform_(-25.0, -7.0, generator=g)
    tw = torch.empty(n_ch, 1, device=device).uniform_(-4.0, 8.0, generator=g)
    fo = torch.empty(n_ch, 2, device=device).uniform_(-0.015, 0.015, generator=g)
    nz = torch.empty(n_ch, 1, device=device).uniform_(-50.0, -25.0, generator=g)
    quiet = torch.rand(n_ch, 1, device=device, generator=g) < 0.25
    a_lo = 32768.0*torch.pow(10.0, (lvl - 3.14)/20.0)
    a_hi = 32768.0*torch.pow(10.0, (lvl - tw - 3.14)/20.0)
    sigma = 32768.0*torch.pow(10.0, (nz - 3.14)/20.0)/(2.0**0.5)
    out = torch.empty(n_frames, n_ch, FRAME, dtype=torch.int16, device=device)
    chunk = 8                                               # frames per synthesis chunk
    for f0 in range(0, n_frames, chunk):
        f1 = min(n_frames, f0 + chunk)
        t = torch.arange(f0*FRAME, f1*FRAME, device=device, dtype=torch.float32).unsqueeze(0)
        rel = t - start
        sym = torch.clamp(torch.floor(rel/period), 0, n_sym - 1).long()
        inside = (rel >= 0) & ((rel - torch.floor(rel/period)*period) < on) & (~quiet)
        k = torch.gather(keys, 1, sym.expand(n_ch, -1))
        f_lo = row[k >> 2]*(1.0 + fo[:, 0:1])
        f_hi = col[k & 3]*(1.0 + fo[:, 1:2])
        # phase in float64 to keep the tones clean over long signals
        ph_lo = (2.0*np.pi/8000.0)*f_lo.double()*t.double()
        ph_hi = (2.0*np.pi/8000.0)*f_hi.double()*t.double() + 1.0
        x = a_lo*torch.sin(ph_lo).float() + a_hi*torch.sin(ph_hi).float()
        x = torch.where(inside, x, torch.zeros_like(x))
        x = x + sigma*torch.randn(x.shape, device=device, generator=g)
        x = torch.clamp(torch.trunc(x), -32768, 32767).to(torch.int16)
        out[f0:f1] = x.view(n_ch, f1 - f0, FRAME).permute(1, 0, 2)
    return out


def cpu_baseline(frames_host, target_s):
    """Time the CPU path on the host cores over a bounded sample of the same workload.

    With the real reference present (oracle/_ref/libspandsp_ref.so -- it travels with the snapshot) dtmf_rx() is
    driven by the pthread driver of oracle/ref_glue/ref_glue_mt.c: one C call per measurement, every thread looping
    over its static slice of channel objects, frames and repetitions inside C.  Two figures: all host cores (the
    headline `value`) and one core.  Without it, the C restatement (oracle/liboracle.so) is driven from one Python
    thread per core (ctypes releases the GIL), one call per frame."""
    import oracle
    from oracle import ref, restated
    n_frames, n_ch, _ = frames_host.shape
    cores = os.cpu_count() or 1
    if oracle.have_ref():
        L = ref.lib()
        per_loop = float(frames_host.size)

        def measure(ch, threads, seconds):
            states = [L.glue_dtmf_rx_new(None, 0, 0) for _ in range(ch)]
            sub = np.ascontiguousarray(frames_host[:, :ch])
            rate, loops, dt = ref.timed_baseline(lambda l: ref.mt_rx(ref.MT_DTMF, states, sub, l, threads), float(sub.size), seconds)
            return rate, loops, dt

        usable, cores_note = ref.usable_cores()
        threads = max(1, min(usable, n_ch))
        all_rate, all_loops, all_dt = measure(n_ch, threads, target_s)
        one_ch = min(n_ch, 256)
        one_rate, one_loops, one_dt = measure(one_ch, 1, min(target_s, 2.0))
        shipped = None
        if oracle.have_ref_fast():
            # the same sources as the library ships them (oracle/Makefile: -O2 -ffast-math -msse2, SPANDSP_USE_SSE2): timing only
            with ref.flavour("fast"):
                L = ref.lib()
                f_all = measure(n_ch, threads, 0.6*target_s)
                f_one = measure(one_ch, 1, min(0.6*target_s, 1.0))
            L = ref.lib()
            shipped = {"kind": "reference-fastmath", "value": f_all[0]/1e6, "single_core": f_one[0]/1e6, "unit": "Msamples/s", "cores": threads,
                       "build": "gcc -std=gnu99 -O2 -ffast-math -msse2 -DSPANDSP_USE_SSE2 (configure.ac:276,346,374-375,509-519); "
                                "never used for parity"}
        return {
            "as_shipped": shipped,
            "value": all_rate/1e6,
            "unit": "Msamples/s",
            "cores": threads,
            "kind": "reference",
            "single_core": one_rate/1e6,
            "host_cores": cores_note,
            "sample": "reference (oracle/_ref) dtmf_rx(), pthread driver: all usable cores = %d channels x %d distinct frames x %d "
                      "passes on %d threads in %.2f s; one core = %d channels x %d frames x %d passes in %.2f s"
                      % (n_ch, n_frames, all_loops, threads, all_dt, one_ch, n_frames, one_loops, one_dt),
        }
    cores = max(1, min(cores, n_ch))
    bounds = np.linspace(0, n_ch, cores + 1).astype(int)
    L = restated.lib()
    L.orc_dtmf_rx_batch.restype = None
    L.orc_dtmf_rx_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int]
    sz = L.orc_dtmf_sizeof()
    blob = np.zeros(n_ch*sz, np.uint8)
    for c in range(n_ch):
        L.orc_dtmf_init(blob.ctypes.data + c*sz, 0)
    loops = 10

    def work(lo, hi):
        for _ in range(loops):
            for f in range(n_frames):
                L.orc_dtmf_rx_batch(blob.ctypes.data + lo*sz, frames_host[f, lo:hi].ctypes.data, hi - lo, FRAME, FRAME)
    threads = [threading.Thread(target=work, args=(int(bounds[i]), int(bounds[i + 1]))) for i in range(cores)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    samples = float(n_frames)*loops*n_ch*FRAME
    return {
        "value": samples/dt/1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d channels x %d frames of %d samples (%d distinct frames cycled %dx), C restatement (oracle/) dtmf_rx on %d "
                  "Python-driven host threads, %.1f s" % (n_ch, n_frames*loops, FRAME, n_frames, loops, cores, dt),
    }


def end_to_end(engine, n_ch, frames_dev, device_index, steps, law=0):
    """SURVEY 8(d)'s second number: a tick as a caller with HOST buffers sees it, through the pipelined feed
    (spangpu_feed_*, csrc/feed_api.hip): the frame sits in one of the feed's pinned slots (where the caller's receive
    path wrote it), commit() queues H2D -> kernel -> digit list -> D2H, collect() hands out the digits of the tick before:
    tick t + 1's copy down overlaps tick t's kernel and the way back of its digits.  law = 0: int16 PCM; 1 / 2: G.711
    bytes decoded on the device (half the PCIe volume).  `with_fill` adds one host thread's copy of the frame from
    ordinary memory into the slot (a caller that cannot receive into the slot directly)."""
    import numpy as np
    bank = engine.ToneBank(engine.DTMF, n_ch, device=device_index)
    feed = engine.Feed(bank, FRAME, law=law, depth=3, device=device_index)
    nf = min(frames_dev.shape[0], 8)
    host = frames_dev[:nf].cpu().numpy()
    if law:
        host = host.astype(np.uint8) if host.dtype != np.uint8 else host

    def run(n, fill):
        digits = 0
        for i in range(n):
            buf = feed.slot()
            if fill or i < feed.depth:
                buf[:, :FRAME] = host[i % nf]
            feed.commit(FRAME)
            if i >= 1:
                digits += len(feed.collect()[0])
        while True:
            got = feed.collect()
            if got is None:
                break
            digits += len(got[0])
        return digits
    run(6, True)
    t0 = time.perf_counter()
    n_digits = run(steps, False)
    dt_py = time.perf_counter() - t0
    # the same loop in C (spangpu_feed_run): what a C caller's media thread gets.  This harness's Python loop pays ~0.1 ms
    # per tick for three ctypes calls and the interpreter lock on top of it, which is time the copy engine then idles.
    ms_c, n_digits = feed.run(FRAME, steps, 1)
    dt = ms_c*1e-3
    t0 = time.perf_counter()
    run(max(4, steps//4), True)
    dt_fill = (time.perf_counter() - t0)/max(4, steps//4)
    feed.close()
    bank.close()
    bps = 1 if law else 2
    # the floor of this path on this box: the same bytes, pinned host memory -> HBM, nothing else
    import torch
    pinned = torch.empty(n_ch*FRAME*bps, dtype=torch.uint8).pin_memory()
    onboard = torch.empty(n_ch*FRAME*bps, dtype=torch.uint8, device="cuda:%d" % device_index)
    onboard.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        onboard.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    h2d_ms = (time.perf_counter() - t0)/20*1e3
    return {
        "ms_per_step": dt*1e3/steps,
        "ms_per_step_python_loop": dt_py*1e3/steps,
        "h2d_copy_alone_ms": h2d_ms,
        "value": float(steps)*n_ch*FRAME/dt/1e6,
        "unit": "Msamples/s",
        "steps": steps,
        "ms_per_step_with_fill": dt_fill*1e3,
        "includes": "per step (tick loop in C, spangpu_feed_run()), pipelined over three pinned slots: H2D copy of the %d x %d %s frame (%.1f MB), the kernel, the digit "
                    "list made on the device, D2H of the list (4 bytes per digit; %d digits in all), one tick of latency"
                    % (n_ch, FRAME, "G.711" if law else "int16", n_ch*FRAME*bps/1e6, n_digits),
    }


def large_banks(engine, dev, local_rank, frames, stream):
    """Not the headline: the same detector on banks large enough for several waves per SIMD (262 144 and 1 048 576 channels x
    160-sample frames; the headline's frames tiled along the channel axis), one launch per tick and in queue mode
    (spangpu_bank_set_queues(bank, 2): the launch cut in two on two hardware queues).  Step time by HIP events on the bank's
    stream around 100 ticks, the second queue joined before the closing event."""
    out = {}
    nf = min(6, frames.shape[0])
    for n_ch in (262144, 1048576):
        try:
            rep = n_ch//frames.shape[1]
            big = frames[:nf].repeat(1, rep, 1).contiguous()
            fb = n_ch*FRAME*2
            res = {}
            for q in (1, 2):
                bank = engine.ToneBank(engine.DTMF, n_ch, device=local_rank)
                bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
                bank.set_queues(q)
                for i in range(10):
                    bank.rx_device(ctypes.c_void_p(big.data_ptr() + (i % nf)*fb), FRAME, FRAME)
                bank.join()
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for i in range(100):
                    bank.rx_device(ctypes.c_void_p(big.data_ptr() + (i % nf)*fb), FRAME, FRAME)
                bank.join()
                e1.record(stream)
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1)*10.0
                res["queues_%d" % q] = {"us_per_step": us, "roofline_frac": n_ch*ALG_READ_BYTES/(us*1e-6)/1e9/HBM_PEAK_GBPS}
                bank.close()
            out[str(n_ch)] = res
            del big
            torch.cuda.empty_cache()
        except Exception as e:                      # never takes the headline line with it
            out[str(n_ch)] = {"error": repr(e)}
    return out


def frames_per_launch(engine, dev, local_rank, frames, stream):
    """Not the headline (which is one 160-sample frame per launch, a 20 ms tick): the same bank given two and three frames per
    launch -- a 40 / 60 ms jitter buffer in front of the detector --, which spreads the launch's fixed cost (launch boundary,
    first burst, state in and out, block ends: DESIGN 4.1) over more samples.  Results are those of one frame at a time (any
    call length is the same detector: tests/test_tone_gpu.py runs ragged lengths).  Per launch length: microseconds per
    launch and per frame, and the roofline fraction both on SURVEY 8(d)'s 400 B per channel and frame and on the bytes such a
    launch actually owes (k x 320 B of PCM + 80 B of state once)."""
    out = {}
    n_ch = frames.shape[1]
    for k in (1, 2, 3):
        try:
            sets = max(2, min(6, frames.shape[0]//k))
            rows = torch.stack([torch.cat([frames[s*k + j] for j in range(k)], dim=1) for s in range(sets)]).contiguous()   # [set][ch][k*160]
            bank = engine.ToneBank(engine.DTMF, n_ch, device=local_rank)
            bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
            rb = n_ch*k*FRAME*2
            for i in range(20):
                bank.rx_device(ctypes.c_void_p(rows.data_ptr() + (i % sets)*rb), k*FRAME, k*FRAME)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            n = 600
            e0.record(stream)
            for i in range(n):
                bank.rx_device(ctypes.c_void_p(rows.data_ptr() + (i % sets)*rb), k*FRAME, k*FRAME)
            e1.record(stream)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1)*1e3/n
            out[str(k)] = {"us_per_launch": us, "us_per_frame": us/k,
                           "roofline_frac_on_contract_bytes": n_ch*k*ALG_READ_BYTES/(us*1e-6)/1e9/HBM_PEAK_GBPS,
                           "roofline_frac_on_bytes_owed": n_ch*(k*FRAME*2 + 80)/(us*1e-6)/1e9/HBM_PEAK_GBPS,
                           "buffered_ms": 20*k}
            bank.close()
            del rows
        except Exception as e:                      # never takes the headline line with it
            out[str(k)] = {"error": repr(e)}
    torch.cuda.empty_cache()
    return out


def run_echo(args, engine, dev, local_rank, rank, world):
    """BASELINE configs[4]: the G.168 canceller (echo.c, 128 taps, ECHO_CAN_USE_ADAPTION) on 131 072 channels per GPU,
    channels sharded over the ranks with no data-path collective; once per second of signal (50 steps) every rank turns
    its per-channel energy sums into ERLE (spangpu_echo_erle, written into the RCCL send buffer) and the floats are
    gathered to rank 0.  A step = one 160-sample frame of every channel of the rank."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_paths as bp
    from spandsp_amd.parallel import FloatGather
    n_ch = args.channels if args.channels != 65536 else 131072
    # SURVEY 8(d)-5's workload: 10 s of continuous signal per line (G.168 echo path models, tools/bench_paths.py: synth_echo);
    # the first second warms up, the other nine are the timed region (no frame is played twice: no seam in any line)
    nf = 50*args.echo_seconds
    warm = 50
    tx, rx = bp.synth_echo(n_ch, nf, dev, seed=0xEC40 + rank)
    clean = torch.empty(n_ch, FRAME, dtype=torch.int16, device=dev)
    bank = engine.EchoBank(n_ch, bp.ECHO_TAPS, bp.ECHO_MODE, device=local_rank)
    lanes = engine.lib().spangpu_echo_lanes_per_channel(bank.h)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    bank.stats(2)                       # energy sums by the update kernel itself (ERLE needs no CRC)
    force_gather = world == 1 and os.environ.get("SPANGPU_BENCH_FORCE_GATHER") == "1"
    gather = FloatGather(world, rank, n_ch, dev) if (world > 1 or force_gather) else None
    erle_dev = gather.send if gather is not None else torch.zeros(n_ch, dtype=torch.float32, device=dev)
    fb = n_ch*FRAME*2
    report_every = 50

    def step(i):
        bank.update_device(ctypes.c_void_p(tx.data_ptr() + i*fb), ctypes.c_void_p(rx.data_ptr() + i*fb),
                           ctypes.c_void_p(clean.data_ptr()), FRAME, FRAME)
        if (i + 1) % report_every == 0:
            if gather is not None:
                gather.result()                     # the previous report has arrived (and the send buffer is free again)
            bank.erle_device(ctypes.c_void_p(erle_dev.data_ptr()))
            if gather is not None:
                gather.gather(bank)                 # (checks that the bank launches on the current stream: the collective waits for that one)
            if i + 1 < nf:
                bank.stats_reset(sums=True, crc=False)

    for i in range(warm):
        step(i)
    timed_steps = nf - warm
    args.steps, args.warmup = timed_steps, warm
    if gather is not None:
        gather.result()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(warm, nf):
        step(i)
    if gather is not None:
        gather.result()
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stream_ms = ev0.elapsed_time(ev1)
    # the last report (step nf - 1) holds the ERLE of every line over the last second
    if gather is not None:
        allr = gather.result()
    else:
        torch.cuda.synchronize()
        allr = erle_dev.unsqueeze(0)
    if rank != 0:
        return
    quiet = (torch.arange(n_ch, device=dev) % 10 != 0)
    erle_single = allr[:, quiet].float()
    state_bytes = 48*4 + bp.ECHO_TAPS*4 + 4*bp.ECHO_TAPS*2 + bp.ECHO_TAPS*2
    alg_read = n_ch*(2*FRAME*2 + state_bytes)
    avg_ms = stream_ms/timed_steps
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        nc = min(4096, n_ch)
        cpu = bp.cpu_echo(tx[:60, :nc].contiguous().cpu().numpy(), rx[:60, :nc].contiguous().cpu().numpy())
    value = float(timed_steps)*n_ch*world*FRAME/dt/1e6
    bp.emit({
        "metric": "Msamples/s of batched G.168 echo cancellation, 128 taps (8 kHz channels at real-time = value*1e6/8000)",
        "value": value, "unit": "Msamples/s", "realtime_channels": value*1e6/8000.0, "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "timed_steps": timed_steps, "timed_region_ms": dt*1e3,
        "ms_per_step": dt*1e3/timed_steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4], SURVEY 8(d)-5's lines: echo_can_update 128 taps, ECHO_CAN_USE_ADAPTION, %d channels/GPU x "
                               "%d-sample frames, %d s of continuous signal (white noise at -15 dBm0 through G.168 echo path models D2..D9 "
                               "by channel mod 8, ERL 6..24 dB, every tenth line with near end talk), per-channel ERLE over the last second "
                               "gathered to rank 0 every %d steps" % (n_ch, FRAME, args.echo_seconds, report_every),
                   "channels_per_gpu": n_ch, "frame_samples": FRAME,
                   "parallelism": ("channels sharded x%d, RCCL gather of one ERLE float per channel per second of signal" % world)
                                  if world > 1 else "single GPU",
                   "erle_db_single_talk_channels": {"median": float(erle_single.median()), "p10": float(erle_single.quantile(0.1)),
                                                    "ranks": int(allr.shape[0])}},
        "roofline": {"bound": "hbm", "kernel": "echo canceller kernel, %d lanes per channel (echo_%s)" % (lanes, "pair_kernel" if lanes == 2 else "bank_kernel"),
                     "achieved": alg_read/(avg_ms*1e-3)/1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg_read/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS, "traffic": None, "alg_read_bytes_per_launch": alg_read,
                     "avg_launch_us": avg_ms*1e3,
                     "note": "integer-VALU bound (2 x 128 MACs per sample per channel); the HBM figure is reported, not targeted; "
                             "avg_launch_us is a whole step (update + statistics kernels)"},
        "cpu_baseline": cpu}, "echo", n_ch)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--channels", type=int, default=65536, help="channels per GPU (BASELINE config 2: 65536)")
    ap.add_argument("--distinct-frames", type=int, default=100, help="distinct 20 ms frames resident in HBM (cycled)")
    ap.add_argument("--gather", choices=["digits", "records"], default="digits",
                    help="multi-GPU: what rank 0 collects per step -- one byte per block and channel (the digit delivered, 0 = "
                         "none; default) or the 32-bit record word of every block; either is written by the detector kernel "
                         "straight into the RCCL send buffer")
    ap.add_argument("--gather-every", type=int, default=25,
                    help="multi-GPU: steps per RCCL gather (25 = one report to rank 0 per 0.5 s of signal; every digit of every "
                         "step travels, a report only batches them: measured on one rank, a collective per 5 steps costs "
                         "3.1 us per step, per 25 steps 1.3 us)")
    ap.add_argument("--queues", type=int, default=1,
                    help="spangpu_bank_set_queues(): 2 = the bank's launches cut in two on two hardware queues (pays from 131072 "
                         "channels: profiles/r5_probe_mq.log), 0 = the library's choice, 1 = one launch per step (the default, and "
                         "what the headline configuration runs)")
    ap.add_argument("--g711", choices=["none", "alaw", "ulaw"], default="none",
                    help="feed the bank G.711 bytes (decoded on the device) instead of 16 bit linear PCM; not the "
                         "BASELINE configuration -- a variant of it with the wire format of a trunk")
    ap.add_argument("--workload", choices=["dtmf", "echo"], default="dtmf",
                    help="dtmf: BASELINE configs[1] (the headline); echo: BASELINE configs[4], one GPU's shard per rank "
                         "(131072 channels of the 128-tap echo canceller), ERLE of every channel gathered to rank 0 "
                         "once per second of signal")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-channels", type=int, default=16384)
    ap.add_argument("--cpu-frames", type=int, default=40)
    ap.add_argument("--cpu-seconds", type=float, default=1.5, help="wall time of the all-core reference measurement")
    ap.add_argument("--min-timed-ms", type=float, default=50.0,
                    help="the timed region is repeated (whole multiples of --steps) until it lasts at least this long")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--echo-seconds", type=int, default=10, help="--workload echo: seconds of continuous signal per line (first second untimed)")
    ap.add_argument("--no-paths", action="store_true", help="leave out the `paths` object (BASELINE configs[2], [3], [4] at full size, ~40 s)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("SPANGPU_BENCH_SPAWN") == "1"):
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU, RCCL over xGMI), the way the
        # driver's own command line does, hand them this command line, and pass on rank 0's JSON line and the exit code
        # (SPANGPU_BENCH_SPAWN=1 takes the same road with one rank: the self-test of this path on a one-GPU box)
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ)
        env.pop("SPANGPU_BENCH_SPAWN", None)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with python -m torch.distributed.run --nproc-per-node %d, "
                         "or without the rendezvous variables: bench.py then starts its ranks itself)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SPANGPU_BENCH_FORCE_GATHER=1 runs the RCCL record gather even with one rank (a self-test of the N > 1 path on a
    # one-GPU box; launch through torch.distributed.run so that the rendezvous variables exist)
    force_gather = world == 1 and os.environ.get("SPANGPU_BENCH_FORCE_GATHER") == "1"
    if world > 1 or force_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    from spandsp_amd import engine
    from spandsp_amd.parallel import DigitGather, ResultGather, shard_range

    if args.workload == "echo":
        run_echo(args, engine, dev, local_rank, rank, world)
        if world > 1 or force_gather:
            dist.destroy_process_group()
        return

    n_ch = args.channels
    lo, hi = shard_range(n_ch*world, world, rank)
    assert hi - lo == n_ch
    frames = synth_dtmf_frames(n_ch, args.distinct_frames, dev, seed=0x5EED0000 + rank)
    law = {"none": 0, "alaw": 1, "ulaw": 2}[args.g711]
    if law:
        # encode to G.711 with the decode table of the reference (tests/golden/g711_decode.npz): nearest code
        tab = np.load(os.path.join(ROOT, "tests", "golden", "g711_decode.npz"))[args.g711].astype(np.int32)
        order = np.argsort(tab, kind="stable")
        vals = torch.tensor(tab[order], device=dev, dtype=torch.int32)
        codes_of = torch.tensor(order.astype(np.uint8), device=dev)
        x = frames.to(torch.int32)
        pos = torch.clamp(torch.searchsorted(vals, x.contiguous()), 1, 255)
        lower = (x - vals[pos - 1]) <= (vals[pos] - x)
        frames = codes_of[torch.where(lower, pos - 1, pos)].contiguous()
    torch.cuda.synchronize()

    bank = engine.ToneBank(engine.DTMF, n_ch, device=local_rank)
    # A dedicated (non-null) torch stream carries every launch, event and collective of the
    # timed region, so torch.cuda.Event timing sees exactly the stream the kernels run on.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    n_queues = bank.set_queues(args.queues)
    gather = None
    if world > 1 or force_gather:
        if args.gather == "digits":
            gather = DigitGather(world, rank, n_ch, 2, dev, every=args.gather_every)
        else:
            gather = ResultGather(world, rank, n_ch, max_blocks=2, device=dev, every=args.gather_every)
    frame_bytes = n_ch*FRAME*(1 if law else 2)
    base_ptr = frames.data_ptr()
    nf = args.distinct_frames

    def step(i):
        if gather is not None:
            gather.aim(bank)                    # the kernel writes its digit list / records straight into the RCCL send buffer
        if law:
            bank.rx_device_g711(ctypes.c_void_p(base_ptr + (i % nf)*frame_bytes), law, FRAME, FRAME)
        else:
            bank.rx_device(ctypes.c_void_p(base_ptr + (i % nf)*frame_bytes), FRAME, FRAME)
        if gather is not None:
            gather.submit(bank)

    for i in range(args.warmup):
        step(i)
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()
    # A region of --steps launches of this kernel lasts a fraction of a millisecond: repeat it (whole multiples of
    # --steps, `reps` of them) until the timed region is at least --min-timed-ms long.  Estimated on an untimed probe.
    ev_a = torch.cuda.Event(enable_timing=True)
    ev_b = torch.cuda.Event(enable_timing=True)
    ev_a.record(stream)
    for i in range(args.steps):
        step(args.warmup + i)
    ev_b.record(stream)
    torch.cuda.synchronize()
    probe_ms = max(ev_a.elapsed_time(ev_b), 1e-3)
    reps = max(1, int(np.ceil(1.15*args.min_timed_ms/probe_ms)))      # (the probe runs cold: margin)
    if world > 1:
        r = torch.tensor([reps], device=dev, dtype=torch.int64)
        dist.all_reduce(r, op=dist.ReduceOp.MAX)
        reps = int(r.item())
    timed_steps = args.steps*reps
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(timed_steps):
        step(args.warmup + i)
    if gather is not None:
        gather.drain()
    bank.join()
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stream_ms = ev0.elapsed_time(ev1)

    if gather is not None:
        bank.set_records_buffer(None, 0)
        bank.set_digits_buffer(None, 0)
    # ---- roofline: the kernel's average launch duration = HIP events on the launch stream around the timed
    # region / launches in it (back-to-back launches of the one kernel; agrees with the rocprofv3 kernel average
    # in profiles/).  A second pass with an event pair around every launch gives the spread; each pair adds the
    # latency of two barrier packets, so its average reads ~2 us high and is reported as such. ----
    roof = None
    if rank == 0:
        k = min(args.steps, 200)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
        for i in range(k):
            evs[i][0].record(stream)
            if law:
                bank.rx_device_g711(ctypes.c_void_p(base_ptr + (i % nf)*frame_bytes), law, FRAME, FRAME)
            else:
                bank.rx_device(ctypes.c_void_p(base_ptr + (i % nf)*frame_bytes), FRAME, FRAME)
            evs[i][1].record(stream)
        torch.cuda.synchronize()
        per = sorted(a.elapsed_time(b) for a, b in evs)
        pair_avg_ms = sum(per)/len(per)
        avg_ms = stream_ms/timed_steps
        alg_read = ALG_READ_BYTES - (FRAME if law else 0)           # G.711: 160 B of codes instead of 320 B of PCM
        achieved = n_ch*alg_read/(avg_ms*1e-3)/1e9
        roof = {
            "bound": "hbm",
            "kernel": "tone_fast_kernel<DtmfDet<false>, 1 channel per lane, 2-slot ring, loader wave>",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved/HBM_PEAK_GBPS,
            "traffic": None,
            "alg_read_bytes_per_launch": n_ch*alg_read,
            "alg_write_bytes_per_launch": n_ch*ALG_WRITE_BYTES,
            "avg_launch_us": avg_ms*1e3,
            "event_pair_avg_launch_us": pair_avg_ms*1e3,
            "event_pair_median_launch_us": per[len(per)//2]*1e3,
        }
        if not law:
            # HBM bytes per launch by the PMC counters (profiles/hbm_traffic.json), nulled when the kernel's sources are not the
            # ones the counters were taken on (spandsp_amd/roofline.py: hbm_traffic)
            from spandsp_amd import roofline as rl_t
            rl_t.add_traffic(roof, "dtmf", n_ch)

    roof_valu = None
    if rank == 0:
        from spandsp_amd import roofline as rl
        rl.add_measured(roof, local_rank)
        roof["bound_note"] = ("graded against the HBM read roofline as north_star asks; at this bank size the launch is bound by "
                              "launch boundary + first-data latency + the VALU issue of the unfusable recurrence (roofline_valu, DESIGN 4.1)")
        roof_valu = rl.valu_roof("dtmf", roof["avg_launch_us"], channels=n_ch)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not law:
        nc = min(args.cpu_channels, n_ch)
        host = frames[:min(args.cpu_frames, nf), :nc].contiguous().cpu().numpy()
        cpu = cpu_baseline(host, args.cpu_seconds)
    e2e = None
    e2e_g711 = None
    if rank == 0 and world == 1 and not args.no_e2e and not law:
        e2e = end_to_end(engine, n_ch, frames, local_rank, 60)
        # the same frames as u-law bytes (nearest code of the reference's decode table)
        tab = np.load(os.path.join(ROOT, "tests", "golden", "g711_decode.npz"))["ulaw"].astype(np.int32)
        order = np.argsort(tab, kind="stable")
        vals = torch.tensor(tab[order], device=dev, dtype=torch.int32)
        codes_of = torch.tensor(order.astype(np.uint8), device=dev)
        x = frames[:8].to(torch.int32)
        posn = torch.clamp(torch.searchsorted(vals, x.contiguous()), 1, 255)
        lower = (x - vals[posn - 1]) <= (vals[posn] - x)
        e2e_g711 = end_to_end(engine, n_ch, codes_of[torch.where(lower, posn - 1, posn)].contiguous(), local_rank, 60, law=2)

    large = None
    fpl = None
    if rank == 0 and world == 1 and not args.no_paths and not law and n_ch == 65536:
        large = large_banks(engine, dev, local_rank, frames, stream)
        fpl = frames_per_launch(engine, dev, local_rank, frames, stream)
    paths = None
    if rank == 0 and world == 1 and not args.no_paths and not law:
        # BASELINE configs[2], [3], [4] under the same clock as the headline (tools/bench_paths.py: paths_for_bench)
        del frames
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_paths as bp
        paths = bp.paths_for_bench(dev, stream, args.no_cpu_baseline, roof.get("measured_stream_peak") if roof else None)

    if rank == 0:
        total_samples = float(timed_steps)*n_ch*world*FRAME
        value = total_samples/dt/1e6
        line = {
            "metric": "Msamples/s of batched DTMF Goertzel detect (8 kHz channels at real-time = value*1e6/8000)",
            "value": value,
            "unit": "Msamples/s",
            "realtime_channels": value*1e6/8000.0,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "timed_steps": timed_steps,
            "timed_region_ms": dt*1e3,
            "ms_per_step": dt*1e3/timed_steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: batched DTMF Goertzel bank, %d channels/GPU x %d-sample frames, "
                            "channel-major %s resident in HBM, %d distinct frames cycled"
                            % (n_ch, FRAME, ("G.711 %s bytes (decoded on the device)" % args.g711) if law else "int16", nf),
                "channels_per_gpu": n_ch,
                "frame_samples": FRAME,
                "queues": n_queues,
                "parallelism": ("channels sharded x%d, RCCL gather of %s every %d steps"
                                % (world, "one digit byte per block and channel" if args.gather == "digits"
                                   else "all block records", args.gather_every))
                               if world > 1 else "single GPU",
            },
            "roofline": roof,
            "roofline_valu": roof_valu,
            "cpu_baseline": cpu,
            "e2e": e2e,
            "e2e_g711": e2e_g711,
            "paths": paths,
            "large_bank": large,
            "frames_per_launch": fpl,
        }
        print(json.dumps(line))
    if world > 1 or force_gather:
        if rank == 0 and gather is not None and gather.latest() is not None:
            assert gather.latest().shape == (world, args.gather_every, gather.n)
            if args.gather == "digits":
                dg = gather.digits()
                assert dg.shape == (world, args.gather_every, 2, n_ch) and int((dg != 0).sum()) > 0, "no rank reported a digit"
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
