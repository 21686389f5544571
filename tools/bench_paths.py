#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json (bench.py carries the headline DTMF line):

    python tools/bench_paths.py --workload v29  [--channels 16384]      # configs[3]
    python tools/bench_paths.py --workload echo [--channels 131072]     # configs[4], one GPU's shard

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


def synth_v29(n_ch, n_frames, dev, seed):
    g = np.load(os.path.join(ROOT, "tests", "golden", "v29_9600.npz"))
    base = torch.tensor(g["amp"].astype(np.float32), device=dev)
    period = base.numel() + 270
    base = torch.cat([base, torch.zeros(270, device=dev)])
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    delay = torch.randint(0, FRAME, (n_ch, 1), device=dev, generator=gen)
    gain = torch.pow(10.0, torch.empty(n_ch, 1, device=dev).uniform_(-14.0, 3.0, generator=gen)/20.0)
    sigma = torch.empty(n_ch, 1, device=dev).uniform_(1.0, 40.0, generator=gen)
    out = torch.empty(n_frames, n_ch, FRAME, dtype=torch.int16, device=dev)
    for f in range(n_frames):
        t = torch.arange(f*FRAME, (f + 1)*FRAME, device=dev).unsqueeze(0) - delay
        x = torch.where(t >= 0, base[torch.remainder(t, period)], torch.zeros((), device=dev))
        x = x*gain + sigma*torch.randn(n_ch, FRAME, device=dev, generator=gen)
        out[f] = torch.clamp(torch.round(x), -32768, 32767).to(torch.int16)
    return out


def cpu_v29(frames_host):
    import oracle
    from oracle import ref
    assert oracle.have_ref(), "cpu baseline for the modem path needs oracle/_ref"
    n_frames, n_ch, _ = frames_host.shape
    cores = max(1, min(os.cpu_count() or 1, n_ch))
    bounds = np.linspace(0, n_ch, cores + 1).astype(int)
    L = ref.lib()
    L.glue_v29_rx_new_quiet.restype = ctypes.c_void_p
    L.glue_v29_rx_new_quiet.argtypes = [ctypes.c_int]
    L.glue_v29_rx_batch.restype = None
    L.glue_v29_rx_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int]
    arr = (ctypes.c_void_p*n_ch)(*[L.glue_v29_rx_new_quiet(9600) for _ in range(n_ch)])

    def work(lo, hi):
        base = ctypes.addressof(arr) + lo*ctypes.sizeof(ctypes.c_void_p)
        for f in range(n_frames):
            L.glue_v29_rx_batch(base, frames_host[f, lo:hi].ctypes.data, hi - lo, FRAME, FRAME)
    th = [threading.Thread(target=work, args=(int(bounds[i]), int(bounds[i + 1]))) for i in range(cores)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    return {"value": n_frames*n_ch*FRAME/dt/1e6, "unit": "Msamples/s", "cores": cores, "kind": "reference",
            "sample": "%d channels x %d frames of %d samples, reference v29_rx on %d host threads, %.1f s"
                      % (n_ch, n_frames, FRAME, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["v29"], default="v29")
    ap.add_argument("--channels", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-channels", type=int, default=2048)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device; the engine has no CPU fallback")
    dev = torch.device("cuda", 0)
    from spandsp_amd import engine
    n_ch = args.channels
    nf = args.steps + args.warmup
    frames = synth_v29(n_ch, nf, dev, seed=0x29290000)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    bank = engine.V29Bank(n_ch, 9600)
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
    alg_read = n_ch*(FRAME*2 + 281*4)
    alg_write = n_ch*(281*4 + 4 + 208)
    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_v29(frames[:, :min(args.cpu_channels, n_ch)].contiguous().cpu().numpy())
    value = args.steps*n_ch*FRAME/dt/1e6
    print(json.dumps({
        "metric": "Msamples/s of batched V.29 9600 bps receive (8 kHz channels at real-time = value*1e6/8000)",
        "value": value, "unit": "Msamples/s", "realtime_channels": value*1e6/8000.0, "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt*1e3/args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: V.29 9600 bps RX, %d channels x %d-sample frames, AWGN line model"
                               % (n_ch, FRAME), "channels_per_gpu": n_ch,
                   "sampled_channels_in_data_mode_at_end": "%d of %d" % (trained, len(range(0, n_ch, max(1, n_ch//256)))),
                   "events_in_last_frame": bits_last},
        "roofline": {"bound": "hbm", "kernel": "v29_bank_kernel", "achieved": alg_read/(avg_ms*1e-3)/1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_read/(avg_ms*1e-3)/1e9/HBM_PEAK_GBPS,
                     "traffic": None, "alg_read_bytes_per_launch": alg_read, "alg_write_bytes_per_launch": alg_write,
                     "avg_launch_us": avg_ms*1e3, "min_launch_us": min(per)*1e3, "max_launch_us": max(per)*1e3,
                     "note": "VALU/LDS-issue bound state machine (SURVEY 8(d)); the HBM figure is reported, not targeted"},
        "cpu_baseline": cpu}))


if __name__ == "__main__":
    main()
