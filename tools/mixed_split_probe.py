#!/usr/bin/env python3
"""configs[2] with the super-tone third cut in two banks (a deployment choice: banks are arbitrary groups of channels): does a tick
get shorter when the bank whose launch carries the cadence matcher -- the longest of the tick -- is two launches on two queues?"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
import bench_paths as bp  # noqa: E402
from spandsp_amd import engine  # noqa: E402

FRAME = 160
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
nf, n_src = 100, 512
srcs = [synth.bell_mf_channels(n_src, nf*FRAME, 21)[0], synth.r2_mf_channels(n_src, nf*FRAME, 22, True)[0],
        synth.cadence_plan_channels(n_src, nf*FRAME, 23, bp.MIXED_ST_LINES)]
fac = [engine.goertzel_fac(float(f)) for f in bp.MIXED_ST_FREQS]


def frames_for(kind, n):
    src = torch.tensor(srcs[kind], device=dev).view(n_src, nf, FRAME)
    idx = torch.arange(n, device=dev)
    fsel = (torch.arange(nf, device=dev).unsqueeze(0) + ((idx//n_src) % nf).unsqueeze(1)) % nf
    return src[(idx % n_src).unsqueeze(1), fsel].permute(1, 0, 2).contiguous()


def run(label, sizes, groups):
    """sizes: [(kind, channels)], groups: lists of bank indices that share a stream (and a launch)"""
    banks, frames = [], []
    for kind, n in sizes:
        b = engine.ToneBank(engine.BELL_MF, n) if kind == 0 else engine.ToneBank(engine.R2_MF, n, r2_fwd=True) if kind == 1 else engine.ToneBank(engine.SUPER_TONE, n, bin_fac=fac)
        if kind == 2:
            b.set_cadences(bp.mixed_st_cadences(), want_segments=True)
        banks.append(b)
        frames.append(frames_for(kind, n))
    heads = [banks[g[0]] for g in groups]
    proven = engine.banks_own_queues(heads)
    for g in groups:
        for i in g[1:]:
            banks[i].share_stream(banks[g[0]])
    own = [torch.cuda.ExternalStream(engine.lib().spangpu_bank_get_stream(b.h), device=dev) for b in heads]
    plan = engine.BanksPlan(banks)
    handles = [plan.frame([frames[k].data_ptr() + f*sizes[k][1]*FRAME*2 for k in range(len(banks))]) for f in range(nf)]
    for i in range(200):
        plan.rx(handles[i % nf], FRAME)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for s in own:
        s.wait_event(e0)
    n = 2000
    for i in range(n):
        plan.rx(handles[(200 + i) % nf], FRAME)
    for s in own:
        e = torch.cuda.Event()
        e.record(s)
        stream.wait_event(e)
    e1.record(stream)
    torch.cuda.synchronize()
    print("%-64s %.2f us a tick (queues proven %d)" % (label, e0.elapsed_time(e1)*1e3/n, proven))
    for b in banks:
        b.close()


A, B, C = 43690, 43690, 43692
for rep in range(2):
    run("Bell+R2 one launch | super-tone                (shipped)", [(0, A), (1, B), (2, C)], [[0, 1], [2]])
    run("Bell+R2 one launch | super-tone a | super-tone b", [(0, A), (1, B), (2, C//2), (2, C - C//2)], [[0, 1], [2], [3]])
    run("Bell | R2 | super-tone a | super-tone b          ", [(0, A), (1, B), (2, C//2), (2, C - C//2)], [[0], [1], [2], [3]])
    run("Bell + super-tone a? no: Bell | R2+... three queues", [(0, A), (1, B), (2, C)], [[0], [1], [2]])
