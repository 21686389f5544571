#!/usr/bin/env python3
"""What the pieces of the cadence matcher cost a super-tone launch (65 536 lines x 160, the lines of tools/bench_paths.py
--workload supertone): the same bank with the whole plan, without segment reports, with one-element tones only, with no tones."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
import test_cadence_gpu as tc  # noqa: E402
from spandsp_amd import engine  # noqa: E402

FRAME = 160
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
n_ch, nf, n_src = 65536, 100, 512
src = torch.tensor(synth.cadence_plan_channels(n_src, nf*FRAME, 81, tc.PLANS), device=dev).view(n_src, nf, FRAME)
frames = src[torch.arange(n_ch, device=dev) % n_src].permute(1, 0, 2).contiguous()
hz = [400, 1100, 350, 440, 480, 620, 950, 1400, 1800]
bins = {0: -1}
bins.update({f: i for i, f in enumerate(hz)})
tones = [[(bins[f1], bins[f2], lo, hi) for f1, f2, lo, hi in t] for t in tc.TONES]
addr = [ctypes.c_void_p(frames.data_ptr() + f*n_ch*FRAME*2) for f in range(nf)]


def run(label, plan, segments):
    bank = engine.ToneBank(engine.SUPER_TONE, n_ch, bin_fac=[engine.goertzel_fac(float(f)) for f in hz])
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    if plan is not None:
        bank.set_cadences(plan, want_segments=segments)
    for i in range(100):
        bank.rx_device(addr[i % nf], FRAME, FRAME)
        if plan is not None:
            bank.cadence_run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    n = 2000
    for i in range(n):
        bank.rx_device(addr[(100 + i) % nf], FRAME, FRAME)
        if plan is not None:
            bank.cadence_run()
    e1.record(stream)
    torch.cuda.synchronize()
    print("%-44s %.2f us a launch" % (label, e0.elapsed_time(e1)*1e3/n))
    bank.close()


for rep in range(2):
    run("detector alone", None, False)
    run("six tones, segment reports", tones, True)
    run("six tones, no segment reports", tones, False)
    run("the three one-element tones, segments", [t for t in tones if len(t) == 1], True)
    run("one tone that never fits (1800 Hz >= 60 s)", [[(8, -1, 60000, 0)]], False)
