#!/usr/bin/env python3
"""Builder's instrument: phase timing of the four-lanes-per-channel receiver kernels.  Needs a library built with
`make -C spandsp_amd/csrc EXTRA=-DSPG_QUAD_PROF` (never the product build).  Prints cycles per wave per round phase."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.argv = [sys.argv[0]] + sys.argv[1:]
import tools.bench_paths as bp  # noqa: E402
from spandsp_amd import engine  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "v29"
n_ch = 16384
steps, warm = 40, 110
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
frames, _, _ = bp.synth_v29_on_device(n_ch, steps + warm, dev, stream, seed=0x2929, modem=workload, line="in_step")
fixture, bit_rate, n_words = bp.MODEMS[workload]
kind = {"v29": engine.V29, "v17": engine.V17, "v27ter": engine.V27TER}[workload]
engine.tune_modem_mapping(4)
bank = engine.ModemBank(kind, n_ch, bit_rate)
bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
fb = n_ch*160*2
L = engine.lib()
L.spangpu_debug_quad_prof.restype = ctypes.c_int
out = (ctypes.c_ulonglong*16)()
for i in range(warm):
    bank.rx_device(ctypes.c_void_p(frames.data_ptr() + i*fb), 160, 160)
torch.cuda.synchronize()
L.spangpu_debug_quad_prof(out)
for i in range(steps):
    bank.rx_device(ctypes.c_void_p(frames.data_ptr() + (warm + i)*fb), 160, 160)
torch.cuda.synchronize()
L.spangpu_debug_quad_prof(out)
v = np.array(list(out)[:10], np.float64)
waves = (n_ch + 15)//16
per = v/steps/waves
names = ["loop top / tail", "pre-reads + candidates", "plan x4 + commit", "RRC", "post x4 (Godard)", "T/2", "Godard baud + EQ", "decode + stage + track", "LMS", "save + carrier"]
tot = per.sum()
for n, x in zip(names, per):
    print("%-26s %9.0f cycles per wave-frame  %5.1f %%   %7.1f per sample" % (n, x, 100*x/tot, x/160))
print("total %.0f cycles per wave-frame = %.1f us at 2.4 GHz" % (tot, tot/2400.0))
