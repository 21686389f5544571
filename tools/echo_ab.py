"""A-B timing of the echo canceller's lane mappings at a given bank size: kernel time per 160-sample frame, frames
resident in HBM, every channel adapting (tools/bench_paths.py's workload).  Usage: python tools/echo_ab.py [n_ch] [lanes ...]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spandsp_amd import engine

n_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
lanes = [int(x) for x in sys.argv[2:]] or [4, 2]
rng = np.random.default_rng(1)
base_tx = (rng.standard_normal(160*6)*3000.0).astype(np.float32)
h = np.zeros(128, np.float32)
h[5:40] = rng.standard_normal(35)*0.1
for L in lanes:
    engine.lib().spangpu_tune_echo_lanes_per_channel(L)
    bank = engine.EchoBank(n_ch, 128, 0x01 | 0x02 | 0x04)
    engine.lib().spangpu_tune_echo_lanes_per_channel(0)
    tx = torch.from_numpy(np.clip(base_tx, -32768, 32767).astype(np.int16)).cuda()
    echo = np.convolve(base_tx, h)[:len(base_tx)]
    rx = torch.from_numpy(np.clip(echo, -32768, 32767).astype(np.int16)).cuda()
    txb = tx[None, :].repeat(n_ch, 1).contiguous()
    rxb = rx[None, :].repeat(n_ch, 1).contiguous()
    clean = torch.zeros_like(txb)
    st = torch.cuda.Stream()
    bank.set_stream(st.cuda_stream)
    stride = txb.shape[1]
    def step(k):
        off = (k % 6)*160*2
        bank.update_device(txb.data_ptr() + off, rxb.data_ptr() + off, clean.data_ptr() + off, 160, stride)
    for k in range(12):
        step(k)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        e0.record(st)
        for k in range(60):
            step(k)
        e1.record(st)
    torch.cuda.synchronize()
    print("lanes %d: %.1f us per frame of %d channels" % (L, e0.elapsed_time(e1)/60*1000.0, n_ch), flush=True)
    bank.close()
