"""Where the host's time goes in a pipelined tick: wall time of slot() / commit() / collect() (tools/e2e_probe.py's setup)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth
from spandsp_amd import engine
n_ch = 65536
sig, _ = synth.dtmf_channels(256, 160*8, seed=5)
host = torch.tensor(sig.reshape(256, 8, 160).transpose(1, 0, 2).copy()).repeat(1, n_ch//256, 1).numpy()
bank = engine.ToneBank(engine.DTMF, n_ch)
feed = engine.Feed(bank, 160, law=0, depth=3)
LAG = int(os.environ.get("LAG", "1"))       # ticks between a commit and the collect of its digits
for i in range(3):
    buf = feed.slot(); buf[:, :160] = host[i % 8]; feed.commit(160)
    if i >= LAG:
        feed.collect()
ts = {"slot": 0.0, "commit": 0.0, "collect": 0.0}
n = 60
t_all = time.perf_counter()
for i in range(n):
    t0 = time.perf_counter(); feed.slot(); t1 = time.perf_counter(); feed.commit(160); t2 = time.perf_counter(); feed.collect(); t3 = time.perf_counter()
    ts["slot"] += t1 - t0; ts["commit"] += t2 - t1; ts["collect"] += t3 - t2
t_all = time.perf_counter() - t_all
print(json.dumps({k: v/n*1e3 for k, v in ts.items()}), "ms per tick; loop", t_all/n*1e3)
while feed.collect() is not None:
    pass
for lag in (1, 2):
    ms, d = feed.run(160, 60, lag)
    print("C loop, lag", lag, ms/60, "ms per tick,", d, "digits")
