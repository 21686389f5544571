"""The pipelined host path alone (bench.py's end_to_end()), for experiments with the feed: python tools/e2e_probe.py [law]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import synth
from spandsp_amd import engine
law = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_ch = 65536
sig, _ = synth.dtmf_channels(256, 160*8, seed=5)
fr = torch.tensor(sig.reshape(256, 8, 160).transpose(1, 0, 2).copy()).repeat(1, n_ch//256, 1).cuda()
if law:
    fr = (fr.to(torch.int32) & 0xFF).to(torch.uint8)
r = bench.end_to_end(engine, n_ch, fr, 0, 60, law=law)
print(json.dumps({k: r[k] for k in ("ms_per_step", "h2d_copy_alone_ms", "ms_per_step_with_fill")}), "piece", os.environ.get("SPANGPU_FEED_PIECE"))
