import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from spandsp_amd import engine
from oracle import restated as orc
from test_echo_gpu import make_channels
taps, mode = 32, 0x61
n_ch = 37
tx, rx = make_channels(n_ch, 160*150, taps, seed=taps + mode)
for frame in (160, 1, 2, 3, 40, 41):
    engine.lib().spangpu_tune_echo_lanes_per_channel(2)
    bank = engine.EchoBank(n_ch, taps, mode)
    dets = [orc.EchoCan(taps, mode) for _ in range(n_ch)]
    pos = 0
    bad = None
    while pos < 160*3 and bad is None:
        got = bank.update_host(tx[:, pos:pos + frame], rx[:, pos:pos + frame], use_hpf_tx=True)
        for c, d in enumerate(dets):
            want = d.run(tx[c, pos:pos + frame], rx[c, pos:pos + frame], True)
            sn = d.snapshot()
            h = bank.get_state(c)["history"]
            if not np.array_equal(got[c], want) or not np.array_equal(h, sn["history"]):
                bad = (pos, c, np.nonzero(got[c] != want)[0][:4], np.nonzero(h != sn["history"])[0][:8])
                break
        pos += frame
    print("frame", frame, "first bad", bad, flush=True)
