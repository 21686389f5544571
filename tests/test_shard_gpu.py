"""spangpu_shard_*: one logical DTMF bank over several devices behind the C ABI.  A one-GPU box names its
device twice -- two, then five shards with a stream each on the same GPU: the digit bytes gathered to the collecting device
equal, tick for tick, the digits of a single bank fed the same channels (itself held to the oracle in test_tone_gpu.py), and
the oracle's dtmf_rx() digit strings on a sample of channels at the end."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shards", [2, 5])
def test_sharded_bank_equals_one_bank(built, shards):
    import ctypes
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch, frame, ticks = 3001, 160, 45
    sig, _ = synth.dtmf_channels(n_ch, frame*ticks, seed=404)
    one = engine.ToneBank(engine.DTMF, n_ch)
    sh = engine.ShardedToneBank(engine.DTMF, n_ch, [0]*shards, max_samples=frame)
    assert sh.shards == shards and sh.ranges[0][1] == 0 and sum(r[2] for r in sh.ranges) == n_ch
    assert all(sh.ranges[i][1] + sh.ranges[i][2] == sh.ranges[i + 1][1] for i in range(shards - 1))
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    bufs = []
    for _, f, n in sh.ranges:
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), n*frame*2) == 0
        bufs.append(p)
    strings = [""]*n_ch
    for t in range(ticks):
        fr = np.ascontiguousarray(sig[:, t*frame:(t + 1)*frame])
        one.rx_host(fr)
        want = np.zeros((2, n_ch), np.uint8)
        for r in one.blocks():
            if (r["flags"] & engine.BLK_CHANGE) and r["code"]:
                want[int(r["block"]), int(r["channel"])] = int(r["code"])
        sh.sync()                                               # (the rows below are overwritten: the last step has read them)
        for (_, f, n), p in zip(sh.ranges, bufs):
            part = np.ascontiguousarray(fr[f:f + n])
            assert hip.hipMemcpy(p, part.ctypes.data, part.nbytes, 1) == 0
        nb = sh.rx_device([p.value for p in bufs], frame, frame)
        got = sh.digits_host()
        assert got.shape == (nb, n_ch) and np.array_equal(got, want[:nb]), t
        for b in range(nb):
            for c in np.nonzero(got[b])[0]:
                strings[c] += chr(int(got[b, c]))
    sh.sync()
    for c in list(range(0, n_ch, 97)) + [n_ch - 1]:
        o = orc.Dtmf(0)
        o.rx(sig[c])
        assert strings[c] == o.get(), c
    assert sum(len(x) for x in strings) > n_ch
    sh.close()
    one.close()
    for p in bufs:
        hip.hipFree(p)
