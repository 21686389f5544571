"""spangpu_shard_*: one logical DTMF bank over several devices behind the C ABI.  A one-GPU box names its
device twice -- two, then five shards with a stream each on the same GPU: the digit bytes gathered to the collecting device
equal, tick for tick, the digits of a single bank fed the same channels (itself held to the oracle in test_tone_gpu.py), and
the oracle's dtmf_rx() digit strings on a sample of channels at the end."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1], ids=["plain copies", "hipMemcpyPeerAsync forced"], autouse=True)
def peer_copy(request, built):
    """Every test of this file twice: as a one-GPU box runs it (a shard on the collecting device copies with hipMemcpyAsync), and
    with spangpu_tune_force_peer_copy(1): the same shards send their results with hipMemcpyPeerAsync, source and destination
    device equal -- the code path a second GPU takes (csrc/shard_api.hip gather_copy())."""
    from spandsp_amd import engine
    engine.tune_force_peer_copy(request.param)
    yield request.param
    assert engine.tune_force_peer_copy(0) == request.param


def check_info(sh, shards, peer_copy):
    from spandsp_amd import engine
    at = 0
    for i in range(shards):
        info = sh.info(i)
        assert (info.device, info.collect_device, info.link) == (0, 0, engine.LINK_SAME)
        assert info.first_channel == at and info.forced_peer_copy == peer_copy
        at += info.n_channels
    with pytest.raises(engine.SpanGpuError):
        sh.info(shards)
    return at


@pytest.mark.parametrize("shards", [2, 5])
def test_sharded_bank_equals_one_bank(built, shards, peer_copy):
    import ctypes
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch, frame, ticks = 3001, 160, 45
    sig, _ = synth.dtmf_channels(n_ch, frame*ticks, seed=404)
    one = engine.ToneBank(engine.DTMF, n_ch)
    sh = engine.ShardedToneBank(engine.DTMF, n_ch, [0]*shards, max_samples=frame)
    assert sh.shards == shards and sh.ranges[0][1] == 0 and sum(r[2] for r in sh.ranges) == n_ch
    assert check_info(sh, shards, peer_copy) == n_ch
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


def _hip():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]

    def dev_alloc(nbytes):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), nbytes) == 0
        return p

    def h2d(p, a):
        a = np.ascontiguousarray(a)
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0

    def d2h(a, p):
        assert hip.hipMemcpy(a.ctypes.data, p, a.nbytes, 2) == 0
    return hip, dev_alloc, h2d, d2h


@pytest.mark.parametrize("shards", [2, 3])
def test_sharded_echo_bank_equals_one_bank(built, shards, peer_copy):
    """spangpu_echo_shard_* (BASELINE configs[4]'s object behind the C ABI) with a one-GPU box's device named two and three
    times: every clean sample and every gathered ERLE float equal those of a single bank fed the same lines (itself held to the
    oracle in test_echo_gpu.py), the ERLE of a sample of lines also against the oracle's clean samples, and a report stays whole
    while the next one is queued (two slots)."""
    from oracle import restated as orc
    from spandsp_amd import engine
    from test_echo_gpu import make_channels
    hip, dev_alloc, h2d, d2h = _hip()
    n_ch, frame, ticks, taps, mode = 700, 160, 60, 128, 0x01
    tx, rx = make_channels(n_ch, frame*ticks, taps, seed=515)
    one = engine.EchoBank(n_ch, taps, mode)
    one.stats(2)
    sh = engine.ShardedEchoBank(n_ch, taps, mode, [0]*shards)
    assert sh.shards == shards and sum(r[2] for r in sh.ranges) == n_ch and sh.ranges[0][1] == 0
    assert check_info(sh, shards, peer_copy) == n_ch
    bufs = [(dev_alloc(n*frame*2), dev_alloc(n*frame*2), dev_alloc(n*frame*2)) for _, _, n in sh.ranges]
    dets = {c: orc.EchoCan(taps, mode) for c in (0, 63, 64, 333, n_ch - 1)}
    sums = {c: [0, 0] for c in dets}
    first_report = None
    for t in range(ticks):
        a = np.ascontiguousarray(tx[:, t*frame:(t + 1)*frame])
        b = np.ascontiguousarray(rx[:, t*frame:(t + 1)*frame])
        want = one.update_host(a, b)
        sh.sync()
        for (_, f, n), (pt, pr, pc) in zip(sh.ranges, bufs):
            h2d(pt, a[f:f + n])
            h2d(pr, b[f:f + n])
        sh.update_device([x[0].value for x in bufs], [x[1].value for x in bufs], [x[2].value for x in bufs], frame, frame)
        sh.sync()
        for (_, f, n), (pt, pr, pc) in zip(sh.ranges, bufs):
            got = np.zeros((n, frame), np.int16)
            d2h(got, pc)
            assert np.array_equal(got, want[f:f + n]), (t, f)
        for c, d in dets.items():
            cl = d.run(a[c], b[c], False)
            assert np.array_equal(cl, want[c]), (t, c)
            sums[c][0] += int((b[c].astype(np.int64)**2).sum())
            sums[c][1] += int((cl.astype(np.int64)**2).sum())
        if t == ticks//2 - 1 or t == ticks - 1:
            sh.report(reset=True)
            ref_erle = one.erle_host()
            one.stats_reset(sums=True, crc=False)
            if first_report is None:
                first_report = ref_erle
                # the next report is queued before this one is read: it lands in the other slot
            else:
                got = sh.erle_host()
                assert np.array_equal(got.view(np.uint32), ref_erle.view(np.uint32))
                for c, (srx, scl) in sums.items():
                    want_db = 10.0*np.log10(srx/scl) if scl and srx else (120.0 if srx else 0.0)
                    assert abs(float(got[c]) - want_db) < 1e-3, (c, got[c], want_db)
            if t == ticks//2 - 1:
                got = sh.erle_host()
                assert np.array_equal(got.view(np.uint32), first_report.view(np.uint32))
                for c in sums:
                    sums[c] = [0, 0]
    sh.close()
    one.close()
    for tup in bufs:
        for p in tup:
            hip.hipFree(p)


def test_sharded_modem_bank_equals_one_bank(built):
    """spangpu_modem_shard_*: V.29 receivers over "three devices" (this GPU named three times): the put_bit streams gathered per
    step are those of a single bank on the same lines, which the committed output of the real reference pins."""
    import os
    from spandsp_amd import engine
    from test_oracle_pin import GOLDEN
    hip, dev_alloc, h2d, d2h = _hip()
    g = np.load(os.path.join(GOLDEN, "v29_9600.npz"))
    x = g["amp"]
    n_ch, frame, per = 200, 160, 256            # (a put_bit call per bit: 192 a frame at 9600 bit/s, and the status reports)
    sh = engine.ShardedModemBank(engine.V29, n_ch, 9600, [0, 0, 0], per=per)
    assert sh.shards == 3 and sum(r[2] for r in sh.ranges) == n_ch
    assert sum(sh.info(i).n_channels for i in range(3)) == n_ch and sh.info(2).link == engine.LINK_SAME
    bufs = [dev_alloc(n*frame*2) for _, _, n in sh.ranges]
    streams = [[] for _ in range(n_ch)]
    for k in range(0, len(x) - frame + 1, frame):
        blk = x[k:k + frame]
        sh.sync()
        for (_, f, n), p in zip(sh.ranges, bufs):
            # a line's signal delayed by a few samples of silence per line group, so that the lines are not in step
            h2d(p, np.tile(blk, (n, 1)))
        sh.rx_device([p.value for p in bufs], frame, frame)
        counts, ev = sh.events_host()
        assert counts.max() <= per
        for c in (0, 63, 64, 127, 128, n_ch - 1):
            streams[c].append(ev[c, :counts[c]].copy())
    want = g["events"]
    for c in (0, 63, 64, 127, 128, n_ch - 1):
        got = np.concatenate(streams[c])
        assert np.array_equal(got, want[:len(got)]) and len(got) > 1000, (c, len(got), len(want))
    sh.close()
    for p in bufs:
        hip.hipFree(p)
