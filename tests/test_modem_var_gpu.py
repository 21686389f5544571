"""Modem receiver banks in a tick where not every receiver has a frame (spangpu_modem_rx_var(), and the group calls of the
spandsp-named shim on top of it): a receiver without a frame is untouched, one with a short frame advances by just that;
every channel equals an oracle receiver fed its own samples only -- event stream and every state word."""
import ctypes as C
import os

import numpy as np
import pytest

from test_oracle_pin import GOLDEN, bits, use_golden_modem_tables

pytestmark = pytest.mark.gpu

KINDS = [("v29", 9600), ("v27ter", 4800), ("v17", 14400)]


def _signals(name, rate, n_ch, seed):
    base = np.load(os.path.join(GOLDEN, "%s_%d.npz" % (name, rate)))["amp"].astype(np.float64)
    rng = np.random.default_rng(seed)
    n = len(base) + 64
    out = np.zeros((n_ch, n), np.int16)
    for c in range(n_ch):
        delay = int(rng.integers(0, 64))
        gain = 10.0**(rng.uniform(-10.0, 3.0)/20.0) if c else 1.0
        noise = rng.normal(0.0, rng.choice([0.0, 3.0, 20.0]), n) if c else 0.0
        x = np.zeros(n)
        x[delay:delay + len(base)] = base
        out[c] = np.clip(np.rint(x*gain + noise), -32768, 32767).astype(np.int16)
    return out


def _make(name, rate, n_ch):
    from oracle import restated as orc
    from spandsp_amd import engine
    bank = {"v29": engine.V29Bank, "v27ter": engine.V27terBank, "v17": engine.V17Bank}[name](n_ch, rate)
    ocls = {"v29": orc.V29, "v27ter": orc.V27ter, "v17": orc.V17}[name]
    return bank, [ocls(rate) for _ in range(n_ch)]


def _lens(rng, n_ch, tick, left):
    r = rng.random(n_ch)
    lens = np.where(r < 0.2, 0, 160).astype(np.int32)
    if tick % 3 == 1:
        short = rng.random(n_ch) < 0.2
        lens[short] = rng.integers(1, 160, int(short.sum()))
    if tick % 9 == 4:
        lens[16:32] = 0                 # a whole wave of the 16-channel kernels sits the tick out
    return np.minimum(lens, left).astype(np.int32)


@pytest.mark.parametrize("name,rate", KINDS)
def test_bank_tick_with_missing_and_short_channels(built, name, rate):
    use_golden_modem_tables()
    n_ch = 50
    sig = _signals(name, rate, n_ch, seed=rate)
    bank, dets = _make(name, rate, n_ch)
    rng = np.random.default_rng(7)
    pos = np.zeros(n_ch, np.int64)
    total = 0
    tick = 0
    while (pos < sig.shape[1]).any() and tick < 400:
        lens = _lens(rng, n_ch, tick, sig.shape[1] - pos)
        frames = rng.integers(-9000, 9000, (n_ch, 160)).astype(np.int16)        # beyond lens[c] a row is never read
        for c in range(n_ch):
            frames[c, :lens[c]] = sig[c, pos[c]:pos[c] + lens[c]]
        bank.rx_host_var(frames, lens)
        got = bank.events() if lens.any() else [np.zeros(0, np.int8)]*n_ch
        for c in range(n_ch):
            d = dets[c]
            d.sink.clear()
            if lens[c]:
                d.rx(sig[c, pos[c]:pos[c] + lens[c]])
            want = d.sink.events()["a"].astype(np.int8)
            assert np.array_equal(got[c], want), (name, "events", c, tick)
            total += len(want)
        pos += lens
        tick += 1
    assert total > 100*n_ch
    for c in range(n_ch):
        f, w = bank.get_state(c)
        fo, wo = dets[c].snapshot()
        assert np.array_equal(w, wo), (name, "int words", c, np.nonzero(w != wo)[0][:8])
        assert np.array_equal(bits(f), bits(fo)), (name, "float words", c)
    bank.close()


PUT_BIT = C.CFUNCTYPE(None, C.c_void_p, C.c_int)


def test_group_tick_with_late_receivers_and_threads(built):
    """The shim's receiver group: the owner of the tick flushes at its deadline with whoever staged; staging is safe from
    several threads; a second frame before the tick ran is refused."""
    import threading
    from oracle import restated as orc
    from spandsp_amd import engine
    use_golden_modem_tables()
    L = C.CDLL(engine.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    L.spangpu_modem_group_create.restype = vp
    L.spangpu_modem_group_create.argtypes = [ci, ci, ci, ci, ci]
    L.spangpu_modem_group_destroy.argtypes = [vp]
    L.spangpu_modem_group_flush.argtypes = [vp]
    L.spangpu_v27ter_rx_attach.restype = vp
    L.spangpu_v27ter_rx_attach.argtypes = [vp, ci, PUT_BIT, vp]
    L.v27ter_rx.argtypes = [vp, vp, ci]
    L.v27ter_rx_free.argtypes = [vp]
    n_thr, per = 4, 6
    n_ch = n_thr*per
    sig = _signals("v27ter", 4800, n_ch, seed=3)
    n_ticks = sig.shape[1]//160
    grp = L.spangpu_modem_group_create(0, engine.V27TER, n_ch, 4800, 160)
    got = [[] for _ in range(n_ch)]
    cbs = [PUT_BIT(lambda u, b, c=c: got[c].append(b)) for c in range(n_ch)]
    objs = [L.spangpu_v27ter_rx_attach(grp, c, cbs[c], None) for c in range(n_ch)]
    assert all(objs)
    rng = np.random.default_rng(11)
    plan = rng.random((n_ticks, n_ch)) >= 0.25          # who has a frame in which tick
    plan[5] = True                                      # a full house: its last stager runs the tick
    pos = np.zeros(n_ch, np.int64)
    frames = [[None]*n_ch for _ in range(n_ticks)]
    for k in range(n_ticks):
        for c in range(n_ch):
            if plan[k, c]:
                frames[k][c] = np.ascontiguousarray(sig[c, pos[c]:pos[c] + 160])
                pos[c] += 160
    gate = threading.Barrier(n_thr + 1)
    errors = []

    def worker(w):
        try:
            for k in range(n_ticks):
                for c in range(w*per, (w + 1)*per):
                    if frames[k][c] is not None and L.v27ter_rx(objs[c], frames[k][c].ctypes.data, 160) != 0:
                        errors.append((c, k))
                gate.wait()             # everybody has staged
                gate.wait()             # the owner has run the tick
        except Exception as e:          # pragma: no cover
            errors.append(repr(e))
            gate.abort()

    th = [threading.Thread(target=worker, args=(w,)) for w in range(n_thr)]
    for x in th:
        x.start()
    for k in range(n_ticks):
        gate.wait()
        n_staged = int(plan[k].sum())
        if 0 < n_staged < n_ch:
            c = int(np.flatnonzero(plan[k])[0])
            assert L.v27ter_rx(objs[c], frames[k][c].ctypes.data, 160) == -1
        assert L.spangpu_modem_group_flush(grp) == (0 if n_staged == n_ch else n_staged)
        gate.wait()
    for x in th:
        x.join()
    assert errors == []
    n_bits = 0
    for c in range(n_ch):
        o = orc.V27ter(4800)
        for k in range(n_ticks):
            if frames[k][c] is not None:
                o.rx(frames[k][c])
        want = [int(v) for v in o.sink.events()["a"]]
        assert got[c] == want, c
        n_bits += len(want)
        L.v27ter_rx_free(objs[c])
    assert n_bits > 100*n_ch
    L.spangpu_modem_group_destroy(grp)


def _var_lens(rng, n_ch, tick, left):
    r = rng.random(n_ch)
    lens = np.where(r < 0.2, 0, 160).astype(np.int32)
    if tick % 3 == 1:
        short = rng.random(n_ch) < 0.2
        lens[short] = rng.integers(1, 160, int(short.sum()))
    if tick % 9 == 4:
        lens[:64] = 0                   # a whole wave sits the tick out
    return np.minimum(lens, left).astype(np.int32)


def test_fsk_bank_tick_with_missing_and_short_channels(built):
    """spangpu_fsk_rx_var(): V.21 channel 2 receivers, every channel against an oracle receiver fed its own samples only."""
    import synth
    from oracle import restated as orc
    from spandsp_amd import engine
    which, mode, n_ch = 1, 1, 100
    sp = engine.fsk_preset(which)
    sig = synth.fsk_channels(n_ch, 160*90, 321, sp.freq_zero, sp.freq_one, sp.baud_rate)
    bank = engine.FskBank(which, n_ch, mode)
    dets = [orc.Fsk(which, mode) for _ in range(n_ch)]
    rng = np.random.default_rng(8)
    pos = np.zeros(n_ch, np.int64)
    got = [[] for _ in range(n_ch)]
    tick = 0
    while (pos < sig.shape[1]).any() and tick < 300:
        lens = _var_lens(rng, n_ch, tick, sig.shape[1] - pos)
        frames = rng.integers(-9000, 9000, (n_ch, 160)).astype(np.int16)
        for c in range(n_ch):
            frames[c, :lens[c]] = sig[c, pos[c]:pos[c] + lens[c]]
        bank.rx_host_var(frames, lens)
        if lens.any():
            ev = bank.events()
            for c in range(n_ch):
                if lens[c]:
                    got[c].append(ev[c])
                    dets[c].rx(sig[c, pos[c]:pos[c] + lens[c]])
                else:
                    assert len(ev[c]) == 0, (c, tick)
        pos += lens
        tick += 1
    total = 0
    for c, o in enumerate(dets):
        want = np.array([e["a"] for e in o.sink.events() if e["kind"] == 3], np.int64)
        have = np.concatenate(got[c]).astype(np.int64) if got[c] else np.zeros(0, np.int64)
        assert np.array_equal(have, want), (c, len(have), len(want))
        assert np.array_equal(bank.get_state(c), o.snapshot()), c
        total += len(want)
    assert total > 20*n_ch


def test_mct_bank_tick_with_missing_and_short_channels(built):
    """spangpu_mct_rx_var(): CED-or-preamble detectors (the 2100 Hz detector and the V.21 receiver in one bank)."""
    import synth
    from oracle import restated as orc
    from spandsp_amd import engine
    tone_type, n_ch = 7, 80
    sig = synth.connect_tone_channels(n_ch, 8000*4, 507, "mix")
    bank = engine.MctBank(tone_type, n_ch)
    dets = [orc.Mct(tone_type) for _ in range(n_ch)]
    rng = np.random.default_rng(9)
    pos = np.zeros(n_ch, np.int64)
    got = [[] for _ in range(n_ch)]
    tick = 0
    while (pos < sig.shape[1]).any() and tick < 400:
        lens = _var_lens(rng, n_ch, tick, sig.shape[1] - pos)
        frames = rng.integers(-9000, 9000, (n_ch, 160)).astype(np.int16)
        for c in range(n_ch):
            frames[c, :lens[c]] = sig[c, pos[c]:pos[c] + lens[c]]
        bank.rx_host_var(frames, lens)
        if lens.any():
            for c, e in enumerate(bank.events()):
                if lens[c]:
                    got[c].extend((int(t), int(lv)) for t, lv in e)
                    dets[c].rx(sig[c, pos[c]:pos[c] + lens[c]])
                else:
                    assert len(e) == 0, (c, tick)
        pos += lens
        tick += 1
    seen = set()
    for c, o in enumerate(dets):
        want = [(int(e["a"]), int(e["b"])) for e in o.sink.events() if e["kind"] == 1]
        assert got[c] == want, (c, got[c][:4], want[:4])
        assert np.array_equal(bank.get_state(c), o.snapshot()), c
        seen.update(t for t, _ in want)
    assert len(seen - {0}) >= 1


def test_fsk_group_tick_with_late_receivers(built):
    """The shim's FSK receiver group: a late receiver stalls nobody, a second frame before the tick ran is refused."""
    import synth
    from oracle import restated as orc
    from spandsp_amd import engine
    L = C.CDLL(engine.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    spec = engine.fsk_preset(1)
    L.spangpu_fsk_group_create.restype = vp
    L.spangpu_fsk_group_create.argtypes = [ci, vp, ci, ci, ci]
    L.spangpu_line_group_flush.argtypes = [vp]
    L.spangpu_line_group_destroy.argtypes = [vp]
    L.spangpu_fsk_rx_attach.restype = vp
    L.spangpu_fsk_rx_attach.argtypes = [vp, ci, PUT_BIT, vp]
    L.fsk_rx.argtypes = [vp, vp, ci]
    L.fsk_rx_free.argtypes = [vp]
    n_ch = 20
    sig = synth.fsk_channels(n_ch, 160*70, 77, spec.freq_zero, spec.freq_one, spec.baud_rate)
    # the spec struct of the shim: spandsp's fsk_spec_t {name, freq_zero, freq_one, tx_level, min_level, baud_rate}
    class FskSpec(C.Structure):
        _fields_ = [("name", C.c_char_p), ("freq_zero", ci), ("freq_one", ci), ("tx_level", ci), ("min_level", ci), ("baud_rate", ci)]
    sp = FskSpec(b"V21 ch 2", spec.freq_zero, spec.freq_one, spec.tx_level, spec.min_level, spec.baud_rate)
    grp = L.spangpu_fsk_group_create(0, C.byref(sp), 1, n_ch, 160)
    assert grp
    got = [[] for _ in range(n_ch)]
    cbs = [PUT_BIT(lambda u, b, c=c: got[c].append(b)) for c in range(n_ch)]
    objs = [L.spangpu_fsk_rx_attach(grp, c, cbs[c], None) for c in range(n_ch)]
    assert all(objs)
    dets = [orc.Fsk(1, 1) for _ in range(n_ch)]
    rng = np.random.default_rng(12)
    pos = np.zeros(n_ch, np.int64)
    for tick in range(90):
        lens = np.where(rng.random(n_ch) < 0.25, 0, np.where(rng.random(n_ch) < 0.2, 80, 160))
        lens = np.minimum(lens, sig.shape[1] - pos)
        staged = 0
        for c in range(n_ch):
            if lens[c]:
                fr = np.ascontiguousarray(sig[c, pos[c]:pos[c] + lens[c]])
                assert L.fsk_rx(objs[c], fr.ctypes.data, int(lens[c])) == 0
                dets[c].rx(fr)
                staged += 1
        if 0 < staged < n_ch:
            c = int(np.flatnonzero(lens)[0])
            fr = np.ascontiguousarray(sig[c, :160])
            assert L.fsk_rx(objs[c], fr.ctypes.data, 160) == -1
        assert L.spangpu_line_group_flush(grp) == (0 if staged == n_ch else staged)
        pos += lens
    n_bits = 0
    for c in range(n_ch):
        want = [int(e["a"]) for e in dets[c].sink.events() if e["kind"] == 3]
        assert got[c] == want, c
        n_bits += len(want)
        L.fsk_rx_free(objs[c])
    assert n_bits > 20*n_ch
    L.spangpu_line_group_destroy(grp)
