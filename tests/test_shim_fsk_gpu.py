"""GPU tests of the spandsp-named entry points of spandsp_amd/csrc/shim_fsk.c (include/spangpu_spandsp.h): what a
caller of fsk_rx() / modem_connect_tones_rx() / dtmf_tx() observes through its callbacks and return values must equal
what the reference delivers -- the committed reference outputs (tests/golden/fsk_*.npz, mct_*.npz, tx_sources via the
oracle) for private objects, and the oracle for grouped objects."""
import ctypes as C
import os

import numpy as np
import pytest

import synth
from test_oracle_pin import GOLDEN, use_golden_modem_tables

pytestmark = pytest.mark.gpu

PUT_BIT = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
STATUS = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
REPORT = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int)


class FskSpec(C.Structure):
    _fields_ = [("name", C.c_char_p), ("freq_zero", C.c_int), ("freq_one", C.c_int), ("tx_level", C.c_int),
                ("min_level", C.c_int), ("baud_rate", C.c_int)]


@pytest.fixture(scope="module")
def L(built):
    from spandsp_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    sig = {
        "fsk_rx_init": (vp, [vp, vp, ci, PUT_BIT, vp]), "spangpu_fsk_rx_attach": (vp, [vp, ci, PUT_BIT, vp]),
        "fsk_rx": (ci, [vp, vp, ci]), "fsk_rx_restart": (ci, [vp, vp, ci]), "fsk_rx_fillin": (ci, [vp, ci]),
        "fsk_rx_free": (ci, [vp]), "fsk_rx_set_put_bit": (None, [vp, PUT_BIT, vp]),
        "fsk_rx_set_modem_status_handler": (None, [vp, STATUS, vp]), "fsk_rx_set_signal_cutoff": (None, [vp, cf]),
        "fsk_rx_set_frame_parameters": (None, [vp, ci, ci, ci]), "fsk_rx_signal_power": (cf, [vp]),
        "fsk_rx_get_parity_errors": (ci, [vp, C.c_bool]), "fsk_rx_get_framing_errors": (ci, [vp, C.c_bool]),
        "spangpu_fsk_group_create": (vp, [ci, vp, ci, ci, ci]),
        "spangpu_modem_connect_tones_group_create": (vp, [ci, ci, ci, ci, ci]),
        "spangpu_line_group_destroy": (ci, [vp]), "spangpu_line_group_flush": (ci, [vp]),
        "modem_connect_tones_rx_init": (vp, [vp, ci, REPORT, vp]),
        "spangpu_modem_connect_tones_rx_attach": (vp, [vp, ci, REPORT, vp]),
        "modem_connect_tones_rx": (ci, [vp, vp, ci]), "modem_connect_tones_rx_get": (ci, [vp]),
        "modem_connect_tones_rx_free": (ci, [vp]), "modem_connect_tone_to_str": (C.c_char_p, [ci]),
        "dtmf_tx_init": (vp, [vp, vp, vp]), "dtmf_tx_put": (ci, [vp, C.c_char_p, ci]), "dtmf_tx": (ci, [vp, vp, ci]),
        "dtmf_tx_set_level": (None, [vp, ci, ci]), "dtmf_tx_set_timing": (None, [vp, ci, ci]), "dtmf_tx_free": (ci, [vp]),
    }
    for name, (res, args) in sig.items():
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
    return lib


def spec_ptr(L, which):
    arr = (FskSpec*11).in_dll(L, "preset_fsk_specs")
    return C.addressof(arr) + which*C.sizeof(FskSpec)


def feed(fn, s, x, chunk):
    x = np.ascontiguousarray(x, np.int16)
    assert fn(s, x.ctypes.data, 0) == 0                     # empty input: nothing happens
    for k in range(0, len(x), chunk):
        blk = x[k:k + chunk]
        assert fn(s, blk.ctypes.data, len(blk)) == 0


@pytest.mark.parametrize("which,mode", [(1, 1), (1, 0), (1, 2), (2, 0), (7, 2)])
def test_fsk_private_object_replays_reference_stream(L, which, mode):
    g = np.load(os.path.join(GOLDEN, "fsk_%d_%d.npz" % (which, mode)))
    ev = []
    put_bit = PUT_BIT(lambda u, b: ev.append(b))
    arr = (FskSpec*11).in_dll(L, "preset_fsk_specs")
    assert arr[1].name == b"V21 ch 2" and arr[1].freq_zero == 1850 and arr[10].baud_rate == 11000
    s = L.fsk_rx_init(None, spec_ptr(L, which), mode, put_bit, None)
    assert s
    feed(L.fsk_rx, s, g["amp"], 517)
    assert np.array_equal(np.array(ev, np.int32), g["events"])
    assert L.fsk_rx_signal_power(s) < -30.0                # the transmission has ended
    assert L.fsk_rx_get_framing_errors(s, True) >= 0 and L.fsk_rx_get_framing_errors(s, False) == 0
    # with a status handler the carrier reports leave the bit stream (fsk.c:343-349)
    ev2, st2 = [], []
    pb2 = PUT_BIT(lambda u, b: ev2.append(b))
    sh2 = STATUS(lambda u, v: st2.append(v))
    assert L.fsk_rx_restart(s, spec_ptr(L, which), mode) == 0
    L.fsk_rx_set_put_bit(s, pb2, None)
    L.fsk_rx_set_modem_status_handler(s, sh2, None)
    feed(L.fsk_rx, s, g["amp"], 160)
    want = g["events"]
    assert st2 == [int(v) for v in want if v in (-1, -2)]
    assert len(ev2) > 10 and all(v >= 0 for v in ev2)
    L.fsk_rx_free(s)


def test_fsk_group_equals_oracle(L):
    from oracle import restated as orc
    n, frames = 40, 60
    sig = synth.fsk_channels(n, 160*frames, 901, 1850, 1650, 30000)
    grp = L.spangpu_fsk_group_create(0, spec_ptr(L, 1), 1, n, 160)
    assert grp
    taps = [[] for _ in range(n)]
    cbs = [PUT_BIT(lambda u, b, t=taps[c]: t.append(b)) for c in range(n)]
    objs = [L.spangpu_fsk_rx_attach(grp, c, cbs[c], None) for c in range(n)]
    assert all(objs) and L.spangpu_fsk_rx_attach(grp, 0, cbs[0], None) is None
    for k in range(frames):
        for c in range(n):
            blk = np.ascontiguousarray(sig[c, k*160:(k + 1)*160])
            L.fsk_rx(objs[c], blk.ctypes.data, 160)
    for c in range(n):
        o = orc.Fsk(1, 1)
        o.rx(sig[c])
        assert taps[c] == [int(e["a"]) for e in o.sink.events()], c
    for o in objs:
        L.fsk_rx_free(o)
    L.spangpu_line_group_destroy(grp)


@pytest.mark.parametrize("rx_type,tx_kind", [(1, 1), (2, 3), (2, 4), (7, "preamble"), (7, 5), (9, 9)])
def test_mct_private_object_replays_reference_reports(L, rx_type, tx_kind):
    g = np.load(os.path.join(GOLDEN, "mct_%d_%s.npz" % (rx_type, tx_kind)))
    rep = []
    cb = REPORT(lambda u, tone, level, delay: rep.append((tone, level, delay)))
    s = L.modem_connect_tones_rx_init(None, rx_type, cb, None)
    assert s
    feed(L.modem_connect_tones_rx, s, g["amp"], 160)       # the committed reports were made with 160-sample calls
    assert rep == [tuple(int(v) for v in e) for e in g["events"]]
    L.modem_connect_tones_rx_free(s)
    # no callback: the hit latch
    s = L.modem_connect_tones_rx_init(None, rx_type, C.cast(None, REPORT), None)
    feed(L.modem_connect_tones_rx, s, g["amp"], 160)
    tones = [int(e[0]) for e in g["events"] if e[0] != 0]
    assert L.modem_connect_tones_rx_get(s) == tones[-1]
    assert L.modem_connect_tones_rx_get(s) == 0
    L.modem_connect_tones_rx_free(s)
    assert L.modem_connect_tone_to_str(7) == b"FAX CED or preamble" and L.modem_connect_tone_to_str(99) == b"???"


def test_mct_group_equals_oracle(L):
    from oracle import restated as orc
    n, frames = 32, 200
    sig = synth.connect_tone_channels(n, 160*frames, 902, "mix")
    grp = L.spangpu_modem_connect_tones_group_create(0, 7, 1, n, 160)
    assert grp
    taps = [[] for _ in range(n)]
    cbs = [REPORT(lambda u, tone, level, delay, t=taps[c]: t.append((tone, level))) for c in range(n)]
    objs = [L.spangpu_modem_connect_tones_rx_attach(grp, c, cbs[c], None) for c in range(n)]
    assert all(objs)
    for k in range(frames):
        for c in range(n):
            blk = np.ascontiguousarray(sig[c, k*160:(k + 1)*160])
            L.modem_connect_tones_rx(objs[c], blk.ctypes.data, 160)
    total = 0
    for c in range(n):
        o = orc.Mct(7)
        for k in range(frames):
            o.rx(sig[c, k*160:(k + 1)*160])
        want = [(int(e["a"]), int(e["b"])) for e in o.sink.events()]
        assert taps[c] == want, c
        total += len(want)
    assert total > n
    for o in objs:
        L.modem_connect_tones_rx_free(o)
    L.spangpu_line_group_destroy(grp)


def test_dtmf_tx_object_equals_oracle(L):
    from oracle import restated as orc
    use_golden_modem_tables()
    s = L.dtmf_tx_init(None, None, None)
    assert s
    o = orc.DtmfTx()
    L.dtmf_tx_set_level(s, -7, 3)
    o.set_level(-7, 3)
    L.dtmf_tx_set_timing(s, 40, 30)
    o.set_timing(40, 30)
    assert L.dtmf_tx_put(s, b"159D*0#x", -1) == o.put("159D*0#x")
    big = b"1"*125
    assert L.dtmf_tx_put(s, big, len(big)) == o.put(big)       # does not fit: the shortfall comes back, nothing is queued
    buf = np.zeros(1000, np.int16)
    for n in (160, 1, 333, 1000, 1000, 1000, 1000, 1000):
        got = L.dtmf_tx(s, buf.ctypes.data, n)
        want = o.tx(n)
        assert got == len(want) and np.array_equal(buf[:got], want)
    L.dtmf_tx_free(s)


def test_bell_and_r2_tx_objects_equal_oracle(L):
    from oracle import restated as orc
    vp, ci = C.c_void_p, C.c_int
    for name, res, args in [("bell_mf_tx_init", vp, [vp]), ("bell_mf_tx_put", ci, [vp, C.c_char_p, ci]), ("bell_mf_tx", ci, [vp, vp, ci]),
                            ("bell_mf_tx_free", ci, [vp]), ("r2_mf_tx_init", vp, [vp, C.c_bool]), ("r2_mf_tx_put", ci, [vp, C.c_char]),
                            ("r2_mf_tx", ci, [vp, vp, ci]), ("r2_mf_tx_free", ci, [vp])]:
        getattr(L, name).restype = res
        getattr(L, name).argtypes = args
    buf = np.zeros(2000, np.int16)
    s = L.bell_mf_tx_init(None)
    assert s
    o = orc.BellMfTx()
    assert L.bell_mf_tx_put(s, b"K1234567890S", -1) == o.put("K1234567890S")
    for n in (160, 1, 777, 2000, 2000, 2000, 2000, 2000, 2000):
        got = L.bell_mf_tx(s, buf.ctypes.data, n)
        want = o.tx(n)
        assert got == len(want) and np.array_equal(buf[:got], want)
    L.bell_mf_tx_free(s)
    for fwd in (True, False):
        s = L.r2_mf_tx_init(None, fwd)
        assert s
        o = orc.R2MfTx(fwd)
        for digit, n in ((b"5", 400), (b"F", 161), (b"\0", 80), (b"x", 40), (b"1", 333)):
            assert L.r2_mf_tx_put(s, digit) == o.put(digit) == 0
            got = L.r2_mf_tx(s, buf.ctypes.data, n)
            want = o.tx(n)
            assert got == len(want) and np.array_equal(buf[:got], want), (fwd, digit)
        L.r2_mf_tx_free(s)


def test_caller_storage_and_a_restart_to_another_modem(L):
    """fsk_rx_init(&state, ...) in the caller's storage (fsk.c:725-733) and fsk_rx_restart() to a different spec (fsk.c:670): an
    object of its own moves to a bank of the new spec and receives that modem's reference stream; an object on a shared
    bank cannot leave it.  modem_connect_tones_rx_init() likewise takes the caller's storage."""
    store = (C.c_char*256)()
    a, m_a = 1, 1
    b, m_b = 2, 0
    ga = np.load(os.path.join(GOLDEN, "fsk_%d_%d.npz" % (a, m_a)))
    gb = np.load(os.path.join(GOLDEN, "fsk_%d_%d.npz" % (b, m_b)))
    ev, st = [], []
    pb = PUT_BIT(lambda u, v: ev.append(v))
    sh = STATUS(lambda u, v: st.append(v))
    s = L.fsk_rx_init(C.addressof(store), spec_ptr(L, a), m_a, pb, None)
    assert s == C.addressof(store)
    L.fsk_rx_set_modem_status_handler(s, sh, None)
    feed(L.fsk_rx, s, ga["amp"], 160)
    want = ga["events"]
    assert st == [int(v) for v in want if v in (-1, -2)] and ev == [int(v) for v in want if v not in (-1, -2)]
    # the same object, another modem
    del ev[:], st[:]
    assert L.fsk_rx_restart(s, spec_ptr(L, b), m_b) == 0
    feed(L.fsk_rx, s, gb["amp"], 160)
    want = gb["events"]
    assert st == [int(v) for v in want if v in (-1, -2)] and ev == [int(v) for v in want if v not in (-1, -2)]
    L.fsk_rx_release.restype = C.c_int
    L.fsk_rx_release.argtypes = [C.c_void_p]
    assert L.fsk_rx_release(s) == 0
    # on a shared bank the spec is the bank's
    grp = L.spangpu_fsk_group_create(0, spec_ptr(L, a), m_a, 4, 160)
    o = L.spangpu_fsk_rx_attach(grp, 2, pb, None)
    assert o and L.fsk_rx_restart(o, spec_ptr(L, b), m_b) == -1 and L.fsk_rx_restart(o, spec_ptr(L, a), m_a) == 0
    L.fsk_rx_free(o)
    L.spangpu_line_group_destroy(grp)
    # connect tones in the caller's storage
    g = np.load(os.path.join(GOLDEN, "mct_2_3.npz"))
    store2 = (C.c_char*256)()
    rep = []
    cb = REPORT(lambda u, tone, level, delay: rep.append((tone, level, delay)))
    m = L.modem_connect_tones_rx_init(C.addressof(store2), 2, cb, None)
    assert m == C.addressof(store2)
    if g is not None:
        feed(L.modem_connect_tones_rx, m, g["amp"], 160)
        assert len(rep) > 0
    L.modem_connect_tones_rx_release.restype = C.c_int
    L.modem_connect_tones_rx_release.argtypes = [C.c_void_p]
    assert L.modem_connect_tones_rx_release(m) == 0


def test_dtmf_tx_asks_for_more_digits(L):
    """dtmf_tx_init() with a digits callback (dtmf.c:566-573, 636): whenever the queue runs dry with room left in the buffer
    the sender calls back; what the callback puts is sent in the same call, seamlessly -- the samples are those of a sender
    that was given all the digits up front."""
    from oracle import restated as orc
    CB = C.CFUNCTYPE(None, C.c_void_p)
    L.dtmf_tx_init.argtypes = [C.c_void_p, CB, C.c_void_p]
    todo = [b"12", b"3", b"", b"45"]
    calls = []
    holder = {}

    def more(user):
        calls.append(len(calls))
        if todo:
            d = todo.pop(0)
            if d:
                assert L.dtmf_tx_put(holder["s"], d, -1) == 0
    cb = CB(more)
    store = (C.c_char*128)()
    s = L.dtmf_tx_init(C.addressof(store), cb, None)
    assert s == C.addressof(store)
    holder["s"] = s
    o = orc.DtmfTx()
    buf = np.zeros(8000, np.int16)
    out = []
    for n in (4000, 4000, 8000):
        got = L.dtmf_tx(s, buf.ctypes.data, n)
        out.append(buf[:got].copy())
    got_all = np.concatenate(out)
    # the reference sender given the same digits at the moments its callback would have supplied them
    o.put("12")
    o.put("3")
    w1 = o.tx(4000)
    w2 = o.tx(4000)
    o.put("45")
    w3 = o.tx(8000)
    want = np.concatenate([w1, w2, w3])
    assert len(calls) >= 4
    assert np.array_equal(got_all, want), (len(got_all), len(want))
    L.dtmf_tx_release = L.dtmf_tx_release
    L.dtmf_tx_release.restype = C.c_int
    L.dtmf_tx_release.argtypes = [C.c_void_p]
    assert L.dtmf_tx_release(s) == 0
    L.dtmf_tx_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]


def test_frames_staged_from_inside_callbacks_run(L):
    """A put_bit handler that feeds the receivers their next frame: frames staged while a tick's callbacks
    are being made complete the next tick, and that tick runs when the callbacks are over -- it does not wait for a flush
    nobody would call, and the fsk_rx() calls that follow are not refused as second frames.  The receivers see every sample
    once, in order: the bit streams are the oracle's.  A private object refuses the nested call (its one staging row is in
    use) instead of losing it."""
    from oracle import restated as orc
    n, frames = 3, 40
    sig = synth.fsk_channels(n, 160*frames, 977, 1850, 1650, 30000)
    grp = L.spangpu_fsk_group_create(0, spec_ptr(L, 1), 1, n, 160)
    taps = [[] for _ in range(n)]
    objs = [None]*n
    shots = {5: None, 17: None}     # ticks whose first callback stages the following frame for everybody
    rcs = []
    state = {"tick": -1, "fired": set()}

    def stage(k):
        for d in range(n):
            blk = np.ascontiguousarray(sig[d, k*160:(k + 1)*160])
            rcs.append(L.fsk_rx(objs[d], blk.ctypes.data, 160))

    def on_bit(c, b):
        taps[c].append(b)
        t = state["tick"]
        if t in shots and t not in state["fired"]:
            state["fired"].add(t)
            state["tick"] = t + 1
            stage(t + 1)            # from inside the callback: completes tick t + 1, which must run when this tick's callbacks are over
    cbs = [PUT_BIT(lambda u, b, c=c: on_bit(c, b)) for c in range(n)]
    for c in range(n):
        objs[c] = L.spangpu_fsk_rx_attach(grp, c, cbs[c], None)
    k = 0
    while k < frames:
        state["tick"] = k
        stage(k)
        k = state["tick"] + 1       # (a shot has advanced it by one frame)
    assert state["fired"] == set(shots) and all(r == 0 for r in rcs), rcs
    for c in range(n):
        o = orc.Fsk(1, 1)
        o.rx(sig[c, :160*frames])
        assert taps[c] == [int(e["a"]) for e in o.sink.events()], c
    for o in objs:
        L.fsk_rx_free(o)
    L.spangpu_line_group_destroy(grp)
    # a private object: the nested call is refused (-1), the outer one is unharmed
    got = []
    inner = []
    holder = [None]

    def private_bit(u, b):
        got.append(b)
        if len(inner) < 3:
            blk = np.zeros(160, np.int16)
            inner.append(L.fsk_rx(holder[0], blk.ctypes.data, 160))
    cb = PUT_BIT(private_bit)
    holder[0] = L.fsk_rx_init(None, spec_ptr(L, 1), 1, cb, None)
    x = np.ascontiguousarray(sig[0])
    assert L.fsk_rx(holder[0], x.ctypes.data, len(x)) == 0
    o = orc.Fsk(1, 1)
    o.rx(sig[0])
    assert inner == [-1, -1, -1] and got == [int(e["a"]) for e in o.sink.events()]
    L.fsk_rx_free(holder[0])
