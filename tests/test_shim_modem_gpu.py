"""GPU tests of the spandsp-named modem entry points (include/spangpu_spandsp.h, spandsp_amd/csrc/shim_modem.c):
what a caller of v29_rx() / v27ter_rx() / v17_rx() observes through put_bit and the modem status handler must equal
what the reference delivers (committed reference outputs, and the oracle for the restart variants)."""
import ctypes as C
import os

import numpy as np
import pytest

from test_oracle_pin import GOLDEN, use_golden_modem_tables

pytestmark = pytest.mark.gpu

PUT_BIT = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
STATUS = C.CFUNCTYPE(None, C.c_void_p, C.c_int)


class LoggingState(C.Structure):
    """src/spandsp/private/logging.h:32-42"""
    _fields_ = [("level", C.c_int), ("samples_per_second", C.c_int), ("elapsed_samples", C.c_int64), ("tag", C.c_char_p),
                ("protocol", C.c_char_p), ("span_message", C.c_void_p), ("user_data", C.c_void_p)]


def check_logging_state(lg, protocol):
    """The descriptor a receiver hands out starts as span_log_init(.., SPAN_LOG_NONE, NULL) + set_protocol leaves it, and
    the reference's own logging functions (when oracle/_ref is built) can work on it: the layout is theirs."""
    assert lg
    c = lg.contents
    assert (c.level, c.samples_per_second, c.elapsed_samples, c.tag, c.protocol) == (0, 8000, 0, None, protocol)
    from oracle import ref
    if ref.available():
        R = C.CDLL(ref.REF_SO)
        R.span_log_set_level.argtypes = [C.c_void_p, C.c_int]
        R.span_log_set_tag.argtypes = [C.c_void_p, C.c_char_p]
        R.span_log_test.argtypes = [C.c_void_p, C.c_int]
        R.span_log_test.restype = C.c_bool
        R.span_log_set_level(lg, 5 | 0x100)
        R.span_log_set_tag(lg, b"trunk 7")
        assert lg.contents.level == 5 | 0x100 and lg.contents.tag == b"trunk 7" and lg.contents.protocol == protocol
        assert R.span_log_test(lg, 4) and not R.span_log_test(lg, 6)


@pytest.fixture(scope="module")
def L(built):
    from spandsp_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    for pfx in ("v29_rx", "v27ter_rx", "v17_rx"):
        sig = {
            pfx + "_init": (vp, [vp, ci, PUT_BIT, vp]),
            "spangpu_" + pfx + "_attach": (vp, [vp, ci, PUT_BIT, vp]),
            pfx: (ci, [vp, vp, ci]),
            pfx + "_restart": (ci, [vp, ci, ci]),
            pfx + "_fillin": (ci, [vp, ci]),
            pfx + "_free": (ci, [vp]),
            pfx + "_set_put_bit": (None, [vp, PUT_BIT, vp]),
            pfx + "_set_modem_status_handler": (None, [vp, STATUS, vp]),
            pfx + "_equalizer_state": (ci, [vp, C.POINTER(vp)]),
            pfx + "_carrier_frequency": (cf, [vp]),
            pfx + "_symbol_timing_correction": (cf, [vp]),
            pfx + "_signal_power": (cf, [vp]),
            pfx + "_set_signal_cutoff": (None, [vp, cf]),
            pfx + "_get_logging_state": (C.POINTER(LoggingState), [vp]),
        }
        for name, (res, args) in sig.items():
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = args
    lib.spangpu_modem_group_create.restype = vp
    lib.spangpu_modem_group_create.argtypes = [ci, ci, ci, ci, ci]
    lib.spangpu_modem_group_destroy.argtypes = [vp]
    lib.spangpu_modem_group_flush.argtypes = [vp]
    return lib


class Tap:
    def __init__(self):
        self.ev = []
        self.status = []
        self.put_bit = PUT_BIT(lambda u, b: self.ev.append(b))
        self.on_status = STATUS(lambda u, s: self.status.append((len(self.ev), s)))


def feed(L, pfx, s, x, chunk=160):
    x = np.ascontiguousarray(x, np.int16)
    assert getattr(L, pfx)(s, x.ctypes.data, 0) == 0                            # empty input: nothing happens
    for k in range(0, len(x), chunk):
        blk = x[k:k + chunk]
        assert getattr(L, pfx)(s, blk.ctypes.data, len(blk)) == 0


@pytest.mark.parametrize("pfx,name,rate", [("v29_rx", "v29", 9600), ("v29_rx", "v29", 4800), ("v27ter_rx", "v27ter", 4800),
                                           ("v27ter_rx", "v27ter", 2400), ("v17_rx", "v17", 14400), ("v17_rx", "v17", 7200)])
def test_private_object_replays_reference_stream(L, pfx, name, rate):
    g = np.load(os.path.join(GOLDEN, "%s_%d.npz" % (name, rate)))
    t = Tap()
    s = getattr(L, pfx + "_init")(None, rate, t.put_bit, None)
    assert s
    feed(L, pfx, s, g["amp"], chunk=517)                    # one odd-sized call size: the stream must not depend on it
    assert np.array_equal(np.array(t.ev, np.int32), g["events"].astype(np.int32))
    # with a status handler installed the status codes leave the bit stream (v29rx.c:171-178)
    t2 = Tap()
    getattr(L, pfx + "_restart")(s, rate, 0)
    getattr(L, pfx + "_set_put_bit")(s, t2.put_bit, None)
    getattr(L, pfx + "_set_modem_status_handler")(s, t2.on_status, None)
    feed(L, pfx, s, g["amp"])
    want = g["events"].astype(np.int32)
    assert np.array_equal(np.array(t2.ev, np.int32), want[want >= 0])
    assert [st for _, st in t2.status] == list(want[want < 0])
    # getters read back sane values after a completed call
    f = getattr(L, pfx + "_carrier_frequency")(s)
    assert 1600.0 < f < 1900.0
    assert abs(getattr(L, pfx + "_symbol_timing_correction")(s)) < 50.0
    p = C.c_void_p()
    n = getattr(L, pfx + "_equalizer_state")(s, C.byref(p))
    assert n == (32 if name == "v27ter" else 33) and p.value
    assert getattr(L, pfx + "_signal_power")(s) < 0.0
    assert getattr(L, pfx + "_restart")(s, 1234, 0) == -1
    assert getattr(L, pfx + "_init")(None, 1234, t.put_bit, None) is None
    getattr(L, pfx + "_free")(s)


def test_v17_short_train_restart_and_rate_change(L):
    """v17_rx_restart(s, rate, short_train) from the host, including a change of bit rate (the object moves to a bank of
    the new rate), against the oracle doing the same."""
    from oracle import restated as orc
    use_golden_modem_tables()
    g1 = np.load(os.path.join(GOLDEN, "v17_9600.npz"))["amp"]
    g2 = np.load(os.path.join(GOLDEN, "v17_14400.npz"))["amp"]
    t = Tap()
    s = L.v17_rx_init(None, 9600, t.put_bit, None)
    o = orc.V17(9600)
    feed(L, "v17_rx", s, g1[:9000])
    o.rx(g1[:9000])
    assert L.v17_rx_restart(s, 9600, 1) == 0                # short train, same rate
    assert o.restart(9600, 1) == 0
    feed(L, "v17_rx", s, g1[9000:])
    o.rx(g1[9000:])
    assert L.v17_rx_restart(s, 14400, 0) == 0               # long train at another rate
    assert o.restart(14400, 0) == 0
    feed(L, "v17_rx", s, g2)
    o.rx(g2)
    assert np.array_equal(np.array(t.ev, np.int32), o.sink.events()["a"].astype(np.int32))
    assert len(t.ev) > 3000
    L.v17_rx_free(s)


def test_v29_old_train_restart_and_fillin(L):
    from oracle import restated as orc
    use_golden_modem_tables()
    x = np.load(os.path.join(GOLDEN, "v29_9600.npz"))["amp"]
    t = Tap()
    s = L.v29_rx_init(None, 9600, t.put_bit, None)
    o = orc.V29(9600)
    feed(L, "v29_rx", s, x)
    o.rx(x)
    assert L.v29_rx_restart(s, 7200, 1) == 0                # old_train restart at a new rate (V.29 banks mix rates)
    from oracle.restated import lib as olib
    assert olib().orc_v29_restart(o.p, 7200, 1) == 0
    y = np.load(os.path.join(GOLDEN, "v29_7200.npz"))["amp"]
    feed(L, "v29_rx", s, y[:3000])
    o.rx(y[:3000])
    L.v29_rx_fillin(s, 160)                                 # a lost packet
    feed(L, "v29_rx", s, y[3160:])
    # the oracle has no fill-in of its own: apply the reference's rule (v29rx.c:967-996) to its state
    f, w = o.snapshot()
    if w[9] > 0 and w[6] != 7:
        st = o.buf[4*238:4*(238 + 43)].view(np.int32)
        for _ in range(160):
            st[10] = np.int32((int(st[10]) + int(st[11])) & 0xFFFFFFFF if (int(st[10]) + int(st[11])) & 0x80000000 == 0
                              else ((int(st[10]) + int(st[11])) & 0xFFFFFFFF) - (1 << 32))
            st[17] -= 48
            if st[17] <= 0:
                st[17] += 48*10//6
    o.rx(y[3160:])
    assert np.array_equal(np.array(t.ev, np.int32), o.sink.events()["a"].astype(np.int32))
    L.v29_rx_free(s)


def test_logging_descriptors(L):
    for pfx, rate, proto in (("v29_rx", 9600, b"V.29 RX"), ("v27ter_rx", 4800, b"V.27ter RX"), ("v17_rx", 14400, b"V.17 RX")):
        tap = Tap()
        s = getattr(L, pfx + "_init")(None, rate, tap.put_bit, None)
        assert s
        check_logging_state(getattr(L, pfx + "_get_logging_state")(s), proto)
        getattr(L, pfx + "_free")(s)


def test_group_of_receivers(L):
    """N receivers on one bank: one launch per tick, callbacks per channel in order."""
    from spandsp_amd import engine
    g = np.load(os.path.join(GOLDEN, "v27ter_4800.npz"))
    x = g["amp"]
    n = 5
    grp = L.spangpu_modem_group_create(0, engine.V27TER, n, 4800, 160)
    assert grp
    taps = [Tap() for _ in range(n)]
    objs = [L.spangpu_v27ter_rx_attach(grp, c, taps[c].put_bit, None) for c in range(n)]
    assert all(objs)
    assert not L.spangpu_v27ter_rx_attach(grp, 0, taps[0].put_bit, None)        # slot taken
    assert not L.spangpu_v29_rx_attach(grp, 1, taps[0].put_bit, None)           # wrong kind
    pad = (-len(x)) % 160
    xs = np.concatenate([x, np.zeros(pad, np.int16)])
    for k in range(0, len(xs), 160):
        for c in range(n):
            blk = np.ascontiguousarray(xs[k:k + 160] if c % 2 == 0 else np.zeros(160, np.int16))
            if k == 320:
                assert L.v27ter_rx(objs[c], blk.ctypes.data, 0) == 0          # an empty block is no block (and no tick)
            L.v27ter_rx(objs[c], blk.ctypes.data, 160)
    for c in range(n):
        if c % 2 == 0:
            assert np.array_equal(np.array(taps[c].ev, np.int32), g["events"].astype(np.int32))
        else:
            assert taps[c].ev == []
    for o in objs:
        L.v27ter_rx_free(o)
    L.spangpu_modem_group_destroy(grp)
