"""GPU tests of the spandsp-named entry points of spandsp_amd/csrc/shim_sigtone.c (include/spangpu_spandsp.h): what a
caller of sig_tone_rx() / sig_tone_tx() observes -- the rewritten frames, the callbacks, in order, and the effect of
setting a mode from inside a callback -- must equal what the reference delivers: the committed reference outputs
(tests/golden/sigtone_*.npz) and, for callbacks that set modes, the oracle (pinned to the reference on exactly that in
test_oracle_pin.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import synth
from test_oracle_pin import GOLDEN, SIGTONE_TX_CASES, zlib_crc

pytestmark = pytest.mark.gpu

REPORT = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int)


@pytest.fixture(scope="module")
def L(built):
    from spandsp_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    sig = {
        "sig_tone_rx_init": (vp, [vp, ci, REPORT, vp]), "sig_tone_rx": (ci, [vp, vp, ci]),
        "sig_tone_rx_set_mode": (None, [vp, ci, ci]), "sig_tone_rx_release": (ci, [vp]), "sig_tone_rx_free": (ci, [vp]),
        "sig_tone_tx_init": (vp, [vp, ci, REPORT, vp]), "sig_tone_tx": (ci, [vp, vp, ci]),
        "sig_tone_tx_set_mode": (None, [vp, ci, ci]), "sig_tone_tx_release": (ci, [vp]), "sig_tone_tx_free": (ci, [vp]),
    }
    for name, (res, args) in sig.items():
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
    return lib


@pytest.mark.parametrize("chunk", [160, 4000])
@pytest.mark.parametrize("tone_type,mode", [(1, 0x40), (2, 0xC0), (3, 0x40)])
def test_sig_tone_rx_against_the_reference_outputs(L, tone_type, mode, chunk):
    g = np.load(os.path.join(GOLDEN, "sigtone_rx_%d_%02x.npz" % (tone_type, mode)))
    got = []
    cb = REPORT(lambda user, what, level, dur: got.append((what, level, dur)))
    s = L.sig_tone_rx_init(None, tone_type, cb, None)
    assert s
    L.sig_tone_rx_set_mode(s, mode, 0)
    x = g["amp"]
    out = []
    for k in range(0, len(x), chunk):
        buf = x[k:k + chunk].copy()
        assert L.sig_tone_rx(s, buf.ctypes.data, len(buf)) == len(buf)
        out.append(buf)
    assert np.array_equal(np.concatenate(out), g["out"])
    assert np.array_equal(np.array(got, np.int32).reshape(-1, 3), g["events"])
    assert L.sig_tone_rx_release(s) == 0 and L.sig_tone_rx_free(s) == 0


@pytest.mark.parametrize("tone_type", [1, 2, 3])
def test_sig_tone_rx_mode_set_from_inside_the_callback(L, tone_type):
    """The reference calls back between a sample's detectors and its media path: a mode set there shows in that sample."""
    from oracle import restated as orc
    sig = synth.sig_tone_channels(6, 8000*4, 50 + tone_type, tone_type)
    for x in sig:                           # a channel whose tone qualifies often enough
        probe = orc.SigToneRx(tone_type, 0x40)
        probe.rx(x)
        if len(probe.sink.events()) >= 8:
            break
    script = [0x00, 0xC0, 0x40, 0xC0, 0x00, 0x40]*12
    o = orc.SigToneRx(tone_type, 0x40)
    o.script(script)
    pos = [0]
    got = []
    holder = {}

    def report(user, what, level, dur):
        got.append((what, level, dur))
        if pos[0] < len(script):
            L.sig_tone_rx_set_mode(holder["s"], script[pos[0]], 0)
            pos[0] += 1
    cb = REPORT(report)
    holder["s"] = s = L.sig_tone_rx_init(None, tone_type, cb, None)
    L.sig_tone_rx_set_mode(s, 0x40, 0)
    for k, m in zip(range(0, len(x), 200), [200]*1000):
        buf = x[k:k + m].copy()
        L.sig_tone_rx(s, buf.ctypes.data, len(buf))
        assert np.array_equal(buf, o.rx(x[k:k + m])), k
    want = [(int(e["a"]), int(e["b"]), int(e["c"])) for e in o.sink.events()]
    assert got == want and len(got) >= 6
    L.sig_tone_rx_free(s)


@pytest.mark.parametrize("tone_type,seed", SIGTONE_TX_CASES)
def test_sig_tone_tx_against_the_reference_outputs(L, tone_type, seed):
    """The run of test_oracle_pin.sigtone_tx_run() through the shim: every update request sets the next scripted mode."""
    g = np.load(os.path.join(GOLDEN, "sigtone_tx_%d.npz" % tone_type))
    script = g["script"]
    pos = [0]
    calls = []
    holder = {}

    def update(user, what, level, dur):
        calls.append((what, level, dur))
        if pos[0] < len(script):
            L.sig_tone_tx_set_mode(holder["s"], int(script[pos[0]][0]), int(script[pos[0]][1]))
            pos[0] += 1
    cb = REPORT(update)
    holder["s"] = s = L.sig_tone_tx_init(None, tone_type, cb, None)
    assert s
    rng = np.random.default_rng(seed + 1000)
    L.sig_tone_tx_set_mode(s, 0x11, 120)
    out = []
    for f in range(200):
        buf = rng.integers(-25000, 25000, [160, 80, 333][f % 3]).astype(np.int16)
        assert L.sig_tone_tx(s, buf.ctypes.data, len(buf)) == len(buf)
        out.append(buf)
    out = np.concatenate(out)
    assert len(out) == int(g["out_len"]) and zlib_crc(out) == int(g["out_crc"])
    assert np.array_equal(out[:4000], g["out_head"])
    assert len(calls) == int(g["requests"]) and set(calls) == {(0x100, 0, 0)}
    assert L.sig_tone_tx_release(s) == 0 and L.sig_tone_tx_free(s) == 0


def test_sig_tone_init_refusals(L):
    cb = REPORT(lambda *a: None)
    none = C.cast(None, REPORT)
    storage = C.create_string_buffer(4096)
    for init in (L.sig_tone_rx_init, L.sig_tone_tx_init):
        assert not init(None, 1, none, None)            # sig_tone.c:352,679: a callback is required
        assert not init(None, 0, cb, None)
        assert not init(None, 4, cb, None)
        assert not init(storage, 1, cb, None)           # caller storage cannot hold state that lives in HBM
    L.sig_tone_rx_free(None)
    L.sig_tone_tx_free(None)
    # an object the library made is re-initialised in place, as the reference re-initialises what it is handed
    for init, free in ((L.sig_tone_rx_init, L.sig_tone_rx_free), (L.sig_tone_tx_init, L.sig_tone_tx_free)):
        s = init(None, 1, cb, None)
        assert s and init(s, 3, cb, None) == s
        assert not init(s, 9, cb, None)
        free(s)
