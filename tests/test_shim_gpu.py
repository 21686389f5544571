"""GPU tests of the spandsp-named C entry points (include/spangpu_spandsp.h,
spandsp_amd/csrc/shim_tone.c): what a caller of dtmf_rx() / bell_mf_rx() / r2_mf_rx() /
super_tone_rx() / goertzel_update() observes -- callbacks, their arguments and order,
digit buffers, status -- must equal what the reference delivers (checked against the
oracle, which test_oracle_pin.py pins to the reference)."""
import ctypes as C

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

DIGITS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int)
TONE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int)
SEG_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int)


@pytest.fixture(scope="module")
def L(built):
    from spandsp_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    sig = {
        "spangpu_group_create": (vp, [ci, ci, ci, ci, vp]),
        "spangpu_group_destroy": (ci, [vp]),
        "spangpu_group_flush": (ci, [vp]),
        "spangpu_dtmf_rx_attach": (vp, [vp, ci, DIGITS_CB, vp]),
        "spangpu_bell_mf_rx_attach": (vp, [vp, ci, DIGITS_CB, vp]),
        "spangpu_r2_mf_rx_attach": (vp, [vp, ci, TONE_CB, vp]),
        "spangpu_super_tone_rx_attach": (vp, [vp, ci, vp, TONE_CB, vp]),
        "spangpu_super_tone_params": (ci, [vp, vp]),
        "dtmf_rx_init": (vp, [vp, DIGITS_CB, vp]),
        "dtmf_rx_free": (ci, [vp]),
        "dtmf_rx_set_realtime_callback": (None, [vp, TONE_CB, vp]),
        "dtmf_rx_parms": (None, [vp, ci, cf, cf, cf]),
        "dtmf_rx": (ci, [vp, vp, ci]),
        "dtmf_rx_fillin": (ci, [vp, ci]),
        "dtmf_rx_status": (ci, [vp]),
        "dtmf_rx_get": (C.c_size_t, [vp, C.c_char_p, ci]),
        "dtmf_rx_get_logging_state": (vp, [vp]),
        "bell_mf_rx_init": (vp, [vp, DIGITS_CB, vp]),
        "bell_mf_rx_free": (ci, [vp]),
        "bell_mf_rx": (ci, [vp, vp, ci]),
        "bell_mf_rx_get": (C.c_size_t, [vp, C.c_char_p, ci]),
        "r2_mf_rx_init": (vp, [vp, C.c_bool, TONE_CB, vp]),
        "r2_mf_rx_free": (ci, [vp]),
        "r2_mf_rx": (ci, [vp, vp, ci]),
        "r2_mf_rx_get": (ci, [vp]),
        "super_tone_rx_make_descriptor": (vp, [vp]),
        "super_tone_rx_free_descriptor": (ci, [vp]),
        "super_tone_rx_add_tone": (ci, [vp]),
        "super_tone_rx_add_element": (ci, [vp, ci, ci, ci, ci, ci]),
        "super_tone_rx_init": (vp, [vp, vp, TONE_CB, vp]),
        "super_tone_rx_free": (ci, [vp]),
        "super_tone_rx_segment_callback": (None, [vp, SEG_CB]),
        "super_tone_rx": (ci, [vp, vp, ci]),
        "make_goertzel_descriptor": (None, [vp, cf, ci]),
        "goertzel_init": (vp, [vp, vp]),
        "goertzel_free": (ci, [vp]),
        "goertzel_update": (ci, [vp, vp, ci]),
        "goertzel_result": (cf, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


class Rec:
    """Collects callback invocations in the oracle's event format."""

    def __init__(self):
        self.events = []
        self.text = ""
        self.digits_cb = DIGITS_CB(self._digits)
        self.tone_cb = TONE_CB(self._tone)
        self.seg_cb = SEG_CB(self._seg)

    def _digits(self, ud, digits, n):
        self.text += digits[:n].decode("latin1")
        self.events.append((2, n, 0, 0))

    def _tone(self, ud, code, level, delay):
        self.events.append((1, code, level, delay))

    def _seg(self, ud, f1, f2, dur):
        self.events.append((4, f1, f2, dur))


def orc_events(det):
    return [tuple(int(x) for x in e) for e in det.sink.events()]


def i16(a):
    return np.ascontiguousarray(a, np.int16)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_dtmf_private_object(L, mode):
    """dtmf_rx_init(NULL, ...) + dtmf_rx(): per-call synchronous behaviour, all three delivery paths."""
    from oracle import restated as orc
    sig, _ = synth.dtmf_channels(6, 160*70, seed=51)
    for c in range(6):
        rec = Rec()
        s = L.dtmf_rx_init(None, rec.digits_cb if mode == 1 else DIGITS_CB(0), None)
        assert s
        if mode == 2:
            L.dtmf_rx_set_realtime_callback(s, rec.tone_cb, None)
        o = orc.Dtmf(mode)
        x = sig[c]
        sizes = [160, 160, 80, 240, 1, 159]
        pos = 0
        k = 0
        while pos < len(x):
            n = min(sizes[k % len(sizes)], len(x) - pos)
            fr = i16(x[pos:pos + n])
            assert L.dtmf_rx(s, fr.ctypes.data, n) == 0
            o.rx(fr)
            assert L.dtmf_rx_status(s) == o.status()
            assert rec.events == orc_events(o), (c, pos)
            pos += n
            k += 1
        buf = C.create_string_buffer(200)
        L.dtmf_rx_get(s, buf, 128)
        assert buf.value.decode() == o.get()
        assert rec.text == o.sink.text()
        L.dtmf_rx_free(s)


def test_dtmf_private_parms_and_fillin(L):
    from oracle import restated as orc
    sig, _ = synth.dtmf_channels(3, 160*50, seed=52)
    t = np.arange(sig.shape[1])
    dial = 3000.0*np.sin(2*np.pi*350.0*t/8000.0) + 3000.0*np.sin(2*np.pi*440.0*t/8000.0)
    for c in range(3):
        x = np.clip(sig[c] + dial, -32768, 32767).astype(np.int16)
        s = L.dtmf_rx_init(None, DIGITS_CB(0), None)
        o = orc.Dtmf(0)
        for i, pos in enumerate(range(0, len(x), 160)):
            if i == 5:
                L.dtmf_rx_parms(s, 1, 9.0, 5.0, -39.0)
                o.parms(1, 9.0, 5.0, -39.0)
            if i == 20:
                L.dtmf_rx_fillin(s, 160)
                o.fillin(160)
            fr = i16(x[pos:pos + 160])
            L.dtmf_rx(s, fr.ctypes.data, len(fr))
            o.rx(fr)
        buf = C.create_string_buffer(200)
        L.dtmf_rx_get(s, buf, 128)
        assert buf.value.decode() == o.get()
        L.dtmf_rx_free(s)


def test_dtmf_logging_descriptor(L):
    from test_shim_modem_gpu import LoggingState, check_logging_state
    s = L.dtmf_rx_init(None, DIGITS_CB(0), None)
    check_logging_state(C.cast(L.dtmf_rx_get_logging_state(s), C.POINTER(LoggingState)), b"DTMF")
    L.dtmf_rx_free(s)


def test_dtmf_group_of_channels(L):
    """N dtmf_rx() callers on one bank: one launch per tick, callbacks replayed per channel."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 100
    sig, _ = synth.dtmf_channels(n_ch, 160*60, seed=53)
    g = L.spangpu_group_create(0, engine.DTMF, n_ch, 160, None)
    assert g
    recs = [Rec() for _ in range(n_ch)]
    hs = [L.spangpu_dtmf_rx_attach(g, c, recs[c].digits_cb, None) for c in range(n_ch)]
    assert all(hs)
    dets = [orc.Dtmf(1) for _ in range(n_ch)]
    for pos in range(0, sig.shape[1], 160):
        for c in range(n_ch):
            fr = i16(sig[c, pos:pos + 160])
            assert L.dtmf_rx(hs[c], fr.ctypes.data, 160) == 0      # the last one triggers the launch
            dets[c].rx(fr)
        for c in range(n_ch):
            assert recs[c].events == orc_events(dets[c]), (c, pos)
    assert sum(len(r.text) for r in recs) > n_ch
    for c in range(n_ch):
        assert recs[c].text == dets[c].sink.text()
        L.dtmf_rx_free(hs[c])
    L.spangpu_group_destroy(g)


def test_dtmf_group_tick_with_late_and_silent_channels(L):
    """A tick runs with the channels that staged when its owner calls spangpu_group_flush(): a late or silent channel
    stalls nobody and is itself untouched; frames of different lengths share a tick; dtmf_rx_parms() on an attached
    object is that channel's own."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 70
    sig, _ = synth.dtmf_channels(n_ch, 160*60, seed=57)
    t = np.arange(sig.shape[1])
    dial = 3000.0*np.sin(2*np.pi*350.0*t/8000.0) + 3000.0*np.sin(2*np.pi*440.0*t/8000.0)
    sig[:8] = np.clip(sig[:8].astype(np.float64) + dial, -32768, 32767).astype(np.int16)
    g = L.spangpu_group_create(0, engine.DTMF, n_ch, 160, None)
    recs = [Rec() for _ in range(n_ch)]
    hs = [L.spangpu_dtmf_rx_attach(g, c, recs[c].digits_cb, None) for c in range(n_ch)]
    dets = [orc.Dtmf(1) for _ in range(n_ch)]
    for c in range(0, 8, 2):
        L.dtmf_rx_parms(hs[c], 1, 0.0, -1.0, -40.0)
        dets[c].parms(1, 0.0, -1.0, -40.0)
    rng = np.random.default_rng(3)
    pos = np.zeros(n_ch, np.int64)
    for tick in range(75):
        r = rng.random(n_ch)
        lens = np.where(r < 0.25, 0, np.where(r < 0.35, 80, 160))
        lens = np.minimum(lens, sig.shape[1] - pos)
        if tick == 40:
            lens[:] = 160
            lens = np.minimum(lens, sig.shape[1] - pos)
        staged = 0
        for c in range(n_ch):
            if lens[c]:
                fr = i16(sig[c, pos[c]:pos[c] + lens[c]])
                assert L.dtmf_rx(hs[c], fr.ctypes.data, int(lens[c])) == 0
                dets[c].rx(fr)
                staged += 1
        if staged and staged < n_ch:
            c = int(np.flatnonzero(lens)[0])
            fr = i16(sig[c, :160])
            assert L.dtmf_rx(hs[c], fr.ctypes.data, 160) == -1         # a second frame before the tick ran is refused
        want = 0 if staged == n_ch else staged                           # a full house ran by itself
        assert L.spangpu_group_flush(g) == want
        assert L.spangpu_group_flush(g) == 0
        pos += lens
        for c in range(n_ch):
            assert recs[c].events == orc_events(dets[c]), (c, tick)
    assert sum(len(r.text) for r in recs) > n_ch//2
    for c in range(n_ch):
        assert recs[c].text == dets[c].sink.text()
        L.dtmf_rx_free(hs[c])
    L.spangpu_group_destroy(g)


def test_dtmf_group_staged_from_many_threads(L):
    """Staging is safe from many threads (one submitter per channel, as for a spandsp object): 8 threads feed 12 channels
    each; whichever completes the tick runs it and replays everybody's callbacks."""
    import threading
    from oracle import restated as orc
    from spandsp_amd import engine
    n_thr, per = 8, 12
    n_ch = n_thr*per
    n_ticks = 50
    sig, _ = synth.dtmf_channels(n_ch, 160*n_ticks, seed=58)
    g = L.spangpu_group_create(0, engine.DTMF, n_ch, 160, None)
    recs = [Rec() for _ in range(n_ch)]
    hs = [L.spangpu_dtmf_rx_attach(g, c, recs[c].digits_cb, None) for c in range(n_ch)]
    frames = [[i16(sig[c, k*160:(k + 1)*160]) for k in range(n_ticks)] for c in range(n_ch)]
    gate = threading.Barrier(n_thr)
    errors = []

    def worker(w):
        try:
            for k in range(n_ticks):
                for c in range(w*per, (w + 1)*per):
                    if L.dtmf_rx(hs[c], frames[c][k].ctypes.data, 160) != 0:
                        errors.append((c, k))
                gate.wait()             # the tick is over (its last stager ran it) before anybody stages the next
        except Exception as e:          # pragma: no cover
            errors.append(repr(e))
            gate.abort()

    th = [threading.Thread(target=worker, args=(w,)) for w in range(n_thr)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert errors == []
    n_digits = 0
    for c in range(n_ch):
        d = orc.Dtmf(1)
        for k in range(n_ticks):
            d.rx(frames[c][k])
        assert recs[c].events == orc_events(d), c
        assert recs[c].text == d.sink.text()
        n_digits += len(recs[c].text)
        L.dtmf_rx_free(hs[c])
    assert n_digits > n_ch
    L.spangpu_group_destroy(g)


def test_bell_and_r2_private_objects(L):
    from oracle import restated as orc
    sig, _ = synth.bell_mf_channels(4, 160*100, seed=54)
    for c in range(4):
        rec = Rec()
        s = L.bell_mf_rx_init(None, rec.digits_cb, None)
        o = orc.BellMf(1)
        for pos in range(0, sig.shape[1], 160):
            fr = i16(sig[c, pos:pos + 160])
            L.bell_mf_rx(s, fr.ctypes.data, 160)
            o.rx(fr)
        assert rec.events == orc_events(o) and rec.text == o.sink.text()
        L.bell_mf_rx_free(s)
    for fwd in (True, False):
        sig, _ = synth.r2_mf_channels(3, 160*80, seed=55, fwd=fwd)
        for c in range(3):
            rec = Rec()
            s = L.r2_mf_rx_init(None, fwd, rec.tone_cb, None)
            o = orc.R2Mf(fwd, True)
            for pos in range(0, sig.shape[1], 160):
                fr = i16(sig[c, pos:pos + 160])
                L.r2_mf_rx(s, fr.ctypes.data, 160)
                o.rx(fr)
                assert L.r2_mf_rx_get(s) == o.snapshot()["current_digit"]
            assert rec.events == orc_events(o)
            L.r2_mf_rx_free(s)


def _build_desc(add_tone, add_element):
    t = add_tone()
    add_element(t, 400, 0, 700, 0)
    t = add_tone()
    add_element(t, 1100, 0, 400, 600)
    add_element(t, 0, 0, 2800, 3200)
    t = add_tone()
    add_element(t, 350, 440, 400, 0)
    t = add_tone()
    add_element(t, 480, 620, 450, 550)
    add_element(t, 0, 0, 450, 550)
    t = add_tone()
    add_element(t, 445, 0, 300, 0)


def test_super_tone_private_and_group(L):
    from oracle import restated as orc
    from spandsp_amd import engine
    desc = L.super_tone_rx_make_descriptor(None)
    _build_desc(lambda: L.super_tone_rx_add_tone(desc), lambda *a: L.super_tone_rx_add_element(desc, *a))
    od = orc.SuperToneDesc()
    _build_desc(od.add_tone, od.add_element)
    n_ch = 40
    sig = synth.call_progress_channels(n_ch, 160*200, seed=56)
    # private object with the segment callback
    rec = Rec()
    s = L.super_tone_rx_init(None, desc, rec.tone_cb, None)
    assert s
    L.super_tone_rx_segment_callback(s, rec.seg_cb)
    o = orc.SuperTone(od, True)
    for pos in range(0, sig.shape[1], 160):
        fr = i16(sig[0, pos:pos + 160])
        assert L.super_tone_rx(s, fr.ctypes.data, 160) == 160
        o.rx(fr)
    assert rec.events == orc_events(o) and len(rec.events) > 0
    L.super_tone_rx_free(s)
    # a group sharing the descriptor
    p = engine.ToneParams()
    assert L.spangpu_super_tone_params(desc, C.byref(p)) == 0
    g = L.spangpu_group_create(0, engine.SUPER_TONE, n_ch, 160, C.byref(p))
    recs = [Rec() for _ in range(n_ch)]
    hs = [L.spangpu_super_tone_rx_attach(g, c, desc, recs[c].tone_cb, None) for c in range(n_ch)]
    dets = [orc.SuperTone(od, False) for _ in range(n_ch)]
    for pos in range(0, sig.shape[1], 160):
        for c in range(n_ch):
            fr = i16(sig[c, pos:pos + 160])
            L.super_tone_rx(hs[c], fr.ctypes.data, 160)
            dets[c].rx(fr)
    n_ev = 0
    for c in range(n_ch):
        assert recs[c].events == orc_events(dets[c]), c
        n_ev += len(recs[c].events)
        L.super_tone_rx_free(hs[c])
    assert n_ev > 0
    L.spangpu_group_destroy(g)
    L.super_tone_rx_free_descriptor(desc)


def _build_wide_desc(add_tone, add_element):
    """22 monitored frequencies plus the resolver's naming quirks (see test_tone_gpu._st_desc_wide)."""
    base = [350, 440, 480, 620, 950, 1100, 1400, 1800, 400, 425, 450, 500, 540, 660, 700, 770, 852, 941, 1004, 1209, 1336, 1477]
    ids = []
    for k in range(0, len(base), 2):
        t = add_tone()
        ids.append(add_element(t, base[k], base[k + 1], 300, 0))
        ids.append(add_element(t, 0, 0, 200, 0))
    t = add_tone()
    ids.append(add_element(t, 355, 0, 400, 0))
    ids.append(add_element(t, 355, 445, 400, 0))
    t = add_tone()
    ids.append(add_element(t, 1100, 0, 400, 600))
    ids.append(add_element(t, 0, 0, 2800, 3200))
    return ids


def test_super_tone_descriptor_of_more_than_16_frequencies(L):
    """A descriptor the reference accepts (up to 64 pitches) with more monitored frequencies than one lane's 16 bins:
    element numbering, tone reports and segment reports equal the oracle's."""
    from oracle import restated as orc
    desc = L.super_tone_rx_make_descriptor(None)
    ids = _build_wide_desc(lambda: L.super_tone_rx_add_tone(desc), lambda *a: L.super_tone_rx_add_element(desc, *a))
    od = orc.SuperToneDesc()
    oids = _build_wide_desc(od.add_tone, od.add_element)
    assert ids == oids
    assert len(od.fac) > 16
    sig = synth.call_progress_channels(6, 160*220, seed=58)
    for c in range(6):
        rec = Rec()
        s = L.super_tone_rx_init(None, desc, rec.tone_cb, None)
        assert s
        L.super_tone_rx_segment_callback(s, rec.seg_cb)
        o = orc.SuperTone(od, True)
        for pos in range(0, sig.shape[1], 160):
            fr = i16(sig[c, pos:pos + 160])
            assert L.super_tone_rx(s, fr.ctypes.data, 160) == 160
            o.rx(fr)
        assert rec.events == orc_events(o), c
        assert len(rec.events) > 0
        L.super_tone_rx_free(s)
    L.super_tone_rx_free_descriptor(desc)


def test_goertzel_object(L):
    """goertzel_update() clamps to the block; goertzel_result() works at a block end and mid-block."""
    from oracle import restated as orc
    sig = synth.call_progress_channels(2, 4000, seed=57)
    d = (C.c_float*2)()
    L.make_goertzel_descriptor(d, 440.0, 205)
    s = L.goertzel_init(None, d)
    assert s
    o = orc.Goertzel(440.0, 205)
    x = sig[1]
    pos = 0
    for n in [100, 100, 100, 205, 7, 50, 300, 64]:
        fr = i16(x[pos:pos + n])
        a = L.goertzel_update(s, fr.ctypes.data, len(fr))
        b = o.update(fr)
        assert a == b
        pos += a
        if a < n or n in (205, 50):
            ra = np.float32(L.goertzel_result(s))
            rb = np.float32(o.result())
            assert ra.tobytes() == rb.tobytes(), (n, ra, rb)
    L.goertzel_free(s)
