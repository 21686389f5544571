"""Super-tone cadences matched on the device (spangpu_bank_set_cadences / _cadence_events) against the oracle's
super_tone_rx(): the tone reports and segment reports of every channel, in order, frame by frame."""
import ctypes as C

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

# (f1 Hz, f2 Hz, min ms, max ms) per element: the reference's own style of descriptor (tests/super_tone_rx_tests.c reads
# them from an XML tone plan; these are a plan of the same shape)
TONES = [
    [(400, 0, 700, 0)],                                         # continuous
    [(1100, 0, 400, 600), (0, 0, 2800, 3200)],                  # ring-back like
    [(350, 440, 400, 0)],                                       # dial tone
    [(480, 620, 450, 550), (0, 0, 450, 550)],                   # busy
    [(480, 620, 200, 300), (0, 0, 200, 300)],                   # congestion (same pair, faster)
    [(950, 0, 300, 360), (1400, 0, 300, 360), (1800, 0, 300, 360), (0, 0, 800, 1200)],      # SIT
]
PLANS = [
    [(400, 0, 1500)],
    [(1100, 0, 500), (0, 0, 3000)],
    [(350, 440, 1200), (0, 0, 300)],
    [(480, 620, 500), (0, 0, 500)],
    [(480, 620, 250), (0, 0, 250)],
    [(950, 0, 330), (1400, 0, 330), (1800, 0, 330), (0, 0, 1000)],
    [(620, 0, 300), (0, 0, 200)],                               # not in the descriptor
]


def build(desc_like):
    for tone in TONES:
        t = desc_like.add_tone()
        for f1, f2, lo, hi in tone:
            desc_like.add_element(t, f1, f2, lo, hi)


class ShimDesc:
    """super_tone_rx_make_descriptor() & co of the library: the bins come out as the reference numbers them."""

    def __init__(self, L):
        self.L = L
        L.super_tone_rx_make_descriptor.restype = C.c_void_p
        L.super_tone_rx_make_descriptor.argtypes = [C.c_void_p]
        L.super_tone_rx_add_tone.argtypes = [C.c_void_p]
        L.super_tone_rx_add_element.argtypes = [C.c_void_p] + [C.c_int]*5
        L.super_tone_rx_free_descriptor.argtypes = [C.c_void_p]
        L.spangpu_super_tone_cadences.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.p = L.super_tone_rx_make_descriptor(None)

    def add_tone(self):
        return self.L.super_tone_rx_add_tone(self.p)

    def add_element(self, *a):
        return self.L.super_tone_rx_add_element(self.p, *a)

    def close(self):
        self.L.super_tone_rx_free_descriptor(self.p)


def orc_events(det):
    ev = [tuple(int(x) for x in e) for e in det.sink.events()]
    det.sink.clear()
    return ev


@pytest.mark.parametrize("segments", [False, True])
def test_cadences_on_the_device_against_the_oracle(built, segments):
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 150
    n_frames = 500                      # 10 s of line
    sig = synth.cadence_plan_channels(n_ch, 160*n_frames, 71, PLANS)
    od = orc.SuperToneDesc()
    build(od)
    fac = list(od.fac)
    L = engine.lib()
    sd = ShimDesc(L)
    build(sd)
    bank = engine.ToneBank(engine.SUPER_TONE, n_ch, bin_fac=fac)
    assert L.spangpu_super_tone_cadences(sd.p, bank.h, int(segments)) == 0
    dets = [orc.SuperTone(od, segments) for _ in range(n_ch)]
    n_on = n_off = n_seg = 0
    for k in range(n_frames):
        fr = np.ascontiguousarray(sig[:, 160*k:160*(k + 1)])
        bank.rx_host(fr)
        got = bank.cadence_events()
        for c in range(n_ch):
            dets[c].rx(fr[c], want_blocks=False)
            want = orc_events(dets[c])
            assert got[c] == want, (k, c, got[c], want)
            n_on += sum(1 for e in want if e[0] == 1 and e[1] >= 0)
            n_off += sum(1 for e in want if e[0] == 1 and e[1] < 0)
            n_seg += sum(1 for e in want if e[0] == 4)
    assert n_on > n_ch and n_off > n_ch//2
    assert (n_seg > 5*n_ch) if segments else (n_seg == 0)
    sd.close()


@pytest.fixture(params=[0, 1, 3], ids=["auto", "general kernel + cadence_kernel", "streaming, no loader waves"])
def tone_variant(request):
    from spandsp_amd import engine
    engine.tune_tone_kernel(request.param)
    yield request.param
    engine.tune_tone_kernel(0)


def test_cadences_given_as_bins_var_lengths_reset_and_state(built, tone_variant):
    """The plain entry point (bins and milliseconds), ragged frame lengths, a channel reset mid-call, the state words out and
    back in, and a second set of cadences given to a live bank -- with the cadences matched in the streaming kernels'
    epilogue and (general detector kernel) by a launch of cadence_kernel."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 70
    sig = synth.cadence_plan_channels(n_ch, 160*300, 72, PLANS)
    od = orc.SuperToneDesc()
    build(od)
    fac = list(od.fac)
    hz = [400, 1100, 350, 440, 480, 620, 950, 1400, 1800]      # the order the descriptor met them in
    bins = {0: -1}
    bins.update({f: i for i, f in enumerate(hz)})
    assert len(fac) == len(hz)
    tones = [[(bins[f1], bins[f2], lo, hi) for f1, f2, lo, hi in t] for t in TONES]
    bank = engine.ToneBank(engine.SUPER_TONE, n_ch, bin_fac=fac)
    bank.set_cadences(tones, want_segments=True)
    dets = [orc.SuperTone(od, True) for _ in range(n_ch)]
    sizes = [160, 96, 256, 31, 128, 129]
    pos = 0
    k = 0
    total = 0
    while pos < sig.shape[1]:
        n = min(sizes[k % len(sizes)], sig.shape[1] - pos)
        if k == 90:
            # channel 5 starts afresh (what super_tone_rx_init() on a live object does)
            bank.reset_channel(5)
            bank.cadence_reset(5)
            dets[5] = orc.SuperTone(od, True)
        if k == 120:
            w = bank.cadence_get_state(7)
            bank.cadence_reset(7)
            assert bank.cadence_get_state(7)[2] == -1 and (bank.cadence_get_state(7)[14:] == 0).all()
            bank.cadence_set_state(7, w)
            assert (bank.cadence_get_state(7) == w).all()
        fr = np.ascontiguousarray(sig[:, pos:pos + n])
        bank.rx_host(fr)
        assert bank.cadence_run() == bank.cadence_run()         # taking a launch twice is harmless
        got = bank.cadence_events()
        for c in range(n_ch):
            dets[c].rx(fr[c], want_blocks=False)
            want = orc_events(dets[c])
            assert got[c] == want, (k, c, got[c], want)
            total += len(want)
        pos += n
        k += 1
    assert total > 5*n_ch
    # no tones at all: nobody follows a tone any more (the numbers belonged to the old set), segments only
    bank.set_cadences([], want_segments=True)
    assert all(bank.cadence_get_state(c)[2] == -1 for c in range(0, n_ch, 7))
    for _ in range(3):
        bank.rx_host(np.zeros((n_ch, 160), np.int16))
        ev = bank.cadence_events()
        assert all(e[0] == 4 for c in range(n_ch) for e in ev[c])
    with pytest.raises(Exception):
        engine.ToneBank(engine.DTMF, 4).set_cadences(tones)
    with pytest.raises(Exception):
        bank.set_cadences([[(300, 0, 100, 0)]])                 # a bin number, not a frequency


def test_cadences_65536_channels(built):
    """A full-size bank: 65 536 lines, two seconds; every 97th channel against the oracle, the rest by replica (the lines
    repeat with period 256)."""
    from oracle import restated as orc
    from spandsp_amd import engine
    base = 256
    n_ch = 65536
    n_frames = 100
    sig = synth.cadence_plan_channels(base, 160*n_frames, 73, PLANS)
    od = orc.SuperToneDesc()
    build(od)
    L = engine.lib()
    sd = ShimDesc(L)
    build(sd)
    bank = engine.ToneBank(engine.SUPER_TONE, n_ch, bin_fac=list(od.fac))
    assert L.spangpu_super_tone_cadences(sd.p, bank.h, 1) == 0
    picks = list(range(0, n_ch, 97))
    dets = {c: orc.SuperTone(od, True) for c in picks}
    total = 0
    for k in range(n_frames):
        fr = np.ascontiguousarray(np.tile(sig[:, 160*k:160*(k + 1)], (n_ch//base, 1)))
        bank.rx_host(fr)
        ev = C.c_void_p()
        cnt = C.c_void_p()
        slots = L.spangpu_bank_cadence_events(bank.h, C.byref(ev), C.byref(cnt))
        assert slots >= 3
        counts = np.ctypeslib.as_array(C.cast(cnt, C.POINTER(C.c_int32)), (n_ch,))
        words = np.ctypeslib.as_array(C.cast(ev, C.POINTER(C.c_uint32)), (slots, n_ch, 2))
        c2 = counts.reshape(-1, base)
        assert (c2 == c2[0]).all()
        w4 = words.reshape(slots, -1, base, 2)
        live = np.arange(slots)[:, None] < c2[0][None, :]
        assert ((w4 == w4[:, :1]) | ~live[:, None, :, None]).all()
        got = bank.cadence_events()
        # the compact list says the same as the slot arrays
        lst = bank.cadence_list()
        assert len(lst) == int(counts.sum())
        by_ch = {}
        for c, w0, w1 in lst.tolist():
            by_ch.setdefault(c, []).append((w0, w1))
        assert sorted(by_ch) == np.nonzero(counts)[0].tolist()
        for c in list(by_ch)[:200] + [c for c in picks if c in by_ch]:
            assert by_ch[c] == [(int(words[k, c, 0]), int(words[k, c, 1])) for k in range(int(counts[c]))]
        for c in picks:
            dets[c].rx(fr[c], want_blocks=False)
            want = orc_events(dets[c])
            assert got[c] == want, (k, c)
            total += len(want)
    assert total > len(picks)
    sd.close()


def test_cadence_state_does_not_depend_on_who_asks(built, tone_variant):
    """A caller that looks at the cadence events only now and then: every block is counted all the same, whichever kernel
    served the frame (the next launch takes the records of the last one to the matcher if nobody has).  Against a bank that is
    asked after every frame: the same cadence state after every frame, the same events whenever both are asked."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 40
    sig = synth.cadence_plan_channels(n_ch, 160*200, 73, PLANS)
    od = orc.SuperToneDesc()
    build(od)
    fac = list(od.fac)
    hz = [400, 1100, 350, 440, 480, 620, 950, 1400, 1800]
    bins = {0: -1}
    bins.update({f: i for i, f in enumerate(hz)})
    tones = [[(bins[f1], bins[f2], lo, hi) for f1, f2, lo, hi in t] for t in TONES]
    asked = engine.ToneBank(engine.SUPER_TONE, n_ch, bin_fac=fac)
    lazy = engine.ToneBank(engine.SUPER_TONE, n_ch, bin_fac=fac)
    for b in (asked, lazy):
        b.set_cadences(tones, want_segments=True)
    sizes = [160, 96, 256, 31, 128, 129]
    pos = 0
    k = 0
    seen = 0
    while pos < sig.shape[1]:
        n = min(sizes[k % len(sizes)], sig.shape[1] - pos)
        fr = np.ascontiguousarray(sig[:, pos:pos + n])
        asked.rx_host(fr)
        want = asked.cadence_events()
        lazy.rx_host(fr)
        if k % 7 == 6:
            got = lazy.cadence_events()
            assert got == want, k
            seen += sum(len(e) for e in got)
            for c in range(0, n_ch, 5):
                assert (lazy.cadence_get_state(c) == asked.cadence_get_state(c)).all(), (k, c)
        pos += n
        k += 1
    lazy.cadence_run()
    for c in range(n_ch):
        assert (lazy.cadence_get_state(c) == asked.cadence_get_state(c)).all(), c
    assert seen > 0
