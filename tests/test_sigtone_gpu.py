"""In-band signalling tone banks (sig_tone.c) against the oracle: sig_tone_rx and sig_tone_tx.

Bar: the rewritten frames, the (signalling_state, duration) reports with the samples they happen at, the sender's
update requests and every state word bit-exact (integer words, the float filter state as bits).  The oracle
(oracle/sigtone_oracle.c) is pinned to the real reference in test_oracle_pin.py.
"""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

MODES = [0x00, 0x40, 0xC0]          # muted; pass-through with the notch while a tone is about; notch always in


def run_rx(bank, orcs, sig, sizes, lens_of=None):
    n = len(orcs)
    pos = 0
    k = 0
    got = [[] for _ in range(n)]
    while pos < sig.shape[1]:
        m = min(sizes[k % len(sizes)], sig.shape[1] - pos)
        frame = sig[:, pos:pos + m]
        if lens_of is None:
            out = bank.rx_host(frame)
            lens = np.full(n, m)
        else:
            lens = lens_of(k, n, m)
            out = bank.rx_host_var(frame, lens)
        ev = bank.events() if lens.max() > 0 else [np.zeros((0, 3), np.int32)]*n
        for c, o in enumerate(orcs):
            before = len(o.sink.events())
            want = o.rx(frame[c, :lens[c]])
            assert np.array_equal(out[c, :lens[c]], want), (c, pos)
            assert np.array_equal(out[c, lens[c]:], frame[c, lens[c]:]), (c, pos)       # past a channel's length: untouched
            new = o.sink.events()[before:]
            assert [(int(s), int(d)) for _, s, d in ev[c]] == [(int(e["a"]), int(e["c"])) for e in new], (c, pos)
            got[c].extend((int(at), int(s), int(d)) for at, s, d in ev[c])
        pos += m
        k += 1
        if k % 5 == 0:
            for c in range(0, n, max(1, n//8)):
                assert np.array_equal(bank.get_state(c), orcs[c].snapshot()), (c, pos)
    for c in range(n):
        assert np.array_equal(bank.get_state(c), orcs[c].snapshot()), c
    return got


@pytest.mark.parametrize("tone_type", [1, 2, 3])
def test_sigtone_rx(built, tone_type):
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 96
    sig = synth.sig_tone_channels(n, 8000*4, 700 + tone_type, tone_type)
    bank = engine.SigToneRxBank(tone_type, n)
    orcs = []
    for c in range(n):
        mode = MODES[c % 3]
        bank.set_mode(mode, c)
        orcs.append(orc.SigToneRx(tone_type, mode))
    assert np.array_equal(bank.thresholds(), orc.sigtone_rx_thresholds(tone_type))
    got = run_rx(bank, orcs, sig, [160, 160, 80, 1, 333, 7, 160])
    states = {s for g in got for _, s, _ in g}
    assert len(states) >= (4 if tone_type == 3 else 2), states       # tones came and went (both, for the two-tone type)
    # the sample a report carries is where the oracle's event happened within its call: all inside the frame
    assert all(0 <= at < 333 for g in got for at, _, _ in g)


def test_sigtone_rx_mode_change_and_all_channels(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 40
    sig = synth.sig_tone_channels(n, 8000*2, 811, 1)
    bank = engine.SigToneRxBank(1, n)
    orcs = [orc.SigToneRx(1, 0) for _ in range(n)]
    half = sig.shape[1]//2
    run_rx(bank, orcs, sig[:, :half], [160])
    bank.set_mode(0xC0)                       # every channel
    for o in orcs:
        o.set_mode(0xC0)
    run_rx(bank, orcs, sig[:, half:], [160])


def test_sigtone_rx_var_lengths(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 70
    sig = synth.sig_tone_channels(n, 8000*2, 905, 3)
    bank = engine.SigToneRxBank(3, n)
    orcs = []
    for c in range(n):
        bank.set_mode(MODES[(c + 1) % 3], c)
        orcs.append(orc.SigToneRx(3, MODES[(c + 1) % 3]))
    rng = np.random.default_rng(5)

    def lens_of(k, n, m):
        lens = rng.integers(0, m + 1, n)
        lens[rng.random(n) < 0.3] = 0
        lens[rng.random(n) < 0.3] = m
        if k % 4 == 3:
            lens[:] = 0
        return lens.astype(np.int32)
    run_rx(bank, orcs, sig, [160, 200, 8, 160], lens_of)


def test_sigtone_rx_device_rows_unaligned(built):
    """Device-resident frames with a stride that is not a multiple of eight samples (the scalar row path)."""
    import ctypes
    from oracle import restated as orc
    from spandsp_amd import engine
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n = 33
    samples = 160
    stride = 163
    sig = synth.sig_tone_channels(n, samples*20, 66, 2)
    bank = engine.SigToneRxBank(2, n)
    bank.set_mode(0x40)
    orcs = [orc.SigToneRx(2, 0x40) for _ in range(n)]
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), n*stride*2) == 0
    for f in range(20):
        frame = np.zeros((n, stride), np.int16)
        frame[:, :samples] = sig[:, f*samples:(f + 1)*samples]
        frame[:, samples:] = 12345
        assert hip.hipMemcpy(buf, frame.ctypes.data, frame.nbytes, 1) == 0
        bank.rx_device(buf, samples, stride)
        bank.sync()
        out = np.zeros_like(frame)
        assert hip.hipMemcpy(out.ctypes.data, buf, out.nbytes, 2) == 0
        for c, o in enumerate(orcs):
            assert np.array_equal(out[c, :samples], o.rx(frame[c, :samples])), (c, f)
        assert (out[:, samples:] == 12345).all()
    hip.hipFree(buf)
    for c in range(n):
        assert np.array_equal(bank.get_state(c), orcs[c].snapshot()), c


def test_sigtone_rx_refuses_other_types(built):
    from spandsp_amd import engine
    for t in (0, 4, -1):
        with pytest.raises(engine.SpanGpuError):
            engine.SigToneRxBank(t, 4)
        with pytest.raises(engine.SpanGpuError):
            engine.SigToneTxBank(t, 4)


def script_for(rng, k):
    modes = [0x00, 0x01, 0x04, 0x05, 0x10, 0x11, 0x14, 0x15]
    out = []
    for i in range(k):
        out.append((int(rng.choice(modes)), int(rng.choice([0, 37, 160, 161, 400, 801, 2400, 3333]) if i == k - 1
                                                else rng.choice([1, 37, 160, 161, 400, 801, 2400, 3333]))))
    return out


@pytest.mark.parametrize("tone_type", [1, 2, 3])
def test_sigtone_tx(built, tone_type):
    """Each channel plays its own script of (mode, duration) pairs, one per update request, as a caller's callback would."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 48
    rng = np.random.default_rng(40 + tone_type)
    scripts = [script_for(rng, int(rng.integers(3, 30))) for _ in range(n)]
    pos = [0]*n
    bank = engine.SigToneTxBank(tone_type, n)
    orcs = [orc.SigToneTx(tone_type, scripts[c]) for c in range(n)]
    first_mode = np.array([int(rng.choice([0x01, 0x11, 0x05, 0x10])) for _ in range(n)], np.int32)
    first_dur = np.array([int(rng.choice([0, 50, 160, 1000])) for _ in range(n)], np.int32)
    bank.set_modes(first_mode, first_dur)
    for c in range(n):
        orcs[c].set_mode(int(first_mode[c]), int(first_dur[c]))
    calls = [0]

    def on_request(who):
        calls[0] += len(who)
        m = []
        d = []
        for c in who:
            if pos[c] < len(scripts[c]):
                m.append(scripts[c][pos[c]][0])
                d.append(scripts[c][pos[c]][1])
                pos[c] += 1
            else:
                m.append(-1)
                d.append(0)
        return np.array(m, np.int32), np.array(d, np.int32)

    for f in range(120):
        samples = [160, 160, 80, 1, 333][f % 5]
        frame = rng.integers(-30000, 30000, (n, samples)).astype(np.int16)
        out = bank.tx_host(frame, on_request)
        for c, o in enumerate(orcs):
            assert np.array_equal(out[c], o.tx(frame[c])), (c, f)
        if f % 10 == 9:
            for c in range(n):
                assert np.array_equal(bank.get_state(c), orcs[c].snapshot()[:5]), (c, f)
    assert calls[0] == sum(o.requests() for o in orcs) and calls[0] > n


def test_sigtone_tx_then_rx_loop(built):
    """A bank of senders feeding a bank of receivers on the device's own frames: the tone a sender turns on and off is
    what the receiver reports, and both ends agree with their oracles."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 64
    tx = engine.SigToneTxBank(1, n)
    rx = engine.SigToneRxBank(1, n)
    rx.set_mode(0x40)
    otx = [orc.SigToneTx(1) for _ in range(n)]
    orx = [orc.SigToneRx(1, 0x40) for _ in range(n)]
    rng = np.random.default_rng(9)
    seen = set()
    for f in range(150):
        if f % 25 == 0:
            on = (f//25) % 2 == 0
            modes = np.where(rng.random(n) < 0.9, 0x11 if on else 0x10, -1).astype(np.int32)
            tx.set_modes(modes, np.zeros(n, np.int32))
            for c in range(n):
                if modes[c] >= 0:
                    otx[c].set_mode(int(modes[c]), 0)
        frame = rng.normal(0, 30, (n, 160)).astype(np.int16)
        sent = tx.tx_host(frame)
        got = rx.rx_host(sent)
        ev = rx.events()
        for c in range(n):
            want_sent = otx[c].tx(frame[c])
            assert np.array_equal(sent[c], want_sent), (c, f)
            before = len(orx[c].sink.events())
            assert np.array_equal(got[c], orx[c].rx(want_sent)), (c, f)
            new = orx[c].sink.events()[before:]
            assert [(int(s), int(d)) for _, s, d in ev[c]] == [(int(e["a"]), int(e["c"])) for e in new]
            seen.update(int(s) for _, s, _ in ev[c])
    assert {0x3, 0x2} <= seen


def test_sigtone_rx_full_size_bank(built):
    """65 536 receivers on one launch per frame: 128 distinct lines, each carried by 512 channels spread over the bank.
    Every replica must leave the same frames, reports and state as the oracle's run of its line."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n_src, n = 128, 65536
    frames = 30
    sig = synth.sig_tone_channels(n_src, 160*frames, 321, 3)
    pick = (np.arange(n)*37) % n_src
    bank = engine.SigToneRxBank(3, n)
    bank.set_mode(0x40)
    orcs = [orc.SigToneRx(3, 0x40) for _ in range(n_src)]
    total = 0
    for f in range(frames):
        src = sig[:, f*160:(f + 1)*160]
        out = bank.rx_host(src[pick])
        ev = bank.events()
        want = []
        want_ev = []
        for c, o in enumerate(orcs):
            before = len(o.sink.events())
            want.append(o.rx(src[c]))
            want_ev.append([(int(e["a"]), int(e["c"])) for e in o.sink.events()[before:]])
        want = np.stack(want)
        assert np.array_equal(out, want[pick]), f
        for c in range(0, n, 97):
            assert [(int(s), int(d)) for _, s, d in ev[c]] == want_ev[pick[c]], (c, f)
        total += sum(len(e) for e in want_ev)
    assert total > 100
    for c in list(range(0, n, 1021)) + [n - 1]:
        assert np.array_equal(bank.get_state(c), orcs[pick[c]].snapshot()), c


def test_sigtone_tx_device_frames_with_stride(built):
    """Sender frames resident on the device, rows a stride apart: only the rows' samples are touched."""
    import ctypes
    from oracle import restated as orc
    from spandsp_amd import engine
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n, samples, stride = 70, 160, 168
    bank = engine.SigToneTxBank(3, n)
    orcs = [orc.SigToneTx(3) for _ in range(n)]
    rng = np.random.default_rng(77)
    modes = rng.choice([0x00, 0x01, 0x04, 0x05, 0x11, 0x15], n).astype(np.int32)
    bank.set_modes(modes, np.zeros(n, np.int32))
    for c in range(n):
        orcs[c].set_mode(int(modes[c]), 0)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), n*stride*2) == 0
    L = engine.lib()
    for f in range(12):
        frame = rng.integers(-20000, 20000, (n, stride)).astype(np.int16)
        assert hip.hipMemcpy(buf, frame.ctypes.data, frame.nbytes, 1) == 0
        assert L.spangpu_sigtone_tx(bank.h, buf, engine.MEM_DEVICE, samples, stride) == 0     # no durations set: no requests
        out = np.zeros_like(frame)
        assert hip.hipMemcpy(out.ctypes.data, buf, out.nbytes, 2) == 0
        for c, o in enumerate(orcs):
            assert np.array_equal(out[c, :samples], o.tx(frame[c, :samples])), (c, f)
        assert np.array_equal(out[:, samples:], frame[:, samples:])
    hip.hipFree(buf)
    for c in range(n):
        assert np.array_equal(bank.get_state(c), orcs[c].snapshot()[:5]), c
