"""FSK receiver banks against the oracle: fsk_rx in its three framing modes.

Bar: bit-exact -- the event stream (bits, carrier up/down, framed characters) and every state word, the
correlation window included.  The oracle (oracle/fsk_oracle.c) is pinned to the real reference in
test_oracle_pin.py.
"""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

SIZES = [160, 160, 77, 1, 8, 333, 160, 1024, 5, 160]


def run_both(bank, orcs, sig, sizes, check_every=1):
    n = len(orcs)
    pos = 0
    k = 0
    got = [[] for _ in range(n)]
    while pos < sig.shape[1]:
        m = min(sizes[k % len(sizes)], sig.shape[1] - pos)
        bank.rx_host(sig[:, pos:pos + m])
        for c, o in enumerate(orcs):
            o.rx(sig[c, pos:pos + m])
        ev = bank.events()
        for c in range(n):
            got[c].append(ev[c])
        pos += m
        k += 1
        if k % check_every == 0:
            for c in range(0, n, max(1, n//9)):
                assert np.array_equal(bank.get_state(c), orcs[c].snapshot()), (c, pos)
    total = 0
    for c, o in enumerate(orcs):
        want = np.array([e["a"] for e in o.sink.events() if e["kind"] == 3], np.int64)
        have = np.concatenate(got[c]).astype(np.int64) if got[c] else np.zeros(0, np.int64)
        assert np.array_equal(have, want), (c, len(have), len(want))
        total += len(want)
    for c in range(n):
        assert np.array_equal(bank.get_state(c), orcs[c].snapshot()), c
    return total


@pytest.mark.parametrize("which,mode", [(1, 1), (1, 0), (0, 1), (2, 0), (6, 1), (3, 0), (7, 0), (10, 1)])
def test_fsk_bit_modes(built, which, mode):
    from oracle import restated as orc
    from spandsp_amd import engine
    sp = engine.fsk_preset(which)
    n = 150 if sp.baud_rate >= 30000 else 70
    n_samples = 160*60 if sp.baud_rate >= 30000 else 160*150
    sig = synth.fsk_channels(n, n_samples, 100 + which*3 + mode, sp.freq_zero, sp.freq_one, sp.baud_rate)
    bank = engine.FskBank(which, n, mode)
    orcs = [orc.Fsk(which, mode) for _ in range(n)]
    total = run_both(bank, orcs, sig, SIZES if which == 1 else [160], check_every=3)
    assert total > n*20                             # carriers came up and bits were delivered


@pytest.mark.parametrize("which,mode,hz,baud_pct", [(1, 1, 12, 1.0), (1, 0, -15, -1.5), (0, 1, -8, 0.7), (6, 1, 20, -1.0), (2, 2, 10, 0.5)])
def test_fsk_off_frequency_and_off_rate(built, which, mode, hz, baud_pct):
    """The far end's modem a few hertz off both frequencies (a carrier system's shift) and its baud clock a per cent off: the
    correlators no longer sit on their bins and the baud phase is nudged at nearly every transition (fsk.c:540-590) -- events and
    every state word against the oracle, sync / async / framed."""
    from oracle import restated as orc
    from spandsp_amd import engine
    sp = engine.fsk_preset(which)
    n = 70
    n_samples = 160*60 if sp.baud_rate >= 30000 else 160*150
    baud = int(round(sp.baud_rate*(1.0 + baud_pct/100.0)))
    sig = synth.fsk_channels(n, n_samples, 300 + which*5 + mode, sp.freq_zero + hz, sp.freq_one + hz, baud, framed=(mode == 2))
    bank = engine.FskBank(which, n, mode)
    orcs = [orc.Fsk(which, mode) for _ in range(n)]
    total = run_both(bank, orcs, sig, SIZES if which == 1 else [160], check_every=3)
    assert total > n*10


@pytest.mark.parametrize("which,data_bits,parity", [(1, 8, 0), (0, 7, 1), (7, 5, 0), (2, 8, 2)])
def test_fsk_framed_mode(built, which, data_bits, parity):
    from oracle import restated as orc
    from spandsp_amd import engine
    sp = engine.fsk_preset(which)
    n = 100
    n_samples = 160*80 if sp.baud_rate >= 30000 else 160*300
    sig = synth.fsk_channels(n, n_samples, 200 + which, sp.freq_zero, sp.freq_one, sp.baud_rate, framed=True,
                             data_bits=data_bits, parity=parity)
    bank = engine.FskBank(which, n, engine.FSK_FRAME_MODE_FRAMED)
    orcs = [orc.Fsk(which, 2) for _ in range(n)]
    if (data_bits, parity) != (8, 0):
        # ASYNC_PARITY_EVEN = 1, ASYNC_PARITY_ODD = 2 (async.h:151-157)
        for c in range(n):
            bank.set_frame_parameters(c, data_bits, parity, 1)
            orcs[c].set_frame_parameters(data_bits, parity, 1)
    total = run_both(bank, orcs, sig, [160], check_every=10)
    assert total > n*4
    errs = sum(int(bank.get_state(c)[27]) for c in range(n))
    assert errs > 0                                 # the framing-error path ran
    if parity:
        assert sum(int(bank.get_state(c)[26]) for c in range(n)) > 0


def test_fsk_control_calls(built):
    """restart / set_signal_cutoff / fillin / mixed framing modes in one bank, against the oracle doing the same."""
    from oracle import restated as orc
    from spandsp_amd import engine
    which = engine.FSK_V21CH2
    sp = engine.fsk_preset(which)
    n = 64 + 9
    sig = synth.fsk_channels(n, 160*90, 301, sp.freq_zero, sp.freq_one, sp.baud_rate)
    bank = engine.FskBank(which, n, engine.FSK_FRAME_MODE_SYNC)
    orcs = [orc.Fsk(which, 1) for _ in range(n)]
    for c in range(0, n, 3):
        bank.restart(c, engine.FSK_FRAME_MODE_ASYNC)
        orcs[c].restart(which, 0)
    for c in range(1, n, 5):
        bank.set_signal_cutoff(c, -45.0)
        orcs[c].set_signal_cutoff(-45.0)
    half = 160*45
    run_both(bank, orcs, sig[:, :half], [160], check_every=5)
    for c in range(0, n, 4):
        bank.fillin(c, 160)
        orcs[c].fillin(160)
    for c in range(2, n, 11):
        bank.restart(c, engine.FSK_FRAME_MODE_SYNC)
        orcs[c].restart(which, 1)
    for o in orcs:
        o.sink.clear()
    total = run_both(bank, orcs, sig[:, half:], [160, 80, 240], check_every=5)
    assert total > n*10
    with pytest.raises(engine.SpanGpuError):
        w = bank.get_state(0)
        w[15] += 1
        bank.set_state(0, w)


def test_fsk_device_frames_and_ragged_bank(built):
    """A bank that does not fill its last wave, frames handed over as a device pointer with a stride."""
    import ctypes
    from oracle import restated as orc
    from spandsp_amd import engine
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    which = engine.FSK_V21CH2
    sp = engine.fsk_preset(which)
    n, frames, stride = 200, 50, 168
    sig = synth.fsk_channels(n, 160*frames, 401, sp.freq_zero, sp.freq_one, sp.baud_rate)
    bank = engine.FskBank(which, n)
    orcs = [orc.Fsk(which, 1) for _ in range(n)]
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), n*stride*2) == 0
    padded = np.zeros((n, stride), np.int16)
    total = 0
    for k in range(frames):
        padded[:, :160] = sig[:, k*160:(k + 1)*160]
        assert hip.hipMemcpy(buf, padded.ctypes.data, padded.nbytes, 1) == 0
        bank.rx_device(buf, 160, stride)
        ev = bank.events()
        for c in range(n):
            orcs[c].sink.clear()
            orcs[c].rx(sig[c, k*160:(k + 1)*160])
            want = np.array([e["a"] for e in orcs[c].sink.events()], np.int64)
            assert np.array_equal(ev[c].astype(np.int64), want), (k, c)
            total += len(want)
    assert total > n*20
    for c in range(n):
        assert np.array_equal(bank.get_state(c), orcs[c].snapshot()), c


@pytest.mark.parametrize("which,mode", [(1, 1), (0, 2), (3, 0)])
def test_fsk_one_and_two_waves_agree(built, which, mode):
    """The receiver cut over two wavefronts (the library's choice) against the whole receiver in one lane: same events,
    same state words, ragged frame sizes and per-channel lengths included."""
    from spandsp_amd import engine
    sp = engine.fsk_preset(which)
    n = 333
    sig = synth.fsk_channels(n, 160*40, 900 + which, sp.freq_zero, sp.freq_one, sp.baud_rate)
    out = []
    try:
        for waves in (1, 2):
            engine.tune_fsk_waves(waves)
            bank = engine.FskBank(which, n, mode)
            pos = 0
            k = 0
            evs = []
            while pos < sig.shape[1]:
                m = min(SIZES[k % len(SIZES)], sig.shape[1] - pos)
                if k % 4 == 3:
                    lens = np.random.default_rng(k).integers(0, m + 1, n).astype(np.int32)
                    bank.rx_host_var(sig[:, pos:pos + m], lens)
                else:
                    bank.rx_host(sig[:, pos:pos + m])
                evs.append([e.copy() for e in bank.events()])
                pos += m
                k += 1
            out.append((evs, [bank.get_state(c) for c in range(n)]))
    finally:
        engine.tune_fsk_waves(0)
    for a, b in zip(out[0][0], out[1][0]):
        for c in range(n):
            assert np.array_equal(a[c], b[c]), c
    for c in range(n):
        assert np.array_equal(out[0][1][c], out[1][1][c]), c
    assert sum(len(e[c]) for e in out[0][0] for c in range(n)) > n*5
