"""Signal-source banks against the oracle: tone_gen / dtmf_tx / bell_mf_tx / r2_mf_tx.

Bar: bit-exact int16 samples, per-channel lengths and generator state (integer words and float bits).
The oracle (oracle/tonegen_oracle.c) is pinned to the real reference in test_oracle_pin.py.
"""
import ctypes

import numpy as np
import pytest

from test_oracle_pin import use_golden_modem_tables

pytestmark = pytest.mark.gpu

FRAMES = [160, 160, 77, 1, 8, 333, 160, 1024, 5, 160]


def oracle_state_words(g):
    """The words of txgen_dev.hpp's layout that mirror tone_gen_state_t."""
    w = np.zeros(19, np.int64)
    for i in range(4):
        w[i] = g.tone[i].phase_rate
        w[4 + i] = np.float32(g.tone[i].gain).view(np.int32)
        w[8 + i] = np.uint32(g.phase[i]).astype(np.int64)
        w[12 + i] = g.duration[i]
    w[16] = g.repeat
    w[17] = g.current_section
    w[18] = g.current_position
    return w


def bank_state_words(bank, ch):
    w = bank.get_state(ch)[:19].astype(np.int64)
    w[8:12] &= 0xFFFFFFFF
    return w


def check_frame(bank, senders, samples, gens):
    pcm, lens = bank.tx_host(samples)
    for c, s in enumerate(senders):
        want = s.tx(samples)
        assert lens[c] == len(want), (c, lens[c], len(want))
        assert np.array_equal(pcm[c, :len(want)], want), c
        assert not pcm[c, len(want):].any()
    for c in range(0, len(senders), max(1, len(senders)//7)):
        g = gens(senders[c])
        if g is not None:
            assert np.array_equal(bank_state_words(bank, c), oracle_state_words(g)), c


def test_tone_gen_bank(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    use_golden_modem_tables()
    descs = [(350, -13, 440, -13, 100, 0, 0, 0, True), (480, -10, 620, -10, 500, 500, 0, 0, True),
             (425, -10, 0, 0, 200, 200, 600, 1000, False), (400, -10, -17, 50, 300, 100, 0, 0, True),
             (1000, 0, 2000, 0, 1, 0, 0, 0, True), (440, -20, 480, -20, 30, 40, 0, 0, False),
             (950, -8, 1400, -8, 330, 30, 330, 1000, True)]
    n = 150
    bank = engine.TxBank(engine.TX_TONE_GEN, n)
    senders = []
    # idle channels first: tone_gen on a finished generator returns 0
    for c in range(n):
        g = orc.ToneGen(orc.tone_desc(1000, -10, 0, 0, 1, 0, 0, 0, False))
        g.s.current_section = -1
        senders.append(g)
    pcm, lens = bank.tx_host(16)
    assert not lens.any() and not pcm.any()
    at = 0
    for k, d in enumerate(descs):
        cnt = 21 if k < len(descs) - 1 else n - at
        bank.tone(*d, first=at, n=cnt)
        for c in range(at, at + cnt):
            senders[c] = orc.ToneGen(orc.tone_desc(*d))
        at += cnt
    for samples in FRAMES*3:
        check_frame(bank, senders, samples, lambda s: s.s)
    with pytest.raises(engine.SpanGpuError):
        bank.tone(440, -10, 0, 0, 0, 0, 0, 0, True)      # the reference would never return from tone_gen()
    with pytest.raises(engine.SpanGpuError):
        bank.put("1")


@pytest.mark.parametrize("kind", ["dtmf", "bell"])
def test_digit_sender_banks(built, kind):
    from oracle import restated as orc
    from spandsp_amd import engine
    use_golden_modem_tables()
    rng = np.random.default_rng(7 if kind == "dtmf" else 8)
    n = 200
    alphabet = list("0123456789ABCD*#xz") if kind == "dtmf" else list("0123456789ABC*#xz")
    bank = engine.TxBank(engine.TX_DTMF if kind == "dtmf" else engine.TX_BELL_MF, n)
    senders = [(orc.DtmfTx() if kind == "dtmf" else orc.BellMfTx()) for _ in range(n)]
    if kind == "dtmf":
        bank.set_level(-7, 3, first=10, n=50)
        bank.set_timing(40, 30, first=40, n=60)
        bank.set_timing(0, 13, first=120, n=10)
        bank.set_timing(0, 0, first=130, n=5)
        bank.set_timing(-1, -1, first=135, n=5)
        for c in range(n):
            if 10 <= c < 60:
                senders[c].set_level(-7, 3)
            if 40 <= c < 100:
                senders[c].set_timing(40, 30)
            if 120 <= c < 130:
                senders[c].set_timing(0, 13)
            if 130 <= c < 135:
                senders[c].set_timing(0, 0)
    for rnd in range(6):
        digs = ["".join(rng.choice(alphabet, int(rng.integers(0, 70 if rnd < 4 else 129)))) for _ in range(n)]
        if rnd == 1:
            digs[3] = ""
        res = bank.put_each(digs)
        for c in range(n):
            assert res[c] == senders[c].put(digs[c]), (rnd, c)
        if rnd == 2:
            # the same digits to a range, some of which have no room left
            rc = bank.put("159D" * 20, first=0, n=n)
            want = [senders[c].put("159D" * 20) for c in range(n)]
            assert rc == max(want)
        for samples in FRAMES[:int(rng.integers(3, len(FRAMES)))]:
            check_frame(bank, senders, samples, lambda s: s.s.tones)
    # run every queue dry
    for _ in range(40):
        check_frame(bank, senders, 4000, lambda s: s.s.tones)
    pcm, lens = bank.tx_host(160)
    assert not lens.any()


@pytest.mark.parametrize("fwd", [True, False])
def test_r2_mf_sender_bank(built, fwd):
    from oracle import restated as orc
    from spandsp_amd import engine
    use_golden_modem_tables()
    n = 96
    bank = engine.TxBank(engine.TX_R2_MF_FWD if fwd else engine.TX_R2_MF_BACK, n)
    senders = [orc.R2MfTx(fwd) for _ in range(n)]
    check_frame(bank, senders, 100, lambda s: None)
    keys = "1234567890BCDEF"
    for rnd in range(5):
        for lo in range(0, n, 8):
            d = keys[(lo//8 + rnd*5) % 15] if (lo//8 + rnd) % 4 else ("\0" if rnd % 2 else "x")
            bank.put(d, first=lo, n=8)
            for c in range(lo, lo + 8):
                senders[c].put(d)
        for samples in (133, 160, 7, 400):
            check_frame(bank, senders, samples, lambda s: s.s.tone if s.s.digit else None)


def test_dtmf_loopback_on_device(built):
    """dtmf_tx bank -> HBM -> dtmf_rx bank without the samples leaving the device: the receivers get exactly the
    digits that were queued, and the samples are those the oracle's sender makes."""
    from oracle import restated as orc
    from spandsp_amd import engine
    use_golden_modem_tables()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n, samples, frames = 1000, 160, 120
    rng = np.random.default_rng(11)
    digs = ["".join(rng.choice(list("0123456789ABCD*#"), int(rng.integers(1, 16)))) for _ in range(n)]
    tx = engine.TxBank(engine.TX_DTMF, n)
    rx = engine.ToneBank(engine.DTMF, n)
    tx.set_stream(engine.lib().spangpu_bank_get_stream(rx.h))
    assert not tx.put_each(digs).any()
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), n*samples*2) == 0
    host = np.zeros((n, samples), np.int16)
    probe = [0, 17, n - 1]
    senders = {c: orc.DtmfTx() for c in probe}
    for c in probe:
        senders[c].put(digs[c])
    got = [""]*n
    for _ in range(frames):
        tx.tx_device(buf, samples, samples)
        rx.rx_device(buf.value, samples)
        rx.sync()
        assert hip.hipMemcpy(host.ctypes.data, buf, host.nbytes, 2) == 0
        for c in probe:
            want = senders[c].tx(samples)
            assert np.array_equal(host[c, :len(want)], want) and not host[c, len(want):].any()
        for r in rx.blocks():
            # a digit is appended on a CHANGE block with a non-zero code (dtmf.c:318-340)
            if (r["flags"] & engine.BLK_CHANGE) and r["code"]:
                got[r["channel"]] += chr(r["code"])
    hip.hipFree(buf)
    assert got == digs
