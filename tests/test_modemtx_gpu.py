"""Modem transmitter banks against the oracle: v29_tx, v27ter_tx and v17_tx.

Bar: bit-exact int16 samples and state words (float words as bits).  The oracle (oracle/v29tx_oracle.c) is pinned to
the real reference in test_oracle_pin.py; the pulse shaper tables are this library's own builder's, which
test_oracle_pin.py pins to the reference's generated tables.
"""
import ctypes

import numpy as np
import pytest

from test_oracle_pin import use_golden_modem_tables

pytestmark = pytest.mark.gpu

FRAMES = [160, 160, 77, 1, 8, 333, 160, 1024, 5, 160]


def setup_oracle():
    from oracle import restated as orc
    from spandsp_amd import engine
    use_golden_modem_tables()
    orc.set_v29_tx_table(engine.modem_tx_table(0))
    orc.set_v27ter_tx_tables(engine.modem_tx_table(1), engine.modem_tx_table(2))
    return orc, engine


@pytest.mark.parametrize("modem,bit_rate,tep", [("v29", 9600, False), ("v29", 7200, True), ("v29", 4800, False), ("v29", 9600, True),
                                                ("v27ter", 4800, False), ("v27ter", 2400, True), ("v27ter", 2400, False),
                                                ("v17", 14400, False), ("v17", 12000, True), ("v17", 9600, False), ("v17", 7200, False),
                                                ("v17", 4800, True)])
def test_modem_tx_bank(built, modem, bit_rate, tep):
    orc, engine = setup_oracle()
    n = 150
    seeds = (np.arange(n)*7919 + 13) & 0x7FFF
    seeds[5] = 0                                     # an all-zero register sends a constant bit: still the reference's behaviour
    if modem == "v29":
        bank = engine.V29TxBank(n, bit_rate, tep, seeds)
        tx = [orc.V29Tx(bit_rate, tep, int(s)) for s in seeds]
        other = 7200 if bit_rate != 7200 else 9600
        train = (480 if tep else 0) + 48 + 128 + 384 + 48
    elif modem == "v17":
        n = 80
        seeds = seeds[:n]
        bank = engine.V17TxBank(n, bit_rate, tep, seeds)
        tx = [orc.V17Tx(bit_rate, tep, int(s)) for s in seeds]
        other = 9600 if bit_rate != 9600 else 14400
        train = ((528 if tep else 0) + 256 + 2976 + 64 + 48)*10//3 + 10
    else:
        bank = engine.V27terTxBank(n, bit_rate, tep, seeds)
        tx = [orc.V27terTx(bit_rate, tep, int(s)) for s in seeds]
        other = 2400 if bit_rate == 4800 else 4800
        train = ((320 if tep else 0) + 32 + 50 + 1074 + 8)*(5 if bit_rate == 4800 else 7)
    for c in range(0, n, 9):
        bank.power(c, -20.0 + (c % 13))
        tx[c].power(-20.0 + (c % 13))
    total = 0
    for k, m in enumerate(FRAMES*{"v29": 3, "v27ter": 6, "v17": 8}[modem]):
        pcm = bank.tx_host(m)
        for c in range(n):
            want = tx[c].tx(m)
            assert np.array_equal(pcm[c], want), (k, c)
        total += m
        if k == 12:
            for c in range(3, n, 17):
                if modem == "v17":
                    bank.restart(c, other, not tep, short_train=True)       # v17_tx_restart(s, rate, tep, short_train)
                    tx[c].restart(other, not tep, True)
                else:
                    bank.restart(c, other, not tep)
                    tx[c].restart(other, not tep)
        if k % 5 == 4:
            for c in range(0, n, 11):
                assert np.array_equal(bank.get_state(c), tx[c].snapshot()), (k, c)
    assert total > train + 1000                                              # well into the data
    assert bank.get_state(0)[24] == 0                                        # in_training is off


def test_v29_tx_feeds_v29_rx_on_device(built):
    """v29_tx bank -> HBM -> v29_rx bank, the samples never leaving the device: every receiver trains and then
    delivers exactly the scrambler-free bit stream its transmitter's LFSR produced."""
    orc, engine = setup_oracle()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n, samples, frames = 512, 160, 40
    seeds = ((np.arange(n)*2654435761 + 12345) & 0x7FFF) | 1
    tx = engine.V29TxBank(n, 9600, False, seeds)
    rx = engine.V29Bank(n, 9600)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), n*samples*2) == 0
    bits = [[] for _ in range(n)]
    for _ in range(frames):
        tx.tx_device(buf, samples, samples)
        tx.sync()
        rx.rx_device(buf, samples, samples)
        for c, e in enumerate(rx.events()):
            bits[c].extend(int(v) for v in e)
    hip.hipFree(buf)
    for c in range(n):
        ev = bits[c]
        assert -4 in ev, c                                  # SIG_STATUS_TRAINING_SUCCEEDED
        data = np.array([b for b in ev[ev.index(-4) + 1:] if b >= 0])
        # the transmitter's bit source, replayed on the host
        st = int(seeds[c])
        want = []
        for _ in range(len(data) + 200):
            b = ((st >> 14) ^ (st >> 13)) & 1
            st = ((st << 1) | b) & 0x7FFF
            want.append(b)
        want = np.array(want)
        assert len(data) > 1000
        # the receiver starts delivering somewhere in the stream: find the alignment once, then demand equality
        hit = [k for k in range(200) if np.array_equal(want[k:k + 64], data[:64])]
        assert hit, c
        assert np.array_equal(want[hit[0]:hit[0] + len(data)], data), c


def test_v27ter_tx_feeds_v27ter_rx_on_device(built):
    """The same loop for V.27ter at 4800 bps: v27ter_tx bank -> HBM -> v27ter_rx bank."""
    orc, engine = setup_oracle()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n, samples, frames = 256, 160, 70
    seeds = ((np.arange(n)*2654435761 + 777) & 0x7FFF) | 1
    tx = engine.V27terTxBank(n, 4800, False, seeds)
    rx = engine.V27terBank(n, 4800)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), n*samples*2) == 0
    bits = [[] for _ in range(n)]
    for _ in range(frames):
        tx.tx_device(buf, samples, samples)
        tx.sync()
        rx.rx_device(buf, samples, samples)
        for c, e in enumerate(rx.events()):
            bits[c].extend(int(v) for v in e)
    hip.hipFree(buf)
    for c in range(n):
        ev = bits[c]
        assert -4 in ev, c
        data = np.array([b for b in ev[ev.index(-4) + 1:] if b >= 0])
        st = int(seeds[c])
        want = []
        for _ in range(len(data) + 200):
            b = ((st >> 14) ^ (st >> 13)) & 1
            st = ((st << 1) | b) & 0x7FFF
            want.append(b)
        want = np.array(want)
        assert len(data) > 500
        hit = [k for k in range(200) if np.array_equal(want[k:k + 64], data[:64])]
        assert hit, c
        assert np.array_equal(want[hit[0]:hit[0] + len(data)], data), c


def test_v17_tx_feeds_v17_rx_on_device(built):
    """And for V.17 at 14400 bps, long training: v17_tx bank -> HBM -> v17_rx bank (trellis decoder and all)."""
    orc, engine = setup_oracle()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n, samples, frames = 128, 160, 95
    seeds = ((np.arange(n)*2654435761 + 4242) & 0x7FFF) | 1
    tx = engine.V17TxBank(n, 14400, False, seeds)
    rx = engine.V17Bank(n, 14400)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), n*samples*2) == 0
    bits = [[] for _ in range(n)]
    for _ in range(frames):
        tx.tx_device(buf, samples, samples)
        tx.sync()
        rx.rx_device(buf, samples, samples)
        for c, e in enumerate(rx.events()):
            bits[c].extend(int(v) for v in e)
    hip.hipFree(buf)
    for c in range(n):
        ev = bits[c]
        assert -4 in ev, c
        data = np.array([b for b in ev[ev.index(-4) + 1:] if b >= 0])
        st = int(seeds[c])
        want = []
        for _ in range(len(data) + 400):
            b = ((st >> 14) ^ (st >> 13)) & 1
            st = ((st << 1) | b) & 0x7FFF
            want.append(b)
        want = np.array(want)
        assert len(data) > 1000
        hit = [k for k in range(400) if np.array_equal(want[k:k + 64], data[:64])]
        assert hit, c
        assert np.array_equal(want[hit[0]:hit[0] + len(data)], data), c


@pytest.mark.parametrize("modem,bit_rate", [("v29", 9600), ("v27ter", 4800), ("v17", 14400)])
def test_modem_tx_line_levels_and_carriers(built, modem, bit_rate):
    """spangpu_modemtx_line(): every channel's level (xxx_tx_power) and carrier frequency in one call.  The level words are the
    oracle's after power(); the carrier word is the DDS phase rate of the frequency asked for, and with that word put into the
    oracle's state the samples are the oracle's, bit for bit (the modulator itself is unchanged)."""
    orc, engine = setup_oracle()
    n = 96
    rng = np.random.default_rng(bit_rate)
    seeds = rng.integers(1, 0x7FFF, n).astype(np.uint32)
    make_bank, make_orc, nominal = {"v29": (engine.V29TxBank, orc.V29Tx, 1700.0), "v27ter": (engine.V27terTxBank, orc.V27terTx, 1800.0),
                                    "v17": (engine.V17TxBank, orc.V17Tx, 1800.0)}[modem]
    bank = make_bank(n, bit_rate, False, seeds)
    tx = [make_orc(bit_rate, False, int(s)) for s in seeds]
    level = rng.uniform(-30.0, -10.0, n).astype(np.float32)
    hz = (nominal + rng.uniform(-7.0, 7.0, n)).astype(np.float32)
    hz[0] = nominal
    before = bank.get_state(0)
    bank.line(level, hz)
    for c in range(n):
        tx[c].power(float(level[c]))
        w = bank.get_state(c)
        rate = int(w[28])
        assert abs(rate - float(hz[c])*2.0**32/8000.0) <= 512.0, (c, rate)       # float32 arithmetic of dds_phase_ratef()
        tx[c].buf[28] = rate
        assert np.array_equal(w, tx[c].snapshot()), c
    assert bank.get_state(0)[28] == before[28]                                   # the nominal carrier gives the word init gave
    for m in (160, 160, 77, 1, 333, 160, 1024, 160):
        pcm = bank.tx_host(m)
        for c in range(n):
            assert np.array_equal(pcm[c], tx[c].tx(m)), (m, c)
    bank.line(None, None)
    with pytest.raises(Exception):
        bank.line(None, np.zeros(n, np.float32))
    bank.close()


def test_v29_tx_with_line_offsets_feeds_v29_rx_on_device(built):
    """SURVEY 8(d)-4's line population without the host: carrier 1700 +- 7 Hz and -30 .. -10 dBm0 per channel from the
    transmitter bank, straight into the receiver bank; every receiver pulls its carrier in, trains and delivers its
    transmitter's bit stream without an error."""
    orc, engine = setup_oracle()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n, samples, frames = 512, 160, 60
    rng = np.random.default_rng(29)
    seeds = ((np.arange(n)*2654435761 + 12345) & 0x7FFF) | 1
    tx = engine.V29TxBank(n, 9600, False, seeds)
    tx.line(rng.uniform(-30.0, -10.0, n), 1700.0 + rng.uniform(-7.0, 7.0, n))
    rx = engine.V29Bank(n, 9600)
    rx.set_signal_cutoff(-1, -45.5)         # as fax_modems.c:416: v29_rx_init()'s -28.5 dBm0 is above the quieter lines (no noise here)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), n*samples*2) == 0
    got = [[] for _ in range(n)]
    for _ in range(frames):
        tx.tx_device(buf, samples, samples)
        tx.sync()
        rx.rx_device(buf, samples, samples)
        for c, e in enumerate(rx.events()):
            got[c].extend(int(v) for v in e)
    hip.hipFree(buf)
    for c in range(n):
        ev = got[c]
        assert -4 in ev and -5 not in ev, c
        data = np.array([b for b in ev[ev.index(-4) + 1:] if b >= 0])
        st = int(seeds[c])
        want = []
        for _ in range(len(data) + 200):
            b = ((st >> 14) ^ (st >> 13)) & 1
            st = ((st << 1) | b) & 0x7FFF
            want.append(b)
        want = np.array(want)
        assert len(data) > 6000
        hit = [k for k in range(200) if np.array_equal(want[k:k + 64], data[:64])]
        assert hit, c
        assert np.array_equal(want[hit[0]:hit[0] + len(data)], data), c
