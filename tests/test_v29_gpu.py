"""V.29 receiver banks on the GPU against the oracle: the put_bit / status event stream of every channel and all 281
state words (float state compared as bit patterns) must be identical after every call."""
import os

import numpy as np
import pytest

from test_oracle_pin import GOLDEN, bits, use_golden_modem_tables

pytestmark = pytest.mark.gpu


def channel_signals(bit_rate, n_ch, seed):
    """Independent channels derived from the committed reference transmission: per-channel delay, gain and noise."""
    g = np.load(os.path.join(GOLDEN, "v29_%d.npz" % bit_rate))
    base = g["amp"].astype(np.float64)
    rng = np.random.default_rng(seed)
    n = len(base) + 64
    out = np.zeros((n_ch, n), np.int16)
    for c in range(n_ch):
        delay = int(rng.integers(0, 64))
        gain = 10.0**(rng.uniform(-14.0, 3.0)/20.0) if c else 1.0
        noise = rng.normal(0.0, rng.choice([0.0, 3.0, 30.0, 120.0]), n) if c else 0.0
        x = np.zeros(n)
        x[delay:delay + len(base)] = base
        out[c] = np.clip(np.rint(x*gain + noise), -32768, 32767).astype(np.int16)
    return out


def oracle_run(bit_rate, x, chunks):
    from oracle import restated as orc
    o = orc.V29(bit_rate)
    per_call = []
    k = i = 0
    while k < len(x):
        n = chunks[i % len(chunks)]
        o.sink.clear()
        o.rx(x[k:k + n])
        ev = o.sink.events()["a"].astype(np.int8)
        f, w = o.snapshot()
        per_call.append((ev, bits(f), w))
        k += n
        i += 1
    return per_call


@pytest.mark.parametrize("bit_rate", [9600, 7200, 4800])
@pytest.mark.parametrize("chunks", [(160,), (400, 3, 1, 97)])
def test_v29_bank_matches_oracle(built, bit_rate, chunks):
    from spandsp_amd import engine
    use_golden_modem_tables()
    n_ch = 70                                   # two workgroups, the second ragged
    sig = channel_signals(bit_rate, n_ch, seed=bit_rate + len(chunks))
    want = [oracle_run(bit_rate, sig[c], chunks) for c in range(n_ch)]
    bank = engine.V29Bank(n_ch, bit_rate)
    k = i = 0
    check = sorted(set([0, 1, 63, 64, n_ch - 1]))
    total_bits = 0
    while k < sig.shape[1]:
        n = chunks[i % len(chunks)]
        bank.rx_host(sig[:, k:k + n])
        got = bank.events()
        for c in range(n_ch):
            assert np.array_equal(got[c], want[c][i][0]), (bit_rate, "events", c, i)
            total_bits += len(got[c])
        if i % 7 == 0 or k + n >= sig.shape[1]:
            for c in check:
                f, w = bank.get_state(c)
                bad_w = np.nonzero(w != want[c][i][2])[0]
                assert bad_w.size == 0, (bit_rate, "int words", c, i, bad_w[:8])
                bad_f = np.nonzero(bits(f) != want[c][i][1])[0]
                assert bad_f.size == 0, (bit_rate, "float words", c, i, bad_f[:8])
        k += n
        i += 1
    assert total_bits > 1500*n_ch//2
    bank.close()


def test_v29_golden_direct(built):
    """The committed reference outputs, straight against the GPU (no oracle in the loop)."""
    from spandsp_amd import engine
    for bit_rate in (9600, 7200, 4800):
        g = np.load(os.path.join(GOLDEN, "v29_%d.npz" % bit_rate))
        x = g["amp"]
        bank = engine.V29Bank(3, bit_rate)
        ev = [[] for _ in range(3)]
        for k in range(0, len(x), 160):
            blk = x[k:k + 160]
            bank.rx_host(np.stack([blk, blk, blk]))
            for c, e in enumerate(bank.events()):
                ev[c].append(e)
        for c in range(3):
            assert np.array_equal(np.concatenate(ev[c]), g["events"])
        f, w = bank.get_state(2)
        assert np.array_equal(w, g["iwords"])
        assert np.array_equal(bits(f), g["fwords"])
        bank.close()


def test_v29_restart(built):
    from spandsp_amd import engine
    g = np.load(os.path.join(GOLDEN, "v29_9600.npz"))
    x = g["amp"]
    bank = engine.V29Bank(2, 9600)
    ev = []
    for rep in range(2):
        for k in range(0, len(x), 160):
            blk = x[k:k + 160]
            bank.rx_host(np.stack([blk, blk]))
            ev.append(bank.events()[1])
        if rep == 0:
            first = np.concatenate(ev)
            ev = []
            bank.restart(1)
    assert np.array_equal(first, g["events"])
    assert np.array_equal(np.concatenate(ev), g["events"])     # a restarted receiver trains again, identically
    bank.close()


def test_modem_edge_cases(built):
    """Empty calls, one-channel banks, one very long call, bad arguments."""
    from oracle import restated as orc
    from spandsp_amd import engine
    use_golden_modem_tables()
    g = np.load(os.path.join(GOLDEN, "v29_7200.npz"))
    x = g["amp"]
    bank = engine.V29Bank(1, 7200)
    bank.rx_host(np.zeros((1, 0), np.int16))                # nothing: no launch, no events yet
    with pytest.raises(engine.SpanGpuError):
        bank.events()
    bank.rx_host(x[None, :])                                # the whole transmission in one call (6130 samples)
    ev = bank.events()[0]
    assert np.array_equal(ev, g["events"])
    f, w = bank.get_state(0)
    assert np.array_equal(w, g["iwords"]) and np.array_equal(bits(f), g["fwords"])
    with pytest.raises(engine.SpanGpuError):
        engine.V29Bank(4, 1234)                             # no such bit rate
    with pytest.raises(engine.SpanGpuError):
        engine.V29Bank(0, 9600)
    with pytest.raises(engine.SpanGpuError):
        bank.get_state(5)
    bank.close()
    # a V.17 bank refuses a restart at a rate its tables are not for; a V.29 channel may change rate
    b17 = engine.V17Bank(2, 9600)
    assert engine.lib().spangpu_modem_restart_ex(b17.h, 0, 14400, 0) < 0
    assert engine.lib().spangpu_modem_restart_ex(b17.h, 0, 9600, 1) == 0
    b17.close()


def test_v29_known_answer(built):
    """BASELINE.md section 2 on the GPU: v29_tx (9600 bps) + AWGN at -50 dBm0, 40 s -> 381 500 bits, the statuses
    CARRIER_UP, TRAINING_IN_PROGRESS, TRAINING_SUCCEEDED and no other, and not one bit in error against the transmitter's
    data source; the whole put_bit stream equal to the oracle's."""
    from oracle import restated as orc
    from spandsp_amd import engine
    from test_oracle_pin import V29_KNOWN, v29_known_signal, prbs15_bits
    K = V29_KNOWN
    y = v29_known_signal(built)
    n_ch = 4
    bank = engine.V29Bank(n_ch, K["bit_rate"])
    o = orc.V29(K["bit_rate"])
    ev = [[] for _ in range(n_ch)]
    frame = 1600
    for k in range(0, len(y), frame):
        bank.rx_host(np.tile(y[k:k + frame], (n_ch, 1)))
        for c, e in enumerate(bank.events()):
            ev[c].append(e)
        o.rx(y[k:k + frame])
    want = o.sink.events()["a"].astype(np.int8)
    for c in range(n_ch):
        a = np.concatenate(ev[c])
        assert np.array_equal(a, want), c
    a = np.concatenate(ev[0])
    assert int((a >= 0).sum()) == K["bits"] and [int(v) for v in a[a < 0]] == K["status"]
    data = a[np.nonzero(a == -4)[0][0] + 1:]
    src = prbs15_bits(K["tx_seed"], len(data) + 400)
    hit = [k for k in range(400) if np.array_equal(src[k:k + 64], data[:64])]
    assert hit and np.array_equal(src[hit[0]:hit[0] + len(data)], data), "bit errors"
    bank.close()
