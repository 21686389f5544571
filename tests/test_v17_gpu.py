"""V.27ter receiver banks on the GPU against the oracle: event stream of every channel and all 547 state words
(float state as bit patterns) identical after every call; plus the committed reference outputs directly."""
import os

import numpy as np
import pytest

from test_oracle_pin import GOLDEN, bits, use_golden_modem_tables

pytestmark = pytest.mark.gpu


def channel_signals(bit_rate, n_ch, seed):
    g = np.load(os.path.join(GOLDEN, "v17_%d.npz" % bit_rate))
    base = g["amp"].astype(np.float64)
    rng = np.random.default_rng(seed)
    n = len(base) + 64
    out = np.zeros((n_ch, n), np.int16)
    for c in range(n_ch):
        delay = int(rng.integers(0, 64))
        gain = 10.0**(rng.uniform(-14.0, 3.0)/20.0) if c else 1.0
        noise = rng.normal(0.0, rng.choice([0.0, 3.0, 20.0, 150.0]), n) if c else 0.0
        x = np.zeros(n)
        x[delay:delay + len(base)] = base
        out[c] = np.clip(np.rint(x*gain + noise), -32768, 32767).astype(np.int16)
    return out


def oracle_run(bit_rate, x, chunks):
    from oracle import restated as orc
    o = orc.V17(bit_rate)
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


@pytest.mark.parametrize("bit_rate", [14400, 12000, 9600, 7200, 4800])
@pytest.mark.parametrize("chunks", [(160,), (400, 3, 1, 97)])
def test_v17_bank_matches_oracle(built, bit_rate, chunks):
    from spandsp_amd import engine
    use_golden_modem_tables()
    n_ch = 40
    sig = channel_signals(bit_rate, n_ch, seed=bit_rate + len(chunks))
    want = [oracle_run(bit_rate, sig[c], chunks) for c in range(n_ch)]
    bank = engine.V17Bank(n_ch, bit_rate)
    k = i = 0
    check = sorted(set([0, 1, 15, 16, 31, 32, n_ch - 1]))
    total = 0
    outcomes = set()
    while k < sig.shape[1]:
        n = chunks[i % len(chunks)]
        bank.rx_host(sig[:, k:k + n])
        got = bank.events()
        for c in range(n_ch):
            assert np.array_equal(got[c], want[c][i][0]), (bit_rate, "events", c, i)
            total += len(got[c])
            outcomes.update(int(v) for v in got[c] if v < 0)
        if i % 7 == 0 or k + n >= sig.shape[1]:
            for c in check:
                f, w = bank.get_state(c)
                bad_w = np.nonzero(w != want[c][i][2])[0]
                assert bad_w.size == 0, (bit_rate, "int words", c, i, bad_w[:8])
                bad_f = np.nonzero(bits(f) != want[c][i][1])[0]
                assert bad_f.size == 0, (bit_rate, "float words", c, i, bad_f[:8])
        k += n
        i += 1
    assert total > 1200*n_ch//3 and {-1, -2, -3, -4} <= outcomes
    bank.close()


def test_v17_golden_direct(built):
    from spandsp_amd import engine
    for bit_rate in (14400, 9600, 4800):
        g = np.load(os.path.join(GOLDEN, "v17_%d.npz" % bit_rate))
        x = g["amp"]
        bank = engine.V17Bank(3, bit_rate)
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
        bank.restart(2)
        ev2 = []
        for k in range(0, len(x), 160):
            blk = x[k:k + 160]
            bank.rx_host(np.stack([blk, blk, blk]))
            ev2.append(bank.events()[2])
        assert np.array_equal(np.concatenate(ev2), g["events"])
        bank.close()
