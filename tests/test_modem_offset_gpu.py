"""The modem receiver banks on lines with a carrier offset and a sample clock offset (tests/impair.py).

Every other receiver test feeds the reference modulator's output with delay, gain and noise: carrier exactly nominal, the
receiver's own sample clock.  There the carrier loop (src/v29rx.c:297-331 track_carrier; v17rx.c:311-339; v27ter_rx.c:296-325)
and the symbol timing loop (src/godard.c:165-220; v27ter_rx.c:486-529) rest at their fixed points.  Here every channel has its
own carrier offset (1700 +- 7 Hz; a few beyond that, some far enough to FAIL training) and its own clock
offset (+-100 ppm: the timing loop walks through the pulse shaper's coefficient sets again and again, the equaliser's
taps rotate).  The put_bit / status stream of every channel after every call and the state words (floats as bit patterns)
must equal the oracle's; the committed outputs of the real reference on such lines (tests/golden/modem_offset_*.npz) are
held against the GPU directly."""
import os

import numpy as np
import pytest

import impair
from test_oracle_pin import (GOLDEN, MODEM_OFFSET_CASES, MODEM_OFFSET_GOLDEN, bits, modem_offset_expectations, modem_offset_name,
                             use_v17_tx_tables)

pytestmark = pytest.mark.gpu


def make_bank(name, n_ch, bit_rate, cutoffs=None):
    from spandsp_amd import engine
    bank = {"v29": engine.V29Bank, "v27ter": engine.V27terBank, "v17": engine.V17Bank}[name](n_ch, bit_rate)
    if cutoffs is not None:
        bank.set_signal_cutoffs(cutoffs)
    return bank


def make_oracle(name, bit_rate, cutoff=None):
    from oracle import restated as orc
    o = {"v29": orc.V29, "v27ter": orc.V27ter, "v17": orc.V17}[name](bit_rate)
    if cutoff is not None:
        o.set_signal_cutoff(float(cutoff))
    return o


def make_tx(name, bit_rate, seed):
    from oracle import restated as orc
    return {"v29": orc.V29Tx, "v27ter": orc.V27terTx, "v17": orc.V17Tx}[name](bit_rate, False, seed)


def offset_channels(name, bit_rate, n_ch, n_signal, seed):
    """One transmission per channel (the oracle's restated modulator: own scrambler seed, own level), each through a line
    of its own: start delay 0..159 samples, carrier offset, clock offset, noise."""
    rng = np.random.default_rng(seed)
    n = 200 + n_signal + 900
    out = np.zeros((n_ch, n), np.int16)
    lines = []
    for c in range(n_ch):
        tx = make_tx(name, bit_rate, int(rng.integers(1, 0x7FFF)))
        level = float(rng.uniform(-30.0, -10.0))
        tx.power(level)
        sig = tx.tx(n_signal)
        delay = int(rng.integers(0, 160))
        if c % 16 == 5:
            hz = float(rng.choice([-1.0, 1.0])*rng.uniform(22.0, 40.0))         # training fails
        elif c % 16 == 9:
            hz = float(rng.choice([-1.0, 1.0])*rng.uniform(8.0, 14.0))          # beyond the contract's range, still pulled in
        else:
            hz = float(rng.uniform(-7.0, 7.0))
        ppm = float(rng.uniform(-100.0, 100.0)) if c % 8 else 0.0
        x = np.zeros(n, np.float64)
        x[40 + delay:40 + delay + len(sig)] = sig
        y = impair.line(x, hz, ppm).astype(np.float64)
        rms = np.sqrt(np.mean(sig.astype(np.float64)**2))
        # (the reference's V.27ter receiver gives up training below some 35 dB: its lines are kept cleaner than SURVEY 8(d)-4's 25..40 dB)
        snr_db = rng.uniform(38.0, 50.0) if name == "v27ter" else rng.uniform(25.0, 40.0)
        y += rng.normal(0.0, rms/10.0**(snr_db/20.0), n)
        out[c] = np.clip(np.rint(y), -32768, 32767).astype(np.int16)
        lines.append((hz, ppm, delay, snr_db, level))
    return out, lines


def oracle_calls(name, bit_rate, x, chunks, cutoff=None):
    o = make_oracle(name, bit_rate, cutoff)
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


@pytest.mark.parametrize("name,bit_rate,n_signal,chunks", [("v29", 9600, 20000, (160,)), ("v29", 7200, 12000, (400, 3, 1, 97)),
                                                           ("v29", 4800, 12000, (160,)),
                                                           ("v27ter", 4800, 20000, (160,)), ("v27ter", 2400, 16000, (400, 3, 1, 97)),
                                                           ("v17", 14400, 22000, (160,)), ("v17", 9600, 18000, (400, 3, 1, 97)),
                                                           ("v17", 7200, 18000, (160,))])
def test_offset_bank_matches_oracle(built, name, bit_rate, n_signal, chunks):
    use_v17_tx_tables(built)
    n_ch = 70                                   # two workgroups of the four-lane kernels, the second ragged
    sig, lines = offset_channels(name, bit_rate, n_ch, n_signal, seed=bit_rate + len(name))
    # v29_rx_init() leaves the carrier detector at -28.5 dBm0 (v29rx.c:1129), above the quieter lines here; at the FAX front
    # end's -45.5 dBm0 (fax_modems.c:416) the noise of the louder lines holds it up for ever.  As an installation would, every
    # V.29 line gets v29_rx_set_signal_cutoff(its level - 12 dB); V.17 and V.27ter stay at their own -45.5 dBm0.
    cutoffs = np.array([ln[4] - 12.0 for ln in lines], np.float32) if name == "v29" else None
    want = [oracle_calls(name, bit_rate, sig[c], chunks, None if cutoffs is None else cutoffs[c]) for c in range(n_ch)]
    bank = make_bank(name, n_ch, bit_rate, cutoffs)
    if cutoffs is not None:
        bank.set_signal_cutoff(n_ch - 1, float(cutoffs[n_ch - 1]))      # (the one-channel form of the same call)
    check = sorted(set([0, 1, 5, 9, 15, 16, 63, 64, n_ch - 1]))
    trained, failed = set(), set()
    k = i = 0
    while k < sig.shape[1]:
        n = chunks[i % len(chunks)]
        bank.rx_host(sig[:, k:k + n])
        got = bank.events()
        for c in range(n_ch):
            assert np.array_equal(got[c], want[c][i][0]), (name, bit_rate, "events", c, i, lines[c])
            trained.update([c] if -4 in got[c] else [])
            failed.update([c] if -5 in got[c] else [])
        if i % 9 == 0 or k + n >= sig.shape[1]:
            for c in check:
                f, w = bank.get_state(c)
                bad_w = np.nonzero(w != want[c][i][2])[0]
                assert bad_w.size == 0, (name, bit_rate, "int words", c, i, bad_w[:8], lines[c])
                bad_f = np.nonzero(bits(f) != want[c][i][1])[0]
                assert bad_f.size == 0, (name, bit_rate, "float words", c, i, bad_f[:8], lines[c])
        k += n
        i += 1
    # the population is what it claims to be: most lines train with their loops off centre, the far-off ones give up
    assert len(trained) >= n_ch*3//4 and len(failed) >= n_ch//16, (len(trained), len(failed))
    bank.close()


@pytest.mark.parametrize("case", [MODEM_OFFSET_CASES[i] for i in MODEM_OFFSET_GOLDEN], ids=modem_offset_name)
def test_offset_golden_direct(built, case):
    """The real reference's outputs on an impaired line, straight against the GPU (no oracle in the loop)."""
    name, bit_rate = case[0], case[1]
    g = np.load(os.path.join(GOLDEN, modem_offset_name(case) + ".npz"))
    x = g["amp"]
    bank = make_bank(name, 3, bit_rate)
    ev = [[] for _ in range(3)]
    for k in range(0, len(x), 160):
        blk = x[k:k + 160]
        bank.rx_host(np.stack([blk, blk, blk]))
        for c, e in enumerate(bank.events()):
            ev[c].append(e)
    for c in range(3):
        got = np.concatenate(ev[c])
        assert np.array_equal(got, g["events"]), (case, c)
    modem_offset_expectations(case, np.concatenate(ev[0]).astype(np.int32))
    f, w = bank.get_state(1)
    assert np.array_equal(w, g["iwords"])
    assert np.array_equal(bits(f), g["fwords"])
    bank.close()
