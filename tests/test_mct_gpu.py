"""Modem connect tone detector banks against the oracle: modem_connect_tones_rx.

Bar: the (tone, level) reports, the `hit` latch and every state word bit-exact (integer words, the float filter
state as bits, and the V.21 receiver's words where it runs).  The oracle (oracle/mct_oracle.c) is pinned to the
real reference in test_oracle_pin.py.
"""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def run_both(bank, orcs, sig, sizes, use_callback=True):
    n = len(orcs)
    pos = 0
    k = 0
    got = [[] for _ in range(n)]
    latched = []
    while pos < sig.shape[1]:
        m = min(sizes[k % len(sizes)], sig.shape[1] - pos)
        bank.rx_host(sig[:, pos:pos + m])
        for c, o in enumerate(orcs):
            o.rx(sig[c, pos:pos + m])
        for c, e in enumerate(bank.events()):
            got[c].extend((int(t), int(lv)) for t, lv in e)
        pos += m
        k += 1
        if k % 7 == 0:
            for c in range(0, n, max(1, n//6)):
                assert np.array_equal(bank.get_state(c), orcs[c].snapshot()), (c, pos)
            if not use_callback:
                for c in range(0, n, 3):
                    h = bank.get(c)
                    assert h == orcs[c].get(), (c, pos)
                    latched.append(h)
    for c in range(n):
        assert np.array_equal(bank.get_state(c), orcs[c].snapshot()), c
    if use_callback:
        for c, o in enumerate(orcs):
            want = [(int(e["a"]), int(e["b"])) for e in o.sink.events() if e["kind"] == 1]
            assert got[c] == want, (c, got[c][:4], want[:4])
        return got
    return latched


@pytest.mark.parametrize("tone_type,kind", [(1, "cng"), (2, "mix"), (3, "ans_pr"), (4, "ansam"), (7, "mix"), (6, "preamble"),
                                            (8, "bell"), (9, "calling"), (1, "mix")])
def test_mct_reports(built, tone_type, kind):
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 100
    sig = synth.connect_tone_channels(n, 8000*5, 500 + tone_type, kind)
    bank = engine.MctBank(tone_type, n)
    orcs = [orc.Mct(tone_type) for _ in range(n)]
    got = run_both(bank, orcs, sig, [160, 160, 80, 1, 333, 160])
    seen = {t for g in got for t, _ in g}
    if (tone_type, kind) != (1, "mix"):
        assert len(seen - {0}) >= 1 and 0 in seen        # tones were declared and withdrawn
    if tone_type in (2, 7) and kind == "mix":
        assert {2, 3, 4, 5} <= seen                      # ANS, ANS/, ANSam, ANSam/ all told apart
    if tone_type == 7:
        assert 6 in seen


@pytest.mark.parametrize("tone_type,kind", [(2, "mix"), (7, "mix"), (9, "calling")])
def test_mct_hit_latch(built, tone_type, kind):
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 70
    sig = synth.connect_tone_channels(n, 8000*4, 600 + tone_type, kind)
    bank = engine.MctBank(tone_type, n, use_callback=False)
    orcs = [orc.Mct(tone_type, use_callback=False) for _ in range(n)]
    latched = run_both(bank, orcs, sig, [160], use_callback=False)
    assert any(latched)


@pytest.mark.parametrize("tone_type,kind", [(2, "mix"), (7, "mix"), (1, "cng"), (6, "preamble")])
def test_mct_on_shifted_lines(built, tone_type, kind):
    """The same lines through a carrier system that shifts every frequency by a few hertz (tests/impair.py; 2100 Hz becomes
    2090 ... 2112 Hz, the AM and the phase reversals stay): the notch and band filters are off their centres, levels move, some
    tones are declared later or not at all -- reports and state words against the oracle all the same."""
    import impair
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 64
    sig = synth.connect_tone_channels(n, 8000*4, 700 + tone_type, kind)
    rng = np.random.default_rng(tone_type)
    for c in range(n):
        sig[c] = impair.line(sig[c], float(rng.choice([-12.0, -6.0, 4.0, 9.0, 15.0])), float(rng.choice([0.0, 80.0, -80.0])))
    bank = engine.MctBank(tone_type, n)
    orcs = [orc.Mct(tone_type) for _ in range(n)]
    got = run_both(bank, orcs, sig, [160, 160, 80, 1, 333, 160])
    assert any(len(g) for g in got)
