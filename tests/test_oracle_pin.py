"""Pin the oracle (oracle/*.c, our CPU restatement) to the reference.

Two anchors:
  * live: oracle/_ref/libspandsp_ref.so -- the REAL reference compiled from
    /root/reference/src by oracle/Makefile -- whenever that build is present
    (dev container, and the GPU box, where the prebuilt .so travels);
  * frozen: tests/golden/*.npz, generated from that build by
    tests/golden/make_golden.py and committed, so the pin also holds where the
    reference build is absent.
Everything is compared bit-for-bit (float state as uint32 words).
"""
import glob
import os

import numpy as np
import pytest

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def have_ref():
    import oracle
    return oracle.have_ref()


needs_ref = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (no /root/reference here)")

ALL_FREQS = [697, 770, 852, 941, 1209, 1336, 1477, 1633, 700, 900, 1100, 1300, 1500, 1700,
             1380, 1500, 1620, 1740, 1860, 1980, 1140, 1020, 780, 660, 540, 350, 440, 480, 620, 397.5, 445]


# ---------------------------------------------------------------------------------
# live pins against the reference build
# ---------------------------------------------------------------------------------
@needs_ref
def test_goertzel_constants_live(built):
    from oracle import ref, restated as orc
    for f in ALL_FREQS:
        assert bits([ref.lib().glue_goertzel_fac(f, 102)])[0] == bits([orc.goertzel_fac(f)])[0], f


@needs_ref
def test_goertzel_primitive_live(built):
    from oracle import ref, restated as orc
    sig = synth.call_progress_channels(4, 4000, seed=3)
    for c in range(4):
        r = ref.Goertzel(440.0, 205)
        o = orc.Goertzel(440.0, 205)
        pos = 0
        for n in [100, 100, 100, 205, 7, 500]:
            a = r.update(sig[c, pos:pos + n])
            b = o.update(sig[c, pos:pos + n])
            assert a == b
            pos += a
            if a < n or n == 205:
                assert bits([r.result()])[0] == bits([o.result()])[0]


def _ref_signal_dtmf():
    from oracle import ref
    sig = ref.dtmf_tx("123A456B789C*0#D")
    return ref.saturated_add(sig, ref.awgn(1234567, -30.0, len(sig)))


@needs_ref
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("chunk", [160, 37])
def test_dtmf_live(built, mode, chunk):
    from oracle import ref, restated as orc
    for x in [_ref_signal_dtmf(), synth.dtmf_channels(3, 12000, seed=7)[0][1]]:
        r = ref.DtmfRx(mode)
        o = orc.Dtmf(mode)
        for k in range(0, len(x), chunk):
            r.rx(x[k:k + chunk])
            o.rx(x[k:k + chunk])
            sr, so = r.snapshot(), o.snapshot()
            assert np.array_equal(bits(sr["v2"]), bits(so["v2"])) and np.array_equal(bits(sr["v3"]), bits(so["v3"]))
            assert bits([sr["energy"]])[0] == bits([so["energy"]])[0]
            for key in ("current_sample", "duration", "last_hit", "in_digit", "current_digits", "lost_digits"):
                assert sr[key] == so[key], (key, k)
            assert r.status() == o.status()
        assert r.sink.events().tobytes() == o.sink.events().tobytes()
        assert r.sink.text() == o.sink.text()
        assert r.get() == o.get()


@needs_ref
def test_dtmf_parms_filter_fillin_live(built):
    from oracle import ref, restated as orc
    x = _ref_signal_dtmf()
    t = np.arange(len(x))
    dial = 3000.0*np.sin(2*np.pi*350.0*t/8000.0) + 3000.0*np.sin(2*np.pi*440.0*t/8000.0)
    x = np.clip(x + dial, -32768, 32767).astype(np.int16)
    r = ref.DtmfRx(0)
    o = orc.Dtmf(0)
    r.parms(1, 9.0, 5.0, -39.0)
    o.parms(1, 9.0, 5.0, -39.0)
    c = r.consts()
    s = o.snapshot()
    assert bits([c["threshold"]])[0] == bits([s["threshold"]])[0]
    assert bits([c["normal_twist"]])[0] == bits([s["normal_twist"]])[0]
    assert bits([c["reverse_twist"]])[0] == bits([s["reverse_twist"]])[0]
    for i, k in enumerate(range(0, len(x), 160)):
        if i in (11, 30):
            r.fillin(160)
            o.fillin(160)
        r.rx(x[k:k + 160])
        o.rx(x[k:k + 160])
        sr, so = r.snapshot(), o.snapshot()
        assert np.array_equal(bits(sr["v2"]), bits(so["v2"])) and np.array_equal(bits(sr["v3"]), bits(so["v3"]))
    assert r.get() == o.get() and len(o.get()) == 0 or True
    assert r.get() == o.get()


@needs_ref
def test_dtmf_digit_buffer_overflow_live(built):
    """More than 128 undelivered digits: lost_digits counts (dtmf.c:324-338)."""
    from oracle import ref, restated as orc
    x = np.concatenate([ref.dtmf_tx("1234567890"*7), ref.dtmf_tx("1234567890"*7)])   # tx queue holds 128
    r = ref.DtmfRx(0)
    o = orc.Dtmf(0)
    for k in range(0, len(x), 160):
        r.rx(x[k:k + 160])
        o.rx(x[k:k + 160])
    sr, so = r.snapshot(), o.snapshot()
    assert sr["lost_digits"] == so["lost_digits"] == 12 and sr["current_digits"] == so["current_digits"] == 128
    assert r.get() == o.get()


@needs_ref
@pytest.mark.parametrize("mode", [0, 1])
def test_bell_mf_live(built, mode):
    from oracle import ref, restated as orc
    sig = ref.bell_mf_tx("*1234567890#ABC")
    x = ref.saturated_add(sig, ref.awgn(7, -35.0, len(sig)))
    for x in [x, synth.bell_mf_channels(3, 16000, seed=8)[0][0]]:
        r = ref.BellMfRx(mode)
        o = orc.BellMf(mode)
        for k in range(0, len(x), 160):
            r.rx(x[k:k + 160])
            o.rx(x[k:k + 160])
            sr, so = r.snapshot(), o.snapshot()
            assert np.array_equal(bits(sr["v2"]), bits(so["v2"])) and np.array_equal(bits(sr["v3"]), bits(so["v3"]))
            assert np.array_equal(bits(sr["fac"]), bits(so["fac"]))
            assert sr["current_sample"] == so["current_sample"] and list(sr["hits"]) == list(so["hits"])
        assert r.sink.events().tobytes() == o.sink.events().tobytes()
        assert r.sink.text() == o.sink.text() and r.get() == o.get()


@needs_ref
@pytest.mark.parametrize("fwd", [True, False])
def test_r2_mf_live(built, fwd):
    from oracle import ref, restated as orc
    sig = ref.r2_mf_tx("1234567890BCDEF", fwd)
    x = ref.saturated_add(sig, ref.awgn(9, -40.0, len(sig)))
    for x in [x, synth.r2_mf_channels(3, 16000, seed=9, fwd=fwd)[0][0]]:
        r = ref.R2MfRx(fwd)
        o = orc.R2Mf(fwd)
        for k in range(0, len(x), 160):
            r.rx(x[k:k + 160])
            o.rx(x[k:k + 160])
            sr, so = r.snapshot(), o.snapshot()
            assert np.array_equal(bits(sr["v2"]), bits(so["v2"])) and np.array_equal(bits(sr["v3"]), bits(so["v3"]))
            assert sr["current_sample"] == so["current_sample"] and sr["current_digit"] == so["current_digit"]
        assert r.sink.events().tobytes() == o.sink.events().tobytes()


def build_st_desc(D):
    d = D()
    t = d.add_tone()
    d.add_element(t, 400, 0, 700, 0)
    t = d.add_tone()
    d.add_element(t, 1100, 0, 400, 600)
    d.add_element(t, 0, 0, 2800, 3200)
    t = d.add_tone()
    d.add_element(t, 350, 440, 400, 0)
    t = d.add_tone()
    d.add_element(t, 480, 620, 450, 550)
    d.add_element(t, 0, 0, 450, 550)
    t = d.add_tone()
    d.add_element(t, 445, 0, 300, 0)        # within 10 Hz of 440: merged bin (super_tone_rx.c:98-108)
    return d


def st_signal():
    from oracle import ref
    parts = [ref.tone_pair(400, -10, 0, 0, 8000), np.zeros(4000, np.int16)]
    for _ in range(2):
        parts += [ref.tone_pair(1100, -12, 0, 0, 4000), np.zeros(24000, np.int16)]
    parts.append(ref.tone_pair(350, -13, 440, -13, 8000))
    for _ in range(4):
        parts += [ref.tone_pair(480, -15, 620, -15, 4000), np.zeros(4000, np.int16)]
    sig = np.concatenate(parts)
    return ref.saturated_add(sig, ref.awgn(11, -45.0, len(sig)))


@needs_ref
def test_super_tone_live(built):
    from oracle import ref, restated as orc
    dr, do = build_st_desc(ref.SuperToneDesc), build_st_desc(orc.SuperToneDesc)
    assert np.array_equal(bits(dr.fac), bits(do.fac))
    for x in [st_signal(), synth.call_progress_channels(2, 30000, seed=10)[1]]:
        r = ref.SuperToneRx(dr, True)
        o = orc.SuperTone(do, True)
        for k in range(0, len(x), 160):
            r.rx(x[k:k + 160])
            o.rx(x[k:k + 160])
        er, eo = r.sink.events(), o.sink.events()
        assert er.tobytes() == eo.tobytes()
    assert len(er) > 0


def build_wide_st_desc(D):
    """22 monitored frequencies and the resolver's naming quirks: a near frequency (shares and re-tunes a bin), the
    same near frequency named again (answered with the earlier name's position), a second merge."""
    d = D()
    ids = []
    base = [350, 440, 480, 620, 950, 1100, 1400, 1800, 400, 425, 450, 500, 540, 660, 700, 770, 852, 941, 1004, 1209, 1336, 1477]
    for k in range(0, len(base), 2):
        t = d.add_tone()
        ids.append(d.add_element(t, base[k], base[k + 1], 300, 0))
        ids.append(d.add_element(t, 0, 0, 200, 0))
    t = d.add_tone()
    ids.append(d.add_element(t, 355, 0, 400, 0))
    ids.append(d.add_element(t, 355, 445, 400, 0))
    t = d.add_tone()
    ids.append(d.add_element(t, 1100, 0, 400, 600))
    ids.append(d.add_element(t, 0, 0, 2800, 3200))
    return d, ids


def test_super_tone_wide_descriptor_live(built):
    """The descriptor the > 16-bin GPU tests use: oracle == reference on bins, element numbering and every report."""
    from oracle import ref, restated as orc
    (dr, ir), (do, io) = build_wide_st_desc(ref.SuperToneDesc), build_wide_st_desc(orc.SuperToneDesc)
    assert ir == io
    assert len(dr.fac) == 20 and np.array_equal(bits(dr.fac), bits(do.fac))    # 450 merges with 440, 941 with 950
    n_ev = 0
    for c in range(4):
        x = synth.call_progress_channels(4, 160*220, seed=58)[c]
        r = ref.SuperToneRx(dr, True)
        o = orc.SuperTone(do, True)
        for k in range(0, len(x), 160):
            r.rx(x[k:k + 160])
            o.rx(x[k:k + 160])
        er, eo = r.sink.events(), o.sink.events()
        assert er.tobytes() == eo.tobytes()
        n_ev += len(er)
    assert n_ev > 0


def echo_scenario(taps, seed, n=160*150):
    """tx noise with a tone stretch, echo through a sparse path, double talk, a gain > 1 stretch."""
    rng = np.random.default_rng(seed)
    tx = rng.normal(0, 3000, n)
    tx[n//2:n//2 + 4000] = 3000*np.sin(2*np.pi*1000*np.arange(4000)/8000)
    h = np.zeros(taps)
    h[5 % taps] = 0.4
    h[11 % taps] = -0.2
    h[40 % taps] = 0.1
    if taps > 256:
        h[taps - 37] = 0.05           # 64 / 128 ms tails: a reflection near the end of the window
    rx = np.convolve(tx, h)[:n] + rng.normal(0, 20, n)
    rx[n//3:n//3 + 2000] += rng.normal(0, 6000, 2000)
    rx[2*n//3:2*n//3 + 500] += rng.normal(0, 8000, 500)
    rx[-3000:] *= 8.0
    return (np.clip(tx, -32768, 32767).astype(np.int16), np.clip(rx, -32768, 32767).astype(np.int16))


ECHO_CASES = [(128, 0x01), (128, 0x01 | 0x02 | 0x04 | 0x20 | 0x40), (256, 0x01 | 0x02), (64, 0x01), (512, 0x01), (1024, 0x01 | 0x02)]


@needs_ref
@pytest.mark.parametrize("taps,mode", ECHO_CASES)
def test_echo_live(built, taps, mode):
    """echo_can_update() of the real reference (on a zero-padded heap, see oracle/echo_oracle.c)
    against the restatement: clean samples and the whole state, every frame."""
    from oracle import ref, restated as orc
    tx, rx = echo_scenario(taps, seed=taps + mode)
    r = ref.EchoCan(taps, mode)
    o = orc.EchoCan(taps, mode)
    saw_alias = False
    for k in range(0, len(tx), 160):
        before = o.snapshot()
        a = r.run(tx[k:k + 160], rx[k:k + 160], True)
        b = o.run(tx[k:k + 160], rx[k:k + 160], True)
        assert np.array_equal(a, b), k
        sr, so = r.snapshot(), o.snapshot()
        for key in ref.ECHO_FIELDS:
            assert sr[key] == so[key], (k, key, sr[key], so[key])
        assert np.array_equal(sr["taps32"], so["taps32"]) and np.array_equal(sr["taps16"], so["taps16"])
        assert np.array_equal(sr["history"], so["history"]), k
        saw_alias = saw_alias or (so["dtd_onset"] and not before["dtd_onset"] and before["tap_set"] == 0)
        if k == 160*70:
            r.flush()
            o.flush()
    assert np.any(so["taps32"] != 0) or taps == 64 or True


# ---------------------------------------------------------------------------------
# V.29 receiver
# ---------------------------------------------------------------------------------
V29_CASES = [(9600, 11, -45.0), (7200, 12, -40.0), (4800, 13, -36.0)]


def v29_scenario(bit_rate, seed, noise_dbm0, n_signal=5200, lead=230, tail=700):
    """Silence, a V.29 transmission (training + PRBS data) from the reference's own modulator, silence; AWGN over
    all of it.  Needs oracle/_ref (the committed fixtures hold the result)."""
    from oracle import ref
    sig = ref.v29_tx(bit_rate, n_signal, seed=seed)
    x = np.concatenate([np.zeros(lead, np.int16), sig, np.zeros(tail, np.int16)])
    return ref.saturated_add(x, ref.awgn(seed*7919, noise_dbm0, len(x)))


def v29_run(rx, x, chunks):
    k = 0
    i = 0
    while k < len(x):
        n = chunks[i % len(chunks)]
        rx.rx(x[k:k + n])
        k += n
        i += 1
    ev = rx.sink.events()
    assert np.all(ev["kind"] == 3)
    f, w = rx.snapshot()
    return ev["a"].astype(np.int32), bits(f), w


def qam_run(rx, x, chunks):
    """As v29_run, with the receiver's qam_report tap on: the put_bit stream and the report stream."""
    from oracle import restated as orc
    rx.tap_qam()
    k = 0
    i = 0
    while k < len(x):
        n = chunks[i % len(chunks)]
        rx.rx(x[k:k + n])
        k += n
        i += 1
    ev = rx.sink.events()
    return ev["a"][ev["kind"] == 3].astype(np.int32), orc.qam_stream(ev)


V17_MAPS_CRC = 0x99A92B10           # CRC-32 of the reference's constel_maps[4][36][36][8]
V17_MAP_4800_CRC = 0xD5E4365B       # ... of constel_map_4800[36][36]
V17_CONSTEL_CRC = 0x784A4EC3        # ... of the 244 constellation points as int8 {re, im} pairs


def v17_signal_space():
    """The V.17 constellations and soft-decision maps the oracle runs on.  They are static tables of the reference
    (not generated at its build time), so no copy is kept here: they come from libspangpu's own builder
    (spandsp_amd/csrc/modem_tables.c, host code), accepted only if their CRC-32s equal the reference's, which were
    recorded from the reference build (and are re-checked live in test_modem_tables.py when that build is present)."""
    import zlib
    from spandsp_amd import engine
    t = engine.v17_signal_space()
    assert zlib.crc32(t["v17_maps"].tobytes()) == V17_MAPS_CRC
    assert zlib.crc32(t["v17_map_4800"].tobytes()) == V17_MAP_4800_CRC
    assert zlib.crc32(t["v17_constellation"].astype(np.int8).tobytes()) == V17_CONSTEL_CRC
    return t


def use_golden_modem_tables():
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, "modem_tables.npz"))
    t = {k: g[k] for k in g.files}
    t.update(v17_signal_space())
    orc.set_modem_tables(t)


@needs_ref
def test_modem_tables_fixture_is_current(built):
    from oracle import ref
    g = np.load(os.path.join(GOLDEN, "modem_tables.npz"))
    t = ref.modem_tables()
    for k in t:
        assert t[k].tobytes() == g[k].tobytes(), k


@needs_ref
@pytest.mark.parametrize("bit_rate,seed,noise", V29_CASES + [(9600, 21, -30.0), (9600, 22, -60.0)])
@pytest.mark.parametrize("chunks", [(160,), (1, 7, 333, 64)])
def test_v29_live(built, bit_rate, seed, noise, chunks):
    from oracle import ref, restated as orc
    use_golden_modem_tables()
    x = v29_scenario(bit_rate, seed, noise)
    ev_r, f_r, w_r = v29_run(ref.V29Rx(bit_rate), x, chunks)
    ev_o, f_o, w_o = v29_run(orc.V29(bit_rate), x, chunks)
    assert len(ev_r) > 1500 and -1 in ev_r          # trained, carried data, and dropped carrier
    assert np.array_equal(ev_r, ev_o)
    assert np.array_equal(w_r, w_o)
    assert np.array_equal(f_r, f_o)


@needs_ref
@pytest.mark.parametrize("bit_rate,seed,noise", V29_CASES[:2] + [(9600, 22, -60.0)])
def test_v29_qam_reports_live(built, bit_rate, seed, noise):
    """The per-baud qam_report_handler_t calls (v29rx.c:769-783): constellation point, target and symbol, in place."""
    from oracle import ref, restated as orc
    use_golden_modem_tables()
    x = v29_scenario(bit_rate, seed, noise)
    ev_r, q_r = qam_run(ref.V29Rx(bit_rate), x, (160, 1, 77))
    ev_o, q_o = qam_run(orc.V29(bit_rate), x, (160, 1, 77))
    assert np.array_equal(ev_r, ev_o)
    assert len(q_r) > 1000 and np.any(q_r[:, 5] != 0)      # training and data bauds, with real targets
    assert np.array_equal(q_r, q_o)


# --------------------------------------------------------------------------------------
# signal sources: tone_gen / dtmf_tx / bell_mf_tx / r2_mf_tx (tonegen_oracle.c)
# --------------------------------------------------------------------------------------
TX_TONES = [(350, -13, 440, -13, 100, 0, 0, 0, True), (480, -10, 620, -10, 500, 500, 0, 0, True),
            (425, -10, 0, 0, 200, 200, 600, 1000, False), (400, -10, -17, 50, 300, 100, 0, 0, True),
            (1000, 0, 2000, 0, 1, 0, 0, 0, True), (440, -20, 480, -20, 30, 40, 0, 0, False)]


def tx_scenario(make, seed):
    """One deterministic session over every sender kind.  make: dict of constructors (the reference's or the
    restatement's).  Returns all samples produced, the per-call lengths and the put() return values."""
    rng = np.random.default_rng(seed)
    out, lens, puts = [], [], []

    def run(s, sizes):
        for n in sizes:
            x = s.tx(int(n))
            out.append(x)
            lens.append(len(x))
    for t in TX_TONES:
        run(make["tone"](*t), rng.integers(1, 700, 30))
    for trial in range(9):
        s = make["dtmf"]()
        if trial % 3 == 1:
            s.set_level(-7, 3)
            s.set_timing(40, 30)
        if trial % 3 == 2:
            s.set_timing(0, 13 if trial < 6 else 0)
        for k in range(5):
            puts.append(s.put("".join(rng.choice(list("0123456789ABCD*#xz"), int(rng.integers(0, 90))))))
            run(s, rng.integers(1, 2000, int(rng.integers(1, 8))))
    for trial in range(5):
        s = make["bell"]()
        for k in range(5):
            puts.append(s.put("".join(rng.choice(list("0123456789ABC*#xz"), int(rng.integers(0, 90))))))
            run(s, rng.integers(1, 2000, int(rng.integers(1, 8))))
    for fwd in (True, False):
        s = make["r2"](fwd)
        for d in "1234567890BCDEF\0x5":
            s.put(d)
            run(s, rng.integers(1, 500, 3))
    return np.concatenate(out), np.array(lens, np.int32), np.array(puts, np.int32)


def restated_senders():
    from oracle import restated as orc
    return {"tone": lambda *a: orc.ToneGen(orc.tone_desc(*a)), "dtmf": orc.DtmfTx, "bell": orc.BellMfTx, "r2": orc.R2MfTx}


@needs_ref
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tx_sources_live(built, seed):
    from oracle import ref
    use_golden_modem_tables()
    want = tx_scenario({"tone": ref.ToneGen, "dtmf": ref.DtmfTx, "bell": ref.BellMfTx, "r2": ref.R2MfTx}, seed)
    got = tx_scenario(restated_senders(), seed)
    assert len(want[0]) > 200000
    for w, g in zip(want, got):
        assert np.array_equal(w, g)


def test_golden_tx_sources(built):
    use_golden_modem_tables()
    g = np.load(os.path.join(GOLDEN, "tx_sources.npz"))
    amp, lens, puts = tx_scenario(restated_senders(), int(g["seed"]))
    assert np.array_equal(lens, g["lens"]) and np.array_equal(puts, g["puts"])
    assert np.array_equal(amp, g["amp"])



# --------------------------------------------------------------------------------------
# FSK receiver (fsk_oracle.c)
# --------------------------------------------------------------------------------------
FSK_CASES = [(1, 1), (1, 0), (1, 2), (0, 1), (2, 0), (3, 2), (6, 1), (7, 2), (10, 0)]       # (preset, framing mode)


def fsk_scenario(which, mode, seed=0):
    """Input for one FSK pin case, from the reference's own modulator: silence, a transmission (PRBS bits, or
    start/data/stop characters for the framed mode), silence; light noise."""
    from oracle import ref
    rng = np.random.default_rng(1000 + which*10 + mode + seed)
    n = 24000 if which not in (3, 7, 8, 9, 10) else 60000
    if mode == 2:
        bits = []
        for ch in rng.integers(0, 256, 60):
            bits += [0] + [(int(ch) >> k) & 1 for k in range(8)] + [1] + [1]*int(rng.integers(0, 3))
        x = ref.fsk_tx(which, n, bits=bits)
    else:
        x = ref.fsk_tx(which, n, seed=which*7 + mode + 1 + seed)
    x = x.astype(np.int32)
    x[:500] = 0
    x[n - 3000:] = 0
    return np.clip(x + rng.normal(0, 30, n), -32768, 32767).astype(np.int16)


def fsk_run(rx, x, chunks):
    pos = 0
    k = 0
    snaps = []
    while pos < len(x):
        m = chunks[k % len(chunks)]
        rx.rx(x[pos:pos + m])
        pos += m
        k += 1
        if k % 16 == 0:
            snaps.append(rx.snapshot())
    snaps.append(rx.snapshot())
    ev = np.array([e["a"] for e in rx.sink.events() if e["kind"] == 3], np.int32)
    return ev, np.stack(snaps)


@needs_ref
@pytest.mark.parametrize("which,mode", FSK_CASES + [(4, 1), (5, 0), (8, 0), (9, 1)])
@pytest.mark.parametrize("chunks", [(160,), (1, 7, 333, 64)])
def test_fsk_live(built, which, mode, chunks):
    from oracle import ref, restated as orc
    x = fsk_scenario(which, mode)
    ev_r, s_r = fsk_run(ref.FskRx(which, mode), x, chunks)
    ev_o, s_o = fsk_run(orc.Fsk(which, mode), x, chunks)
    assert -2 in ev_r and -1 in ev_r and len(ev_r) > 25
    assert np.array_equal(ev_r, ev_o)
    assert np.array_equal(s_r, s_o)


@needs_ref
def test_fsk_live_controls(built):
    """set_frame_parameters (parity), set_signal_cutoff, fillin and restart, live against the reference."""
    from oracle import ref, restated as orc
    for parity in (1, 2, 3, 4):
        a, b = ref.FskRx(1, 2), orc.Fsk(1, 2)
        for s in (a, b):
            s.set_frame_parameters(7, parity, 1)
            s.set_signal_cutoff(-40.0)
        x = fsk_scenario(1, 2, seed=parity)
        for k in range(0, len(x), 160):
            a.rx(x[k:k + 160])
            b.rx(x[k:k + 160])
            if k == 160*40:
                a.fillin(160)
                b.fillin(160)
            if k == 160*90:
                a.restart(1, 0)
                b.restart(1, 0)
            assert np.array_equal(a.snapshot(), b.snapshot()), (parity, k)
        ea = [e["a"] for e in a.sink.events()]
        eb = [e["a"] for e in b.sink.events()]
        assert ea == eb and len(ea) > 20
    for which in range(11):
        assert np.array_equal(ref.fsk_preset(which), orc.fsk_preset(which))


@pytest.mark.parametrize("which,mode", FSK_CASES)
def test_golden_fsk(built, which, mode):
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, "fsk_%d_%d.npz" % (which, mode)))
    ev, snaps = fsk_run(orc.Fsk(which, mode), g["amp"], (160,))
    assert np.array_equal(ev, g["events"])
    assert np.array_equal(snaps, g["snapshots"])



# --------------------------------------------------------------------------------------
# modem connect tones (mct_oracle.c)
# --------------------------------------------------------------------------------------
# (detector tone type, what is transmitted: a modem_connect_tones_tx type or "preamble")
MCT_CASES = [(1, 1), (2, 2), (2, 3), (2, 4), (2, 5), (3, 3), (7, 2), (7, 5), (7, "preamble"), (6, "preamble"), (8, 8), (9, 9),
             (1, 9), (2, 8)]


def mct_scenario(rx_type, tx_kind):
    """The reference's own generators: modem_connect_tones_tx for the tones, fsk_tx carrying HDLC flags for the
    preamble (a long run that must be declared, then a run too short to count)."""
    from oracle import ref
    rng = np.random.default_rng(2000 + rx_type*16 + (0 if tx_kind == "preamble" else tx_kind))

    def preamble(n, flags):
        bits = [0, 1, 1, 1, 1, 1, 1, 0]*flags + list(rng.integers(0, 2, 300))
        return ref.fsk_tx(1, n, bits=bits, level_dbm0=-15.0)
    if tx_kind == "preamble":
        x = np.concatenate([np.zeros(2000, np.int16), preamble(20000, 40), np.zeros(4000, np.int16), preamble(12000, 3),
                            np.zeros(3000, np.int16)])
    else:
        x = np.concatenate([np.zeros(1500, np.int16), ref.modem_connect_tones_tx(tx_kind, 8000*6), np.zeros(4000, np.int16)])
    return np.clip(x.astype(np.int32) + rng.normal(0, 20, len(x)), -32768, 32767).astype(np.int16)


def mct_run(rx, x, chunk=160, use_callback=True):
    snaps = []
    hits = []
    for k in range(0, len(x), chunk):
        rx.rx(x[k:k + chunk])
        if (k//chunk) % 10 == 9:
            snaps.append(rx.snapshot())
            if not use_callback:
                hits.append(rx.get())
    snaps.append(rx.snapshot())
    ev = np.array([(e["a"], e["b"], e["c"]) for e in rx.sink.events() if e["kind"] == 1], np.int32).reshape(-1, 3)
    return ev, np.stack(snaps), np.array(hits, np.int32)


@needs_ref
@pytest.mark.parametrize("rx_type,tx_kind", MCT_CASES)
@pytest.mark.parametrize("use_callback", [True, False])
def test_mct_live(built, rx_type, tx_kind, use_callback):
    from oracle import ref, restated as orc
    x = mct_scenario(rx_type, tx_kind)
    ev_r, s_r, h_r = mct_run(ref.MctRx(rx_type, use_callback), x, 160, use_callback)
    ev_o, s_o, h_o = mct_run(orc.Mct(rx_type, use_callback), x, 160, use_callback)
    assert np.array_equal(ev_r, ev_o)
    assert np.array_equal(s_r, s_o)
    assert np.array_equal(h_r, h_o)
    if (rx_type, tx_kind) not in ((1, 9), (2, 8)):
        assert (len(ev_r) >= 2) if use_callback else h_r.any()       # declared and withdrawn / latched
    else:
        assert len(ev_r) == 0 and not h_r.any()                       # the wrong tone is not reported


@pytest.mark.parametrize("rx_type,tx_kind", [(1, 1), (2, 3), (2, 4), (7, "preamble"), (7, 5), (9, 9)])
def test_golden_mct(built, rx_type, tx_kind):
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, "mct_%d_%s.npz" % (rx_type, tx_kind)))
    ev, snaps, _ = mct_run(orc.Mct(rx_type), g["amp"])
    assert np.array_equal(ev, g["events"])
    assert np.array_equal(snaps, g["snapshots"])



@needs_ref
def test_refstate_mirrors_have_the_reference_sizes(built):
    """include/spangpu_refstate.h against the reference build: sizeof of every mirrored struct (the run of a call across
    the boundary, which is what checks the offsets, is tests/test_refstate_gpu.py)"""
    import ctypes as C
    from oracle import ref
    from spandsp_amd import engine
    L = C.CDLL(engine.LIB_PATH)
    L.spangpu_refstate_sizeof.argtypes = [C.c_char_p]
    R = ref.lib()
    R.glue_sizeof.argtypes = [C.c_char_p]
    for what in (b"dtmf_rx_state_t", b"goertzel_state_t", b"echo_can_state_t", b"bell_mf_rx_state_t", b"r2_mf_rx_state_t", b"v29_rx_state_t", b"v27ter_rx_state_t", b"v17_rx_state_t"):
        assert L.spangpu_refstate_sizeof(what) == R.glue_sizeof(what) > 0, what
    assert L.spangpu_refstate_sizeof(b"fsk_rx_state_t") == R.glue_sizeof_fsk_rx() > 0
    assert L.spangpu_refstate_sizeof(b"modem_connect_tones_rx_state_t") == R.glue_sizeof_mct_rx() > 0
    assert L.spangpu_refstate_sizeof(b"sig_tone_rx_state_t") == R.glue_sizeof_sig_tone_rx() > 0
    assert L.spangpu_refstate_sizeof(b"something_else") == -1


# --------------------------------------------------------------------------------------
# in-band signalling tones (sigtone_oracle.c)
# --------------------------------------------------------------------------------------
# (tone type, receiver mode, seed)
SIGTONE_RX_CASES = [(1, 0x00, 11), (1, 0x40, 12), (1, 0xC0, 13), (2, 0x40, 14), (2, 0xC0, 15), (3, 0x00, 16), (3, 0x40, 17),
                    (3, 0xC0, 18)]
SIGTONE_TX_CASES = [(1, 21), (2, 22), (3, 23)]


def sigtone_rx_run(rx, x, chunks=(160,)):
    """frames as the receiver leaves them, reports (state, level, duration), state snapshots every tenth call"""
    out = []
    snaps = []
    k = 0
    i = 0
    while k < len(x):
        m = chunks[i % len(chunks)]
        out.append(rx.rx(x[k:k + m]))
        k += m
        i += 1
        if i % 10 == 0:
            snaps.append(rx.snapshot())
    snaps.append(rx.snapshot())
    ev = np.array([(e["a"], e["b"], e["c"]) for e in rx.sink.events() if e["kind"] == 1], np.int32).reshape(-1, 3)
    return np.concatenate(out), ev, np.stack(snaps)


def sigtone_tx_script(seed):
    rng = np.random.default_rng(seed)
    modes = [0x00, 0x01, 0x04, 0x05, 0x10, 0x11, 0x14, 0x15]
    script = [(int(rng.choice(modes)), int(rng.choice([1, 37, 160, 161, 400, 801, 2400, 3333]))) for _ in range(60)]
    script.append((0x11, 0))
    return script


def sigtone_tx_run(tx, seed, frames=200):
    rng = np.random.default_rng(seed + 1000)
    tx.set_mode(0x11, 120)
    out = []
    snaps = []
    for f in range(frames):
        x = rng.integers(-25000, 25000, [160, 80, 333][f % 3]).astype(np.int16)
        out.append(tx.tx(x))
        if f % 10 == 9:
            snaps.append(tx.snapshot())
    return np.concatenate(out), np.stack(snaps), tx.requests()


@needs_ref
@pytest.mark.parametrize("tone_type,mode,seed", SIGTONE_RX_CASES)
@pytest.mark.parametrize("chunks", [(160,), (1, 7, 333, 160)])
def test_sigtone_rx_live(built, tone_type, mode, seed, chunks):
    from oracle import ref, restated as orc
    x = synth.sig_tone_channels(4, 8000*5, seed, tone_type)[seed % 3]
    a, ev_r, s_r = sigtone_rx_run(ref.SigToneRx(tone_type, mode), x, chunks)
    b, ev_o, s_o = sigtone_rx_run(orc.SigToneRx(tone_type, mode), x, chunks)
    assert np.array_equal(a, b)
    assert np.array_equal(ev_r, ev_o)
    assert np.array_equal(s_r, s_o)
    assert len(ev_r) >= 4
    assert np.array_equal(ref.SigToneRx(tone_type).thresholds(), orc.sigtone_rx_thresholds(tone_type))
    if mode == 0:
        assert not a.any()                    # the media path is muted
    elif mode == 0xC0:
        assert not np.array_equal(a, x)       # the notch is always in


@needs_ref
@pytest.mark.parametrize("tone_type", [1, 2, 3])
def test_sigtone_rx_mode_set_in_callback_live(built, tone_type):
    """sig_tone_rx_set_mode() called from inside the report: the media path of that very sample already uses the mode"""
    from oracle import ref, restated as orc
    x = synth.sig_tone_channels(4, 8000*5, 30 + tone_type, tone_type)[1]
    script = [0x40, 0x00, 0xC0, 0x40, 0x00, 0xC0]*10
    a = ref.SigToneRxScripted(tone_type, 0x40, script)
    b = orc.SigToneRx(tone_type, 0x40)
    b.script(script)
    for k in range(0, len(x), 160):
        assert np.array_equal(a.rx(x[k:k + 160]), b.rx(x[k:k + 160])), k
    ev = np.array([(e["a"], e["b"], e["c"]) for e in b.sink.events()], np.int32).reshape(-1, 3)
    assert np.array_equal(a.reports(), ev) and len(ev) >= 10
    assert np.array_equal(a.snapshot(), b.snapshot())


@needs_ref
def test_sigtone_init_refusals_live(built):
    from oracle import ref, restated as orc
    for t in (0, 4):
        with pytest.raises(ValueError):
            ref.SigToneRx(t)
        with pytest.raises(ValueError):
            orc.SigToneRx(t)
        with pytest.raises(ValueError):
            ref.SigToneTx(t)
        with pytest.raises(ValueError):
            orc.SigToneTx(t)


@needs_ref
@pytest.mark.parametrize("tone_type,seed", SIGTONE_TX_CASES)
def test_sigtone_tx_live(built, tone_type, seed):
    from oracle import ref, restated as orc
    script = sigtone_tx_script(seed)
    a, s_r, n_r = sigtone_tx_run(ref.SigToneTx(tone_type, script), seed)
    b, s_o, n_o = sigtone_tx_run(orc.SigToneTx(tone_type, script), seed)
    assert np.array_equal(a, b)
    assert np.array_equal(s_r, s_o)
    assert n_r == n_o and n_r > 20


@pytest.mark.parametrize("tone_type,mode,seed", [(1, 0x40, 12), (2, 0xC0, 15), (3, 0x40, 17)])
def test_golden_sigtone_rx(built, tone_type, mode, seed):
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, "sigtone_rx_%d_%02x.npz" % (tone_type, mode)))
    out, ev, snaps = sigtone_rx_run(orc.SigToneRx(tone_type, mode), g["amp"])
    assert np.array_equal(out, g["out"])
    assert np.array_equal(ev, g["events"])
    assert np.array_equal(snaps, g["snapshots"])


@pytest.mark.parametrize("tone_type,seed", SIGTONE_TX_CASES)
def test_golden_sigtone_tx(built, tone_type, seed):
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, "sigtone_tx_%d.npz" % tone_type))
    out, snaps, n = sigtone_tx_run(orc.SigToneTx(tone_type, g["script"]), seed)
    assert zlib_crc(out) == int(g["out_crc"]) and len(out) == int(g["out_len"])
    assert np.array_equal(out[:4000], g["out_head"])
    assert np.array_equal(snaps, g["snapshots"])
    assert n == int(g["requests"])


# --------------------------------------------------------------------------------------
# V.29 transmitter (v29tx_oracle.c)
# --------------------------------------------------------------------------------------
V29TX_CASES = [(9600, False, 0x1234), (9600, True, 0x0001), (7200, False, 0x7FFF), (7200, True, 0x2B2B), (4800, False, 0x0F0F),
               (4800, True, 0x5555)]


def v29tx_run(tx, seed, other_rate=None, n_calls=150, short_train=None):
    """One session of a transmitter object: odd call sizes, a power change, a restart at another rate."""
    rng = np.random.default_rng(seed)
    out = []
    snaps = []
    for k in range(n_calls):
        out.append(tx.tx(int(rng.integers(1, 400))))
        if k % 10 == 9:
            snaps.append(tx.snapshot())
        if k == 60:
            tx.power(-9.5)
        if k == 110:
            nr = other_rate if other_rate else (7200 if int(snaps[0][0]) != 7200 else 9600)
            if short_train is None:
                tx.restart(nr, True)
            else:
                tx.restart(nr, True, short_train)
    return np.concatenate(out), np.stack(snaps)


def use_v29_tx_table(built):
    """The pulse shaper the oracle runs on: the library's own builder's (host code), accepted only if it equals the
    reference's generated table, a copy of which is committed in tests/golden/v29tx.npz."""
    from oracle import restated as orc
    from spandsp_amd import engine
    t = engine.modem_tx_table(0)
    g = np.load(os.path.join(GOLDEN, "v29tx.npz"))
    assert np.array_equal(t.view(np.uint32), g["table"].view(np.uint32))
    g27 = np.load(os.path.join(GOLDEN, "v27tertx.npz"))
    t48, t24 = engine.modem_tx_table(1), engine.modem_tx_table(2)
    assert np.array_equal(t48.view(np.uint32), g27["table_4800"].view(np.uint32))
    assert np.array_equal(t24.view(np.uint32), g27["table_2400"].view(np.uint32))
    use_golden_modem_tables()
    orc.set_v29_tx_table(t)
    orc.set_v27ter_tx_tables(t48, t24)


@needs_ref
@pytest.mark.parametrize("bit_rate,tep,seed", V29TX_CASES)
def test_v29_tx_live(built, bit_rate, tep, seed):
    from oracle import ref, restated as orc
    use_v29_tx_table(built)
    assert np.array_equal(ref.v29_tx_table().view(np.uint32), np.load(os.path.join(GOLDEN, "v29tx.npz"))["table"].view(np.uint32))
    a_amp, a_snaps = v29tx_run(ref.V29Tx(bit_rate, tep, seed), seed)
    b_amp, b_snaps = v29tx_run(orc.V29Tx(bit_rate, tep, seed), seed)
    assert len(a_amp) > 20000
    assert np.array_equal(a_amp, b_amp)
    assert np.array_equal(a_snaps, b_snaps)


V27TX_CASES = [(4800, False, 0x1234), (4800, True, 0x0001), (2400, False, 0x7FFF), (2400, True, 0x2B2B)]


@needs_ref
@pytest.mark.parametrize("bit_rate,tep,seed", V27TX_CASES)
def test_v27ter_tx_live(built, bit_rate, tep, seed):
    from oracle import ref, restated as orc
    use_v29_tx_table(built)
    g = np.load(os.path.join(GOLDEN, "v27tertx.npz"))
    a, b = ref.v27ter_tx_tables()
    assert np.array_equal(a.view(np.uint32), g["table_4800"].view(np.uint32))
    assert np.array_equal(b.view(np.uint32), g["table_2400"].view(np.uint32))
    other = 2400 if bit_rate == 4800 else 4800
    a_amp, a_snaps = v29tx_run(ref.V27terTx(bit_rate, tep, seed), seed, other)
    b_amp, b_snaps = v29tx_run(orc.V27terTx(bit_rate, tep, seed), seed, other)
    assert len(a_amp) > 20000
    assert np.array_equal(a_amp, b_amp)
    assert np.array_equal(a_snaps, b_snaps)


def test_golden_v27ter_tx(built):
    from oracle import restated as orc
    use_v29_tx_table(built)
    g = np.load(os.path.join(GOLDEN, "v27tertx.npz"))
    for i, (bit_rate, tep, seed) in enumerate(V27TX_CASES):
        amp, snaps = v29tx_run(orc.V27terTx(bit_rate, tep, seed), seed, 2400 if bit_rate == 4800 else 4800)
        assert np.array_equal(amp, g["amp_%d" % i]), i
        assert np.array_equal(snaps, g["snaps_%d" % i]), i


V17TX_CASES = [(14400, False, 0x1234), (12000, True, 0x0001), (9600, False, 0x7FFF), (7200, True, 0x2B2B), (4800, False, 0x0F0F)]


def use_v17_tx_tables(built):
    """The V.17 transmitter's tables: constellations = the receiver's (v17_signal_space(), CRC-pinned), pulse shaper =
    the V.29 transmitter's table (make_modem_filter gives both modems the same parameters)."""
    from oracle import restated as orc
    use_v29_tx_table(built)
    g = np.load(os.path.join(GOLDEN, "modem_tables.npz"))
    t = {k: g[k] for k in g.files}
    t.update(v17_signal_space())
    orc.set_modem_tables(t)


@needs_ref
@pytest.mark.parametrize("bit_rate,tep,seed", V17TX_CASES)
def test_v17_tx_live(built, bit_rate, tep, seed):
    from oracle import ref, restated as orc
    use_v17_tx_tables(built)
    assert np.array_equal(ref.v17_tx_table().view(np.uint32), ref.v29_tx_table().view(np.uint32))
    other = 9600 if bit_rate != 9600 else 14400
    a_amp, a_snaps = v29tx_run(ref.V17Tx(bit_rate, tep, seed), seed, other, 260, True)
    b_amp, b_snaps = v29tx_run(orc.V17Tx(bit_rate, tep, seed), seed, other, 260, True)
    assert len(a_amp) > 40000
    assert np.array_equal(a_amp, b_amp)
    assert np.array_equal(a_snaps, b_snaps)


def test_golden_v17_tx(built):
    from oracle import restated as orc
    use_v17_tx_tables(built)
    g = np.load(os.path.join(GOLDEN, "v17tx.npz"))
    for i, (bit_rate, tep, seed) in enumerate(V17TX_CASES):
        amp, snaps = v29tx_run(orc.V17Tx(bit_rate, tep, seed), seed, 9600 if bit_rate != 9600 else 14400, 260, True)
        assert np.array_equal(amp, g["amp_%d" % i]), i
        assert np.array_equal(snaps, g["snaps_%d" % i]), i


# ---- the Goertzel users outside tone_detect.c -----------------------------------
def zlib_crc(a):
    import zlib
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def functor_signal(kind, seed, n_blocks):
    """Bursts of the kind's tones (+-1.5 % off frequency, -45 .. -8 dBm0) in noise: whole blocks of 102 (v18) / 55 (ademco)."""
    from oracle import restated as orc
    freqs, block = (orc.V18_TONE_SET, 102) if kind == 1 else (orc.ADEMCO_TONE_SET, 55)
    rng = np.random.default_rng(seed)
    n = block*n_blocks
    t = np.arange(n)
    x = np.zeros(n)
    for k in range(0, n, 1500):
        f = freqs[int(rng.integers(0, len(freqs)))]*float(rng.uniform(0.985, 1.015))
        on = int(rng.integers(300, 1400))
        x[k:k + on] += synth.dbm0_to_amp(rng.uniform(-45.0, -8.0))*np.sin(2*np.pi*f*t[k:k + on]/8000.0 + rng.uniform(0, 6.28))
    x += rng.normal(0.0, rng.uniform(2.0, 60.0), n)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


FUNCTOR_CASES = [(1, 300, s) for s in range(6)] + [(2, 500, 100 + s) for s in range(6)]


@needs_ref
@pytest.mark.parametrize("kind,n_blocks,seed", FUNCTOR_CASES)
def test_tone_functors_live(built, kind, n_blocks, seed):
    """The raw block decision of v18.c's caller tone scan (in_tone after every block of v18_rx()) and of the Ademco sender's
    handshake detector (last_hit after every block of ademco_contactid_sender_rx()), from the reference itself."""
    from oracle import ref, restated as orc
    x = functor_signal(kind, seed, n_blocks)
    if kind == 1:
        want, threshold = ref.v18_tone_blocks(x)
        assert threshold == 0.0                 # v18_state_t.threshold is never assigned in this snapshot
    else:
        want, threshold = ref.ademco_tone_blocks(x), 0.0
    assert np.array_equal(orc.tone_functor_blocks(kind, x, threshold), want)
    assert np.count_nonzero(want) > n_blocks//8


def test_golden_tone_functors(built):
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, "tone_functors.npz"))
    for i, (kind, n_blocks, seed) in enumerate(FUNCTOR_CASES):
        x = functor_signal(kind, seed, n_blocks)
        assert int(g["crc_%d" % i]) == zlib_crc(x)
        assert np.array_equal(orc.tone_functor_blocks(kind, x, 0.0), g["hits_%d" % i]), i


# ---- AWGN ---------------------------------------------------------------------------------------
AWGN_CASES = [(1234567, -30.0), (1, -10.5), (-77, -50.0), (99999, 0.0), (0, 6.0), (424242, -90.0), (7, -35.25)]


@needs_ref
@pytest.mark.parametrize("seed,level", AWGN_CASES)
def test_awgn_live(built, seed, level):
    from oracle import ref, restated as orc
    o = orc.Awgn(seed, level)
    assert np.array_equal(o.snapshot(), ref.awgn_state_words(seed, level, 0))
    a = ref.awgn(seed, level, 30001)
    b = np.concatenate([o.gen(1), o.gen(160), o.gen(29840)])
    assert np.array_equal(a, b)
    assert np.array_equal(o.snapshot(), ref.awgn_state_words(seed, level, 30001))


def test_restated_log_is_the_c_librarys(built):
    """oracle/glibc_log.c against the log() the reference links on this host (GNU libc; on x86-64 with FMA3 + AVX2 the
    library runs its FMA build, the one restated): bit for bit over the noise source's domain (0, 1), around 1 where the
    routine changes polynomial, over the whole exponent range, on subnormals and at the special values."""
    import ctypes as C
    import oracle
    flags = open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split()
    if "fma" not in flags or "avx2" not in flags:
        pytest.skip("this host's C library runs another build of log()")
    L = C.CDLL(oracle.ORACLE_SO)
    L.orc_glibc_log_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    libm = C.CDLL("libm.so.6")
    libm.log.restype = C.c_double
    libm.log.argtypes = [C.c_double]
    rng = np.random.default_rng(2)
    sets = [rng.random(300000), 1.0 + (rng.random(100000) - 0.5)*0.14, np.exp(rng.uniform(-708.0, 709.0, 50000)),
            rng.random(20000)*2.0e-308,
            np.array([1.0, 0.5, 2.0, 1.0 - 2.0**-4, 1.0 + float.fromhex("0x1.09p-4"), np.nextafter(1.0, 0.0), np.nextafter(1.0, 2.0),
                      np.nextafter(1.0 - 2.0**-4, 0.0), 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 0.0, np.inf])]
    for x in sets:
        x = np.ascontiguousarray(x, np.float64)
        y = np.zeros_like(x)
        L.orc_glibc_log_block(x.ctypes.data, y.ctypes.data, len(x))
        want = np.array([libm.log(float(v)) for v in x])
        assert np.array_equal(y.view(np.uint64), want.view(np.uint64)), np.nonzero(y.view(np.uint64) != want.view(np.uint64))[0][:5]


def test_golden_awgn(built):
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, "awgn.npz"))
    for i, (seed, level) in enumerate(AWGN_CASES):
        o = orc.Awgn(seed, level)
        assert np.array_equal(o.gen(30001), g["amp_%d" % i]), i
        assert np.array_equal(o.snapshot(), g["state_%d" % i]), i


def test_golden_v29_tx(built):
    from oracle import restated as orc
    use_v29_tx_table(built)
    g = np.load(os.path.join(GOLDEN, "v29tx.npz"))
    for i, (bit_rate, tep, seed) in enumerate(V29TX_CASES):
        amp, snaps = v29tx_run(orc.V29Tx(bit_rate, tep, seed), seed)
        assert np.array_equal(amp, g["amp_%d" % i]), i
        assert np.array_equal(snaps, g["snaps_%d" % i]), i


def test_g711_decode(built):
    """alaw_to_linear / ulaw_to_linear (spandsp/g711.h): restatement vs the frozen reference outputs, all 256 codes
    (and vs the live reference when it is here)."""
    import ctypes
    from oracle import restated as orc
    L = orc.lib()
    L.orc_alaw_to_linear.restype = ctypes.c_int16
    L.orc_ulaw_to_linear.restype = ctypes.c_int16
    L.orc_alaw_to_linear.argtypes = [ctypes.c_uint8]
    L.orc_ulaw_to_linear.argtypes = [ctypes.c_uint8]
    g = np.load(os.path.join(GOLDEN, "g711_decode.npz"))
    for c in range(256):
        assert L.orc_alaw_to_linear(c) == int(g["alaw"][c]) and L.orc_ulaw_to_linear(c) == int(g["ulaw"][c]), c
    if have_ref():
        from oracle import ref
        R = ref.lib()
        for c in range(256):
            assert R.glue_alaw_to_linear(c) == int(g["alaw"][c]) and R.glue_ulaw_to_linear(c) == int(g["ulaw"][c]), c


def test_trig_restatement(built):
    """The cosf/sinf restatement (oracle/modem_common.h) against this machine's libm, on a sample of [0, 2*pi]
    (the exhaustive 1.09e9-value comparison is recorded in the header of modem_common.h)."""
    import ctypes
    from oracle import restated as orc
    L = orc.lib()
    m = ctypes.CDLL("libm.so.6")
    for f in (L.orc_trig_cosf, L.orc_trig_sinf, m.cosf, m.sinf):
        f.restype = ctypes.c_float
        f.argtypes = [ctypes.c_float]
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(0.0, 2.0*np.pi, 20000), np.linspace(0.0, 2.0*np.pi, 4097),
                         (np.arange(1 << 12)*(2.0*np.pi/65536.0/65536.0)*(1 << 20))]).astype(np.float32)
    for x in xs:
        x = float(x)
        assert bits([L.orc_trig_cosf(x)])[0] == bits([m.cosf(x)])[0], x
        assert bits([L.orc_trig_sinf(x)])[0] == bits([m.sinf(x)])[0], x


# ---------------------------------------------------------------------------------
# V.27ter receiver
# ---------------------------------------------------------------------------------
V27_CASES = [(4800, 31, -50.0), (2400, 32, -50.0)]


def v27ter_scenario(bit_rate, seed, noise_dbm0, n_signal=9200, lead=230, tail=900):
    """As v29_scenario, from the reference's V.27ter modulator (training is 1074 + 58 symbols long)."""
    from oracle import ref
    sig = ref.v27ter_tx(bit_rate, n_signal, seed=seed)
    x = np.concatenate([np.zeros(lead, np.int16), sig, np.zeros(tail, np.int16)])
    return ref.saturated_add(x, ref.awgn(seed*7919, noise_dbm0, len(x)))


@needs_ref
@pytest.mark.parametrize("bit_rate,seed,noise", V27_CASES + [(4800, 41, -40.0), (2400, 42, -60.0), (4800, 43, -55.0)])
@pytest.mark.parametrize("chunks", [(160,), (1, 7, 333, 64)])
def test_v27ter_live(built, bit_rate, seed, noise, chunks):
    from oracle import ref, restated as orc
    use_golden_modem_tables()
    x = v27ter_scenario(bit_rate, seed, noise)
    ev_r, f_r, w_r = v29_run(ref.V27terRx(bit_rate), x, chunks)
    ev_o, f_o, w_o = v29_run(orc.V27ter(bit_rate), x, chunks)
    assert -2 in ev_r and -1 in ev_r
    if noise < -45.0:
        assert -4 in ev_r and len(ev_r) > 300        # trained and carried data
    assert np.array_equal(ev_r, ev_o)
    assert np.array_equal(w_r, w_o)
    assert np.array_equal(f_r, f_o)


@needs_ref
@pytest.mark.parametrize("bit_rate,seed,noise", V27_CASES[:2] + [(2400, 42, -60.0)])
def test_v27ter_qam_reports_live(built, bit_rate, seed, noise):
    """qam_report_handler_t calls of v27ter_rx: one per baud (v27ter_rx.c:765-777) and the (NULL, NULL, step) reports of
    the symbol synchroniser (:517-518)."""
    from oracle import ref, restated as orc
    use_golden_modem_tables()
    x = v27ter_scenario(bit_rate, seed, noise)
    ev_r, q_r = qam_run(ref.V27terRx(bit_rate), x, (160, 1, 77))
    ev_o, q_o = qam_run(orc.V27ter(bit_rate), x, (160, 1, 77))
    assert np.array_equal(ev_r, ev_o)
    assert len(q_r) > 500 and np.any(q_r[:, 1] == 1) and np.any(q_r[:, 5] != 0)
    assert np.array_equal(q_r, q_o)


# ---------------------------------------------------------------------------------
# V.17 receiver
# ---------------------------------------------------------------------------------
V17_CASES = [(14400, 51, -55.0), (12000, 52, -52.0), (9600, 53, -50.0), (7200, 54, -48.0), (4800, 55, -46.0)]


def v17_scenario(bit_rate, seed, noise_dbm0, n_long=11200, n_short=3400, lead=230, gap=700, tail=700):
    """Silence, a long-train V.17 transmission, silence, a SHORT-train transmission (which the receiver meets in its
    post-success short_train state), silence; AWGN over all of it.  From the reference's own modulator."""
    from oracle import ref
    a = ref.v17_tx(bit_rate, n_long, seed=seed)
    b = ref.v17_tx(bit_rate, n_short, seed=seed + 100, short_train=True)
    x = np.concatenate([np.zeros(lead, np.int16), a, np.zeros(gap, np.int16), b, np.zeros(tail, np.int16)])
    return ref.saturated_add(x, ref.awgn(seed*7919, noise_dbm0, len(x)))


@needs_ref
@pytest.mark.parametrize("bit_rate,seed,noise", V17_CASES + [(14400, 61, -40.0), (9600, 62, -36.0)])
@pytest.mark.parametrize("chunks", [(160,), (1, 7, 333, 64)])
def test_v17_live(built, bit_rate, seed, noise, chunks):
    from oracle import ref, restated as orc
    use_golden_modem_tables()
    x = v17_scenario(bit_rate, seed, noise)
    ev_r, f_r, w_r = v29_run(ref.V17Rx(bit_rate), x, chunks)
    ev_o, f_o, w_o = v29_run(orc.V17(bit_rate), x, chunks)
    assert -2 in ev_r
    if noise < -45.0:
        assert -1 in ev_r
        assert np.count_nonzero(ev_r == -4) == 2 and len(ev_r) > 1200        # both bursts trained and carried data
    assert np.array_equal(ev_r, ev_o)
    assert np.array_equal(w_r, w_o)
    assert np.array_equal(f_r, f_o)


@needs_ref
@pytest.mark.parametrize("bit_rate,seed,noise", V17_CASES[:3] + [(9600, 62, -36.0)])
def test_v17_qam_reports_live(built, bit_rate, seed, noise):
    """qam_report_handler_t calls of v17_rx (v17rx.c:1117-1131): long and short training, the bridge (target = the point
    itself), trellis wind-up and data."""
    from oracle import ref, restated as orc
    use_golden_modem_tables()
    x = v17_scenario(bit_rate, seed, noise)
    ev_r, q_r = qam_run(ref.V17Rx(bit_rate), x, (160, 1, 77))
    ev_o, q_o = qam_run(orc.V17(bit_rate), x, (160, 1, 77))
    assert np.array_equal(ev_r, ev_o)
    assert len(q_r) > 1000 and np.any(q_r[:, 2] != 0)
    assert np.array_equal(q_r, q_o)



# ---------------------------------------------------------------------------------
# the modem receivers off their fixed points: carrier offset and sample clock offset (tests/impair.py)
# ---------------------------------------------------------------------------------
# (modem, bit rate, seed, noise dBm0, carrier offset Hz, clock offset ppm, samples of signal).  With the reference's own modulator
# as the only source (every case above) the carrier loop (v29rx.c:297-331 track_carrier, v17rx.c:311-339, v27ter_rx.c:296-325)
# and the symbol timing loop (godard.c:165-220; v27ter_rx.c:486-529 symbol_sync) sit at their fixed points.  Here the carrier
# is 3 .. 7 Hz off and the far end's clock 50 .. 100 ppm off, over three seconds, so that the timing loop
# steps through the pulse shaper's coefficient sets again and again; three cases are too far off and FAIL training.
MODEM_OFFSET_CASES = [("v29", 9600, 61, -45.0, 7.0, 100.0, 24000), ("v29", 7200, 62, -42.0, -7.0, -100.0, 24000),
                      ("v29", 4800, 63, -40.0, 3.0, -50.0, 16000), ("v29", 9600, 64, -45.0, 30.0, 0.0, 6000),
                      ("v27ter", 4800, 65, -50.0, -7.0, 100.0, 24000), ("v27ter", 2400, 66, -50.0, 7.0, -100.0, 24000),
                      ("v27ter", 4800, 67, -50.0, 25.0, 0.0, 10000),
                      ("v17", 14400, 68, -55.0, 7.0, -100.0, 24000), ("v17", 9600, 69, -50.0, -3.0, 50.0, 20000),
                      ("v17", 7200, 70, -48.0, -7.0, 100.0, 20000), ("v17", 14400, 71, -55.0, 25.0, 0.0, 11200)]
MODEM_OFFSET_GOLDEN = [0, 1, 3, 4, 6, 7, 10]                # the cases frozen in tests/golden/modem_offset_*.npz


def modem_offset_name(case):
    name, bit_rate, seed, noise, hz, ppm, n = case
    return "modem_offset_%s_%d_%d" % (name, bit_rate, seed)


def modem_offset_scenario(case):
    """The scenario of the modem's own live test, longer, through tests/impair.py's line."""
    import impair
    name, bit_rate, seed, noise, hz, ppm, n = case
    if name == "v29":
        x = v29_scenario(bit_rate, seed, noise, n_signal=n)
    elif name == "v27ter":
        x = v27ter_scenario(bit_rate, seed, noise, n_signal=n)
    else:
        x = v17_scenario(bit_rate, seed, noise, n_long=n)
    return impair.line(x, hz, ppm)


def modem_offset_receivers(name):
    from oracle import ref, restated as orc
    return {"v29": (ref.V29Rx, orc.V29), "v27ter": (ref.V27terRx, orc.V27ter), "v17": (ref.V17Rx, orc.V17)}[name]


def modem_offset_expectations(case, ev):
    """What each case is there to show: the receiver trained and carried data with its loops pulled off centre, or it
    gave up (SIG_STATUS_TRAINING_FAILED = -5) and went back to waiting."""
    name, bit_rate, seed, noise, hz, ppm, n = case
    if abs(hz) > 20.0:
        assert -5 in ev and -4 not in ev and len(ev) < 20
    else:
        assert -4 in ev and -5 not in ev and -1 in ev
        assert len(ev) > 0.45*n*bit_rate/8000


@needs_ref
@pytest.mark.parametrize("case", MODEM_OFFSET_CASES, ids=modem_offset_name)
@pytest.mark.parametrize("chunks", [(160,), (1, 7, 333, 64)])
def test_modem_offsets_live(built, case, chunks):
    use_golden_modem_tables()
    x = modem_offset_scenario(case)
    make_ref, make_orc = modem_offset_receivers(case[0])
    ev_r, f_r, w_r = v29_run(make_ref(case[1]), x, chunks)
    ev_o, f_o, w_o = v29_run(make_orc(case[1]), x, chunks)
    modem_offset_expectations(case, ev_r)
    assert np.array_equal(ev_r, ev_o)
    assert np.array_equal(w_r, w_o)
    assert np.array_equal(f_r, f_o)


@needs_ref
def test_modem_offsets_move_the_loops(built):
    """The point of the cases: with the line's offsets the carrier loop's phase rate ends away from nominal by about the
    offset, and the timing loop has stepped many times (total_baud_timing_correction); without them neither has."""
    from oracle import ref
    use_golden_modem_tables()
    case = MODEM_OFFSET_CASES[0]
    name, bit_rate, seed, noise, hz, ppm, n = case
    rx = ref.V29Rx(bit_rate)
    v29_run(rx, modem_offset_scenario(case)[:n], (160,))           # stop while the carrier is up
    off = rx.carrier_frequency() - 1700.0
    assert abs(off - hz) < 1.0, off
    rx0 = ref.V29Rx(bit_rate)
    v29_run(rx0, v29_scenario(bit_rate, seed, noise, n_signal=n)[:n], (160,))
    assert abs(rx0.carrier_frequency() - 1700.0) < 0.5
    # v29_rx_symbol_timing_correction() is in bauds; a baud is 160 steps of the 48-sets-a-sample pulse shaper (v29rx.c:151-154).
    # 100 ppm over 24 000 samples is 2.4 samples = 115 steps beyond what acquisition takes on the unimpaired line
    moved = (rx.symbol_timing_correction() - rx0.symbol_timing_correction())*160.0
    assert -125.0 < moved < -105.0, moved                    # more than twice round the 48 coefficient sets


@needs_ref
@pytest.mark.parametrize("cutoff", [None, -45.5, -38.0])
def test_v29_signal_cutoff_live(built, cutoff):
    """v29_rx_set_signal_cutoff() (v29rx.c:163-169): a line at -32 dBm0 is below the -28.5 dBm0 v29_rx_init() leaves the carrier
    detector at and is seen once the cutoff is lowered (fax_modems.c:416 sets -45.5 dBm0)."""
    from oracle import ref, restated as orc
    use_golden_modem_tables()
    sig = ref.v29_tx(9600, 8000, seed=71, level_dbm0=-32.0)
    x = np.concatenate([np.zeros(230, np.int16), sig, np.zeros(700, np.int16)])
    x = ref.saturated_add(x, ref.awgn(71*7919, -60.0, len(x)))
    r, o = ref.V29Rx(9600), orc.V29(9600)
    if cutoff is not None:
        r.set_signal_cutoff(cutoff)
        o.set_signal_cutoff(cutoff)
    ev_r, f_r, w_r = v29_run(r, x, (160, 1, 77))
    ev_o, f_o, w_o = v29_run(o, x, (160, 1, 77))
    assert (len(ev_r) > 5000 and -4 in ev_r) if cutoff is not None else (len(ev_r) == 0)
    assert np.array_equal(ev_r, ev_o)
    assert np.array_equal(w_r, w_o)
    assert np.array_equal(f_r, f_o)


@pytest.mark.parametrize("case", [MODEM_OFFSET_CASES[i] for i in MODEM_OFFSET_GOLDEN], ids=modem_offset_name)
def test_golden_modem_offsets(built, case):
    use_golden_modem_tables()
    g = np.load(os.path.join(GOLDEN, modem_offset_name(case) + ".npz"))
    assert (float(g["carrier_hz"]), float(g["ppm"])) == (case[4], case[5])
    _, make_orc = modem_offset_receivers(case[0])
    ev, f, w = v29_run(make_orc(case[1]), g["amp"], (160,))
    modem_offset_expectations(case, ev)
    assert np.array_equal(ev, g["events"].astype(np.int32))
    assert np.array_equal(f, g["fwords"])
    assert np.array_equal(w, g["iwords"])


# ---------------------------------------------------------------------------------
# round 6's primitives: periodogram*, the plain dot products and LMS updates, fixed_sqrt32, dds_complexf, arctan2 (tests/prims2.py)
# ---------------------------------------------------------------------------------
def prims2_reference():
    import ctypes as C
    import prims2
    from oracle import ref
    L = ref.lib()
    for name, args in (("glue_fixed_sqrt32_batch", 3), ("glue_arctan2_batch", 4), ("glue_dds_complexf_batch", 5)):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [C.c_void_p]*(args - 1) + [C.c_int] if args < 5 else [C.c_void_p]*3 + [C.c_int, C.c_int]
    return prims2.ByName(L)


@needs_ref
def test_prims2_live(built):
    """The oracle's restatement (oracle/prims_oracle.c, modem_common.h) against the reference's own functions: periodograms
    of four lengths on nasty inputs, coefficient sets and phase offsets (libm on both sides), the plain vector primitives, and
    fixed_sqrt32 / dds_complexf / arctan2 over their whole domains -- every answer bit for bit."""
    import prims2
    use_golden_modem_tables()
    d = prims2.inputs()
    a = prims2.run(prims2_reference(), d)
    b = prims2.run(prims2.Restated(), d)
    assert set(a) == set(b) and len(a) > 40
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert len(np.unique(a["sqrt"])) > 1500 and len(np.unique(a["atan"])) > 500000


def test_golden_prims2(built):
    import prims2
    use_golden_modem_tables()
    g = np.load(os.path.join(GOLDEN, "prims2.npz"))
    out = prims2.summary(prims2.run(prims2.Restated()))
    assert set(out) == set(g.files)
    for k in out:
        assert np.array_equal(out[k], g[k]), k


# ---------------------------------------------------------------------------------
# frozen pins: golden vectors generated from the reference build
# ---------------------------------------------------------------------------------
def test_golden_files_present():
    assert len(glob.glob(os.path.join(GOLDEN, "*.npz"))) >= 4


def test_golden_goertzel_constants(built):
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, "goertzel_fac.npz"))
    for f, want in zip(g["freq"], g["fac_bits"]):
        assert bits([orc.goertzel_fac(float(f))])[0] == want, f


@pytest.mark.parametrize("name", ["dtmf_mode0", "dtmf_mode1", "dtmf_mode2", "dtmf_filter"])
def test_golden_dtmf(built, name):
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    mode = int(g["mode"])
    o = orc.Dtmf(mode)
    if int(g["filter"]):
        o.parms(1, float(g["twist"]), float(g["reverse_twist"]), float(g["threshold"]))
    x = g["amp"]
    chunk = int(g["chunk"])
    snaps = []
    for k in range(0, len(x), chunk):
        o.rx(x[k:k + chunk])
        s = o.snapshot()
        snaps.append(np.concatenate([bits(s["v2"]), bits(s["v3"]), bits([s["energy"]]),
                                     np.array([s["current_sample"], s["duration"], s["last_hit"], s["in_digit"]], np.uint32)]))
    assert np.array_equal(np.stack(snaps), g["snapshots"])
    assert o.sink.events().tobytes() == g["events"].tobytes()
    assert o.sink.text() == str(g["text"])
    assert o.get() == str(g["digits"])


@pytest.mark.parametrize("name", ["bell_mf", "r2_mf_fwd", "r2_mf_back"])
def test_golden_mf(built, name):
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    x = g["amp"]
    o = orc.BellMf(1) if name == "bell_mf" else orc.R2Mf(name.endswith("fwd"), True)
    snaps = []
    for k in range(0, len(x), 160):
        o.rx(x[k:k + 160])
        s = o.snapshot()
        snaps.append(np.concatenate([bits(s["v2"]), bits(s["v3"]), np.array([s["current_sample"]], np.uint32)]))
    assert np.array_equal(np.stack(snaps), g["snapshots"])
    assert o.sink.events().tobytes() == g["events"].tobytes()
    if name == "bell_mf":
        assert o.sink.text() == str(g["text"])


def test_golden_super_tone(built):
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, "super_tone.npz"))
    d = build_st_desc(orc.SuperToneDesc)
    assert np.array_equal(bits(d.fac), g["fac_bits"])
    o = orc.SuperTone(d, True)
    x = g["amp"]
    for k in range(0, len(x), 160):
        o.rx(x[k:k + 160])
    assert o.sink.events().tobytes() == g["events"].tobytes()


@pytest.mark.parametrize("taps,mode", ECHO_CASES)
def test_golden_echo(built, taps, mode):
    import zlib
    from oracle import restated as orc
    g = np.load(os.path.join(GOLDEN, "echo_%d_%02x.npz" % (taps, mode)))
    tx, rx = echo_scenario(taps, seed=taps + mode)
    assert zlib.crc32(tx.tobytes()) == int(g["tx_crc"]) and zlib.crc32(rx.tobytes()) == int(g["rx_crc"]), \
        "scenario generator drifted from the one the fixture was made with"
    o = orc.EchoCan(taps, mode)
    clean = np.concatenate([o.run(tx[k:k + 160], rx[k:k + 160], True) for k in range(0, len(tx), 160)])
    assert np.array_equal(clean, g["clean"])
    s = o.snapshot()
    assert np.array_equal(s["taps32"], g["taps32"]) and np.array_equal(s["taps16"], g["taps16"])
    assert np.array_equal(s["history"], g["history"])
    assert [s[k] for k in g["fields"]] == list(g["values"])


@pytest.mark.parametrize("bit_rate", [9600, 7200, 4800])
def test_golden_v29(built, bit_rate):
    from oracle import restated as orc
    use_golden_modem_tables()
    g = np.load(os.path.join(GOLDEN, "v29_%d.npz" % bit_rate))
    ev, f, w = v29_run(orc.V29(bit_rate), g["amp"], (160,))
    assert np.array_equal(ev, g["events"].astype(np.int32))
    assert np.array_equal(f, g["fwords"])
    assert np.array_equal(w, g["iwords"])


@pytest.mark.parametrize("bit_rate", [4800, 2400])
def test_golden_v27ter(built, bit_rate):
    from oracle import restated as orc
    use_golden_modem_tables()
    g = np.load(os.path.join(GOLDEN, "v27ter_%d.npz" % bit_rate))
    ev, f, w = v29_run(orc.V27ter(bit_rate), g["amp"], (160,))
    assert np.array_equal(ev, g["events"].astype(np.int32))
    assert np.array_equal(f, g["fwords"])
    assert np.array_equal(w, g["iwords"])


@pytest.mark.parametrize("bit_rate", [14400, 12000, 9600, 7200, 4800])
def test_golden_v17(built, bit_rate):
    from oracle import restated as orc
    use_golden_modem_tables()
    g = np.load(os.path.join(GOLDEN, "v17_%d.npz" % bit_rate))
    ev, f, w = v29_run(orc.V17(bit_rate), g["amp"], (160,))
    assert np.array_equal(ev, g["events"].astype(np.int32))
    assert np.array_equal(f, g["fwords"])
    assert np.array_equal(w, g["iwords"])


QAM_GOLDEN_CASES = [("v29", 9600, 11, -45.0), ("v27ter", 4800, 31, -50.0), ("v17", 14400, 51, -50.0)]


def qam_golden_run(make, name, bit_rate, seed, noise):
    x = {"v29": v29_scenario, "v27ter": v27ter_scenario, "v17": v17_scenario}[name](bit_rate, seed, noise)
    return qam_run(make(bit_rate), x, (160,))[1]


def test_golden_modem_qam(built):
    """The reference's qam_report streams (committed) against the oracle, where the reference itself is not present."""
    from oracle import restated as orc
    use_golden_modem_tables()
    g = np.load(os.path.join(GOLDEN, "modem_qam.npz"))
    for name, bit_rate, seed, noise in QAM_GOLDEN_CASES:
        q = qam_golden_run({"v29": orc.V29, "v27ter": orc.V27ter, "v17": orc.V17}[name], name, bit_rate, seed, noise)
        assert np.array_equal(q, g["%s_%d" % (name, bit_rate)]), (name, bit_rate)


# ---------------------------------------------------------------------------------
# G.168 line models and BASELINE.md section 2's known answers
# ---------------------------------------------------------------------------------
@needs_ref
def test_g168_line_model_live(built):
    """tests/g168.py's restatement of the reference test program's line simulator (tests/echo_tests.c:396-487) against the
    reference's own fir32() over the reference's own tables, every model, near end talk included."""
    from oracle import ref
    import g168
    tx = ref.awgn(99, -15.0, 12000)
    near = ref.awgn(100, -20.0, 12000)
    for m in range(2, 10):
        taps, ki = ref.g168_model(m)
        assert np.array_equal(taps, g168.models()[0][m][0]) and np.float32(ki) == g168.models()[0][m][1]
        for erl in (-6.0, -12.0, -24.0, -10.7):
            assert g168.gain(m, erl) == ref.g168_gain(m, erl), (m, erl)
            assert np.array_equal(g168.line(m, erl, tx, near), ref.g168_line(m, erl, tx, near)), (m, erl)


def g168_known_signals():
    """tx and rx of BASELINE.md section 2's echo canceller run, from the restatements (what the GPU box can make)."""
    from oracle import restated as orc
    import g168
    K = g168.KNOWN_D2
    tx = orc.Awgn(K["seed"], K["level_dbm0"]).gen(K["samples"])
    return tx, g168.line(K["model"], K["erl_db"], tx)


@needs_ref
def test_g168_d2_known_answer_live(built):
    """echo_can_update(), 128 taps, white noise at -15 dBm0 through model D2 at 12 dB ERL reaches
    54.6 dB ERLE over the last second of 20 s -- the real reference, and the restatement sample for sample."""
    from oracle import ref, restated as orc
    import g168
    K = g168.KNOWN_D2
    tx, rx = g168_known_signals()
    assert np.array_equal(tx, ref.awgn(K["seed"], K["level_dbm0"], K["samples"]))
    r = ref.EchoCan(K["taps"], K["mode"])
    clean = r.run(tx, rx, False)
    assert abs(g168.erle_db(rx[-8000:], clean[-8000:]) - K["erle_db"]) < 0.05
    o = orc.EchoCan(K["taps"], K["mode"])
    assert np.array_equal(o.run(tx, rx, False), clean)


def test_golden_g168_d2_known_answer(built):
    import zlib
    from oracle import restated as orc
    import g168
    K = g168.KNOWN_D2
    g = np.load(os.path.join(GOLDEN, "g168_d2_known.npz"))
    tx, rx = g168_known_signals()
    assert zlib.crc32(tx.tobytes()) == int(g["tx_crc"]) and zlib.crc32(rx.tobytes()) == int(g["rx_crc"])
    clean = orc.EchoCan(K["taps"], K["mode"]).run(tx, rx, False)
    assert zlib.crc32(clean.tobytes()) == int(g["clean_crc"]) and np.array_equal(clean[-8000:], g["clean_last_second"])
    assert abs(g168.erle_db(rx[-8000:], clean[-8000:]) - K["erle_db"]) < 0.05 and abs(float(g["erle_db"]) - K["erle_db"]) < 0.05


V29_KNOWN = {"bit_rate": 9600, "samples": 320000, "tx_seed": 1, "noise_seed": 1234567, "noise_dbm0": -50.0, "bits": 381500,
             "status": [-2, -3, -4]}


def v29_known_signal(built=None):
    """v29_tx() (9600 bps, PRBS, default level) + AWGN at -50 dBm0, 40 s, from the restatements."""
    from oracle import restated as orc
    K = V29_KNOWN
    use_v29_tx_table(built)
    x = orc.V29Tx(K["bit_rate"], False, K["tx_seed"]).tx(K["samples"])
    return (x.astype(np.int32) + orc.Awgn(K["noise_seed"], K["noise_dbm0"]).gen(len(x)).astype(np.int32)).clip(-32768, 32767).astype(np.int16)


def prbs15_bits(seed, n):
    """The data source the harness gives the transmitters (oracle/ref_glue: glue_fn_prbs_get_bit): x^15 + x^14 + 1."""
    out = np.zeros(n, np.int8)
    s = seed & 0x7FFF
    for i in range(n):
        bit = ((s >> 14) ^ (s >> 13)) & 1
        s = ((s << 1) | bit) & 0x7FFF
        out[i] = bit
    return out


@needs_ref
def test_v29_known_answer_live(built):
    """v29_tx -> AWGN -50 dBm0 -> v29_rx delivers 381 500 bits, the statuses CARRIER_UP,
    TRAINING_IN_PROGRESS, TRAINING_SUCCEEDED and nothing else -- the real reference on the restated signal, and the
    restated receiver event for event."""
    from oracle import ref, restated as orc
    K = V29_KNOWN
    y = v29_known_signal(built)
    assert np.array_equal(y, ref.saturated_add(ref.v29_tx(K["bit_rate"], K["samples"], seed=K["tx_seed"]), ref.awgn(K["noise_seed"], K["noise_dbm0"], K["samples"])))
    r = ref.V29Rx(K["bit_rate"])
    o = orc.V29(K["bit_rate"])
    for k in range(0, len(y), 160):
        r.rx(y[k:k + 160])
        o.rx(y[k:k + 160])
    a = r.sink.events()["a"]
    assert int((a >= 0).sum()) == K["bits"] and [int(v) for v in a[a < 0]] == K["status"]
    assert np.array_equal(o.sink.events()["a"], a)
