"""GPU parity of the batched G.168 echo canceller (spangpu_echo_*) against the oracle
(oracle/echo_oracle.c, pinned to the reference by test_oracle_pin.py): every clean sample
and the complete per-channel state (control words, 32-bit taps, all four 16-bit tap sets,
FIR history in the reference's physical order) bit-exact, frame after frame."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_channels(n_ch, n, taps, seed):
    """tx = noise (some channels get a tone burst), rx = echo through a random sparse path
    + near-end noise + double-talk bursts."""
    rng = np.random.default_rng(seed)
    tx = np.zeros((n_ch, n), np.int16)
    rx = np.zeros((n_ch, n), np.int16)
    t = np.arange(n)
    for c in range(n_ch):
        x = rng.normal(0, rng.uniform(800, 4000), n)
        if c % 3 == 1:
            a, b = sorted(rng.integers(n//4, n, 2))
            x[a:b] = 3000*np.sin(2*np.pi*rng.uniform(400, 2500)*t[a:b]/8000.0)       # narrow-band stretch
        if c % 5 == 4:
            x[:] = 0.3*x + 1500                                                     # DC offset for the HPFs
        h = np.zeros(taps)
        for k in rng.integers(0, taps, 5):
            h[k] = rng.uniform(-0.4, 0.4)
        y = np.convolve(x, h)[:n] + rng.normal(0, 15, n)
        if c % 2 == 0:
            a = int(rng.integers(n//5, n - 3000))
            y[a:a + 2500] += rng.normal(0, 7000, 2500)                               # double talk
        if c % 7 == 6:
            y *= 6.0                                                                 # gain > 1: drives the divergence zap
        tx[c] = np.clip(x, -32768, 32767).astype(np.int16)
        rx[c] = np.clip(y, -32768, 32767).astype(np.int16)
    return tx, rx


def frames_of(total, sizes):
    pos = 0
    i = 0
    while pos < total:
        n = min(sizes[i % len(sizes)], total - pos)
        yield pos, n
        pos += n
        i += 1


def compare_state(bank, dets, what):
    from spandsp_amd import engine
    for c, d in enumerate(dets):
        g = bank.get_state(c)
        o = d.snapshot()
        for key in engine.ECHO_FIELDS:
            assert g[key] == o[key], (what, c, key, g[key], o[key])
        assert np.array_equal(g["last_acf"], o["last_acf"]), (what, c, "last_acf")
        assert np.array_equal(g["taps32"], o["taps32"]), (what, c, "taps32")
        assert np.array_equal(g["taps16"], o["taps16"]), (what, c, "taps16", np.nonzero(g["taps16"] != o["taps16"]))
        assert np.array_equal(g["history"], o["history"]), (what, c, "history")


@pytest.mark.parametrize("taps,mode,sizes,lanes", [
    (128, 0x01, [160], 0),
    (128, 0x01 | 0x02 | 0x04 | 0x20 | 0x40, [160], 0),
    (128, 0x01 | 0x02, [1, 7, 160, 333, 64, 5], 0),
    (256, 0x01 | 0x20 | 0x40, [160], 0),
    (64, 0x01 | 0x02 | 0x04, [80, 240], 0),
    (32, 0x01, [160], 0),
    # 64 and 128 ms tails (echo_can_init(len): any power of two, echo.c:254-300): sixteen lanes per channel, slices of 32 / 64 taps;
    # narrowband_detect()'s walk wraps at its hard-coded 256 inside the longer history (echo.c:133-139)
    (512, 0x01 | 0x02, [160, 77], 0),
    (512, 0x01 | 0x02 | 0x04 | 0x20 | 0x40, [160, 31], 0),
    (1024, 0x01 | 0x20 | 0x40, [160], 0),
    # the other lane mappings (spangpu_tune_echo_lanes_per_channel): identical results.  Eight lanes (half a DPP row)
    (128, 0x01 | 0x02 | 0x04 | 0x20 | 0x40, [160, 77], 8),
    (64, 0x01 | 0x02, [160], 8),
    (32, 0x01, [160], 8),
    (128, 0x01, [160, 77], 8),              # the eight-lane kernel compiled for mode 0x01
    # sixteen lanes per channel (a DPP row)
    (128, 0x01, [160, 31], 16),             # the sixteen-lane kernel compiled for mode 0x01
    (128, 0x01 | 0x02 | 0x04 | 0x20 | 0x40, [160, 77], 16),
    (64, 0x01 | 0x02, [160], 16),
    # four lanes per channel (a DPP quad; the default up to 128 taps)
    (128, 0x01 | 0x02 | 0x04 | 0x20 | 0x40, [160, 77], 4),
    (128, 0x01, [160], 4),
    (128, 0x01, [160, 1, 31, 77], 4),       # the kernel compiled for mode 0x01: partial rounds of the common body
    (128, 0x01 | 0x02, [160, 77], 4),       # ... for adaption + NLP
    (128, 0x01 | 0x02 | 0x04, [160, 77], 4),    # ... for adaption + NLP + CNG
    (64, 0x01 | 0x02, [160, 1, 31], 4),
    (32, 0x01 | 0x20 | 0x40, [160], 4),
    # two lanes per channel, 16-bit quantities packed in pairs (echo_pair.hpp; the default for big banks)
    (128, 0x01 | 0x02 | 0x04 | 0x20 | 0x40, [160, 77], 2),
    (128, 0x01, [160, 1, 31], 2),
    (64, 0x01 | 0x02 | 0x04, [160], 2),
    (32, 0x01 | 0x20 | 0x40, [160, 3], 2),
])
def test_echo_bank_parity(built, taps, mode, sizes, lanes):
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 37                               # ragged: 9 full wavefronts + one with a single channel
    n = 160*150
    tx, rx = make_channels(n_ch, n, taps, seed=taps + mode)
    assert engine.lib().spangpu_tune_echo_lanes_per_channel(lanes) == 0
    try:
        bank = engine.EchoBank(n_ch, taps, mode)
    finally:
        engine.lib().spangpu_tune_echo_lanes_per_channel(0)
    dets = [orc.EchoCan(taps, mode) for _ in range(n_ch)]
    events = {"rot": 0, "dtd": 0, "nb": 0}
    for fi, (pos, m) in enumerate(frames_of(n, sizes)):
        got = bank.update_host(tx[:, pos:pos + m], rx[:, pos:pos + m], use_hpf_tx=True)
        for c, d in enumerate(dets):
            want = d.run(tx[c, pos:pos + m], rx[c, pos:pos + m], True)
            assert np.array_equal(got[c], want), (fi, c, np.nonzero(got[c] != want)[0][:5])
        if fi % 10 == 0 or fi < 3:
            compare_state(bank, dets, (taps, hex(mode), fi))
    compare_state(bank, dets, "final")
    # the scenario must really have driven the rare paths somewhere in the bank
    snaps = [d.snapshot() for d in dets]
    assert any(s["tap_set"] != 0 or s["tap_rotate_counter"] != 1600 for s in snaps)
    assert any(np.any(s["taps32"] != 0) for s in snaps)


def test_echo_flush_and_mode_change(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch, taps, mode = 8, 128, 0x01
    tx, rx = make_channels(n_ch, 160*60, taps, seed=99)
    bank = engine.EchoBank(n_ch, taps, mode)
    dets = [orc.EchoCan(taps, mode) for _ in range(n_ch)]
    for fi, pos in enumerate(range(0, tx.shape[1], 160)):
        if fi == 25:
            # after >= 1 rotation: flush leaves fir_state.coeffs on the old set (echo.c:331-372)
            for c in (1, 5):
                bank.flush(c)
                dets[c].flush()
        if fi == 40:
            bank.adaption_mode(0x01 | 0x02 | 0x04)
            for d in dets:
                d.adaption_mode(0x01 | 0x02 | 0x04)
        got = bank.update_host(tx[:, pos:pos + 160], rx[:, pos:pos + 160], use_hpf_tx=False)
        for c, d in enumerate(dets):
            want = d.run(tx[c, pos:pos + 160], rx[c, pos:pos + 160], False)
            assert np.array_equal(got[c], want), (fi, c)
    compare_state(bank, dets, "flush")


def test_echo_kernel_follows_the_modes_of_the_bank(built):
    """A four-lane bank created in mode ECHO_CAN_USE_ADAPTION runs the kernel compiled for that mode; one channel given
    another mode sends the bank to the kernel that reads each channel's mode word, every channel back on 0x01 returns it
    (echo_api.hip, `uniform_mode`).  Same results whichever kernel ran."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch, taps = 37, 128
    tx, rx = make_channels(n_ch, 160*50, taps, seed=31337)
    assert engine.lib().spangpu_tune_echo_lanes_per_channel(4) == 0
    try:
        bank = engine.EchoBank(n_ch, taps, 0x01)
    finally:
        engine.lib().spangpu_tune_echo_lanes_per_channel(0)
    dets = [orc.EchoCan(taps, 0x01) for _ in range(n_ch)]
    for fi, pos in enumerate(range(0, tx.shape[1], 160)):
        if fi == 10:
            bank.adaption_mode(0x01 | 0x02 | 0x04 | 0x40, channel=3)
            dets[3].adaption_mode(0x01 | 0x02 | 0x04 | 0x40)
        if fi == 30:
            bank.adaption_mode(0x01)
            for d in dets:
                d.adaption_mode(0x01)
        if fi == 42:
            bank.adaption_mode(0x01 | 0x02, channel=7)
            dets[7].adaption_mode(0x01 | 0x02)
        if fi == 45:
            # ... and one channel at a time: the channels agree again, which the bank finds out at its next update
            bank.adaption_mode(0x01, channel=7)
            dets[7].adaption_mode(0x01)
        if fi in (20, 38):
            # echo_can_flush() in either kernel: the FIR stays on the old tap set until the next rotation, every sample of
            # the wave takes the complete routine meanwhile
            for c in (1, 5, 36):
                bank.flush(c)
                dets[c].flush()
        got = bank.update_host(tx[:, pos:pos + 160], rx[:, pos:pos + 160], use_hpf_tx=False)
        for c, d in enumerate(dets):
            want = d.run(tx[c, pos:pos + 160], rx[c, pos:pos + 160], False)
            assert np.array_equal(got[c], want), (fi, c)
        if fi in (9, 10, 20, 21, 29, 30, 38, 39):
            compare_state(bank, dets, ("modes", fi))
    compare_state(bank, dets, "modes")


def test_echo_line_statistics(built):
    """The per-channel result of a multi-GPU echo run: energy of rx and of the cleaned signal over the
    last second (exact 64-bit sums; ERLE from them) and the CRC-32 of the whole clean stream, against the oracle."""
    import zlib
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch, taps, mode = 70, 128, 0x01
    n = 160*120
    tx, rx = make_channels(n_ch, n, taps, seed=4242)
    bank = engine.EchoBank(n_ch, taps, mode)
    bank.stats(True)
    dets = [orc.EchoCan(taps, mode) for _ in range(n_ch)]
    clean_o = np.zeros((n_ch, n), np.int16)
    last_second = n - 8000
    for pos in range(0, n, 160):
        if pos == last_second:
            bank.stats_reset(sums=True, crc=False)
        got = bank.update_host(tx[:, pos:pos + 160], rx[:, pos:pos + 160], use_hpf_tx=False)
        for c, d in enumerate(dets):
            clean_o[c, pos:pos + 160] = d.run(tx[c, pos:pos + 160], rx[c, pos:pos + 160], False)
        assert np.array_equal(got, clean_o[:, pos:pos + 160])
    st = bank.stats_get()
    r = rx[:, last_second:].astype(np.int64)
    c = clean_o[:, last_second:].astype(np.int64)
    assert np.array_equal(st["sum_rx2"], (r*r).sum(axis=1).astype(np.uint64))
    assert np.array_equal(st["sum_clean2"], (c*c).sum(axis=1).astype(np.uint64))
    assert np.all(st["samples"] == 8000)
    want_crc = np.array([zlib.crc32(clean_o[k].astype("<i2").tobytes()) for k in range(n_ch)], np.uint32)
    assert np.array_equal(st["crc"], want_crc)
    erle = bank.erle_host()
    want = 10.0*np.log10((r*r).sum(axis=1)/np.maximum((c*c).sum(axis=1), 1))
    assert np.allclose(erle, want, rtol=1e-5, atol=1e-4)
    assert np.max(erle) > 20.0 and np.sum(erle > 10.0) >= n_ch//4       # the single-talk lines did converge
    # a sub-range, and a second reset
    part = bank.stats_get(5, 9)
    assert np.array_equal(part["crc"], want_crc[5:14])
    bank.stats_reset(sums=True, crc=True)
    z = bank.stats_get()
    assert not z["sum_rx2"].any() and not z["crc"].any() and not z["samples"].any()


@pytest.mark.parametrize("lanes", [0, 2, 4, 8, 16])
def test_echo_energy_sums_by_the_update_kernel(built, lanes):
    """spangpu_echo_stats(ec, 2): the update kernel itself adds up the frame's energy sums (no second pass, no CRC), under
    every lane mapping, and they are the sums of the samples that went in and came out."""
    from spandsp_amd import engine
    n_ch, taps, mode = 200, 128, 0x01
    n = 160*30
    tx, rx = make_channels(n_ch, n, taps, seed=777)
    engine.lib().spangpu_tune_echo_lanes_per_channel(lanes)
    try:
        bank = engine.EchoBank(n_ch, taps, mode)
    finally:
        engine.lib().spangpu_tune_echo_lanes_per_channel(0)
    bank.stats(2)
    clean = np.zeros((n_ch, n), np.int16)
    for pos in range(0, n, 160):
        clean[:, pos:pos + 160] = bank.update_host(tx[:, pos:pos + 160], rx[:, pos:pos + 160], use_hpf_tx=False)
    st = bank.stats_get()
    r = rx.astype(np.int64)
    c = clean.astype(np.int64)
    assert np.array_equal(st["sum_rx2"], (r*r).sum(axis=1).astype(np.uint64))
    assert np.array_equal(st["sum_clean2"], (c*c).sum(axis=1).astype(np.uint64))
    assert np.all(st["samples"] == n) and not st["crc"].any()
    bank.close()


def test_g168_lines_and_the_known_answer(built):
    """The contract's echo workload and BASELINE.md section 2's known answer on the GPU: 64 lines through
    the eight G.168 echo path models (tests/g168.py: the reference test program's line simulator, models by c mod 8, ERL
    6 ... 24 dB), white noise at -15 dBm0 from the reference's awgn(), 20 s without a break.  Line 0 is the known answer's
    own set-up: model D2, ERL 12 dB, seed 1234567 -- 54.6 dB ERLE over the last second.  Every clean sample of every line
    against the oracle, the ERLE of every line from the bank's own statistics."""
    import zlib
    from oracle import restated as orc
    from spandsp_amd import engine
    import g168
    K = g168.KNOWN_D2
    n_ch, n = 64, K["samples"]
    tx = np.zeros((n_ch, n), np.int16)
    rx = np.zeros((n_ch, n), np.int16)
    for c in range(n_ch):
        model = 2 + c % 8
        erl = K["erl_db"] if c == 0 else -(6.0 + 18.0*((c*37) % 64)/63.0)
        tx[c] = orc.Awgn(K["seed"] + 1000*c, K["level_dbm0"]).gen(n)
        near = None
        if c % 10 == 9:                                   # one line in ten has the near end talking now and then
            near = np.zeros(n, np.int16)
            for k in range(3):
                a = 20000 + 45000*k
                near[a:a + 6000] = orc.Awgn(77 + c + k, -18.0).gen(6000)
        rx[c] = g168.line(model, erl, tx[c], near)
    g = np.load(os.path.join(g168.GOLDEN, "g168_d2_known.npz"))
    assert zlib.crc32(tx[0].tobytes()) == int(g["tx_crc"]) and zlib.crc32(rx[0].tobytes()) == int(g["rx_crc"])
    bank = engine.EchoBank(n_ch, K["taps"], K["mode"])
    bank.stats(True)
    dets = [orc.EchoCan(K["taps"], K["mode"]) for _ in range(n_ch)]
    want = np.stack([d.run(tx[c], rx[c], False) for c, d in enumerate(dets)])
    got = np.zeros_like(want)
    step = 1600                                             # ten frames a call: the reference's callers may hand over any length
    for pos in range(0, n, step):
        if pos == n - 8000:
            bank.stats_reset(sums=True, crc=False)
        got[:, pos:pos + step] = bank.update_host(tx[:, pos:pos + step], rx[:, pos:pos + step], use_hpf_tx=False)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, bad[:8]
    assert zlib.crc32(got[0].tobytes()) == int(g["clean_crc"])
    erle = bank.erle_host()
    assert abs(float(erle[0]) - K["erle_db"]) < 0.05, erle[0]
    for c in range(n_ch):
        assert abs(float(erle[c]) - g168.erle_db(rx[c, -8000:], want[c, -8000:])) < 1e-3, c
    single = np.array([erle[c] for c in range(n_ch) if c % 10 != 9])
    assert np.median(single) > 40.0, np.median(single)
    bank.close()


def test_echo_golden_direct(built):
    """The committed outputs of the real reference (tests/golden/echo_*.npz: every clean sample, the final taps, history and
    control words of echo_can_update() at 64 .. 1024 taps), straight against the GPU -- no oracle in the loop."""
    import zlib
    from spandsp_amd import engine
    from test_oracle_pin import ECHO_CASES, GOLDEN, echo_scenario
    for taps, mode in ECHO_CASES:
        g = np.load(os.path.join(GOLDEN, "echo_%d_%02x.npz" % (taps, mode)))
        tx, rx = echo_scenario(taps, seed=taps + mode)
        assert zlib.crc32(tx.tobytes()) == int(g["tx_crc"]) and zlib.crc32(rx.tobytes()) == int(g["rx_crc"])
        n_ch = 5
        bank = engine.EchoBank(n_ch, taps, mode)
        clean = []
        for k in range(0, len(tx), 160):
            got = bank.update_host(np.tile(tx[k:k + 160], (n_ch, 1)), np.tile(rx[k:k + 160], (n_ch, 1)), use_hpf_tx=True)
            clean.append(got)
        clean = np.concatenate(clean, axis=1)
        for c in range(n_ch):
            assert np.array_equal(clean[c], g["clean"]), (taps, hex(mode), c, np.nonzero(clean[c] != g["clean"])[0][:5])
        s = bank.get_state(n_ch - 1)
        assert np.array_equal(s["taps32"], g["taps32"]) and np.array_equal(s["taps16"], g["taps16"]), (taps, hex(mode))
        assert np.array_equal(s["history"], g["history"]), (taps, hex(mode))
        for key, want in zip(g["fields"], g["values"]):
            assert s[str(key)] == int(want), (taps, hex(mode), key, s[str(key)], int(want))
        bank.close()
