"""GPU parity: the HIP Goertzel-bank kernels (through the C ABI) against the CPU
oracle (oracle/tone_oracle.c, itself pinned to the reference by test_oracle_pin.py)
on the same seeded inputs.  Bar: every float state word, every Goertzel energy and
every decision BIT-EXACT (the tolerance north_star allows for energies, 1e-5
relative, is not needed: the kernels evaluate the same unfused fp32 expression tree).
"""
import ctypes as C
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

_libm = C.CDLL("libm.so.6")
_libm.log10f.restype = C.c_float
_libm.log10f.argtypes = [C.c_float]


# (lanes per channel, kernel family): the general kernel (tone_dev.hpp) under both lane mappings, the streaming kernels
# (tone_fast.hpp) with a loader wave per workgroup and without, the latter also with two lanes per channel.
_KERNELS = [(1, 1), (2, 1), (1, 2), (1, 3), (2, 3)]
_FAMILY = {1: "general", 2: "loader", 3: "stream"}


@pytest.fixture(params=_KERNELS, ids=lambda p: "lpc%d-%s" % (p[0], _FAMILY[p[1]]), autouse=True)
def lanes_per_channel(request, built):
    """Every parity test runs under every kernel family and lane mapping the library can pick."""
    from spandsp_amd import engine
    lpc, variant = request.param
    engine.tune_lanes_per_channel(lpc)
    engine.tune_tone_kernel(variant)
    yield lpc
    engine.tune_lanes_per_channel(0)
    engine.tune_tone_kernel(0)


FMA_MODE = os.environ.get("SPANGPU_TEST_FMA") == "1"


def same_f32(a, b):
    """Bit for bit -- unless the library under test is the v_pk_fma_f32 BUILD VARIANT of round 6 (SPANGPU_TEST_FMA=1 with SPANGPU_LIB
    pointing at it: tools/gpu_fma.sh): then within 1e-5 of the largest magnitude of the vector compared (north_star's tolerance for
    the Goertzel energies; decisions, digits and every integer stay exact in that mode too)."""
    a = np.atleast_1d(np.asarray(a, np.float32))
    b = np.atleast_1d(np.asarray(b, np.float32))
    if not FMA_MODE:
        return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    if a.shape != b.shape:
        return False
    scale = max(float(np.max(np.abs(b.astype(np.float64)), initial=0.0)), 1e-30)
    return bool(np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= 1e-5*scale))


def f32_bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def frames_of(total, sizes):
    pos = 0
    i = 0
    while pos < total:
        n = min(sizes[i % len(sizes)], total - pos)
        yield pos, n
        pos += n
        i += 1


def run_gpu(bank, sig, sizes, layout=0, hook=None):
    """Returns per-channel list of (hit, code, flags, duration, energy, e[nb+1]) in block order."""
    n_ch = sig.shape[0]
    per_ch = [[] for _ in range(n_ch)]
    for fi, (pos, n) in enumerate(frames_of(sig.shape[1], sizes)):
        if hook:
            hook(fi, bank)
        fr = sig[:, pos:pos + n]
        if layout == 1:
            bank.rx_host(np.ascontiguousarray(fr.T), layout=1)
        else:
            bank.rx_host(fr)
        blk = bank.blocks()
        tr = bank.trace(max_blocks=n//64 + 4) if blk.size else None
        for r in blk:
            e = tr[r["block"], :, r["channel"]].copy()
            per_ch[r["channel"]].append((int(r["hit"]), int(r["code"]), int(r["flags"]), int(r["duration"]),
                                         np.float32(r["energy"]), e))
    return per_ch


def check_blocks(per_ch_gpu, per_ch_orc, nb, what, total_energy=True):
    for c, (g, o) in enumerate(zip(per_ch_gpu, per_ch_orc)):
        assert len(g) == len(o), (what, c, len(g), len(o))
        for k, (gb, ob) in enumerate(zip(g, o)):
            assert gb[0] == ob["hit"], (what, "hit", c, k, gb[0], ob["hit"])
            assert gb[1] == ob["aux"], (what, "code", c, k, gb[1], ob["aux"])
            assert same_f32(gb[5][:nb], ob["e"][:nb]), (what, "energies", c, k)
            if total_energy:
                assert same_f32(gb[5][-1:], [ob["total_energy"]]), (what, "total", c, k)


# --------------------------------------------------------------------------------------
# DTMF
# --------------------------------------------------------------------------------------
def _dtmf_oracle(sig, sizes, mode=0, parms=None, hook=None):
    from oracle import restated as orc
    dets = []
    for c in range(sig.shape[0]):
        d = orc.Dtmf(mode)
        if parms:
            d.parms(**parms)
        dets.append(d)
    per_ch = [[] for _ in dets]
    for fi, (pos, n) in enumerate(frames_of(sig.shape[1], sizes)):
        if hook:
            hook(fi, dets)
        for c, d in enumerate(dets):
            per_ch[c].extend(list(d.rx(sig[c, pos:pos + n])))
    return dets, per_ch


def _dtmf_state_check(bank, dets, what):
    for c, d in enumerate(dets):
        f, i = bank.get_state(c)
        s = d.snapshot()
        assert same_f32(f[0:8], s["v2"]), (what, "v2", c)
        assert same_f32(f[8:16], s["v3"]), (what, "v3", c)
        assert same_f32(f[16:17], [s["energy"]]), (what, "energy", c)
        assert i[0] == s["current_sample"], (what, "cs", c)
        assert i[1] == s["last_hit"] and i[2] == s["in_digit"], (what, "hits", c, i, s)
        assert i[3] == s["duration"], (what, "duration", c, i[3], s["duration"])


@pytest.mark.parametrize("sizes", [[160], [1, 7, 101, 102, 103, 160, 333, 64]])
def test_dtmf_digits_mode(built, sizes):
    from spandsp_amd import engine
    n_ch = 200                      # ragged last wavefront (200 = 3*64 + 8)
    sig, _ = synth.dtmf_channels(n_ch, 160*75, seed=11)
    bank = engine.ToneBank(engine.DTMF, n_ch, trace=True)
    g = run_gpu(bank, sig, sizes)
    dets, o = _dtmf_oracle(sig, sizes)
    check_blocks(g, o, 8, "dtmf")
    _dtmf_state_check(bank, dets, "dtmf")
    # digits appended = CHANGE blocks with a non-zero code (dtmf.c:318-340)
    n_digits = 0
    for c in range(n_ch):
        digits = "".join(chr(b[1]) for b in g[c] if (b[2] & engine.BLK_CHANGE) and b[1])
        assert digits == dets[c].get(), (c, digits)
        n_digits += len(digits)
    assert n_digits > n_ch          # the workload really contains detectable digits


def test_dtmf_realtime_mode(built):
    from spandsp_amd import engine
    n_ch = 130
    sig, _ = synth.dtmf_channels(n_ch, 160*75, seed=12)
    bank = engine.ToneBank(engine.DTMF, n_ch, report_mode=engine.REPORT_REALTIME, trace=True)
    g = run_gpu(bank, sig, [160])
    dets, o = _dtmf_oracle(sig, [160], mode=2)
    check_blocks(g, o, 8, "dtmf-rt")
    _dtmf_state_check(bank, dets, "dtmf-rt")
    n_events = 0
    for c in range(n_ch):
        ev = []
        for b in g[c]:
            if b[2] & engine.BLK_REPORT:
                # the host shim computes the level exactly as dtmf.c:314 does, with libm's log10f
                level = -99 if (b[2] & engine.BLK_TONE_OFF) else int(np.float32(10.0)*np.float32(_libm.log10f(b[4]))
                                                                     - np.float32(107.255))
                ev.append((b[1], level, b[3]))
        oe = [(int(e["a"]), int(e["b"]), int(e["c"])) for e in dets[c].sink.events() if e["kind"] == 1]
        assert ev == oe, (c, ev[:4], oe[:4])
        n_events += len(ev)
    assert n_events > n_ch


def test_dtmf_dialtone_filter_and_parms(built):
    from spandsp_amd import engine
    n_ch = 96
    sig, _ = synth.dtmf_channels(n_ch, 160*50, seed=13)
    # add a strong 350+440 Hz dial tone under everything
    t = np.arange(sig.shape[1])
    dial = 3000.0*np.sin(2*np.pi*350.0*t/8000.0) + 3000.0*np.sin(2*np.pi*440.0*t/8000.0)
    sig = np.clip(sig.astype(np.float64) + dial, -32768, 32767).astype(np.int16)
    bank = engine.ToneBank(engine.DTMF, n_ch, filter_dialtone=True, twist_db=9.0, reverse_twist_db=5.0,
                           threshold_dbm0=-39.0, trace=True)
    g = run_gpu(bank, sig, [160])
    dets, o = _dtmf_oracle(sig, [160], parms=dict(filter_dialtone=1, twist=9.0, reverse_twist=5.0, threshold=-39.0))
    check_blocks(g, o, 8, "dtmf-filter")
    for c, d in enumerate(dets):
        f, i = bank.get_state(c)
        s = d.snapshot()
        assert same_f32(f[17:19], s["z350"]), c
        assert same_f32(f[19:21], s["z440"]), c


def test_dtmf_divergent_block_phase_and_fillin(built):
    """Channels whose 102-sample block phase differs inside one wavefront (after a
    dtmf_rx_fillin() on some of them) take the per-lane path; results must not change."""
    from spandsp_amd import engine
    n_ch = 128
    sig, _ = synth.dtmf_channels(n_ch, 160*60, seed=14)
    victims = {3: [5, 17, 70], 10: [5, 64, 127], 11: list(range(0, 128, 3))}

    def ghook(fi, bank):
        for c in victims.get(fi, []):
            bank.reset_channel(c, fillin_only=True)

    def ohook(fi, dets):
        for c in victims.get(fi, []):
            dets[c].fillin(160)

    bank = engine.ToneBank(engine.DTMF, n_ch, trace=True)
    g = run_gpu(bank, sig, [160], hook=ghook)
    dets, o = _dtmf_oracle(sig, [160], hook=ohook)
    check_blocks(g, o, 8, "dtmf-divergent")
    _dtmf_state_check(bank, dets, "dtmf-divergent")


def _var_ticks(n_ch, n_ticks, seed, ragged):
    """lens[tick][channel]: most channels bring a whole 160-sample frame, some none, and (ragged) some a short one."""
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n_ticks):
        r = rng.random(n_ch)
        lens = np.where(r < 0.2, 0, 160).astype(np.int32)
        if ragged and t % 3 == 1:
            short = rng.random(n_ch) < 0.15
            lens[short] = rng.integers(1, 160, int(short.sum()))
        if t % 7 == 3:
            lens[64:128] = 0                # a whole wavefront of one-lane-per-channel kernels sits the tick out
        if t % 11 == 5:
            lens[:] = 0
            lens[rng.integers(0, n_ch)] = 160
        out.append(lens)
    return out


@pytest.mark.parametrize("ragged", [False, True], ids=["mask", "ragged"])
def test_dtmf_tick_with_missing_and_short_channels(built, ragged):
    """spangpu_bank_rx_var(): a channel without a frame in a tick is untouched by it (filters, block phase, debounce,
    duration), one with a short frame advances by just that; every channel equals an oracle detector fed its own samples
    only.  Frames of 0 / 160 keep the streaming kernels (active mask), other lengths take the general one."""
    from spandsp_amd import engine
    from oracle import restated as orc
    n_ch = 200
    n_ticks = 70
    sig, _ = synth.dtmf_channels(n_ch, 160*n_ticks, seed=21)
    bank = engine.ToneBank(engine.DTMF, n_ch, trace=True)
    dets = [orc.Dtmf(0) for _ in range(n_ch)]
    pos = np.zeros(n_ch, np.int64)
    rng = np.random.default_rng(99)
    g = [[] for _ in range(n_ch)]
    o = [[] for _ in range(n_ch)]
    for lens in _var_ticks(n_ch, n_ticks, 5, ragged):
        frames = rng.integers(-20000, 20000, (n_ch, 160)).astype(np.int16)     # what a row holds beyond lens[c] is never read
        for c in range(n_ch):
            frames[c, :lens[c]] = sig[c, pos[c]:pos[c] + lens[c]]
        bank.rx_host_var(frames, lens)
        blk = bank.blocks()
        tr = bank.trace(max_blocks=4) if blk.size else None
        for r in blk:
            assert lens[r["channel"]] > 0
            g[r["channel"]].append((int(r["hit"]), int(r["code"]), int(r["flags"]), int(r["duration"]), np.float32(r["energy"]),
                                    tr[r["block"], :, r["channel"]].copy()))
        for c in range(n_ch):
            if lens[c]:
                o[c].extend(list(dets[c].rx(sig[c, pos[c]:pos[c] + lens[c]])))
        pos += lens
    check_blocks(g, o, 8, "dtmf-var")
    _dtmf_state_check(bank, dets, "dtmf-var")
    assert sum(len(x) for x in g) > 50*n_ch
    # wrong lengths are refused before anything runs
    bad = np.full(n_ch, 160, np.int32)
    bad[7] = 161
    with pytest.raises(engine.SpanGpuError):
        bank.rx_host_var(np.zeros((n_ch, 160), np.int16), bad)
    bank.rx_host_var(np.zeros((n_ch, 160), np.int16), np.zeros(n_ch, np.int32))        # an empty tick is no tick
    _dtmf_state_check(bank, dets, "dtmf-var-empty")


def test_dtmf_parameters_per_channel(built):
    """spangpu_bank_set_channel_params() = dtmf_rx_parms() on one detector of the bank (dtmf.c:421-445): dial tone filter,
    twists (0 dB included) and threshold differ between channels of one launch, and change in mid-call."""
    from spandsp_amd import engine
    n_ch = 150
    sig, _ = synth.dtmf_channels(n_ch, 160*50, seed=23)
    t = np.arange(sig.shape[1])
    dial = 3000.0*np.sin(2*np.pi*350.0*t/8000.0) + 3000.0*np.sin(2*np.pi*440.0*t/8000.0)
    sig = np.clip(sig.astype(np.float64) + dial, -32768, 32767).astype(np.int16)
    at_start = {c: dict(filter_dialtone=1) for c in range(0, n_ch, 3)}
    at_start.update({c: dict(filter_dialtone=1, twist=0.0, reverse_twist=0.0, threshold=-30.0) for c in range(1, n_ch, 7)})
    at_start.update({c: dict(twist=12.0, threshold=0.0) for c in range(2, n_ch, 11)})
    later = {20: {c: dict(filter_dialtone=0) for c in range(0, n_ch, 6)}, 31: {5: dict(filter_dialtone=1, twist=3.0), 149: dict(threshold=-50.0)}}
    gp = lambda d: dict(filter_dialtone=d.get("filter_dialtone", -1), twist_db=d.get("twist", -1.0),
                        reverse_twist_db=d.get("reverse_twist", -1.0), threshold_dbm0=d.get("threshold", -99.0))

    def ghook(fi, bank):
        for c, d in (at_start if fi == 0 else later.get(fi, {})).items():
            bank.set_channel_params(c, **gp(d))

    def ohook(fi, dets):
        for c, d in (at_start if fi == 0 else later.get(fi, {})).items():
            dets[c].parms(**d)

    bank = engine.ToneBank(engine.DTMF, n_ch, trace=True)
    g = run_gpu(bank, sig, [160], hook=ghook)
    dets, o = _dtmf_oracle(sig, [160], hook=ohook)
    check_blocks(g, o, 8, "dtmf-chan-parms")
    _dtmf_state_check(bank, dets, "dtmf-chan-parms")
    for c, d in enumerate(dets):
        f, i = bank.get_state(c)
        sn = d.snapshot()
        assert same_f32(f[17:19], sn["z350"]) and same_f32(f[19:21], sn["z440"]), c
    n_diff = sum(1 for c in range(n_ch) if [b[1] for b in g[c]] != [b[1] for b in g[(c + 3) % n_ch]])
    assert n_diff > 0
    with pytest.raises(engine.SpanGpuError):
        engine.ToneBank(engine.BELL_MF, 8).set_channel_params(0, twist_db=3.0)


def test_dtmf_zero_db_parameters_through_set_mask(built):
    """A 0 dB twist or 0 dBm0 threshold is a value, not "unset", once set_mask names the field."""
    from spandsp_amd import engine
    n_ch = 64
    sig, _ = synth.dtmf_channels(n_ch, 160*30, seed=24)
    bank = engine.ToneBank(engine.DTMF, n_ch, twist_db=0.0, reverse_twist_db=6.0, threshold_dbm0=0.0, trace=True,
                           set_mask=engine.TP_TWIST | engine.TP_REVERSE_TWIST)
    g = run_gpu(bank, sig, [160])
    dets, o = _dtmf_oracle(sig, [160], parms=dict(twist=0.0, reverse_twist=6.0))
    check_blocks(g, o, 8, "dtmf-mask")
    assert dets[0].snapshot()["normal_twist"] == 1.0


def test_digit_events_are_the_digits_of_the_records(built):
    """spangpu_bank_digit_events() (a compact list made by a small kernel over the last launch's records) and
    spangpu_bank_set_digits_buffer() (one byte per block and channel written by the detector kernel itself: what a
    multi-GPU run gathers) hold exactly the digits the full records report; a list cut by its capacity still says how many."""
    import ctypes
    from spandsp_amd import engine
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n_ch = 1000
    sig, _ = synth.dtmf_channels(n_ch, 160*40, seed=31)
    bank = engine.ToneBank(engine.DTMF, n_ch)
    cap = n_ch
    dev = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dev), 4*(1 + cap)) == 0
    dev2 = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dev2), 4*(1 + cap)) == 0
    total = 0
    armed = False
    for pos in range(0, sig.shape[1], 160):
        bank.rx_host(sig[:, pos:pos + 160])
        blk = bank.blocks()
        want = sorted((int(r["channel"]), int(r["code"]), int(r["block"])) for r in blk
                      if (r["flags"] & engine.BLK_CHANGE) and r["code"])
        small = 3 if pos == 160*20 else cap
        bank.digit_events_device(dev, small)
        bank.sync()
        out = np.zeros(1 + cap, np.uint32)
        assert hip.hipMemcpy(out.ctypes.data, dev, 4*(1 + small), 2) == 0
        assert int(out[0]) == len(want), (pos, int(out[0]), len(want))
        w = out[1:1 + min(small, len(want))].astype(np.int64)
        got = sorted(zip((w & 0xFFFFF).tolist(), ((w >> 20) & 0xFF).tolist(), ((w >> 28) & 0xF).tolist()))
        if small == cap:
            assert got == want, pos
        else:
            assert set(got) <= set(want) and len(got) == min(small, len(want))
        # the one-byte-per-block report written by the detector kernel itself (this launch wrote into dev2)
        if armed:
            nb_max = 2
            dg = np.zeros((nb_max, n_ch), np.uint8)
            assert hip.hipMemcpy(dg.ctypes.data, dev2, dg.nbytes, 2) == 0
            exp = np.zeros((nb_max, n_ch), np.uint8)
            for c, code, b in want:
                exp[b, c] = code
            assert np.array_equal(dg, exp), pos
        fill = np.full((2, n_ch), 0xEE, np.uint8)                       # stale bytes must be overwritten, zeros included
        assert hip.hipMemcpy(dev2, fill.ctypes.data, fill.nbytes, 1) == 0
        bank.set_digits_buffer(dev2, 2*n_ch)
        armed = True
        total += len(want)
    assert total > n_ch
    bank.set_digits_buffer(None, 0)
    hip.hipFree(dev)
    hip.hipFree(dev2)


def test_dtmf_sample_major_layout(built):
    from spandsp_amd import engine
    n_ch = 70
    sig, _ = synth.dtmf_channels(n_ch, 160*30, seed=15)
    bank = engine.ToneBank(engine.DTMF, n_ch, trace=True)
    g = run_gpu(bank, sig, [160], layout=1)
    dets, o = _dtmf_oracle(sig, [160])
    check_blocks(g, o, 8, "dtmf-sample-major")
    _dtmf_state_check(bank, dets, "dtmf-sample-major")


def test_dtmf_full_size_replica_property(built):
    """BASELINE config 2 size (65 536 channels x 160-sample frames): the oracle is too
    slow for all channels, so (a) channels are 256 distinct signals tiled 256 times and
    every replica must agree bit-for-bit with the first copy, and (b) the first 256
    channels are checked against the oracle."""
    from spandsp_amd import engine
    base, _ = synth.dtmf_channels(256, 160*20, seed=16)
    n_ch = 65536
    sig = np.tile(base, (n_ch//256, 1))
    bank = engine.ToneBank(engine.DTMF, n_ch, trace=False)
    recs = []
    for pos in range(0, sig.shape[1], 160):
        bank.rx_host(sig[:, pos:pos + 160])
        b = bank.blocks()
        recs.append(b)
    dets, o = _dtmf_oracle(base, [160])
    total = 0
    n_rep = n_ch//256
    for fi, b in enumerate(recs):
        # blocks() returns (channel, block) order, so each replica is one contiguous slice
        assert len(b) % n_rep == 0, fi
        r = b.reshape(n_rep, -1)
        for name in ("block", "hit", "code", "flags"):
            assert np.array_equal(r[name], np.broadcast_to(r[name][0], r[name].shape)), (fi, name)
        assert np.array_equal(r["channel"] % 256, np.broadcast_to(r["channel"][0], r["channel"].shape)), fi
        total += len(b)
    # oracle check on the first 256 channels
    per_ch = [[] for _ in range(256)]
    for b in recs:
        for r in b[b["channel"] < 256]:
            per_ch[r["channel"]].append((int(r["hit"]), int(r["code"])))
    for c in range(256):
        assert per_ch[c] == [(int(x["hit"]), int(x["aux"])) for x in o[c]], c
    assert total > 0


# --------------------------------------------------------------------------------------
# Bell MF / R2 MF
# --------------------------------------------------------------------------------------
def test_bell_mf(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 150
    sig, _ = synth.bell_mf_channels(n_ch, 160*100, seed=21)
    bank = engine.ToneBank(engine.BELL_MF, n_ch, trace=True)
    sizes = [160, 160, 37, 240]
    g = run_gpu(bank, sig, sizes)
    dets = [orc.BellMf(0) for _ in range(n_ch)]
    o = [[] for _ in dets]
    for pos, n in frames_of(sig.shape[1], sizes):
        for c, d in enumerate(dets):
            o[c].extend(list(d.rx(sig[c, pos:pos + n])))
    check_blocks(g, o, 6, "bell", total_energy=False)
    n_digits = 0
    for c, d in enumerate(dets):
        digits = "".join(chr(b[1]) for b in g[c] if b[2] & engine.BLK_REPORT)
        assert digits == d.get(), (c, digits)
        n_digits += len(digits)
        f, i = bank.get_state(c)
        s = d.snapshot()
        assert same_f32(f[0:6], s["v2"]) and same_f32(f[6:12], s["v3"])
        hits = [i[1], i[2], i[3] & 0xFF, (i[3] >> 8) & 0xFF, (i[3] >> 16) & 0xFF]
        assert i[0] == s["current_sample"] and hits == list(s["hits"]), (c, i, s["hits"])
    assert n_digits > n_ch//2


@pytest.mark.parametrize("fwd", [True, False])
def test_r2_mf(built, fwd):
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 100
    sig, _ = synth.r2_mf_channels(n_ch, 160*80, seed=22 + int(fwd), fwd=fwd)
    bank = engine.ToneBank(engine.R2_MF, n_ch, r2_fwd=fwd, trace=True)
    g = run_gpu(bank, sig, [160])
    dets = [orc.R2Mf(fwd, True) for _ in range(n_ch)]
    o = [[] for _ in dets]
    for pos, n in frames_of(sig.shape[1], [160]):
        for c, d in enumerate(dets):
            o[c].extend(list(d.rx(sig[c, pos:pos + n])))
    check_blocks(g, o, 6, "r2", total_energy=False)
    n_ev = 0
    for c, d in enumerate(dets):
        ev = [(b[1], -10 if b[1] else -99, 0) for b in g[c] if b[2] & engine.BLK_REPORT]
        oe = [(int(e["a"]), int(e["b"]), int(e["c"])) for e in d.sink.events()]
        assert ev == oe, (c, ev[:4], oe[:4])
        n_ev += len(ev)
    assert n_ev > n_ch


# --------------------------------------------------------------------------------------
# Super tone bank and the generic Goertzel bank
# --------------------------------------------------------------------------------------
def _st_desc(D):
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
    d.add_element(t, 950, 0, 300, 0)
    return d


def test_super_tone_bank(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 90
    sig = synth.call_progress_channels(n_ch, 160*120, seed=31)
    desc = _st_desc(orc.SuperToneDesc)
    fac = desc.fac
    assert len(fac) == 7
    bank = engine.ToneBank(engine.SUPER_TONE, n_ch, bin_fac=list(fac), trace=True)
    g = run_gpu(bank, sig, [160])
    dets = [orc.SuperTone(desc) for _ in range(n_ch)]
    o = [[] for _ in dets]
    for pos, n in frames_of(sig.shape[1], [160]):
        for c, d in enumerate(dets):
            o[c].extend(list(d.rx(sig[c, pos:pos + n])))
    check_blocks(g, o, len(fac), "super-tone")
    assert sum(1 for c in range(n_ch) for b in g[c] if b[0] >= 0) > n_ch


def _st_desc_wide(D):
    """A call-progress plan of 22 monitored frequencies (more than one lane's 16 bins), with the naming quirks of
    the reference's resolver in it: a frequency within 10 Hz of an earlier one (shares and re-tunes its bin), and
    the same near frequency named twice (the second time it gets the earlier NAME's position, not the bin)."""
    d = D()
    base = [350, 440, 480, 620, 950, 1100, 1400, 1800, 400, 425, 450, 500, 540, 660, 700, 770, 852, 941, 1004, 1209, 1336, 1477]
    for k in range(0, len(base), 2):
        t = d.add_tone()
        d.add_element(t, base[k], base[k + 1], 300, 0)
        d.add_element(t, 0, 0, 200, 0)
    t = d.add_tone()
    d.add_element(t, 355, 0, 400, 0)        # within 10 Hz of 350: merged into its bin
    d.add_element(t, 355, 445, 400, 0)      # 355 named again; 445 merges with 440
    t = d.add_tone()
    d.add_element(t, 1100, 0, 400, 600)
    d.add_element(t, 0, 0, 2800, 3200)
    return d


def test_super_tone_bank_of_more_than_16_bins(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 70
    sig = synth.call_progress_channels(n_ch, 160*100, seed=33)
    desc = _st_desc_wide(orc.SuperToneDesc)
    fac = list(desc.fac)
    assert 16 < len(fac) <= engine.MAX_BINS
    bank = engine.ToneBank(engine.SUPER_TONE, n_ch, bin_fac=fac, trace=True)
    g = run_gpu(bank, sig, [160, 96, 256])
    dets = [orc.SuperTone(desc) for _ in range(n_ch)]
    o = [[] for _ in dets]
    for pos, n in frames_of(sig.shape[1], [160, 96, 256]):
        for c, d in enumerate(dets):
            o[c].extend(list(d.rx(sig[c, pos:pos + n])))
    check_blocks(g, o, len(fac), "super-tone, %d bins" % len(fac))
    assert sum(1 for c in range(n_ch) for b in g[c] if b[0] >= 0) > n_ch


def _st_desc_64(D):
    """A descriptor that names all 64 pitches super_tone_rx.h:44 has room for (more than two lanes' 32 bins): call-progress
    pairs from 300 Hz up in 25 Hz steps, the pairs the synthetic lines really send among them."""
    d = D()
    sent = [350, 440, 480, 620, 950, 1100, 1400, 1800]
    rest = [f for f in range(300, 300 + 25*80, 25) if all(abs(f - x) > 10 for x in sent)][:56]
    base = sent + rest
    assert len(base) == 64
    for k in range(0, 64, 2):
        t = d.add_tone()
        d.add_element(t, base[k], base[k + 1], 300, 0)
        d.add_element(t, 0, 0, 200, 0)
    return d


@pytest.mark.parametrize("sizes", [[160], [160, 96, 256, 31]])
def test_super_tone_bank_of_64_bins(built, sizes):
    """Four lanes per channel (tone_bank_kernel<MultiDet<64>, 4>): block energies of all 64 bins, the pair picked, bit for bit."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 53                                       # a last wave that is not full (16 channels per wave)
    sig = synth.call_progress_channels(n_ch, 160*80, seed=35)
    desc = _st_desc_64(orc.SuperToneDesc)
    fac = list(desc.fac)
    assert len(fac) == 64 == engine.MAX_BINS
    bank = engine.ToneBank(engine.SUPER_TONE, n_ch, bin_fac=fac, trace=True)
    g = run_gpu(bank, sig, sizes)
    dets = [orc.SuperTone(desc) for _ in range(n_ch)]
    o = [[] for _ in dets]
    for pos, n in frames_of(sig.shape[1], sizes):
        for c, d in enumerate(dets):
            o[c].extend(list(d.rx(sig[c, pos:pos + n])))
    check_blocks(g, o, len(fac), "super-tone, 64 bins")
    assert sum(1 for c in range(n_ch) for b in g[c] if b[0] >= 0) > n_ch


def test_goertzel_bank_of_40_bins_and_its_state(built):
    """More than 32 plain Goertzel bins: block energies against goertzel_update() / goertzel_result() for frames that do not
    line up with the blocks, with every channel's state words moved into a second bank half way."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 37
    freqs = [300.0 + 43.0*i for i in range(40)]
    block = 205
    sig = synth.call_progress_channels(n_ch, 160*30, seed=43)
    facs = [engine.goertzel_fac(f) for f in freqs]
    banks = [engine.ToneBank(engine.GOERTZEL, n_ch, bin_fac=facs, block_len=block, trace=True) for _ in range(2)]
    got = [[] for _ in range(n_ch)]
    frames = list(frames_of(sig.shape[1], [160, 96, 31, 256]))
    for k, (pos, n) in enumerate(frames):
        if k == len(frames)//2:
            for c in range(n_ch):
                banks[1].set_state(c, *banks[0].get_state(c))
        bank = banks[0] if k < len(frames)//2 else banks[1]
        bank.rx_host(sig[:, pos:pos + n])
        blk = bank.blocks()
        if blk.size:
            tr = bank.trace()
            for r in blk:
                got[r["channel"]].append(tr[r["block"], :len(freqs), r["channel"]].copy())
    for c in range(n_ch):
        gs = [orc.Goertzel(f, block) for f in freqs]
        want = []
        pos = 0
        while pos + block <= sig.shape[1]:
            for gz in gs:
                assert gz.update(sig[c, pos:pos + block]) == block
            want.append(np.array([gz.result() for gz in gs], np.float32))
            pos += block
        assert len(got[c]) == len(want), c
        for x, y in zip(got[c], want):
            assert same_f32(x, y), c


def test_goertzel_bank(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 77
    freqs = [350.0, 440.0, 480.0, 620.0, 1400.0, 2100.0]
    block = 205
    sig = synth.call_progress_channels(n_ch, 160*40, seed=41)
    bank = engine.ToneBank(engine.GOERTZEL, n_ch, bin_fac=[engine.goertzel_fac(f) for f in freqs], block_len=block)
    got = [[] for _ in range(n_ch)]
    for pos, n in frames_of(sig.shape[1], [160]):
        bank.rx_host(sig[:, pos:pos + n])
        blk = bank.blocks()
        if blk.size:
            tr = bank.trace()
            for r in blk:
                got[r["channel"]].append(tr[r["block"], :len(freqs), r["channel"]].copy())
    for c in range(n_ch):
        gs = [orc.Goertzel(f, block) for f in freqs]
        want = []
        pos = 0
        while pos + block <= sig.shape[1]:
            for gz in gs:
                assert gz.update(sig[c, pos:pos + block]) == block
            want.append(np.array([gz.result() for gz in gs], np.float32))
            pos += block
        assert len(got[c]) == len(want), c
        for a, b in zip(got[c], want):
            assert same_f32(a, b), c


@pytest.mark.parametrize("name,freqs,block", [
    ("v18", [390.0, 980.0, 1180.0, 1270.0, 1300.0, 1400.0, 1650.0, 1800.0, 2225.0], 102),     # v18.c:177,200-211
    ("ademco", [1400.0, 2300.0], 55)])                                                        # ademco_contactid.c:446,467-468
def test_goertzel_bank_serves_the_other_goertzel_users(built, name, freqs, block):
    """the tone front ends of v18.c (:1546-1600) and ademco_contactid.c (:890-935) are a Goertzel tone
    set + the block's total energy + a few comparisons.  The generic bank delivers the energies and the total energy
    bit-exact; the comparisons (a host functor here) then give the reference's raw block decision."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch = 130
    rng = np.random.default_rng(5)
    n = block*60
    sig = np.zeros((n_ch, n))
    t = np.arange(n)
    for c in range(n_ch):
        for k in range(0, n, 1500):
            f = freqs[int(rng.integers(0, len(freqs)))]*float(rng.uniform(0.985, 1.015))
            on = int(rng.integers(300, 1400))
            sig[c, k:k + on] += synth.dbm0_to_amp(rng.uniform(-35.0, -8.0))*np.sin(2*np.pi*f*t[k:k + on]/8000.0 + rng.uniform(0, 6.28))
        sig[c] += rng.normal(0.0, rng.uniform(2.0, 60.0), n)
    sig = synth._finish(sig)
    bank = engine.ToneBank(engine.GOERTZEL, n_ch, bin_fac=[engine.goertzel_fac(f) for f in freqs], block_len=block)
    got = [[] for _ in range(n_ch)]
    for pos, m in frames_of(n, [160]):
        bank.rx_host(sig[:, pos:pos + m])
        blk = bank.blocks()
        if blk.size:
            tr = bank.trace()
            for r in blk:
                # trace rows: the compiled bins (bank.nbins >= len(freqs)), then the block's total energy
                row = tr[r["block"], :, r["channel"]]
                got[r["channel"]].append(np.concatenate([row[:len(freqs)], row[bank.nbins:bank.nbins + 1]]))
    # the literal constants of the float builds: ademco_contactid.c:461-462, v18.c:192 (its threshold is per object)
    threshold = np.float32(49728296.6) if name == "ademco" else np.float32(10.0**((-42.0 - 3.14)/10.0)*(block*32768.0*32768.0/2.0))
    fraction = np.float32(45.2233) if name == "ademco" else np.float32(83.868)

    def decide(e, total):
        if name == "ademco":
            # ademco_contactid.c:918-935
            if e[0] > threshold or e[1] > threshold:
                if e[0] > e[1]:
                    return 1 if e[0] > fraction*total else 0
                return 2 if e[1] > fraction*total else 0
            return 0
        # v18.c:1580-1600: strict > scan from zero, then the level and fraction-of-total tests
        best, at = np.float32(0.0), 0
        for i in range(len(e)):
            if e[i] > best:
                best, at = e[i], i
        return 0 if (best < threshold or best <= fraction*total) else at
    hits = 0
    for c in range(n_ch):
        gs = [orc.Goertzel(f, block) for f in freqs]
        for b in range(n//block):
            seg = sig[c, b*block:(b + 1)*block]
            for gz in gs:
                gz.update(seg)
            e = np.array([gz.result() for gz in gs], np.float32)
            total = np.float32(0.0)
            for v in seg.astype(np.float32):
                total = np.float32(total + v*v)
            assert same_f32(got[c][b][:len(freqs)], e), (c, b)
            assert same_f32(got[c][b][len(freqs):], np.array([total])), (c, b)
            d = decide(got[c][b][:len(freqs)], got[c][b][len(freqs)])
            assert d == decide(e, total)
            hits += int(d != 0)
    assert hits > n_ch


@pytest.mark.parametrize("kind", [1, 2], ids=["v18", "ademco"])
def test_goertzel_bank_functors(built, kind):
    """The raw block decisions of v18.c's tone scan and of the Ademco sender's handshake detector made on the device
    (spangpu_tone_params_t.functor): a block's `hit` equals the oracle's decision (oracle/tone_oracle.c, held to the
    reference's own in_tone / last_hit in test_oracle_pin.py), across frames that cut the blocks anywhere."""
    from oracle import restated as orc
    from spandsp_amd import engine
    from test_oracle_pin import functor_signal
    freqs, block = (orc.V18_TONE_SET, 102) if kind == 1 else (orc.ADEMCO_TONE_SET, 55)
    n_ch, n_blocks = 100, 160
    sig = np.stack([functor_signal(kind, 1000*kind + c, n_blocks) for c in range(n_ch)])
    bank = engine.ToneBank(engine.GOERTZEL, n_ch, bin_fac=[engine.goertzel_fac(f) for f in freqs], block_len=block,
                           functor=kind, functor_threshold=0.0)
    got = [[] for _ in range(n_ch)]
    for pos, m in frames_of(sig.shape[1], [160, 33, 160, 401]):
        bank.rx_host(sig[:, pos:pos + m])
        for r in bank.blocks():
            got[r["channel"]].append(int(r["hit"]))
    hits = 0
    for c in range(n_ch):
        want = orc.tone_functor_blocks(kind, sig[c], 0.0)
        assert np.array_equal(np.array(got[c], np.int32), want), (c, np.nonzero(np.array(got[c]) != want)[0][:5])
        hits += int(np.count_nonzero(want))
    assert hits > 10*n_ch
    with pytest.raises(engine.SpanGpuError):
        engine.ToneBank(engine.GOERTZEL, 4, bin_fac=[1.0, 1.1, 1.2], block_len=55, functor=engine.FUNCTOR_ADEMCO)


# --------------------------------------------------------------------------------------
# several banks in one launch
# --------------------------------------------------------------------------------------
def test_multi_bank_launch_equals_separate_launches(built):
    """spangpu_banks_rx(): DTMF + Bell MF + R2 MF + super-tone banks advanced by one kernel launch give exactly the
    records and state that separate launches give (ragged channel counts, several frames)."""
    import ctypes
    from spandsp_amd import engine
    # device buffers through the HIP runtime libspangpu already runs on (no second runtime in this process)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]

    def to_device(a):
        a = np.ascontiguousarray(a)
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), a.nbytes) == 0
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0          # hipMemcpyHostToDevice
        return p
    n = [70, 131, 64, 33]
    n_frames = 40
    sigs = [synth.dtmf_channels(n[0], 160*n_frames, seed=41)[0], synth.bell_mf_channels(n[1], 160*n_frames, seed=42)[0],
            synth.r2_mf_channels(n[2], 160*n_frames, seed=43, fwd=True)[0], synth.call_progress_channels(n[3], 160*n_frames, seed=44)]
    fac = [engine.goertzel_fac(f) for f in (350.0, 400.0, 440.0, 480.0, 620.0, 950.0, 1100.0, 1400.0)]

    def make():
        return [engine.ToneBank(engine.DTMF, n[0]), engine.ToneBank(engine.BELL_MF, n[1]),
                engine.ToneBank(engine.R2_MF, n[2], r2_fwd=True), engine.ToneBank(engine.SUPER_TONE, n[3], bin_fac=fac)]
    sep = make()
    fused = make()
    for b in fused[1:]:
        b.share_stream(fused[0])
    total = 0
    for k in range(n_frames):
        for b, x in zip(sep, sigs):
            b.rx_host(x[:, k*160:(k + 1)*160])
        frames = [to_device(x[:, k*160:(k + 1)*160]) for x in sigs]
        engine.banks_rx_device(fused, [f.value for f in frames], 160)
        for b0, b1 in zip(sep, fused):
            r0 = b0.blocks()
            r1 = b1.blocks()
            assert r0.tobytes() == r1.tobytes(), k
            total += int((r0["hit"] != 0).sum())
        for f in frames:
            hip.hipFree(f)
    for b0, b1, nn in zip(sep, fused, n):
        for c in (0, nn//2, nn - 1):
            f0, i0 = b0.get_state(c)
            f1, i1 = b1.get_state(c)
            assert same_f32(f0, f1) and np.array_equal(i0, i1), c
    assert total > 200
    # kinds that cannot share a launch are refused, not silently run some other way
    g = engine.ToneBank(engine.GOERTZEL, 8, bin_fac=fac[:4], block_len=100)
    g.share_stream(fused[0])
    buf = to_device(sigs[0][:, :160])
    with pytest.raises(engine.SpanGpuError):
        engine.banks_rx_device([fused[0], g], [buf.value, buf.value], 160)
    hip.hipFree(buf)


def test_records_straight_into_a_caller_buffer(built):
    """spangpu_bank_set_records_buffer(): the kernel writes its block records into a caller-owned device buffer (the
    zero-copy path of the multi-GPU gather); same records as the bank's own buffer gives."""
    import ctypes
    from spandsp_amd import engine
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n_ch = 300
    sig, _ = synth.dtmf_channels(n_ch, 160*30, seed=51)
    a = engine.ToneBank(engine.DTMF, n_ch)
    b = engine.ToneBank(engine.DTMF, n_ch)
    buf = ctypes.c_void_p()
    nbytes = 2*n_ch*4
    assert hip.hipMalloc(ctypes.byref(buf), 3*nbytes) == 0
    words = np.zeros(2*n_ch, np.uint32)
    hits = 0
    for k in range(30):
        frame = sig[:, k*160:(k + 1)*160]
        a.rx_host(frame)
        b.set_records_buffer(ctypes.c_void_p(buf.value + (k % 3)*nbytes), nbytes)
        b.rx_host(frame)
        ra = a.blocks()
        rb = b.blocks()
        assert ra.tobytes() == rb.tobytes(), k
        nb = len(ra)//n_ch
        assert hip.hipMemcpy(words.ctypes.data, ctypes.c_void_p(buf.value + (k % 3)*nbytes), nbytes, 2) == 0
        got = words[:nb*n_ch].reshape(nb, n_ch)
        want = (ra["hit"].astype(np.uint32) | (ra["code"].astype(np.uint32) << 8) | (ra["flags"].astype(np.uint32) << 16)).reshape(n_ch, nb).T
        assert np.array_equal(got, want), k
        hits += int((ra["hit"] != 0).sum())
    assert hits > 50
    b.set_records_buffer(None, 0)
    with pytest.raises(engine.SpanGpuError):
        b.set_records_buffer(buf, 8)
        b.rx_host(sig[:, :160])
    hip.hipFree(buf)


# --------------------------------------------------------------------------------------
# G.711 front end
# --------------------------------------------------------------------------------------
def _g711_encode(x, table):
    """Some G.711 code whose decoded value is nearest to x (test input only; the decode is what is under test)."""
    order = np.argsort(table.astype(np.int32), kind="stable")
    vals = table[order].astype(np.int32)
    pos = np.clip(np.searchsorted(vals, x.astype(np.int32)), 1, 255)
    lower = (x - vals[pos - 1]) <= (vals[pos] - x)
    return order[np.where(lower, pos - 1, pos)].astype(np.uint8)


@pytest.mark.parametrize("law", ["alaw", "ulaw"])
def test_g711_input_equals_decoded_linear_input(built, law):
    """A bank fed A-law / u-law bytes (decoded on the device) gives exactly the records, energies and state of a bank
    fed the reference's decode of the same bytes; DTMF and Bell MF, aligned and ragged frame lengths."""
    from spandsp_amd import engine
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g711_decode.npz"))
    table = g[law]
    code = engine.G711_ALAW if law == "alaw" else engine.G711_ULAW
    for kind, make_sig, n_ch in ((engine.DTMF, synth.dtmf_channels, 150), (engine.BELL_MF, synth.bell_mf_channels, 70)):
        sig, _ = make_sig(n_ch, 160*40, seed=61)
        codes = _g711_encode(sig, table)
        lin = table[codes]                                   # what the reference's decoder makes of those bytes
        a = engine.ToneBank(kind, n_ch, trace=True)
        b = engine.ToneBank(kind, n_ch, trace=True)
        hits = 0
        for pos, n in frames_of(sig.shape[1], [160, 160, 37, 240, 400]):
            a.rx_host(lin[:, pos:pos + n])
            b.rx_host_g711(codes[:, pos:pos + n], code)
            ra = a.blocks()
            rb = b.blocks()
            assert ra.tobytes() == rb.tobytes(), (kind, pos)
            ta = a.trace(4)
            tb = b.trace(4)
            assert same_f32(ta, tb), (kind, pos)
            hits += int((ra["hit"] != 0).sum())
        for c in (0, n_ch//2, n_ch - 1):
            fa, ia = a.get_state(c)
            fb, ib = b.get_state(c)
            assert same_f32(fa, fb) and np.array_equal(ia, ib), (kind, c)
        assert hits > 50


# --------------------------------------------------------------------------------------
# Round 5: the production block ends of the MF detectors, queue mode, banks on streams of their own
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["dtmf", "bell", "r2"])
def test_lean_block_end_equals_the_scan(built, kind, lanes_per_channel):
    """Det::decide_plain() (what the streaming kernels run when nothing but the decision is asked for) against Det::decide()
    (the reference's scans with their tie-breaks: dtmf.c:209-258, bell_r2_mf.c:556-661,793-876) on energies a synthesised
    signal rarely produces: equal values in every position, zeros, values a rounding away from every threshold and ratio, and
    every history the debounce / five-block rules can be in.  Record word and both state words, bit for bit."""
    from spandsp_amd import engine
    if lanes_per_channel != 1:
        pytest.skip("one run is enough: the hook does not go through the bank kernels")
    L = engine.lib()
    L.spangpu_debug_decide.restype = C.c_int
    L.spangpu_debug_decide.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(1234)
    n = 400000
    thr = {"dtmf": 171029200.0, "bell": 3343803100.0, "r2": 1031766650.0}[kind]
    levels = np.array([0.0, thr*0.01, thr/12.589, thr/6.309, thr*0.5, thr, thr*1.5, thr*2.512, thr*3.981, thr*5.012, thr*6.309,
                       thr*12.589, thr*40.0, thr*83.868], np.float32)
    e = np.empty((n, 8), np.float32)
    # a third: values from a small set of levels (ties everywhere); a third: the same, each a few ulps off; a third: log-uniform
    third = n//3
    e[:third] = levels[rng.integers(0, len(levels), (third, 8))]
    base = levels[rng.integers(0, len(levels), (third, 8))]
    e[third:2*third] = (base.view(np.uint32) + rng.integers(-2, 3, (third, 8)).astype(np.int64)).clip(0, 0x7F000000).astype(np.uint32).view(np.float32)
    e[2*third:] = (thr*np.exp(rng.uniform(-6.0, 6.0, (n - 2*third, 8)))).astype(np.float32)
    # mostly two-tone shapes: push all but two (MF) / one per group (DTMF) down in half of the cases
    quiet = rng.random(n) < 0.5
    for i in np.nonzero(quiet)[0][:n//4]:
        keep = rng.choice(6, 2, replace=False) if kind != "dtmf" else np.array([rng.integers(0, 4), 4 + rng.integers(0, 4)])
        m = np.ones(8, bool)
        m[keep] = False
        e[i, m] *= np.float32(1.0e-3)
    energy = (e[:, :8].sum(axis=1)*rng.choice([0.001, 1.0/83.868, 0.012, 0.02], n)).astype(np.float32)
    keys = {"dtmf": b"\x00123A456B789C*0#D", "bell": b"\x00123456789*0#ABC", "r2": b"\x00123456789BCDEF"}[kind]
    pick = lambda: np.frombuffer(keys, np.uint8)[rng.integers(0, len(keys), n)].astype(np.uint32)
    w0 = (pick() << 16) | (pick() << 24) | rng.integers(0, 102, n).astype(np.uint32)
    w1 = (pick() | (pick() << 8) | (pick() << 16)).astype(np.int32) if kind == "bell" else rng.integers(0, 100000, n).astype(np.int32)
    # histories that agree with each other, so that reports happen: copy one key into several positions for a quarter of the cases
    same = rng.random(n) < 0.25
    k = pick()
    w0 = np.where(same, (w0 & 0xFFFF) | (k << 16) | (k << 24), w0).astype(np.uint32)
    if kind == "bell":
        w1 = np.where(same & (rng.random(n) < 0.7), (k | (k << 8) | (k << 16)).astype(np.int32), w1).astype(np.int32)
    out = np.zeros((6, n), np.uint32)
    kid = {"dtmf": engine.DTMF, "bell": engine.BELL_MF, "r2": engine.R2_MF}[kind]

    def both(w0, w1):
        w0 = np.ascontiguousarray(w0, np.uint32)
        w1 = np.ascontiguousarray(w1, np.int32)
        rc = L.spangpu_debug_decide(kid, e.ctypes.data, energy.ctypes.data, w0.ctypes.data, w1.ctypes.data, out.ctypes.data, n)
        assert rc == 0, L.spangpu_last_error()
        bad = np.nonzero((out[0] != out[3]) | (out[1] != out[4]) | (out[2] != out[5]))[0]
        assert bad.size == 0, (kind, bad[:5], e[bad[:2]], [hex(int(x)) for x in out[:, bad[0]]])
        return int(((out[0] & 0xFF) != 0).sum()), int((((out[0] >> 16) & engine.BLK_REPORT) != 0).sum())
    hits, _ = both(w0, w1)
    assert hits > n//100, hits
    # again with histories built around the hit each case produced, so that the debounce / five-block / change rules fire:
    # the hit in the newest positions, the older ones the hit or something else
    hit = (out[0] & 0xFF).astype(np.uint32)
    other = pick()
    a = np.where(rng.random(n) < 0.5, hit, other).astype(np.uint32)
    b = np.where(rng.random(n) < 0.5, hit, other).astype(np.uint32)
    c = np.where(rng.random(n) < 0.3, hit, pick()).astype(np.uint32)
    if kind == "bell":
        hits, reports = both((c << 16) | (b << 24), (a | (hit << 8) | (hit << 16)).astype(np.int32))      # hits[0..4] = c, b, a, hit, hit
    else:
        hits, reports = both((a << 16) | (b << 24), w1)                                                     # last_hit / current = a, in_digit = b
    assert hits > n//100 and reports > n//50, (hits, reports)


def test_queue_mode_equals_one_launch(built, lanes_per_channel):
    """spangpu_bank_set_queues(bank, 2): the streaming kernel's launch cut in two on two hardware queues leaves the records
    and the state of one launch, with launches of the other kind (sample-major frames: the general kernel), state reads and a
    parameter change in between -- every one of which has to join the second queue first."""
    from spandsp_amd import engine
    n_ch = 2048 + 70                    # nine workgroups: four on the bank's stream, five on the second
    n_frames = 40
    sig, _ = synth.dtmf_channels(n_ch, 160*n_frames, seed=91)
    one = engine.ToneBank(engine.DTMF, n_ch)
    two = engine.ToneBank(engine.DTMF, n_ch)
    assert two.set_queues(2) == 2 and one.set_queues(0) == 1 and one.set_queues(1) == 1
    total = 0
    for k in range(n_frames):
        fr = sig[:, k*160:(k + 1)*160]
        for b in (one, two):
            if k % 7 == 3:
                b.rx_host(np.ascontiguousarray(fr.T), layout=1)
            else:
                b.rx_host(fr)
        r0 = one.blocks()
        r1 = two.blocks()
        assert r0.tobytes() == r1.tobytes(), k
        total += int(((r0["flags"] & engine.BLK_CHANGE) != 0).sum())
        if k % 5 == 4:
            for c in (0, 1023, 1024, n_ch - 1):
                f0, i0 = one.get_state(c)
                f1, i1 = two.get_state(c)
                assert same_f32(f0, f1) and np.array_equal(i0, i1), (k, c)
        if k == 20:
            for b in (one, two):
                b.set_channel_params(1500, threshold_dbm0=-30.0)
    assert total > n_ch//2
    assert two.set_queues(1) == 1
    two.rx_host(sig[:, :160])
    one.rx_host(sig[:, :160])
    assert one.blocks().tobytes() == two.blocks().tobytes()


def test_banks_on_streams_of_their_own(built, lanes_per_channel):
    """spangpu_banks_rx() with banks that were given streams of their own: a launch each, every one on its bank's stream --
    the records and state of the shared launch."""
    import ctypes
    from spandsp_amd import engine
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]

    def to_device(a):
        a = np.ascontiguousarray(a)
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), a.nbytes) == 0
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        return p
    n = [300, 131, 257]
    n_frames = 30
    sigs = [synth.bell_mf_channels(n[0], 160*n_frames, seed=52)[0], synth.r2_mf_channels(n[1], 160*n_frames, seed=53, fwd=True)[0],
            synth.call_progress_channels(n[2], 160*n_frames, seed=54)]
    fac = [engine.goertzel_fac(f) for f in (350.0, 400.0, 440.0, 480.0, 620.0, 950.0, 1100.0, 1400.0)]

    def make():
        return [engine.ToneBank(engine.BELL_MF, n[0]), engine.ToneBank(engine.R2_MF, n[1], r2_fwd=True),
                engine.ToneBank(engine.SUPER_TONE, n[2], bin_fac=fac)]
    shared = make()
    for b in shared[1:]:
        b.share_stream(shared[0])
    own = make()
    streams = []
    for b in own:
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0           # hipStreamNonBlocking
        streams.append(s)
        b.set_stream(s)
    dev = [[to_device(x[:, k*160:(k + 1)*160]) for x in sigs] for k in range(n_frames)]
    # the banks on their own streams run all their ticks before anything is read: the queues run free
    for k in range(n_frames):
        engine.banks_rx_device(own, [f.value for f in dev[k]], 160)
    last = [b.blocks() for b in own]
    for k in range(n_frames):
        engine.banks_rx_device(shared, [f.value for f in dev[k]], 160)
    hits = 0
    for b0, b1, r1, nn in zip(shared, own, last, n):
        r0 = b0.blocks()
        assert r0.tobytes() == r1.tobytes()
        hits += int((r0["hit"] != 0).sum())
        for c in (0, nn//2, nn - 1):
            f0, i0 = b0.get_state(c)
            f1, i1 = b1.get_state(c)
            assert same_f32(f0, f1) and np.array_equal(i0, i1), c
    assert hits > 20
    # ... and on the streams spangpu_banks_own_queues() makes for them (a hardware queue each, by stream priority), a fourth
    # bank included (more banks than priority levels: it gets a plain stream)
    third = make() + [engine.ToneBank(engine.DTMF, 97)]
    levels = engine.banks_own_queues(third)
    assert 1 <= levels <= 4
    assert len(set(engine.lib().spangpu_bank_get_stream(b.h) for b in third)) == 4
    for k in range(n_frames):
        engine.banks_rx_device(third[:3], [f.value for f in dev[k]], 160)
    for b0, b2 in zip(shared, third):
        assert b0.blocks().tobytes() == b2.blocks().tobytes()
    assert engine.banks_own_queues(third[:2]) >= 1         # again: the streams of before are given back
    # banks are grouped by stream: two of the three on one stream share a launch, the third has its own
    fourth = make()
    assert engine.banks_own_queues([fourth[0], fourth[2]]) >= 1
    fourth[1].share_stream(fourth[0])
    for k in range(n_frames):
        engine.banks_rx_device(fourth, [f.value for f in dev[k]], 160)
    for b0, b4 in zip(shared, fourth):
        assert b0.blocks().tobytes() == b4.blocks().tobytes()
    for b in fourth:
        b.close()
    for b in own + third:
        b.close()
    for s in streams:
        hip.hipStreamDestroy(s)
    for fr in dev:
        for f in fr:
            hip.hipFree(f)
