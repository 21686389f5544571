"""A live call changes sides: state in the reference's own struct layout (include/spangpu_refstate.h).  A detector /
canceller made and run by the REAL reference (oracle/_ref/libspandsp_ref.so) is imported into a bank channel in
mid-stream and both go on -- what the bank reports must be what the reference reports; and the other way: a bank
channel is exported into a fresh reference object, which then carries on.  The reference struct is never described to
Python: its pointer is handed to the library as it is, so the layout in the header is what is under test."""
import ctypes as C

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

DIGITS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int)


@pytest.fixture(scope="module")
def libs(built):
    from oracle import ref
    from spandsp_amd import engine
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    R = C.CDLL(ref.REF_SO)
    vp, ci = C.c_void_p, C.c_int
    R.dtmf_rx_init.restype = vp
    R.dtmf_rx_init.argtypes = [vp, vp, vp]
    R.dtmf_rx.argtypes = [vp, vp, ci]
    R.dtmf_rx_get.restype = C.c_size_t
    R.dtmf_rx_get.argtypes = [vp, C.c_char_p, ci]
    R.dtmf_rx_parms.argtypes = [vp, ci, C.c_float, C.c_float, C.c_float]
    R.dtmf_rx_free.argtypes = [vp]
    R.echo_can_init.restype = vp
    R.echo_can_init.argtypes = [ci, ci]
    R.echo_can_update.restype = C.c_int16
    R.echo_can_update.argtypes = [vp, C.c_int16, C.c_int16]
    R.echo_can_free.argtypes = [vp]
    L = engine.lib()
    L.spangpu_dtmf_import_state.argtypes = [vp, ci, vp]
    L.spangpu_dtmf_export_state.argtypes = [vp, ci, vp]
    L.spangpu_echo_import_state.argtypes = [vp, ci, vp]
    L.spangpu_echo_export_state.argtypes = [vp, ci, vp]
    return R, L


def ref_digits(R, s, x):
    """The digits the reference collects over x, fed in 160-sample frames."""
    out = ""
    buf = C.create_string_buffer(200)
    for k in range(0, len(x), 160):
        fr = np.ascontiguousarray(x[k:k + 160])
        R.dtmf_rx(s, fr.ctypes.data, len(fr))
        n = R.dtmf_rx_get(s, buf, 128)
        out += buf.raw[:n].decode("latin1")
    return out


def bank_digits(bank, ch, x, n_ch):
    from spandsp_amd import engine
    out = ""
    for k in range(0, len(x), 160):
        fr = np.zeros((n_ch, 160), np.int16)
        m = min(160, len(x) - k)
        lens = np.zeros(n_ch, np.int32)
        fr[ch, :m] = x[k:k + m]
        lens[ch] = m
        bank.rx_host_var(fr, lens)
        for r in bank.blocks():
            if r["channel"] == ch and (r["flags"] & engine.BLK_CHANGE) and r["code"]:
                out += chr(int(r["code"]))
    return out


@pytest.mark.parametrize("parms", [None, (1, 5.0, 3.0, -35.0)], ids=["defaults", "own-parms"])
def test_dtmf_call_changes_sides(libs, parms):
    from spandsp_amd import engine
    R, L = libs
    sig, _ = synth.dtmf_channels(4, 160*120, seed=71)
    t = np.arange(sig.shape[1])
    dial = 2500.0*np.sin(2*np.pi*350.0*t/8000.0) + 2500.0*np.sin(2*np.pi*440.0*t/8000.0)
    n_ch, ch = 5, 3
    total = 0
    for c in range(4):
        x = np.clip(sig[c].astype(np.float64) + (dial if parms else 0.0), -32768, 32767).astype(np.int16)
        cut = 160*47 + 61                       # in the middle of a frame and of a 102-sample block
        # the whole call on the reference
        whole = R.dtmf_rx_init(None, None, None)
        if parms:
            R.dtmf_rx_parms(whole, *parms)
        want = ref_digits(R, whole, x[:cut]) + ref_digits(R, whole, x[cut:])
        total += len(want)
        # reference -> bank
        a = R.dtmf_rx_init(None, None, None)
        if parms:
            R.dtmf_rx_parms(a, *parms)
        first = ref_digits(R, a, x[:cut])
        bank = engine.ToneBank(engine.DTMF, n_ch)
        assert L.spangpu_dtmf_import_state(bank.h, ch, a) == 0
        assert first + bank_digits(bank, ch, x[cut:], n_ch) == want, c
        # bank -> reference
        bank2 = engine.ToneBank(engine.DTMF, n_ch)
        if parms:
            bank2.set_channel_params(ch, filter_dialtone=parms[0], twist_db=parms[1], reverse_twist_db=parms[2], threshold_dbm0=parms[3])
        first2 = bank_digits(bank2, ch, x[:cut], n_ch)
        b = R.dtmf_rx_init(None, None, None)
        assert L.spangpu_dtmf_export_state(bank2.h, ch, b) == 0
        assert first2 + ref_digits(R, b, x[cut:]) == want, c
        # and the exported struct is, byte for byte in its signal-processing part, the one the reference built itself
        probe = R.dtmf_rx_init(None, None, None)
        assert L.spangpu_dtmf_export_state(bank.h, ch, probe) == 0
        R.dtmf_rx(a, np.ascontiguousarray(x[cut:]).ctypes.data, len(x) - cut)      # `a` catches up with the bank
        lo, hi = 4*C.sizeof(C.c_void_p), 4*C.sizeof(C.c_void_p) + 4 + 4*8 + 8*20 + 4 + 8       # filter_dialtone .. duration
        assert C.string_at(probe + lo, hi - lo) == C.string_at(a + lo, hi - lo), c
        for s in (whole, a, b, probe):
            R.dtmf_rx_free(s)
        bank.close()
        bank2.close()
    assert total > 15                           # (some of the synthetic channels are too poor to yield a digit: part of the test)


def test_echo_call_changes_sides(libs):
    from spandsp_amd import engine
    from test_echo_gpu import make_channels
    R, L = libs
    taps, mode = 128, 0x01 | 0x02 | 0x04 | 0x40
    tx, rx = make_channels(3, 160*40, taps, seed=73)
    tx, rx = tx[2, 160*14:160*30], rx[2, 160*14:160*30]
    cut = 160*9 + 77
    n_ch, ch = 4, 2

    def ref_run(ec, a, b):
        return np.array([R.echo_can_update(ec, int(t), int(r)) for t, r in zip(a, b)], np.int16)

    def bank_run(bank, a, b):
        out = []
        for k in range(0, len(a), 160):
            m = min(160, len(a) - k)
            t2 = np.zeros((n_ch, m), np.int16)
            r2 = np.zeros((n_ch, m), np.int16)
            t2[ch], r2[ch] = a[k:k + m], b[k:k + m]
            out.append(bank.update_host(t2, r2, False)[ch])
        return np.concatenate(out)

    whole = R.echo_can_init(taps, mode)
    want = ref_run(whole, tx, rx)
    assert np.any(want[cut:] != rx[cut:])
    # reference -> bank
    a = R.echo_can_init(taps, mode)
    ref_run(a, tx[:cut], rx[:cut])
    bank = engine.EchoBank(n_ch, taps, mode)
    assert L.spangpu_echo_import_state(bank.h, ch, a) == 0
    assert np.array_equal(bank_run(bank, tx[cut:], rx[cut:]), want[cut:])
    # bank -> reference
    bank2 = engine.EchoBank(n_ch, taps, mode)
    first = bank_run(bank2, tx[:cut], rx[:cut])
    assert np.array_equal(first, want[:cut])
    b = R.echo_can_init(taps, mode)
    assert L.spangpu_echo_export_state(bank2.h, ch, b) == 0
    assert np.array_equal(ref_run(b, tx[cut:], rx[cut:]), want[cut:])
    other = R.echo_can_init(64, mode)
    assert L.spangpu_echo_import_state(bank.h, ch, other) != 0             # another length: refused
    for ec in (whole, a, b, other):
        R.echo_can_free(ec)
    bank.close()
    bank2.close()


MODEMS = {"v29": ("v29_9600.npz", 9600, "V29Rx", "V29Bank", "V27terBank", 4800),
          "v27ter": ("v27ter_4800.npz", 4800, "V27terRx", "V27terBank", "V29Bank", 9600),
          "v17": ("v17_14400.npz", 14400, "V17Rx", "V17Bank", "V29Bank", 9600)}


DATA_CUT = {"v29": 160*20 + 77, "v27ter": 160*45 + 77, "v17": 160*85 + 77}          # well after each modem's training


@pytest.mark.parametrize("where", ["in-training", "in-data"])
@pytest.mark.parametrize("modem", ["v29", "v27ter", "v17"])
def test_modem_call_changes_sides(libs, modem, where):
    """A modem receiver handed over in mid-call, trained equaliser, loops, scrambler and all: reference -> bank channel and
    bank channel -> reference; the bits and status events delivered across the change are those of one receiver that ran
    the whole call."""
    import os
    from oracle import ref
    from spandsp_amd import engine
    from test_oracle_pin import GOLDEN
    R, L = libs
    fixture, rate, ref_cls, bank_cls, other_cls, other_rate = MODEMS[modem]
    vp, ci = C.c_void_p, C.c_int
    imp = getattr(L, "spangpu_%s_import_state" % modem)
    exp = getattr(L, "spangpu_%s_export_state" % modem)
    imp.argtypes = [vp, ci, vp]
    exp.argtypes = [vp, ci, vp]
    x = np.load(os.path.join(GOLDEN, fixture))["amp"]
    cut = DATA_CUT[modem] if where == "in-data" else 160*7 + 5
    assert cut + 1000 < len(x)
    n_ch, ch = 3, 1
    make_ref = getattr(ref, ref_cls)
    make_bank = getattr(engine, bank_cls)

    def ref_bits(rx, seg):
        rx.sink.clear()
        for k in range(0, len(seg), 160):
            rx.rx(seg[k:k + 160])
        return [int(e["a"]) for e in rx.sink.events()]

    def bank_bits(bank, seg):
        out = []
        for k in range(0, len(seg), 160):
            m = min(160, len(seg) - k)
            fr = np.zeros((n_ch, 160), np.int16)
            lens = np.zeros(n_ch, np.int32)
            fr[ch, :m] = seg[k:k + m]
            lens[ch] = m
            bank.rx_host_var(fr, lens)
            out.extend(int(b) for b in bank.events()[ch])
        return out

    whole = make_ref(rate)
    want = ref_bits(whole, x[:cut]) + ref_bits(whole, x[cut:])
    assert -4 in want and sum(1 for b in want if b >= 0) > 2000          # trained, and data came through
    # reference -> bank
    a = make_ref(rate)
    first = ref_bits(a, x[:cut])
    assert (-4 in first) == (where == "in-data")
    bank = make_bank(n_ch, rate)
    assert imp(bank.h, ch, a.p) == 0
    assert first + bank_bits(bank, x[cut:]) == want
    # bank -> reference
    bank2 = make_bank(n_ch, rate)
    first2 = bank_bits(bank2, x[:cut])
    b = make_ref(rate)
    assert exp(bank2.h, ch, b.p) == 0
    assert first2 + ref_bits(b, x[cut:]) == want
    # and what the bank exports after the whole call is, word for word, the state of the receiver that ran it all
    probe = make_ref(rate)
    assert exp(bank.h, ch, probe.p) == 0
    fw, iw = whole.snapshot()
    fp, ip = probe.snapshot()
    assert np.array_equal(fw.view(np.uint32), fp.view(np.uint32)) and np.array_equal(iw, ip)
    # a bank of another modem refuses
    other = getattr(engine, other_cls)(n_ch, other_rate)
    assert imp(other.h, ch, a.p) < 0
    if modem != "v29":
        # and so does a bank of the same modem at another rate (tables, constellation and space map are per rate)
        slow = make_bank(n_ch, 2400 if modem == "v27ter" else 9600)
        assert imp(slow.h, ch, a.p) < 0


def test_bell_mf_and_r2_mf_calls_change_sides(libs):
    """The two MF detectors: six Goertzel states, the block position, and the hit history (Bell MF) or the digit being
    reported (R2) move between the real reference and a bank channel in the middle of a block."""
    from oracle import ref
    from spandsp_amd import engine
    R, L = libs
    vp, ci = C.c_void_p, C.c_int
    Rl = ref.lib()
    for name in ("spangpu_bell_mf_import_state", "spangpu_bell_mf_export_state", "spangpu_r2_mf_import_state", "spangpu_r2_mf_export_state"):
        getattr(L, name).argtypes = [vp, ci, vp]
    n_ch, ch = 4, 2
    cut = 160*23 + 71

    def bank_hits(bank, seg):
        """(block digit, flags) of every block that reports something, in order"""
        out = []
        for k in range(0, len(seg), 160):
            m = min(160, len(seg) - k)
            fr = np.zeros((n_ch, 160), np.int16)
            lens = np.zeros(n_ch, np.int32)
            fr[ch, :m] = seg[k:k + m]
            lens[ch] = m
            bank.rx_host_var(fr, lens)
            for r in bank.blocks():
                if r["channel"] == ch and (r["flags"] & engine.BLK_REPORT):
                    out.append(int(r["code"]))
        return out

    # ---- Bell MF: the digits collected ----
    sig, _ = synth.bell_mf_channels(3, 160*60, seed=81)
    for x in sig:
        def ref_digits(rx, seg):
            out = ""
            for k in range(0, len(seg), 160):
                rx.rx(seg[k:k + 160])
                out += rx.get()
            return out
        whole = ref.BellMfRx(0)
        want = ref_digits(whole, x[:cut]) + ref_digits(whole, x[cut:])
        assert len(want) >= 3
        a = ref.BellMfRx(0)
        first = ref_digits(a, x[:cut])
        bank = engine.ToneBank(engine.BELL_MF, n_ch)
        assert L.spangpu_bell_mf_import_state(bank.h, ch, a.p) == 0
        assert first + "".join(chr(c) for c in bank_hits(bank, x[cut:])) == want
        bank2 = engine.ToneBank(engine.BELL_MF, n_ch)
        first2 = "".join(chr(c) for c in bank_hits(bank2, x[:cut]))
        b = ref.BellMfRx(0)
        assert L.spangpu_bell_mf_export_state(bank2.h, ch, b.p) == 0
        assert first2 + ref_digits(b, x[cut:]) == want
        # the exported detector is, in what moved, the one the reference built itself
        probe = ref.BellMfRx(0)
        assert L.spangpu_bell_mf_export_state(bank.h, ch, probe.p) == 0
        sw, sp = whole.snapshot(), probe.snapshot()
        for key in ("v2", "v3", "fac", "hits"):
            assert np.array_equal(np.asarray(sw[key]).view(np.uint32), np.asarray(sp[key]).view(np.uint32)), key
        assert sw["current_sample"] == sp["current_sample"]
        assert L.spangpu_bell_mf_import_state(engine.ToneBank(engine.DTMF, n_ch).h, ch, a.p) < 0
    # ---- R2: the changes of the digit present ----
    for fwd in (True, False):
        sig, _ = synth.r2_mf_channels(4, 160*60, seed=83, fwd=fwd)
        moved = 0
        for x in sig:
            def ref_changes(rx, seg):
                rx.sink.clear()
                for k in range(0, len(seg), 160):
                    rx.rx(seg[k:k + 160])
                return [int(e["a"]) for e in rx.sink.events() if e["kind"] == 1]
            whole = ref.R2MfRx(fwd, use_callback=True)
            want = ref_changes(whole, x[:cut]) + ref_changes(whole, x[cut:])
            if len(want) < 4:
                continue                    # (some of the synthetic channels are too poor to yield a digit)
            moved += 1
            a = ref.R2MfRx(fwd, use_callback=True)
            first = ref_changes(a, x[:cut])
            bank = engine.ToneBank(engine.R2_MF, n_ch, r2_fwd=fwd)
            assert L.spangpu_r2_mf_import_state(bank.h, ch, a.p) == 0
            assert first + bank_hits(bank, x[cut:]) == want
            bank2 = engine.ToneBank(engine.R2_MF, n_ch, r2_fwd=fwd)
            first2 = bank_hits(bank2, x[:cut])
            b = ref.R2MfRx(fwd, use_callback=True)
            assert L.spangpu_r2_mf_export_state(bank2.h, ch, b.p) == 0
            assert first2 + ref_changes(b, x[cut:]) == want
            wrong = engine.ToneBank(engine.R2_MF, n_ch, r2_fwd=not fwd)
            assert L.spangpu_r2_mf_import_state(wrong.h, ch, a.p) < 0
        assert moved >= 2


@pytest.mark.parametrize("which,mode", [(1, 1), (1, 0), (1, 2)], ids=["v21ch2-sync", "v21ch2-async", "v21ch2-framed"])
def test_fsk_call_changes_sides(libs, which, mode):
    """An FSK receiver handed over in the middle of a transmission, correlation window, oscillators, baud phase and
    framing state included: reference -> bank channel and bank channel -> reference."""
    from oracle import ref
    from spandsp_amd import engine
    from test_oracle_pin import fsk_scenario
    R, L = libs
    vp, ci = C.c_void_p, C.c_int
    L.spangpu_fsk_import_state.argtypes = [vp, ci, vp]
    L.spangpu_fsk_export_state.argtypes = [vp, ci, vp]
    x = fsk_scenario(which, mode)
    cut = 160*31 + 53
    assert cut + 2000 < len(x)
    n_ch, ch = 3, 2

    def ref_bits(rx, seg):
        rx.sink.clear()
        for k in range(0, len(seg), 160):
            rx.rx(seg[k:k + 160])
        return [int(e["a"]) for e in rx.sink.events()]

    def bank_bits(bank, seg):
        out = []
        for k in range(0, len(seg), 160):
            m = min(160, len(seg) - k)
            fr = np.zeros((n_ch, 160), np.int16)
            lens = np.zeros(n_ch, np.int32)
            fr[ch, :m] = seg[k:k + m]
            lens[ch] = m
            bank.rx_host_var(fr, lens)
            out.extend(int(b) for b in bank.events()[ch])
        return out

    whole = ref.FskRx(which, mode)
    want = ref_bits(whole, x[:cut]) + ref_bits(whole, x[cut:])
    assert len(want) > 40
    a = ref.FskRx(which, mode)
    first = ref_bits(a, x[:cut])
    bank = engine.FskBank(which, n_ch, mode)
    assert L.spangpu_fsk_import_state(bank.h, ch, a.p) == 0
    assert first + bank_bits(bank, x[cut:]) == want
    bank2 = engine.FskBank(which, n_ch, mode)
    first2 = bank_bits(bank2, x[:cut])
    b = ref.FskRx(which, mode)
    assert L.spangpu_fsk_export_state(bank2.h, ch, b.p) == 0
    assert first2 + ref_bits(b, x[cut:]) == want
    probe = ref.FskRx(which, mode)
    assert L.spangpu_fsk_export_state(bank.h, ch, probe.p) == 0
    assert np.array_equal(whole.snapshot(), probe.snapshot())
    other = engine.FskBank(0 if which else 1, n_ch, mode)           # another spec: refused
    assert L.spangpu_fsk_import_state(other.h, ch, a.p) < 0


@pytest.mark.parametrize("rx_type,tx_kind", [(2, 4), (7, "preamble"), (7, 5), (1, 1)])
def test_connect_tone_call_changes_sides(libs, rx_type, tx_kind):
    """A modem connect tone detector -- its V.21 receiver inside it where it hunts for the FAX preamble -- handed over in
    mid-signal, both ways."""
    from oracle import ref
    from spandsp_amd import engine
    from test_oracle_pin import mct_scenario
    R, L = libs
    vp, ci = C.c_void_p, C.c_int
    L.spangpu_mct_import_state.argtypes = [vp, ci, vp]
    L.spangpu_mct_export_state.argtypes = [vp, ci, vp]
    x = mct_scenario(rx_type, tx_kind)
    n_ch, ch = 3, 0
    for cut in (160*20 + 9, 160*60 + 101):
        def ref_reports(rx, seg):
            rx.sink.clear()
            for k in range(0, len(seg), 160):
                rx.rx(seg[k:k + 160])
            return [(int(e["a"]), int(e["b"])) for e in rx.sink.events() if e["kind"] == 1]

        def bank_reports(bank, seg):
            out = []
            for k in range(0, len(seg), 160):
                m = min(160, len(seg) - k)
                fr = np.zeros((n_ch, 160), np.int16)
                lens = np.zeros(n_ch, np.int32)
                fr[ch, :m] = seg[k:k + m]
                lens[ch] = m
                bank.rx_host_var(fr, lens)
                out.extend((int(t), int(lv)) for t, lv in bank.events()[ch])
            return out
        whole = ref.MctRx(rx_type)
        want = ref_reports(whole, x[:cut]) + ref_reports(whole, x[cut:])
        assert len(want) >= 2
        a = ref.MctRx(rx_type)
        first = ref_reports(a, x[:cut])
        bank = engine.MctBank(rx_type, n_ch)
        assert L.spangpu_mct_import_state(bank.h, ch, a.p) == 0
        assert first + bank_reports(bank, x[cut:]) == want
        bank2 = engine.MctBank(rx_type, n_ch)
        first2 = bank_reports(bank2, x[:cut])
        b = ref.MctRx(rx_type)
        assert L.spangpu_mct_export_state(bank2.h, ch, b.p) == 0
        assert first2 + ref_reports(b, x[cut:]) == want
        probe = ref.MctRx(rx_type)
        assert L.spangpu_mct_export_state(bank.h, ch, probe.p) == 0
        assert np.array_equal(whole.snapshot(), probe.snapshot())
        assert L.spangpu_mct_import_state(engine.MctBank(8, n_ch).h, ch, a.p) < 0        # another tone type: refused


@pytest.mark.parametrize("tone_type,mode", [(1, 0x40), (2, 0xC0), (3, 0x40)])
def test_sig_tone_call_changes_sides(libs, tone_type, mode):
    """A signalling tone receiver handed over in mid-frame with a tone present: filter states, meters, timers, mode."""
    from oracle import ref
    from spandsp_amd import engine
    R, L = libs
    vp, ci = C.c_void_p, C.c_int
    L.spangpu_sig_tone_rx_import_state.argtypes = [vp, ci, vp]
    L.spangpu_sig_tone_rx_export_state.argtypes = [vp, ci, vp]
    sig = synth.sig_tone_channels(4, 8000*4, 140 + tone_type, tone_type)
    n_ch, ch = 3, 1
    moved = 0
    for x in sig:
        def ref_run(rx, seg):
            rx.sink.clear()
            out = [rx.rx(seg[k:k + 160]) for k in range(0, len(seg), 160)]
            return np.concatenate(out), [(int(e["a"]), int(e["c"])) for e in rx.sink.events()]

        def bank_run(bank, seg):
            out = []
            ev = []
            for k in range(0, len(seg), 160):
                m = min(160, len(seg) - k)
                fr = np.zeros((n_ch, 160), np.int16)
                lens = np.zeros(n_ch, np.int32)
                fr[ch, :m] = seg[k:k + m]
                lens[ch] = m
                got = bank.rx_host_var(fr, lens)
                out.append(got[ch, :m])
                ev.extend((int(s), int(d)) for _, s, d in bank.events()[ch])
            return np.concatenate(out), ev
        whole = ref.SigToneRx(tone_type, mode)
        probe_run = ref_run(whole, x)
        if len(probe_run[1]) < 4:
            continue
        moved += 1
        cut = 160*41 + 37
        want_out, want_ev = probe_run
        a = ref.SigToneRx(tone_type, mode)
        o1, e1 = ref_run(a, x[:cut])
        bank = engine.SigToneRxBank(tone_type, n_ch)
        assert L.spangpu_sig_tone_rx_import_state(bank.h, ch, a.p) == 0
        o2, e2 = bank_run(bank, x[cut:])
        assert np.array_equal(np.concatenate([o1, o2]), want_out) and e1 + e2 == want_ev
        bank2 = engine.SigToneRxBank(tone_type, n_ch)
        bank2.set_mode(mode)
        o1, e1 = bank_run(bank2, x[:cut])
        b = ref.SigToneRx(tone_type, 0)
        assert L.spangpu_sig_tone_rx_export_state(bank2.h, ch, b.p) == 0
        o2, e2 = ref_run(b, x[cut:])
        assert np.array_equal(np.concatenate([o1, o2]), want_out) and e1 + e2 == want_ev
        probe = ref.SigToneRx(tone_type, 0)
        assert L.spangpu_sig_tone_rx_export_state(bank.h, ch, probe.p) == 0
        assert np.array_equal(whole.snapshot(), probe.snapshot())
        other = engine.SigToneRxBank(1 if tone_type != 1 else 2, n_ch)
        assert L.spangpu_sig_tone_rx_import_state(other.h, ch, a.p) < 0
    assert moved >= 1


def test_super_tone_call_changes_sides(libs):
    """A super-tone receiver: the Goertzel states, the block's energy and position (device side) and the cadence bookkeeping
    (host side) move between a receiver of the real reference and one made by this library on an equal descriptor."""
    from oracle import ref
    from spandsp_amd import engine
    from test_oracle_pin import build_st_desc, st_signal
    R, L0 = libs
    L = C.CDLL(engine.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    TONE_CB = C.CFUNCTYPE(None, vp, ci, ci, ci)
    SEG_CB = C.CFUNCTYPE(None, vp, ci, ci, ci)
    for name, (res, args) in {"super_tone_rx_make_descriptor": (vp, [vp]), "super_tone_rx_free_descriptor": (ci, [vp]),
                              "super_tone_rx_add_tone": (ci, [vp]), "super_tone_rx_add_element": (ci, [vp, ci, ci, ci, ci, ci]),
                              "super_tone_rx_init": (vp, [vp, vp, TONE_CB, vp]), "super_tone_rx_free": (ci, [vp]),
                              "super_tone_rx_segment_callback": (None, [vp, SEG_CB]), "super_tone_rx": (ci, [vp, vp, ci]),
                              "spangpu_super_tone_rx_import_state": (ci, [vp, vp]),
                              "spangpu_super_tone_rx_export_state": (ci, [vp, vp])}.items():
        getattr(L, name).restype = res
        getattr(L, name).argtypes = args
    desc = L.super_tone_rx_make_descriptor(None)

    class D:
        def add_tone(self):
            return L.super_tone_rx_add_tone(desc)

        def add_element(self, *a):
            return L.super_tone_rx_add_element(desc, *a)
    build_st_desc(D)
    rdesc = build_st_desc(ref.SuperToneDesc)
    x = st_signal()

    def ref_events(rx, seg):
        rx.sink.clear()
        for k in range(0, len(seg), 160):
            rx.rx(seg[k:k + 160])
        return [tuple(int(v) for v in e) for e in rx.sink.events()]

    class Shim:
        def __init__(self):
            self.ev = []
            self.tcb = TONE_CB(lambda ud, code, level, delay: self.ev.append((1, code, level, delay)))
            self.scb = SEG_CB(lambda ud, f1, f2, dur: self.ev.append((4, f1, f2, dur)))
            self.s = L.super_tone_rx_init(None, desc, self.tcb, None)
            assert self.s
            L.super_tone_rx_segment_callback(self.s, self.scb)

        def run(self, seg):
            self.ev = []
            for k in range(0, len(seg), 160):
                fr = np.ascontiguousarray(seg[k:k + 160])
                assert L.super_tone_rx(self.s, fr.ctypes.data, len(fr)) == len(fr)
            return list(self.ev)
    whole = ref.SuperToneRx(rdesc, True)
    want = ref_events(whole, x)
    assert len(want) >= 20
    for cut in (160*61 + 45, 160*300 + 3, 160*520 + 127):
        assert cut + 1000 < len(x)
        a = ref.SuperToneRx(rdesc, True)
        first = ref_events(a, x[:cut])
        sh = Shim()
        assert L.spangpu_super_tone_rx_import_state(sh.s, a.p) == 0
        assert first + sh.run(x[cut:]) == want, cut
        sh2 = Shim()
        first2 = sh2.run(x[:cut])
        b = ref.SuperToneRx(rdesc, True)
        assert L.spangpu_super_tone_rx_export_state(sh2.s, b.p) == 0
        assert first2 + ref_events(b, x[cut:]) == want, cut
        L.super_tone_rx_free(sh.s)
        L.super_tone_rx_free(sh2.s)
    # a receiver on another descriptor (another number of monitored frequencies) is refused
    small = ref.SuperToneDesc()
    t = small.add_tone()
    small.add_element(t, 400, 0, 700, 0)
    small.add_element(t, 620, 0, 700, 0)
    other = ref.SuperToneRx(small, True)
    sh = Shim()
    assert L.spangpu_super_tone_rx_import_state(sh.s, other.p) < 0
    L.super_tone_rx_free(sh.s)
