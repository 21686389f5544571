"""BASELINE configs[0]: dtmf_rx() on the Mitel CM7291 side-1 test sequence, exactly as the reference's own test program
drives it (tests/dtmf_rx_tests.c:357-655, restated in tests/mitel.py).  The golden file holds what the REAL reference
answered to every one of the 4173 dtmf_rx() calls (tests/golden/make_golden.py: mitel_side1, from oracle/_ref) and the
CRC of every signal it was given.  Here the signals are regenerated with the restated tone generator and noise source
(the CRC proves they are the reference's), and

  * on the CPU the restated oracle must give the reference's answers (one more pin of oracle/tone_oracle.c), and
  * on the GPU the dtmf_rx() / dtmf_rx_get() shim over the HIP engine (a private one-channel object, the plumbing
    configuration) must give them too, call for call,

plus the summary figures BASELINE.md section 2 records for the reference: twist 8.0 / 8.5 / 8.3 / 8.6 dB (reverse
4.0 / 4.4 / 4.5 / 4.6), dynamic range 39 dB, guard time 27 ms, acceptable S/N 9 dB."""
import ctypes as C
import os

import numpy as np
import pytest

import mitel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mitel_side1.npz")


def _burst(f1, l1, f2, l2, on_ms, off_ms):
    from oracle import restated as orc
    return orc.ToneGen(orc.tone_desc(f1, l1, f2, l2, on_ms, off_ms, 0, 0, False)).tx(1000)


def _noise(seed, level):
    from oracle import restated as orc
    return orc.Awgn(seed, level)


def _check(run, res):
    g = np.load(GOLDEN)
    assert run.calls == int(g["calls"])
    assert np.uint32(run.crc) == g["signal_crc"], "the regenerated test signals differ from the reference's"
    assert "|".join(run.log) == bytes(g["answers"]).decode("latin1")
    assert res["decode_ok"] and int(g["decode_ok"]) == 1
    assert np.array_equal(res["bandwidth"], g["bandwidth"])
    assert np.array_equal(res["twist"], g["twist"])
    assert res["dynamic_range"] == int(g["dynamic_range"])
    assert res["guard_time_ms"] == int(g["guard_time_ms"]) and res["guard_responses"] == int(g["guard_responses"])
    assert np.array_equal(res["snr_levels"], g["snr_levels"])
    assert res["acceptable_snr_db"] == int(g["acceptable_snr_db"])
    # the known answers of BASELINE.md section 2 (the reference's own program, run unmodified in the survey container)
    assert [t[0] for t in res["twist"].tolist()] == [80, 85, 83, 86]
    assert [t[1] for t in res["twist"].tolist()] == [40, 44, 45, 46]
    assert res["dynamic_range"] == 39
    assert res["guard_time_ms"] == 27
    assert res["acceptable_snr_db"] == 9
    # and the pass limits the reference's program applies (dtmf_rx_tests.c:463,529,546,579,650)
    for nplus, nminus in res["bandwidth"].tolist():
        rrb = (nplus + nminus)/10.0
        rcfo = (nplus - nminus)/10.0
        assert 3.0 + rcfo <= rrb < 15.0 + rcfo


def test_mitel_side1_oracle(built):
    from oracle import restated as orc
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()               # the restated tone generator's sine table
    run = mitel.Run(_burst, _noise, orc.Dtmf(0))
    _check(run, run.run())


class _ShimRx:
    def __init__(self, lib):
        self.L = lib
        self.s = lib.dtmf_rx_init(None, None, None)
        assert self.s

    def rx(self, amp):
        assert self.L.dtmf_rx(self.s, amp.ctypes.data, len(amp)) == 0

    def get(self):
        buf = C.create_string_buffer(129)
        n = self.L.dtmf_rx_get(self.s, buf, 128)
        assert n == len(buf.value)
        return buf.value.decode("latin1")

    def close(self):
        self.L.dtmf_rx_free(self.s)


@pytest.mark.gpu
def test_mitel_side1_dtmf_rx_shim(built):
    from spandsp_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    lib.dtmf_rx_init.restype = vp
    lib.dtmf_rx_init.argtypes = [vp, vp, vp]
    lib.dtmf_rx.restype = ci
    lib.dtmf_rx.argtypes = [vp, vp, ci]
    lib.dtmf_rx_get.restype = C.c_size_t
    lib.dtmf_rx_get.argtypes = [vp, C.c_char_p, ci]
    lib.dtmf_rx_free.restype = ci
    lib.dtmf_rx_free.argtypes = [vp]
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()
    rx = _ShimRx(lib)
    run = mitel.Run(_burst, _noise, rx)
    res = run.run()
    rx.close()
    _check(run, res)


# ---- dial_tone_tolerance_tests() of the same program (dtmf_rx_tests.c:744-800), dial tone filter off and on --------
DIAL_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dtmf_dial_tone.npz")


class _Dial:
    def __init__(self, level):
        from oracle import restated as orc
        self.g = orc.ToneGen(orc.tone_desc(350, level, 440, level, 1, 0, 0, 0, True))

    def gen(self, n):
        return self.g.tx(n)


def _check_dial(run, res, filt):
    g = np.load(DIAL_GOLDEN)
    k = int(filt)
    assert run.calls == int(g["calls_%d" % k])
    assert np.uint32(run.crc) == g["signal_crc_%d" % k], "the regenerated test signals differ from the reference's"
    assert "|".join(run.log) == bytes(g["answers_%d" % k]).decode("latin1")
    assert np.array_equal(res["rounds"], g["rounds_%d" % k])
    assert res["signal_to_dial_tone_db"] == int(g["ratio_%d" % k])
    # the reference's answers, and its pass limits (dtmf_rx_tests.c:790-791)
    assert res["signal_to_dial_tone_db"] == (-12 if filt else 9)
    assert not (res["signal_to_dial_tone_db"] > (-12 if filt else 10))


@pytest.mark.parametrize("filt", [False, True], ids=["filter-off", "filter-on"])
def test_dial_tone_tolerance_oracle(built, filt):
    from oracle import restated as orc
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()
    run = mitel.DialToneRun(_burst, _Dial, orc.Dtmf(0), filt)
    _check_dial(run, run.run(), filt)


@pytest.mark.gpu
@pytest.mark.parametrize("filt", [False, True], ids=["filter-off", "filter-on"])
def test_dial_tone_tolerance_dtmf_rx_shim(built, filt):
    from spandsp_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    lib.dtmf_rx_init.restype = vp
    lib.dtmf_rx_init.argtypes = [vp, vp, vp]
    lib.dtmf_rx.restype = ci
    lib.dtmf_rx.argtypes = [vp, vp, ci]
    lib.dtmf_rx_get.restype = C.c_size_t
    lib.dtmf_rx_get.argtypes = [vp, C.c_char_p, ci]
    lib.dtmf_rx_parms.restype = None
    lib.dtmf_rx_parms.argtypes = [vp, ci, C.c_float, C.c_float, C.c_float]
    lib.dtmf_rx_free.restype = ci
    lib.dtmf_rx_free.argtypes = [vp]
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()
    rx = _ShimRx(lib)
    rx.parms = lambda f, t, r, th: lib.dtmf_rx_parms(rx.s, f, t, r, th)
    run = mitel.DialToneRun(_burst, _Dial, rx, filt)
    res = run.run()
    rx.close()
    _check_dial(run, res, filt)


# ---- callback_function_tests() of the same program (dtmf_rx_tests.c:805-893): digits and realtime callbacks --------
CB_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dtmf_callbacks.npz")


def _flatten(log):
    rows = []
    text = []
    for entry in log:
        at, (ev, t) = entry if isinstance(entry[0], int) else (-1, entry)
        rows.append((9, at, len(ev), len(t)))
        rows.extend(ev)
        text.append(t)
    return np.array(rows, np.int32).reshape(-1, 4), np.frombuffer("".join(text).encode("latin1"), np.uint8)


def _check_callbacks(run, log):
    g = np.load(CB_GOLDEN)
    assert np.uint32(run.crc) == g["signal_crc"], "the regenerated test signals differ from the reference's"
    rows, text = _flatten(log)
    assert np.array_equal(rows, g["rows"]) and np.array_equal(text, g["text"])
    # what the reference's program checks (:219-318): the digits arrive in order, round after round; the realtime reports
    # alternate digit / off with the sender's level (-10 dBm0 per tone) to within 1 dB
    assert bytes(text).decode("latin1") == mitel.POSITIONS*45
    reports = [e for entry in log if isinstance(entry[0], int) for e in entry[1][0]]
    assert len(reports) == 2*16*45
    assert all(r[1] == (ord(mitel.POSITIONS[(k//2) % 16]) if k % 2 == 0 else 0) for k, r in enumerate(reports))
    assert all(-11 <= r[2] <= -9 for r in reports[::2])


def test_dtmf_callback_modes_oracle(built):
    from oracle import restated as orc
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()

    class Rx:
        def __init__(self, mode):
            self.d = orc.Dtmf(mode)
            self.n = 0
            self.t = 0

        def rx(self, amp):
            self.d.rx(amp)

        def drain(self):
            ev = self.d.sink.events()
            txt = self.d.sink.text()
            new = [tuple(int(x) for x in e) for e in ev[self.n:]]
            t = txt[self.t:]
            self.n = len(ev)
            self.t = len(txt)
            return new, t
    run = mitel.CallbackRun(_burst, Rx)
    _check_callbacks(run, run.run())


@pytest.mark.gpu
def test_dtmf_callback_modes_dtmf_rx_shim(built):
    """The digits callback inside single dtmf_rx() calls of up to 115 200 samples, and the realtime callback over
    160-sample chunks, on private shim objects."""
    from spandsp_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    DIGITS_CB = C.CFUNCTYPE(None, vp, C.c_char_p, ci)
    TONE_CB = C.CFUNCTYPE(None, vp, ci, ci, ci)
    lib.dtmf_rx_init.restype = vp
    lib.dtmf_rx_init.argtypes = [vp, DIGITS_CB, vp]
    lib.dtmf_rx_set_realtime_callback.restype = None
    lib.dtmf_rx_set_realtime_callback.argtypes = [vp, TONE_CB, vp]
    lib.dtmf_rx.restype = ci
    lib.dtmf_rx.argtypes = [vp, vp, ci]
    lib.dtmf_rx_free.restype = ci
    lib.dtmf_rx_free.argtypes = [vp]
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()
    made = []

    class Rx:
        def __init__(self, mode):
            self.ev = []
            self.text = ""
            self.dcb = DIGITS_CB(self._digits)
            self.tcb = TONE_CB(lambda ud, code, level, delay: self.ev.append((1, code, level, delay)))
            if mode == 1:
                self.s = lib.dtmf_rx_init(None, self.dcb, None)
            else:
                self.s = lib.dtmf_rx_init(None, C.cast(None, DIGITS_CB), None)
                lib.dtmf_rx_set_realtime_callback(self.s, self.tcb, None)
            assert self.s
            made.append(self)

        def _digits(self, ud, digits, n):
            self.text += digits[:n].decode("latin1")
            self.ev.append((2, n, 0, 0))

        def rx(self, amp):
            amp = np.ascontiguousarray(amp)
            assert lib.dtmf_rx(self.s, amp.ctypes.data, len(amp)) == 0

        def drain(self):
            out = (self.ev, self.text)
            self.ev = []
            self.text = ""
            return out
    run = mitel.CallbackRun(_burst, Rx)
    log = run.run()
    for r in made:
        lib.dtmf_rx_free(r.s)
    _check_callbacks(run, log)
