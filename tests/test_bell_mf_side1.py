"""bell_mf_rx() on the seven-test sequence of the reference's own Bell MF receiver test program, exactly as that program
drives it (tests/bell_mf_rx_tests.c:236-560, restated in tests/mf_side1.py).  The golden file holds what the REAL
reference answered to every one of the 13 623 bell_mf_rx() calls (tests/golden/make_golden.py: bell_mf_side1, from
oracle/_ref) and the CRC of every signal it was given.  Here the signals are regenerated with the restated tone
generator and noise source (the CRC proves they are the reference's), and

  * on the CPU the restated oracle must give the reference's answers (one more pin of oracle/tone_oracle.c), and
  * on the GPU the bell_mf_rx() / bell_mf_rx_get() shim over the HIP engine (a private one-channel object) must give
    them too, call for call,

plus the summary figures BASELINE.md section 2 records for the reference: dynamic range -30 ... -3 dBm0, guard time
43 ms, acceptable S/N 7 dB."""
import ctypes as C
import os

import numpy as np
import pytest

import mf_side1

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bell_mf_side1.npz")


def _burst(f1, l1, f2, l2, on_ms, off_ms):
    from oracle import restated as orc
    return orc.ToneGen(orc.tone_desc(f1, l1, f2, l2, on_ms, off_ms, 0, 0, False)).tx(9999)


def _noise(seed, level):
    from oracle import restated as orc
    return orc.Awgn(seed, level)


def _check(run, res):
    g = np.load(GOLDEN)
    assert run.calls == int(g["calls"])
    assert np.uint32(run.crc) == g["signal_crc"], "the regenerated test signals differ from the reference's"
    assert "|".join(run.log) == bytes(g["answers"]).decode("latin1")
    assert res["decode_ok"] and int(g["decode_ok"]) == 1
    for key in ("bandwidth", "twist", "dynamic_rounds", "dynamic_range", "guard_rounds", "snr_levels"):
        assert np.array_equal(res[key], g[key]), key
    assert res["guard_time_ms"] == int(g["guard_time_ms"]) and res["acceptable_snr_db"] == int(g["acceptable_snr_db"])
    # the known answers of BASELINE.md section 2 (the reference's own program, run unmodified in the survey container)
    assert res["dynamic_range"].tolist() == [-30, -3]
    assert res["guard_time_ms"] == 43
    assert res["acceptable_snr_db"] == 7
    # and the pass limits the reference's program applies (bell_mf_rx_tests.c:331,367,400,418,465,506,543)
    k = 0
    for d in range(15):
        for which in (0, 1):
            nplus, nminus = res["bandwidth"][k].tolist()
            k += 1
            rrb = (nplus + nminus)/10.0
            rcfo = (nplus - nminus)/10.0
            assert 3.0 + rcfo + 2.0*100.0*10.0/mf_side1.TONES[d][which] <= rrb < 15.0 + rcfo
    assert (res["twist"] >= 60).all()
    assert res["dynamic_range"][0] <= -22 and res["dynamic_range"][1] + 1 > -3
    assert res["guard_time_ms"] <= 61 and res["acceptable_snr_db"] <= 26


def test_bell_mf_side1_oracle(built):
    from oracle import restated as orc
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()               # the restated tone generator's sine table
    run = mf_side1.Run(_burst, _noise, orc.BellMf(0))
    _check(run, run.run())


class _ShimRx:
    def __init__(self, lib):
        self.L = lib
        self.s = lib.bell_mf_rx_init(None, None, None)
        assert self.s

    def rx(self, amp):
        assert self.L.bell_mf_rx(self.s, amp.ctypes.data, len(amp)) == 0

    def get(self):
        buf = C.create_string_buffer(129)
        n = self.L.bell_mf_rx_get(self.s, buf, 128)
        assert n == len(buf.value)
        return buf.value.decode("latin1")

    def close(self):
        self.L.bell_mf_rx_free(self.s)


@pytest.mark.gpu
def test_bell_mf_side1_bell_mf_rx_shim(built):
    from spandsp_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    lib.bell_mf_rx_init.restype = vp
    lib.bell_mf_rx_init.argtypes = [vp, vp, vp]
    lib.bell_mf_rx.restype = ci
    lib.bell_mf_rx.argtypes = [vp, vp, ci]
    lib.bell_mf_rx_get.restype = C.c_size_t
    lib.bell_mf_rx_get.argtypes = [vp, C.c_char_p, ci]
    lib.bell_mf_rx_free.restype = ci
    lib.bell_mf_rx_free.argtypes = [vp]
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()
    rx = _ShimRx(lib)
    run = mf_side1.Run(_burst, _noise, rx)
    res = run.run()
    rx.close()
    _check(run, res)
