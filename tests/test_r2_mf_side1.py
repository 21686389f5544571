"""r2_mf_rx() on the seven-test sequence of the reference's own MFC/R2 receiver test program, forward and backward tone
sets, exactly as that program drives it (tests/r2_mf_rx_tests.c:246-587, restated in tests/mf_side1.py: R2Run).  The
golden files hold what the REAL reference answered to every r2_mf_rx() call (83 k forward, 76 k backward;
tests/golden/make_golden.py: r2_mf_side1, from oracle/_ref) and the CRC of every signal it was given.  The signals are
regenerated with the restated tone generator and noise source (the CRC proves they are the reference's); the restated
oracle (CPU) and the r2_mf_rx() / r2_mf_rx_get() shim over the HIP engine (GPU) must give the reference's answers call
for call, and the summary figures of BASELINE.md section 2: dynamic range -36 ... -3 dBm0, guard time 33 ms, acceptable
S/N 4 dB."""
import ctypes as C
import os

import numpy as np
import pytest

import mf_side1

HERE = os.path.dirname(os.path.abspath(__file__))


def _burst(f1, l1, f2, l2, on_ms, off_ms):
    from oracle import restated as orc
    return orc.ToneGen(orc.tone_desc(f1, l1, f2, l2, on_ms, off_ms, 0, 0, False)).tx(9999)


def _noise(seed, level):
    from oracle import restated as orc
    return orc.Awgn(seed, level)


def _check(run, res, fwd):
    g = np.load(os.path.join(HERE, "golden", "r2_mf_side1_%s.npz" % ("fwd" if fwd else "back")))
    assert run.calls == int(g["calls"])
    assert np.uint32(run.crc) == g["signal_crc"], "the regenerated test signals differ from the reference's"
    assert np.array_equal(np.array(run.log, np.uint8), g["answers"])
    assert res["decode_ok"] and int(g["decode_ok"]) == 1
    for key in ("bandwidth", "twist", "dynamic_rounds", "dynamic_range", "guard_rounds", "snr_levels"):
        assert np.array_equal(res[key], g[key]), key
    assert res["guard_time_ms"] == int(g["guard_time_ms"]) and res["acceptable_snr_db"] == int(g["acceptable_snr_db"])
    # BASELINE.md section 2, and the pass limits of the reference's program (r2_mf_rx_tests.c:418,437,493,535,580)
    assert res["dynamic_range"].tolist() == [-36, -3]
    assert res["guard_time_ms"] == 33
    assert res["acceptable_snr_db"] == 4
    assert (res["twist"] >= 70).all()


class _OracleRx:
    def __init__(self, fwd):
        from oracle import restated as orc
        self.d = orc.R2Mf(fwd, use_callback=False)

    def rx(self, amp):
        self.d.rx(amp)

    def get(self):
        return self.d.snapshot()["current_digit"]


@pytest.mark.parametrize("fwd", [True, False], ids=["forward", "backward"])
def test_r2_mf_side1_oracle(built, fwd):
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()
    run = mf_side1.R2Run(_burst, _noise, _OracleRx(fwd), fwd)
    _check(run, run.run(), fwd)


class _ShimRx:
    def __init__(self, lib, fwd):
        self.L = lib
        self.s = lib.r2_mf_rx_init(None, fwd, None, None)
        assert self.s

    def rx(self, amp):
        self.L.r2_mf_rx(self.s, amp.ctypes.data, len(amp))

    def get(self):
        return self.L.r2_mf_rx_get(self.s)

    def close(self):
        self.L.r2_mf_rx_free(self.s)


@pytest.mark.gpu
@pytest.mark.parametrize("fwd", [True, False], ids=["forward", "backward"])
def test_r2_mf_side1_r2_mf_rx_shim(built, fwd):
    from spandsp_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    lib.r2_mf_rx_init.restype = vp
    lib.r2_mf_rx_init.argtypes = [vp, C.c_bool, vp, vp]
    lib.r2_mf_rx.restype = ci
    lib.r2_mf_rx.argtypes = [vp, vp, ci]
    lib.r2_mf_rx_get.restype = ci
    lib.r2_mf_rx_get.argtypes = [vp]
    lib.r2_mf_rx_free.restype = ci
    lib.r2_mf_rx_free.argtypes = [vp]
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()
    rx = _ShimRx(lib, fwd)
    run = mf_side1.R2Run(_burst, _noise, rx, fwd)
    res = run.run()
    rx.close()
    _check(run, res, fwd)
