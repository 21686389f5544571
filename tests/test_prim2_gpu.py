"""Round 6's primitives on the GPU: the batched entry points of csrc/prim2_api.hip (periodogram*, fixed_sqrt32, dds_complexf,
arctan2) and the plain dot products / LMS updates, against the oracle's restatement (oracle/prims_oracle.c, pinned to the
reference in tests/test_oracle_pin.py::test_prims2_live) on tests/prims2.py's cases -- whole-domain sweeps of the three
helpers the receivers use inside their kernels included --, against the committed answers of the real reference
(tests/golden/prims2.npz), and by their spandsp names through libspangpu_prims.so (against the reference's functions of the
same names where oracle/_ref travelled with the snapshot).  Bar: bit-exact (any NaN counting as NaN)."""
import ctypes as C
import os

import numpy as np
import pytest

import prims2
from test_oracle_pin import GOLDEN, use_golden_modem_tables

pytestmark = pytest.mark.gpu


class Batched:
    """the GPU's batched entry points behind tests/prims2.py's interface; table making (coefficient sets, phase offsets) is host
    code in libspangpu_prims.so"""

    def __init__(self):
        from spandsp_amd import engine
        self.e = engine
        self.names = prims2.ByName(C.CDLL(engine.PRIMS_LIB_PATH))

    def periodogram(self, co, amp, n):
        return self.e.periodogram(co, amp)

    def prepare(self, amp, n):
        return self.e.periodogram_prepare(amp)

    def apply(self, co, s, d, n):
        return self.e.periodogram_apply(co, s, d, n)

    def gen_coeffs(self, freq, rate, n):
        return self.names.gen_coeffs(freq, rate, n)

    def gen_phase_offset(self, freq, rate, interval):
        return self.names.gen_phase_offset(freq, rate, interval)

    def freq_error(self, off, scale, last, now):
        return self.e.periodogram_freq_error(off, scale, last, now)

    def vec_dot(self, x, y):
        return self.e.vec_circular_dot_prodf(x, y, np.zeros(len(x), np.int32))

    def vec_lms(self, x, y, err):
        return self.e.vec_circular_lmsf(x, y, np.zeros(len(x), np.int32), err)

    def cvec_dot(self, x, y):
        z = self.e.cvec_circular_dot_prodf(x.view(np.complex64)[..., 0], y.view(np.complex64)[..., 0], np.zeros(len(x), np.int32))
        return z.view(np.float32).reshape(-1, 2)

    def cvec_lms(self, x, y, err):
        z = self.e.cvec_circular_lmsf(x.view(np.complex64)[..., 0], y.view(np.complex64)[..., 0], np.zeros(len(x), np.int32),
                                      err.view(np.complex64)[..., 0])
        return z.view(np.float32).reshape(y.shape)

    def sqrt32(self, x):
        return self.e.fixed_sqrt32(x)

    def dds(self, acc, rate, n):
        return self.e.dds_complexf(acc, rate, n)

    def arctan2(self, y, x):
        return self.e.arctan2(y, x)


def test_batches_match_the_oracle_and_the_committed_reference_answers(built):
    use_golden_modem_tables()
    d = prims2.inputs()
    got = prims2.run(Batched(), d)
    want = prims2.run(prims2.Restated(), d)
    assert set(got) == set(want) and len(got) > 40
    for k in want:
        bad = np.nonzero(got[k].reshape(-1) != want[k].reshape(-1))[0]
        assert bad.size == 0, (k, bad[:8])
    g = np.load(os.path.join(GOLDEN, "prims2.npz"))
    s = prims2.summary(got)
    for k in g.files:
        assert np.array_equal(s[k], g[k]), k


def test_by_name_through_the_opt_in_library(built):
    """libspangpu_prims.so: every name of round 6 on a slice of the cases (a launch per call), against the oracle and -- where
    oracle/_ref is present -- the reference's function of the same name."""
    import oracle
    from spandsp_amd import engine
    use_golden_modem_tables()
    lib = C.CDLL(engine.PRIMS_LIB_PATH)
    by_name = prims2.ByName(lib)
    orc = prims2.Restated()
    ref = None
    if oracle.have_ref():
        from test_oracle_pin import prims2_reference
        ref = prims2_reference()
    d = prims2.inputs()
    k = 6
    for n in (16, 33):
        amp, co = d["amp_%d" % n][:k], d["coeffs_%d" % n][:k]
        for other in [orc] + ([ref] if ref else []):
            assert np.array_equal(prims2.nan_canon(by_name.periodogram(co, amp, n)), prims2.nan_canon(other.periodogram(co, amp, n)))
            s1, d1 = by_name.prepare(amp, n)
            s2, d2 = other.prepare(amp, n)
            assert np.array_equal(prims2.nan_canon(s1), prims2.nan_canon(s2)) and np.array_equal(prims2.nan_canon(d1), prims2.nan_canon(d2))
            assert np.array_equal(prims2.nan_canon(by_name.apply(co, s1, d1, n)), prims2.nan_canon(other.apply(co, s2, d2, n)))
            assert np.array_equal(by_name.gen_coeffs(1100.0, 8000, n).view(np.uint32), other.gen_coeffs(1100.0, 8000, n).view(np.uint32))
    off, scale = by_name.gen_phase_offset(1100.0, 8000, 80)
    for other in [orc] + ([ref] if ref else []):
        assert (off, scale) == other.gen_phase_offset(1100.0, 8000, 80)
        a = by_name.freq_error(np.array(off, np.float32), scale, d["fe_last"][:40], d["fe_now"][:40])
        b = other.freq_error(np.array(off, np.float32), scale, d["fe_last"][:40], d["fe_now"][:40])
        assert np.array_equal(prims2.nan_canon(a), prims2.nan_canon(b))
        for n in (27, 33, 1):
            x, y, cx, cy = d["vx_%d" % n][:k], d["vy_%d" % n][:k], d["cx_%d" % n][:k], d["cy_%d" % n][:k]
            assert np.array_equal(prims2.nan_canon(by_name.vec_dot(x, y)), prims2.nan_canon(other.vec_dot(x, y)))
            assert np.array_equal(prims2.nan_canon(by_name.vec_lms(x, y, d["verr_%d" % n][:k])), prims2.nan_canon(other.vec_lms(x, y, d["verr_%d" % n][:k])))
            assert np.array_equal(prims2.nan_canon(by_name.cvec_dot(cx, cy)), prims2.nan_canon(other.cvec_dot(cx, cy)))
            assert np.array_equal(prims2.nan_canon(by_name.cvec_lms(cx, cy, d["cerr_%d" % n][:k])), prims2.nan_canon(other.cvec_lms(cx, cy, d["cerr_%d" % n][:k])))
        xs = np.ascontiguousarray(d["sqrt_x"][::40000])
        assert np.array_equal(np.array([lib.fixed_sqrt32(int(v)) for v in xs], np.uint16), other.sqrt32(xs))
        ph = d["dds_phase"][::500].copy()
        want, _ = other.dds(ph.copy(), np.zeros(len(ph), np.int32), 1)
        for i, p in enumerate(ph):
            z = lib.dds_lookup_complexf(int(p))
            assert (np.float32(z.re), np.float32(z.im)) == (want[i, 0, 0], want[i, 0, 1])
        acc = C.c_uint32(12345)
        run, end = other.dds(np.array([12345], np.uint32), np.array([-987654321], np.int32), 5)
        for j in range(5):
            z = lib.dds_complexf(C.byref(acc), -987654321)
            assert (np.float32(z.re), np.float32(z.im)) == (run[0, j, 0], run[0, j, 1])
        assert acc.value == int(end[0])
    lib.dds_advancef.argtypes = [C.c_void_p, C.c_int32]
    acc = C.c_uint32(0xFFFFFFF0)
    lib.dds_advancef(C.byref(acc), 0x20)
    assert acc.value == 0x10


def test_bad_arguments_and_positions(built):
    from spandsp_amd import engine
    x = np.ones((3, 8), np.float32)
    with pytest.raises(engine.SpanGpuError):
        engine.vec_circular_dot_prodf(x, x, np.array([0, 9, 0], np.int32))          # a position outside its row, host arrays: refused
    with pytest.raises(engine.SpanGpuError):
        engine.vec_circular_lmsf(x, x, np.array([0, -1, 0], np.int32), np.zeros(3, np.float32))
    with pytest.raises(engine.SpanGpuError):
        engine.periodogram(np.zeros((2, 4, 2), np.float32), np.zeros((0, 8, 2), np.float32))
    L = engine.lib()
    st = np.zeros((4, 8), np.uint32)
    desc = np.zeros((4, 12), np.uint32)
    xs = np.zeros((4, 5), np.float32)
    L.spangpu_godard_ted_rx_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int]
    assert L.spangpu_godard_ted_rx_batch(0, st.ctypes.data, desc.ctypes.data, 5, xs.ctypes.data, 5, 4, 5, 0) == -2      # descriptor stride 1..11
    assert L.spangpu_godard_ted_rx_batch(0, st.ctypes.data, desc.ctypes.data, 12, xs.ctypes.data, -5, 4, 5, 0) == -2    # negative sample stride
    assert L.spangpu_godard_ted_rx_batch(0, st.ctypes.data, desc.ctypes.data, 12, xs.ctypes.data, 5, 4, 5, 0) == 0
