"""The receivers' inner primitives under their spandsp names (csrc/shim_prims.c: vec_circular_dot_prodf(), vec_circular_lmsf(),
cvec_circular_dot_prodf(), cvec_circular_lmsf(), power_meter_*()), as a caller that links them by name finds them: against the
real reference's functions of the same names (oracle/_ref, which travels with the snapshot) where it is present, and against
the batched entry points they run through (held to the reference in test_prim_gpu.py) in any case."""
import ctypes as C

import numpy as np
import pytest

from test_prim_gpu import nasty, same

pytestmark = pytest.mark.gpu


class Complexf(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class PowerMeter(C.Structure):
    _fields_ = [("shift", C.c_int), ("reading", C.c_int32)]


def bind(L):
    fp = C.POINTER(C.c_float)
    L.vec_circular_dot_prodf.restype = C.c_float
    L.vec_circular_dot_prodf.argtypes = [fp, fp, C.c_int, C.c_int]
    L.vec_circular_lmsf.restype = None
    L.vec_circular_lmsf.argtypes = [fp, fp, C.c_int, C.c_int, C.c_float]
    L.cvec_circular_dot_prodf.restype = Complexf
    L.cvec_circular_dot_prodf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.cvec_circular_lmsf.restype = None
    L.cvec_circular_lmsf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Complexf)]
    L.power_meter_init.restype = C.POINTER(PowerMeter)
    L.power_meter_init.argtypes = [C.c_void_p, C.c_int]
    L.power_meter_update.restype = C.c_int32
    L.power_meter_update.argtypes = [C.POINTER(PowerMeter), C.c_int16]
    L.power_meter_rx.restype = C.c_int32
    L.power_meter_rx.argtypes = [C.POINTER(PowerMeter), C.c_void_p, C.c_int]
    L.power_meter_current.restype = C.c_int32
    L.power_meter_current.argtypes = [C.POINTER(PowerMeter)]
    L.power_meter_free.argtypes = [C.POINTER(PowerMeter)]
    return L


def libs():
    import oracle
    from spandsp_amd import engine
    gpu = bind(C.CDLL(engine.LIB_PATH))
    ref = None
    if oracle.have_ref():
        from oracle import ref as r
        ref = bind(C.CDLL(r.REF_SO)) if hasattr(r, "REF_SO") else None
    return gpu, ref


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_the_spandsp_named_primitives(built):
    gpu, ref = libs()
    rng = np.random.default_rng(99)
    checked = 0
    for n in (27, 33, 8):
        for pos in (0, 1, n//2, n - 1):
            x = nasty(rng, (n,))
            y = nasty(rng, (n,))
            z = np.float32(gpu.vec_circular_dot_prodf(fptr(x), fptr(y), n, pos))
            if ref is not None:
                assert same(np.array([z]), np.array([np.float32(ref.vec_circular_dot_prodf(fptr(x), fptr(y), n, pos))])), (n, pos)
                checked += 1
            y1, y2 = y.copy(), y.copy()
            gpu.vec_circular_lmsf(fptr(x), fptr(y1), n, pos, C.c_float(0.0123))
            if ref is not None:
                ref.vec_circular_lmsf(fptr(x), fptr(y2), n, pos, C.c_float(0.0123))
                assert same(y1, y2), (n, pos)
            assert not same(y1, y) or n == 0
            cx = nasty(rng, (n, 2))
            cy = nasty(rng, (n, 2))
            cz = gpu.cvec_circular_dot_prodf(cx.ctypes.data, cy.ctypes.data, n, pos)
            if ref is not None:
                rz = ref.cvec_circular_dot_prodf(cx.ctypes.data, cy.ctypes.data, n, pos)
                assert same(np.array([cz.re, cz.im], np.float32), np.array([rz.re, rz.im], np.float32)), (n, pos)
            err = Complexf(0.25, -0.0625)
            c1, c2 = cy.copy(), cy.copy()
            gpu.cvec_circular_lmsf(cx.ctypes.data, c1.ctypes.data, n, pos, C.byref(err))
            if ref is not None:
                ref.cvec_circular_lmsf(cx.ctypes.data, c2.ctypes.data, n, pos, C.byref(err))
                assert same(c1, c2), (n, pos)
    amp = rng.integers(-32768, 32768, 400).astype(np.int16)
    for shift in (3, 5, 8):
        pm = gpu.power_meter_init(None, shift)
        want = 0
        for a in amp[:50]:
            got = gpu.power_meter_update(pm, int(a))
            want = (want + ((int(a)*int(a) - want) >> shift))
            want = (want + 2**31) % 2**32 - 2**31
            assert got == want and gpu.power_meter_current(pm) == want
        assert gpu.power_meter_rx(pm, amp[50:].ctypes.data, 350) == 0
        for a in amp[50:]:
            want = (want + ((int(a)*int(a) - want) >> shift))
            want = (want + 2**31) % 2**32 - 2**31
        assert gpu.power_meter_current(pm) == want
        if ref is not None:
            rp = ref.power_meter_init(None, shift)
            for a in amp[:50]:
                ref.power_meter_update(rp, int(a))
            ref.power_meter_rx(rp, amp[50:].ctypes.data, 350)
            assert ref.power_meter_current(rp) == want
            ref.power_meter_free(rp)
        gpu.power_meter_free(pm)
    import oracle
    assert checked > 0 or not oracle.have_ref()
