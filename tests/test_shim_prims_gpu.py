"""The receivers' inner primitives under their spandsp names (libspangpu_prims.so, the opt-in library; csrc/shim_prims.c: vec_circular_dot_prodf(), vec_circular_lmsf(),
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
    gpu = bind(C.CDLL(engine.PRIMS_LIB_PATH))
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


class GodardDesc(C.Structure):
    _fields_ = [("low", C.c_float*3), ("high", C.c_float*3), ("mixed", C.c_float), ("coarse_trigger", C.c_float), ("fine_trigger", C.c_float),
                ("coarse_step", C.c_int), ("fine_step", C.c_int)]


class GodardState(C.Structure):
    _fields_ = [("desc", GodardDesc), ("low", C.c_float*2), ("high", C.c_float*2), ("dc", C.c_float*2), ("baud_phase", C.c_float),
                ("total", C.c_int)]


def bind_godard(L):
    L.godard_ted_make_descriptor.restype = C.POINTER(GodardDesc)
    L.godard_ted_make_descriptor.argtypes = [C.c_void_p] + [C.c_float]*6 + [C.c_int, C.c_int]
    L.godard_ted_init.restype = C.POINTER(GodardState)
    L.godard_ted_init.argtypes = [C.c_void_p, C.POINTER(GodardDesc)]
    L.godard_ted_rx.restype = None
    L.godard_ted_rx.argtypes = [C.POINTER(GodardState), C.c_float]
    L.godard_ted_per_baud.restype = C.c_int
    L.godard_ted_per_baud.argtypes = [C.POINTER(GodardState)]
    L.godard_ted_correction.restype = C.c_int
    L.godard_ted_correction.argtypes = [C.POINTER(GodardState)]
    L.godard_ted_free.argtypes = [C.POINTER(GodardState)]
    L.godard_ted_free_descriptor.argtypes = [C.POINTER(GodardDesc)]
    return L


def godard_np(state, desc, samples, baud_at):
    """godard.c:144-220 restated in binary32 (every product and sum rounded by itself): the word-level twin of the struct."""
    f = np.float32
    l0, l1, h0, h1, d0, d1, ph = (f(v) for v in state[:7])
    total = int(state[7])
    lc, hc, mixed, coarse, fine, cstep, fstep = desc
    rets = []
    for k, x in enumerate(samples):
        x = f(x)
        v = f(f(f(l0*lc[0]) + f(l1*lc[1])) + x)
        l1, l0 = l0, v
        v = f(f(f(h0*hc[0]) + f(h1*hc[1])) + x)
        h1, h0 = h0, v
        if k in baud_at:
            v = f(f(f(f(l1*h0)*lc[2]) - f(f(l0*h1)*hc[2])) + f(f(l1*h1)*mixed))
            p = f(v - d1)
            d1, d0 = d0, v
            ph = f(ph - p)
            a = abs(ph)
            corr = 0
            if a > fine:
                corr = cstep if a > coarse else fstep
                if ph < 0:
                    corr = -corr
                total += corr
            rets.append(corr)
    return np.array([l0, l1, h0, h1, d0, d1, ph], np.float32), total, rets


def test_godard_ted_names(built):
    """godard_ted_*() under their spandsp names (csrc/shim_prims.c) against the reference's own (src/godard.c, in oracle/_ref) and
    against a binary32 restatement: descriptor words, every state word after every call, every return value."""
    import oracle
    gpu, ref = libs()
    bind_godard(gpu)
    if ref is not None:
        bind_godard(ref)
    rng = np.random.default_rng(2400)
    par = (8000.0, 2400.0, 1700.0, 0.99, 0.35, 0.05, 15, 1)
    gd = gpu.godard_ted_make_descriptor(None, *par)
    lc = np.array(gd.contents.low[:], np.float32)
    hc = np.array(gd.contents.high[:], np.float32)
    desc = (lc, hc, np.float32(gd.contents.mixed), np.float32(gd.contents.coarse_trigger), np.float32(gd.contents.fine_trigger),
            gd.contents.coarse_step, gd.contents.fine_step)
    if ref is not None:
        rd = ref.godard_ted_make_descriptor(None, *par)
        assert bytes(gd.contents) == bytes(rd.contents)
    gs = gpu.godard_ted_init(None, gd)
    rs = ref.godard_ted_init(None, rd) if ref is not None else None
    samples = (rng.normal(0.0, 0.6, 240)*np.where(np.arange(240) < 120, 1.0, 40.0)).astype(np.float32)
    baud_at = set(int(v) for v in np.cumsum(rng.choice([3, 3, 4], 70)) if v < 240)
    st0 = np.zeros(8, np.float32)
    want_f, want_total, want_rets = godard_np(st0, desc, samples, baud_at)
    rets = []
    for k, x in enumerate(samples):
        gpu.godard_ted_rx(gs, C.c_float(float(x)))
        if rs is not None:
            ref.godard_ted_rx(rs, C.c_float(float(x)))
        if k in baud_at:
            r = gpu.godard_ted_per_baud(gs)
            rets.append(r)
            if rs is not None:
                assert r == ref.godard_ted_per_baud(rs), k
        if rs is not None:
            assert bytes(gs.contents) == bytes(rs.contents), k
    got = np.array(list(gs.contents.low) + list(gs.contents.high) + list(gs.contents.dc) + [gs.contents.baud_phase], np.float32)
    assert same(got, want_f) and gs.contents.total == want_total and rets == want_rets
    assert gpu.godard_ted_correction(gs) == want_total
    assert any(r != 0 for r in rets) and any(abs(r) == 15 for r in rets) and any(abs(r) == 1 for r in rets)
    gpu.godard_ted_free(gs)
    gpu.godard_ted_free_descriptor(gd)
    if ref is not None:
        ref.godard_ted_free(rs)
        ref.godard_ted_free_descriptor(rd)
    assert ref is not None or not oracle.have_ref()


def test_godard_ted_batch(built):
    """spangpu_godard_ted_rx_batch() / _per_baud_batch(): 300 detectors with descriptors and states of their own, rows of samples, against
    the binary32 restatement item by item."""
    from spandsp_amd import engine
    L = C.CDLL(engine.LIB_PATH)
    L.spangpu_godard_ted_rx_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int]
    L.spangpu_godard_ted_per_baud_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(17)
    items, n = 300, 7
    state = np.zeros((items, 8), np.uint32)
    state[:, :7] = rng.normal(0.0, 3.0, (items, 7)).astype(np.float32).view(np.uint32)
    state[:, 7] = rng.integers(-50, 50, items).astype(np.int32).view(np.uint32)
    desc = np.zeros((items, 12), np.uint32)
    desc[:, :9] = rng.normal(0.0, 1.0, (items, 9)).astype(np.float32).view(np.uint32)
    desc[:, 7] = np.abs(rng.normal(6.0, 2.0, items)).astype(np.float32).view(np.uint32)          # coarse trigger
    desc[:, 8] = np.abs(rng.normal(1.0, 0.5, items)).astype(np.float32).view(np.uint32)          # fine trigger
    desc[:, 9] = 15
    desc[:, 10] = 1
    x = rng.normal(0.0, 2.0, (items, n)).astype(np.float32)
    st = state.copy()
    corr = np.zeros(items, np.int32)
    assert L.spangpu_godard_ted_rx_batch(0, st.ctypes.data, desc.ctypes.data, 12, x.ctypes.data, n, items, n, 0) == 0
    assert L.spangpu_godard_ted_per_baud_batch(0, st.ctypes.data, desc.ctypes.data, 12, corr.ctypes.data, items, 0) == 0
    seen = set()
    for i in range(items):
        d = desc[i].view(np.float32)
        dd = (d[0:3], d[3:6], d[6], d[7], d[8], int(desc[i, 9]), int(desc[i, 10]))
        s0 = list(state[i, :7].view(np.float32)) + [int(state[i, 7:8].view(np.int32)[0])]
        wf, wt, wr = godard_np(s0, dd, x[i], {n - 1})
        assert same(st[i, :7].view(np.float32), wf), i
        assert int(st[i, 7:8].view(np.int32)[0]) == wt and corr[i] == wr[0], i
        seen.add(abs(int(corr[i])))
    assert seen == {0, 1, 15}
    # one descriptor shared by all items (stride 0)
    st2 = state.copy()
    assert L.spangpu_godard_ted_rx_batch(0, st2.ctypes.data, desc[:1].ctypes.data, 0, x.ctypes.data, n, items, n, 0) == 0
    d = desc[0].view(np.float32)
    wf, _, _ = godard_np(list(state[5, :7].view(np.float32)) + [0], (d[0:3], d[3:6], d[6], d[7], d[8], 15, 1), x[5], set())
    assert same(st2[5, :7].view(np.float32), wf)
