"""The receivers' inner primitives as batched entry points (csrc/prim_api.hip), bit for bit against
the reference's own vec_circular_dot_prodf() / vec_circular_lmsf() / cvec_circular_dot_prodf() / cvec_circular_lmsf() /
power_meter_update() (oracle/_ref: src/vector_float.c:890-1000, src/complex_vector_float.c:137-219, src/power_meter.c:65-70)
where that build is present, and against a binary32 restatement of their loops in any case -- on the tap counts the receivers
use (27 real, 32 / 33 complex) and on inputs a receiver never makes: denormals, cancellation, infinities, NaNs."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

f32 = np.float32


def nasty(rng, shape):
    """random values with denormals, huge and tiny magnitudes, exact cancellations, a few infinities and NaNs"""
    x = rng.normal(0, 1, shape).astype(np.float32)
    scale = rng.choice([1e-42, 1e-38, 1e-20, 1e-3, 1.0, 1e3, 1e19, 3e38], shape).astype(np.float32)
    x = (x*scale).astype(np.float32)
    flat = x.reshape(-1)
    k = flat.size
    idx = rng.integers(0, k - 1, k//16)
    flat[idx + 1] = -flat[idx]                          # cancellation partners
    flat[rng.integers(0, k, 3)] = np.inf
    flat[rng.integers(0, k, 2)] = np.nan
    flat[rng.integers(0, k, 4)] = -0.0
    return x


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same(a, b):
    """bit-equal, any NaN counting as a NaN (the sign and payload of a NaN made by an invalid operation differ between x86 and
    the GPU and nothing in the reference depends on them)"""
    a = np.ascontiguousarray(a).view(np.float32).reshape(-1)
    b = np.ascontiguousarray(b).view(np.float32).reshape(-1)
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((bits(a) == bits(b)) | nan))


def ref_dot(x, y, pos):
    n = len(x)
    a = f32(0.0)
    for k in range(n - pos):
        a = f32(a + f32(x[pos + k]*y[k]))
    b = f32(0.0)
    for k in range(pos):
        b = f32(b + f32(x[k]*y[n - pos + k]))
    return f32(a + b)


@pytest.mark.parametrize("n", [27, 33, 8])
def test_vec_circular_dot_prodf_and_lmsf(built, n):
    from spandsp_amd import engine
    from oracle import ref
    import oracle
    rng = np.random.default_rng(n)
    items = 700
    with np.errstate(all="ignore"):
        x = nasty(rng, (items, n))
        y = nasty(rng, (items, n))
        x[:200] = rng.normal(0, 1000, (200, n)).astype(np.float32)        # a receiver's kind of values as well
        y[:200] = rng.normal(0, 0.1, (200, n)).astype(np.float32)
        pos = rng.integers(0, n, items).astype(np.int32)
        pos[:3] = [0, n - 1, 1]
        got = engine.vec_circular_dot_prodf(x, y, pos)
        want = np.array([ref_dot(x[i], y[i], int(pos[i])) for i in range(items)], np.float32)
        assert same(got, want)
        # one row of coefficients shared by all items (stride 0), as a pulse shaper is
        got1 = engine.vec_circular_dot_prodf(x, y[5], pos)
        want1 = np.array([ref_dot(x[i], y[5], int(pos[i])) for i in range(items)], np.float32)
        assert same(got1, want1)
        err = nasty(rng, items)
        err[:200] = rng.normal(0, 0.01, 200).astype(np.float32)
        got2 = engine.vec_circular_lmsf(x, y, pos, err)
        want2 = y.copy()
        for i in range(items):
            p = int(pos[i])
            for k in range(n):
                xv = x[i, p + k] if k < n - p else x[i, k - (n - p)]
                want2[i, k] = f32(f32(y[i, k]*f32(0.9999)) + f32(xv*err[i]))
        assert same(got2, want2)
    if oracle.have_ref():
        L = ref.lib()
        zr = np.array([L.vec_circular_dot_prodf(x[i].ctypes.data, y[i].ctypes.data, n, int(pos[i])) for i in range(items)], np.float32)
        assert same(got, zr)
        yr = y.copy()
        for i in range(items):
            L.vec_circular_lmsf(x[i].ctypes.data, yr[i].ctypes.data, n, int(pos[i]), float(err[i]))
        assert same(got2, yr)


@pytest.mark.parametrize("n", [33, 32])
def test_cvec_circular_dot_prodf_and_lmsf(built, n):
    from spandsp_amd import engine
    from oracle import ref
    import oracle
    rng = np.random.default_rng(100 + n)
    items = 500
    with np.errstate(all="ignore"):
        x = (nasty(rng, (items, n)) + 1j*nasty(rng, (items, n))).astype(np.complex64)
        y = (nasty(rng, (items, n)) + 1j*nasty(rng, (items, n))).astype(np.complex64)
        x[:150] = (rng.normal(0, 3, (150, n)) + 1j*rng.normal(0, 3, (150, n))).astype(np.complex64)
        y[:150] = (rng.normal(0, 0.3, (150, n)) + 1j*rng.normal(0, 0.3, (150, n))).astype(np.complex64)
        pos = rng.integers(0, n, items).astype(np.int32)
        pos[:2] = [0, n - 1]
        got = engine.cvec_circular_dot_prodf(x, y, pos)
        want = np.zeros(items, np.complex64)
        for i in range(items):
            p = int(pos[i])
            acc = [[f32(0), f32(0)], [f32(0), f32(0)]]
            for part, (xo, yo, cnt) in enumerate(((p, 0, n - p), (0, n - p, p))):
                for k in range(cnt):
                    a, b = x[i, xo + k], y[i, yo + k]
                    acc[part][0] = f32(acc[part][0] + f32(f32(a.real*b.real) - f32(a.imag*b.imag)))
                    acc[part][1] = f32(acc[part][1] + f32(f32(a.real*b.imag) + f32(a.imag*b.real)))
            want[i] = complex(f32(acc[0][0] + acc[1][0]), f32(acc[0][1] + acc[1][1]))
        assert same(got.view(np.float32), want.view(np.float32))
        err = (nasty(rng, items) + 1j*nasty(rng, items)).astype(np.complex64)
        err[:150] = (rng.normal(0, 0.01, 150) + 1j*rng.normal(0, 0.01, 150)).astype(np.complex64)
        got2 = engine.cvec_circular_lmsf(x, y, pos, err)
        want2 = y.copy()
        for i in range(items):
            p = int(pos[i])
            e = err[i]
            for k in range(n):
                xv = x[i, p + k] if k < n - p else x[i, k - (n - p)]
                re = f32(f32(y[i, k].real*f32(0.9999)) + f32(f32(xv.imag*e.imag) + f32(xv.real*e.real)))
                im = f32(f32(y[i, k].imag*f32(0.9999)) + f32(f32(xv.real*e.imag) - f32(xv.imag*e.real)))
                want2[i, k] = complex(re, im)
        assert same(got2.view(np.float32), want2.view(np.float32))
    if oracle.have_ref():
        L = ref.lib()
        zr = np.zeros(items, np.complex64)
        yr = y.copy()
        for i in range(items):
            r = L.cvec_circular_dot_prodf(x[i].ctypes.data, y[i].ctypes.data, n, int(pos[i]))
            zr[i] = complex(r.re, r.im)
            e = (C.c_float*2)(float(err[i].real), float(err[i].imag))
            L.cvec_circular_lmsf(x[i].ctypes.data, yr[i].ctypes.data, n, int(pos[i]), C.addressof(e))
        assert same(got.view(np.float32), zr.view(np.float32))
        assert same(got2.view(np.float32), yr.view(np.float32))


def test_power_meter_update(built):
    from spandsp_amd import engine
    from oracle import ref
    import oracle
    rng = np.random.default_rng(9)
    items, n = 300, 160
    amp = rng.integers(-32768, 32768, (items, n)).astype(np.int16)
    amp[:50] = (rng.normal(0, 300, (50, n))).astype(np.int16)
    reading = rng.integers(0, 1 << 30, items).astype(np.int32)
    shift = rng.integers(1, 12, items).astype(np.int32)
    got = engine.power_meter_update(amp, reading, shift)
    want = reading.copy()
    for i in range(items):
        r = int(want[i])
        for k in range(n):
            a = int(amp[i, k])
            r = r + ((a*a - r) >> int(shift[i]))
            r = (r + 2**31) % 2**32 - 2**31
        want[i] = r
    assert np.array_equal(got, want)
    if oracle.have_ref():
        L = ref.lib()
        for i in range(0, items, 7):
            st = (C.c_int32*2)(int(shift[i]), int(reading[i]))              # power_meter_t: {int shift; int32_t reading;}
            r = 0
            for k in range(n):
                r = L.power_meter_update(C.addressof(st), int(amp[i, k]))
            assert r == got[i], i
