"""Round 6's primitives (periodogram*, the plain dot products / LMS updates, fixed_sqrt32, dds_complexf, arctan2): one set of
deterministic inputs and three ways to compute the answers -- the real reference (oracle/_ref), the oracle's restatement
(oracle/prims_oracle.c) and the GPU's batched entry points (spandsp_amd/csrc/prim2_api.hip) -- so that the pins
(tests/test_oracle_pin.py), the golden file (tests/golden/prims2.npz, tests/golden/make_golden.py) and the GPU tests
(tests/test_prim2_gpu.py) all talk about the same cases.  Test infrastructure."""
import ctypes as C
import zlib

import numpy as np

PERIODOGRAM_LENS = (16, 102, 160, 33)
N_ITEMS = 48


def nasty_f32(rng, shape, scale=1000.0):
    """normals with a sprinkling of zeros, denormals, huge values and exact cancellations"""
    x = (rng.normal(0.0, scale, shape)).astype(np.float32)
    flat = x.reshape(-1)
    k = max(1, flat.size//40)
    for v in (0.0, -0.0, 1.0e-42, -3.0e-39, 3.0e38, -3.0e38, 1.0, -1.0):
        flat[rng.integers(0, flat.size, k)] = np.float32(v)
    return x


def inputs():
    rng = np.random.default_rng(0x9E71)
    d = {}
    for n in PERIODOGRAM_LENS:
        d["amp_%d" % n] = nasty_f32(rng, (N_ITEMS, n, 2))
        d["coeffs_%d" % n] = nasty_f32(rng, (N_ITEMS, n//2, 2), 0.05)
        d["freq_%d" % n] = rng.uniform(300.0, 3400.0, N_ITEMS).astype(np.float32)
    d["fe_last"] = nasty_f32(rng, (4096, 2))
    d["fe_now"] = nasty_f32(rng, (4096, 2))
    for n in (27, 33, 8, 1):
        d["vx_%d" % n] = nasty_f32(rng, (N_ITEMS, n), 30.0)
        d["vy_%d" % n] = nasty_f32(rng, (N_ITEMS, n), 30.0)
        d["cx_%d" % n] = nasty_f32(rng, (N_ITEMS, n, 2), 30.0)
        d["cy_%d" % n] = nasty_f32(rng, (N_ITEMS, n, 2), 30.0)
        d["verr_%d" % n] = rng.normal(0.0, 0.05, N_ITEMS).astype(np.float32)
        d["cerr_%d" % n] = rng.normal(0.0, 0.05, (N_ITEMS, 2)).astype(np.float32)
    m = np.arange(65536, dtype=np.uint64)
    sq = np.concatenate([(m << s) & 0xFFFFFFFF for s in range(17)] + [rng.integers(0, 1 << 32, 1 << 20, dtype=np.uint64),
                         np.array([0, 1, 2, 3, 4, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0x40000000, 0x3FFFFFFF], np.uint64)])
    d["sqrt_x"] = sq.astype(np.uint32)
    k = np.arange(2048, dtype=np.uint64) << 21
    d["dds_phase"] = (np.concatenate([k - 1, k, k + 1, k + (1 << 20), rng.integers(0, 1 << 32, 4096, dtype=np.uint64)]) & 0xFFFFFFFF).astype(np.uint32)
    d["dds_acc"] = rng.integers(0, 1 << 32, 256, dtype=np.uint64).astype(np.uint32)
    d["dds_rate"] = rng.integers(-(1 << 31), 1 << 31, 256).astype(np.int32)
    v = np.array([0.0, -0.0, 1.4e-45, -1.4e-45, 1e-38, -1e-38, 1e-20, -1e-20, 0.5, -0.5, 1.0, -1.0, 3.0, -3.0, 1e20, -1e20,
                  3.4e38, -3.4e38, np.inf, -np.inf, np.nan], np.float32)
    yy, xx = np.meshgrid(v, v, indexing="ij")
    ry = (rng.normal(0.0, 1.0, 1 << 20)*10.0**rng.uniform(-6.0, 6.0, 1 << 20)).astype(np.float32)
    rx = (rng.normal(0.0, 1.0, 1 << 20)*10.0**rng.uniform(-6.0, 6.0, 1 << 20)).astype(np.float32)
    d["atan_y"] = np.concatenate([yy.reshape(-1), ry]).astype(np.float32)
    d["atan_x"] = np.concatenate([xx.reshape(-1), rx]).astype(np.float32)
    return d


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def nan_canon(a):
    """float arrays as bit patterns with every NaN made the same one (an invalid operation's NaN differs between x86 and the GPU)"""
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b


def run(impl, d=None):
    """every answer, as bit patterns / integers"""
    d = d or inputs()
    out = {}
    for n in PERIODOGRAM_LENS:
        amp, co = d["amp_%d" % n], d["coeffs_%d" % n]
        out["pg_%d" % n] = nan_canon(impl.periodogram(co, amp, n))
        s, df = impl.prepare(amp, n)
        out["pg_sum_%d" % n] = nan_canon(s)
        out["pg_diff_%d" % n] = nan_canon(df)
        out["pg_apply_%d" % n] = nan_canon(impl.apply(co, s, df, n))
        gen = np.stack([impl.gen_coeffs(float(f), 8000, n) for f in d["freq_%d" % n][:8]])
        out["pg_gen_%d" % n] = nan_canon(gen)
        out["pg_matched_%d" % n] = nan_canon(impl.periodogram(np.repeat(gen[:1], N_ITEMS, 0), amp, n))
    off, scale = impl.gen_phase_offset(1100.0, 8000, 80)
    out["fe_offset"] = nan_canon(np.array(list(off) + [scale], np.float32))
    out["fe"] = nan_canon(impl.freq_error(np.array(off, np.float32), scale, d["fe_last"], d["fe_now"]))
    for n in (27, 33, 8, 1):
        out["vdot_%d" % n] = nan_canon(impl.vec_dot(d["vx_%d" % n], d["vy_%d" % n]))
        out["vlms_%d" % n] = nan_canon(impl.vec_lms(d["vx_%d" % n], d["vy_%d" % n], d["verr_%d" % n]))
        out["cdot_%d" % n] = nan_canon(impl.cvec_dot(d["cx_%d" % n], d["cy_%d" % n]))
        out["clms_%d" % n] = nan_canon(impl.cvec_lms(d["cx_%d" % n], d["cy_%d" % n], d["cerr_%d" % n]))
    out["sqrt"] = impl.sqrt32(d["sqrt_x"]).astype(np.uint16)
    z, acc = impl.dds(d["dds_phase"].copy(), np.zeros(len(d["dds_phase"]), np.int32), 1)
    assert np.array_equal(acc, d["dds_phase"])
    out["dds_lookup"] = nan_canon(z)
    z, acc = impl.dds(d["dds_acc"].copy(), d["dds_rate"], 40)
    out["dds_run"] = nan_canon(z)
    out["dds_acc"] = acc.astype(np.uint32)
    out["atan"] = impl.arctan2(d["atan_y"], d["atan_x"]).astype(np.int32)
    return out


def summary(out):
    """what the golden file keeps: a CRC-32 per answer and the small ones whole"""
    g = {"crc_" + k: np.uint32(crc(v)) for k, v in out.items()}
    for k in ("fe_offset", "pg_16", "pg_gen_16", "vdot_27", "cdot_33"):
        g[k] = out[k]
    return g


def fp(a):
    return a.ctypes.data_as(C.c_void_p)


class Complexf(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class ByName:
    """An implementation that exports the reference's own names (the reference build itself, or libspangpu_prims.so)."""

    def __init__(self, L, helpers=None):
        self.L = L
        self.H = helpers or L
        L.periodogram.restype = Complexf
        L.periodogram.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.periodogram_prepare.restype = C.c_int
        L.periodogram_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.periodogram_apply.restype = Complexf
        L.periodogram_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.periodogram_generate_coeffs.restype = C.c_int
        L.periodogram_generate_coeffs.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
        L.periodogram_generate_phase_offset.restype = C.c_float
        L.periodogram_generate_phase_offset.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
        L.periodogram_freq_error.restype = C.c_float
        L.periodogram_freq_error.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.vec_dot_prodf.restype = C.c_float
        L.vec_dot_prodf.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vec_lmsf.restype = None
        L.vec_lmsf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.cvec_dot_prodf.restype = Complexf
        L.cvec_dot_prodf.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.cvec_lmsf.restype = None
        L.cvec_lmsf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.fixed_sqrt32.restype = C.c_uint16
        L.fixed_sqrt32.argtypes = [C.c_uint32]
        L.dds_complexf.restype = Complexf
        L.dds_complexf.argtypes = [C.c_void_p, C.c_int32]
        L.dds_lookup_complexf.restype = Complexf
        L.dds_lookup_complexf.argtypes = [C.c_uint32]

    def periodogram(self, co, amp, n):
        out = np.zeros((len(amp), 2), np.float32)
        for i in range(len(amp)):
            c, a = np.ascontiguousarray(co[i]), np.ascontiguousarray(amp[i])
            z = self.L.periodogram(fp(c), fp(a), n)
            out[i] = (z.re, z.im)
        return out

    def prepare(self, amp, n):
        s = np.zeros((len(amp), n//2, 2), np.float32)
        d = np.zeros_like(s)
        for i in range(len(amp)):
            a = np.ascontiguousarray(amp[i])
            si, di = s[i].copy(), d[i].copy()
            assert self.L.periodogram_prepare(fp(si), fp(di), fp(a), n) == n//2
            s[i], d[i] = si, di
        return s, d

    def apply(self, co, s, d, n):
        out = np.zeros((len(s), 2), np.float32)
        for i in range(len(s)):
            c, si, di = np.ascontiguousarray(co[i]), np.ascontiguousarray(s[i]), np.ascontiguousarray(d[i])
            z = self.L.periodogram_apply(fp(c), fp(si), fp(di), n)
            out[i] = (z.re, z.im)
        return out

    def gen_coeffs(self, freq, rate, n):
        c = np.zeros((n//2, 2), np.float32)
        assert self.L.periodogram_generate_coeffs(fp(c), freq, rate, n) == n//2
        return c

    def gen_phase_offset(self, freq, rate, interval):
        off = np.zeros(2, np.float32)
        scale = self.L.periodogram_generate_phase_offset(fp(off), freq, rate, interval)
        return (float(off[0]), float(off[1])), float(np.float32(scale))

    def freq_error(self, off, scale, last, now):
        out = np.zeros(len(last), np.float32)
        off = np.ascontiguousarray(off, np.float32)
        for i in range(len(last)):
            a, b = np.ascontiguousarray(last[i]), np.ascontiguousarray(now[i])
            out[i] = self.L.periodogram_freq_error(fp(off), scale, fp(a), fp(b))
        return out

    def vec_dot(self, x, y):
        return np.array([self.L.vec_dot_prodf(fp(np.ascontiguousarray(x[i])), fp(np.ascontiguousarray(y[i])), x.shape[1]) for i in range(len(x))], np.float32)

    def vec_lms(self, x, y, err):
        out = y.copy()
        for i in range(len(x)):
            yi = np.ascontiguousarray(out[i])
            self.L.vec_lmsf(fp(np.ascontiguousarray(x[i])), fp(yi), x.shape[1], float(err[i]))
            out[i] = yi
        return out

    def cvec_dot(self, x, y):
        out = np.zeros((len(x), 2), np.float32)
        for i in range(len(x)):
            z = self.L.cvec_dot_prodf(fp(np.ascontiguousarray(x[i])), fp(np.ascontiguousarray(y[i])), x.shape[1])
            out[i] = (z.re, z.im)
        return out

    def cvec_lms(self, x, y, err):
        out = y.copy()
        for i in range(len(x)):
            yi = np.ascontiguousarray(out[i])
            e = np.ascontiguousarray(err[i])
            self.L.cvec_lmsf(fp(np.ascontiguousarray(x[i])), fp(yi), x.shape[1], fp(e))
            out[i] = yi
        return out

    # the whole-domain sweeps go through array loops where the library has them (the reference's glue)
    def sqrt32(self, x):
        out = np.zeros(len(x), np.uint16)
        if hasattr(self.H, "glue_fixed_sqrt32_batch"):
            self.H.glue_fixed_sqrt32_batch(fp(x), fp(out), len(x))
        else:
            for i in range(len(x)):
                out[i] = self.L.fixed_sqrt32(int(x[i]))
        return out

    def dds(self, acc, rate, n):
        out = np.zeros((len(acc), n, 2), np.float32)
        if hasattr(self.H, "glue_dds_complexf_batch"):
            self.H.glue_dds_complexf_batch(fp(acc), fp(rate), fp(out), len(acc), n)
        else:
            for i in range(len(acc)):
                a = C.c_uint32(int(acc[i]))
                for k in range(n):
                    z = self.L.dds_complexf(C.byref(a), int(rate[i]))
                    out[i, k] = (z.re, z.im)
                acc[i] = a.value
        return out, acc

    def arctan2(self, y, x):
        out = np.zeros(len(y), np.int32)
        self.H.glue_arctan2_batch(fp(y), fp(x), fp(out), len(y))
        return out


class Restated:
    """oracle/prims_oracle.c"""

    def __init__(self):
        from oracle import restated as orc
        self.L = orc.lib()
        self.L.orc_periodogram_generate_phase_offset.restype = C.c_float
        self.L.orc_periodogram_generate_phase_offset.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
        self.L.orc_periodogram_generate_coeffs.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
        self.L.orc_periodogram_freq_error.restype = C.c_float
        self.L.orc_periodogram_freq_error.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        self.L.orc_periodogram.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self.L.orc_periodogram_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.L.orc_periodogram_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self.L.orc_vec_dot_prodf.restype = C.c_float
        self.L.orc_vec_dot_prodf.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.L.orc_vec_lmsf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        self.L.orc_cvec_dot_prodf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self.L.orc_cvec_lmsf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self.L.orc_fixed_sqrt32_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.L.orc_arctan2_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.L.orc_dds_complexf_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]

    def periodogram(self, co, amp, n):
        out = np.zeros((len(amp), 2), np.float32)
        for i in range(len(amp)):
            o = np.zeros(2, np.float32)
            self.L.orc_periodogram(fp(np.ascontiguousarray(co[i])), fp(np.ascontiguousarray(amp[i])), n, fp(o))
            out[i] = o
        return out

    def prepare(self, amp, n):
        s = np.zeros((len(amp), n//2, 2), np.float32)
        d = np.zeros_like(s)
        for i in range(len(amp)):
            si, di = s[i].copy(), d[i].copy()
            self.L.orc_periodogram_prepare(fp(si), fp(di), fp(np.ascontiguousarray(amp[i])), n)
            s[i], d[i] = si, di
        return s, d

    def apply(self, co, s, d, n):
        out = np.zeros((len(s), 2), np.float32)
        for i in range(len(s)):
            o = np.zeros(2, np.float32)
            self.L.orc_periodogram_apply(fp(np.ascontiguousarray(co[i])), fp(np.ascontiguousarray(s[i])), fp(np.ascontiguousarray(d[i])), n, fp(o))
            out[i] = o
        return out

    def gen_coeffs(self, freq, rate, n):
        c = np.zeros((n//2, 2), np.float32)
        self.L.orc_periodogram_generate_coeffs(fp(c), freq, rate, n)
        return c

    def gen_phase_offset(self, freq, rate, interval):
        off = np.zeros(2, np.float32)
        scale = self.L.orc_periodogram_generate_phase_offset(fp(off), freq, rate, interval)
        return (float(off[0]), float(off[1])), float(np.float32(scale))

    def freq_error(self, off, scale, last, now):
        out = np.zeros(len(last), np.float32)
        off = np.ascontiguousarray(off, np.float32)
        for i in range(len(last)):
            out[i] = self.L.orc_periodogram_freq_error(fp(off), scale, fp(np.ascontiguousarray(last[i])), fp(np.ascontiguousarray(now[i])))
        return out

    def vec_dot(self, x, y):
        return np.array([self.L.orc_vec_dot_prodf(fp(np.ascontiguousarray(x[i])), fp(np.ascontiguousarray(y[i])), x.shape[1]) for i in range(len(x))], np.float32)

    def vec_lms(self, x, y, err):
        out = y.copy()
        for i in range(len(x)):
            yi = np.ascontiguousarray(out[i])
            self.L.orc_vec_lmsf(fp(np.ascontiguousarray(x[i])), fp(yi), x.shape[1], float(err[i]))
            out[i] = yi
        return out

    def cvec_dot(self, x, y):
        out = np.zeros((len(x), 2), np.float32)
        for i in range(len(x)):
            o = np.zeros(2, np.float32)
            self.L.orc_cvec_dot_prodf(fp(np.ascontiguousarray(x[i])), fp(np.ascontiguousarray(y[i])), x.shape[1], fp(o))
            out[i] = o
        return out

    def cvec_lms(self, x, y, err):
        out = y.copy()
        for i in range(len(x)):
            yi = np.ascontiguousarray(out[i])
            self.L.orc_cvec_lmsf(fp(np.ascontiguousarray(x[i])), fp(yi), x.shape[1], fp(np.ascontiguousarray(err[i])))
            out[i] = yi
        return out

    def sqrt32(self, x):
        out = np.zeros(len(x), np.uint16)
        self.L.orc_fixed_sqrt32_batch(fp(x), fp(out), len(x))
        return out

    def dds(self, acc, rate, n):
        out = np.zeros((len(acc), n, 2), np.float32)
        self.L.orc_dds_complexf_batch(fp(acc), fp(rate), fp(out), len(acc), n)
        return out, acc

    def arctan2(self, y, x):
        out = np.zeros(len(y), np.int32)
        self.L.orc_arctan2_batch(fp(y), fp(x), fp(out), len(y))
        return out
