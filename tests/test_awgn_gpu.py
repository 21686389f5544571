"""Noise source banks (awgn) against the oracle.

Bar: every int16 sample and every state word (the three LCGs, the 97 entry shuffle table, rms, the carried half pair
amp2 as a binary64 bit pattern) identical.  log(), the one library call on the path, is GNU libc's routine restated on
the device (csrc/glibc_log_dev.hpp) and in the oracle (oracle/glibc_log.c, held to the host's libm and to the golden
vectors of the real reference in test_oracle_pin.py).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AMP2 = slice(2, 4)


def as_double(words):
    return np.frombuffer(np.asarray(words, np.uint32).tobytes(), np.float64)[0]


def check_state(got, want):
    assert np.array_equal(got, want), (np.nonzero(got != want)[0][:8], as_double(got[AMP2]), as_double(want[AMP2]))


def check_samples(got, want):
    assert np.array_equal(got, want), np.count_nonzero(got != want)


def make(n, seed):
    rng = np.random.default_rng(seed)
    seeds = rng.integers(-2_000_000, 2_000_000, n).astype(np.int32)
    seeds[:4] = [0, 1, -1, 1234567]
    levels = rng.uniform(-60.0, -5.0, n).astype(np.float32)
    levels[:3] = [0.0, 6.0, -90.0]           # clipping, hard clipping, rms of half a step
    return seeds, levels


def test_awgn_matches_oracle(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 200
    seeds, levels = make(n, 11)
    bank = engine.AwgnBank(seeds, levels)
    orcs = [orc.Awgn(int(s), float(lv)) for s, lv in zip(seeds, levels)]
    for c in range(0, n, 17):
        check_state(bank.get_state(c), orcs[c].snapshot())
    for m in [160, 1, 160, 3, 1, 1, 333, 2, 160, 4000, 7, 160]:       # odd sizes leave the spare value pending
        got = bank.tx_host(m)
        want = np.stack([o.gen(m) for o in orcs])
        check_samples(got, want)
        for c in range(0, n, 9):
            check_state(bank.get_state(c), orcs[c].snapshot())
    # the levels are what was asked for
    long = bank.tx_host(16000).astype(np.float64)
    for c in range(3, n, 13):
        dbm0 = 10.0*np.log10(np.mean(long[c]**2)/(32768.0**2)) + 3.14 + 3.02
        assert abs(dbm0 - levels[c]) < 0.5, (c, dbm0, levels[c])
    assert np.all(np.abs(long[2]) <= 3) and np.any(long[2] != 0)
    assert np.mean(np.abs(long[1]) >= 32767) > 0.2


def test_awgn_mix_reinit_and_device_buffers(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 130
    seeds, levels = make(n, 12)
    bank = engine.AwgnBank(seeds, levels)
    orcs = [orc.Awgn(int(s), float(lv)) for s, lv in zip(seeds, levels)]
    rng = np.random.default_rng(5)
    base = rng.integers(-32768, 32768, (n, 161)).astype(np.int16)
    got = bank.tx_host(161, mix_into=base)
    want = np.stack([np.clip(base[c].astype(np.int32) + o.gen(161), -32768, 32767) for c, o in enumerate(orcs)]).astype(np.int16)
    check_samples(got, want)
    assert np.any((got == 32767) | (got == -32768))
    # a channel reseeded in place
    bank.reinit(7, 4242, -20.0)
    orcs[7] = orc.Awgn(4242, -20.0)
    check_state(bank.get_state(7), orcs[7].snapshot())
    # device resident buffer with a row stride, written in place then mixed into
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    stride = 192
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), n*stride*2) == 0
    marker = np.full((n, stride), 77, np.int16)
    assert hip.hipMemcpy(d, marker.ctypes.data, marker.nbytes, 1) == 0
    bank.tx_device(d, stride, 160)
    bank.tx_device(d, stride, 160, mix=True)
    bank.sync()
    out = np.zeros((n, stride), np.int16)
    assert hip.hipMemcpy(out.ctypes.data, d, out.nbytes, 2) == 0
    hip.hipFree(d)
    first = np.stack([o.gen(160) for o in orcs]).astype(np.int32)
    second = np.stack([o.gen(160) for o in orcs]).astype(np.int32)
    check_samples(out[:, :160], np.clip(first + second, -32768, 32767).astype(np.int16))
    assert np.all(out[:, 160:] == 77)
    for c in range(0, n, 11):
        check_state(bank.get_state(c), orcs[c].snapshot())
