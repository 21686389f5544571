"""The G.168 line simulator of the reference's echo canceller test program, restated (TEST INFRASTRUCTURE).

tests/echo_tests.c of the reference simulates a line as channel_model() does (:396-487): the far end talker's signal goes
through an FIR with one of the eight echo path models of ITU-T G.168 (tables in src/spandsp/g168models.h) at a chosen echo
return loss, and the near end talker's signal is added with saturation:

    gain = 32768.0f*powf(10.0f, erl/20.0f)*ki[model]                      (:443, binary32 throughout; erl is negative dB)
    echo = fir32(&impulse, rout*gain)                                     (:487; the product is truncated to int16)
    sin  = sat_add16(echo, sgen)                                          (:488)

with fir32() = spandsp/fir.h:232-251: y = sum coeffs[i]*x[n - i] in wrap-around int32, result (int16) (y >> 15).

The model tables are data of the reference's test program, committed as tests/golden/g168_models.npz by
tests/golden/make_golden.py (read out of the header through oracle/_ref, where the gains for the ERLs used here are also
computed by the C library's powf); `line()` below is pinned to the reference's own fir32() in tests/test_oracle_pin.py.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_models = None


def models():
    """{model number 2 .. 9: (taps int32[], ki float32)} and the committed gains {(model, erl_db): float32}."""
    global _models
    if _models is None:
        g = np.load(os.path.join(GOLDEN, "g168_models.npz"))
        m = {int(k): (g["taps_%d" % k].astype(np.int32), np.float32(g["ki"][i])) for i, k in enumerate(g["models"])}
        gains = {(int(a), float(b)): np.float32(c) for a, b, c in zip(g["gain_model"], g["gain_erl"], g["gain_value"])}
        _models = (m, gains)
    return _models


def gain(model, erl_db):
    """32768.0f*powf(10.0f, erl/20.0f)*ki in binary32; the committed value where the fixture holds one (powf is libm's)."""
    m, gains = models()
    key = (int(model), float(erl_db))
    if key in gains:
        return gains[key]
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    p = np.float32(libm.powf(10.0, float(np.float32(erl_db)/np.float32(20.0))))
    return np.float32(np.float32(np.float32(32768.0)*p)*m[int(model)][1])


def line(model, erl_db, rout, sgen=None):
    """sin[] for the far end signal rout[] (and the near end signal sgen[], silence if None) on line model D`model`."""
    m, _ = models()
    taps = m[int(model)][0].astype(np.int64)
    rout = np.asarray(rout, np.int16)
    g = gain(model, erl_db)
    x = (rout.astype(np.float32)*g).astype(np.int64)            # float -> int16 of values well inside its range: truncation
    y = np.convolve(x, taps)[:len(x)]                            # exact in int64; int32 wrap-around below
    y = ((y + 2**31) % 2**32) - 2**31
    echo = ((y >> 15) & 0xFFFF).astype(np.uint16).view(np.int16).astype(np.int32)
    near = np.zeros(len(x), np.int32) if sgen is None else np.asarray(sgen, np.int16).astype(np.int32)
    return np.clip(echo + near, -32768, 32767).astype(np.int16)


def erle_db(rx, clean):
    """10 log10 (sum rx^2 / sum clean^2) -- SURVEY 8(d)-5's per-line figure of merit."""
    a = np.asarray(rx, np.float64)
    b = np.asarray(clean, np.float64)
    return float(10.0*np.log10(max((a*a).sum(), 1e-9)/max((b*b).sum(), 1e-9)))


# BASELINE.md section 2's known answer: echo_can_update(), 128 taps, ECHO_CAN_USE_ADAPTION, white noise at -15 dBm0 (the
# reference's awgn(), seed 1234567) through model D2 at an ERL of 12 dB, 20 s; ERLE over the last second.
KNOWN_D2 = {"model": 2, "erl_db": -12.0, "seed": 1234567, "level_dbm0": -15.0, "samples": 160000, "taps": 128, "mode": 0x01,
            "erle_db": 54.6}
