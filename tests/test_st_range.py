"""super_tone_rx() on the detection range test of the reference's own test program (tests/super_tone_rx_tests.c:436-471,
restated in tests/st_range.py): 350 Hz + 440 Hz swept from -80 to -1 dBm0 into a receiver built on that program's
two-tone descriptor, both callbacks installed.  The golden file holds every callback of the REAL reference
(tests/golden/make_golden.py: super_tone_range) and a CRC of the signal (regenerated here with numpy; the generator was
held against the reference's own dds() when the golden was made).  The restated oracle (CPU) and the super_tone_rx()
shim over the HIP engine (GPU) must deliver the same callbacks, in order."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import st_range

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "super_tone_range.npz")
TONE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int)
SEG_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int)


def _run(rx_frame):
    crc = 0
    for level, frames in st_range.sweep():
        crc = zlib.crc32(frames.tobytes(), crc)
        for fr in frames:
            rx_frame(np.ascontiguousarray(fr))
    return crc


def _check(events, crc):
    g = np.load(GOLDEN)
    assert np.uint32(crc) == g["signal_crc"], "the regenerated sweep differs from the reference's"
    assert np.array_equal(np.array(events, np.int32).reshape(-1, 4), g["events"])
    assert len(events) >= 20


def test_super_tone_range_oracle(built):
    from oracle import restated as orc
    rx = orc.SuperTone(st_range.fill_descriptor(orc.SuperToneDesc()), True)
    crc = _run(rx.rx)
    _check([tuple(int(x) for x in e) for e in rx.sink.events()], crc)


@pytest.mark.gpu
def test_super_tone_range_super_tone_rx_shim(built):
    from spandsp_amd import engine
    L = C.CDLL(engine.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    for name, (res, args) in {"super_tone_rx_make_descriptor": (vp, [vp]), "super_tone_rx_free_descriptor": (ci, [vp]),
                              "super_tone_rx_add_tone": (ci, [vp]), "super_tone_rx_add_element": (ci, [vp, ci, ci, ci, ci, ci]),
                              "super_tone_rx_init": (vp, [vp, vp, TONE_CB, vp]), "super_tone_rx_free": (ci, [vp]),
                              "super_tone_rx_segment_callback": (None, [vp, SEG_CB]),
                              "super_tone_rx": (ci, [vp, vp, ci])}.items():
        getattr(L, name).restype = res
        getattr(L, name).argtypes = args
    desc = L.super_tone_rx_make_descriptor(None)

    class D:
        def add_tone(self):
            return L.super_tone_rx_add_tone(desc)

        def add_element(self, *a):
            return L.super_tone_rx_add_element(desc, *a)
    st_range.fill_descriptor(D())
    events = []
    tone_cb = TONE_CB(lambda ud, code, level, delay: events.append((1, code, level, delay)))
    seg_cb = SEG_CB(lambda ud, f1, f2, dur: events.append((4, f1, f2, dur)))
    s = L.super_tone_rx_init(None, desc, tone_cb, None)
    assert s
    L.super_tone_rx_segment_callback(s, seg_cb)

    def rx(fr):
        assert L.super_tone_rx(s, fr.ctypes.data, len(fr)) == len(fr)
    crc = _run(rx)
    L.super_tone_rx_free(s)
    L.super_tone_rx_free_descriptor(desc)
    _check(events, crc)
