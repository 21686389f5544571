"""The receive front end of a FAX terminal over banks: what fax_modems_v29_v21_rx() does per channel
(/root/reference/src/fax_modems.c:290-323 -- the fast modem and the V.21 receiver both get every frame until one of
them proves to be the one in use, then only that one runs) done for N channels with two banks and per-tick active
masks (spangpu_modem_rx_var(), spangpu_fsk_rx_var()).

Each channel's two receivers must deliver what oracle receivers deliver when they are handed exactly the frames the
reference's handler switching would hand them: all frames while both run, none after the channel has switched away.
"""
import os

import numpy as np
import pytest

import synth
from test_oracle_pin import GOLDEN, bits, use_golden_modem_tables

pytestmark = pytest.mark.gpu

TRAINING_SUCCEEDED = -4         # SIG_STATUS_TRAINING_SUCCEEDED, spandsp/async.h
BOTH, FAST_ONLY, V21_ONLY = 0, 1, 2


def test_v29_v21_front_end_with_per_channel_switching(built):
    from oracle import restated as orc
    from spandsp_amd import engine
    use_golden_modem_tables()
    n = 48
    frames = 260
    sp = engine.fsk_preset(engine.FSK_V21CH2)
    v29 = np.load(os.path.join(GOLDEN, "v29_9600.npz"))["amp"]
    v21 = synth.fsk_channels(n, 160*frames, 91, sp.freq_zero, sp.freq_one, sp.baud_rate)
    rng = np.random.default_rng(4)
    sig = np.zeros((n, 160*frames), np.int16)
    kind = np.arange(n) % 3                              # 0: a V.29 page, 1: V.21 signalling, 2: an idle line
    for c in range(n):
        if kind[c] == 0:
            d = int(rng.integers(0, 900))
            m = min(len(v29), sig.shape[1] - d)
            sig[c, d:d + m] = v29[:m]
        elif kind[c] == 1:
            sig[c] = v21[c]
        sig[c] = np.clip(sig[c].astype(np.int32) + rng.normal(0, 6, sig.shape[1]), -32768, 32767).astype(np.int16)
    fast = engine.V29Bank(n, 9600)
    slow = engine.FskBank(engine.FSK_V21CH2, n, engine.FSK_FRAME_MODE_SYNC)
    ofast = [orc.V29(9600) for _ in range(n)]
    oslow = [orc.Fsk(engine.FSK_V21CH2, engine.FSK_FRAME_MODE_SYNC) for _ in range(n)]
    handler = np.full(n, BOTH)
    v21_bits = np.zeros(n, np.int64)
    switched = {FAST_ONLY: 0, V21_ONLY: 0}
    for f in range(frames):
        x = sig[:, f*160:(f + 1)*160]
        lens_fast = np.where(handler != V21_ONLY, 160, 0).astype(np.int32)
        lens_slow = np.where(handler != FAST_ONLY, 160, 0).astype(np.int32)
        # both banks take the tick's frames on their own streams; nothing orders one after the other
        fast.rx_host_var(x, lens_fast)
        slow.rx_host_var(x, lens_slow)
        ef = fast.events() if lens_fast.any() else [np.zeros(0, np.int8)]*n
        es = slow.events() if lens_slow.any() else [np.zeros(0, np.int32)]*n
        for c in range(n):
            ofast[c].sink.clear()
            oslow[c].sink.clear()
            if lens_fast[c]:
                ofast[c].rx(x[c])
            if lens_slow[c]:
                oslow[c].rx(x[c])
            want_f = ofast[c].sink.events()["a"].astype(np.int8)
            want_s = oslow[c].sink.events()["a"].astype(np.int64)
            assert np.array_equal(ef[c], want_f), ("fast", c, f)
            assert np.array_equal(np.asarray(es[c]).astype(np.int64), want_s), ("v21", c, f)
            # the switching of fax_modems.c:275-287 and :299-305, with "a frame was received" stood in for by a run of bits
            if handler[c] == BOTH:
                v21_bits[c] += int((want_s >= 0).sum())
                if TRAINING_SUCCEEDED in want_f:
                    handler[c] = FAST_ONLY
                    switched[FAST_ONLY] += 1
                elif v21_bits[c] >= 96:
                    handler[c] = V21_ONLY
                    switched[V21_ONLY] += 1
    assert switched[FAST_ONLY] >= n//3 - 1 and switched[V21_ONLY] >= n//3 - 1
    assert (handler[kind == 2] == BOTH).all()
    for c in range(n):
        fw, iw = fast.get_state(c)
        of, oi = ofast[c].snapshot()
        assert np.array_equal(bits(fw), bits(of)) and np.array_equal(iw, oi), c
        assert np.array_equal(slow.get_state(c), oslow[c].snapshot()), c
