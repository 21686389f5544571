"""Bank-size soaks: minutes of signal through full-size banks, tick by tick, looking for what twenty-frame tests cannot
show -- drift, counter wrap, ring-slot reuse.

* 65 536 DTMF channels x 60 s (3 000 ticks of 160 samples): 1 024 distinct lines, every one held to the oracle digit for
  digit (and its final state words), the other 63 copies of each line held to their original record for record on the
  device, every tick.
* 16 384 V.29 channels through three calls on the same lines (train, carry a page, drop carrier; twice more): 256
  distinct lines against the oracle event for event, the 63 copies of each against their original."""
import ctypes
import os

import numpy as np
import pytest

import synth
from test_oracle_pin import GOLDEN, bits, use_golden_modem_tables

pytestmark = [pytest.mark.gpu, pytest.mark.soak]


def test_dtmf_bank_sixty_seconds(built):
    import torch
    from oracle import restated as orc
    from spandsp_amd import engine
    dev = torch.device("cuda", 0)
    distinct, copies, frame, ticks = 1024, 64, 160, 3000
    n_ch = distinct*copies
    sig, _ = synth.dtmf_channels(distinct, frame*ticks, seed=2026)
    # the oracle, on the distinct lines (C: a few seconds)
    dets = [orc.Dtmf(0) for _ in range(distinct)]
    want_digits = []
    for c, d in enumerate(dets):
        parts = []
        for k in range(0, sig.shape[1], 8000):      # a second at a time: dtmf_rx_get() holds 128 digits
            d.rx(sig[c, k:k + 8000])
            parts.append(d.get())
        want_digits.append("".join(parts))
    assert sum(len(w) for w in want_digits) > 50*distinct      # a digit every half second or so, for a minute
    base = torch.tensor(sig.reshape(distinct, ticks, frame).transpose(1, 0, 2).copy(), device=dev)      # [tick][line][160]
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    bank = engine.ToneBank(engine.DTMF, n_ch)
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    rec = torch.zeros(2, n_ch, dtype=torch.int32, device=dev)
    bank.set_records_buffer(rec.data_ptr(), rec.numel()*4)
    frame_buf = torch.empty(n_ch, frame, dtype=torch.int16, device=dev)
    got = [[] for _ in range(distinct)]
    bad_copies = torch.zeros((), dtype=torch.int64, device=dev)
    for t in range(ticks):
        frame_buf.view(copies, distinct, frame).copy_(base[t].unsqueeze(0).expand(copies, distinct, frame))
        rec.zero_()
        bank.rx_device(ctypes.c_void_p(frame_buf.data_ptr()), frame, frame)
        r = rec.view(2, copies, distinct)
        bad_copies += (r != r[:, :1, :]).sum()
        first = r[:, 0, :].cpu().numpy().view(np.uint32)        # [block][line]
        flags = (first >> 16) & 0xFF
        code = (first >> 8) & 0xFF
        hit = (flags & engine.BLK_CHANGE).astype(bool) & (code != 0)
        for b, c in zip(*np.nonzero(hit)):
            got[c].append((t, b, chr(code[b, c])))
    assert int(bad_copies.item()) == 0
    for c in range(distinct):
        assert "".join(ch for _, _, ch in sorted(got[c])) == want_digits[c], c
    # and the state of a few lines (first copy and last copy) after the minute
    for c in (0, 511, 1023):
        for k in (0, copies - 1):
            f, i = bank.get_state(k*distinct + c)
            o = dets[c].snapshot()
            assert np.array_equal(bits(f[0:8]), bits(o["v2"])) and np.array_equal(bits(f[8:16]), bits(o["v3"])), (c, k)
            assert bits(f[16:17])[0] == bits(np.array([o["energy"]], np.float32))[0], (c, k)
            assert i[0] == o["current_sample"] and i[1] == o["last_hit"] and i[2] == o["in_digit"] and i[3] == o["duration"], (c, k)
    bank.set_records_buffer(None, 0)
    bank.close()


def test_v29_bank_three_calls(built):
    import torch
    from oracle import restated as orc
    from spandsp_amd import engine
    use_golden_modem_tables()
    dev = torch.device("cuda", 0)
    distinct, copies, frame = 256, 64, 160
    n_ch = distinct*copies
    g = np.load(os.path.join(GOLDEN, "v29_9600.npz"))
    call = g["amp"].astype(np.float64)
    rng = np.random.default_rng(929)
    gap = 1200
    n = 3*(len(call) + gap) + 160
    n -= n % frame
    sig = np.zeros((distinct, n), np.int16)
    for c in range(distinct):
        x = np.zeros(n)
        for k in range(3):
            s = k*(len(call) + gap) + int(rng.integers(0, 300))
            x[s:s + len(call)] = call*10.0**(rng.uniform(-12.0, 2.0)/20.0)
        x += rng.normal(0.0, rng.choice([0.0, 2.0, 15.0, 60.0]), n)
        sig[c] = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    ticks = n//frame
    want = []
    trained = 0
    for c in range(distinct):
        o = orc.V29(9600)
        per = []
        for t in range(ticks):
            o.sink.clear()
            o.rx(sig[c, t*frame:(t + 1)*frame])
            per.append(o.sink.events()["a"].astype(np.int8))
        want.append((per, o.snapshot()))
        trained += int(sum(int((p == -4).sum()) for p in per))
    assert trained >= 2*distinct                # most lines train on every one of the three calls
    base = torch.tensor(sig.reshape(distinct, ticks, frame).transpose(1, 0, 2).copy(), device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    bank = engine.ModemBank(engine.V29, n_ch, 9600)
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    frame_buf = torch.empty(n_ch, frame, dtype=torch.int16, device=dev)
    for t in range(ticks):
        frame_buf.view(copies, distinct, frame).copy_(base[t].unsqueeze(0).expand(copies, distinct, frame))
        bank.rx_device(ctypes.c_void_p(frame_buf.data_ptr()), frame, frame)
        ev = bank.events()
        for c in range(distinct):
            assert np.array_equal(ev[c], want[c][0][t]), (t, c)
        if t % 16 == 0 or t == ticks - 1:
            for k in (1, 31, copies - 1):
                for c in range(0, distinct, 37):
                    assert np.array_equal(ev[k*distinct + c], ev[c]), (t, k, c)
    for c in (0, 100, 255):
        for k in (0, copies - 1):
            f, w = bank.get_state(k*distinct + c)
            of, ow = want[c][1]
            assert np.array_equal(bits(f), bits(of)) and np.array_equal(w, ow), (c, k)
    bank.close()
