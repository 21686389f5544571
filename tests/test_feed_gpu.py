"""The pipelined host-buffer path (spangpu_feed_*, csrc/feed_api.hip): ticks queued three deep deliver exactly the digits
the serial path (rx_host + blocks, itself held to the oracle elsewhere) delivers, tick for tick, for 16 bit PCM and for
G.711 bytes; a slot cannot be committed twice, and an empty feed says so."""
import os

import numpy as np
import pytest

import synth
from test_oracle_pin import GOLDEN

pytestmark = pytest.mark.gpu


def serial_digits(engine, sig, n_ch, frame, law=0, codes=None):
    bank = engine.ToneBank(engine.DTMF, n_ch)
    out = []
    for pos in range(0, sig.shape[1], frame):
        if law:
            bank.rx_host_g711(codes[:, pos:pos + frame], law)
        else:
            bank.rx_host(sig[:, pos:pos + frame])
        blk = bank.blocks()
        out.append(sorted((int(r["channel"]), int(r["code"]), int(r["block"])) for r in blk
                          if (r["flags"] & engine.BLK_CHANGE) and r["code"]))
    bank.close()
    return out


@pytest.mark.parametrize("law", [0, 2])
def test_pipelined_feed_delivers_the_serial_path_digits(built, law):
    from spandsp_amd import engine
    n_ch, frame, ticks = 3000, 160, 50
    sig, _ = synth.dtmf_channels(n_ch, frame*ticks, seed=91)
    codes = None
    if law:
        tab = np.load(os.path.join(GOLDEN, "g711_decode.npz"))["ulaw"].astype(np.int32)
        order = np.argsort(tab, kind="stable")
        vals = tab[order]
        pos = np.clip(np.searchsorted(vals, sig.astype(np.int32)), 1, 255)
        lower = (sig - vals[pos - 1]) <= (vals[pos] - sig)
        codes = order[np.where(lower, pos - 1, pos)].astype(np.uint8)
    want = serial_digits(engine, sig, n_ch, frame, law, codes)
    bank = engine.ToneBank(engine.DTMF, n_ch)
    feed = engine.Feed(bank, frame, law=law, depth=3)
    got = []
    for t in range(ticks):
        buf = feed.slot()
        buf[:, :frame] = (codes if law else sig)[:, t*frame:(t + 1)*frame]
        feed.commit(frame)
        if t >= 2:                              # three ticks in flight
            c, d, b = feed.collect()
            got.append(sorted(zip(c.tolist(), d.tolist(), b.tolist())))
    while True:
        r = feed.collect()
        if r is None:
            break
        got.append(sorted(zip(r[0].tolist(), r[1].tolist(), r[2].tolist())))
    assert len(got) == ticks
    assert got == want
    assert sum(len(g) for g in got) > n_ch//2
    assert feed.collect() is None
    feed.close()
    bank.close()


def test_feed_slots_are_bounded(built):
    from spandsp_amd import engine
    bank = engine.ToneBank(engine.DTMF, 100)
    feed = engine.Feed(bank, 160, depth=2)
    for _ in range(2):
        feed.slot()[:] = 0
        feed.commit(160)
    with pytest.raises(engine.SpanGpuError):
        feed.slot()                             # both slots hold a tick nobody has collected
    assert len(feed.collect()[0]) == 0
    feed.slot()
    feed.close()
    bank.close()


def test_feed_run_is_the_same_loop_in_c(built):
    """spangpu_feed_run(): the tick loop of a C caller over the frames the slots hold -- the same digits as the loop written
    out with acquire / commit / collect on a second bank fed the same frames."""
    from spandsp_amd import engine
    n_ch, frame, depth, ticks = 2000, 160, 3, 12
    sig, _ = synth.dtmf_channels(n_ch, frame*depth, seed=92)
    counts = []
    for use_c in (False, True):
        bank = engine.ToneBank(engine.DTMF, n_ch)
        feed = engine.Feed(bank, frame, depth=depth)
        for t in range(depth):                  # fill every slot once (and take those ticks)
            feed.slot()[:, :frame] = sig[:, t*frame:(t + 1)*frame]
            feed.commit(frame)
        n = 0
        while True:
            r = feed.collect()
            if r is None:
                break
            n += len(r[0])
        if use_c:
            ms, d = feed.run(frame, ticks, 1)
            assert ms > 0.0
            n += d
        else:
            for t in range(ticks):
                feed.slot()
                feed.commit(frame)
                if t >= 1:
                    n += len(feed.collect()[0])
            while True:
                r = feed.collect()
                if r is None:
                    break
                n += len(r[0])
        with pytest.raises(Exception):
            feed.run(frame, 1, depth)           # lag must leave a slot free
        counts.append(n)
        feed.close()
        bank.close()
    assert counts[0] == counts[1] > 0


def test_feed_tick_with_more_digits_than_travel_with_a_tick(built):
    """Only a bounded part of the digit list comes back with every tick; when every line of a big bank delivers a digit in
    the same 20 ms (lines in step), collect() fetches the rest."""
    from spandsp_amd import engine
    n_ch, frame, ticks = 40000, 160, 10
    one, _ = synth.dtmf_channels(1, frame*ticks, seed=93)
    bank = engine.ToneBank(engine.DTMF, n_ch)
    feed = engine.Feed(bank, frame, depth=2)
    per_tick = []
    for t in range(ticks):
        buf = feed.slot()
        buf[:, :frame] = one[0, t*frame:(t + 1)*frame]          # every line the same signal
        feed.commit(frame)
        c, d, b = feed.collect()
        per_tick.append((c, d, b))
    full = [x for x in per_tick if len(x[0]) > 0]
    assert full, "the line carries at least one digit"
    for c, d, b in full:
        assert len(c) == n_ch and np.array_equal(np.sort(c), np.arange(n_ch, dtype=np.uint32))
        assert len(set(d.tolist())) == 1 and len(set(b.tolist())) == 1
    feed.close()
    bank.close()


@pytest.mark.parametrize("law", [0, 1, 2], ids=["pcm", "alaw", "ulaw"])
def test_pipelined_echo_feed_equals_the_serial_path(built, law):
    """spangpu_echo_feed_*: ticks three deep (H2D, kernel and D2H of successive ticks on three streams) return exactly the
    clean rows of the serial path (update_host, itself held to the oracle in test_echo_gpu.py).  With a G.711 law the rows
    travel as bytes both ways: decoded on the device (the reference's tables, golden/g711_decode.npz), cancelled, and encoded
    on the device with the reference's linear_to_alaw() / linear_to_ulaw() (golden/g711_encode.npz: every int16 value)."""
    from spandsp_amd import engine
    from test_echo_gpu import make_channels
    n_ch, taps, frame, ticks = 700, 128, 160, 40
    tx, rx = make_channels(n_ch, frame*ticks, taps, seed=515)
    if law:
        name = "alaw" if law == 1 else "ulaw"
        dec = np.load(os.path.join(GOLDEN, "g711_decode.npz"))[name].astype(np.int16)
        enc = np.load(os.path.join(GOLDEN, "g711_encode.npz"))[name]
        tx_c, rx_c = enc[tx.astype(np.int32) + 32768], enc[rx.astype(np.int32) + 32768]
        tx, rx = dec[tx_c], dec[rx_c]                                     # what the canceller sees
    serial = engine.EchoBank(n_ch, taps, 0x01)
    want = [serial.update_host(tx[:, t*frame:(t + 1)*frame], rx[:, t*frame:(t + 1)*frame], False) for t in range(ticks)]
    serial.close()
    bank = engine.EchoBank(n_ch, taps, 0x01)
    feed = engine.EchoFeed(bank, frame, law=law, depth=3)
    got = []

    def take():
        clean, n = feed.collect()
        assert n == frame
        got.append(clean[:, :frame].copy())
    for t in range(ticks):
        a, b = feed.slots()
        a[:, :frame] = (tx_c if law else tx)[:, t*frame:(t + 1)*frame]
        b[:, :frame] = (rx_c if law else rx)[:, t*frame:(t + 1)*frame]
        feed.commit(frame)
        if t >= 2:
            take()
    while feed.outstanding():
        take()
    assert len(got) == ticks
    for t in range(ticks):
        w = enc[want[t].astype(np.int32) + 32768] if law else want[t]
        assert np.array_equal(got[t], w), t
    with pytest.raises(engine.SpanGpuError):
        for _ in range(4):
            feed.slots()
            feed.commit(frame)
    feed.close()
    bank.close()


def test_packed_modem_events_and_the_pipelined_modem_feed(built):
    """spangpu_modem_pack_events / _unpack_events and spangpu_modem_feed_*: the put_bit stream of every tick, packed on the
    device (a header word + the data bits per channel, one sparse list of status reports per bank), brought back three ticks
    deep and unpacked on the host, equals the serial path's events call for call -- through training (status reports), data,
    the end of the carrier and a second call."""
    from spandsp_amd import engine
    from test_v29_gpu import channel_signals
    from test_oracle_pin import use_golden_modem_tables
    use_golden_modem_tables()
    n_ch, frame = 300, 160
    base = channel_signals(9600, n_ch, seed=23)
    sig = np.concatenate([base, np.zeros((n_ch, 480), np.int16), base[:, :3200]], axis=1)
    ticks = sig.shape[1]//frame
    serial = engine.V29Bank(n_ch, 9600)
    want = []
    for t in range(ticks):
        serial.rx_host(sig[:, t*frame:(t + 1)*frame])
        want.append(serial.events())
        if t % 3 == 0:                                          # spangpu_modem_events_packed(): the same answer by the packed way
            again = serial.events(packed=True)
            assert all(np.array_equal(a, b) for a, b in zip(again, want[-1])), t
    serial.close()
    bank = engine.V29Bank(n_ch, 9600)
    feed = engine.ModemFeed(bank, frame, 9600, depth=3)
    assert feed.wpc == 8                                        # 32 bytes per channel and tick: a header word + 192 bits + slack
    got = []
    for t in range(ticks):
        feed.slot()[:, :frame] = sig[:, t*frame:(t + 1)*frame]
        feed.commit(frame)
        if t >= 2:
            got.append(feed.collect())
    while feed.outstanding():
        got.append(feed.collect())
    assert len(got) == ticks
    n_status = 0
    for t in range(ticks):
        for c in range(n_ch):
            assert np.array_equal(got[t][c], want[t][c]), (t, c)
            n_status += int((want[t][c] < 0).sum())
    assert n_status >= 5*n_ch                                   # carrier up, training, trained, carrier down, up again ...
    feed.close()
    bank.close()


def test_modem_feed_of_a_bank_in_step_above_4096_channels(built):
    """Every channel of a 4 200-channel bank fed the same line reports its carrier and training changes in the same tick, two and
    three reports a channel in an 800-sample tick: the feed's status list holds them all (it was sized one report a channel)."""
    import os
    from spandsp_amd import engine
    from test_oracle_pin import GOLDEN, use_golden_modem_tables
    use_golden_modem_tables()
    g = np.load(os.path.join(GOLDEN, "v29_9600.npz"))
    x = g["amp"]
    frame = 800                                                 # 100 ms ticks: training begins and the carrier is up inside one
    line = np.concatenate([x[:frame*8], np.zeros(frame, np.int16), x[:frame*5]])
    ticks = len(line)//frame
    serial = engine.V29Bank(2, 9600)
    want = []
    for t in range(ticks):
        blk = line[t*frame:(t + 1)*frame]
        serial.rx_host(np.stack([blk, blk]))
        want.append(serial.events()[1])
    serial.close()
    n_ch = 4200
    bank = engine.V29Bank(n_ch, 9600)
    feed = engine.ModemFeed(bank, frame, 9600, depth=3)
    got = []
    for t in range(ticks):
        feed.slot()[:, :frame] = line[t*frame:(t + 1)*frame]
        feed.commit(frame)
        if t >= 2:
            got.append(feed.collect())
    while feed.outstanding():
        got.append(feed.collect())
    most = 0
    for t in range(ticks):
        most = max(most, int((want[t] < 0).sum()))
        for c in (0, 1, 2047, 4095, 4096, n_ch - 1):
            assert np.array_equal(got[t][c], want[t]), (t, c)
    assert most >= 2                                            # a tick with two or more reports on every channel at once
    feed.close()
    bank.close()
