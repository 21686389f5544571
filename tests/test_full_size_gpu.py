"""Parity at BASELINE.json's full sizes for the configurations whose oracle is too slow for every channel: channels
are V distinct signals tiled over the bank; every replica must agree bit-for-bit with the first copy (a size-independent
property that also exercises every workgroup, wave and lane position), and the first V channels are checked against
the oracle.  (DTMF 65 536 channels: tests/test_tone_gpu.py::test_dtmf_full_size_replica_property.)"""
import os

import numpy as np
import pytest

import synth
from test_oracle_pin import GOLDEN, bits, use_golden_modem_tables

pytestmark = pytest.mark.gpu


def test_v29_16384_channels(built):
    """configs[3]: V.29 9600 bps, 16 384 channels."""
    from oracle import restated as orc
    from spandsp_amd import engine
    from test_v29_gpu import channel_signals
    use_golden_modem_tables()
    n_ch, V, n_frames = 16384, 64, 22                       # 22 frames = training + 1 500 samples of data
    base = channel_signals(9600, V, seed=77)[:, :n_frames*160]
    sig = np.tile(base, (n_ch//V, 1))
    bank = engine.V29Bank(n_ch, 9600)
    want = []
    for c in range(V):
        o = orc.V29(9600)
        per = []
        for k in range(n_frames):
            o.sink.clear()
            o.rx(base[c, k*160:(k + 1)*160])
            per.append(o.sink.events()["a"].astype(np.int8))
        want.append((per, o.snapshot()))
    total = 0
    for k in range(n_frames):
        bank.rx_host(sig[:, k*160:(k + 1)*160])
        ev = bank.events()
        for c in range(n_ch):
            assert np.array_equal(ev[c], want[c % V][0][k]), (k, c)
        total += sum(len(e) for e in ev[:V])
    assert total > 400*V//2
    for c in (0, 63, 64, 8191, n_ch - 1):
        f, w = bank.get_state(c)
        of, ow = want[c % V][1]
        assert np.array_equal(w, ow) and np.array_equal(bits(f), bits(of)), c
    bank.close()


@pytest.mark.parametrize("modem,bit_rate,n_frames", [("v29", 9600, 22), ("v27ter", 4800, 48), ("v17", 14400, 84)])
def test_quad_kernels_40037_channels(built, modem, bit_rate, n_frames):
    """A bank between the headline size and the full-wave kernels' (four lanes per channel up to 65 535 channels: several
    workgroups per CU one after the other, a last workgroup and a last wave that are not full): every channel's events
    against the oracle's for its signal, state words of a few."""
    from oracle import restated as orc
    from spandsp_amd import engine
    import importlib
    use_golden_modem_tables()
    n_ch, V = 40037, 53
    sigs = importlib.import_module("test_%s_gpu" % modem).channel_signals
    base = sigs(bit_rate, V, seed=79)[:, :n_frames*160]
    reps = (n_ch + V - 1)//V
    sig = np.tile(base, (reps, 1))[:n_ch]
    O = {"v29": orc.V29, "v27ter": orc.V27ter, "v17": orc.V17}[modem]
    B = {"v29": engine.V29Bank, "v27ter": engine.V27terBank, "v17": engine.V17Bank}[modem]
    bank = B(n_ch, bit_rate)
    want = []
    for c in range(V):
        o = O(bit_rate)
        per = []
        for k in range(n_frames):
            o.sink.clear()
            o.rx(base[c, k*160:(k + 1)*160])
            per.append(o.sink.events()["a"].astype(np.int8))
        want.append((per, o.snapshot()))
    total = 0
    for k in range(n_frames):
        bank.rx_host(sig[:, k*160:(k + 1)*160])
        ev = bank.events()
        for c in range(n_ch):
            assert np.array_equal(ev[c], want[c % V][0][k]), (k, c)
        total += sum(len(e) for e in ev[:V])
    assert total > 100*V
    for c in (0, 15, 16, 63, 64, 16383, 16384, 32767, 32768, n_ch - 2, n_ch - 1):
        f, w = bank.get_state(c)
        of, ow = want[c % V][1]
        assert np.array_equal(w, ow) and np.array_equal(bits(f), bits(of)), c
    bank.close()


def test_v29_full_wave_kernel_65797_channels(built):
    """Banks of 64 K channels and more run the full-wave kernel (four waves per workgroup sharing the tables, the RRC delay
    line as packed int16 pairs): a bank that does not fill its last workgroup nor its last wave, every channel against
    the oracle's run of its line, events of every frame and the state at the end."""
    from oracle import restated as orc
    from spandsp_amd import engine
    from test_v29_gpu import channel_signals
    use_golden_modem_tables()
    n_ch, V, n_frames = 65536 + 64*4 + 5, 61, 24
    base = channel_signals(9600, V, seed=78)[:, :n_frames*160]
    pick = (np.arange(n_ch)*7) % V
    bank = engine.V29Bank(n_ch, 9600)
    want = []
    for c in range(V):
        o = orc.V29(9600)
        per = []
        for k in range(n_frames):
            o.sink.clear()
            o.rx(base[c, k*160:(k + 1)*160])
            per.append(o.sink.events()["a"].astype(np.int8))
        want.append((per, o.snapshot()))
    total = 0
    for k in range(n_frames):
        bank.rx_host(base[pick, k*160:(k + 1)*160])           # (a frame at a time: 21 MB, not the whole call for every channel)
        ev = bank.events()
        for c in range(n_ch):
            assert np.array_equal(ev[c], want[pick[c]][0][k]), (k, c)
        total += sum(len(e) for e in ev[:V])
    assert total > 100*V
    for c in list(range(0, n_ch, 509)) + [65535, 65536, n_ch - 6, n_ch - 1]:
        f, w = bank.get_state(c)
        fo, wo = want[pick[c]][1]
        assert np.array_equal(w, wo), c
        assert np.array_equal(bits(f), bits(fo)), c


@pytest.mark.parametrize("bit_rate", [4800, 2400])
def test_v27ter_full_wave_kernel_65700_channels(built, bit_rate):
    """The same for V.27ter (both rates: their pulse shaping tables differ in size): the full-wave kernel on a bank that
    fills neither its last workgroup nor its last wave."""
    from oracle import restated as orc
    from spandsp_amd import engine
    from test_v27ter_gpu import channel_signals
    use_golden_modem_tables()
    n_ch, V, n_frames = 65536 + 64*2 + 36, 47, (48 if bit_rate == 4800 else 58)      # the training (0.7 s / 0.94 s) and some data
    base = channel_signals(bit_rate, V, seed=79)[:, :n_frames*160]
    pick = (np.arange(n_ch)*5) % V
    bank = engine.V27terBank(n_ch, bit_rate)
    want = []
    for c in range(V):
        o = orc.V27ter(bit_rate)
        per = []
        for k in range(n_frames):
            o.sink.clear()
            o.rx(base[c, k*160:(k + 1)*160])
            per.append(o.sink.events()["a"].astype(np.int8))
        want.append((per, o.snapshot()))
    total = 0
    for k in range(n_frames):
        bank.rx_host(base[pick, k*160:(k + 1)*160])           # (a frame at a time: 21 MB, not the whole call for every channel)
        ev = bank.events()
        for c in range(n_ch):
            assert np.array_equal(ev[c], want[pick[c]][0][k]), (k, c)
        total += sum(len(e) for e in ev[:V])
    assert total > 30*V
    for c in list(range(0, n_ch, 997)) + [65535, 65536, n_ch - 37, n_ch - 1]:
        f, w = bank.get_state(c)
        fo, wo = want[pick[c]][1]
        assert np.array_equal(w, wo), c
        assert np.array_equal(bits(f), bits(fo)), c


def test_v17_full_wave_kernel_65650_channels(built):
    """And for V.17 (trellis survivor memory in LDS, three waves per workgroup): 14 400 bps through its long training
    into data, on a bank that fills neither its last workgroup nor its last wave."""
    from oracle import restated as orc
    from spandsp_amd import engine
    from test_v17_gpu import channel_signals
    use_golden_modem_tables()
    n_ch, V, n_frames = 65536 + 64 + 50, 37, 84              # 84 frames: the 1.4 s of training and some data
    base = channel_signals(14400, V, seed=80)[:, :n_frames*160]
    pick = (np.arange(n_ch)*11) % V
    bank = engine.V17Bank(n_ch, 14400)
    want = []
    for c in range(V):
        o = orc.V17(14400)
        per = []
        for k in range(n_frames):
            o.sink.clear()
            o.rx(base[c, k*160:(k + 1)*160])
            per.append(o.sink.events()["a"].astype(np.int8))
        want.append((per, o.snapshot()))
    total = 0
    for k in range(n_frames):
        bank.rx_host(base[pick, k*160:(k + 1)*160])
        ev = bank.events()
        for c in range(n_ch):
            assert np.array_equal(ev[c], want[pick[c]][0][k]), (k, c)
        total += sum(len(e) for e in ev[:V])
    assert total > 100*V
    for c in list(range(0, n_ch, 1499)) + [65535, 65536, 65599, 65600, n_ch - 1]:
        f, w = bank.get_state(c)
        fo, wo = want[pick[c]][1]
        assert np.array_equal(w, wo), c
        assert np.array_equal(bits(f), bits(fo)), c


ST_PLAN = ([(400, 0, 700, 0)], [(1100, 0, 400, 600), (0, 0, 2800, 3200)], [(350, 440, 400, 0)],
           [(480, 620, 450, 550), (0, 0, 450, 550)], [(950, 0, 300, 0)], [(1400, 0, 300, 0)])
ST_LINES = ([(400, 0, 1500)], [(1100, 0, 500), (0, 0, 3000)], [(350, 440, 1200), (0, 0, 300)], [(480, 620, 500), (0, 0, 500)],
            [(950, 0, 330), (1400, 0, 330), (0, 0, 1000)], [(620, 0, 300), (0, 0, 200)])


def test_mixed_banks_131072_channels(built):
    """configs[2]: Bell MF + R2 MF + super_tone_rx() -- block decisions AND, for the super-tone third, the cadence matcher's tone
    and segment reports (super_tone_rx.c:164-228, :369-445) -- 131 072 channels in all."""
    from oracle import restated as orc
    from spandsp_amd import engine
    V = 128
    n_frames = [12, 12, 200]                         # 4 s for the super-tone lines: the ring-back cadence needs 3.5 s
    n_each = [43690, 43690, 43692]
    srcs = [synth.bell_mf_channels(V, n_frames[0]*160, 31)[0], synth.r2_mf_channels(V, n_frames[1]*160, 32, True)[0],
            synth.cadence_plan_channels(V, n_frames[2]*160, 33, ST_LINES)]
    # the call-progress plan of the third bank as a super-tone descriptor (the one of tests/super_tone_rx_tests.c:361-374 and
    # four more tones): 8 monitored frequencies
    desc = orc.SuperToneDesc()
    for tone in ST_PLAN:
        t = desc.add_tone()
        for f1, f2, lo, hi in tone:
            desc.add_element(t, f1, f2, lo, hi)
    fac = [float(f) for f in desc.fac]
    assert len(fac) == 8
    hz = [400, 1100, 350, 440, 480, 620, 950, 1400]                  # the order the descriptor met them in
    assert [engine.goertzel_fac(float(f)) for f in hz] == fac
    bins = {0: -1}
    bins.update({f: i for i, f in enumerate(hz)})
    banks = [engine.ToneBank(engine.BELL_MF, n_each[0]), engine.ToneBank(engine.R2_MF, n_each[1], r2_fwd=True),
             engine.ToneBank(engine.SUPER_TONE, n_each[2], bin_fac=fac)]
    banks[2].set_cadences([[(bins[f1], bins[f2], lo, hi) for f1, f2, lo, hi in t] for t in ST_PLAN], want_segments=True)
    hits = 0
    reports = segments = 0
    for kind in range(3):
        n = n_each[kind]
        reps = -(-n//V)
        sig = np.tile(srcs[kind], (reps, 1))[:n]
        per_frame = []
        per_frame_cad = []
        for k in range(n_frames[kind]):
            banks[kind].rx_host(sig[:, k*160:(k + 1)*160])
            per_frame.append(banks[kind].blocks())
            if kind == 2:
                per_frame_cad.append(banks[kind].cadence_events())
        # replica property: channel c and channel c % V report the same blocks (and the same cadence events)
        for k, b in enumerate(per_frame):
            first = b[b["channel"] < V]
            nb = len(first)//V
            assert len(b) == nb*n, (kind, k)
            rr = b.reshape(n, nb)
            ref_rows = np.stack([first[first["channel"] == c] for c in range(V)])
            for name in ("block", "hit", "code", "flags"):
                assert np.array_equal(rr[name], ref_rows[name][np.arange(n) % V]), (kind, k, name)
            hits += int((first["hit"] != 0).sum())
            if kind == 2:
                cad = per_frame_cad[k]
                for c in range(V, n):
                    assert cad[c] == cad[c % V], (k, c)
        # oracle on the first V channels: hit and code of every block, in order
        for c in range(V):
            if kind == 0:
                o = orc.BellMf(0)
            elif kind == 1:
                o = orc.R2Mf(True, True)
            else:
                o = orc.SuperTone(desc, True)
            blocks = []
            for k in range(n_frames[kind]):
                blocks.extend(o.rx(srcs[kind][c, k*160:(k + 1)*160]))
                if kind == 2:
                    want = [tuple(int(x) for x in e) for e in o.sink.events()]
                    o.sink.clear()
                    assert per_frame_cad[k][c] == want, (k, c, per_frame_cad[k][c], want)
                    reports += sum(1 for e in want if e[0] == 1)
                    segments += sum(1 for e in want if e[0] == 4)
            got = [(int(r["hit"]), int(r["code"])) for b in per_frame for r in b[b["channel"] == c]]
            assert got == [(int(x["hit"]), int(x["aux"])) for x in blocks], (kind, c)
            if kind == 0:
                digits = "".join(chr(int(r["code"])) for b in per_frame for r in b[b["channel"] == c] if r["flags"] & engine.BLK_REPORT)
                assert digits == o.get(), (c, digits)
    assert hits > 100
    assert reports > V and segments > 4*V, (reports, segments)
    for b in banks:
        b.close()


def test_echo_131072_channels(built):
    """configs[4], one GPU's shard: 128-tap echo cancellers, 131 072 channels."""
    from oracle import restated as orc
    from spandsp_amd import engine
    from test_echo_gpu import make_channels
    n_ch, V, n_frames = 131072, 64, 6
    tx, rx = make_channels(V, 160*40, 128, seed=404)
    tx, rx = tx[:, 160*30:160*(30 + n_frames)], rx[:, 160*30:160*(30 + n_frames)]        # a stretch with double talk in it
    dets = [orc.EchoCan(128, 0x01) for _ in range(V)]
    bank = engine.EchoBank(n_ch, 128, 0x01)
    txb = np.tile(tx, (n_ch//V, 1))
    rxb = np.tile(rx, (n_ch//V, 1))
    for k in range(n_frames):
        clean = bank.update_host(txb[:, k*160:(k + 1)*160], rxb[:, k*160:(k + 1)*160], True)
        want = np.stack([d.run(tx[c, k*160:(k + 1)*160], rx[c, k*160:(k + 1)*160], True) for c, d in enumerate(dets)])
        assert np.array_equal(clean.reshape(n_ch//V, V, 160), np.broadcast_to(want, (n_ch//V, V, 160))), k
    for c in (0, 63, 64, 70000, n_ch - 1):
        g = bank.get_state(c)
        o = dets[c % V].snapshot()
        assert np.array_equal(g["taps32"], o["taps32"]) and np.array_equal(g["history"], o["history"]), c
    bank.close()


def test_fsk_and_connect_tones_65536_channels(built):
    """The widened receivers at bank sizes that fill the chip: V distinct signals tiled over 65 536 channels; every
    replica's events and final state must equal the first copy's, and the first V channels equal the oracle."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n_ch, V = 65536, 64
    for what in ("fsk", "mct"):
        n_frames = 40 if what == "fsk" else 110         # a connect tone needs 0.4 - 0.55 s before it is declared
        if what == "fsk":
            base = synth.fsk_channels(V, 160*n_frames, 811, 1850, 1650, 30000)
            bank = engine.FskBank(engine.FSK_V21CH2, n_ch)
            orcs = [orc.Fsk(1, 1) for _ in range(V)]
        else:
            base = synth.connect_tone_channels(V, 160*n_frames, 812, "mix")
            bank = engine.MctBank(engine.MCT_FAX_CED_OR_PREAMBLE, n_ch)
            orcs = [orc.Mct(7) for _ in range(V)]
        sig = np.tile(base, (n_ch//V, 1))
        total = 0
        for k in range(n_frames):
            bank.rx_host(sig[:, k*160:(k + 1)*160])
            ev = bank.events()
            want = []
            for c in range(V):
                orcs[c].sink.clear()
                orcs[c].rx(base[c, k*160:(k + 1)*160])
                e = orcs[c].sink.events()
                want.append(e["a"].astype(np.int64) if what == "fsk" else np.stack([e["a"], e["b"]], 1).astype(np.int64).reshape(-1, 2))
            for c in range(n_ch):
                assert np.array_equal(ev[c].astype(np.int64), want[c % V]), (what, k, c)
            total += sum(len(w) for w in want)
        assert total > V//2
        for c in list(range(V)) + list(range(n_ch - V, n_ch)) + list(range(0, n_ch, 4099)):
            assert np.array_equal(bank.get_state(c), orcs[c % V].snapshot()), (what, c)
        bank.close()


def test_dtmf_sender_feeds_detector_65536_channels(built):
    """dtmf_tx bank -> HBM -> dtmf_rx bank at BASELINE configs[1]'s size without the samples leaving the device:
    every channel's detector reports exactly the digits its sender was given (a round trip that holds at any size)."""
    import ctypes
    from spandsp_amd import engine
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n, samples = 65536, 160
    rng = np.random.default_rng(13)
    keys = np.frombuffer(b"0123456789ABCD*#", np.uint8)
    ndig = rng.integers(1, 9, n)
    digs = keys[rng.integers(0, 16, (n, 8))]
    want = [bytes(digs[c, :ndig[c]]).decode() for c in range(n)]
    tx = engine.TxBank(engine.TX_DTMF, n)
    rx = engine.ToneBank(engine.DTMF, n)
    tx.set_stream(engine.lib().spangpu_bank_get_stream(rx.h))
    assert not tx.put_each(want).any()
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), n*samples*2) == 0
    got = [[] for _ in range(n)]
    for _ in range(8*840//samples + 4):
        tx.tx_device(buf, samples, samples)
        rx.rx_device(buf.value, samples)
        blk = rx.blocks()
        sel = blk[((blk["flags"] & engine.BLK_CHANGE) != 0) & (blk["code"] != 0)]
        for ch, code in zip(sel["channel"], sel["code"]):
            got[ch].append(chr(code))
    hip.hipFree(buf)
    bad = [c for c in range(n) if "".join(got[c]) != want[c]]
    assert not bad, (len(bad), bad[:5])


@pytest.mark.parametrize("modem,bit_rate,frames", [("v29", 9600, 40), ("v27ter", 4800, 70), ("v17", 14400, 95)])
def test_modem_round_trip_16384_channels(built, modem, bit_rate, frames):
    """BASELINE configs[3]'s size as a round trip that holds at any size: 16 384 transmitters (own LFSR seed each) ->
    HBM -> 16 384 receivers, the samples never leaving the device.  Every receiver must report training success and
    then deliver exactly its transmitter's bit stream (checked bit for bit on every 16th channel, and by a bit count
    and a spot comparison on all the others)."""
    import ctypes
    from spandsp_amd import engine
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n, samples = 16384, 160
    seeds = (((np.arange(n, dtype=np.uint64)*2654435761 + 99) & 0x7FFF) | 1).astype(np.uint32)
    tx = {"v29": engine.V29TxBank, "v27ter": engine.V27terTxBank, "v17": engine.V17TxBank}[modem](n, bit_rate, False, seeds)
    rx = {"v29": engine.V29Bank, "v27ter": engine.V27terBank, "v17": engine.V17Bank}[modem](n, bit_rate)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), n*samples*2) == 0
    chunks = [[] for _ in range(n)]
    for _ in range(frames):
        tx.tx_device(buf, samples, samples)
        tx.sync()
        rx.rx_device(buf, samples, samples)
        for c, e in enumerate(rx.events()):
            if len(e):
                chunks[c].append(e)
    hip.hipFree(buf)

    def lfsr(seed, count):
        st = int(seed)
        out = np.empty(count, np.int8)
        for i in range(count):
            b = ((st >> 14) ^ (st >> 13)) & 1
            st = ((st << 1) | b) & 0x7FFF
            out[i] = b
        return out
    counts = []
    for c in range(n):
        ev = np.concatenate(chunks[c]) if chunks[c] else np.zeros(0, np.int8)
        ok = np.nonzero(ev == -4)[0]                    # SIG_STATUS_TRAINING_SUCCEEDED
        assert len(ok) == 1, c
        data = ev[ok[0] + 1:]
        assert (data >= 0).all(), c                     # no carrier drop, no training failure afterwards
        counts.append(len(data))
        if c % 16 == 0:
            want = lfsr(seeds[c], len(data) + 400)
            hit = [k for k in range(400) if np.array_equal(want[k:k + 64], data[:64])]
            assert hit, c
            assert np.array_equal(want[hit[0]:hit[0] + len(data)], data), c
    counts = np.array(counts)
    assert counts.min() > 500 and counts.max() - counts.min() <= 2*bit_rate//2400 + 8      # every channel ran in step


def test_awgn_65536_channels(built):
    """The noise source bank at full size: a spread of channels against the oracle, and every channel against the
    properties the domain offers (its own level, zero mean, channels with different seeds uncorrelated, channels
    with the same seed identical)."""
    from oracle import restated as orc
    from spandsp_amd import engine
    n = 65536
    rng = np.random.default_rng(77)
    seeds = rng.integers(1, 250000, n).astype(np.int32)
    seeds[n//2:] = seeds[:n//2]                      # every generator has a twin
    levels = rng.uniform(-40.0, -15.0, n//2).astype(np.float32)
    levels = np.concatenate([levels, levels])
    bank = engine.AwgnBank(seeds, levels)
    frames = 12
    out = np.concatenate([bank.tx_host(160) for _ in range(frames)], axis=1)
    picks = list(range(0, n, 997)) + [n - 1]
    for c in picks:
        want = orc.Awgn(int(seeds[c]), float(levels[c])).gen(160*frames)
        assert np.array_equal(out[c], want), (c, np.count_nonzero(out[c] != want))
    assert np.array_equal(out[:n//2], out[n//2:])
    x = out[:n//2].astype(np.float64)
    dbm0 = 10.0*np.log10(np.mean(x*x, axis=1)/32768.0**2) + 3.14 + 3.02
    assert np.max(np.abs(dbm0 - levels[:n//2])) < 1.0          # 1920 samples: sigma of the estimate ~0.14 dB
    assert abs(np.mean(x/np.std(x, axis=1, keepdims=True))) < 0.01
    u = x/np.linalg.norm(x, axis=1, keepdims=True)
    distinct = np.unique(seeds[:n//2], return_index=True)[1][:2048]
    g = u[distinct] @ u[distinct].T
    np.fill_diagonal(g, 0.0)
    assert np.abs(g).max() < 0.2                               # 1/sqrt(1920) = 0.023 per pair, 4M pairs
