#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libspandsp_ref.so,
compiled from /root/reference/src by oracle/Makefile).  Run in the dev container:

    python tests/golden/make_golden.py

Fixtures are data only: int16 input signals (produced by the reference's own
transmitters / AWGN, or by tests/synth.py) and the reference receiver's outputs
(state snapshots as uint32 words, event streams, digit strings)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import synth  # noqa: E402
from oracle import ref  # noqa: E402
from test_oracle_pin import (SIGTONE_TX_CASES, sigtone_rx_run, sigtone_tx_run, sigtone_tx_script, ALL_FREQS, AWGN_CASES, QAM_GOLDEN_CASES, qam_golden_run, V29TX_CASES, V27TX_CASES, V17TX_CASES, v29tx_run, FSK_CASES, fsk_run, fsk_scenario, mct_run, mct_scenario, ECHO_CASES, V17_CASES, V27_CASES, V29_CASES, bits, build_st_desc, echo_scenario, st_signal,  # noqa: E402
                             tx_scenario, v17_scenario, v27ter_scenario, v29_run, v29_scenario)


def save(name, **kw):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **kw)
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in kw.items()})


def echo_goldens():
    """echo_can_update() of the real reference over tests/test_oracle_pin.py's scenario, every ECHO_CASES length and mode
    (32 ms to 128 ms tails): the clean samples, the final taps, history and control words"""
    import zlib
    for taps, mode in ECHO_CASES:
        tx, rx = echo_scenario(taps, seed=taps + mode)
        r = ref.EchoCan(taps, mode)
        clean = np.concatenate([r.run(tx[k:k + 160], rx[k:k + 160], True) for k in range(0, len(tx), 160)])
        s = r.snapshot()
        save("echo_%d_%02x" % (taps, mode), tx_crc=zlib.crc32(tx.tobytes()), rx_crc=zlib.crc32(rx.tobytes()),
             clean=clean, taps32=s["taps32"], taps16=s["taps16"], history=s["history"],
             fields=np.array(ref.ECHO_FIELDS), values=np.array([s[k] for k in ref.ECHO_FIELDS]))


def mitel_side1():
    """BASELINE configs[0]: Tests 2-7 of tests/dtmf_rx_tests.c on the real reference (signals from its tone_gen / awgn,
    answers from its dtmf_rx / dtmf_rx_get): every answer, a CRC of every signal, and the summary figures."""
    import mitel

    def burst(f1, l1, f2, l2, on_ms, off_ms):
        return ref.ToneGen(f1, l1, f2, l2, on_ms, off_ms, 0, 0, False).tx(1000)

    class Noise:
        def __init__(self, seed, level):
            self.cache = ref.awgn(seed, level, 1000*1000)
            self.pos = 0

        def gen(self, n):
            out = self.cache[self.pos:self.pos + n]
            self.pos += n
            return out
    run = mitel.Run(burst, Noise, ref.DtmfRx(0))
    res = run.run()
    save("mitel_side1", answers=np.frombuffer("|".join(run.log).encode("latin1"), np.uint8), calls=run.calls,
         signal_crc=np.uint32(run.crc), decode_ok=int(res["decode_ok"]), bandwidth=res["bandwidth"], twist=res["twist"],
         dynamic_range=res["dynamic_range"], guard_time_ms=res["guard_time_ms"], guard_responses=res["guard_responses"],
         snr_levels=res["snr_levels"], acceptable_snr_db=res["acceptable_snr_db"])


def dial_tone_tolerance():
    """dial_tone_tolerance_tests() of tests/dtmf_rx_tests.c on the real reference, dial tone filter off and on"""
    import mitel

    def burst(f1, l1, f2, l2, on_ms, off_ms):
        return ref.ToneGen(f1, l1, f2, l2, on_ms, off_ms, 0, 0, False).tx(1000)

    class Dial:
        def __init__(self, level):
            self.g = ref.ToneGen(350, level, 440, level, 1, 0, 0, 0, True)

        def gen(self, n):
            return self.g.tx(n)
    kw = {}
    for filt in (0, 1):
        run = mitel.DialToneRun(burst, Dial, ref.DtmfRx(0), bool(filt))
        res = run.run()
        kw["answers_%d" % filt] = np.frombuffer("|".join(run.log).encode("latin1"), np.uint8)
        kw["calls_%d" % filt] = run.calls
        kw["signal_crc_%d" % filt] = np.uint32(run.crc)
        kw["rounds_%d" % filt] = res["rounds"]
        kw["ratio_%d" % filt] = res["signal_to_dial_tone_db"]
    save("dtmf_dial_tone", **kw)


def flatten_callback_log(log):
    """CallbackRun's log as arrays: per dtmf_rx() call its position (or -1), then its callbacks, then its digits"""
    rows = []
    text = []
    for entry in log:
        at, (ev, t) = entry if isinstance(entry[0], int) else (-1, entry)
        rows.append((9, at, len(ev), len(t)))
        rows.extend(ev)
        text.append(t)
    return np.array(rows, np.int32).reshape(-1, 4), np.frombuffer("".join(text).encode("latin1"), np.uint8)


def dtmf_callbacks():
    """callback_function_tests() of tests/dtmf_rx_tests.c on the real reference: every callback of every call"""
    import mitel

    def burst(f1, l1, f2, l2, on_ms, off_ms):
        return ref.ToneGen(f1, l1, f2, l2, on_ms, off_ms, 0, 0, False).tx(1000)

    class Rx:
        def __init__(self, mode):
            self.d = ref.DtmfRx(mode)
            self.n = 0
            self.t = 0

        def rx(self, amp):
            self.d.rx(amp)

        def drain(self):
            ev = self.d.sink.events()
            txt = self.d.sink.text()
            new = [tuple(int(x) for x in e) for e in ev[self.n:]]
            t = txt[self.t:]
            self.n = len(ev)
            self.t = len(txt)
            return new, t
    run = mitel.CallbackRun(burst, Rx)
    rows, text = flatten_callback_log(run.run())
    save("dtmf_callbacks", rows=rows, text=text, signal_crc=np.uint32(run.crc))


def super_tone_range():
    """detection_range_tests() of tests/super_tone_rx_tests.c on the real reference: every callback, a CRC of the signal"""
    import zlib
    import st_range
    rx = ref.SuperToneRx(st_range.fill_descriptor(ref.SuperToneDesc()), True)
    crc = 0
    for level, frames in st_range.sweep():
        crc = zlib.crc32(frames.tobytes(), crc)
        for fr in frames:
            rx.rx(fr)
    ev = rx.sink.events()
    save("super_tone_range", events=np.array([tuple(int(x) for x in e) for e in ev], np.int32).reshape(-1, 4),
         signal_crc=np.uint32(crc))


def bell_mf_side1():
    """Tests 2-7 of tests/bell_mf_rx_tests.c on the real reference: every answer, a CRC of every signal, the summary figures"""
    import mf_side1

    def burst(f1, l1, f2, l2, on_ms, off_ms):
        return ref.ToneGen(f1, l1, f2, l2, on_ms, off_ms, 0, 0, False).tx(9999)

    class Noise:
        def __init__(self, seed, level):
            self.cache = ref.awgn(seed, level, 9000000)
            self.pos = 0

        def gen(self, n):
            out = self.cache[self.pos:self.pos + n]
            self.pos += n
            assert len(out) == n
            return out
    run = mf_side1.Run(burst, Noise, ref.BellMfRx(0))
    res = run.run()
    save("bell_mf_side1", answers=np.frombuffer("|".join(run.log).encode("latin1"), np.uint8), calls=run.calls,
         signal_crc=np.uint32(run.crc), decode_ok=int(res["decode_ok"]), bandwidth=res["bandwidth"], twist=res["twist"],
         dynamic_rounds=res["dynamic_rounds"], dynamic_range=res["dynamic_range"], guard_rounds=res["guard_rounds"],
         guard_time_ms=res["guard_time_ms"], snr_levels=res["snr_levels"], acceptable_snr_db=res["acceptable_snr_db"])


def r2_mf_side1():
    """Tests 2-7 of tests/r2_mf_rx_tests.c on the real reference, forward and backward tone sets"""
    import mf_side1

    def burst(f1, l1, f2, l2, on_ms, off_ms):
        return ref.ToneGen(f1, l1, f2, l2, on_ms, off_ms, 0, 0, False).tx(9999)

    class Noise:
        def __init__(self, seed, level):
            self.cache = ref.awgn(seed, level, 400000)
            self.pos = 0

        def gen(self, n):
            out = self.cache[self.pos:self.pos + n]
            self.pos += n
            assert len(out) == n
            return out
    for fwd in (True, False):
        run = mf_side1.R2Run(burst, Noise, ref.R2MfRx(fwd, use_callback=False), fwd)
        res = run.run()
        save("r2_mf_side1_%s" % ("fwd" if fwd else "back"), answers=np.array(run.log, np.uint8), calls=run.calls,
             signal_crc=np.uint32(run.crc), decode_ok=int(res["decode_ok"]), bandwidth=res["bandwidth"], twist=res["twist"],
             dynamic_rounds=res["dynamic_rounds"], dynamic_range=res["dynamic_range"], guard_rounds=res["guard_rounds"],
             guard_time_ms=res["guard_time_ms"], snr_levels=res["snr_levels"], acceptable_snr_db=res["acceptable_snr_db"])


def sigtone_goldens():
    """sig_tone_rx / sig_tone_tx of the real reference (oracle/ref.py: SigToneRx, SigToneTx)"""
    from test_oracle_pin import zlib_crc
    for tone_type, mode, seed in [(1, 0x40, 12), (2, 0xC0, 15), (3, 0x40, 17)]:
        x = synth.sig_tone_channels(4, 8000*5, seed, tone_type)[seed % 3]
        out, ev, snaps = sigtone_rx_run(ref.SigToneRx(tone_type, mode), x)
        assert len(ev) >= 4
        save("sigtone_rx_%d_%02x" % (tone_type, mode), amp=x, out=out, events=ev, snapshots=snaps)
    for tone_type, seed in SIGTONE_TX_CASES:
        script = np.array(sigtone_tx_script(seed), np.int32)
        out, snaps, n = sigtone_tx_run(ref.SigToneTx(tone_type, script), seed)
        save("sigtone_tx_%d" % tone_type, script=script, out_crc=np.uint32(zlib_crc(out)), out_len=np.int64(len(out)),
             out_head=out[:4000], snapshots=snaps, requests=np.int32(n))


def make_g711_encode():
    """linear_to_alaw() / linear_to_ulaw() of the reference (spandsp/g711.h:124-237) for every int16 value."""
    L = ref.lib()
    v = np.arange(-32768, 32768)
    save("g711_encode", alaw=np.array([L.glue_linear_to_alaw(int(x)) for x in v], np.uint8),
         ulaw=np.array([L.glue_linear_to_ulaw(int(x)) for x in v], np.uint8))


def modem_offset_goldens():
    """The three receivers of the real reference on lines with a carrier offset and a sample clock offset (tests/impair.py),
    three seconds each, and on lines too far off to train: input, every put_bit / status call, the final state words."""
    from test_oracle_pin import MODEM_OFFSET_CASES, MODEM_OFFSET_GOLDEN, modem_offset_expectations, modem_offset_name, modem_offset_receivers, modem_offset_scenario
    for i in MODEM_OFFSET_GOLDEN:
        case = MODEM_OFFSET_CASES[i]
        x = modem_offset_scenario(case)
        make_ref, _ = modem_offset_receivers(case[0])
        ev, f, w = v29_run(make_ref(case[1]), x, (160,))
        modem_offset_expectations(case, ev)
        save(modem_offset_name(case), amp=x, carrier_hz=case[4], ppm=case[5], events=ev.astype(np.int8), fwords=f, iwords=w)


def prims2_golden():
    """tests/prims2.py's cases on the real reference: a CRC-32 of every answer (whole-domain sweeps included) and the small answers whole"""
    import prims2
    from test_oracle_pin import prims2_reference
    save("prims2", **prims2.summary(prims2.run(prims2_reference())))


def make_g168():
    import zlib
    # the G.168 echo path models (test data of the reference: src/spandsp/g168models.h) and the known answer of
    # BASELINE.md section 2 (tests/g168.py: KNOWN_D2) from the real echo canceller
    import g168
    ms = list(range(2, 10))
    kw = {"models": np.array(ms), "ki": np.array([ref.g168_model(m)[1] for m in ms], np.float32)}
    for m in ms:
        kw["taps_%d" % m] = ref.g168_model(m)[0]
    erls = [-6.0, -9.0, -12.0, -15.0, -18.0, -21.0, -24.0]
    kw["gain_model"] = np.array([m for m in ms for _ in erls])
    kw["gain_erl"] = np.array([e for _ in ms for e in erls])
    kw["gain_value"] = np.array([ref.g168_gain(m, e) for m in ms for e in erls], np.float32)
    save("g168_models", **kw)
    K = g168.KNOWN_D2
    tx = ref.awgn(K["seed"], K["level_dbm0"], K["samples"])
    rx = ref.g168_line(K["model"], K["erl_db"], tx, np.zeros(len(tx), np.int16))
    r = ref.EchoCan(K["taps"], K["mode"])
    clean = r.run(tx, rx, False)
    save("g168_d2_known", tx_crc=zlib.crc32(tx.tobytes()), rx_crc=zlib.crc32(rx.tobytes()), clean_crc=zlib.crc32(clean.tobytes()),
         clean_last_second=clean[-8000:], erle_db=g168.erle_db(rx[-8000:], clean[-8000:]))


def main():
    assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
    L = ref.lib()
    mitel_side1()
    dial_tone_tolerance()
    dtmf_callbacks()
    super_tone_range()
    bell_mf_side1()
    r2_mf_side1()
    save("goertzel_fac", freq=np.array(ALL_FREQS, np.float32),
         fac_bits=np.array([bits([L.glue_goertzel_fac(f, 102)])[0] for f in ALL_FREQS], np.uint32))

    sig = ref.dtmf_tx("123A456B789C*0#D")
    x = ref.saturated_add(sig, ref.awgn(1234567, -30.0, len(sig)))
    x = np.concatenate([x, synth.dtmf_channels(2, 8000, seed=7)[0][1]])
    for name, mode, filt, chunk in [("dtmf_mode0", 0, 0, 160), ("dtmf_mode1", 1, 0, 37), ("dtmf_mode2", 2, 0, 160),
                                    ("dtmf_filter", 0, 1, 160)]:
        xx = x
        r = ref.DtmfRx(mode)
        if filt:
            t = np.arange(len(x))
            dial = 3000.0*np.sin(2*np.pi*350.0*t/8000.0) + 3000.0*np.sin(2*np.pi*440.0*t/8000.0)
            xx = np.clip(x + dial, -32768, 32767).astype(np.int16)
            r.parms(1, 9.0, 5.0, -39.0)
        snaps = []
        for k in range(0, len(xx), chunk):
            r.rx(xx[k:k + chunk])
            s = r.snapshot()
            snaps.append(np.concatenate([bits(s["v2"]), bits(s["v3"]), bits([s["energy"]]),
                                         np.array([s["current_sample"], s["duration"], s["last_hit"], s["in_digit"]], np.uint32)]))
        save(name, amp=xx, mode=mode, filter=filt, twist=9.0, reverse_twist=5.0, threshold=-39.0, chunk=chunk,
             snapshots=np.stack(snaps), events=r.sink.events(), text=r.sink.text(), digits=r.get())

    sig = ref.bell_mf_tx("*1234567890#ABC")
    x = ref.saturated_add(sig, ref.awgn(7, -35.0, len(sig)))
    r = ref.BellMfRx(1)
    snaps = []
    for k in range(0, len(x), 160):
        r.rx(x[k:k + 160])
        s = r.snapshot()
        snaps.append(np.concatenate([bits(s["v2"]), bits(s["v3"]), np.array([s["current_sample"]], np.uint32)]))
    save("bell_mf", amp=x, snapshots=np.stack(snaps), events=r.sink.events(), text=r.sink.text())

    for fwd in (True, False):
        sig = ref.r2_mf_tx("1234567890BCDEF", fwd)
        x = ref.saturated_add(sig, ref.awgn(9, -40.0, len(sig)))
        r = ref.R2MfRx(fwd)
        snaps = []
        for k in range(0, len(x), 160):
            r.rx(x[k:k + 160])
            s = r.snapshot()
            snaps.append(np.concatenate([bits(s["v2"]), bits(s["v3"]), np.array([s["current_sample"]], np.uint32)]))
        save("r2_mf_fwd" if fwd else "r2_mf_back", amp=x, snapshots=np.stack(snaps), events=r.sink.events())

    x = st_signal()
    d = build_st_desc(ref.SuperToneDesc)
    r = ref.SuperToneRx(d, True)
    for k in range(0, len(x), 160):
        r.rx(x[k:k + 160])
    save("super_tone", amp=x, fac_bits=bits(d.fac), events=r.sink.events())

    echo_goldens()
    make_g168()
    make_g711_encode()

    codes = np.arange(256)
    save("g711_decode", alaw=np.array([L.glue_alaw_to_linear(int(c)) for c in codes], np.int16),
         ulaw=np.array([L.glue_ulaw_to_linear(int(c)) for c in codes], np.int16))
    save("modem_tables", **ref.modem_tables())
    for bit_rate, seed, noise in V29_CASES:
        x = v29_scenario(bit_rate, seed, noise)
        ev, f, w = v29_run(ref.V29Rx(bit_rate), x, (160,))
        assert len(ev) > 1500 and -1 in ev
        save("v29_%d" % bit_rate, amp=x, events=ev.astype(np.int8), fwords=f, iwords=w)
    for bit_rate, seed, noise in V27_CASES:
        x = v27ter_scenario(bit_rate, seed, noise)
        ev, f, w = v29_run(ref.V27terRx(bit_rate), x, (160,))
        assert len(ev) > 400 and -4 in ev and -1 in ev
        save("v27ter_%d" % bit_rate, amp=x, events=ev.astype(np.int8), fwords=f, iwords=w)
    for bit_rate, seed, noise in V17_CASES:
        x = v17_scenario(bit_rate, seed, noise)
        ev, f, w = v29_run(ref.V17Rx(bit_rate), x, (160,))
        assert np.count_nonzero(ev == -4) == 2 and -1 in ev
        save("v17_%d" % bit_rate, amp=x, events=ev.astype(np.int8), fwords=f, iwords=w)
    for which, mode in FSK_CASES:
        x = fsk_scenario(which, mode)
        ev, snaps = fsk_run(ref.FskRx(which, mode), x, (160,))
        assert -2 in ev and -1 in ev
        save("fsk_%d_%d" % (which, mode), amp=x, events=ev, snapshots=snaps)
    for rx_type, tx_kind in [(1, 1), (2, 3), (2, 4), (7, "preamble"), (7, 5), (9, 9)]:
        x = mct_scenario(rx_type, tx_kind)
        ev, snaps, _ = mct_run(ref.MctRx(rx_type), x)
        assert len(ev) >= 2
        save("mct_%d_%s" % (rx_type, tx_kind), amp=x, events=ev, snapshots=snaps)
    sigtone_goldens()
    modem_offset_goldens()
    prims2_golden()
    kw = {"table": ref.v29_tx_table()}
    for i, (bit_rate, tep, seed) in enumerate(V29TX_CASES):
        kw["amp_%d" % i], kw["snaps_%d" % i] = v29tx_run(ref.V29Tx(bit_rate, tep, seed), seed)
    save("v29tx", **kw)
    t48, t24 = ref.v27ter_tx_tables()
    kw = {"table_4800": t48, "table_2400": t24}
    for i, (bit_rate, tep, seed) in enumerate(V27TX_CASES):
        kw["amp_%d" % i], kw["snaps_%d" % i] = v29tx_run(ref.V27terTx(bit_rate, tep, seed), seed, 2400 if bit_rate == 4800 else 4800)
    save("v27tertx", **kw)
    kw = {}
    for i, (bit_rate, tep, seed) in enumerate(V17TX_CASES):
        kw["amp_%d" % i], kw["snaps_%d" % i] = v29tx_run(ref.V17Tx(bit_rate, tep, seed), seed, 9600 if bit_rate != 9600 else 14400, 260, True)
    save("v17tx", **kw)
    kw = {}
    for name, bit_rate, seed, noise in QAM_GOLDEN_CASES:
        kw["%s_%d" % (name, bit_rate)] = qam_golden_run({"v29": ref.V29Rx, "v27ter": ref.V27terRx, "v17": ref.V17Rx}[name], name, bit_rate, seed, noise)
    save("modem_qam", **kw)
    kw = {}
    for i, (seed, level) in enumerate(AWGN_CASES):
        kw["amp_%d" % i] = ref.awgn(seed, level, 30001)
        kw["state_%d" % i] = ref.awgn_state_words(seed, level, 30001)
    save("awgn", **kw)
    from test_oracle_pin import FUNCTOR_CASES, functor_signal, zlib_crc
    kw = {}
    for i, (kind, n_blocks, seed) in enumerate(FUNCTOR_CASES):
        x = functor_signal(kind, seed, n_blocks)
        kw["crc_%d" % i] = np.uint32(zlib_crc(x))
        kw["hits_%d" % i] = ref.v18_tone_blocks(x)[0] if kind == 1 else ref.ademco_tone_blocks(x)
    save("tone_functors", **kw)
    amp, lens, puts = tx_scenario({"tone": ref.ToneGen, "dtmf": ref.DtmfTx, "bell": ref.BellMfTx, "r2": ref.R2MfTx}, 5)
    save("tx_sources", seed=5, amp=amp, lens=lens, puts=puts)


if __name__ == "__main__":
    if sys.argv[1:] == ["sigtone"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        sigtone_goldens()
    elif sys.argv[1:] == ["prims2"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        prims2_golden()
    elif sys.argv[1:] == ["modem_offsets"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        modem_offset_goldens()
    elif sys.argv[1:] == ["dtmf_callbacks"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        dtmf_callbacks()
    elif sys.argv[1:] == ["super_tone_range"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        super_tone_range()
    elif sys.argv[1:] == ["g711_encode"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        make_g711_encode()
    elif sys.argv[1:] == ["echo"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        echo_goldens()
    elif sys.argv[1:] == ["g168"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        make_g168()
    elif sys.argv[1:] == ["dial_tone"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        dial_tone_tolerance()
    elif sys.argv[1:] == ["r2_mf_side1"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        r2_mf_side1()
    elif sys.argv[1:] == ["bell_mf_side1"]:
        assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
        bell_mf_side1()
    else:
        main()
