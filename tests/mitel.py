"""BASELINE configs[0]: the Mitel CM7291 side-1 sequence of the reference's own DTMF receiver test program
(tests/dtmf_rx_tests.c:357-655 -- Tests 2 to 7: decode check, recognition bandwidth, twist, dynamic range, guard time,
signal to noise), restated as a driver over three pluggable parts:

    burst(f1, l1, f2, l2, on_ms, off_ms) -> int16 samples of one tone pair pulse plus its gap
                                            (tone_gen_descriptor_init + tone_gen(.., 1000): my_dtmf_gen_init /
                                            my_dtmf_generate, dtmf_rx_tests.c:165-217)
    noise(seed, level_dbm0)              -> an object whose gen(n) returns n awgn() samples (awgn_init_dbm0(.., 1234567, ..))
    rx                                    -> an object with rx(amp) and get() (dtmf_rx / dtmf_rx_get)

make_golden.py runs it on the real reference (oracle/_ref) and stores what the receiver answered to every call plus a CRC
of every generated signal; the tests run it on the restated oracle (CPU) and on the dtmf_rx() shim over the HIP engine
(GPU) and must reproduce both, call for call.  Test data only -- nothing here is part of the product."""
import zlib

import numpy as np

ROW = [697.0, 770.0, 852.0, 941.0]
COL = [1209.0, 1336.0, 1477.0, 1633.0]
POSITIONS = "123A456B789C*0#D"


def _f32(x):
    return np.float32(x)


def tone_freqs(digit, low_fudge, high_fudge):
    """The two integer frequencies the test program hands to tone_gen_descriptor_init(): float products, truncated by the
    implicit conversion to its int parameters (dtmf_rx_tests.c:178-188)."""
    k = POSITIONS.index(digit)
    f1 = _f32(ROW[k >> 2])*(_f32(1.0) + _f32(low_fudge))
    f2 = _f32(COL[k & 3])*(_f32(1.0) + _f32(high_fudge))
    return int(f1), int(f2)


class Run:
    def __init__(self, burst, noise, rx):
        self.burst = burst
        self.noise = noise
        self.rx = rx
        self.log = []               # what dtmf_rx_get() returned after every dtmf_rx() call
        self.crc = 0                # CRC-32 of every sample handed to the receiver, in order
        self.calls = 0

    def _send(self, amp):
        amp = np.ascontiguousarray(amp, np.int16)
        self.crc = zlib.crc32(amp.tobytes(), self.crc)
        self.rx.rx(amp)
        got = self.rx.get()
        self.log.append(got)
        self.calls += 1
        return got

    def _pulse(self, digit, low_fudge, low_level, high_fudge, high_level, on_ms, off_ms, add=None):
        f1, f2 = tone_freqs(digit, low_fudge, high_fudge)
        amp = self.burst(f1, low_level, f2, high_level, on_ms, off_ms)
        if add is not None:
            n = add.gen(len(amp))
            amp = np.clip(amp.astype(np.int32) + n.astype(np.int32), -32768, 32767).astype(np.int16)      # sat_add16()
        return self._send(amp)

    def run(self):
        res = {}
        # Test 2 (:359-384): every digit ten times, 50 ms bursts, -4 dBm0 per tone
        ok = True
        for d in POSITIONS:
            for _ in range(10):
                ok = ok and (self._pulse(d, 0.0, -4, 0.0, -4, 50, 50) == d)
        res["decode_ok"] = ok
        # Test 3 (:415-480): recognition bandwidth, digits 1 5 9 D, low then high tone swept +-0.1 % .. 6 %
        bw = []
        for d in "159D":
            for which in (0, 1):
                nplus = 0
                for i in range(1, 61):
                    fu = _f32(i)/_f32(1000.0)
                    nplus += len(self._pulse(d, fu if which == 0 else 0.0, -17, 0.0 if which == 0 else fu, -17, 50, 50))
                nminus = 0
                for i in range(-1, -61, -1):
                    fu = _f32(i)/_f32(1000.0)
                    nminus += len(self._pulse(d, fu if which == 0 else 0.0, -17, 0.0 if which == 0 else fu, -17, 50, 50))
                bw.append((nplus, nminus))
        res["bandwidth"] = np.array(bw, np.int32)           # [digit*2 + (0 low, 1 high)] = (N+, N-)
        # Test 4 (:507-550): twist, the other tone attenuated from -3 to -23 dBm0 in 0.1 steps of an integer level (C division)
        tw = []
        for d in "159D":
            nplus = 0
            for i in range(-30, -231, -1):
                nplus += len(self._pulse(d, 0.0, -3, 0.0, int(i/10), 50, 50))
            nminus = 0
            for i in range(-30, -231, -1):
                nminus += len(self._pulse(d, 0.0, int(i/10), 0.0, -3, 50, 50))
            tw.append((nplus, nminus))
        res["twist"] = np.array(tw, np.int32)               # per digit (normal, reverse) in 1/10 dB
        # Test 5 (:563-585): dynamic range, digit 1 from +3 down to -50 dBm0 per tone
        n = 0
        for i in range(3, -51, -1):
            n += len(self._pulse("1", 0.0, i, 0.0, i, 50, 50))
        res["dynamic_range"] = n
        # Test 6 (:597-610): guard time, the pulse shortened from 49 ms to 10 ms (integer ms: C division of i by 10)
        n = 0
        for i in range(490, 99, -1):
            n += len(self._pulse("1", 0.0, -3, 0.0, -3, i//10, 50))
        res["guard_time_ms"] = (500 - n)//10
        res["guard_responses"] = n
        # Test 7 (:625-655): digit 1 at -4 dBm0 per tone over noise, from -13 dBm0 down, a thousand pulses per level or
        # until the first one the receiver misses
        per_level = []
        j = -13
        while j > -50:
            src = self.noise(1234567, float(j))
            i = 0
            while i < 1000:
                if len(self._pulse("1", 0.0, -4, 0.0, -4, 50, 50, add=src)) != 1:
                    break
                i += 1
            per_level.append((j, i))
            if i == 1000:
                break
            j -= 1
        res["snr_levels"] = np.array(per_level, np.int32)
        res["acceptable_snr_db"] = -4 - j
        return res


def c_int_div(a, b):
    """C integer division (truncation toward zero)."""
    return int(a/b)


class DialToneRun:
    """dial_tone_tolerance_tests() of the same program (dtmf_rx_tests.c:744-800): all sixteen digits at -15 dBm0 per tone,
    50 ms on / 50 ms off, over a continuous 350 Hz + 440 Hz dial tone whose level rises from -30 dBm0 per tone until a
    round of ten is no longer received whole -- with the receiver's dial tone filter off (the program's default) or on.

        burst(f1, l1, f2, l2, on_ms, off_ms) as above;  dial(level) -> an object whose gen(n) returns the next n samples
        of the dial tone (tone_gen_descriptor_init(350, level, 440, level, 1, 0, 0, 0, true));  rx with rx(amp), get() and
        parms(filter_dialtone, twist, reverse_twist, threshold) = dtmf_rx_parms()"""

    def __init__(self, burst, dial, rx, use_filter):
        self.burst = burst
        self.dial = dial
        self.rx = rx
        self.use_filter = use_filter
        self.log = []
        self.crc = 0
        self.calls = 0

    def run(self):
        if self.use_filter:
            self.rx.parms(1, -1.0, -1.0, -99.0)            # (:756-757)
        digits = np.concatenate([self.burst(*tone_freqs(d, 0.0, 0.0)[:1], -15, tone_freqs(d, 0.0, 0.0)[1], -15, 50, 50)
                                 for d in POSITIONS])
        rounds = []
        j = -30
        while j < -3:
            tone = self.dial(j)
            i = 0
            while i < 10:
                amp = np.clip(digits.astype(np.int32) + tone.gen(len(digits)).astype(np.int32), -32768, 32767).astype(np.int16)
                self.crc = zlib.crc32(amp.tobytes(), self.crc)
                self.rx.rx(amp)
                got = self.rx.get()
                self.log.append(got)
                self.calls += 1
                if len(got) != len(POSITIONS):
                    break
                i += 1
            rounds.append(i)
            if i != 10:
                break
            j += 1
        return {"rounds": np.array(rounds, np.int32), "signal_to_dial_tone_db": -15 - j}


class CallbackRun:
    """callback_function_tests() of the same program (dtmf_rx_tests.c:805-893): one to nine rounds of all sixteen digits
    (-10 dBm0 per tone, 50 ms on / 50 ms off) handed to a receiver with a digits callback in ONE dtmf_rx() call per
    round count -- up to 115 200 samples a call -- and then to a receiver with a realtime callback in 160-sample chunks.

        make_rx(mode) -> an object with rx(amp) and drain() -> (the callback calls since the last drain as tuples
                         (kind, a, b, c): kind 2 = digits callback (a = len), kind 1 = realtime callback (a = code,
                         b = level, c = delay); the digits delivered since the last drain as a string)
    The log is, per dtmf_rx() call, what drain() returned."""

    def __init__(self, burst, make_rx):
        self.burst = burst
        self.make_rx = make_rx
        self.log = []
        self.crc = 0

    def run(self):
        one = np.concatenate([self.burst(*tone_freqs(d, 0.0, 0.0)[:1], -10, tone_freqs(d, 0.0, 0.0)[1], -10, 50, 50)
                              for d in POSITIONS])
        rx = self.make_rx(1)
        for i in range(1, 10):
            amp = np.ascontiguousarray(np.tile(one, i))
            self.crc = zlib.crc32(amp.tobytes(), self.crc)
            rx.rx(amp)
            self.log.append(rx.drain())
        rx = self.make_rx(2)
        for i in range(1, 10):
            amp = np.ascontiguousarray(np.tile(one, i))
            for k in range(0, len(amp), 160):
                rx.rx(amp[k:k + 160])
                ev = rx.drain()
                if ev[0]:
                    self.log.append((k, ev))
        return self.log
