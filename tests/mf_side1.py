"""The seven-test sequence of the reference's own Bell MF receiver test program (tests/bell_mf_rx_tests.c:236-560 -- the
Mitel DTMF procedure carried over to Bell MF: decode check, recognition bandwidth, twist, dynamic range, guard time,
signal to noise), restated as a driver over three pluggable parts:

    burst(f1, l1, f2, l2, on_ms, off_ms) -> int16 samples of one tone pair pulse plus its gap
                                            (tone_gen_descriptor_init + tone_gen(.., 9999): my_mf_gen_init /
                                            my_mf_generate, bell_mf_rx_tests.c:126-174)
    noise(seed, level_dbm0)              -> an object whose gen(n) returns n awgn() samples (awgn_init_dbm0(.., 1234567, ..))
    rx                                    -> an object with rx(amp) and get() (bell_mf_rx / bell_mf_rx_get)

make_golden.py runs it on the real reference (oracle/_ref) and stores what the receiver answered to every call plus a CRC
of every generated signal; the tests run it on the restated oracle (CPU) and on the bell_mf_rx() shim over the HIP engine
(GPU) and must reproduce both, call for call.  Test data only -- nothing here is part of the product."""
import zlib

import numpy as np

# bell_mf_rx_tests.c:95-113: f1, f2 (float), on time in ms; levels and off time do not vary
TONES = [(700.0, 900.0, 68), (700.0, 1100.0, 68), (900.0, 1100.0, 68), (700.0, 1300.0, 68), (900.0, 1300.0, 68),
         (1100.0, 1300.0, 68), (700.0, 1500.0, 68), (900.0, 1500.0, 68), (1100.0, 1500.0, 68), (1300.0, 1500.0, 68),
         (700.0, 1700.0, 68), (900.0, 1700.0, 68), (1100.0, 1700.0, 100), (1300.0, 1700.0, 68), (1500.0, 1700.0, 68)]
CODES = "1234567890CA*B#"
ALL = CODES


def tone_freqs(k, low_fudge, high_fudge):
    """my_mf_gen_init(): float frequency times (1.0 + float fudge) in double, truncated by the implicit conversion to
    tone_gen_descriptor_init()'s int parameters (bell_mf_rx_tests.c:141-151)"""
    f1 = np.float64(np.float32(TONES[k][0]))*(1.0 + np.float64(np.float32(low_fudge)))
    f2 = np.float64(np.float32(TONES[k][1]))*(1.0 + np.float64(np.float32(high_fudge)))
    return int(f1), int(f2)


class Run:
    def __init__(self, burst, noise, rx):
        self.burst = burst
        self.noise = noise
        self.rx = rx
        self.log = []               # what bell_mf_rx_get() returned after every bell_mf_rx() call
        self.crc = 0                # CRC-32 of every sample handed to the receiver, in order
        self.calls = 0

    def _send(self, digits, low_fudge, low_level, high_fudge, high_level, duration, gap, add=None):
        parts = []
        for d in digits:
            k = CODES.index(d)
            f1, f2 = tone_freqs(k, low_fudge, high_fudge)
            parts.append(self.burst(f1, low_level, f2, high_level, (3*duration)//2 if k == 12 else duration, gap))
        amp = np.ascontiguousarray(np.concatenate(parts), np.int16)
        if add is not None:
            n = add.gen(len(amp))
            amp = np.clip(amp.astype(np.int32) + n.astype(np.int32), -32768, 32767).astype(np.int16)      # sat_add16()
        self.crc = zlib.crc32(amp.tobytes(), self.crc)
        self.rx.rx(amp)
        got = self.rx.get()
        self.log.append(got)
        self.calls += 1
        return got

    def run(self):
        res = {}
        # Test 2 (:245-270): every digit ten times, 68 ms bursts, -3 dBm0 per tone
        ok = True
        for d in ALL:
            for _ in range(10):
                ok = ok and (self._send(d, 0.0, -3, 0.0, -3, 68, 68) == d)
        res["decode_ok"] = ok
        # Test 3 (:300-376): recognition bandwidth, every digit, low then high tone swept +-0.1 % .. 6 %
        bw = []
        for d in ALL:
            for which in (0, 1):
                counts = []
                for sweep in (range(1, 61), range(-1, -61, -1)):
                    n = 0
                    for i in sweep:
                        fu = np.float32(np.float64(np.float32(i))/1000.0)
                        n += len(self._send(d, fu if which == 0 else 0.0, -17, 0.0 if which == 0 else fu, -17, 68, 68))
                    counts.append(n)
                bw.append(tuple(counts))
        res["bandwidth"] = np.array(bw, np.int32)           # [digit*2 + (0 low, 1 high)] = (N+, N-)
        # Test 4 (:383-424): twist, the other tone from -5 to -25 dBm0 in 0.1 steps of an integer level (C division)
        tw = []
        for d in ALL:
            nplus = 0
            for i in range(-50, -251, -1):
                nplus += len(self._send(d, 0.0, -5, 0.0, int(i/10), 68, 68))
            nminus = 0
            for i in range(-50, -251, -1):
                nminus += len(self._send(d, 0.0, int(i/10), 0.0, -5, 68, 68))
            tw.append((nplus, nminus))
        res["twist"] = np.array(tw, np.int32)
        # Test 5 (:433-472): dynamic range, all digits from -50 to +3 dBm0 per tone, a hundred rounds per level or until
        # the first round that is not received whole
        nplus = nminus = -1000
        rounds = []
        for i in range(-50, 4):
            j = 0
            while j < 100:
                if self._send(ALL, 0.0, i, 0.0, i, 68, 68) != ALL:
                    break
                j += 1
            rounds.append(j)
            if j == 100:
                if nplus == -1000:
                    nplus = i
            elif nplus != -1000 and nminus == -1000:
                nminus = i
        res["dynamic_rounds"] = np.array(rounds, np.int32)
        res["dynamic_range"] = np.array([nplus, nminus - 1], np.int32)
        # Test 6 (:482-511): guard time, the pulses lengthened from 30 ms until five hundred rounds come through whole
        rounds = []
        i = 30
        while i < 62:
            j = 0
            while j < 500:
                if self._send(ALL, 0.0, -5, 0.0, -3, i, 68) != ALL:
                    break
                j += 1
            rounds.append(j)
            if j == 500:
                break
            i += 1
        res["guard_rounds"] = np.array(rounds, np.int32)
        res["guard_time_ms"] = i
        # Test 7 (:517-548): all digits at -3 dBm0 per tone over noise from -10 dBm0 down, five hundred rounds per level or
        # until the first one that is not received whole
        per_level = []
        i = -10
        while i > -50:
            src = self.noise(1234567, float(i))
            j = 0
            while j < 500:
                if self._send(ALL, 0.0, -3, 0.0, -3, 68, 68, add=src) != ALL:
                    break
                j += 1
            per_level.append((i, j))
            if j == 500:
                break
            i -= 1
        res["snr_levels"] = np.array(per_level, np.int32)
        res["acceptable_snr_db"] = -3 - i
        return res


# ---- MFC/R2 (tests/r2_mf_rx_tests.c:100-145, 246-587): the same procedure, one continuous pulse per call ------------
R2_FWD = [(1380.0, 1500.0), (1380.0, 1620.0), (1500.0, 1620.0), (1380.0, 1740.0), (1500.0, 1740.0), (1620.0, 1740.0),
          (1380.0, 1860.0), (1500.0, 1860.0), (1620.0, 1860.0), (1740.0, 1860.0), (1380.0, 1980.0), (1500.0, 1980.0),
          (1620.0, 1980.0), (1740.0, 1980.0), (1860.0, 1980.0)]
R2_BACK = [(1140.0, 1020.0), (1140.0, 900.0), (1020.0, 900.0), (1140.0, 780.0), (1020.0, 780.0), (900.0, 780.0),
           (1140.0, 660.0), (1020.0, 660.0), (900.0, 660.0), (780.0, 660.0), (1140.0, 540.0), (1020.0, 540.0),
           (900.0, 540.0), (780.0, 540.0), (660.0, 540.0)]
R2_CODES = "1234567890BCDEF"


class R2Run:
    """rx: an object with rx(amp) and get() -> the digit present (r2_mf_rx / r2_mf_rx_get); noise(seed, level) as above"""

    def __init__(self, burst, noise, rx, fwd):
        self.burst = burst
        self.noise = noise
        self.rx = rx
        self.tones = R2_FWD if fwd else R2_BACK
        self.log = []               # what r2_mf_rx_get() returned after every r2_mf_rx() call
        self.crc = 0
        self.calls = 0

    def _send(self, d, low_fudge, low_level, high_fudge, high_level, duration, add=None):
        f1, f2 = self.tones[R2_CODES.index(d)]
        a = int(np.float64(np.float32(f1))*(1.0 + np.float64(np.float32(low_fudge))))
        b = int(np.float64(np.float32(f2))*(1.0 + np.float64(np.float32(high_fudge))))
        amp = np.ascontiguousarray(self.burst(a, low_level, b, high_level, duration, 0), np.int16)
        if add is not None:
            n = add.gen(len(amp))
            amp = np.clip(amp.astype(np.int32) + n.astype(np.int32), -32768, 32767).astype(np.int16)
        self.crc = zlib.crc32(amp.tobytes(), self.crc)
        self.rx.rx(amp)
        got = int(self.rx.get())
        self.log.append(got)
        self.calls += 1
        return got

    def run(self):
        res = {}
        # Test 2 (:259-283)
        ok = True
        for d in R2_CODES:
            for _ in range(10):
                ok = ok and (self._send(d, 0.0, -3, 0.0, -3, 68) == ord(d))
        res["decode_ok"] = ok
        # Test 3 (:313-396)
        bw = []
        for d in R2_CODES:
            for which in (0, 1):
                counts = []
                for sweep in (range(1, 61), range(-1, -61, -1)):
                    n = 0
                    for i in sweep:
                        fu = np.float32(np.float64(np.float32(i))/1000.0)
                        n += (self._send(d, fu if which == 0 else 0.0, -17, 0.0 if which == 0 else fu, -17, 68) == ord(d))
                    counts.append(n)
                bw.append(tuple(counts))
        res["bandwidth"] = np.array(bw, np.int32)
        # Test 4 (:403-448)
        tw = []
        for d in R2_CODES:
            nplus = sum(self._send(d, 0.0, -5, 0.0, int(i/10), 68) == ord(d) for i in range(-50, -251, -1))
            nminus = sum(self._send(d, 0.0, int(i/10), 0.0, -5, 68) == ord(d) for i in range(-50, -251, -1))
            tw.append((nplus, nminus))
        res["twist"] = np.array(tw, np.int32)
        # Test 5 (:455-500): per level, digit after digit a hundred times each until one is missed
        nplus = nminus = -1000
        rounds = []
        for i in range(-50, 4):
            j = 0
            for d in R2_CODES:
                j = 0
                while j < 100:
                    if self._send(d, 0.0, i, 0.0, i, 68) != ord(d):
                        break
                    j += 1
                if j < 100:
                    break
            rounds.append(j)
            if j == 100:
                if nplus == -1000:
                    nplus = i
            elif nplus != -1000 and nminus == -1000:
                nminus = i
        res["dynamic_rounds"] = np.array(rounds, np.int32)
        res["dynamic_range"] = np.array([nplus, nminus - 1], np.int32)
        # Test 6 (:506-542)
        rounds = []
        i = 30
        while i < 62:
            j = 0
            for d in R2_CODES:
                j = 0
                while j < 500:
                    if self._send(d, 0.0, -5, 0.0, -3, i) != ord(d):
                        break
                    j += 1
                if j < 500:
                    break
            rounds.append(j)
            if j == 500:
                break
            i += 1
        res["guard_rounds"] = np.array(rounds, np.int32)
        res["guard_time_ms"] = i
        # Test 7 (:548-587): a fresh noise source (same seed) for every digit of every level
        per_level = []
        i = -3
        while i > -50:
            j = 0
            for d in R2_CODES:
                src = self.noise(1234567, float(i))
                j = 0
                while j < 500:
                    if self._send(d, 0.0, -3, 0.0, -3, 68, add=src) != ord(d):
                        break
                    j += 1
                if j < 500:
                    break
            per_level.append((i, j))
            if j == 500:
                break
            i -= 1
        res["snr_levels"] = np.array(per_level, np.int32)
        res["acceptable_snr_db"] = -3 - i
        return res
