"""Seeded synthetic telephony signals for the parity tests (numpy, host side).

Recipes follow SURVEY.md section 8(d): tone pairs at per-channel level / twist /
frequency offset with AWGN, cadenced like the reference's transmitters
(dtmf.c:67-69: 50 ms on / 55 ms off; bell_r2_mf.c:105-124: 68 ms on / 68 ms off).
"""
import numpy as np

DTMF_KEYS = "123A456B789C*0#D"
DTMF_ROW = [697.0, 770.0, 852.0, 941.0]
DTMF_COL = [1209.0, 1336.0, 1477.0, 1633.0]
BELL_FREQS = [700.0, 900.0, 1100.0, 1300.0, 1500.0, 1700.0]
R2_FWD = [1380.0, 1500.0, 1620.0, 1740.0, 1860.0, 1980.0]
R2_BACK = [1140.0, 1020.0, 900.0, 780.0, 660.0, 540.0]


def dbm0_to_amp(db):
    # a 3.14 dBm0 sine has peak 32768 (telephony.h:129); amplitude = peak
    return 32768.0*np.power(10.0, (np.asarray(db) - 3.14)/20.0)


def _finish(x):
    return np.clip(np.trunc(x), -32768, 32767).astype(np.int16)


def tone_pair_channels(n_ch, n_samples, pairs, seed, on=400, off=440, level_db=(-25.0, -7.0),
                       twist_db=(-4.0, 8.0), freq_off=0.015, noise_db=(-50.0, -25.0), quiet_fraction=0.25):
    """Generic cadenced two-tone generator.

    pairs: list of (f_low, f_high) tuples a channel may send.  Returns (int16
    [n_ch, n_samples], keys [n_ch, n_symbols] indices into pairs (-1 = silent channel)).
    """
    rng = np.random.default_rng(seed)
    period = on + off
    n_sym = n_samples//period + 2
    t = np.arange(n_samples, dtype=np.float64)
    keys = rng.integers(0, len(pairs), size=(n_ch, n_sym))
    start = rng.integers(0, period, size=n_ch)
    lvl = rng.uniform(level_db[0], level_db[1], size=n_ch)
    tw = rng.uniform(twist_db[0], twist_db[1], size=n_ch)
    fo = rng.uniform(-freq_off, freq_off, size=(n_ch, 2))
    nz = rng.uniform(noise_db[0], noise_db[1], size=n_ch)
    quiet = rng.uniform(size=n_ch) < quiet_fraction
    out = np.zeros((n_ch, n_samples), np.int16)
    pl = np.array([p[0] for p in pairs])
    ph = np.array([p[1] for p in pairs])
    for c in range(n_ch):
        rel = t - start[c]
        sym = np.floor(rel/period).astype(np.int64)
        inside = (rel >= 0) & ((rel - sym*period) < on)
        sym = np.clip(sym, 0, n_sym - 1)
        k = keys[c][sym]
        f1 = pl[k]*(1.0 + fo[c, 0])
        f2 = ph[k]*(1.0 + fo[c, 1])
        a_lo = dbm0_to_amp(lvl[c])
        a_hi = dbm0_to_amp(lvl[c] - tw[c])      # positive twist: low tone stronger
        x = a_lo*np.sin(2.0*np.pi*f1*t/8000.0) + a_hi*np.sin(2.0*np.pi*f2*t/8000.0 + 1.0)
        x = np.where(inside & (not quiet[c]), x, 0.0)
        sigma = dbm0_to_amp(nz[c])/np.sqrt(2.0)
        x = x + rng.normal(0.0, sigma, size=n_samples)
        out[c] = _finish(x)
        if quiet[c]:
            keys[c] = -1
    return out, keys


def dtmf_channels(n_ch, n_samples, seed):
    pairs = [(DTMF_ROW[i >> 2], DTMF_COL[i & 3]) for i in range(16)]
    return tone_pair_channels(n_ch, n_samples, pairs, seed)


def bell_mf_channels(n_ch, n_samples, seed):
    pairs = [(BELL_FREQS[i], BELL_FREQS[j]) for i in range(6) for j in range(i + 1, 6)]
    return tone_pair_channels(n_ch, n_samples, pairs, seed, on=544, off=544, level_db=(-20.0, -5.0),
                              twist_db=(-4.0, 4.0), noise_db=(-55.0, -35.0))


def r2_mf_channels(n_ch, n_samples, seed, fwd=True):
    fr = R2_FWD if fwd else R2_BACK
    pairs = [(fr[i], fr[j]) for i in range(6) for j in range(i + 1, 6)]
    return tone_pair_channels(n_ch, n_samples, pairs, seed, on=800, off=480, level_db=(-25.0, -5.0),
                              twist_db=(-5.0, 5.0), noise_db=(-55.0, -35.0))


def call_progress_channels(n_ch, n_samples, seed):
    """Cadenced call-progress tones for the super-tone detector (single and dual)."""
    pairs = [(400.0, 0.0), (1100.0, 0.0), (350.0, 440.0), (480.0, 620.0), (440.0, 480.0), (950.0, 0.0)]
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64)
    out = np.zeros((n_ch, n_samples), np.int16)
    for c in range(n_ch):
        k = rng.integers(0, len(pairs))
        on = int(rng.integers(1600, 6400))
        off = int(rng.integers(0, 6400))
        start = int(rng.integers(0, 2000))
        rel = t - start
        inside = (rel >= 0) & ((rel % (on + off)) < on)
        a = dbm0_to_amp(rng.uniform(-30.0, -8.0))
        f1, f2 = pairs[k]
        x = a*np.sin(2.0*np.pi*f1*t/8000.0)
        if f2 > 0.0:
            x = x + a*np.sin(2.0*np.pi*f2*t/8000.0 + 0.5)
        x = np.where(inside, x, 0.0)
        x = x + rng.normal(0.0, dbm0_to_amp(rng.uniform(-60.0, -40.0))/np.sqrt(2.0), size=n_samples)
        out[c] = _finish(x)
    return out


def fsk_channels(n_ch, n_samples, seed, freq_zero, freq_one, baud_x100, framed=False, data_bits=8, parity=0):
    """Phase-continuous FSK test signals (not the reference's modulator: any input serves a parity test).
    Each channel: silence, a burst of random bits (or of start/data/parity/stop characters when framed=True) at a
    random level with a little noise, silence again, sometimes a second burst.  parity: 0 none, 1 even, 2 odd."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_ch, n_samples), np.float64)
    spb = 800000.0/baud_x100
    for c in range(n_ch):
        pos = int(rng.integers(0, n_samples//6))
        while pos < n_samples - 200:
            length = int(rng.integers(n_samples//4, n_samples//2))
            nbits = int(length/spb) + 2
            if framed:
                bits = []
                while len(bits) < nbits:
                    ch = int(rng.integers(0, 1 << data_bits))
                    data = [(ch >> k) & 1 for k in range(data_bits)]
                    word = [0] + data
                    if parity:
                        p = sum(data) & 1
                        if rng.random() < 0.1:
                            p ^= 1                      # a parity error now and then
                        word.append(p if parity == 1 else p ^ 1)
                    word += [1]*int(rng.integers(1, 4))
                    if rng.random() < 0.05:
                        word[-1] = 0                    # and a framing error
                    bits += word
                bits = np.array(bits[:nbits])
            else:
                bits = rng.integers(0, 2, nbits)
            t = np.arange(length)
            f = np.where(bits[np.minimum((t/spb).astype(int), nbits - 1)] == 1, freq_one, freq_zero)
            ph = 2*np.pi*np.cumsum(f)/8000.0 + rng.uniform(0, 2*np.pi)
            amp = dbm0_to_amp(rng.uniform(-28.0, -6.0))
            end = min(n_samples, pos + length)
            out[c, pos:end] += amp*np.sin(ph[:end - pos])
            pos = end + int(rng.integers(300, n_samples//3))
        out[c] += rng.normal(0.0, rng.uniform(1.0, 40.0), n_samples)
        if c % 7 == 3:
            out[c] += 300.0                             # a DC offset: the power meter sits behind a DC blocker
    return _finish(out)


def connect_tone_channels(n_ch, n_samples, seed, kind):
    """Test signals for the modem connect tone detectors.  kind: 'cng' (1100 Hz, 0.5 s on / 3 s off), 'calling'
    (1300 Hz, 0.6 s / 1.75 s), 'ans' (2100 Hz), 'ans_pr' (2100 Hz with a phase reversal every 450 ms), 'ansam' /
    'ansam_pr' (the same with 20 % AM at 15 Hz), 'bell' (2225 Hz), 'preamble' (V.21 channel 2 carrying HDLC flags, then
    random bits), 'mix' (a different one of the above per channel).  Frequencies, levels, start times and noise vary per
    channel; some channels get a tone that is off frequency or too short to count."""
    rng = np.random.default_rng(seed)
    kinds = ["cng", "calling", "ans", "ans_pr", "ansam", "ansam_pr", "bell", "preamble"]
    out = np.zeros((n_ch, n_samples), np.float64)
    t = np.arange(n_samples)
    for c in range(n_ch):
        k = kinds[c % len(kinds)] if kind == "mix" else kind
        amp = dbm0_to_amp(rng.uniform(-30.0, -8.0))
        start = int(rng.integers(200, 4000))
        wrong = (c % 11 == 5)
        if k == "preamble":
            flags = int(rng.integers(2, 60)) if not wrong else 3
            bits = np.array([0, 1, 1, 1, 1, 1, 1, 0]*flags + list(rng.integers(0, 2, 400)))
            spb = 8000.0/300.0
            length = min(n_samples - start, int(len(bits)*spb))
            f = np.where(bits[np.minimum((np.arange(length)/spb).astype(int), len(bits) - 1)] == 1, 1650.0, 1850.0)
            out[c, start:start + length] = amp*np.sin(2*np.pi*np.cumsum(f)/8000.0)
        else:
            f0 = {"cng": 1100.0, "calling": 1300.0, "bell": 2225.0}.get(k, 2100.0)
            f0 *= (1.06 if wrong else float(rng.uniform(0.997, 1.003)))
            ph = 2*np.pi*f0*t/8000.0
            env = np.zeros(n_samples)
            if k in ("cng", "calling"):
                on, off = (4000, 24000) if k == "cng" else (4800, 14000)
                pos = start
                while pos < n_samples:
                    env[pos:pos + on] = 1.0
                    pos += on + off
            else:
                env[start:start + int(rng.integers(8000, 30000))] = 1.0
            if k in ("ans_pr", "ansam_pr"):
                ph = ph + np.pi*(((t - start)//3600) % 2)
            if k in ("ansam", "ansam_pr"):
                env = env*(1.0 + 0.2*np.sin(2*np.pi*15.0*t/8000.0))
            out[c] = amp*env*np.sin(ph)
        out[c] += rng.normal(0.0, rng.uniform(1.0, 30.0), n_samples)
    return _finish(out)


def sig_tone_channels(n_ch, n_samples, seed, tone_type):
    """Test signals for the in-band signalling tone receivers (sig_tone.c): bursts of the type's tone(s) -- 2280 Hz,
    2600 Hz, or 2400 / 2600 Hz alone and together -- from 2 ms to 1.2 s long with pauses of the same range, over
    speech-like noise whose level varies per channel (so that some channels never qualify and some chatter); level,
    exact frequency and start differ per channel, every seventh channel carries a tone the type does not listen for,
    every ninth noise only."""
    rng = np.random.default_rng(seed)
    freqs = {1: [(2280.0,)], 2: [(2600.0,)], 3: [(2400.0,), (2600.0,), (2400.0, 2600.0)]}[tone_type]
    out = np.zeros((n_ch, n_samples), np.float64)
    t = np.arange(n_samples)
    for c in range(n_ch):
        out[c] = rng.normal(0.0, dbm0_to_amp(rng.uniform(-60.0, -22.0)), n_samples)
        if c % 9 == 8:
            continue
        amp = dbm0_to_amp(rng.uniform(-24.0, -6.0))
        pos = int(rng.integers(0, 3000))
        while pos < n_samples:
            on = int(rng.choice([16, 40, 200, 700, 2500, 9600], p=[0.1, 0.1, 0.2, 0.25, 0.25, 0.1]))
            off = int(rng.choice([16, 80, 400, 1600, 4000], p=[0.1, 0.2, 0.3, 0.3, 0.1]))
            fs = freqs[int(rng.integers(0, len(freqs)))]
            if c % 7 == 6:
                fs = (1900.0,)
            seg = np.zeros(min(on, n_samples - pos))
            for f in fs:
                f = f*float(rng.uniform(0.998, 1.002))
                seg += amp*np.sin(2*np.pi*f*t[:len(seg)]/8000.0 + rng.uniform(0, 6.28))
            out[c, pos:pos + len(seg)] += seg
            pos += on + off
    return _finish(out)


def cadence_plan_channels(n_ch, n_samples, seed, plans):
    """Call-progress lines that follow a cadence for a few cycles and then change to another: plans = list of cycles,
    each a list of (f1, f2, ms) with 0 Hz = silent.  Segment lengths jitter by a few per cent; now and then a segment is
    cut short or stretched, which breaks the cadence a detector is following."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_ch, n_samples), np.int16)
    for c in range(n_ch):
        x = np.zeros(n_samples)
        pos = int(rng.integers(0, 3000))
        a = dbm0_to_amp(rng.uniform(-28.0, -9.0))
        while pos < n_samples:
            plan = plans[int(rng.integers(0, len(plans)))]
            for _ in range(int(rng.integers(1, 5))):
                for f1, f2, ms in plan:
                    n = int(8*ms*rng.uniform(0.97, 1.03))
                    if rng.uniform() < 0.08:
                        n = int(n*rng.choice([0.4, 1.9]))
                    n = min(n, n_samples - pos)
                    if n <= 0:
                        break
                    t = np.arange(pos, pos + n)
                    if f1:
                        x[pos:pos + n] += a*np.sin(2.0*np.pi*f1*t/8000.0)
                    if f2:
                        x[pos:pos + n] += a*np.sin(2.0*np.pi*f2*t/8000.0 + 0.7)
                    pos += n
        x += rng.normal(0.0, dbm0_to_amp(rng.uniform(-62.0, -45.0))/np.sqrt(2.0), size=n_samples)
        out[c] = _finish(x)
    return out
