"""The detection range test of the reference's super-tone receiver test program (tests/super_tone_rx_tests.c:436-471):
350 Hz + 440 Hz from the integer DDS, swept from -80 to -1 dBm0 per tone, a hundred 160-sample chunks per level, into a
receiver built on that program's own two-tone descriptor (:361-374) with both callbacks installed (:545-551).  Test
data only."""
import numpy as np

QUARTER = np.array([int(round(32767.0*np.sin(i*np.pi/512.0))) for i in range(257)], np.int32)


def dds_lookup(phase):
    """dds_lookup(), src/dds_int.c:340-355, on an array of 32 bit phases"""
    p = (phase >> np.uint32(22)).astype(np.int64)
    step = p & 255
    step = np.where(p & 256, 256 - step, step)
    amp = QUARTER[step]
    return np.where(p & 512, -amp, amp)


def fill_descriptor(d):
    """super_tone_rx_fill_descriptor(), super_tone_rx_tests.c:361-374"""
    t = d.add_tone()
    d.add_element(t, 400, 0, 700, 0)
    t = d.add_tone()
    d.add_element(t, 1100, 0, 400, 600)
    d.add_element(t, 0, 0, 2800, 3200)
    return d


def sweep():
    """Yields (level, [100, 160] int16): detection_range_tests(), :436-471.  dds_phase_rate(f) = (int32)(f*2^32/8000) and
    dds_scaling_dbm0(level) = (int16)(10^((level - 3.14)/20)*32767) in binary32 (dds_int.c:316-331); the phases run on
    from level to level."""
    inc = [np.uint32(np.int32(np.float32(f)*np.float32(65536.0)*np.float32(65536.0)/np.float32(8000.0))) for f in (350.0, 440.0)]
    phase = [np.uint32(0), np.uint32(0)]
    n = 100*160
    k = np.arange(n, dtype=np.uint64)
    for level in range(-80, 0):
        scale = int(np.int16(np.float32(np.power(np.float32(10.0), (np.float32(level) - np.float32(3.14))/np.float32(20.0)))*np.float32(32767.0)))
        out = np.zeros(n, np.int64)
        for t in range(2):
            ph = ((np.uint64(phase[t]) + k*np.uint64(inc[t])) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            out = ((out + ((dds_lookup(ph)*scale) >> 15) + 32768) & 0xFFFF) - 32768        # int16 arithmetic
            phase[t] = np.uint32((np.uint64(phase[t]) + np.uint64(n)*np.uint64(inc[t])) & np.uint64(0xFFFFFFFF))
        yield level, out.astype(np.int16).reshape(100, 160)
