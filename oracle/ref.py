"""ctypes binding of oracle/_ref/libspandsp_ref.so -- the REAL reference, compiled
from /root/reference/src by oracle/Makefile (sources are not copied here).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The .so is built in the
dev container and travels to the GPU box with the snapshot; /root/reference
itself is never read at run time.
"""
import contextlib
import ctypes as C
import os

import numpy as np

from . import REF_SO, REF_FAST_SO

EVENT_DTYPE = np.dtype([("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("c", "<i4")])

class _Cf(C.Structure):
    """complexf_t (spandsp/complex.h), as returned by value"""
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


_libs = {}
_flavour = "strict"


def available():
    return os.path.exists(REF_SO)


@contextlib.contextmanager
def flavour(name):
    """Everything inside the block talks to the named build of the reference: "strict" (-O2 -ffp-contract=off: the oracle
    every parity test uses) or "fast" (as the library ships: -O2 -ffast-math -msse2 with its SSE2 paths -- for the
    cpu_baseline legs to time, never to compare results with).  Objects made inside the block belong to that build."""
    global _flavour
    assert name in ("strict", "fast")
    old = _flavour
    _flavour = name
    try:
        yield
    finally:
        _flavour = old


def lib():
    if _flavour not in _libs:
        L = C.CDLL(REF_SO if _flavour == "strict" else REF_FAST_SO)
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        sigs = {
            # glue
            "glue_sink_new": (vp, []), "glue_sink_free": (None, [vp]), "glue_sink_clear": (None, [vp]),
            "glue_sink_count": (ci, [vp]), "glue_sink_events": (vp, [vp]),
            "glue_sink_ndigits": (ci, [vp]), "glue_sink_digits": (C.c_char_p, [vp]),
            "glue_dtmf_rx_new": (vp, [vp, ci, ci]),
            "glue_dtmf_rx_snapshot": (None, [vp, vp, vp]),
            "glue_dtmf_rx_consts": (None, [vp, vp]),
            "glue_bell_mf_rx_new": (vp, [vp, ci]),
            "glue_bell_mf_rx_snapshot": (None, [vp, vp, vp]),
            "glue_r2_mf_rx_new": (vp, [vp, ci, ci]),
            "glue_r2_mf_rx_snapshot": (None, [vp, vp, vp]),
            "glue_super_tone_rx_new": (vp, [vp, vp, ci]),
            "glue_super_tone_desc_bins": (ci, [vp, vp]),
            "glue_super_tone_rx_snapshot": (None, [vp, vp, vp]),
            "glue_goertzel_fac": (cf, [cf, ci]),
            "glue_goertzel_new": (vp, [cf, ci]),
            "glue_goertzel_snapshot": (None, [vp, vp, vp]),
            "glue_install_padded_allocator": (None, [ci]),
            "glue_echo_run": (None, [vp, vp, vp, vp, ci, ci]),
            "glue_echo_taps": (ci, [vp]),
            "glue_echo_snapshot": (None, [vp, vp, vp, vp, vp]),
            "glue_v29_rx_tap_qam": (None, [vp, vp]), "glue_v27ter_rx_tap_qam": (None, [vp, vp]), "glue_v17_rx_tap_qam": (None, [vp, vp]),
            "glue_v29_rx_new": (vp, [ci, vp]), "glue_v27ter_rx_new": (vp, [ci, vp]), "glue_v17_rx_new": (vp, [ci, vp]),
            "glue_v29_tx_new": (vp, [ci, ci, vp]), "glue_v27ter_tx_new": (vp, [ci, ci, vp]), "glue_v17_tx_new": (vp, [ci, ci, vp]),
            "glue_sizeof": (ci, [C.c_char_p]),
            "glue_mt_rx": (C.c_double, [ci, vp, vp, ci, ci, ci, C.c_longlong, C.c_longlong, ci, ci]),
            "glue_mt_echo": (C.c_double, [vp, vp, vp, vp, ci, ci, ci, C.c_longlong, C.c_longlong, ci, ci]),
            "glue_modem_tables": (None, [vp, vp, vp, vp, vp, vp]),
            "glue_v29_rx_snapshot": (None, [vp, vp, vp]),
            "glue_v27ter_rx_snapshot": (None, [vp, vp, vp]),
            "glue_v17_rx_snapshot": (None, [vp, vp, vp]),
            "glue_v17_tables": (None, [vp, vp, vp, vp]),
            "glue_v17_constellations": (None, [vp]),
            "glue_v17_constel_maps": (None, [vp, vp]),
            "v17_rx": (ci, [vp, vp, ci]), "v17_rx_free": (ci, [vp]), "v17_rx_restart": (ci, [vp, ci, ci]),
            "v17_tx": (ci, [vp, vp, ci]), "v17_tx_free": (ci, [vp]), "v17_tx_power": (None, [vp, cf]),
            "v17_tx_restart": (ci, [vp, ci, ci, ci]),
            "glue_v27ter_tables": (None, [vp, vp, vp, vp]),
            # reference public API (src/spandsp/*.h)
            "dtmf_rx": (ci, [vp, vp, ci]), "dtmf_rx_get": (C.c_size_t, [vp, C.c_char_p, ci]),
            "dtmf_rx_status": (ci, [vp]), "dtmf_rx_fillin": (ci, [vp, ci]),
            "dtmf_rx_parms": (None, [vp, ci, cf, cf, cf]), "dtmf_rx_free": (ci, [vp]),
            "dtmf_tx_init": (vp, [vp, vp, vp]), "dtmf_tx_put": (ci, [vp, C.c_char_p, ci]),
            "dtmf_tx": (ci, [vp, vp, ci]), "dtmf_tx_set_level": (None, [vp, ci, ci]),
            "dtmf_tx_set_timing": (None, [vp, ci, ci]), "dtmf_tx_free": (ci, [vp]),
            "bell_mf_rx": (ci, [vp, vp, ci]), "bell_mf_rx_get": (C.c_size_t, [vp, C.c_char_p, ci]),
            "bell_mf_rx_free": (ci, [vp]),
            "bell_mf_tx_init": (vp, [vp]), "bell_mf_tx_put": (ci, [vp, C.c_char_p, ci]),
            "bell_mf_tx": (ci, [vp, vp, ci]), "bell_mf_tx_free": (ci, [vp]),
            "r2_mf_rx": (ci, [vp, vp, ci]), "r2_mf_rx_get": (ci, [vp]), "r2_mf_rx_free": (ci, [vp]),
            "r2_mf_tx_init": (vp, [vp, ci]), "r2_mf_tx_put": (ci, [vp, C.c_char]),
            "r2_mf_tx": (ci, [vp, vp, ci]), "r2_mf_tx_free": (ci, [vp]),
            "super_tone_rx_make_descriptor": (vp, [vp]), "super_tone_rx_free_descriptor": (ci, [vp]),
            "super_tone_rx_add_tone": (ci, [vp]), "super_tone_rx_add_element": (ci, [vp, ci, ci, ci, ci, ci]),
            "super_tone_rx": (ci, [vp, vp, ci]), "super_tone_rx_free": (ci, [vp]),
            "goertzel_update": (ci, [vp, vp, ci]), "goertzel_result": (cf, [vp]), "goertzel_free": (ci, [vp]),
            "tone_gen_descriptor_init": (vp, [vp, ci, ci, ci, ci, ci, ci, ci, ci, ci]),
            "tone_gen_descriptor_free": (None, [vp]),
            "tone_gen_init": (vp, [vp, vp]), "tone_gen": (ci, [vp, vp, ci]), "tone_gen_free": (ci, [vp]),
            "awgn_init_dbm0": (vp, [vp, ci, cf]), "awgn": (C.c_int16, [vp]), "awgn_free": (ci, [vp]),
            "glue_awgn_run": (ci, [ci, cf, vp, ci, vp]),
            "glue_g168_model": (ci, [ci, vp, vp]), "glue_g168_line": (ci, [ci, cf, vp, vp, vp, ci]), "glue_g168_gain": (cf, [ci, cf]),
            "echo_can_init": (vp, [ci, ci]), "echo_can_free": (ci, [vp]), "echo_can_flush": (None, [vp]),
            "echo_can_adaption_mode": (None, [vp, ci]),
            "echo_can_update": (C.c_int16, [vp, C.c_int16, C.c_int16]),
            "echo_can_hpf_tx": (C.c_int16, [vp, C.c_int16]),
            "v29_rx": (ci, [vp, vp, ci]), "v29_rx_free": (ci, [vp]), "v29_rx_restart": (ci, [vp, ci, ci]),
            "v29_rx_set_signal_cutoff": (None, [vp, cf]), "v27ter_rx_set_signal_cutoff": (None, [vp, cf]), "v17_rx_set_signal_cutoff": (None, [vp, cf]),
            "v29_rx_carrier_frequency": (cf, [vp]), "v29_rx_symbol_timing_correction": (cf, [vp]),
            "v29_tx": (ci, [vp, vp, ci]), "v29_tx_free": (ci, [vp]), "v29_tx_power": (None, [vp, cf]),
            "v27ter_rx": (ci, [vp, vp, ci]), "v27ter_rx_free": (ci, [vp]),
            "v27ter_tx": (ci, [vp, vp, ci]), "v27ter_tx_free": (ci, [vp]), "v27ter_tx_power": (None, [vp, cf]),
            "v17_rx": (ci, [vp, vp, ci]), "v17_rx_free": (ci, [vp]),
            "v17_tx": (ci, [vp, vp, ci]), "v17_tx_free": (ci, [vp]), "v17_tx_power": (None, [vp, cf]),
            "glue_fn_prbs_get_bit": (vp, []), "glue_fn_put_bit": (vp, []), "glue_fn_tone_report": (vp, []),
            "modem_connect_tones_rx_init": (vp, [vp, ci, vp, vp]), "modem_connect_tones_rx": (ci, [vp, vp, ci]),
            "modem_connect_tones_rx_get": (ci, [vp]), "modem_connect_tones_rx_free": (ci, [vp]),
            "modem_connect_tones_tx_init": (vp, [vp, ci]), "modem_connect_tones_tx": (ci, [vp, vp, ci]),
            "modem_connect_tones_tx_free": (ci, [vp]), "glue_mct_rx_snapshot": (ci, [vp, vp]),
            "glue_v27ter_tx_tables": (None, [vp, vp]), "glue_v27ter_tx_snapshot": (ci, [vp, vp]), "v27ter_tx_restart": (ci, [vp, ci, ci]),
            "glue_v17_tx_table": (None, [vp]), "glue_v17_tx_snapshot": (ci, [vp, vp]),
            "glue_v29_tx_table": (None, [vp]), "glue_v29_tx_snapshot": (ci, [vp, vp]), "v29_tx_restart": (ci, [vp, ci, ci]),
            "glue_mct_rx_batch_frames": (None, [vp, vp, ci, C.c_longlong, C.c_longlong, ci, ci, ci]),
            "sig_tone_rx_init": (vp, [vp, ci, vp, vp]), "sig_tone_rx": (ci, [vp, vp, ci]),
            "sig_tone_rx_set_mode": (None, [vp, ci, ci]), "sig_tone_rx_free": (ci, [vp]),
            "glue_sigtone_rx_snapshot": (ci, [vp, vp]), "glue_sigtone_rx_thresholds": (None, [vp, vp]),
            "glue_sigtone_rx_new_quiet": (vp, [ci, ci, vp]),
            "glue_sigtone_rx_batch_frames": (None, [vp, vp, ci, C.c_longlong, C.c_longlong, ci, ci, ci]),
            "glue_sigtone_rx_scripted_new": (vp, [ci, ci, vp, ci]), "glue_sigtone_rx_scripted_free": (None, [vp]),
            "glue_sigtone_rx_scripted": (ci, [vp, vp, ci]), "glue_sigtone_rx_scripted_reports": (ci, [vp, C.POINTER(vp)]),
            "glue_sigtone_rx_scripted_state": (vp, [vp]),
            "glue_sigtone_tx_new": (vp, [ci, vp, ci]), "glue_sigtone_tx_free": (None, [vp]),
            "glue_sigtone_tx_set_mode": (None, [vp, ci, ci]), "glue_sigtone_tx": (ci, [vp, vp, ci]),
            "glue_sigtone_tx_requests": (ci, [vp]), "glue_sigtone_tx_snapshot": (ci, [vp, vp]),
            "glue_fsk_preset": (ci, [ci, vp]), "glue_fsk_rx_new": (vp, [ci, ci, vp, vp]),
            "glue_fsk_rx_restart": (ci, [vp, ci, ci]), "glue_fsk_tx_new": (vp, [ci, vp, vp]),
            "glue_fsk_rx_snapshot": (ci, [vp, vp]),
            "glue_fsk_rx_new_quiet": (vp, [ci, ci, vp]), "glue_fsk_rx_batch": (None, [vp, vp, ci, C.c_longlong, ci]),
            "glue_fsk_rx_batch_frames": (None, [vp, vp, ci, C.c_longlong, C.c_longlong, ci, ci, ci]),
            "fsk_rx": (ci, [vp, vp, ci]), "fsk_rx_free": (ci, [vp]), "fsk_rx_fillin": (ci, [vp, ci]),
            "fsk_rx_set_signal_cutoff": (None, [vp, cf]), "fsk_rx_set_frame_parameters": (None, [vp, ci, ci, ci]),
            "fsk_tx": (ci, [vp, vp, ci]), "fsk_tx_free": (ci, [vp]), "fsk_tx_power": (None, [vp, cf]),
            "vec_dot_prodf": (cf, [vp, vp, ci]), "vec_circular_dot_prodf": (cf, [vp, vp, ci, ci]),
            "vec_lmsf": (None, [vp, vp, ci, cf]), "vec_circular_lmsf": (None, [vp, vp, ci, ci, cf]),
            "cvec_circular_dot_prodf": (_Cf, [vp, vp, ci, ci]), "cvec_circular_lmsf": (None, [vp, vp, ci, ci, vp]),
            "power_meter_update": (C.c_int32, [vp, C.c_int16]),
        }
        for name, (res, args) in sigs.items():
            if not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _libs[_flavour] = L
    return _libs[_flavour]


@contextlib.contextmanager
def quiet_stdout():
    """echo.c of this snapshot printf()s on every sample; silence fd 1 around calls."""
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    null = os.open(os.devnull, os.O_WRONLY)
    try:
        os.dup2(null, 1)
        yield
    finally:
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(null)
        os.close(saved)


def _i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


class Sink:
    def __init__(self):
        self.p = lib().glue_sink_new()

    def __del__(self):
        try:
            lib().glue_sink_free(self.p)
        except Exception:
            pass

    def clear(self):
        lib().glue_sink_clear(self.p)

    def events(self):
        n = lib().glue_sink_count(self.p)
        if n == 0:
            return np.zeros(0, EVENT_DTYPE)
        addr = lib().glue_sink_events(self.p)
        buf = (C.c_char*(n*EVENT_DTYPE.itemsize)).from_address(addr)
        return np.frombuffer(buf, dtype=EVENT_DTYPE).copy()

    def text(self):
        return lib().glue_sink_digits(self.p)[:lib().glue_sink_ndigits(self.p)].decode("latin1")


class DtmfRx:
    """mode 0 = buffered digits, 1 = digits callback, 2 = realtime callback (same numbering as the oracle)."""

    def __init__(self, mode=0):
        self.sink = Sink()
        self.p = lib().glue_dtmf_rx_new(self.sink.p, int(mode == 1), int(mode == 2))

    def __del__(self):
        try:
            lib().dtmf_rx_free(self.p)
        except Exception:
            pass

    def parms(self, filter_dialtone=-1, twist=-1.0, reverse_twist=-1.0, threshold=-99.0):
        lib().dtmf_rx_parms(self.p, filter_dialtone, twist, reverse_twist, threshold)

    def rx(self, amp):
        amp = _i16(amp)
        return lib().dtmf_rx(self.p, amp.ctypes.data, len(amp))

    def get(self, maxlen=128):
        out = C.create_string_buffer(maxlen + 1)
        lib().dtmf_rx_get(self.p, out, maxlen)
        return out.value.decode("latin1")

    def status(self):
        return lib().dtmf_rx_status(self.p)

    def fillin(self, samples):
        lib().dtmf_rx_fillin(self.p, samples)

    def snapshot(self):
        f = np.zeros(17, np.float32)
        i = np.zeros(6, np.int32)
        lib().glue_dtmf_rx_snapshot(self.p, f.ctypes.data, i.ctypes.data)
        return {"v2": f[0:8].copy(), "v3": f[8:16].copy(), "energy": float(f[16]),
                "current_sample": int(i[0]), "duration": int(i[1]), "last_hit": int(i[2]),
                "in_digit": int(i[3]), "lost_digits": int(i[4]), "current_digits": int(i[5])}

    def consts(self):
        f = np.zeros(11, np.float32)
        lib().glue_dtmf_rx_consts(self.p, f.ctypes.data)
        return {"fac": f[0:8].copy(), "threshold": float(f[8]), "normal_twist": float(f[9]), "reverse_twist": float(f[10])}


class BellMfRx:
    def __init__(self, mode=0):
        self.sink = Sink()
        self.p = lib().glue_bell_mf_rx_new(self.sink.p, int(mode == 1))

    def __del__(self):
        try:
            lib().bell_mf_rx_free(self.p)
        except Exception:
            pass

    def rx(self, amp):
        amp = _i16(amp)
        return lib().bell_mf_rx(self.p, amp.ctypes.data, len(amp))

    def get(self, maxlen=128):
        out = C.create_string_buffer(maxlen + 1)
        lib().bell_mf_rx_get(self.p, out, maxlen)
        return out.value.decode("latin1")

    def snapshot(self):
        f = np.zeros(18, np.float32)
        i = np.zeros(8, np.int32)
        lib().glue_bell_mf_rx_snapshot(self.p, f.ctypes.data, i.ctypes.data)
        return {"v2": f[0:6].copy(), "v3": f[6:12].copy(), "fac": f[12:18].copy(),
                "current_sample": int(i[0]), "hits": i[1:6].copy(), "lost_digits": int(i[6]),
                "current_digits": int(i[7])}


class R2MfRx:
    def __init__(self, fwd=True, use_callback=True):
        self.sink = Sink()
        self.p = lib().glue_r2_mf_rx_new(self.sink.p, int(fwd), int(use_callback))

    def __del__(self):
        try:
            lib().r2_mf_rx_free(self.p)
        except Exception:
            pass

    def rx(self, amp):
        amp = _i16(amp)
        return lib().r2_mf_rx(self.p, amp.ctypes.data, len(amp))

    def get(self):
        return lib().r2_mf_rx_get(self.p)

    def snapshot(self):
        f = np.zeros(18, np.float32)
        i = np.zeros(2, np.int32)
        lib().glue_r2_mf_rx_snapshot(self.p, f.ctypes.data, i.ctypes.data)
        return {"v2": f[0:6].copy(), "v3": f[6:12].copy(), "fac": f[12:18].copy(),
                "current_sample": int(i[0]), "current_digit": int(i[1])}


class SuperToneDesc:
    def __init__(self):
        self.p = lib().super_tone_rx_make_descriptor(None)

    def add_tone(self):
        return lib().super_tone_rx_add_tone(self.p)

    def add_element(self, tone, f1, f2, min_ms, max_ms):
        return lib().super_tone_rx_add_element(self.p, tone, f1, f2, min_ms, max_ms)

    @property
    def fac(self):
        f = np.zeros(64, np.float32)
        n = lib().glue_super_tone_desc_bins(self.p, f.ctypes.data)
        return f[:n].copy()


class SuperToneRx:
    def __init__(self, desc, use_segment_cb=False):
        self.desc = desc
        self.sink = Sink()
        self.p = lib().glue_super_tone_rx_new(desc.p, self.sink.p, int(use_segment_cb))

    def rx(self, amp):
        amp = _i16(amp)
        return lib().super_tone_rx(self.p, amp.ctypes.data, len(amp))

    def snapshot(self):
        m = len(self.desc.fac)
        f = np.zeros(1 + 2*m, np.float32)
        i = np.zeros(36, np.int32)
        lib().glue_super_tone_rx_snapshot(self.p, f.ctypes.data, i.ctypes.data)
        return {"energy": float(f[0]), "v2": f[1:1 + m].copy(), "v3": f[1 + m:1 + 2*m].copy(),
                "detected_tone": int(i[0]), "rotation": int(i[1]), "current_sample": int(i[2]),
                "segments": i[3:36].reshape(11, 3).copy()}


class Goertzel:
    def __init__(self, freq, samples):
        self.p = lib().glue_goertzel_new(freq, samples)

    def update(self, amp):
        amp = _i16(amp)
        return lib().goertzel_update(self.p, amp.ctypes.data, len(amp))

    def result(self):
        return lib().goertzel_result(self.p)


# ---- signal sources (the reference's own transmitters / noise) ----------------
def dtmf_tx(digits, level=None, twist=None, on_ms=None, off_ms=None, max_samples=1 << 22):
    L = lib()
    tx = L.dtmf_tx_init(None, None, None)
    if level is not None:
        L.dtmf_tx_set_level(tx, level, twist or 0)
    if on_ms is not None:
        L.dtmf_tx_set_timing(tx, on_ms, off_ms)
    L.dtmf_tx_put(tx, digits.encode(), -1)
    out = []
    buf = np.zeros(4096, np.int16)
    total = 0
    while total < max_samples:
        n = L.dtmf_tx(tx, buf.ctypes.data, len(buf))
        if n <= 0:
            break
        out.append(buf[:n].copy())
        total += n
    L.dtmf_tx_free(tx)
    return np.concatenate(out) if out else np.zeros(0, np.int16)


def bell_mf_tx(digits, max_samples=1 << 22):
    L = lib()
    tx = L.bell_mf_tx_init(None)
    L.bell_mf_tx_put(tx, digits.encode(), -1)
    out = []
    buf = np.zeros(4096, np.int16)
    total = 0
    while total < max_samples:
        n = L.bell_mf_tx(tx, buf.ctypes.data, len(buf))
        if n <= 0:
            break
        out.append(buf[:n].copy())
        total += n
    L.bell_mf_tx_free(tx)
    return np.concatenate(out) if out else np.zeros(0, np.int16)


def r2_mf_tx(digits, fwd=True, on_samples=800, off_samples=400):
    """R2 tones are continuous; key each digit for on_samples then silence."""
    L = lib()
    tx = L.r2_mf_tx_init(None, int(fwd))
    out = []
    for d in digits:
        L.r2_mf_tx_put(tx, d.encode())
        buf = np.zeros(on_samples, np.int16)
        L.r2_mf_tx(tx, buf.ctypes.data, on_samples)
        out.append(buf)
        L.r2_mf_tx_put(tx, b"\0")
        buf = np.zeros(off_samples, np.int16)
        L.r2_mf_tx(tx, buf.ctypes.data, off_samples)
        out.append(buf)
    L.r2_mf_tx_free(tx)
    return np.concatenate(out)


def tone_pair(f1, l1, f2, l2, samples, d1=None):
    """tone_gen with one cadence section held on (tone_generate.c:67-229)."""
    L = lib()
    d = L.tone_gen_descriptor_init(None, f1, l1, f2, l2, d1 if d1 is not None else 1, 0, 0, 0, 1)
    g = L.tone_gen_init(None, d)
    buf = np.zeros(samples, np.int16)
    L.tone_gen(g, buf.ctypes.data, samples)
    L.tone_gen_free(g)
    L.tone_gen_descriptor_free(d)
    return buf



class _RefSender:
    """A live transmitter of the real reference: put() digits, tx(n) samples."""
    _free = None

    def __del__(self):
        try:
            getattr(lib(), self._free)(self.p)
        except Exception:
            pass

    def tx(self, n):
        buf = np.zeros(max(n, 1), np.int16)
        got = getattr(lib(), self._tx)(self.p, buf.ctypes.data, n)
        return buf[:got].copy()


class DtmfTx(_RefSender):
    _free, _tx = "dtmf_tx_free", "dtmf_tx"

    def __init__(self):
        self.p = lib().dtmf_tx_init(None, None, None)

    def set_level(self, level, twist):
        lib().dtmf_tx_set_level(self.p, level, twist)

    def set_timing(self, on_ms, off_ms):
        lib().dtmf_tx_set_timing(self.p, on_ms, off_ms)

    def put(self, digits):
        b = digits if isinstance(digits, bytes) else digits.encode()
        return lib().dtmf_tx_put(self.p, b, len(b))


class BellMfTx(_RefSender):
    _free, _tx = "bell_mf_tx_free", "bell_mf_tx"

    def __init__(self):
        self.p = lib().bell_mf_tx_init(None)

    def put(self, digits):
        b = digits if isinstance(digits, bytes) else digits.encode()
        return lib().bell_mf_tx_put(self.p, b, len(b))


class R2MfTx(_RefSender):
    _free, _tx = "r2_mf_tx_free", "r2_mf_tx"

    def __init__(self, fwd=True):
        self.p = lib().r2_mf_tx_init(None, int(fwd))

    def put(self, digit):
        b = digit if isinstance(digit, bytes) else digit.encode()
        return lib().r2_mf_tx_put(self.p, b[:1] if b else b"\0")


class ToneGen(_RefSender):
    _free, _tx = "tone_gen_free", "tone_gen"

    def __init__(self, f1, l1, f2, l2, d1, d2=0, d3=0, d4=0, repeat=False):
        d = lib().tone_gen_descriptor_init(None, f1, l1, f2, l2, d1, d2, d3, d4, int(repeat))
        self.p = lib().tone_gen_init(None, d)
        lib().tone_gen_descriptor_free(d)


def _awgn_run(seed, level_dbm0, samples):
    out = np.zeros(max(samples, 1), np.int16)
    words = np.zeros(202, np.uint32)
    assert lib().glue_awgn_run(seed, level_dbm0, out.ctypes.data, samples, words.ctypes.data) == 202
    return out[:samples], words


def awgn(seed, level_dbm0, samples):
    return _awgn_run(seed, level_dbm0, samples)[0]


def awgn_state_words(seed, level_dbm0, samples):
    """The generator's state after `samples` calls, in the oracle / device word layout."""
    return _awgn_run(seed, level_dbm0, samples)[1]


def g168_model(model):
    """Taps (int32) and gain constant of G.168 line model D`model` (2 .. 9), src/spandsp/g168models.h."""
    out = np.zeros(256, np.int32)
    ki = C.c_float(0.0)
    n = lib().glue_g168_model(model, out.ctypes.data, C.byref(ki))
    assert n > 0, model
    return out[:n].copy(), float(ki.value)


def g168_gain(model, erl_db):
    return np.float32(lib().glue_g168_gain(model, erl_db))


def g168_line(model, erl_db, rout, sgen):
    """What comes back from the line: the echo of `rout` through model D`model` at `erl_db` (negative) plus the near end
    signal `sgen`, by the reference's own fir32() as tests/echo_tests.c's channel_model() uses it (no codec)."""
    rout = _i16(rout)
    sgen = _i16(sgen)
    out = np.zeros(len(rout), np.int16)
    assert lib().glue_g168_line(model, erl_db, rout.ctypes.data, sgen.ctypes.data, out.ctypes.data, len(rout)) > 0
    return out


def saturated_add(a, b):
    return np.clip(a.astype(np.int32) + b.astype(np.int32), -32768, 32767).astype(np.int16)


ECHO_FIELDS = ["tx_power0", "tx_power1", "tx_power2", "tx_power3", "rx_power0", "rx_power1", "rx_power2",
               "clean_rx_power", "rx_power_threshold", "nonupdate_dwell", "curr_pos", "taps", "tap_mask",
               "adaption_mode", "supp_test1", "supp_test2", "supp1", "supp2", "vad", "cng", "geigel_max",
               "geigel_lag", "dtd_onset", "tap_set", "tap_rotate_counter", "latest_correction",
               "narrowband_count", "narrowband_score", "fir_curr_pos", "tx_hpf0", "tx_hpf1", "rx_hpf0",
               "rx_hpf1", "cng_level", "cng_rndnum", "cng_filter"]


class EchoCan:
    """echo_can_init()/echo_can_update() of the real reference, on a zero-padded heap and
    with its per-sample debug printf()s sent to /dev/null."""

    def __init__(self, taps, mode):
        lib().glue_install_padded_allocator(1)
        self.p = lib().echo_can_init(taps, mode)
        lib().glue_install_padded_allocator(0)
        self.taps = taps

    def __del__(self):
        try:
            lib().echo_can_free(self.p)
        except Exception:
            pass

    def flush(self):
        lib().echo_can_flush(self.p)

    def adaption_mode(self, mode):
        lib().echo_can_adaption_mode(self.p, mode)

    def run(self, tx, rx, use_hpf_tx=False):
        tx = _i16(tx)
        rx = _i16(rx)
        out = np.zeros(len(tx), np.int16)
        with quiet_stdout():
            lib().glue_echo_run(self.p, tx.ctypes.data, rx.ctypes.data, out.ctypes.data, len(tx), int(use_hpf_tx))
        return out

    def snapshot(self):
        i = np.zeros(64, np.int32)
        t32 = np.zeros(self.taps, np.int32)
        t16 = np.zeros(4*self.taps, np.int16)
        h = np.zeros(self.taps, np.int16)
        lib().glue_echo_snapshot(self.p, i.ctypes.data, t32.ctypes.data, t16.ctypes.data, h.ctypes.data)
        d = {k: int(v) for k, v in zip(ECHO_FIELDS, i)}
        d["taps32"] = t32
        d["taps16"] = t16.reshape(4, self.taps)
        d["history"] = h
        return d


def modem_tables():
    """The constant tables of the reference build's V.29 receiver, as numpy arrays."""
    t = {"rrc_re": np.zeros(48*27, np.float32), "rrc_im": np.zeros(48*27, np.float32),
         "sine": np.zeros(2048, np.float32), "sqrt_tab": np.zeros(193, np.uint16),
         "godard": np.zeros(9, np.float32), "steps": np.zeros(2, np.int32)}
    lib().glue_modem_tables(t["rrc_re"].ctypes.data, t["rrc_im"].ctypes.data, t["sine"].ctypes.data,
                            t["sqrt_tab"].ctypes.data, t["godard"].ctypes.data, t["steps"].ctypes.data)
    for k, n in (("v27_4800_re", 8), ("v27_4800_im", 8), ("v27_2400_re", 12), ("v27_2400_im", 12)):
        t[k] = np.zeros(n*27, np.float32)
    lib().glue_v27ter_tables(t["v27_4800_re"].ctypes.data, t["v27_4800_im"].ctypes.data,
                             t["v27_2400_re"].ctypes.data, t["v27_2400_im"].ctypes.data)
    t["v17_re"] = np.zeros(192*27, np.float32)
    t["v17_im"] = np.zeros(192*27, np.float32)
    t["v17_godard"] = np.zeros(9, np.float32)
    t["v17_steps"] = np.zeros(2, np.int32)
    lib().glue_v17_tables(t["v17_re"].ctypes.data, t["v17_im"].ctypes.data, t["v17_godard"].ctypes.data,
                          t["v17_steps"].ctypes.data)
    return t


def v17_signal_space():
    """The reference's V.17 constellations (244 points, 14400/12000/9600/7200/4800) and receiver soft-decision maps."""
    c = np.zeros(244*2, np.float32)
    maps = np.zeros(4*36*36*8, np.uint8)
    m48 = np.zeros(36*36, np.uint8)
    lib().glue_v17_constellations(c.ctypes.data)
    lib().glue_v17_constel_maps(maps.ctypes.data, m48.ctypes.data)
    return {"v17_constellation": c, "v17_maps": maps, "v17_map_4800": m48}


class V29Rx:
    def __init__(self, bit_rate=9600):
        self.sink = Sink()
        self.p = lib().glue_v29_rx_new(bit_rate, self.sink.p)

    def __del__(self):
        try:
            lib().v29_rx_free(self.p)
        except Exception:
            pass

    def tap_qam(self):
        lib().glue_v29_rx_tap_qam(self.p, self.sink.p)

    def rx(self, amp):
        amp = _i16(amp)
        return lib().v29_rx(self.p, amp.ctypes.data, len(amp))

    def snapshot(self):
        f = np.zeros(238, np.float32)
        w = np.zeros(43, np.int32)
        lib().glue_v29_rx_snapshot(self.p, f.ctypes.data, w.ctypes.data)
        return f, w

    def set_signal_cutoff(self, cutoff_dbm0):
        lib().v29_rx_set_signal_cutoff(self.p, cutoff_dbm0)

    def carrier_frequency(self):
        return float(lib().v29_rx_carrier_frequency(self.p))

    def symbol_timing_correction(self):
        """in steps of the pulse shaper's coefficient sets (48 to a sample), v29rx.c:180-183"""
        return float(lib().v29_rx_symbol_timing_correction(self.p))


class V27terRx:
    def __init__(self, bit_rate=4800):
        self.sink = Sink()
        self.p = lib().glue_v27ter_rx_new(bit_rate, self.sink.p)

    def __del__(self):
        try:
            lib().v27ter_rx_free(self.p)
        except Exception:
            pass

    def tap_qam(self):
        lib().glue_v27ter_rx_tap_qam(self.p, self.sink.p)

    def rx(self, amp):
        amp = _i16(amp)
        return lib().v27ter_rx(self.p, amp.ctypes.data, len(amp))

    def snapshot(self):
        f = np.zeros(225, np.float32)
        w = np.zeros(45, np.int32)
        lib().glue_v27ter_rx_snapshot(self.p, f.ctypes.data, w.ctypes.data)
        return f, w


class V17Rx:
    def __init__(self, bit_rate=14400):
        self.sink = Sink()
        self.p = lib().glue_v17_rx_new(bit_rate, self.sink.p)

    def __del__(self):
        try:
            lib().v17_rx_free(self.p)
        except Exception:
            pass

    def restart(self, bit_rate, short_train):
        return lib().v17_rx_restart(self.p, bit_rate, int(short_train))

    def tap_qam(self):
        lib().glue_v17_rx_tap_qam(self.p, self.sink.p)

    def rx(self, amp):
        amp = _i16(amp)
        return lib().v17_rx(self.p, amp.ctypes.data, len(amp))

    def snapshot(self):
        f = np.zeros(246, np.float32)
        w = np.zeros(301, np.int32)
        lib().glue_v17_rx_snapshot(self.p, f.ctypes.data, w.ctypes.data)
        return f, w


def v17_tx(bit_rate, n_samples, seed=1, tep=False, short_train=False, level_dbm0=None):
    """v17_tx() of the reference carrying a PRBS; returns int16 samples."""
    L = lib()
    st = C.c_uint32(seed & 0x7FFF or 1)
    tx = L.glue_v17_tx_new(bit_rate, int(tep), C.addressof(st))
    if short_train:
        L.v17_tx_restart(tx, bit_rate, int(tep), 1)
    if level_dbm0 is not None:
        L.v17_tx_power(tx, level_dbm0)
    buf = np.zeros(n_samples, np.int16)
    n = L.v17_tx(tx, buf.ctypes.data, n_samples)
    L.v17_tx_free(tx)
    return buf[:n]


def v27ter_tx(bit_rate, n_samples, seed=1, tep=False, level_dbm0=None):
    """v27ter_tx() of the reference carrying a PRBS; returns int16 samples."""
    L = lib()
    st = C.c_uint32(seed & 0x7FFF or 1)
    tx = L.glue_v27ter_tx_new(bit_rate, int(tep), C.addressof(st))
    if level_dbm0 is not None:
        L.v27ter_tx_power(tx, level_dbm0)
    buf = np.zeros(n_samples, np.int16)
    n = L.v27ter_tx(tx, buf.ctypes.data, n_samples)
    L.v27ter_tx_free(tx)
    return buf[:n]


def v29_tx(bit_rate, n_samples, seed=1, tep=False, level_dbm0=None):
    """v29_tx() of the reference carrying a PRBS; returns int16 samples."""
    L = lib()
    st = C.c_uint32(seed & 0x7FFF or 1)
    tx = L.glue_v29_tx_new(bit_rate, int(tep), C.addressof(st))
    if level_dbm0 is not None:
        L.v29_tx_power(tx, level_dbm0)
    buf = np.zeros(n_samples, np.int16)
    n = L.v29_tx(tx, buf.ctypes.data, n_samples)
    L.v29_tx_free(tx)
    return buf[:n]


# ---- FSK (src/fsk.c) ---------------------------------------------------------------------
FSK_PRESETS = ["V21CH1", "V21CH2", "V23CH1", "V23CH2", "BELL103CH1", "BELL103CH2", "BELL202", "WEITBRECHT_4545",
               "WEITBRECHT_50", "WEITBRECHT_476", "V21CH1_110"]


def v18_tone_blocks(amp):
    """in_tone after every 102-sample block of amp through v18_rx() (caller_tone_scan), and the object's threshold."""
    amp = _i16(amp)
    nb = len(amp)//102
    out = np.zeros(nb, np.int32)
    lib().glue_v18_tone_blocks.restype = C.c_float
    lib().glue_v18_tone_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    with quiet_stdout():
        thr = lib().glue_v18_tone_blocks(amp.ctypes.data, nb, out.ctypes.data)
    return out, float(thr)


def ademco_tone_blocks(amp):
    """last_hit after every 55-sample block of amp through ademco_contactid_sender_rx()."""
    amp = _i16(amp)
    nb = len(amp)//55
    out = np.zeros(nb, np.int32)
    lib().glue_ademco_tone_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    with quiet_stdout():
        assert lib().glue_ademco_tone_blocks(amp.ctypes.data, nb, out.ctypes.data) == 0
    return out


def fsk_preset(which):
    out = np.zeros(5, np.int32)
    assert lib().glue_fsk_preset(which, out.ctypes.data) == 0
    return out


class FskRx:
    """fsk_rx_init(NULL, &preset_fsk_specs[which], framing_mode, put_bit, sink) of the real reference."""

    def __init__(self, which, framing_mode):
        self.sink = Sink()
        self.p = lib().glue_fsk_rx_new(which, framing_mode, lib().glue_fn_put_bit(), self.sink.p)

    def __del__(self):
        try:
            lib().fsk_rx_free(self.p)
        except Exception:
            pass

    def restart(self, which, framing_mode):
        return lib().glue_fsk_rx_restart(self.p, which, framing_mode)

    def set_signal_cutoff(self, cutoff):
        lib().fsk_rx_set_signal_cutoff(self.p, cutoff)

    def set_frame_parameters(self, data_bits, parity, stop_bits):
        lib().fsk_rx_set_frame_parameters(self.p, data_bits, parity, stop_bits)

    def rx(self, amp):
        amp = _i16(amp)
        return lib().fsk_rx(self.p, amp.ctypes.data, len(amp))

    def fillin(self, n):
        return lib().fsk_rx_fillin(self.p, n)

    def snapshot(self):
        out = np.zeros(28 + 4*128, np.int32)
        n = lib().glue_fsk_rx_snapshot(self.p, out.ctypes.data)
        return out[:n].copy()


def fsk_tx(which, n_samples, seed=1, level_dbm0=None, bits=None):
    """fsk_tx() fed by the glue's PRBS (or by the 0/1 array `bits`, then idle marks)."""
    L = lib()
    if bits is None:
        st = (C.c_uint32*1)(seed & 0x7FFF or 1)
        tx = L.glue_fsk_tx_new(which, L.glue_fn_prbs_get_bit(), C.cast(st, C.c_void_p))
        keep = st
    else:
        seq = [int(b) for b in bits]
        pos = [0]
        GET = C.CFUNCTYPE(C.c_int, C.c_void_p)

        def get_bit(_):
            pos[0] += 1
            return seq[pos[0] - 1] if pos[0] <= len(seq) else 1
        keep = GET(get_bit)
        tx = L.glue_fsk_tx_new(which, C.cast(keep, C.c_void_p), None)
    if level_dbm0 is not None:
        L.fsk_tx_power(tx, level_dbm0)
    out = np.zeros(n_samples, np.int16)
    got = L.fsk_tx(tx, out.ctypes.data, n_samples)
    L.fsk_tx_free(tx)
    del keep
    return out[:got]


# ---- modem connect tones (src/modem_connect_tones.c) --------------------------------------
class MctRx:
    """modem_connect_tones_rx_init(NULL, tone_type, callback, sink); use_callback=False leaves the hit latch."""

    def __init__(self, tone_type, use_callback=True):
        self.sink = Sink()
        self.p = lib().modem_connect_tones_rx_init(None, tone_type, lib().glue_fn_tone_report() if use_callback else None,
                                                   self.sink.p if use_callback else None)

    def __del__(self):
        try:
            lib().modem_connect_tones_rx_free(self.p)
        except Exception:
            pass

    def rx(self, amp):
        amp = _i16(amp)
        return lib().modem_connect_tones_rx(self.p, amp.ctypes.data, len(amp))

    def get(self):
        return lib().modem_connect_tones_rx_get(self.p)

    def snapshot(self):
        out = np.zeros(18 + 28 + 4*128, np.int32)
        n = lib().glue_mct_rx_snapshot(self.p, out.ctypes.data)
        return out[:n].copy()


# ---- in-band signalling tones (src/sig_tone.c) ---------------------------------------------
class SigToneRx:
    """sig_tone_rx_init(NULL, tone_type, callback, sink); rx() returns the frame as the receiver left it."""

    def __init__(self, tone_type, mode=0):
        self.sink = Sink()
        self.p = lib().sig_tone_rx_init(None, tone_type, lib().glue_fn_tone_report(), self.sink.p)
        if not self.p:
            raise ValueError("sig_tone_rx_init refused tone type %d" % tone_type)
        self.set_mode(mode)

    def __del__(self):
        try:
            lib().sig_tone_rx_free(self.p)
        except Exception:
            pass

    def set_mode(self, mode):
        lib().sig_tone_rx_set_mode(self.p, mode, 0)

    def rx(self, amp):
        buf = _i16(amp).copy()
        lib().sig_tone_rx(self.p, buf.ctypes.data, len(buf))
        return buf

    def snapshot(self):
        out = np.zeros(27, np.int32)
        n = lib().glue_sigtone_rx_snapshot(self.p, out.ctypes.data)
        return out[:n].copy()

    def thresholds(self):
        out = np.zeros(3, np.int32)
        lib().glue_sigtone_rx_thresholds(self.p, out.ctypes.data)
        return out


class SigToneRxScripted:
    """A receiver whose callback sets the next mode of `script` at every report (sig_tone_rx_set_mode from inside it)."""

    def __init__(self, tone_type, mode, script):
        self.script = np.ascontiguousarray(script, np.int32)
        self.p = lib().glue_sigtone_rx_scripted_new(tone_type, mode, self.script.ctypes.data, len(self.script))

    def __del__(self):
        try:
            lib().glue_sigtone_rx_scripted_free(self.p)
        except Exception:
            pass

    def rx(self, amp):
        buf = _i16(amp).copy()
        lib().glue_sigtone_rx_scripted(self.p, buf.ctypes.data, len(buf))
        return buf

    def reports(self):
        ptr = C.c_void_p()
        n = lib().glue_sigtone_rx_scripted_reports(self.p, C.byref(ptr))
        if n == 0:
            return np.zeros((0, 3), np.int32)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), (n*3,)).reshape(n, 3).copy()

    def snapshot(self):
        out = np.zeros(27, np.int32)
        lib().glue_sigtone_rx_snapshot(lib().glue_sigtone_rx_scripted_state(self.p), out.ctypes.data)
        return out


class SigToneTx:
    """sig_tone_tx_init(NULL, tone_type, callback, ...) with a callback that sets the next (mode, duration) of `script`."""

    def __init__(self, tone_type, script=()):
        self.script = np.ascontiguousarray(np.asarray(script, np.int32).reshape(-1, 2))
        self.p = lib().glue_sigtone_tx_new(tone_type, self.script.ctypes.data, len(self.script))
        if not self.p:
            raise ValueError("sig_tone_tx_init refused tone type %d" % tone_type)

    def __del__(self):
        try:
            lib().glue_sigtone_tx_free(self.p)
        except Exception:
            pass

    def set_mode(self, mode, duration):
        lib().glue_sigtone_tx_set_mode(self.p, mode, duration)

    def tx(self, amp):
        buf = _i16(amp).copy()
        lib().glue_sigtone_tx(self.p, buf.ctypes.data, len(buf))
        return buf

    def requests(self):
        return lib().glue_sigtone_tx_requests(self.p)

    def snapshot(self):
        out = np.zeros(11, np.int32)
        n = lib().glue_sigtone_tx_snapshot(self.p, out.ctypes.data)
        return out[:n].copy()


def modem_connect_tones_tx(tone_type, n_samples):
    L = lib()
    tx = L.modem_connect_tones_tx_init(None, tone_type)
    out = np.zeros(n_samples, np.int16)
    got = L.modem_connect_tones_tx(tx, out.ctypes.data, n_samples)
    L.modem_connect_tones_tx_free(tx)
    return out[:got]


# ---- V.29 transmitter as a live object (src/v29tx.c) ------------------------------------------
def v29_tx_table():
    out = np.zeros(90, np.float32)
    lib().glue_v29_tx_table(out.ctypes.data)
    return out


class V29Tx:
    """v29_tx_init(NULL, bit_rate, tep, prbs_get_bit, &state) of the real reference."""

    def __init__(self, bit_rate, tep=False, seed=1):
        self.st = (C.c_uint32*1)(seed & 0x7FFF)
        self.p = lib().glue_v29_tx_new(bit_rate, int(tep), C.cast(self.st, C.c_void_p))

    def __del__(self):
        try:
            lib().v29_tx_free(self.p)
        except Exception:
            pass

    def power(self, level_dbm0):
        lib().v29_tx_power(self.p, level_dbm0)

    def restart(self, bit_rate, tep):
        return lib().v29_tx_restart(self.p, bit_rate, int(tep))

    def tx(self, n):
        out = np.zeros(max(n, 1), np.int16)
        got = lib().v29_tx(self.p, out.ctypes.data, n)
        return out[:got].copy()

    def snapshot(self):
        out = np.zeros(32, np.uint32)
        n = lib().glue_v29_tx_snapshot(self.p, out.ctypes.data)
        out[n] = self.st[0]
        return out[:n + 1].copy()


def v27ter_tx_tables():
    a = np.zeros(45, np.float32)
    b = np.zeros(180, np.float32)
    lib().glue_v27ter_tx_tables(a.ctypes.data, b.ctypes.data)
    return a, b


class V27terTx(V29Tx):
    """v27ter_tx_init(NULL, bit_rate, tep, prbs_get_bit, &state) of the real reference."""

    def __init__(self, bit_rate, tep=False, seed=1):
        self.st = (C.c_uint32*1)(seed & 0x7FFF)
        self.p = lib().glue_v27ter_tx_new(bit_rate, int(tep), C.cast(self.st, C.c_void_p))

    def __del__(self):
        try:
            lib().v27ter_tx_free(self.p)
        except Exception:
            pass

    def power(self, level_dbm0):
        lib().v27ter_tx_power(self.p, level_dbm0)

    def restart(self, bit_rate, tep):
        return lib().v27ter_tx_restart(self.p, bit_rate, int(tep))

    def tx(self, n):
        out = np.zeros(max(n, 1), np.int16)
        got = lib().v27ter_tx(self.p, out.ctypes.data, n)
        return out[:got].copy()

    def snapshot(self):
        out = np.zeros(32, np.uint32)
        n = lib().glue_v27ter_tx_snapshot(self.p, out.ctypes.data)
        out[n] = self.st[0]
        return out[:n + 1].copy()


def v17_tx_table():
    out = np.zeros(90, np.float32)
    lib().glue_v17_tx_table(out.ctypes.data)
    return out


class V17Tx(V29Tx):
    """v17_tx_init(NULL, bit_rate, tep, prbs_get_bit, &state) of the real reference."""

    def __init__(self, bit_rate, tep=False, seed=1):
        self.st = (C.c_uint32*1)(seed & 0x7FFF)
        self.p = lib().glue_v17_tx_new(bit_rate, int(tep), C.cast(self.st, C.c_void_p))

    def __del__(self):
        try:
            lib().v17_tx_free(self.p)
        except Exception:
            pass

    def power(self, level_dbm0):
        lib().v17_tx_power(self.p, level_dbm0)

    def restart(self, bit_rate, tep, short_train=False):
        return lib().v17_tx_restart(self.p, bit_rate, int(tep), int(short_train))

    def tx(self, n):
        out = np.zeros(max(n, 1), np.int16)
        got = lib().v17_tx(self.p, out.ctypes.data, n)
        return out[:got].copy()

    def snapshot(self):
        out = np.zeros(32, np.uint32)
        n = lib().glue_v17_tx_snapshot(self.p, out.ctypes.data)
        out[n] = self.st[0]
        return out[:n + 1].copy()


# ---- the pthread driver of the cpu_baseline legs (ref_glue/ref_glue_mt.c) ---------------------------------------
MT_DTMF, MT_BELL_MF, MT_R2_MF, MT_SUPER_TONE, MT_V29, MT_V27TER, MT_V17, MT_FSK, MT_MCT, MT_SIGTONE = range(10)


def mt_rx(kind, states, frames, loops, threads):
    """Runs `loops` passes over frames (int16 [n_frames, n_ch, samples], C order) through the reference receiver
    objects `states` (list of pointers, one per channel) of `kind` on `threads` host threads, all inside C.
    Returns the elapsed seconds."""
    frames = np.ascontiguousarray(frames, np.int16)
    n_frames, n_ch, samples = frames.shape
    assert len(states) == n_ch
    arr = (C.c_void_p*n_ch)(*states)
    return lib().glue_mt_rx(kind, C.addressof(arr), frames.ctypes.data, n_ch, n_frames, samples, samples, n_ch*samples,
                            loops, threads)


def mt_echo(states, tx, rx, loops, threads):
    """The same for echo_can_update(): tx / rx int16 [n_frames, n_ch, samples].  Returns (seconds, clean samples of
    the last frame [n_ch, samples])."""
    tx = np.ascontiguousarray(tx, np.int16)
    rx = np.ascontiguousarray(rx, np.int16)
    n_frames, n_ch, samples = tx.shape
    assert rx.shape == tx.shape and len(states) == n_ch
    arr = (C.c_void_p*n_ch)(*states)
    out = np.zeros((n_ch, samples), np.int16)
    dt = lib().glue_mt_echo(C.addressof(arr), tx.ctypes.data, rx.ctypes.data, out.ctypes.data, n_ch, n_frames, samples,
                            samples, n_ch*samples, loops, threads)
    return dt, out


def timed_baseline(run, samples_per_loop, target_s):
    """run(loops) -> seconds.  Calibrates on a short run, then runs enough loops for about target_s seconds.
    Returns (samples per second, loops, seconds)."""
    loops = 1
    dt = run(loops)
    while dt < 0.05 and loops < (1 << 20):
        loops *= 4
        dt = run(loops)
    want = max(1, int(loops*target_s/max(dt, 1e-9)))
    if want > loops:
        loops = want
        dt = run(loops)
    return samples_per_loop*loops/dt, loops, dt


def usable_cores():
    """Host cores this process can actually run on at once: the smaller of its CPU affinity and the cgroup CPU quota
    (the GPU boxes show 256 CPUs but cap the container at 16 cores' worth of time: cpu.max = "1600000 100000";
    measured in tools/cpu_scaling.py -- the reference scales linearly to 16 threads and not beyond).  Returns
    (cores, description)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    note = "%d CPUs visible" % n
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q)/float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q/per
        except Exception:
            pass
    if quota is not None:
        note += ", cgroup CPU quota %.1f cores" % quota
        n = max(1, min(n, int(quota + 0.5)))
    return n, note
