"""ctypes binding of oracle/liboracle.so (our plain-C restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes as C
import os

import numpy as np

from . import ORACLE_SO, build

EVENT_DTYPE = np.dtype([("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("c", "<i4")])
BLOCK_DTYPE = np.dtype([("hit", "<i4"), ("aux", "<i4"), ("total_energy", "<f4"), ("e", "<f4", (64,))])

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        sigs = {
            "orc_sink_new": (vp, []),
            "orc_sink_free": (None, [vp]),
            "orc_sink_clear": (None, [vp]),
            "orc_sink_want_qam": (None, [vp, ci]),
            "orc_sink_count": (ci, [vp]),
            "orc_sink_events": (vp, [vp]),
            "orc_sink_ntext": (ci, [vp]),
            "orc_sink_text": (C.c_char_p, [vp]),
            "orc_goertzel_fac": (cf, [cf]),
            "orc_goertzel_init": (None, [vp, cf, ci]),
            "orc_goertzel_update": (ci, [vp, vp, ci]),
            "orc_goertzel_result": (cf, [vp]),
            "orc_dtmf_sizeof": (ci, []),
            "orc_dtmf_init": (None, [vp, ci]),
            "orc_dtmf_parms": (None, [vp, ci, cf, cf, cf]),
            "orc_dtmf_rx": (ci, [vp, vp, ci, vp, vp, ci]),
            "orc_dtmf_get": (ci, [vp, C.c_char_p, ci]),
            "orc_dtmf_status": (ci, [vp]),
            "orc_dtmf_fillin": (None, [vp, ci]),
            "orc_bell_mf_sizeof": (ci, []),
            "orc_bell_mf_init": (None, [vp, ci]),
            "orc_bell_mf_rx": (ci, [vp, vp, ci, vp, vp, ci]),
            "orc_bell_mf_get": (ci, [vp, C.c_char_p, ci]),
            "orc_r2_mf_sizeof": (ci, []),
            "orc_r2_mf_init": (None, [vp, ci, ci]),
            "orc_r2_mf_rx": (ci, [vp, vp, ci, vp, vp, ci]),
            "orc_st_desc_sizeof": (ci, []),
            "orc_st_sizeof": (ci, []),
            "orc_st_desc_init": (None, [vp]),
            "orc_st_add_tone": (ci, [vp]),
            "orc_st_add_element": (ci, [vp, ci, ci, ci, ci, ci]),
            "orc_st_init": (None, [vp, vp, ci]),
            "orc_st_rx": (ci, [vp, vp, ci, vp, vp, ci]),
            "orc_modem_set_tables": (None, [vp]),
            "orc_v29_sizeof": (ci, []),
            "orc_v29_init": (ci, [vp, ci]),
            "orc_v29_restart": (ci, [vp, ci, ci]),
            "orc_v29_rx": (ci, [vp, vp, ci, vp]),
            "orc_v17_sizeof": (ci, []),
            "orc_v17_init": (ci, [vp, ci]),
            "orc_v17_restart": (ci, [vp, ci, ci]),
            "orc_v17_rx": (ci, [vp, vp, ci, vp]),
            "orc_v27ter_sizeof": (ci, []),
            "orc_v27ter_init": (ci, [vp, ci]),
            "orc_v27ter_restart": (ci, [vp, ci, ci]),
            "orc_v27ter_rx": (ci, [vp, vp, ci, vp]),
            "orc_fsk_sizeof": (ci, []),
            "orc_fsk_preset": (ci, [ci, vp]),
            "orc_fsk_init": (ci, [vp, vp, ci]),
            "orc_fsk_restart": (ci, [vp, vp, ci]),
            "orc_fsk_set_signal_cutoff": (None, [vp, cf]),
            "orc_fsk_set_frame_parameters": (None, [vp, ci, ci, ci]),
            "orc_fsk_rx": (ci, [vp, vp, ci, vp]),
            "orc_fsk_fillin": (ci, [vp, ci]),
            "orc_v27ter_tx_sizeof": (ci, []),
            "orc_v27ter_tx_set_tables": (None, [vp, vp]),
            "orc_v27ter_tx_init": (ci, [vp, ci, ci, C.c_uint32]),
            "orc_v27ter_tx_restart": (ci, [vp, ci, ci]),
            "orc_v27ter_tx_power": (None, [vp, cf]),
            "orc_v27ter_tx": (ci, [vp, vp, ci]),
            "orc_v17_tx_sizeof": (ci, []),
            "orc_v17_tx_init": (ci, [vp, ci, ci, C.c_uint32]),
            "orc_v17_tx_restart": (ci, [vp, ci, ci, ci]),
            "orc_v17_tx_power": (None, [vp, cf]),
            "orc_v17_tx": (ci, [vp, vp, ci]),
            "orc_awgn_sizeof": (ci, []),
            "orc_awgn_init_dbm0": (None, [vp, ci, cf]),
            "orc_awgn_block": (None, [vp, vp, ci]),
            "orc_v29_tx_sizeof": (ci, []),
            "orc_v29_tx_set_table": (None, [vp]),
            "orc_v29_tx_init": (ci, [vp, ci, ci, C.c_uint32]),
            "orc_v29_tx_restart": (ci, [vp, ci, ci]),
            "orc_v29_tx_power": (None, [vp, cf]),
            "orc_v29_tx": (ci, [vp, vp, ci]),
            "orc_mct_sizeof": (ci, []),
            "orc_mct_init": (None, [vp, ci, vp]),
            "orc_mct_rx": (ci, [vp, vp, ci]),
            "orc_mct_get": (ci, [vp]),
            "orc_sigtone_rx_sizeof": (ci, []), "orc_sigtone_rx_init": (ci, [vp, ci, vp]),
            "orc_sigtone_rx_set_mode": (None, [vp, ci]), "orc_sigtone_rx_script": (None, [vp, vp, ci]), "orc_sigtone_rx_thresholds": (None, [ci, vp]),
            "orc_sigtone_rx": (ci, [vp, vp, ci]),
            "orc_sigtone_tx_sizeof": (ci, []), "orc_sigtone_tx_init": (ci, [vp, ci, vp]),
            "orc_sigtone_tx_set_mode": (None, [vp, ci, ci]), "orc_sigtone_tx_script": (None, [vp, vp, ci]),
            "orc_sigtone_tx": (ci, [vp, vp, ci]),
            "orc_echo_sizeof": (ci, []),
            "orc_echo_init": (ci, [vp, ci, ci]),
            "orc_echo_adaption_mode": (None, [vp, ci]),
            "orc_echo_flush": (None, [vp]),
            "orc_echo_run": (None, [vp, vp, vp, vp, ci, ci]),
            "orc_echo_run_batch": (None, [vp, vp, vp, vp, ci, C.c_longlong, ci, ci]),
        }
        for name, (res, args) in sigs.items():
            if not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _i16(a):
    a = np.ascontiguousarray(a, dtype=np.int16)
    return a


class Sink:
    def __init__(self):
        self.p = lib().orc_sink_new()

    def __del__(self):
        try:
            lib().orc_sink_free(self.p)
        except Exception:
            pass

    def clear(self):
        lib().orc_sink_clear(self.p)

    def events(self):
        n = lib().orc_sink_count(self.p)
        if n == 0:
            return np.zeros(0, EVENT_DTYPE)
        addr = lib().orc_sink_events(self.p)
        buf = (C.c_char*(n*EVENT_DTYPE.itemsize)).from_address(addr)
        return np.frombuffer(buf, dtype=EVENT_DTYPE).copy()

    def text(self):
        return lib().orc_sink_text(self.p).decode("latin1")

    def want_qam(self, on=True):
        lib().orc_sink_want_qam(self.p, int(on))


def qam_stream(events):
    """The qam_report_handler_t calls among a modem's events as uint32 [n, 7]: events (put_bit calls) before the report,
    1 if its pointers were NULL, symbol, constel re / im and target re / im as bits -- the device's record layout."""
    out = []
    pos = 0
    pend = None
    for e in events:
        if e["kind"] == 3:
            pos += 1
        elif e["kind"] == 6:
            pend = (pos, int(e["a"]), int(e["b"]), int(e["c"]))
        elif e["kind"] == 7:
            out.append((pend[0], int(e["a"]), pend[1], pend[2], pend[3], int(e["b"]), int(e["c"])))
    return np.array(out, np.int64).astype(np.uint32).reshape(-1, 7)


class _Detector:
    """Common shape: opaque state blob + rx(amp) -> per-block trace."""
    _rx = None

    def __init__(self, size):
        self.buf = np.zeros(size + 16, np.uint8)
        self.p = self.buf.ctypes.data
        self.sink = Sink()

    def rx(self, amp, want_blocks=True):
        amp = _i16(amp)
        maxb = len(amp)//64 + 4 if want_blocks else 0
        blocks = np.zeros(maxb, BLOCK_DTYPE)
        n = getattr(lib(), self._rx)(self.p, amp.ctypes.data, len(amp), self.sink.p,
                                     blocks.ctypes.data if want_blocks else None, maxb)
        return blocks[:n] if want_blocks else n


class Dtmf(_Detector):
    _rx = "orc_dtmf_rx"
    # field offsets of orc_dtmf_t (floats then ints), see oracle.h
    def __init__(self, mode=0):
        super().__init__(lib().orc_dtmf_sizeof())
        lib().orc_dtmf_init(self.p, mode)

    def parms(self, filter_dialtone=-1, twist=-1.0, reverse_twist=-1.0, threshold=-99.0):
        lib().orc_dtmf_parms(self.p, filter_dialtone, twist, reverse_twist, threshold)

    def get(self, maxlen=128):
        out = C.create_string_buffer(maxlen + 1)
        lib().orc_dtmf_get(self.p, out, maxlen)
        return out.value.decode("latin1")

    def status(self):
        return lib().orc_dtmf_status(self.p)

    def fillin(self, samples):
        lib().orc_dtmf_fillin(self.p, samples)

    def snapshot(self):
        f = self.buf[:4*35].view(np.float32)
        i = self.buf[4*35:4*35 + 4*9].view(np.int32)
        return {
            "fac": f[0:8].copy(), "v2": f[8:16].copy(), "v3": f[16:24].copy(), "energy": float(f[24]),
            "threshold": float(f[25]), "normal_twist": float(f[26]), "reverse_twist": float(f[27]),
            "z350": f[28:30].copy(), "z440": f[30:32].copy(),
            "filter_dialtone": int(f[32:33].view(np.int32)[0]),
            "current_sample": int(f[33:34].view(np.int32)[0]),
            "duration": int(f[34:35].view(np.int32)[0]),
            "last_hit": int(i[0]), "in_digit": int(i[1]), "mode": int(i[2]),
            "lost_digits": int(i[3]), "current_digits": int(i[4]),
        }


class BellMf(_Detector):
    _rx = "orc_bell_mf_rx"

    def __init__(self, mode=0):
        super().__init__(lib().orc_bell_mf_sizeof())
        lib().orc_bell_mf_init(self.p, mode)

    def get(self, maxlen=128):
        out = C.create_string_buffer(maxlen + 1)
        lib().orc_bell_mf_get(self.p, out, maxlen)
        return out.value.decode("latin1")

    def snapshot(self):
        f = self.buf[:4*18].view(np.float32)
        i = self.buf[4*18:4*18 + 4*9].view(np.int32)
        return {"fac": f[0:6].copy(), "v2": f[6:12].copy(), "v3": f[12:18].copy(),
                "hits": i[0:5].copy(), "current_sample": int(i[5]), "mode": int(i[6]),
                "lost_digits": int(i[7]), "current_digits": int(i[8])}


class R2Mf(_Detector):
    _rx = "orc_r2_mf_rx"

    def __init__(self, fwd=True, use_callback=True):
        super().__init__(lib().orc_r2_mf_sizeof())
        lib().orc_r2_mf_init(self.p, int(fwd), int(use_callback))

    def snapshot(self):
        f = self.buf[:4*18].view(np.float32)
        i = self.buf[4*18:4*18 + 4*4].view(np.int32)
        return {"fac": f[0:6].copy(), "v2": f[6:12].copy(), "v3": f[12:18].copy(),
                "fwd": int(i[0]), "current_sample": int(i[1]), "current_digit": int(i[2])}


class SuperToneDesc:
    def __init__(self):
        self.buf = np.zeros(lib().orc_st_desc_sizeof() + 16, np.uint8)
        self.p = self.buf.ctypes.data
        lib().orc_st_desc_init(self.p)

    def add_tone(self):
        return lib().orc_st_add_tone(self.p)

    def add_element(self, tone, f1, f2, min_ms, max_ms):
        return lib().orc_st_add_element(self.p, tone, f1, f2, min_ms, max_ms)

    @property
    def monitored(self):
        return int(self.buf[4:8].view(np.int32)[0])

    @property
    def fac(self):
        off = 4*(2 + 128)
        return self.buf[off:off + 4*64].view(np.float32)[:self.monitored].copy()


class SuperTone(_Detector):
    _rx = "orc_st_rx"

    def __init__(self, desc, use_segment_cb=False):
        super().__init__(lib().orc_st_sizeof())
        self.desc = desc
        lib().orc_st_init(self.p, desc.p, int(use_segment_cb))


class Goertzel:
    def __init__(self, freq, samples):
        self.buf = np.zeros(32, np.uint8)
        self.p = self.buf.ctypes.data
        lib().orc_goertzel_init(self.p, freq, samples)

    def update(self, amp):
        amp = _i16(amp)
        return lib().orc_goertzel_update(self.p, amp.ctypes.data, len(amp))

    def result(self):
        return lib().orc_goertzel_result(self.p)


def goertzel_fac(freq):
    return lib().orc_goertzel_fac(freq)


V18_TONE_SET = [390.0, 980.0, 1180.0, 1270.0, 1300.0, 1400.0, 1650.0, 1800.0, 2225.0]      # v18.c:200-211
ADEMCO_TONE_SET = [1400.0, 2300.0]                                                           # ademco_contactid.c:1179-1180


def tone_functor_blocks(kind, amp, threshold=0.0):
    """Raw block decisions of the v18.c (kind 1, 102-sample blocks) or ademco_contactid.c (kind 2, 55) tone scan over the
    whole blocks of amp."""
    freqs = np.array(V18_TONE_SET if kind == 1 else ADEMCO_TONE_SET, np.float32)
    block = 102 if kind == 1 else 55
    amp = _i16(amp)
    nb = len(amp)//block
    out = np.zeros(nb, np.int32)
    L = lib()
    L.orc_tone_functor_blocks.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_tone_functor_blocks(kind, freqs.ctypes.data, len(freqs), block, threshold, amp.ctypes.data, nb, out.ctypes.data)
    return out


ECHO_FIELDS = ["tx_power0", "tx_power1", "tx_power2", "tx_power3", "rx_power0", "rx_power1", "rx_power2",
               "clean_rx_power", "rx_power_threshold", "nonupdate_dwell", "curr_pos", "taps", "tap_mask",
               "adaption_mode", "supp_test1", "supp_test2", "supp1", "supp2", "vad", "cng", "geigel_max",
               "geigel_lag", "dtd_onset", "tap_set", "tap_rotate_counter", "latest_correction",
               "narrowband_count", "narrowband_score", "fir_curr_pos", "tx_hpf0", "tx_hpf1", "rx_hpf0",
               "rx_hpf1", "cng_level", "cng_rndnum", "cng_filter", "fir_set"]
ECHO_MAX_TAPS = 1024


class EchoCan:
    def __init__(self, taps, mode):
        self.buf = np.zeros(lib().orc_echo_sizeof() + 16, np.uint8)
        self.p = self.buf.ctypes.data
        assert lib().orc_echo_init(self.p, taps, mode) == 0
        self.taps = taps

    def flush(self):
        lib().orc_echo_flush(self.p)

    def adaption_mode(self, mode):
        lib().orc_echo_adaption_mode(self.p, mode)

    def run(self, tx, rx, use_hpf_tx=False):
        tx = _i16(tx)
        rx = _i16(rx)
        out = np.zeros(len(tx), np.int16)
        lib().orc_echo_run(self.p, tx.ctypes.data, rx.ctypes.data, out.ctypes.data, len(tx), int(use_hpf_tx))
        return out

    def snapshot(self):
        n = len(ECHO_FIELDS)
        w = self.buf[:4*(n + 9)].view(np.int32)
        d = {k: int(v) for k, v in zip(ECHO_FIELDS, w[:n])}
        d["last_acf"] = w[n:n + 9].copy()
        off = 4*(n + 9)
        d["taps32"] = self.buf[off:off + 4*ECHO_MAX_TAPS].view(np.int32)[:self.taps].copy()
        off += 4*ECHO_MAX_TAPS
        d["taps16"] = self.buf[off:off + 2*4*ECHO_MAX_TAPS].view(np.int16).reshape(4, ECHO_MAX_TAPS)[:, :self.taps].copy()
        off += 2*4*ECHO_MAX_TAPS
        d["history"] = self.buf[off:off + 2*ECHO_MAX_TAPS].view(np.int16)[:self.taps].copy()
        return d


class _ModemTables(C.Structure):
    _fields_ = [("rrc_re", C.c_void_p), ("rrc_im", C.c_void_p), ("sine", C.c_void_p), ("sqrt_tab", C.c_void_p),
                ("godard", C.c_float*7), ("coarse_trigger", C.c_float), ("fine_trigger", C.c_float),
                ("coarse_step", C.c_int), ("fine_step", C.c_int),
                ("v27_4800_re", C.c_void_p), ("v27_4800_im", C.c_void_p),
                ("v27_2400_re", C.c_void_p), ("v27_2400_im", C.c_void_p),
                ("v17_re", C.c_void_p), ("v17_im", C.c_void_p),
                ("v17_godard", C.c_float*7), ("v17_coarse_trigger", C.c_float), ("v17_fine_trigger", C.c_float),
                ("v17_coarse_step", C.c_int), ("v17_fine_step", C.c_int),
                ("v17_constellation", C.c_void_p), ("v17_maps", C.c_void_p), ("v17_map_4800", C.c_void_p)]


_tables_keepalive = None


def set_modem_tables(t):
    """t: dict of numpy arrays as returned by oracle.ref.modem_tables() / the golden fixture."""
    global _tables_keepalive
    keep = {k: np.ascontiguousarray(t[k]) for k in ("rrc_re", "rrc_im", "sine", "sqrt_tab")}
    m = _ModemTables()
    m.rrc_re = keep["rrc_re"].ctypes.data
    m.rrc_im = keep["rrc_im"].ctypes.data
    m.sine = keep["sine"].ctypes.data
    m.sqrt_tab = keep["sqrt_tab"].ctypes.data
    for i in range(7):
        m.godard[i] = float(t["godard"][i])
    m.coarse_trigger = float(t["godard"][7])
    m.fine_trigger = float(t["godard"][8])
    m.coarse_step = int(t["steps"][0])
    m.fine_step = int(t["steps"][1])
    for k in ("v27_4800_re", "v27_4800_im", "v27_2400_re", "v27_2400_im", "v17_re", "v17_im", "v17_constellation",
              "v17_maps", "v17_map_4800"):
        if k in t:
            keep[k] = np.ascontiguousarray(t[k])
            setattr(m, k, keep[k].ctypes.data)
    if "v17_godard" in t:
        for i in range(7):
            m.v17_godard[i] = float(t["v17_godard"][i])
        m.v17_coarse_trigger = float(t["v17_godard"][7])
        m.v17_fine_trigger = float(t["v17_godard"][8])
        m.v17_coarse_step = int(t["v17_steps"][0])
        m.v17_fine_step = int(t["v17_steps"][1])
    lib().orc_modem_set_tables(C.byref(m))
    _tables_keepalive = (keep, m)


class V29:
    N_FLOATS = 238
    N_INTS = 43

    def __init__(self, bit_rate=9600):
        self.buf = np.zeros(lib().orc_v29_sizeof() + 16, np.uint8)
        self.p = self.buf.ctypes.data
        self.sink = Sink()
        assert lib().orc_v29_init(self.p, bit_rate) == 0

    def tap_qam(self):
        self.sink.want_qam(True)

    def set_signal_cutoff(self, cutoff_dbm0):
        """v29_rx_set_signal_cutoff(), v29rx.c:163-169 (init gives -28.5 dBm0; fax_modems.c:416 sets -45.5)"""
        fn = lib().orc_v29_set_signal_cutoff
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_float]
        fn(self.p, cutoff_dbm0)

    def rx(self, amp):
        amp = _i16(amp)
        return lib().orc_v29_rx(self.p, amp.ctypes.data, len(amp), self.sink.p)

    def snapshot(self):
        f = self.buf[:4*self.N_FLOATS].view(np.float32).copy()
        w = self.buf[4*self.N_FLOATS:4*(self.N_FLOATS + self.N_INTS)].view(np.int32).copy()
        return f, w


class V27ter:
    N_FLOATS = 225
    N_INTS = 45

    def __init__(self, bit_rate=4800):
        self.buf = np.zeros(lib().orc_v27ter_sizeof() + 16, np.uint8)
        self.p = self.buf.ctypes.data
        self.sink = Sink()
        assert lib().orc_v27ter_init(self.p, bit_rate) == 0

    def tap_qam(self):
        self.sink.want_qam(True)

    def rx(self, amp):
        amp = _i16(amp)
        return lib().orc_v27ter_rx(self.p, amp.ctypes.data, len(amp), self.sink.p)

    def snapshot(self):
        f = self.buf[:4*self.N_FLOATS].view(np.float32).copy()
        w = self.buf[4*self.N_FLOATS:4*(self.N_FLOATS + self.N_INTS)].view(np.int32).copy()
        return f, w


class V17:
    N_FLOATS = 246
    N_INTS = 301

    def __init__(self, bit_rate=14400):
        self.buf = np.zeros(lib().orc_v17_sizeof() + 16, np.uint8)
        self.p = self.buf.ctypes.data
        self.sink = Sink()
        assert lib().orc_v17_init(self.p, bit_rate) == 0

    def restart(self, bit_rate, short_train):
        return lib().orc_v17_restart(self.p, bit_rate, int(short_train))

    def tap_qam(self):
        self.sink.want_qam(True)

    def rx(self, amp):
        amp = _i16(amp)
        return lib().orc_v17_rx(self.p, amp.ctypes.data, len(amp), self.sink.p)

    def snapshot(self):
        f = self.buf[:4*self.N_FLOATS].view(np.float32).copy()
        w = self.buf[4*self.N_FLOATS:4*(self.N_FLOATS + self.N_INTS)].view(np.int32).copy()
        return f, w


# ---- signal sources (tonegen_oracle.c) ------------------------------------------
class _Tone(C.Structure):
    _fields_ = [("phase_rate", C.c_int32), ("gain", C.c_float)]


class ToneDesc(C.Structure):
    _fields_ = [("tone", _Tone*4), ("duration", C.c_int32*4), ("repeat", C.c_int32)]


class ToneGenState(C.Structure):
    _fields_ = [("tone", _Tone*4), ("phase", C.c_uint32*4), ("duration", C.c_int32*4), ("repeat", C.c_int32),
                ("current_section", C.c_int32), ("current_position", C.c_int32)]


class _DigitQueue(C.Structure):
    _fields_ = [("data", C.c_uint8*128), ("rd", C.c_int32), ("count", C.c_int32)]


class _DtmfTxState(C.Structure):
    _fields_ = [("tones", ToneGenState), ("low_level", C.c_float), ("high_level", C.c_float),
                ("on_time", C.c_int32), ("off_time", C.c_int32), ("queue", _DigitQueue)]


class _BellMfTxState(C.Structure):
    _fields_ = [("tones", ToneGenState), ("queue", _DigitQueue)]


class _R2MfTxState(C.Structure):
    _fields_ = [("tone", ToneGenState), ("fwd", C.c_int32), ("digit", C.c_int32)]


def _tx_call(fn, state, n):
    buf = np.zeros(max(n, 1), np.int16)
    fn.restype = C.c_int
    got = fn(C.byref(state), C.c_void_p(buf.ctypes.data), C.c_int(n))
    return buf[:got].copy()


def tone_desc(f1, l1, f2, l2, d1, d2=0, d3=0, d4=0, repeat=False):
    d = ToneDesc()
    lib().orc_tone_desc_init(C.byref(d), f1, l1, f2, l2, d1, d2, d3, d4, int(repeat))
    return d


class ToneGen:
    def __init__(self, desc):
        self.s = ToneGenState()
        lib().orc_tone_gen_init(C.byref(self.s), C.byref(desc))

    def tx(self, n):
        return _tx_call(lib().orc_tone_gen, self.s, n)


class DtmfTx:
    def __init__(self):
        self.s = _DtmfTxState()
        lib().orc_dtmf_tx_init(C.byref(self.s))

    def set_level(self, level, twist):
        lib().orc_dtmf_tx_set_level(C.byref(self.s), level, twist)

    def set_timing(self, on_ms, off_ms):
        lib().orc_dtmf_tx_set_timing(C.byref(self.s), on_ms, off_ms)

    def put(self, digits):
        b = digits if isinstance(digits, bytes) else digits.encode()
        return lib().orc_dtmf_tx_put(C.byref(self.s), b, len(b))

    def tx(self, n):
        return _tx_call(lib().orc_dtmf_tx, self.s, n)


class BellMfTx:
    def __init__(self):
        self.s = _BellMfTxState()
        lib().orc_bell_mf_tx_init(C.byref(self.s))

    def put(self, digits):
        b = digits if isinstance(digits, bytes) else digits.encode()
        return lib().orc_bell_mf_tx_put(C.byref(self.s), b, len(b))

    def tx(self, n):
        return _tx_call(lib().orc_bell_mf_tx, self.s, n)


class R2MfTx:
    def __init__(self, fwd=True):
        self.s = _R2MfTxState()
        lib().orc_r2_mf_tx_init(C.byref(self.s), int(fwd))

    def put(self, digit):
        b = digit if isinstance(digit, bytes) else digit.encode()
        return lib().orc_r2_mf_tx_put(C.byref(self.s), C.c_char(b[:1] if b else b"\0"))

    def tx(self, n):
        return _tx_call(lib().orc_r2_mf_tx, self.s, n)


# ---- FSK receiver (fsk_oracle.c) --------------------------------------------------------
def fsk_preset(which):
    out = np.zeros(5, np.int32)
    assert lib().orc_fsk_preset(which, out.ctypes.data) == 0
    return out


class Fsk:
    SCALARS = 28

    def __init__(self, which, framing_mode):
        self.buf = np.zeros(lib().orc_fsk_sizeof()//4, np.int32)
        self.p = self.buf.ctypes.data
        self.sink = Sink()
        self.spec = fsk_preset(which)
        assert lib().orc_fsk_init(self.p, self.spec.ctypes.data, framing_mode) == 0

    def restart(self, which, framing_mode):
        self.spec = fsk_preset(which)
        return lib().orc_fsk_restart(self.p, self.spec.ctypes.data, framing_mode)

    def set_signal_cutoff(self, cutoff):
        lib().orc_fsk_set_signal_cutoff(self.p, cutoff)

    def set_frame_parameters(self, data_bits, parity, stop_bits):
        lib().orc_fsk_set_frame_parameters(self.p, data_bits, parity, stop_bits)

    def rx(self, amp):
        amp = _i16(amp)
        return lib().orc_fsk_rx(self.p, amp.ctypes.data, len(amp), self.sink.p)

    def fillin(self, n):
        return lib().orc_fsk_fillin(self.p, n)

    def snapshot(self):
        span = int(self.buf[15])
        return self.buf[:self.SCALARS + 4*span].copy()


# ---- modem connect tones (mct_oracle.c) ---------------------------------------------------
class Mct:
    WORDS = 18

    def __init__(self, tone_type, use_callback=True):
        self.buf = np.zeros(lib().orc_mct_sizeof()//4 + 2, np.int32)
        self.p = self.buf.ctypes.data
        self.sink = Sink()
        lib().orc_mct_init(self.p, tone_type, self.sink.p if use_callback else None)

    def rx(self, amp):
        amp = _i16(amp)
        return lib().orc_mct_rx(self.p, amp.ctypes.data, len(amp))

    def get(self):
        return lib().orc_mct_get(self.p)

    def snapshot(self):
        w = self.buf[:self.WORDS].copy()
        if int(w[0]) in (6, 7):
            span = int(self.buf[self.WORDS + 15])
            return np.concatenate([w, self.buf[self.WORDS:self.WORDS + 28 + 4*span]])
        return w


# ---- in-band signalling tones (sigtone_oracle.c) ------------------------------------------
class SigToneRx:
    WORDS = 27

    def __init__(self, tone_type, mode=0):
        self.buf = np.zeros(lib().orc_sigtone_rx_sizeof()//4 + 2, np.int32)
        self.p = self.buf.ctypes.data
        self.sink = Sink()
        if lib().orc_sigtone_rx_init(self.p, tone_type, self.sink.p) != 0:
            raise ValueError("not a signalling tone type: %d" % tone_type)
        self.set_mode(mode)

    def set_mode(self, mode):
        lib().orc_sigtone_rx_set_mode(self.p, mode)

    def script(self, modes):
        """modes set from inside the reports, one each, as a caller's callback might"""
        self._script = np.ascontiguousarray(modes, np.int32)
        lib().orc_sigtone_rx_script(self.p, self._script.ctypes.data, len(self._script))

    def rx(self, amp):
        buf = _i16(amp).copy()
        lib().orc_sigtone_rx(self.p, buf.ctypes.data, len(buf))
        return buf

    def snapshot(self):
        return self.buf[:self.WORDS].copy()

    def thresholds(self):
        return self.buf[self.WORDS + 1:self.WORDS + 4].copy()


def sigtone_rx_thresholds(tone_type):
    out = np.zeros(3, np.int32)
    lib().orc_sigtone_rx_thresholds(tone_type, out.ctypes.data)
    return out


class SigToneTx:
    WORDS = 5

    def __init__(self, tone_type, script=()):
        self.buf = np.zeros(lib().orc_sigtone_tx_sizeof()//4 + 2, np.int32)
        self.p = self.buf.ctypes.data
        self.sink = Sink()
        if lib().orc_sigtone_tx_init(self.p, tone_type, self.sink.p) != 0:
            raise ValueError("not a signalling tone type: %d" % tone_type)
        self.script = np.ascontiguousarray(np.asarray(script, np.int32).reshape(-1, 2))
        lib().orc_sigtone_tx_script(self.p, self.script.ctypes.data, len(self.script))

    def set_mode(self, mode, duration):
        lib().orc_sigtone_tx_set_mode(self.p, mode, duration)

    def tx(self, amp):
        buf = _i16(amp).copy()
        lib().orc_sigtone_tx(self.p, buf.ctypes.data, len(buf))
        return buf

    def requests(self):
        return len(self.sink.events())

    def snapshot(self):
        """phase_acc[2], high_low_timer, current_tx_tone, current_tx_timeout, phase_rate[2], the four scalings"""
        w = self.buf[:7].copy()
        sc = self.buf[7:9].view(np.int16).astype(np.int32)
        return np.concatenate([w, sc])


# ---- V.29 transmitter (v29tx_oracle.c) ----------------------------------------------------------
def set_v29_tx_table(table):
    t = np.ascontiguousarray(table, np.float32)
    assert t.size == 90
    lib().orc_v29_tx_set_table(t.ctypes.data)


class V29Tx:
    WORDS = 32

    def __init__(self, bit_rate, tep=False, seed=1):
        self.buf = np.zeros(self.WORDS, np.uint32)
        self.p = self.buf.ctypes.data
        assert lib().orc_v29_tx_sizeof() == 4*self.WORDS
        assert lib().orc_v29_tx_init(self.p, bit_rate, int(tep), seed & 0x7FFF) == 0

    def power(self, level_dbm0):
        lib().orc_v29_tx_power(self.p, level_dbm0)

    def restart(self, bit_rate, tep):
        return lib().orc_v29_tx_restart(self.p, bit_rate, int(tep))

    def tx(self, n):
        out = np.zeros(max(n, 1), np.int16)
        got = lib().orc_v29_tx(self.p, out.ctypes.data, n)
        return out[:got].copy()

    def snapshot(self):
        return self.buf.copy()


def set_v27ter_tx_tables(t4800, t2400):
    a = np.ascontiguousarray(t4800, np.float32)
    b = np.ascontiguousarray(t2400, np.float32)
    assert a.size == 45 and b.size == 180
    lib().orc_v27ter_tx_set_tables(a.ctypes.data, b.ctypes.data)


class V27terTx(V29Tx):
    def __init__(self, bit_rate, tep=False, seed=1):
        self.buf = np.zeros(self.WORDS, np.uint32)
        self.p = self.buf.ctypes.data
        assert lib().orc_v27ter_tx_sizeof() == 4*self.WORDS
        assert lib().orc_v27ter_tx_init(self.p, bit_rate, int(tep), seed & 0x7FFF) == 0

    def power(self, level_dbm0):
        lib().orc_v27ter_tx_power(self.p, level_dbm0)

    def restart(self, bit_rate, tep):
        return lib().orc_v27ter_tx_restart(self.p, bit_rate, int(tep))

    def tx(self, n):
        out = np.zeros(max(n, 1), np.int16)
        got = lib().orc_v27ter_tx(self.p, out.ctypes.data, n)
        return out[:got].copy()


class V17Tx(V29Tx):
    """Needs set_modem_tables() (constellations, sine) and set_v29_tx_table() (the shared 10 x 9 pulse shaper)."""

    def __init__(self, bit_rate, tep=False, seed=1):
        self.buf = np.zeros(self.WORDS, np.uint32)
        self.p = self.buf.ctypes.data
        assert lib().orc_v17_tx_sizeof() == 4*self.WORDS
        assert lib().orc_v17_tx_init(self.p, bit_rate, int(tep), seed & 0x7FFF) == 0

    def power(self, level_dbm0):
        lib().orc_v17_tx_power(self.p, level_dbm0)

    def restart(self, bit_rate, tep, short_train=False):
        return lib().orc_v17_tx_restart(self.p, bit_rate, int(tep), int(short_train))

    def tx(self, n):
        out = np.zeros(max(n, 1), np.int16)
        got = lib().orc_v17_tx(self.p, out.ctypes.data, n)
        return out[:got].copy()


# ---- AWGN (awgn_oracle.c) ---------------------------------------------------------------------
class Awgn:
    WORDS = 2*(2 + 97) + 4

    def __init__(self, seed, level_dbm0):
        assert lib().orc_awgn_sizeof() == 4*self.WORDS
        self.buf = np.zeros(self.WORDS, np.uint32)
        self.p = self.buf.ctypes.data
        lib().orc_awgn_init_dbm0(self.p, seed, level_dbm0)

    def gen(self, n):
        out = np.zeros(n, np.int16)
        lib().orc_awgn_block(self.p, out.ctypes.data, n)
        return out

    def snapshot(self):
        return self.buf.copy()
