"""ctypes harness over libspangpu.so (include/spangpu.h).

Plumbing only: every call goes to the HIP library; nothing is computed here.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PRIMS_LIB_PATH = os.path.join(HERE, "libspangpu_prims.so")        # the opt-in library of spandsp-named primitives (include/spangpu_prims.h)
LIB_PATH = os.environ.get("SPANGPU_LIB", os.path.join(HERE, "libspangpu.so"))      # the variable: instrumented builds (tools/quad_prof.py)

# include/spangpu.h
DTMF, BELL_MF, R2_MF, SUPER_TONE, GOERTZEL, V29, V27TER, V17, ECHO = range(1, 10)
CADENCE_TONE_ON, CADENCE_TONE_OFF, CADENCE_SEGMENT = 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1
CHANNEL_MAJOR, SAMPLE_MAJOR = 0, 1
REPORT_DIGITS, REPORT_REALTIME = 0, 2
BLK_VALID, BLK_CHANGE, BLK_REPORT, BLK_TONE_OFF = 1, 2, 4, 8
MAX_BINS = 64

BLOCK_DTYPE = np.dtype([("channel", "<i4"), ("block", "<i4"), ("hit", "<i4"), ("code", "<i4"),
                        ("flags", "<i4"), ("duration", "<i4"), ("energy", "<f4")])


class ToneParams(C.Structure):
    _fields_ = [("report_mode", C.c_int32), ("filter_dialtone", C.c_int32),
                ("twist_db", C.c_float), ("reverse_twist_db", C.c_float), ("threshold_dbm0", C.c_float),
                ("r2_fwd", C.c_int32), ("n_bins", C.c_int32), ("block_len", C.c_int32),
                ("bin_fac", C.c_float*MAX_BINS), ("trace", C.c_int32), ("set_mask", C.c_int32),
                ("functor", C.c_int32), ("functor_threshold", C.c_float)]


TP_TWIST, TP_REVERSE_TWIST, TP_THRESHOLD = 1, 2, 4
FUNCTOR_NONE, FUNCTOR_V18, FUNCTOR_ADEMCO = 0, 1, 2


class SpanGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("spangpu error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load libspangpu.so; raise loudly if it is missing (there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, ci, cf, ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
        sigs = {
            "spangpu_device_count": (ci, []),
            "spangpu_probe_stream_read": (ci, [ci, C.c_size_t, ci, C.POINTER(C.c_double)]),
            "spangpu_last_error": (C.c_char_p, []),
            "spangpu_version": (C.c_char_p, []),
            "spangpu_goertzel_fac": (cf, [cf]),
            "spangpu_tune_lanes_per_channel": (ci, [ci]),
            "spangpu_tune_tone_kernel": (ci, [ci]),
            "spangpu_bank_create": (ci, [C.POINTER(vp), ci, ci, ci, vp, C.c_size_t]),
            "spangpu_bank_destroy": (ci, [vp]),
            "spangpu_bank_kind": (ci, [vp]),
            "spangpu_bank_channels": (ci, [vp]),
            "spangpu_bank_set_stream": (ci, [vp, vp]),
            "spangpu_bank_set_queues": (ci, [vp, ci]),
            "spangpu_bank_join": (ci, [vp]),
            "spangpu_bank_rx": (ci, [vp, vp, ci, ci, ci, ll]),
            "spangpu_bank_rx_var": (ci, [vp, vp, ci, vp, ci, ll]),
            "spangpu_bank_set_channel_params": (ci, [vp, ci, vp, C.c_size_t]),
            "spangpu_bank_sync": (ci, [vp]),
            "spangpu_bank_blocks": (ci, [vp, vp, ci]),
            "spangpu_bank_trace": (ci, [vp, vp, C.c_size_t]),
            "spangpu_bank_copy_records": (ll, [vp, vp, C.c_size_t]),
            "spangpu_bank_digit_events": (ci, [vp, vp, ci]),
            "spangpu_bank_set_cadences": (ci, [vp, vp, ci, vp, ci]),
            "spangpu_bank_cadence_run": (ci, [vp]),
            "spangpu_bank_cadence_events": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_bank_cadence_device": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_bank_cadence_list": (ci, [vp, C.POINTER(vp)]),
            "spangpu_bank_cadence_reset": (ci, [vp, ci]),
            "spangpu_bank_cadence_state_words": (ci, []),
            "spangpu_bank_cadence_get_state": (ci, [vp, ci, vp]),
            "spangpu_bank_cadence_set_state": (ci, [vp, ci, vp]),
            "spangpu_bank_set_digits_buffer": (ci, [vp, vp, C.c_size_t]),
            "spangpu_bank_set_digits_ring": (ci, [vp, vp, C.c_size_t, ci]),
            "spangpu_bank_reset_channel": (ci, [vp, ci, ci]),
            "spangpu_bank_get_state": (ci, [vp, ci, vp, ci, vp, ci]),
            "spangpu_bank_set_state": (ci, [vp, ci, vp, ci, vp, ci]),
            "spangpu_bank_last_kernel_ms": (cf, [vp]),
            "spangpu_bank_set_timing": (ci, [vp, ci]),
            "spangpu_bank_bins": (ci, [vp]),
            "spangpu_bank_force_block": (ci, [vp]),
            "spangpu_banks_rx": (ci, [vp, vp, ci, ci, vp]), "spangpu_banks_own_queues": (ci, [vp, ci]),
            "spangpu_bank_rx_g711": (ci, [vp, vp, ci, ci, ci, ll]),
            "spangpu_bank_set_records_buffer": (ci, [vp, vp, C.c_size_t]),
            "spangpu_bank_get_stream": (vp, [vp]), "spangpu_echo_get_stream": (vp, [vp]), "spangpu_modem_get_stream": (vp, [vp]),
            "spangpu_modemtx_create": (ci, [C.POINTER(vp), ci, ci, ci, ci, ci, vp]),
            "spangpu_modemtx_destroy": (None, [vp]),
            "spangpu_modemtx_channels": (ci, [vp]),
            "spangpu_modemtx_set_stream": (ci, [vp, vp]),
            "spangpu_modemtx_sync": (ci, [vp]),
            "spangpu_modemtx_power": (ci, [vp, ci, cf]),
            "spangpu_modemtx_line": (ci, [vp, vp, vp]),
            "spangpu_modemtx_restart": (ci, [vp, ci, ci, ci]),
            "spangpu_modemtx_restart_ex": (ci, [vp, ci, ci, ci, ci]),
            "spangpu_modemtx_tx": (ci, [vp, ci, vp, ll, ci]),
            "spangpu_modemtx_state_words": (ci, []),
            "spangpu_modemtx_get_state": (ci, [vp, ci, vp]),
            "spangpu_modemtx_table": (ci, [ci, vp, ci]),
            "spangpu_awgn_create": (ci, [C.POINTER(vp), ci, ci, vp, vp]),
            "spangpu_awgn_destroy": (None, [vp]),
            "spangpu_awgn_channels": (ci, [vp]),
            "spangpu_awgn_set_stream": (ci, [vp, vp]),
            "spangpu_awgn_sync": (ci, [vp]),
            "spangpu_awgn_reinit": (ci, [vp, ci, ci, cf]),
            "spangpu_awgn_tx": (ci, [vp, ci, vp, ll, ci, ci]),
            "spangpu_awgn_state_words": (ci, [vp]),
            "spangpu_awgn_get_state": (ci, [vp, ci, vp]),
            "spangpu_mct_create": (ci, [C.POINTER(vp), ci, ci, ci, ci]),
            "spangpu_mct_destroy": (None, [vp]),
            "spangpu_mct_channels": (ci, [vp]),
            "spangpu_mct_set_stream": (ci, [vp, vp]),
            "spangpu_mct_sync": (ci, [vp]),
            "spangpu_mct_rx": (ci, [vp, vp, ci, ci, ll]),
            "spangpu_mct_rx_var": (ci, [vp, vp, ci, vp, ci, ll]),
            "spangpu_mct_events": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_mct_get": (ci, [vp, ci]),
            "spangpu_mct_state_words": (ci, [vp]),
            "spangpu_mct_get_state": (ci, [vp, ci, vp]),
            "spangpu_sigtone_rx_create": (ci, [C.POINTER(vp), ci, ci, ci]),
            "spangpu_sigtone_rx_destroy": (None, [vp]),
            "spangpu_sigtone_rx_channels": (ci, [vp]),
            "spangpu_sigtone_rx_set_stream": (ci, [vp, vp]),
            "spangpu_sigtone_rx_sync": (ci, [vp]),
            "spangpu_sigtone_rx_set_mode": (ci, [vp, ci, ci]),
            "spangpu_sigtone_rx": (ci, [vp, vp, ci, ci, ll]),
            "spangpu_sigtone_rx_var": (ci, [vp, vp, ci, vp, ci, ll]),
            "spangpu_sigtone_rx_events": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_sigtone_rx_state_words": (ci, [vp]),
            "spangpu_sigtone_rx_get_state": (ci, [vp, ci, vp]),
            "spangpu_sigtone_rx_set_state": (ci, [vp, ci, vp]),
            "spangpu_sigtone_rx_thresholds": (ci, [vp, vp]),
            "spangpu_sigtone_tx_create": (ci, [C.POINTER(vp), ci, ci, ci]),
            "spangpu_sigtone_tx_destroy": (None, [vp]),
            "spangpu_sigtone_tx_channels": (ci, [vp]),
            "spangpu_sigtone_tx_set_stream": (ci, [vp, vp]),
            "spangpu_sigtone_tx_set_mode": (ci, [vp, ci, ci, ci]),
            "spangpu_sigtone_tx_set_modes": (ci, [vp, vp, vp]),
            "spangpu_sigtone_tx": (ci, [vp, vp, ci, ci, ll]),
            "spangpu_sigtone_tx_continue": (ci, [vp, vp, ci, ll]),
            "spangpu_sigtone_tx_requests": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_sigtone_tx_state_words": (ci, [vp]),
            "spangpu_sigtone_tx_get_state": (ci, [vp, ci, vp]),
            "spangpu_fsk_preset": (ci, [ci, vp]),
            "spangpu_fsk_create": (ci, [C.POINTER(vp), ci, ci, vp, ci]),
            "spangpu_fsk_destroy": (None, [vp]),
            "spangpu_fsk_channels": (ci, [vp]),
            "spangpu_fsk_set_stream": (ci, [vp, vp]),
            "spangpu_fsk_sync": (ci, [vp]),
            "spangpu_fsk_rx": (ci, [vp, vp, ci, ci, ll]),
            "spangpu_fsk_rx_var": (ci, [vp, vp, ci, vp, ci, ll]),
            "spangpu_fsk_events": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_fsk_state_words": (ci, [vp]),
            "spangpu_fsk_get_state": (ci, [vp, ci, vp]),
            "spangpu_fsk_set_state": (ci, [vp, ci, vp]),
            "spangpu_fsk_restart": (ci, [vp, ci, ci]),
            "spangpu_fsk_set_signal_cutoff": (ci, [vp, ci, cf]),
            "spangpu_fsk_set_frame_parameters": (ci, [vp, ci, ci, ci, ci]),
            "spangpu_fsk_fillin": (ci, [vp, ci, ci]),
            "spangpu_tune_echo_lanes_per_channel": (ci, [ci]),
            "spangpu_feed_create": (ci, [C.POINTER(vp), vp, ci, ci, ci, ci]),
            "spangpu_feed_destroy": (ci, [vp]),
            "spangpu_feed_stride": (ll, [vp]),
            "spangpu_feed_acquire": (vp, [vp]),
            "spangpu_feed_commit": (ci, [vp, ci]),
            "spangpu_feed_collect": (ci, [vp, C.POINTER(vp)]),
            "spangpu_feed_outstanding": (ci, [vp]),
            "spangpu_feed_run": (ci, [vp, ci, ci, ci, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
            "spangpu_vec_circular_dot_prodf_batch": (ci, [ci, vp, ll, vp, ll, vp, vp, ci, ci, ci]),
            "spangpu_vec_circular_lmsf_batch": (ci, [ci, vp, ll, vp, ll, vp, vp, ci, ci, ci]),
            "spangpu_cvec_circular_dot_prodf_batch": (ci, [ci, vp, ll, vp, ll, vp, vp, ci, ci, ci]),
            "spangpu_cvec_circular_lmsf_batch": (ci, [ci, vp, ll, vp, ll, vp, vp, ci, ci, ci]),
            "spangpu_power_meter_update_batch": (ci, [ci, vp, ll, vp, vp, ci, ci, ci]),
            "spangpu_periodogram_batch": (ci, [ci, vp, ll, vp, ll, vp, ci, ci, ci]),
            "spangpu_periodogram_prepare_batch": (ci, [ci, vp, ll, vp, vp, ci, ci, ci]),
            "spangpu_periodogram_apply_batch": (ci, [ci, vp, ll, vp, vp, vp, ci, ci, ci]),
            "spangpu_periodogram_freq_error_batch": (ci, [ci, vp, cf, vp, vp, vp, ci, ci]),
            "spangpu_fixed_sqrt32_batch": (ci, [ci, vp, vp, ci, ci]),
            "spangpu_dds_complexf_batch": (ci, [ci, vp, vp, vp, ci, ci, ci]),
            "spangpu_arctan2_batch": (ci, [ci, vp, vp, vp, ci, ci]),
            "spangpu_tune_force_peer_copy": (ci, [ci]),
            "spangpu_shard_info": (ci, [vp, ci, vp]), "spangpu_echo_shard_info": (ci, [vp, ci, vp]), "spangpu_modem_shard_info": (ci, [vp, ci, vp]),
            "spangpu_shard_create": (ci, [C.POINTER(vp), vp, ci, ci, ci, ci, vp, C.c_size_t]),
            "spangpu_shard_destroy": (ci, [vp]),
            "spangpu_shard_count": (ci, [vp]),
            "spangpu_shard_channels": (ci, [vp]),
            "spangpu_shard_range": (ci, [vp, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]),
            "spangpu_shard_bank": (vp, [vp, ci]),
            "spangpu_shard_rx": (ci, [vp, vp, ci, ll]),
            "spangpu_shard_digits_device": (ci, [vp, vp, C.POINTER(vp), C.POINTER(ci), C.POINTER(ci)]),
            "spangpu_shard_digits_host": (ci, [vp, vp, C.c_size_t]),
            "spangpu_shard_sync": (ci, [vp]),
            "spangpu_echo_shard_create": (ci, [C.POINTER(vp), vp, ci, ci, ci, ci]),
            "spangpu_echo_shard_destroy": (ci, [vp]),
            "spangpu_echo_shard_count": (ci, [vp]),
            "spangpu_echo_shard_range": (ci, [vp, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]),
            "spangpu_echo_shard_bank": (vp, [vp, ci]),
            "spangpu_echo_shard_update": (ci, [vp, vp, vp, vp, ci, ll]),
            "spangpu_echo_shard_report": (ci, [vp, ci]),
            "spangpu_echo_shard_erle_device": (ci, [vp, vp, C.POINTER(vp), C.POINTER(ci)]),
            "spangpu_echo_shard_erle_host": (ci, [vp, vp, C.c_size_t]),
            "spangpu_echo_shard_sync": (ci, [vp]),
            "spangpu_modem_shard_create": (ci, [C.POINTER(vp), vp, ci, ci, ci, ci, ci]),
            "spangpu_modem_shard_destroy": (ci, [vp]),
            "spangpu_modem_shard_range": (ci, [vp, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]),
            "spangpu_modem_shard_bank": (vp, [vp, ci]),
            "spangpu_modem_shard_rx": (ci, [vp, vp, ci, ll]),
            "spangpu_modem_shard_events_host": (ci, [vp, vp, vp]),
            "spangpu_modem_shard_sync": (ci, [vp]),
            "spangpu_echo_feed_create": (ci, [C.POINTER(vp), vp, ci, ci, ci, ci]),
            "spangpu_echo_feed_destroy": (ci, [vp]),
            "spangpu_echo_feed_stride": (ll, [vp]),
            "spangpu_echo_feed_acquire": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_echo_feed_commit": (ci, [vp, ci]),
            "spangpu_echo_feed_collect": (ci, [vp, C.POINTER(vp)]),
            "spangpu_echo_feed_outstanding": (ci, [vp]),
            "spangpu_echo_feed_run": (ci, [vp, ci, ci, ci, C.POINTER(C.c_double)]),
            "spangpu_modem_packed_words": (ci, [ci, ci]),
            "spangpu_modem_pack_events": (ci, [vp, vp, ci, vp, ci]),
            "spangpu_modem_events_packed": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_modem_unpack_events": (ci, [vp, ci, vp, ci, ci, vp, ci, vp]),
            "spangpu_modem_feed_create": (ci, [C.POINTER(vp), vp, ci, ci, ci]),
            "spangpu_modem_feed_destroy": (ci, [vp]),
            "spangpu_modem_feed_stride": (ll, [vp]),
            "spangpu_modem_feed_words_per_channel": (ci, [vp]),
            "spangpu_modem_feed_status_cap": (ci, [vp]),
            "spangpu_modem_feed_acquire": (vp, [vp]),
            "spangpu_modem_feed_commit": (ci, [vp, ci]),
            "spangpu_modem_feed_collect": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_modem_feed_outstanding": (ci, [vp]),
            "spangpu_modem_feed_run": (ci, [vp, ci, ci, ci, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
            "spangpu_tune_modem_mapping": (ci, [ci]),
            "spangpu_tune_fsk_waves": (ci, [ci]),
            "spangpu_echo_lanes_per_channel": (ci, [vp]),
            "spangpu_echo_stats": (ci, [vp, ci]),
            "spangpu_echo_stats_reset": (ci, [vp, ci]),
            "spangpu_echo_stats_get": (ci, [vp, ci, ci, vp]),
            "spangpu_echo_erle": (ci, [vp, vp, ci]),
            "spangpu_txbank_create": (ci, [C.POINTER(vp), ci, ci, ci]),
            "spangpu_txbank_destroy": (None, [vp]),
            "spangpu_txbank_channels": (ci, [vp]),
            "spangpu_txbank_set_stream": (ci, [vp, vp]),
            "spangpu_txbank_sync": (ci, [vp]),
            "spangpu_txbank_tone": (ci, [vp, ci, ci, vp]),
            "spangpu_txbank_set_level": (ci, [vp, ci, ci, ci, ci]),
            "spangpu_txbank_set_timing": (ci, [vp, ci, ci, ci, ci]),
            "spangpu_txbank_put": (ci, [vp, ci, ci, C.c_char_p, ci]),
            "spangpu_txbank_put_each": (ci, [vp, ci, ci, vp, ci, vp, vp]),
            "spangpu_txbank_tx": (ci, [vp, ci, vp, ll, ci, vp]),
            "spangpu_txbank_state_words": (ci, []),
            "spangpu_txbank_get_state": (ci, [vp, ci, vp]),
            "spangpu_modem_create": (ci, [C.POINTER(vp), ci, ci, ci, ci]),
            "spangpu_modem_destroy": (ci, [vp]),
            "spangpu_modem_channels": (ci, [vp]),
            "spangpu_modem_set_stream": (ci, [vp, vp]),
            "spangpu_modem_sync": (ci, [vp]),
            "spangpu_modem_rx": (ci, [vp, vp, ci, ci, ll]),
            "spangpu_modem_rx_var": (ci, [vp, vp, ci, vp, ci, ll]),
            "spangpu_modem_events": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_modem_copy_events": (ci, [vp, vp, C.c_size_t, ci]),
            "spangpu_modem_qam_tap": (ci, [vp, ci]),
            "spangpu_modem_qam_reports": (ci, [vp, C.POINTER(vp), C.POINTER(vp)]),
            "spangpu_modem_state_words": (ci, [ci, C.POINTER(ci), C.POINTER(ci)]),
            "spangpu_modem_get_state": (ci, [vp, ci, vp]),
            "spangpu_modem_restart": (ci, [vp, ci]),
            "spangpu_modem_set_signal_cutoff": (ci, [vp, ci, cf]), "spangpu_modem_set_signal_cutoffs": (ci, [vp, vp]),
            "spangpu_modem_table": (ci, [ci, vp, ci]),
            "spangpu_v17_rx_maps": (ci, [vp, vp]),
            "spangpu_echo_create": (ci, [C.POINTER(vp), ci, ci, ci, ci]),
            "spangpu_echo_destroy": (ci, [vp]),
            "spangpu_echo_channels": (ci, [vp]),
            "spangpu_echo_taps": (ci, [vp]),
            "spangpu_echo_set_stream": (ci, [vp, vp]),
            "spangpu_echo_sync": (ci, [vp]),
            "spangpu_echo_update": (ci, [vp, vp, vp, vp, ci, ci, ll, ci]),
            "spangpu_echo_adaption_mode": (ci, [vp, ci, ci]),
            "spangpu_echo_flush": (ci, [vp, ci]),
            "spangpu_echo_get_state": (ci, [vp, ci, vp, vp, vp, vp]),
            "spangpu_echo_set_state": (ci, [vp, ci, vp, vp, vp, vp]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise SpanGpuError(rc, lib().spangpu_last_error().decode("latin1"))
    return rc


def device_count():
    return lib().spangpu_device_count()


def probe_stream_read(device=0, nbytes=1 << 30, reps=10):
    """The streaming read rate (GB/s) a plain kernel reaches on this device: the practical ceiling beside the 8 TB/s peak."""
    v = C.c_double(0.0)
    _check(lib().spangpu_probe_stream_read(device, nbytes, reps, C.byref(v)))
    return v.value


def tune_lanes_per_channel(lpc):
    _check(lib().spangpu_tune_lanes_per_channel(lpc))


def tune_tone_kernel(variant):
    """0 = per-call choice, 1 = always the general kernel, 2 = streaming with loader waves, 3 = streaming without."""
    _check(lib().spangpu_tune_tone_kernel(variant))


def tune_modem_mapping(mapping):
    """0 = by bank size, 1 = one channel per lane, 4 = four lanes per channel (16 channels per wavefront)."""
    _check(lib().spangpu_tune_modem_mapping(mapping))


def tune_fsk_waves(waves):
    """FSK / connect-tone / signalling-tone receiver banks: 0 = the library's choice, 1 = a receiver in one lane of one
    wavefront, 2 = a receiver cut into two instruction streams on two wavefronts.  Results are identical."""
    _check(lib().spangpu_tune_fsk_waves(waves))


def goertzel_fac(freq):
    return lib().spangpu_goertzel_fac(freq)


class ToneBank:
    """N channels of one tone detector kind, state resident in HBM."""

    def __init__(self, kind, n_channels, device=0, report_mode=REPORT_DIGITS, filter_dialtone=False,
                 twist_db=0.0, reverse_twist_db=0.0, threshold_dbm0=0.0, r2_fwd=True,
                 bin_fac=None, block_len=0, trace=False, set_mask=0, functor=FUNCTOR_NONE, functor_threshold=0.0):
        p = ToneParams()
        p.set_mask = set_mask
        p.functor = functor
        p.functor_threshold = functor_threshold
        p.report_mode = report_mode
        p.filter_dialtone = int(filter_dialtone)
        p.twist_db = twist_db
        p.reverse_twist_db = reverse_twist_db
        p.threshold_dbm0 = threshold_dbm0
        p.r2_fwd = int(r2_fwd)
        if bin_fac is not None:
            p.n_bins = len(bin_fac)
            for i, f in enumerate(bin_fac):
                p.bin_fac[i] = f
        p.block_len = block_len
        p.trace = int(trace)
        self.kind = kind
        self.n = n_channels
        self.h = C.c_void_p()
        _check(lib().spangpu_bank_create(C.byref(self.h), device, kind, n_channels, C.byref(p), C.sizeof(p)))
        self.nbins = lib().spangpu_bank_bins(self.h)

    def close(self):
        if self.h:
            lib().spangpu_bank_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(lib().spangpu_bank_set_stream(self.h, hip_stream))

    def get_stream(self):
        """the hipStream_t the bank launches on, as an integer"""
        return lib().spangpu_bank_get_stream(self.h) or 0

    def set_queues(self, queues):
        """Queue mode (spangpu_bank_set_queues): 2 = the streaming kernel's launch cut in two on two hardware queues, 1 = one
        launch, 0 = the library's choice.  Returns the number in use."""
        rc = lib().spangpu_bank_set_queues(self.h, int(queues))
        if rc < 0:
            _check(rc)
        return rc

    def join(self):
        """Make the bank's stream wait for the second queue's last launch (before work of the caller's own on that stream)."""
        _check(lib().spangpu_bank_join(self.h))

    def share_stream(self, other):
        """Launch on the same HIP stream as `other` (needed for banks that share a launch)."""
        _check(lib().spangpu_bank_set_stream(self.h, lib().spangpu_bank_get_stream(other.h)))

    def set_timing(self, on=True):
        _check(lib().spangpu_bank_set_timing(self.h, int(on)))

    def rx_host(self, frames, layout=CHANNEL_MAJOR):
        """frames: int16 [n_channels, samples] (channel-major) or [samples, n_channels]."""
        frames = np.ascontiguousarray(frames, dtype=np.int16)
        if layout == CHANNEL_MAJOR:
            assert frames.shape[0] == self.n
            samples, stride = frames.shape[1], frames.shape[1]
        else:
            assert frames.shape[1] == self.n
            samples, stride = frames.shape[0], frames.shape[1]
        _check(lib().spangpu_bank_rx(self.h, frames.ctypes.data, MEM_HOST, layout, samples, stride))
        # the H2D copy is queued asynchronously from pageable memory: keep the source alive until done
        self.sync()

    def rx_device(self, ptr, samples, stride=0, layout=CHANNEL_MAJOR):
        _check(lib().spangpu_bank_rx(self.h, ptr, MEM_DEVICE, layout, samples, stride))

    def rx_host_var(self, frames, lens):
        """A tick with per-channel frame lengths: frames int16 [n_channels, max_samples], lens[c] samples of row c count
        (0 = channel c sits the tick out, untouched)."""
        frames = np.ascontiguousarray(frames, dtype=np.int16)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        assert frames.shape[0] == self.n and lens.shape == (self.n,)
        _check(lib().spangpu_bank_rx_var(self.h, frames.ctypes.data, MEM_HOST, lens.ctypes.data, frames.shape[1], frames.shape[1]))
        self.sync()

    def rx_device_var(self, ptr, lens, max_samples, stride):
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        _check(lib().spangpu_bank_rx_var(self.h, ptr, MEM_DEVICE, lens.ctypes.data, max_samples, stride))

    def set_channel_params(self, channel, filter_dialtone=-1, twist_db=-1.0, reverse_twist_db=-1.0, threshold_dbm0=-99.0):
        """dtmf_rx_parms() for one channel of a DTMF bank, with its argument conventions (negative = leave alone)."""
        p = ToneParams()
        p.filter_dialtone = int(filter_dialtone)
        p.twist_db, p.reverse_twist_db, p.threshold_dbm0 = twist_db, reverse_twist_db, threshold_dbm0
        p.set_mask = TP_TWIST | TP_REVERSE_TWIST | TP_THRESHOLD
        _check(lib().spangpu_bank_set_channel_params(self.h, channel, C.byref(p), C.sizeof(p)))

    def rx_host_g711(self, codes, law):
        """codes: uint8 [n_channels, samples] of A-law (law = G711_ALAW) or u-law (G711_ULAW) bytes."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        assert codes.shape[0] == self.n
        _check(lib().spangpu_bank_rx_g711(self.h, codes.ctypes.data, MEM_HOST, law, codes.shape[1], codes.shape[1]))

    def rx_device_g711(self, ptr, law, samples, stride=0):
        _check(lib().spangpu_bank_rx_g711(self.h, ptr, MEM_DEVICE, law, samples, stride))

    def sync(self):
        _check(lib().spangpu_bank_sync(self.h))

    def blocks(self):
        n = _check(lib().spangpu_bank_blocks(self.h, None, 0))
        out = np.zeros(n, BLOCK_DTYPE)
        if n:
            _check(lib().spangpu_bank_blocks(self.h, out.ctypes.data, n))
        return out

    def set_records_buffer(self, dev_ptr, nbytes):
        """Later launches write their block records straight into this device buffer (None: the bank's own)."""
        _check(lib().spangpu_bank_set_records_buffer(self.h, dev_ptr, nbytes))

    def copy_records(self, dst_ptr, dst_bytes):
        return _check(lib().spangpu_bank_copy_records(self.h, dst_ptr, dst_bytes))

    def set_digits_buffer(self, dev_ptr, nbytes):
        """From now on the detector kernel also writes digits[block][channel] bytes (0 = none) at dev_ptr; None: off."""
        _check(lib().spangpu_bank_set_digits_buffer(self.h, dev_ptr, nbytes))

    def set_digits_ring(self, dev_ptr, slice_bytes, n_slices):
        """... successive launches fill successive slices of dev_ptr (n_slices of slice_bytes), round and round."""
        _check(lib().spangpu_bank_set_digits_ring(self.h, dev_ptr, slice_bytes, n_slices))

    def digit_events_device(self, dst_ptr, cap_entries):
        """The digits of the last launch as a compact list at dst_ptr (1 + cap_entries uint32 words of device memory)."""
        _check(lib().spangpu_bank_digit_events(self.h, dst_ptr, cap_entries))

    # ---- super-tone cadences matched on the device ----
    def set_cadences(self, tones, want_segments=False):
        """tones = [[(f1_bin, f2_bin, min_ms, max_ms), ...], ...] as super_tone_rx_add_element() resolved them."""
        counts = np.array([len(t) for t in tones], np.int32)
        el = np.array([e for t in tones for e in t], np.int32).reshape(-1, 4)
        _check(lib().spangpu_bank_set_cadences(self.h, counts.ctypes.data, len(tones), el.ctypes.data if len(el) else None,
                                               int(want_segments)))

    def cadence_run(self):
        return _check(lib().spangpu_bank_cadence_run(self.h))

    def cadence_events(self):
        """Per channel, the callbacks super_tone_rx() would have made over the last launch, in the oracle's event format:
        (1, tone, -10, 0) for tone reports (tone -1 = lost), (4, f1, f2, ms) for segments."""
        ev = C.c_void_p()
        cnt = C.c_void_p()
        slots = _check(lib().spangpu_bank_cadence_events(self.h, C.byref(ev), C.byref(cnt)))
        counts = np.ctypeslib.as_array(C.cast(cnt, C.POINTER(C.c_int32)), (self.n,))
        out = [[] for _ in range(self.n)]
        if slots == 0 or not counts.any():
            return out
        words = np.ctypeslib.as_array(C.cast(ev, C.POINTER(C.c_uint32)), (slots, self.n, 2))
        for c in np.nonzero(counts)[0]:
            for k in range(int(counts[c])):
                w0 = int(words[k, c, 0])
                w1 = int(np.int32(words[k, c, 1]))
                kind = w0 & 0xFF
                if kind == CADENCE_TONE_ON:
                    out[c].append((1, w1, -10, 0))
                elif kind == CADENCE_TONE_OFF:
                    out[c].append((1, -1, -10, 0))
                else:
                    out[c].append((4, ((w0 >> 8) & 0xFF) - 1, ((w0 >> 16) & 0xFF) - 1, w1))
        return out

    def cadence_list(self):
        """The events of the last launch as an [n, 3] array of (channel, word 0, word 1), a channel's events together and in
        order (a view of the library's pinned buffer: good until the next call)."""
        p = C.c_void_p()
        n = _check(lib().spangpu_bank_cadence_list(self.h, C.byref(p)))
        if n == 0:
            return np.zeros((0, 3), np.uint32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), (n, 3))

    def cadence_reset(self, channel=-1):
        _check(lib().spangpu_bank_cadence_reset(self.h, channel))

    def cadence_get_state(self, channel):
        w = np.zeros(lib().spangpu_bank_cadence_state_words(), np.int32)
        _check(lib().spangpu_bank_cadence_get_state(self.h, channel, w.ctypes.data))
        return w

    def cadence_set_state(self, channel, words):
        w = np.ascontiguousarray(words, np.int32)
        _check(lib().spangpu_bank_cadence_set_state(self.h, channel, w.ctypes.data))

    def trace(self, max_blocks=8):
        buf = np.zeros(max_blocks*(self.nbins + 1)*self.n, np.float32)
        nb = _check(lib().spangpu_bank_trace(self.h, buf.ctypes.data, buf.size))
        return buf[:nb*(self.nbins + 1)*self.n].reshape(nb, self.nbins + 1, self.n)

    def get_state(self, channel):
        f = np.zeros(2*MAX_BINS + 8, np.float32)
        i = np.zeros(4, np.int32)
        nsf = _check(lib().spangpu_bank_get_state(self.h, channel, f.ctypes.data, len(f), i.ctypes.data, 4))
        return f[:nsf].copy(), i

    def set_state(self, channel, f, i):
        f = np.ascontiguousarray(f, np.float32)
        i = np.ascontiguousarray(i, np.int32)
        _check(lib().spangpu_bank_set_state(self.h, channel, f.ctypes.data, len(f), i.ctypes.data, len(i)))

    def reset_channel(self, channel, fillin_only=False):
        _check(lib().spangpu_bank_reset_channel(self.h, channel, int(fillin_only)))

    def last_kernel_ms(self):
        return lib().spangpu_bank_last_kernel_ms(self.h)


ECHO_SCALARS = 48
ECHO_FIELDS = ["tx_power0", "tx_power1", "tx_power2", "tx_power3", "rx_power0", "rx_power1", "rx_power2",
               "clean_rx_power", "rx_power_threshold", "nonupdate_dwell", "curr_pos", "taps", "tap_mask",
               "adaption_mode", "supp_test1", "supp_test2", "supp1", "supp2", "vad", "cng", "geigel_max",
               "geigel_lag", "dtd_onset", "tap_set", "tap_rotate_counter", "latest_correction",
               "narrowband_count", "narrowband_score", "fir_curr_pos", "tx_hpf0", "tx_hpf1", "rx_hpf0",
               "rx_hpf1", "cng_level", "cng_rndnum", "cng_filter", "fir_set"]


def banks_own_queues(banks):
    """A stream of its own for every bank of a tick, on hardware queues that are certainly different (spangpu_banks_own_queues)."""
    hb = (C.c_void_p*len(banks))(*[b.h for b in banks])
    return _check(lib().spangpu_banks_own_queues(hb, len(banks)))


def banks_rx_device(banks, ptrs, samples, strides=None):
    """Advance several ToneBanks (same stream) with one kernel launch; ptrs = device addresses of their frames."""
    n = len(banks)
    hb = (C.c_void_p*n)(*[b.h for b in banks])
    pa = (C.c_void_p*n)(*[C.c_void_p(int(p)) for p in ptrs])
    st = None
    if strides is not None:
        st = (C.c_longlong*n)(*strides)
    _check(lib().spangpu_banks_rx(hb, pa, n, samples, st))


class BanksPlan:
    """banks_rx_device() with the argument arrays built once: plan = BanksPlan(banks); plan.frame(ptrs) -> a handle
    for one set of frame addresses; plan.rx(handle, samples) queues the launch (no Python-side allocation per tick)."""

    def __init__(self, banks, strides=None):
        self.n = len(banks)
        self.hb = (C.c_void_p*self.n)(*[b.h for b in banks])
        self.st = (C.c_longlong*self.n)(*strides) if strides is not None else None
        self.fn = lib().spangpu_banks_rx

    def frame(self, ptrs):
        return (C.c_void_p*self.n)(*[C.c_void_p(int(p)) for p in ptrs])

    def rx(self, frame, samples):
        rc = self.fn(self.hb, frame, self.n, samples, self.st)
        if rc < 0:
            _check(rc)


ECHO_STATS_DTYPE = np.dtype([("sum_rx2", "<u8"), ("sum_clean2", "<u8"), ("crc", "<u4"), ("samples", "<u4")])


class Feed:
    """The pipelined host-buffer path of a tone bank (spangpu_feed_*): `depth` pinned slots; slot() hands out the numpy view
    of the next tick's staging buffer ([n_ch, stride] int16, or uint8 for G.711) for the caller to fill, commit() queues the
    tick, collect() returns the digits of the oldest tick as (channel, digit, block) arrays."""

    def __init__(self, bank, max_samples, law=0, depth=2, device=0):
        self.bank = bank
        self.law = law
        self.h = C.c_void_p()
        _check(lib().spangpu_feed_create(C.byref(self.h), bank.h, device, max_samples, law, depth))
        self.stride = int(lib().spangpu_feed_stride(self.h))
        self.n_ch = bank.n
        self.depth = depth

    def close(self):
        if self.h:
            lib().spangpu_feed_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def slot(self):
        p = lib().spangpu_feed_acquire(self.h)
        if not p:
            raise SpanGpuError(-6, lib().spangpu_last_error().decode())
        if self.law:
            buf = (C.c_uint8*(self.n_ch*self.stride)).from_address(p)
            return np.frombuffer(buf, np.uint8).reshape(self.n_ch, self.stride)
        buf = (C.c_int16*(self.n_ch*self.stride)).from_address(p)
        return np.frombuffer(buf, np.int16).reshape(self.n_ch, self.stride)

    def commit(self, samples):
        _check(lib().spangpu_feed_commit(self.h, samples))

    def run(self, samples, ticks, lag=1):
        """The tick loop in C over the frames the slots hold -> (milliseconds, digits collected)."""
        ms = C.c_double()
        digits = C.c_longlong()
        _check(lib().spangpu_feed_run(self.h, samples, ticks, lag, C.byref(ms), C.byref(digits)))
        return ms.value, digits.value

    def collect(self):
        """-> (channel, digit, block) uint32 arrays of the oldest outstanding tick, or None if there is none."""
        if lib().spangpu_feed_outstanding(self.h) <= 0:
            return None
        ent = C.c_void_p()
        n = _check(lib().spangpu_feed_collect(self.h, C.byref(ent)))
        if n == 0:
            z = np.zeros(0, np.uint32)
            return z, z, z
        w = np.frombuffer((C.c_uint32*n).from_address(ent.value), np.uint32).copy()
        return w & 0xFFFFF, (w >> 20) & 0xFF, (w >> 28) & 0xF


# ---- the receivers' inner primitives, batched (csrc/prim_api.hip); host arrays in, host arrays out ----------------------------
def vec_circular_dot_prodf(x, y, pos, device=0):
    """x, y: float32 [items, n] (or [n] shared by all items); pos int32 [items] -> float32 [items]."""
    pos = np.ascontiguousarray(pos, np.int32)
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    n = x.shape[-1]
    z = np.zeros(len(pos), np.float32)
    _check(lib().spangpu_vec_circular_dot_prodf_batch(device, x.ctypes.data, n if x.ndim == 2 else 0, y.ctypes.data, n if y.ndim == 2 else 0,
                                                      pos.ctypes.data, z.ctypes.data, len(pos), n, MEM_HOST))
    return z


def vec_circular_lmsf(x, y, pos, error, device=0):
    """y (float32 [items, n]) after y = y*0.9999f + x*error in circular order; x [items, n] or [n]."""
    pos = np.ascontiguousarray(pos, np.int32)
    x = np.ascontiguousarray(x, np.float32)
    y = np.array(y, np.float32, order="C")
    e = np.ascontiguousarray(error, np.float32)
    n = y.shape[1]
    _check(lib().spangpu_vec_circular_lmsf_batch(device, x.ctypes.data, n if x.ndim == 2 else 0, y.ctypes.data, n, pos.ctypes.data, e.ctypes.data,
                                                 len(pos), n, MEM_HOST))
    return y


def cvec_circular_dot_prodf(x, y, pos, device=0):
    """x, y: complex64 [items, n] (or [n]); -> complex64 [items]."""
    pos = np.ascontiguousarray(pos, np.int32)
    x = np.ascontiguousarray(x, np.complex64)
    y = np.ascontiguousarray(y, np.complex64)
    n = x.shape[-1]
    z = np.zeros(len(pos), np.complex64)
    _check(lib().spangpu_cvec_circular_dot_prodf_batch(device, x.ctypes.data, n if x.ndim == 2 else 0, y.ctypes.data, n if y.ndim == 2 else 0,
                                                       pos.ctypes.data, z.ctypes.data, len(pos), n, MEM_HOST))
    return z


def cvec_circular_lmsf(x, y, pos, error, device=0):
    pos = np.ascontiguousarray(pos, np.int32)
    x = np.ascontiguousarray(x, np.complex64)
    y = np.array(y, np.complex64, order="C")
    e = np.ascontiguousarray(error, np.complex64)
    n = y.shape[1]
    _check(lib().spangpu_cvec_circular_lmsf_batch(device, x.ctypes.data, n if x.ndim == 2 else 0, y.ctypes.data, n, pos.ctypes.data, e.ctypes.data,
                                                  len(pos), n, MEM_HOST))
    return y


def power_meter_update(amp, reading, shift, device=0):
    """amp int16 [items, n]; reading, shift int32 [items] -> the readings after the rows."""
    amp = np.ascontiguousarray(amp, np.int16)
    r = np.array(reading, np.int32)
    sh = np.ascontiguousarray(shift, np.int32)
    _check(lib().spangpu_power_meter_update_batch(device, amp.ctypes.data, amp.shape[1], r.ctypes.data, sh.ctypes.data, amp.shape[0], amp.shape[1], MEM_HOST))
    return r


# ---- ... and the rest of them (csrc/prim2_api.hip): periodograms, fixed_sqrt32, dds_complexf, arctan2 ------------------------------
def periodogram(coeffs, amp, device=0):
    """coeffs float32 [items, len/2, 2] (or [len/2, 2] for all items), amp float32 [items, len, 2] -> float32 [items, 2]"""
    c = np.ascontiguousarray(coeffs, np.float32)
    a = np.ascontiguousarray(amp, np.float32)
    items, n = a.shape[0], a.shape[1]
    out = np.zeros((items, 2), np.float32)
    _check(lib().spangpu_periodogram_batch(device, c.ctypes.data, n//2 if c.ndim == 3 else 0, a.ctypes.data, n, out.ctypes.data, items, n, MEM_HOST))
    return out


def periodogram_prepare(amp, device=0):
    a = np.ascontiguousarray(amp, np.float32)
    items, n = a.shape[0], a.shape[1]
    s = np.zeros((items, n//2, 2), np.float32)
    d = np.zeros((items, n//2, 2), np.float32)
    assert _check(lib().spangpu_periodogram_prepare_batch(device, a.ctypes.data, n, s.ctypes.data, d.ctypes.data, items, n, MEM_HOST)) == n//2
    return s, d


def periodogram_apply(coeffs, s, d, n, device=0):
    c = np.ascontiguousarray(coeffs, np.float32)
    s = np.ascontiguousarray(s, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    out = np.zeros((s.shape[0], 2), np.float32)
    _check(lib().spangpu_periodogram_apply_batch(device, c.ctypes.data, n//2 if c.ndim == 3 else 0, s.ctypes.data, d.ctypes.data, out.ctypes.data,
                                                 s.shape[0], n, MEM_HOST))
    return out


def periodogram_freq_error(phase_offset, scale, last, now, device=0):
    off = np.ascontiguousarray(phase_offset, np.float32)
    a = np.ascontiguousarray(last, np.float32)
    b = np.ascontiguousarray(now, np.float32)
    out = np.zeros(a.shape[0], np.float32)
    _check(lib().spangpu_periodogram_freq_error_batch(device, off.ctypes.data, scale, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.shape[0], MEM_HOST))
    return out


def fixed_sqrt32(x, device=0):
    x = np.ascontiguousarray(x, np.uint32)
    out = np.zeros(len(x), np.uint16)
    _check(lib().spangpu_fixed_sqrt32_batch(device, x.ctypes.data, out.ctypes.data, len(x), MEM_HOST))
    return out


def dds_complexf(phase_acc, phase_rate, n, device=0):
    """-> (float32 [items, n, 2] phasors, uint32 [items] accumulators after n steps)"""
    acc = np.array(phase_acc, np.uint32)
    rate = np.ascontiguousarray(phase_rate, np.int32)
    out = np.zeros((len(acc), n, 2), np.float32)
    _check(lib().spangpu_dds_complexf_batch(device, acc.ctypes.data, rate.ctypes.data, out.ctypes.data, len(acc), n, MEM_HOST))
    return out, acc


def arctan2(y, x, device=0):
    y = np.ascontiguousarray(y, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(len(y), np.int32)
    _check(lib().spangpu_arctan2_batch(device, y.ctypes.data, x.ctypes.data, out.ctypes.data, len(y), MEM_HOST))
    return out


class ShardedToneBank:
    """One logical DTMF / Bell MF / R2 MF bank over several devices (spangpu_shard_*): contiguous channel ranges, a bank and
    a stream per shard, the digit byte of every block and channel gathered device to device to the first shard's device."""

    def __init__(self, kind, n_channels, devices, max_samples=160):
        self.n = n_channels
        self.max_blocks = max_samples//102 + 2          # (shard_api.hip: the shortest block of the kinds a shard takes)
        self.h = C.c_void_p()
        dv = (C.c_int*len(devices))(*devices)
        _check(lib().spangpu_shard_create(C.byref(self.h), dv, len(devices), kind, n_channels, max_samples, None, 0))
        self.shards = _check(lib().spangpu_shard_count(self.h))
        self.ranges = []
        for i in range(self.shards):
            d, f, n = C.c_int(), C.c_int(), C.c_int()
            _check(lib().spangpu_shard_range(self.h, i, C.byref(d), C.byref(f), C.byref(n)))
            self.ranges.append((d.value, f.value, n.value))

    def close(self):
        if self.h:
            lib().spangpu_shard_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self, i):
        """where shard i sits and how its bytes reach the collecting device (LINK_SAME / LINK_PEER / LINK_STAGED)"""
        return shard_info("spangpu_shard", self.h, i)

    def rx_device(self, ptrs, samples, stride):
        """ptrs[i]: device address of shard i's rows (on shard i's device); queues the step, returns blocks per channel."""
        arr = (C.c_void_p*len(ptrs))(*ptrs)
        return _check(lib().spangpu_shard_rx(self.h, arr, samples, stride))

    def digits_host(self):
        """[blocks, n_channels] uint8 of the last step, in the whole bank's channel order."""
        out = np.zeros((self.max_blocks, self.n), np.uint8)
        nb = _check(lib().spangpu_shard_digits_host(self.h, out.ctypes.data, out.nbytes))
        return out[:nb].copy()

    def sync(self):
        _check(lib().spangpu_shard_sync(self.h))


LINK_SAME, LINK_PEER, LINK_STAGED = 0, 1, 2


class ShardInfo(C.Structure):
    _fields_ = [("device", C.c_int), ("first_channel", C.c_int), ("n_channels", C.c_int), ("collect_device", C.c_int), ("link", C.c_int),
                ("forced_peer_copy", C.c_int)]


def tune_force_peer_copy(on):
    """Debug knob: shards on the collecting device send their results with hipMemcpyPeerAsync too.  Returns the previous setting."""
    return lib().spangpu_tune_force_peer_copy(int(on))


def shard_info(prefix, handle, i):
    info = ShardInfo()
    _check(getattr(lib(), prefix + "_info")(handle, i, C.byref(info)))
    return info


class _Sharded:
    """Common to the sharded echo and modem objects: the ranges, close()."""
    _prefix = ""

    def info(self, i):
        """where shard i sits and how its results reach the collecting device (LINK_SAME / LINK_PEER / LINK_STAGED)"""
        return shard_info(self._prefix, self.h, i)

    def _ranges(self):
        self.ranges = []
        i = 0
        while True:
            d, f, n = C.c_int(), C.c_int(), C.c_int()
            if getattr(lib(), self._prefix + "_range")(self.h, i, C.byref(d), C.byref(f), C.byref(n)) < 0:
                break
            self.ranges.append((d.value, f.value, n.value))
            i += 1
        self.shards = len(self.ranges)

    def close(self):
        if self.h:
            getattr(lib(), self._prefix + "_destroy")(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(getattr(lib(), self._prefix + "_sync")(self.h))


class ShardedEchoBank(_Sharded):
    """The echo cancellers of n_channels lines over several devices (spangpu_echo_shard_*, BASELINE configs[4]'s object):
    update_device() queues a step (rows per shard on its device), report() has every shard compute its lines' ERLE and send
    it to the first shard's device, erle_host() returns the last report in the whole bank's channel order."""
    _prefix = "spangpu_echo_shard"

    def __init__(self, n_channels, taps, mode, devices):
        self.n = n_channels
        self.h = C.c_void_p()
        dv = (C.c_int*len(devices))(*devices)
        _check(lib().spangpu_echo_shard_create(C.byref(self.h), dv, len(devices), n_channels, taps, mode))
        self._ranges()

    def update_device(self, tx_ptrs, rx_ptrs, clean_ptrs, samples, stride):
        a = (C.c_void_p*len(tx_ptrs))(*tx_ptrs)
        b = (C.c_void_p*len(rx_ptrs))(*rx_ptrs)
        c = (C.c_void_p*len(clean_ptrs))(*clean_ptrs)
        _check(lib().spangpu_echo_shard_update(self.h, a, b, c, samples, stride))

    def report(self, reset=True):
        _check(lib().spangpu_echo_shard_report(self.h, int(reset)))

    def erle_host(self):
        out = np.zeros(self.n, np.float32)
        _check(lib().spangpu_echo_shard_erle_host(self.h, out.ctypes.data, out.size))
        return out


class ShardedModemBank(_Sharded):
    """Modem receivers over several devices (spangpu_modem_shard_*): rx_device() queues a step, events_host() returns the
    step's put_bit streams as (counts [n_channels] int32, events [n_channels, per] int8) in the whole bank's channel order."""
    _prefix = "spangpu_modem_shard"

    def __init__(self, kind, n_channels, bit_rate, devices, per=64):
        self.n = n_channels
        self.per = per
        self.h = C.c_void_p()
        dv = (C.c_int*len(devices))(*devices)
        _check(lib().spangpu_modem_shard_create(C.byref(self.h), dv, len(devices), kind, n_channels, bit_rate, per))
        self._ranges()

    def rx_device(self, ptrs, samples, stride):
        arr = (C.c_void_p*len(ptrs))(*ptrs)
        _check(lib().spangpu_modem_shard_rx(self.h, arr, samples, stride))

    def events_host(self):
        counts = np.zeros(self.n, np.int32)
        ev = np.zeros((self.n, self.per), np.int8)
        _check(lib().spangpu_modem_shard_events_host(self.h, counts.ctypes.data, ev.ctypes.data))
        return counts, ev


class EchoFeed:
    """The pipelined host path of an echo canceller bank (spangpu_echo_feed_*): slots() hands out the numpy views of the next
    tick's tx and rx staging rows ([n_ch, stride] int16, or uint8 with a G.711 law), commit() queues the tick (H2D, kernel and
    D2H on three streams), collect() returns the clean rows of the oldest tick ([n_ch, stride], good until that slot is
    committed again)."""

    def __init__(self, bank, max_samples, law=0, depth=3, use_hpf_tx=False):
        self.bank = bank
        self.law = law
        self.h = C.c_void_p()
        _check(lib().spangpu_echo_feed_create(C.byref(self.h), bank.h, max_samples, law, depth, int(use_hpf_tx)))
        self.stride = int(lib().spangpu_echo_feed_stride(self.h))
        self.n_ch = bank.n
        self.depth = depth

    def close(self):
        if self.h:
            lib().spangpu_echo_feed_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _view(self, p):
        if self.law:
            return np.frombuffer((C.c_uint8*(self.n_ch*self.stride)).from_address(p), np.uint8).reshape(self.n_ch, self.stride)
        return np.frombuffer((C.c_int16*(self.n_ch*self.stride)).from_address(p), np.int16).reshape(self.n_ch, self.stride)

    def slots(self):
        tx, rx = C.c_void_p(), C.c_void_p()
        _check(lib().spangpu_echo_feed_acquire(self.h, C.byref(tx), C.byref(rx)))
        return self._view(tx.value), self._view(rx.value)

    def commit(self, samples):
        _check(lib().spangpu_echo_feed_commit(self.h, samples))

    def outstanding(self):
        return _check(lib().spangpu_echo_feed_outstanding(self.h))

    def collect(self):
        """-> (clean rows view, samples) of the oldest outstanding tick, or None."""
        if self.outstanding() <= 0:
            return None
        p = C.c_void_p()
        n = _check(lib().spangpu_echo_feed_collect(self.h, C.byref(p)))
        return self._view(p.value), n

    def run(self, samples, ticks, lag=1):
        ms = C.c_double()
        _check(lib().spangpu_echo_feed_run(self.h, samples, ticks, lag, C.byref(ms)))
        return ms.value


class ModemFeed:
    """The pipelined host path of a modem receiver bank (spangpu_modem_feed_*): slot() = the next tick's PCM staging rows
    ([n_ch, stride] int16), commit() queues the tick, collect() returns the oldest tick's put_bit stream -- as the packed rows
    and status list (raw=True) or unpacked into per-channel int8 arrays like ModemBank.events()."""

    def __init__(self, bank, max_samples, max_bit_rate, depth=3):
        self.bank = bank
        self.h = C.c_void_p()
        _check(lib().spangpu_modem_feed_create(C.byref(self.h), bank.h, max_samples, max_bit_rate, depth))
        self.stride = int(lib().spangpu_modem_feed_stride(self.h))
        self.wpc = _check(lib().spangpu_modem_feed_words_per_channel(self.h))
        self.status_cap = _check(lib().spangpu_modem_feed_status_cap(self.h))
        self.n_ch = bank.n
        self.depth = depth
        self.max_samples = max_samples

    def close(self):
        if self.h:
            lib().spangpu_modem_feed_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def slot(self):
        p = lib().spangpu_modem_feed_acquire(self.h)
        if not p:
            raise SpanGpuError(-6, lib().spangpu_last_error().decode())
        return np.frombuffer((C.c_int16*(self.n_ch*self.stride)).from_address(p), np.int16).reshape(self.n_ch, self.stride)

    def commit(self, samples):
        _check(lib().spangpu_modem_feed_commit(self.h, samples))

    def outstanding(self):
        return _check(lib().spangpu_modem_feed_outstanding(self.h))

    def collect(self, raw=False):
        if self.outstanding() <= 0:
            return None
        pk, st = C.c_void_p(), C.c_void_p()
        _check(lib().spangpu_modem_feed_collect(self.h, C.byref(pk), C.byref(st)))
        packed = np.frombuffer((C.c_uint32*(self.n_ch*self.wpc)).from_address(pk.value), np.uint32).reshape(self.n_ch, self.wpc)
        status = np.frombuffer((C.c_uint32*(1 + 2*self.status_cap)).from_address(st.value), np.uint32)
        if raw:
            return packed, status
        return unpack_modem_events(packed, status, self.status_cap, self.max_samples*4 + 64)

    def run(self, samples, ticks, lag=1):
        ms = C.c_double()
        bits = C.c_longlong()
        _check(lib().spangpu_modem_feed_run(self.h, samples, ticks, lag, C.byref(ms), C.byref(bits)))
        return ms.value, bits.value


def unpack_modem_events(packed, status, status_cap, cap):
    """spangpu_modem_unpack_events(): the packed put_bit stream of a tick -> a list of per-channel int8 arrays."""
    packed = np.ascontiguousarray(packed, np.uint32)
    status = np.ascontiguousarray(status, np.uint32)
    n_ch, wpc = packed.shape
    ev = np.zeros((n_ch, cap), np.int8)
    cnt = np.zeros(n_ch, np.int32)
    _check(lib().spangpu_modem_unpack_events(packed.ctypes.data, wpc, status.ctypes.data, status_cap, n_ch, ev.ctypes.data, cap, cnt.ctypes.data))
    return [ev[c, :cnt[c]].copy() for c in range(n_ch)]


class EchoBank:
    """N G.168 line echo cancellers, state resident in HBM."""

    def __init__(self, n_channels, taps, adaption_mode, device=0):
        self.n = n_channels
        self.taps = taps
        self.h = C.c_void_p()
        _check(lib().spangpu_echo_create(C.byref(self.h), device, n_channels, taps, adaption_mode))

    def close(self):
        if self.h:
            lib().spangpu_echo_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(lib().spangpu_echo_set_stream(self.h, hip_stream))

    def get_stream(self):
        """the hipStream_t the bank launches on, as an integer"""
        return lib().spangpu_echo_get_stream(self.h) or 0

    def sync(self):
        _check(lib().spangpu_echo_sync(self.h))

    def update_host(self, tx, rx, use_hpf_tx=False):
        tx = np.ascontiguousarray(tx, np.int16)
        rx = np.ascontiguousarray(rx, np.int16)
        assert tx.shape == rx.shape and tx.shape[0] == self.n
        clean = np.zeros_like(tx)
        _check(lib().spangpu_echo_update(self.h, tx.ctypes.data, rx.ctypes.data, clean.ctypes.data, MEM_HOST,
                                         tx.shape[1], tx.shape[1], int(use_hpf_tx)))
        return clean

    def update_device(self, tx_ptr, rx_ptr, clean_ptr, samples, stride, use_hpf_tx=False):
        _check(lib().spangpu_echo_update(self.h, tx_ptr, rx_ptr, clean_ptr, MEM_DEVICE, samples, stride, int(use_hpf_tx)))

    def adaption_mode(self, mode, channel=-1):
        _check(lib().spangpu_echo_adaption_mode(self.h, channel, mode))

    # per-channel line statistics: energy of rx and of the clean signal, CRC-32 of the clean stream
    def stats(self, enable=True):
        _check(lib().spangpu_echo_stats(self.h, int(enable)))

    def stats_reset(self, sums=True, crc=False):
        _check(lib().spangpu_echo_stats_reset(self.h, (1 if sums else 0) | (2 if crc else 0)))

    def stats_get(self, first=0, n=None):
        n = self.n - first if n is None else n
        out = np.zeros(n, ECHO_STATS_DTYPE)
        _check(lib().spangpu_echo_stats_get(self.h, first, n, out.ctypes.data))
        return out

    def erle_host(self):
        out = np.zeros(self.n, np.float32)
        _check(lib().spangpu_echo_erle(self.h, out.ctypes.data, MEM_HOST))
        return out

    def erle_device(self, dev_ptr):
        _check(lib().spangpu_echo_erle(self.h, dev_ptr, MEM_DEVICE))

    def flush(self, channel):
        _check(lib().spangpu_echo_flush(self.h, channel))

    def get_state(self, channel):
        s = np.zeros(ECHO_SCALARS, np.int32)
        t32 = np.zeros(self.taps, np.int32)
        t16 = np.zeros(4*self.taps, np.int16)
        h = np.zeros(self.taps, np.int16)
        _check(lib().spangpu_echo_get_state(self.h, channel, s.ctypes.data, t32.ctypes.data, t16.ctypes.data, h.ctypes.data))
        d = {k: int(v) for k, v in zip(ECHO_FIELDS, s)}
        d["last_acf"] = s[37:46].copy()
        d["taps32"] = t32
        d["taps16"] = t16.reshape(4, self.taps)
        d["history"] = h
        return d


V29 = 6
V27TER = 7
V17 = 8
G711_ALAW = 1
G711_ULAW = 2

_TABLES = {"sine": 0, "sqrt_tab": 1, "rrc_re": 10, "rrc_im": 11, "godard": 12, "v27_4800_re": 20, "v27_4800_im": 21,
           "v27_2400_re": 22, "v27_2400_im": 23, "v17_re": 30, "v17_im": 31, "v17_godard": 32, "v17_constellation": 33}


def modem_tables():
    """The constant modem tables as built by libspangpu (host code, no GPU needed)."""
    t = {}
    buf = np.zeros(192*27, np.float32)
    for name, which in _TABLES.items():
        n = _check(lib().spangpu_modem_table(which, buf.ctypes.data, len(buf)))
        t[name] = buf[:n].astype(np.uint16) if name == "sqrt_tab" else buf[:n].copy()
    return t


def v17_signal_space():
    """V.17 constellations (244 x {re, im}) and receiver soft-decision maps as built by libspangpu (host code)."""
    buf = np.zeros(488, np.float32)
    _check(lib().spangpu_modem_table(33, buf.ctypes.data, len(buf)))
    maps = np.zeros(4*36*36*8, np.uint8)
    m48 = np.zeros(36*36, np.uint8)
    _check(lib().spangpu_v17_rx_maps(maps.ctypes.data, m48.ctypes.data))
    return {"v17_constellation": buf, "v17_maps": maps, "v17_map_4800": m48}


class ModemBank:
    """N modem receivers of one kind and bit rate, state resident in HBM."""

    def __init__(self, kind, n_channels, bit_rate, device=0):
        self.n = n_channels
        self.kind = kind
        nf = C.c_int()
        ni = C.c_int()
        self.n_words = _check(lib().spangpu_modem_state_words(kind, C.byref(nf), C.byref(ni)))
        self.n_floats, self.n_ints = nf.value, ni.value
        self.h = C.c_void_p()
        _check(lib().spangpu_modem_create(C.byref(self.h), device, kind, n_channels, bit_rate))

    def close(self):
        if self.h:
            lib().spangpu_modem_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(lib().spangpu_modem_set_stream(self.h, hip_stream))

    def get_stream(self):
        """the hipStream_t the bank launches on, as an integer"""
        return lib().spangpu_modem_get_stream(self.h) or 0

    def sync(self):
        _check(lib().spangpu_modem_sync(self.h))

    def rx_host(self, frames):
        frames = np.ascontiguousarray(frames, np.int16)
        assert frames.shape[0] == self.n
        _check(lib().spangpu_modem_rx(self.h, frames.ctypes.data, MEM_HOST, frames.shape[1], frames.shape[1]))

    def rx_device(self, ptr, samples, stride):
        _check(lib().spangpu_modem_rx(self.h, ptr, MEM_DEVICE, samples, stride))

    def rx_host_var(self, frames, lens):
        """A tick with per-channel frame lengths (0 = the receiver sits it out, untouched)."""
        frames = np.ascontiguousarray(frames, np.int16)
        lens = np.ascontiguousarray(lens, np.int32)
        assert frames.shape[0] == self.n and lens.shape == (self.n,)
        _check(lib().spangpu_modem_rx_var(self.h, frames.ctypes.data, MEM_HOST, lens.ctypes.data, frames.shape[1], frames.shape[1]))

    def events(self, packed=False):
        """List (per channel) of int8 arrays: 0/1 bits and negative SIG_STATUS codes, in order.  packed: by way of the
        packed form (spangpu_modem_events_packed: the same answer, a fraction of the bytes over PCIe)."""
        ev = C.c_void_p()
        cnt = C.c_void_p()
        fn = lib().spangpu_modem_events_packed if packed else lib().spangpu_modem_events
        cap = _check(fn(self.h, C.byref(ev), C.byref(cnt)))
        counts = np.frombuffer((C.c_char*(4*self.n)).from_address(cnt.value), dtype=np.int32).copy()
        raw = np.frombuffer((C.c_char*(cap*self.n)).from_address(ev.value), dtype=np.int8).reshape(self.n, cap)
        assert counts.max(initial=0) <= cap, "event buffer overflow"
        return [raw[c, :counts[c]].copy() for c in range(self.n)]

    def copy_events(self, dst_ptr, nbytes, per_channel):
        """The last call's events, device to device: int32 counts[n_ch] then int8 events[n_ch][per_channel]."""
        _check(lib().spangpu_modem_copy_events(self.h, C.c_void_p(dst_ptr), nbytes, per_channel))

    def qam_tap(self, on=True):
        """Record the qam_report_handler_t calls of the following rx calls (xxx_rx_set_qam_report_handler)."""
        _check(lib().spangpu_modem_qam_tap(self.h, int(on)))

    def qam_reports(self):
        """List (per channel) of uint32 [n, 7]: events before the report, NULL flag, symbol, constel re / im, target re / im."""
        rec = C.c_void_p()
        cnt = C.c_void_p()
        cap = _check(lib().spangpu_modem_qam_reports(self.h, C.byref(rec), C.byref(cnt)))
        counts = np.frombuffer((C.c_char*(4*self.n)).from_address(cnt.value), dtype=np.int32).copy()
        raw = np.frombuffer((C.c_char*(28*cap*self.n)).from_address(rec.value), dtype=np.uint32).reshape(self.n, cap, 7)
        assert counts.max(initial=0) <= cap, "report buffer overflow"
        return [raw[c, :counts[c]].copy() for c in range(self.n)]

    def get_state(self, channel):
        w = np.zeros(self.n_words, np.uint32)
        _check(lib().spangpu_modem_get_state(self.h, channel, w.ctypes.data))
        return w[:self.n_floats].view(np.float32).copy(), w[self.n_floats:].view(np.int32).copy()

    def restart(self, channel):
        _check(lib().spangpu_modem_restart(self.h, channel))

    def set_signal_cutoff(self, channel, cutoff_dbm0):
        """xxx_rx_set_signal_cutoff(); channel -1 = every channel of the bank"""
        _check(lib().spangpu_modem_set_signal_cutoff(self.h, channel, cutoff_dbm0))

    def set_signal_cutoffs(self, cutoffs_dbm0):
        c = np.ascontiguousarray(cutoffs_dbm0, np.float32)
        assert len(c) == self.n
        _check(lib().spangpu_modem_set_signal_cutoffs(self.h, c.ctypes.data))


class V29Bank(ModemBank):
    def __init__(self, n_channels, bit_rate=9600, device=0):
        super().__init__(V29, n_channels, bit_rate, device)


class V27terBank(ModemBank):
    def __init__(self, n_channels, bit_rate=4800, device=0):
        super().__init__(V27TER, n_channels, bit_rate, device)


class V17Bank(ModemBank):
    def __init__(self, n_channels, bit_rate=14400, device=0):
        super().__init__(V17, n_channels, bit_rate, device)


# ---- signal source banks (include/spangpu.h "signal source banks") --------------------
TX_TONE_GEN, TX_DTMF, TX_BELL_MF, TX_R2_MF_FWD, TX_R2_MF_BACK = 1, 2, 3, 4, 5


class ToneDesc(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("f1", "l1", "f2", "l2", "d1", "d2", "d3", "d4", "repeat")]


class TxBank:
    """N tone generators / digit senders (tone_gen, dtmf_tx, bell_mf_tx, r2_mf_tx), state in HBM."""

    def __init__(self, kind, n_channels, device=0):
        self.kind = kind
        self.n = n_channels
        self.h = C.c_void_p()
        _check(lib().spangpu_txbank_create(C.byref(self.h), device, kind, n_channels))

    def close(self):
        if self.h:
            lib().spangpu_txbank_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _range(self, first, n):
        return first, (self.n - first) if n is None else n

    def set_stream(self, hip_stream):
        _check(lib().spangpu_txbank_set_stream(self.h, hip_stream))

    def sync(self):
        _check(lib().spangpu_txbank_sync(self.h))

    def tone(self, f1, l1, f2, l2, d1, d2=0, d3=0, d4=0, repeat=False, first=0, n=None):
        d = ToneDesc(f1, l1, f2, l2, d1, d2, d3, d4, int(repeat))
        first, n = self._range(first, n)
        _check(lib().spangpu_txbank_tone(self.h, first, n, C.byref(d)))

    def set_level(self, level, twist, first=0, n=None):
        first, n = self._range(first, n)
        _check(lib().spangpu_txbank_set_level(self.h, first, n, level, twist))

    def set_timing(self, on_ms, off_ms, first=0, n=None):
        first, n = self._range(first, n)
        _check(lib().spangpu_txbank_set_timing(self.h, first, n, on_ms, off_ms))

    def put(self, digits, first=0, n=None):
        """The same digits to every channel of the range; returns what xxx_tx_put() returns."""
        b = digits if isinstance(digits, bytes) else digits.encode()
        first, n = self._range(first, n)
        rc = lib().spangpu_txbank_put(self.h, first, n, b, len(b))
        if rc < 0:
            _check(rc)
        return rc

    def put_each(self, digit_strings, first=0):
        """One digit string per channel; returns the per-channel xxx_tx_put() results."""
        bs = [d if isinstance(d, bytes) else d.encode() for d in digit_strings]
        n = len(bs)
        stride = max(1, max(len(b) for b in bs))
        buf = np.zeros((n, stride), np.uint8)
        lens = np.zeros(n, np.int32)
        for i, b in enumerate(bs):
            buf[i, :len(b)] = np.frombuffer(b, np.uint8)
            lens[i] = len(b)
        res = np.zeros(n, np.int32)
        rc = lib().spangpu_txbank_put_each(self.h, first, n, buf.ctypes.data, stride, lens.ctypes.data, res.ctypes.data)
        if rc < 0:
            _check(rc)
        return res

    def tx_host(self, samples):
        pcm = np.zeros((self.n, samples), np.int16)
        lens = np.zeros(self.n, np.int32)
        _check(lib().spangpu_txbank_tx(self.h, MEM_HOST, pcm.ctypes.data, samples, samples, lens.ctypes.data))
        return pcm, lens

    def tx_device(self, pcm_ptr, stride, samples, lens_ptr=None):
        _check(lib().spangpu_txbank_tx(self.h, MEM_DEVICE, pcm_ptr, stride, samples, lens_ptr))

    def get_state(self, channel):
        w = np.zeros(lib().spangpu_txbank_state_words(), np.int32)
        _check(lib().spangpu_txbank_get_state(self.h, channel, w.ctypes.data))
        return w


# ---- FSK receiver banks (include/spangpu.h "FSK receiver banks") ------------------------
(FSK_V21CH1, FSK_V21CH2, FSK_V23CH1, FSK_V23CH2, FSK_BELL103CH1, FSK_BELL103CH2, FSK_BELL202, FSK_WEITBRECHT_4545,
 FSK_WEITBRECHT_50, FSK_WEITBRECHT_476, FSK_V21CH1_110) = range(11)
FSK_FRAME_MODE_ASYNC, FSK_FRAME_MODE_SYNC, FSK_FRAME_MODE_FRAMED = 0, 1, 2


class FskSpec(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("freq_zero", "freq_one", "tx_level", "min_level", "baud_rate")]


def fsk_preset(which):
    sp = FskSpec()
    _check(lib().spangpu_fsk_preset(which, C.byref(sp)))
    return sp


class FskBank:
    """N FSK receivers of one spec (fsk_rx), state in HBM."""

    def __init__(self, spec, n_channels, framing_mode=FSK_FRAME_MODE_SYNC, device=0):
        self.spec = fsk_preset(spec) if isinstance(spec, int) else spec
        self.n = n_channels
        self.h = C.c_void_p()
        _check(lib().spangpu_fsk_create(C.byref(self.h), device, n_channels, C.byref(self.spec), framing_mode))
        self.words = lib().spangpu_fsk_state_words(self.h)

    def close(self):
        if self.h:
            lib().spangpu_fsk_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(lib().spangpu_fsk_set_stream(self.h, hip_stream))

    def sync(self):
        _check(lib().spangpu_fsk_sync(self.h))

    def rx_host(self, amp):
        amp = np.ascontiguousarray(amp, np.int16)
        assert amp.shape[0] == self.n
        _check(lib().spangpu_fsk_rx(self.h, amp.ctypes.data, MEM_HOST, amp.shape[1], amp.shape[1]))

    def rx_device(self, ptr, samples, stride=0):
        _check(lib().spangpu_fsk_rx(self.h, ptr, MEM_DEVICE, samples, stride))

    def rx_host_var(self, amp, lens):
        """A tick with per-channel frame lengths (0 = the receiver sits it out, untouched)."""
        amp = np.ascontiguousarray(amp, np.int16)
        lens = np.ascontiguousarray(lens, np.int32)
        _check(lib().spangpu_fsk_rx_var(self.h, amp.ctypes.data, MEM_HOST, lens.ctypes.data, amp.shape[1], amp.shape[1]))

    def events(self):
        """Per channel: the int16 put_bit() values of the last frame, in order."""
        ev = C.c_void_p()
        cnt = C.c_void_p()
        cap = _check(lib().spangpu_fsk_events(self.h, C.byref(ev), C.byref(cnt)))
        counts = np.ctypeslib.as_array(C.cast(cnt, C.POINTER(C.c_int32)), (self.n,)).copy()
        assert counts.max(initial=0) <= cap
        flat = np.ctypeslib.as_array(C.cast(ev, C.POINTER(C.c_int16)), (self.n*cap,)).reshape(self.n, cap)
        return [flat[c, :counts[c]].copy() for c in range(self.n)]

    def get_state(self, channel):
        w = np.zeros(self.words, np.int32)
        _check(lib().spangpu_fsk_get_state(self.h, channel, w.ctypes.data))
        return w

    def set_state(self, channel, w):
        w = np.ascontiguousarray(w, np.int32)
        assert len(w) == self.words
        _check(lib().spangpu_fsk_set_state(self.h, channel, w.ctypes.data))

    def restart(self, channel, framing_mode):
        _check(lib().spangpu_fsk_restart(self.h, channel, framing_mode))

    def set_signal_cutoff(self, channel, cutoff_dbm0):
        _check(lib().spangpu_fsk_set_signal_cutoff(self.h, channel, cutoff_dbm0))

    def set_frame_parameters(self, channel, data_bits, parity, stop_bits):
        _check(lib().spangpu_fsk_set_frame_parameters(self.h, channel, data_bits, parity, stop_bits))

    def fillin(self, channel, n):
        _check(lib().spangpu_fsk_fillin(self.h, channel, n))


# ---- modem connect tone banks (include/spangpu.h "modem connect tone banks") -------------
(MCT_NONE, MCT_FAX_CNG, MCT_ANS, MCT_ANS_PR, MCT_ANSAM, MCT_ANSAM_PR, MCT_FAX_PREAMBLE, MCT_FAX_CED_OR_PREAMBLE,
 MCT_BELL_ANS, MCT_CALLING_TONE) = range(10)


class MctBank:
    """N modem connect tone detectors of one tone type (modem_connect_tones_rx), state in HBM."""

    def __init__(self, tone_type, n_channels, use_callback=True, device=0):
        self.n = n_channels
        self.h = C.c_void_p()
        _check(lib().spangpu_mct_create(C.byref(self.h), device, tone_type, n_channels, int(use_callback)))
        self.words = lib().spangpu_mct_state_words(self.h)

    def close(self):
        if self.h:
            lib().spangpu_mct_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(lib().spangpu_mct_set_stream(self.h, hip_stream))

    def sync(self):
        _check(lib().spangpu_mct_sync(self.h))

    def rx_host(self, amp):
        amp = np.ascontiguousarray(amp, np.int16)
        assert amp.shape[0] == self.n
        _check(lib().spangpu_mct_rx(self.h, amp.ctypes.data, MEM_HOST, amp.shape[1], amp.shape[1]))

    def rx_device(self, ptr, samples, stride=0):
        _check(lib().spangpu_mct_rx(self.h, ptr, MEM_DEVICE, samples, stride))

    def rx_host_var(self, amp, lens):
        """A tick with per-channel frame lengths (0 = the detector sits it out, untouched)."""
        amp = np.ascontiguousarray(amp, np.int16)
        lens = np.ascontiguousarray(lens, np.int32)
        _check(lib().spangpu_mct_rx_var(self.h, amp.ctypes.data, MEM_HOST, lens.ctypes.data, amp.shape[1], amp.shape[1]))

    def events(self):
        """Per channel: [k, 2] int32 (tone, level) reports of the last frame, in order."""
        ev = C.c_void_p()
        cnt = C.c_void_p()
        cap = _check(lib().spangpu_mct_events(self.h, C.byref(ev), C.byref(cnt)))
        counts = np.ctypeslib.as_array(C.cast(cnt, C.POINTER(C.c_int32)), (self.n,)).copy()
        assert counts.max(initial=0) <= cap
        flat = np.ctypeslib.as_array(C.cast(ev, C.POINTER(C.c_int32)), (self.n*cap*2,)).reshape(self.n, cap, 2)
        return [flat[c, :counts[c]].copy() for c in range(self.n)]

    def get(self, channel):
        return _check(lib().spangpu_mct_get(self.h, channel))

    def get_state(self, channel):
        w = np.zeros(self.words, np.int32)
        _check(lib().spangpu_mct_get_state(self.h, channel, w.ctypes.data))
        return w


# ---- signalling tone banks (include/spangpu.h "signalling tone banks") ---------------------
(SIG_TONE_2280HZ, SIG_TONE_2600HZ, SIG_TONE_2400HZ_2600HZ) = (1, 2, 3)
SIG_TONE_1_PRESENT, SIG_TONE_1_CHANGE, SIG_TONE_2_PRESENT, SIG_TONE_2_CHANGE = 0x001, 0x002, 0x004, 0x008
SIG_TONE_TX_PASSTHROUGH, SIG_TONE_RX_PASSTHROUGH, SIG_TONE_RX_FILTER_TONE = 0x010, 0x040, 0x080
SIG_TONE_TX_UPDATE_REQUEST = 0x100


class SigToneRxBank:
    """N in-band signalling tone receivers of one tone type (sig_tone_rx), state in HBM; frames are rewritten in place."""

    def __init__(self, tone_type, n_channels, device=0):
        self.n = n_channels
        self.h = C.c_void_p()
        _check(lib().spangpu_sigtone_rx_create(C.byref(self.h), device, tone_type, n_channels))
        self.words = lib().spangpu_sigtone_rx_state_words(self.h)

    def close(self):
        if self.h:
            lib().spangpu_sigtone_rx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(lib().spangpu_sigtone_rx_set_stream(self.h, hip_stream))

    def sync(self):
        _check(lib().spangpu_sigtone_rx_sync(self.h))

    def set_mode(self, mode, channel=-1):
        _check(lib().spangpu_sigtone_rx_set_mode(self.h, channel, mode))

    def rx_host(self, amp):
        """Returns the frames as the receivers left them."""
        buf = np.array(amp, np.int16, order="C", copy=True)
        assert buf.shape[0] == self.n
        _check(lib().spangpu_sigtone_rx(self.h, buf.ctypes.data, MEM_HOST, buf.shape[1], buf.shape[1]))
        return buf

    def rx_device(self, ptr, samples, stride=0):
        _check(lib().spangpu_sigtone_rx(self.h, ptr, MEM_DEVICE, samples, stride))

    def rx_host_var(self, amp, lens):
        """A tick with per-channel frame lengths (0 = the receiver sits it out, state and row untouched)."""
        buf = np.array(amp, np.int16, order="C", copy=True)
        lens = np.ascontiguousarray(lens, np.int32)
        _check(lib().spangpu_sigtone_rx_var(self.h, buf.ctypes.data, MEM_HOST, lens.ctypes.data, buf.shape[1], buf.shape[1]))
        return buf

    def events(self):
        """Per channel: [k, 3] int32 (sample of the call, signalling_state, duration) reports of the last call, in order."""
        ev = C.c_void_p()
        cnt = C.c_void_p()
        cap = _check(lib().spangpu_sigtone_rx_events(self.h, C.byref(ev), C.byref(cnt)))
        counts = np.ctypeslib.as_array(C.cast(cnt, C.POINTER(C.c_int32)), (self.n,)).copy()
        flat = np.ctypeslib.as_array(C.cast(ev, C.POINTER(C.c_int32)), (self.n*cap*3,)).reshape(self.n, cap, 3)
        return [flat[c, :counts[c]].copy() for c in range(self.n)]

    def get_state(self, channel):
        w = np.zeros(self.words, np.int32)
        _check(lib().spangpu_sigtone_rx_get_state(self.h, channel, w.ctypes.data))
        return w

    def set_state(self, channel, words):
        w = np.ascontiguousarray(words, np.int32)
        assert len(w) == self.words
        _check(lib().spangpu_sigtone_rx_set_state(self.h, channel, w.ctypes.data))

    def thresholds(self):
        out = np.zeros(3, np.int32)
        _check(lib().spangpu_sigtone_rx_thresholds(self.h, out.ctypes.data))
        return out


class SigToneTxBank:
    """N in-band signalling tone senders of one tone type (sig_tone_tx), state in HBM; frames are rewritten in place."""

    def __init__(self, tone_type, n_channels, device=0):
        self.n = n_channels
        self.h = C.c_void_p()
        _check(lib().spangpu_sigtone_tx_create(C.byref(self.h), device, tone_type, n_channels))
        self.words = lib().spangpu_sigtone_tx_state_words(self.h)

    def close(self):
        if self.h:
            lib().spangpu_sigtone_tx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mode(self, mode, duration, channel=-1):
        _check(lib().spangpu_sigtone_tx_set_mode(self.h, channel, mode, duration))

    def set_modes(self, modes, durations):
        modes = np.ascontiguousarray(modes, np.int32)
        durations = np.ascontiguousarray(durations, np.int32)
        assert len(modes) == self.n and len(durations) == self.n
        _check(lib().spangpu_sigtone_tx_set_modes(self.h, modes.ctypes.data, durations.ctypes.data))

    def tx_host(self, amp, on_request=None):
        """One frame for every channel.  on_request(channels) -> (modes, durations) arrays for those channels is what the
        reference's update-request callback does; without it a request is simply acknowledged."""
        buf = np.array(amp, np.int16, order="C", copy=True)
        assert buf.shape[0] == self.n
        pending = _check(lib().spangpu_sigtone_tx(self.h, buf.ctypes.data, MEM_HOST, buf.shape[1], buf.shape[1]))
        rounds = 0
        while pending > 0:
            req, _ = self.requests()
            who = np.nonzero(req)[0]
            if on_request is not None:
                modes = np.full(self.n, -1, np.int32)
                durs = np.zeros(self.n, np.int32)
                m, d = on_request(who)
                modes[who] = m
                durs[who] = d
                self.set_modes(modes, durs)
            pending = _check(lib().spangpu_sigtone_tx_continue(self.h, buf.ctypes.data, MEM_HOST, buf.shape[1]))
            rounds += 1
            assert rounds <= buf.shape[1] + 1
        return buf

    def requests(self):
        req = C.c_void_p()
        stop = C.c_void_p()
        _check(lib().spangpu_sigtone_tx_requests(self.h, C.byref(req), C.byref(stop)))
        r = np.ctypeslib.as_array(C.cast(req, C.POINTER(C.c_int32)), (self.n,)).copy()
        s = np.ctypeslib.as_array(C.cast(stop, C.POINTER(C.c_int32)), (self.n,)).copy()
        return r, s

    def get_state(self, channel):
        w = np.zeros(self.words, np.int32)
        _check(lib().spangpu_sigtone_tx_get_state(self.h, channel, w.ctypes.data))
        return w


# ---- modem transmitter banks (include/spangpu.h "modem transmitter banks") -----------------
def modem_tx_table(which):
    """which: 0 V.29 [10, 9], 1 V.27ter 4800 bps [5, 9], 2 V.27ter 2400 bps [20, 9] -- flat float32."""
    out = np.zeros(180, np.float32)
    n = _check(lib().spangpu_modemtx_table(which, out.ctypes.data, 180))
    return out[:n].copy()


def v29_tx_table():
    return modem_tx_table(0)


class ModemTxBank:
    """N V.29 / V.27ter modulators (v29_tx, v27ter_tx), state in HBM; data bits from a per-channel LFSR."""

    def __init__(self, modem, n_channels, bit_rate, tep=False, seeds=None, device=0):
        self.n = n_channels
        self.h = C.c_void_p()
        sp = None
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, np.uint32)
            assert len(seeds) == n_channels
            sp = seeds.ctypes.data
        _check(lib().spangpu_modemtx_create(C.byref(self.h), device, modem, n_channels, bit_rate, int(tep), sp))

    def close(self):
        if self.h:
            lib().spangpu_modemtx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(lib().spangpu_modemtx_set_stream(self.h, hip_stream))

    def sync(self):
        _check(lib().spangpu_modemtx_sync(self.h))

    def power(self, channel, level_dbm0):
        _check(lib().spangpu_modemtx_power(self.h, channel, level_dbm0))

    def line(self, power_dbm0=None, carrier_hz=None):
        """every channel's level and / or carrier frequency at once (the carrier: a line model, see include/spangpu.h)"""
        p = None if power_dbm0 is None else np.ascontiguousarray(power_dbm0, np.float32)
        f = None if carrier_hz is None else np.ascontiguousarray(carrier_hz, np.float32)
        assert (p is None or len(p) == self.n) and (f is None or len(f) == self.n)
        _check(lib().spangpu_modemtx_line(self.h, None if p is None else p.ctypes.data, None if f is None else f.ctypes.data))

    def restart(self, channel, bit_rate, tep, short_train=False):
        _check(lib().spangpu_modemtx_restart_ex(self.h, channel, bit_rate, int(tep), int(short_train)))

    def tx_host(self, samples):
        pcm = np.zeros((self.n, samples), np.int16)
        _check(lib().spangpu_modemtx_tx(self.h, MEM_HOST, pcm.ctypes.data, samples, samples))
        return pcm

    def tx_device(self, pcm_ptr, stride, samples):
        _check(lib().spangpu_modemtx_tx(self.h, MEM_DEVICE, pcm_ptr, stride, samples))

    def get_state(self, channel):
        w = np.zeros(lib().spangpu_modemtx_state_words(), np.uint32)
        _check(lib().spangpu_modemtx_get_state(self.h, channel, w.ctypes.data))
        return w


class V29TxBank(ModemTxBank):
    def __init__(self, n_channels, bit_rate=9600, tep=False, seeds=None, device=0):
        super().__init__(V29, n_channels, bit_rate, tep, seeds, device)


class V27terTxBank(ModemTxBank):
    def __init__(self, n_channels, bit_rate=4800, tep=False, seeds=None, device=0):
        super().__init__(V27TER, n_channels, bit_rate, tep, seeds, device)


class V17TxBank(ModemTxBank):
    def __init__(self, n_channels, bit_rate=14400, tep=False, seeds=None, device=0):
        super().__init__(V17, n_channels, bit_rate, tep, seeds, device)


class AwgnBank:
    """N Gaussian noise generators (awgn_init_dbm0 / awgn), state in HBM."""

    def __init__(self, seeds, levels_dbm0, device=0):
        seeds = np.ascontiguousarray(seeds, np.int32)
        levels = np.ascontiguousarray(levels_dbm0, np.float32)
        assert len(seeds) == len(levels)
        self.n = len(seeds)
        self.h = C.c_void_p()
        _check(lib().spangpu_awgn_create(C.byref(self.h), device, self.n, seeds.ctypes.data, levels.ctypes.data))

    def close(self):
        if self.h:
            lib().spangpu_awgn_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(lib().spangpu_awgn_set_stream(self.h, hip_stream))

    def sync(self):
        _check(lib().spangpu_awgn_sync(self.h))

    def reinit(self, channel, seed, level_dbm0):
        _check(lib().spangpu_awgn_reinit(self.h, channel, seed, level_dbm0))

    def tx_host(self, samples, mix_into=None):
        if mix_into is None:
            pcm = np.zeros((self.n, samples), np.int16)
        else:
            pcm = np.ascontiguousarray(mix_into, np.int16).copy()
            assert pcm.shape == (self.n, samples)
        _check(lib().spangpu_awgn_tx(self.h, MEM_HOST, pcm.ctypes.data, samples, samples, int(mix_into is not None)))
        return pcm

    def tx_device(self, pcm_ptr, stride, samples, mix=False):
        _check(lib().spangpu_awgn_tx(self.h, MEM_DEVICE, pcm_ptr, stride, samples, int(mix)))

    def get_state(self, channel):
        w = np.zeros(lib().spangpu_awgn_state_words(self.h), np.uint32)
        _check(lib().spangpu_awgn_get_state(self.h, channel, w.ctypes.data))
        return w
