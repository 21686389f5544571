"""Bounds quoted beside the HBM roofline in the benchmark lines (bench.py, tools/bench_paths.py):

* `measured_stream_peak()` -- what a plain streaming read kernel reaches on THIS device in THIS run (the practical ceiling
  under the 8 TB/s datasheet peak);
* `valu_roof()` -- the VALU issue floor of a workload's dominant kernel: its VALU instruction count per launch (rocprofv3
  SQ_INSTS_VALU, committed in profiles/valu_counters.json by tools/gpu_valu.sh) spread over the chip's 1 024 SIMDs at
  one instruction per four cycles (what one wavefront per SIMD can issue; packed fp32 and 64-lane integer operations
  also occupy the SIMD-32 pipe for four), at the 2.4 GHz peak clock.  frac = floor / measured launch time: how much of
  the launch is explained by instruction issue alone -- the bound that matters for the state machine kernels (modem
  receivers, echo canceller), whose HBM fraction is by construction a few per cent."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 1024
CLOCK_MHZ = 2400.0
CYCLES_PER_VALU = 4.0
_stream = {}


def measured_stream_peak(device=0):
    """GB/s of a 1 GiB streaming read with 16-byte loads (cached per process); None if it cannot be measured."""
    if device not in _stream:
        try:
            from . import engine
            _stream[device] = engine.probe_stream_read(device, 1 << 30, 8)
        except Exception:
            _stream[device] = None
    return _stream[device]


def valu_roof(key, avg_launch_us, channels=None, samples=160):
    """The VALU issue floor of workload `key` from profiles/valu_counters.json, scaled to `channels` if the counters were
    taken at another bank size (instruction counts are per channel-sample)."""
    path = os.path.join(ROOT, "profiles", "valu_counters.json")
    try:
        rec = json.load(open(path))["workloads"].get(key)
    except Exception:
        rec = None
    if not rec or not avg_launch_us:
        return None
    valu = float(rec["valu_insts_per_launch"])
    if channels and rec.get("channels") and channels != rec["channels"]:
        valu *= float(channels)/float(rec["channels"])
    floor_us = valu/SIMDS*CYCLES_PER_VALU/CLOCK_MHZ
    out = {"bound": "valu_issue", "kernel": rec.get("kernel"), "valu_insts_per_launch": valu,
           "valu_insts_per_wave_sample": rec.get("valu_insts_per_wave_sample"), "issue_floor_us": floor_us,
           "frac": floor_us/avg_launch_us, "cycles_per_valu": CYCLES_PER_VALU, "simds": SIMDS, "clock_mhz": CLOCK_MHZ,
           # (for the reader: a lone wavefront also pays about four cycles for every scalar instruction, branch and wait --
           # DESIGN 4.5; the floor counts vector instructions only)
           "salu_insts_per_wave_sample": (rec.get("salu_insts_per_launch", 0.0)/max(1.0, rec.get("waves", 1.0))/float(samples)),
           "waves_per_simd": rec.get("waves", 0.0)/SIMDS, "wait_frac": rec.get("wait_frac"),
           "source": rec.get("source")}
    return out


def add_measured(roof, device=0):
    """roofline object + the measured stream ceiling and the fraction of it."""
    if roof is None:
        return None
    peak = measured_stream_peak(device)
    roof["measured_stream_peak"] = peak
    roof["frac_of_measured_stream"] = (roof["achieved"]/peak) if peak else None
    return roof


# ---- HBM bytes per launch from the PMC counters (profiles/hbm_traffic.json, written by tools/hbm_traffic.py from the rocprofv3
# FETCH_SIZE / WRITE_SIZE passes of tools/gpu_round5.sh hbm) ---------------------------------------------------------------
# The counters are taken once per round, in passes of their own; what makes them a statement about the kernel a bench run has
# just timed is the hash of the kernel's sources recorded with them: sources that differ null the figure.
_KERNEL_SOURCES = {
    "dtmf": ["tone_fast.hpp", "tone_dev.hpp", "tone_pairs_asm.inc"],
    "mixed": ["tone_fast.hpp", "tone_dev.hpp", "tone_pairs_asm.inc", "cadence_dev.hpp"],
    "v29": ["v29_quad.hpp", "v29_common.hpp", "quad_round_front.inc", "quad_ctx.hpp"],
    "v17": ["v17_quad.hpp", "v17_common.hpp", "v29_common.hpp", "quad_round_front.inc", "quad_ctx.hpp"],
    "v27ter": ["v27ter_quad.hpp", "v27ter_common.hpp", "v29_common.hpp", "quad_ctx.hpp"],
    "echo": ["echo_dev.hpp"],
}


def source_hash(key):
    import hashlib
    h = hashlib.sha256()
    for name in _KERNEL_SOURCES.get(key, []):
        try:
            h.update(open(os.path.join(ROOT, "spandsp_amd", "csrc", name), "rb").read())
        except OSError:
            h.update(b"missing:" + name.encode())
    return h.hexdigest()[:16]


def hbm_traffic(key, channels=None):
    """{"bytes": read + written per launch, "read": .., "write": .., ...} of workload `key` from profiles/hbm_traffic.json, scaled
    to `channels`; None when there is no record or the kernel's sources have changed since the counters were taken."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))["workloads"].get(key)
    except Exception:
        rec = None
    if not rec or rec.get("source_hash") != source_hash(key):
        return None
    scale = 1.0
    if channels and rec.get("channels") and channels != rec["channels"]:
        scale = float(channels)/float(rec["channels"])
    return {"bytes": (rec["read_bytes_per_launch"] + rec["write_bytes_per_launch"])*scale,
            "read": rec["read_bytes_per_launch"]*scale, "write": rec["write_bytes_per_launch"]*scale,
            "kernel": rec.get("kernel"), "source_hash": rec.get("source_hash"), "round": rec.get("round")}


def add_traffic(roof, key, channels=None):
    """Fill roofline.traffic (HBM bytes per launch, read + written) and the ratios to the algorithmic bytes."""
    if roof is None:
        return None
    t = hbm_traffic(key, channels)
    if t is None:
        roof["traffic"] = None
        return roof
    roof["traffic"] = t["bytes"]
    roof["traffic_read"] = t["read"]
    roof["traffic_write"] = t["write"]
    if roof.get("alg_read_bytes_per_launch"):
        roof["traffic_read_over_algorithmic"] = t["read"]/roof["alg_read_bytes_per_launch"]
    if roof.get("state_bytes_actual_read_per_launch"):
        roof["traffic_read_over_actual_state"] = t["read"]/roof["state_bytes_actual_read_per_launch"]
    roof["traffic_source"] = "profiles/hbm_traffic.json (rocprofv3 FETCH_SIZE x 2, WRITE_SIZE x 1: profiles/r5_hbm_calibration.json), kernel sources %s" % t["source_hash"]
    return roof
