"""profiles/hbm_traffic.json from the raw rocprofv3 counters of tools/gpu_round5.sh hbm (gpurun_out/r5/hbm_traffic_raw.json) and
the calibration of the counters on this repository's access patterns (profiles/r5_hbm_calibration.json: FETCH_SIZE reports half
the bytes of every load shape the kernels use -- dword per lane, 16 bytes per lane, a lane's own row 16 bytes at a time; WRITE_SIZE
the bytes of the store shapes).  Usage: python3 tools/hbm_traffic.py <hbm_traffic_raw.json> [round]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spandsp_amd import roofline as rl  # noqa: E402

WANT = {
    "dtmf": ("tone_fast_kernel<spg::DtmfDet<false>", 65536),
    "mixed": ("_fast_kernel<", 131072),              # a tick's launches (round 6: the super-tone bank with its cadence matcher | Bell MF + R2 MF in one launch): summed below
    "v29": ("v29_quad_kernel", 16384),
    "v17": ("v17_quad_kernel", 16384),
    "v27ter": ("v27ter_quad_kernel", 16384),
    "echo": ("echo_bank_kernel", 131072),
}


def main():
    raw = json.load(open(sys.argv[1]))
    rnd = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over tools/bench_paths.py "
                     "--workload <w> and bench.py; tools/gpu_round5.sh hbm, tools/hbm_traffic.py",
           "correction": "FETCH_SIZE (KiB) x 2: on gfx950 it reports half the bytes of coalesced reads (MI355X_MICROARCH.md, HBM section), "
                         "and of every load shape of these kernels (profiles/r5_hbm_calibration.json: 0.5000 for a dword per lane, 16 bytes "
                         "per lane, and a lane's own 128-byte row); WRITE_SIZE (KiB) as is (1.000 for the same store shapes)",
           "round": rnd, "workloads": {}}
    # workloads this run did not measure keep their record (and its source hash: it still says what it was taken on)
    try:
        out["workloads"].update(json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get("workloads", {}))
    except Exception:
        pass
    for key, (pat, channels) in WANT.items():
        ks = {k: v for k, v in raw.get(key, {}).items() if pat in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v}
        if not ks:
            continue
        if key == "mixed":
            # the banks' launches of a tick (or the one launch that serves all three): per tick = sum over kernels of mean x launches / ticks
            multi = {k: v for k, v in raw.get(key, {}).items() if "tone_multi_fast_kernel" in k and "FETCH_SIZE" in v}
            ticks = min(v["FETCH_SIZE"]["launches"] for v in ks.values())
            rd = sum(v["FETCH_SIZE"]["mean_KiB"]*v["FETCH_SIZE"]["launches"] for v in ks.values())/ticks*1024.0*2.0
            wr = sum(v["WRITE_SIZE"]["mean_KiB"]*v["WRITE_SIZE"]["launches"] for v in ks.values())/min(v["WRITE_SIZE"]["launches"] for v in ks.values())*1024.0
            name = " + ".join(sorted(k.split("spg::")[-1][:40] for k in ks))
            if multi and not ks:
                continue
        else:
            k, v = max(ks.items(), key=lambda kv: kv[1]["FETCH_SIZE"]["launches"])
            rd = v["FETCH_SIZE"]["mean_KiB"]*1024.0*2.0
            wr = v["WRITE_SIZE"]["mean_KiB"]*1024.0
            name = k
        out["workloads"][key] = {"kernel": name, "channels": channels, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                                 "source_hash": rl.source_hash(key), "round": rnd}
    # (bench.py's older reader)
    if "dtmf" in out["workloads"]:
        d = out["workloads"]["dtmf"]
        out["dtmf_read_bytes_per_launch"] = d["read_bytes_per_launch"]
        out["dtmf_write_bytes_per_launch"] = d["write_bytes_per_launch"]
        out["dtmf_bytes_per_launch"] = d["read_bytes_per_launch"] + d["write_bytes_per_launch"]
    json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out["workloads"], indent=1))


if __name__ == "__main__":
    main()
