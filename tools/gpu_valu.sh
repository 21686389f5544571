# VALU instruction counts of every workload's dominant kernel (rocprofv3 --pmc, kernel-trace only) -> profiles/valu_counters.json,
# which spandsp_amd/roofline.py turns into the `roofline_valu` object of the bench lines.  Run on the GPU box: bash tools/gpu_valu.sh
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/valu
rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/dtmf -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-paths > $R/dtmf.log 2>&1
for w in ${VALU_MODEMS-v29 v17 v27ter}; do
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/$w -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --steps 40 --warmup 110 --no-cpu-baseline --no-e2e > $R/$w.log 2>&1
done
for w in ${VALU_W:-echo mixed fsk mct sigtone supertone dtmf_tx v29_tx awgn}; do
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/$w -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --steps 30 --no-cpu-baseline --no-e2e --echo-seconds 2 $( [ $w = mixed ] && echo --single-mode ) > $R/$w.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json, os
R = "gpurun_out/valu"
want = {"dtmf": ("tone_fast_kernel", 65536), "v29": ("v29_quad_kernel", 16384), "v17": ("v17_quad_kernel", 16384), "v27ter": ("v27ter_", 16384),
        "echo": ("echo_", 131072), "mixed": ("_fast_kernel", 131072), "fsk": ("fsk_", 65536), "mct": ("mct_", 65536),
        "sigtone": ("sigtone_rx_kernel", 65536), "supertone": ("tone_fast_kernel", 65536), "dtmf_tx": ("tx_bank_kernel", 65536),
        "v29_tx": ("modemtx_bank_kernel", 65536), "awgn": ("awgn_bank_kernel", 65536)}
out = {"note": "rocprofv3 --pmc SQ_* (kernel-trace only) means per launch of each workload's dominant kernel (the one with the most "
               "SQ_WAVE_CYCLES among the names matched); tools/gpu_valu.sh", "workloads": {}}
for key, (pat, n_ch) in want.items():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (R, key)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat in k and "stats_kernel" not in k and "erle" not in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        continue
    best = max(acc, key=lambda k: sum(acc[k].get("SQ_WAVE_CYCLES", [0])))
    m = {c: sum(v)/len(v) for c, v in acc[best].items()}
    if key == "mixed":
        # a tick is a launch per bank (three kernels on three streams): the tick's counts are the sums of the kernels' means
        m = collections.defaultdict(float)
        for k in acc:
            for c, v in acc[k].items():
                m[c] += sum(v)/len(v)
        best = " + ".join(sorted(k.split("spg::")[-1].split(",")[0] for k in acc))
    waves = m.get("SQ_WAVES", 0) or 1
    out["workloads"][key] = {"kernel": best.split("(")[0][:90], "channels": n_ch, "valu_insts_per_launch": m.get("SQ_INSTS_VALU"),
                             "salu_insts_per_launch": m.get("SQ_INSTS_SALU"), "lds_insts_per_launch": m.get("SQ_INSTS_LDS"),
                             "waves": waves, "valu_insts_per_wave_sample": m.get("SQ_INSTS_VALU", 0)/waves/160.0,
                             "wave_cycles_per_wave_sample": 4.0*m.get("SQ_WAVE_CYCLES", 0)/waves/160.0,
                             "active_frac": m.get("SQ_ACTIVE_INST_ANY", 0)/max(1.0, m.get("SQ_WAVE_CYCLES", 1)),
                             "wait_frac": m.get("SQ_WAIT_ANY", 0)/max(1.0, m.get("SQ_WAVE_CYCLES", 1)),
                             "launches": len(acc[best].get("SQ_INSTS_VALU", [])), "source": "tools/gpu_valu.sh (round %s)" % os.environ.get("ROUND", "6")}
# workloads not measured in this run keep their record (profiles/valu_counters.json)
try:
    old = json.load(open("profiles/valu_counters.json"))["workloads"]
    for k, v in old.items():
        out["workloads"].setdefault(k, v)
except Exception:
    pass
json.dump(out, open(os.path.join(R, "valu_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
# (the raw per-dispatch tables are tens of MB a workload: only the summary travels back)
find gpurun_out/valu -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
