# Run on the GPU box (gpurun): kernel-trace stats for bench.py and the secondary workloads, and the HBM traffic counters
# (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, kernel-trace only) for bench.py.  Outputs under gpurun_out/prof/.
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $R
export TMPDIR=/tmp
cd /tmp
timeout 300 python $GRAFT_REPO_ROOT/bench.py > $R/bench.json 2> $R/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline > $R/bench_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/bench_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/bench_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/bench_write -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/bench_write.log 2>&1
for w in mixed v29 v17 v27ter echo dtmf_tx fsk mct v29_tx awgn; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${w}_stats -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --no-cpu-baseline > $R/${w}_stats.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json, os
R = "gpurun_out/prof"
out = {}
for name in ("bench_fetch", "bench_write"):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (R, name)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "tone_bank_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = {"launches": len(v), "mean": sum(v)/len(v), "min": min(v), "max": max(v)}
print(json.dumps(out))
json.dump(out, open(R + "/hbm_counters_raw.json", "w"), indent=1)
for f in sorted(glob.glob(R + "/*_stats/*/*kernel_stats.csv")):
    print(f)
    print(open(f).read()[:1500])
PY
tail -2 $R/bench.json
