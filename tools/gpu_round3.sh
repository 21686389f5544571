# Round 3 measurements (run on the GPU box through gpurun): bash tools/gpu_round3.sh tests | bench | paths | valu
# Everything lands under gpurun_out/r3/; the summaries that are judged are copied into profiles/ (r3_*).
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $R
export TMPDIR=/tmp
case "$1" in
tests)
  python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
  timeout 1500 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
  tail -4 $R/pytest_gpu.log; tail -2 $R/smoke.log
  ;;
bench)
  cd /tmp
  timeout 400 python $GRAFT_REPO_ROOT/bench.py > $R/bench.json 2> $R/bench.err
  timeout 400 python $GRAFT_REPO_ROOT/bench.py --workload echo > $R/bench_echo.json 2> $R/bench_echo.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-e2e > $R/bench_stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_echo_stats -- python $GRAFT_REPO_ROOT/bench.py --workload echo --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $R/bench_echo_stats.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/bench_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $R/bench_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/bench_write -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $R/bench_write.log 2>&1
  cd $GRAFT_REPO_ROOT
  python3 - <<'PY'
import csv, glob, collections, json
R = "gpurun_out/r3"
out = {}
for name in ("bench_fetch", "bench_write"):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (R, name)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "tone_fast_kernel" in k or "tone_bank_kernel" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(name, {})[k] = {c: {"launches": len(x), "mean": sum(x)/len(x)} for c, x in v.items()}
json.dump(out, open(R + "/counters_raw.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
  for d in bench_stats bench_echo_stats; do cp $(ls $R/$d/*/*kernel_stats.csv | head -1) $R/$d.csv; done
  tail -c 1500 $R/bench.json; tail -c 600 $R/bench_echo.json
  ;;
paths)
  mkdir -p $R/paths
  for w in mixed v29 v17 v27ter echo dtmf_tx fsk mct sigtone supertone fax_rx v29_tx awgn; do
    timeout 500 python tools/bench_paths.py --workload $w > $R/paths/$w.json 2> $R/paths/$w.err; echo "$w rc=$?"
    python3 -c "import json;d=json.load(open('$R/paths/$w.json'));print('$w', d['ms_per_step'], d['roofline'].get('avg_launch_us'), d['value'], d['unit'])"
  done
  for w in fsk mct; do
    timeout 300 python tools/bench_paths.py --workload $w --fsk-waves 1 --no-cpu-baseline > $R/paths/${w}_one_wave.json 2> $R/paths/${w}_one_wave.err
  done
  SUPERTONE_QUIET=1 timeout 300 python tools/bench_paths.py --workload supertone --no-cpu-baseline > $R/paths/supertone_quiet.json 2> $R/paths/supertone_quiet.err
  cd /tmp
  for w in v29 v17 v27ter fsk mct sigtone; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${w}_stats -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --no-cpu-baseline > $R/${w}_stats.log 2>&1
    cp $(ls $R/${w}_stats/*/*kernel_stats.csv | head -1) $R/${w}_kernel_stats.csv
  done
  ;;
valu)
  bash tools/gpu_valu.sh > $R/valu.log 2>&1
  cp gpurun_out/valu/valu_counters.json $R/ 2>/dev/null
  tail -30 $R/valu.log
  ;;
esac
