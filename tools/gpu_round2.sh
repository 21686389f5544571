# Round 2 measurements (run on the GPU box through gpurun): the bench lines, their rocprofv3 kernel statistics, the HBM
# traffic counters of the headline kernel (separate --pmc passes, kernel-trace only), the secondary workloads, and the
# probes quoted in DESIGN.md.  Everything lands under gpurun_out/r2/; the summaries that are judged are copied into profiles/.
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $R
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 900 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
cd /tmp
timeout 400 python $GRAFT_REPO_ROOT/bench.py > $R/bench.json 2> $R/bench.err
timeout 400 python $GRAFT_REPO_ROOT/bench.py --workload echo > $R/bench_echo.json 2> $R/bench_echo.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-e2e > $R/bench_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_echo_stats -- python $GRAFT_REPO_ROOT/bench.py --workload echo --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $R/bench_echo_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/bench_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $R/bench_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/bench_write -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $R/bench_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/bench_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $R/bench_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/bench_echo_sq -- python $GRAFT_REPO_ROOT/bench.py --workload echo --steps 60 --warmup 10 --no-cpu-baseline --no-e2e > $R/bench_echo_sq.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p $R/paths
for w in mixed v29 v17 v27ter echo dtmf_tx fsk mct v29_tx awgn; do
  timeout 500 python tools/bench_paths.py --workload $w > $R/paths/$w.json 2> $R/paths/$w.err; echo "$w rc=$?"
done
cd /tmp
for w in mixed v29 v17 v27ter; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${w}_stats -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --no-cpu-baseline > $R/${w}_stats.log 2>&1
done
cd $GRAFT_REPO_ROOT
{
  echo "### tools/probe2 (tone kernels: variants, sizes; eager)"; timeout 200 ./tools/probe2
  echo "### tools/probe2 quick graph (same launches from a hipGraph)"; timeout 100 ./tools/probe2 quick graph
  echo "### tools/probe5 (launch boundary)"; timeout 100 ./tools/probe5
  echo "### tools/probe6 (integer instruction issue, echo kernels)"; timeout 100 ./tools/probe6; echo "(rc=$?)"
  echo "### tools/probe3 x (Goertzel step2 block issue)"; timeout 100 ./tools/probe3 x
  echo "### tools/echo_ab.py (every channel adapting in step)"; for n in 32768 65536 131072; do python tools/echo_ab.py $n 8 4 2 2>&1 | grep lanes; done
  echo "### tools/echo_wl.sh (mixed lines: tools/bench_paths.py --workload echo --echo-lanes G)"; bash tools/echo_wl.sh 2>&1 | grep channels
} > $R/probe.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
R = "gpurun_out/r2"
out = {}
for name in ("bench_fetch", "bench_write", "bench_sq", "bench_echo_sq"):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (R, name)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "tone_fast_kernel" in k or "tone_bank_kernel" in k or "echo_" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(name, {})[k] = {c: {"launches": len(x), "mean": sum(x)/len(x)} for c, x in v.items()}
json.dump(out, open(R + "/counters_raw.json", "w"), indent=1)
print(json.dumps(out)[:3000])
PY
tail -c 1200 $R/bench.json; tail -c 1200 $R/bench_echo.json; tail -5 $R/bench.err
