# Round 6 measurements (run on the GPU box through gpurun): bash tools/gpu_round6.sh <what> [...]
# Everything lands under gpurun_out/r6/; the summaries that are judged are copied into profiles/ (r6_*).
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r6
mkdir -p $R
export TMPDIR=/tmp
for what in "$@"; do
case "$what" in
final)
  # smoke, the whole GPU suite, bench.py (the driver's command), and its kernel statistics
  python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
  timeout 2400 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $R/pytest_gpu.log
  cd /tmp
  timeout 600 python $GRAFT_REPO_ROOT/bench.py > $R/bench.json 2> $R/bench.err; echo "bench rc=$?"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-e2e --no-paths > $R/bench_stats.log 2>&1
  find $R/bench_stats -name "*kernel_stats.csv" -exec cp {} $R/bench_kernel_stats.csv \;
  rm -rf $R/bench_stats
  cd $GRAFT_REPO_ROOT
  tail -2 $R/smoke.log; head -3 $R/bench_kernel_stats.csv | cut -c1-220
  ;;
paths)
  # every BASELINE path (and the other receivers) by tools/bench_paths.py, with the rocprofv3 kernel statistics of the same commands
  mkdir -p $R/paths
  for w in ${PATHS_W:-mixed v29 v17 v27ter echo supertone}; do
    timeout 600 python tools/bench_paths.py --workload $w > $R/paths/$w.json 2> $R/paths/$w.err; echo "$w rc=$?"
    cd /tmp
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_$w -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --no-cpu-baseline --no-e2e --echo-seconds 3 $( [ $w = mixed ] && echo --single-mode ) > $R/stats_$w.log 2>&1
    cd $GRAFT_REPO_ROOT
    find $R/stats_$w -name "*kernel_stats.csv" -exec cp {} $R/${w}_kernel_stats.csv \;
    if [ $w = mixed ]; then find $R/stats_$w -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_overlap.py {} _fast_kernel > $R/mixed_trace_overlap.txt 2>&1; fi
    rm -rf $R/stats_$w
  done
  python tools/bench_paths.py --workload mixed --no-cpu-baseline --three-queues > $R/paths/mixed_three_queues.json 2>/dev/null
  python tools/bench_paths.py --workload v29 --no-cpu-baseline --line in_step > $R/paths/v29_in_step.json 2>/dev/null
  ;;
hbm)
  # FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of the kernels whose sources changed this round
  cd /tmp
  for w in ${HBM_W:-mixed}; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/hbm_${w}_$c -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --steps 40 --no-cpu-baseline --no-e2e --echo-seconds 2 $( [ $w = mixed ] && echo --single-mode ) > $R/hbm_${w}_$c.log 2>&1
      echo "$w $c rc=$?"
    done
  done
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/hbm_dtmf_FETCH_SIZE -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-paths > $R/hbm_dtmf_FETCH_SIZE.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/hbm_dtmf_WRITE_SIZE -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-paths > $R/hbm_dtmf_WRITE_SIZE.log 2>&1
  cd $GRAFT_REPO_ROOT
  python3 tools/hbm_summary.py $R > $R/hbm_traffic_raw.json
  cat $R/hbm_traffic_raw.json | head -60
  find $R -mindepth 1 -maxdepth 1 -type d -name 'hbm_*' -exec rm -rf {} +
  ;;
valu)
  VALU_MODEMS="" VALU_W="mixed supertone" ROUND=6 bash tools/gpu_valu.sh > $R/valu.log 2>&1; tail -5 $R/valu.log
  cp gpurun_out/valu/valu_counters.json $R/valu_counters.json
  ;;
esac
done
