#!/bin/bash
# round 6: the echo bank with its control words word-major -- parity tests, the bench line, kernel statistics and the HBM counters
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export GRAFT_REPO_ROOT=$PWD
R=$PWD/gpurun_out/r6
mkdir -p $R
python -m pytest tests/test_echo_gpu.py tests/test_shard_gpu.py tests/test_refstate_gpu.py tests/test_shim_gpu.py -x -q -m gpu -k "echo or Echo" > $R/echo_tests.log 2>&1; tail -3 $R/echo_tests.log
python tools/bench_paths.py --workload echo --no-cpu-baseline > $R/paths_echo.json 2> $R/paths_echo.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/paths_echo.json').read().strip().splitlines()[-1])
print('echo ms_per_step', d['ms_per_step'], 'avg_launch_us', d['roofline'].get('avg_launch_us'), d['config'].get('erle_db_last_second_single_talk_lines'))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_echo -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --no-cpu-baseline --echo-seconds 3 > $R/stats_echo.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/hbm_echo_$c -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --steps 40 --no-cpu-baseline --no-e2e --echo-seconds 2 > $R/hbm_echo_$c.log 2>&1
  echo "echo $c rc=$?"
done
cd $GRAFT_REPO_ROOT
find $R/stats_echo -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/echo_kernel_stats.csv
head -4 $R/echo_kernel_stats.csv | cut -c1-200
python3 tools/hbm_summary.py $R > $R/hbm_traffic_raw_echo.json
python3 -c "
import json
d=json.load(open('$R/hbm_traffic_raw_echo.json'))
for k,v in d.get('echo',{}).items():
    if 'echo_bank' in k: print(k[:60], {c:(x['launches'], x['mean_KiB']) for c,x in v.items()})
"
find $R -mindepth 1 -maxdepth 1 -type d \( -name 'hbm_*' -o -name 'stats_*' \) -exec rm -rf {} +
