#!/bin/bash
# where the echo canceller's read excess comes from: FETCH_SIZE / WRITE_SIZE of its kernel with 4, 8 and 16 lanes per line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export GRAFT_REPO_ROOT=$PWD
R=$PWD/gpurun_out/r6e
mkdir -p $R
python -m pytest tests/test_fsk_gpu.py -x -q -m gpu -k off_frequency 2>&1 | tail -2
cd /tmp
for g in 4 8 16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/hbm_echo${g}_$c -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --steps 40 --no-cpu-baseline --no-e2e --echo-seconds 2 --echo-lanes $g > $R/hbm_echo${g}_$c.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python3 tools/hbm_summary.py $R > $R/hbm_echo_lanes.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r6e/hbm_echo_lanes.json'))
for w,ks in d.items():
    for k,v in ks.items():
        if 'echo_bank' in k and 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            print(w, k[:50], 'read MB %.1f write MB %.1f' % (v['FETCH_SIZE']['mean_KiB']*2*1024/1e6, v['WRITE_SIZE']['mean_KiB']*1024/1e6))
PY
find $R -mindepth 1 -maxdepth 1 -type d -name 'hbm_*' -exec rm -rf {} +
