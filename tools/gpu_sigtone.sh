#!/bin/bash
# one gpurun call: the signalling tone tests, the bench line with its cpu baseline, its rocprofv3 kernel statistics
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/sigtone
mkdir -p $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sigtone_gpu.py tests/test_shim_sigtone_gpu.py -q 2>&1 | tail -5 > $R/pytest.log
timeout 300 python tools/bench_paths.py --workload sigtone > $R/paths_sigtone.json 2> $R/paths_sigtone.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload sigtone --no-cpu-baseline > $R/stats.log 2>&1
find $R/stats -name "*kernel_stats.csv" -exec cp {} $R/kernel_stats.csv \;
cat $R/pytest.log; cat $R/paths_sigtone.json; head -4 $R/kernel_stats.csv
