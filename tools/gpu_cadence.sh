# super-tone cadences on the device: the bench_paths line and the kernel statistics of the same command
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/cadence
mkdir -p $R
export TMPDIR=/tmp
timeout 300 python tools/bench_paths.py --workload supertone > $R/supertone.json 2> $R/supertone.err; echo "rc=$?"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload supertone --no-cpu-baseline > $R/stats.log 2>&1
find $R/stats -name "*kernel_stats.csv" -exec cp {} $R/kernel_stats.csv \;
cat $R/supertone.json; tail -3 $R/supertone.err; head -6 $R/kernel_stats.csv | cut -c1-250
