# The modem banks of 131 072 channels (full-wave kernels): bench_paths lines and rocprofv3 kernel statistics of the same commands
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/modem_big
mkdir -p $R
export TMPDIR=/tmp
cd /tmp
for w in v29 v27ter v17; do
  timeout 200 python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --channels 131072 --no-cpu-baseline --steps 100 > $R/$w.json 2> $R/$w.err; echo "$w rc=$?"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_$w -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --channels 131072 --no-cpu-baseline --steps 100 > $R/stats_$w.log 2>&1
  find $R/stats_$w -name "*kernel_stats.csv" -exec cp {} $R/${w}_kernel_stats.csv \;
  head -3 $R/${w}_kernel_stats.csv | cut -c1-200
done
