# The end-of-round refresh (run on the GPU box through gpurun): smoke, the GPU tests, the two bench.py lines with the
# rocprofv3 kernel statistics of the headline one, and every tools/bench_paths.py workload.  Lands under gpurun_out/final/.
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $R/paths
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $R/pytest_gpu.log
cd /tmp
timeout 400 python $GRAFT_REPO_ROOT/bench.py > $R/bench.json 2> $R/bench.err
timeout 400 python $GRAFT_REPO_ROOT/bench.py --workload echo > $R/bench_echo.json 2> $R/bench_echo.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-e2e > $R/bench_stats.log 2>&1
find $R/bench_stats -name "*kernel_stats.csv" -exec cp {} $R/bench_kernel_stats.csv \;
cd $GRAFT_REPO_ROOT
for w in mixed v29 v17 v27ter echo dtmf_tx fsk mct sigtone supertone fax_rx v29_tx awgn; do
  timeout 500 python tools/bench_paths.py --workload $w > $R/paths/$w.json 2> $R/paths/$w.err; echo "$w rc=$?"
done
cat $R/smoke.log | tail -2; cat $R/pytest_gpu.log | tail -3; cat $R/bench.json; head -3 $R/bench_kernel_stats.csv
