set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 120 ./tools/probe > gpurun_out/probe.log 2>&1
timeout 300 python bench.py > gpurun_out/bench.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_r1 -name "*stats*" | head; head -5 gpurun_out/prof_r1/*/*kernel_stats.csv; 
tail -3 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; cat gpurun_out/probe.log; tail -3 gpurun_out/bench.log
