# Round 4 measurements (run on the GPU box through gpurun): bash tools/gpu_round4.sh <what>
# Everything lands under gpurun_out/r4/; the summaries that are judged are copied into profiles/ (r4_*).
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r4
mkdir -p $R
export TMPDIR=/tmp
case "$1" in
probe)
  echo "### tools/probe2 r4 (eager)" > $R/probe.log
  timeout 300 ./tools/probe2 r4 >> $R/probe.log 2>&1
  echo "### tools/probe2 r4 graph" >> $R/probe.log
  timeout 300 ./tools/probe2 r4 graph >> $R/probe.log 2>&1
  grep -v "^      \|loader waves" $R/probe.log | tail -60
  timeout 900 python -m pytest tests/test_tone_gpu.py tests/test_mitel.py tests/test_feed_gpu.py -m gpu -q -x > $R/pytest_tone.log 2>&1; echo "pytest rc=$?" >> $R/pytest_tone.log
  tail -5 $R/pytest_tone.log
  cd /tmp; timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e > $R/bench_quick.json 2> $R/bench_quick.err; tail -c 900 $R/bench_quick.json
  ;;
tests)
  python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
  timeout 1500 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
  tail -4 $R/pytest_gpu.log; tail -2 $R/smoke.log
  ;;
esac
