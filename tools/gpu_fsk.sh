# FSK-family kernels over two waves: parity tests, then A-B timing.  Output: gpurun_out/fsk/.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/fsk
rm -rf $R; mkdir -p $R
timeout 900 python -m pytest ${TESTS:-tests/test_fsk_gpu.py tests/test_shim_fsk_gpu.py tests/test_fax_front_end_gpu.py} -m gpu -q -x > $R/pytest.log 2>&1; echo "pytest rc=$?" >> $R/pytest.log
tail -5 $R/pytest.log
for w in ${WORKLOADS:-fsk}; do
  for v in 1 2; do
    timeout 200 python tools/bench_paths.py --workload $w --steps 60 --no-cpu-baseline --fsk-waves $v > $R/${w}_w$v.json 2> $R/${w}_w$v.err
    python3 -c "import json;d=json.load(open('$R/${w}_w$v.json'));print('$w waves', $v, d['ms_per_step'], d['roofline']['avg_launch_us'], d['value'])"
  done
done
