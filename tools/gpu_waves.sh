# Secondary kernels: throughput against waves per SIMD (65 536 channels = one wave per SIMD, 262 144 = four).  Output: gpurun_out/waves/.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/waves
rm -rf $R; mkdir -p $R
for w in fsk mct sigtone v29_tx dtmf_tx; do
  for n in 65536 131072 262144; do
    timeout 200 python tools/bench_paths.py --workload $w --channels $n --steps 60 --no-cpu-baseline > $R/${w}_$n.json 2> $R/${w}_$n.err
    python3 -c "import json;d=json.load(open('$R/${w}_$n.json'));print('$w', $n, d['ms_per_step'], d['roofline']['avg_launch_us'], d['value'])"
  done
done
