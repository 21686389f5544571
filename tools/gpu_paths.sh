set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/paths
for w in mixed v29 v17 v27ter echo dtmf_tx fsk mct v29_tx awgn; do
  timeout 500 python tools/bench_paths.py --workload $w > gpurun_out/paths/$w.json 2> gpurun_out/paths/$w.err; echo "$w rc=$?"; tail -c 1500 gpurun_out/paths/$w.json; tail -3 gpurun_out/paths/$w.err
done
