cd $GRAFT_REPO_ROOT
for n in 32768 65536 131072; do for L in 8 4 2; do
python tools/bench_paths.py --workload echo --channels $n --no-cpu-baseline --echo-lanes $L 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('channels $n lanes $L: ms/step %.4f kernel_us %.1f' % (d['ms_per_step'], d['roofline']['avg_launch_us']))"
done; done
