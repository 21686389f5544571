#!/bin/bash
# round 6: where configs[2]'s tick goes with the cadence matcher on (kernel trace), and the super-tone bank by itself
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mixed -o mixed -- python $R/tools/bench_paths.py --workload mixed --no-cpu-baseline > $R/gpurun_out/r6_paths_mixed_prof.json 2> $R/gpurun_out/r6_prof_mixed.err
cd $R
find gpurun_out/prof_mixed -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r6_mixed_kernel_stats.csv
find gpurun_out/prof_mixed -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_overlap.py {} > gpurun_out/r6_mixed_trace_overlap.txt 2>&1
head -12 gpurun_out/r6_mixed_kernel_stats.csv | cut -c1-260
python -m pytest tests/test_tone_gpu.py -x -q -m gpu -k "streams_of_their_own" 2>&1 | tail -3
python tools/bench_paths.py --workload mixed --no-cpu-baseline --no-cadences > gpurun_out/r6_paths_mixed_nocad.json 2>/dev/null
python tools/bench_paths.py --workload mixed --no-cpu-baseline --one-launch > gpurun_out/r6_paths_mixed_one.json 2>/dev/null
python tools/bench_paths.py --workload supertone --no-cpu-baseline > gpurun_out/r6_paths_supertone.json 2>/dev/null
for f in mixed_prof mixed_nocad mixed_one supertone; do python - <<PY
import json
d=json.loads(open('gpurun_out/r6_paths_$f.json').read().strip().splitlines()[-1])
print('$f', d['ms_per_step'], d['roofline'].get('avg_launch_us'), d['roofline'].get('one_launch_us'), {k:v for k,v in d['config'].items() if k.endswith('_us')})
PY
done
rm -rf gpurun_out/prof_mixed
