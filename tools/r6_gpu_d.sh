#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mixed -o mixed -- python $R/tools/bench_paths.py --workload mixed --no-cpu-baseline > $R/gpurun_out/r6_paths_mixed_prof.json 2> $R/gpurun_out/r6_prof_mixed.err
cd $R
find gpurun_out/prof_mixed -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r6_mixed_kernel_stats.csv
find gpurun_out/prof_mixed -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/r6_mixed_kernel_trace.csv
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/r6_mixed_kernel_trace.csv')))
rows=[r for r in rows if 'tone_' in r['Kernel_Name'] or 'cadence' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last 6000 launches = the timed two_queues region (2000 ticks x 2 launches) + some before
tail=rows[-4000:]
byq=collections.defaultdict(list)
for r in tail: byq[(r['Queue_Id'], r['Kernel_Name'][:60])].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for k,v in byq.items():
    d=[e-s for s,e in v]; g=[v[i+1][0]-v[i][1] for i in range(len(v)-1)]
    per=[v[i+1][0]-v[i][0] for i in range(len(v)-1)]
    print(k, len(v), 'dur %.2f'%(sum(d)/len(d)/1e3), 'gap %.2f'%(sum(g)/len(g)/1e3), 'period %.2f'%(sum(per)/len(per)/1e3))
PY
rm -rf gpurun_out/prof_mixed gpurun_out/r6_mixed_kernel_trace.csv
