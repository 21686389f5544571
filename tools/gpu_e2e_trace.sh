# Timeline of the pipelined host path: copy and kernel records of a short bench.py run.  Output: gpurun_out/e2e/.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/e2e
rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 > $R/bench.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $R/bench.log | cut -c1-400
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/e2e/trace/*/*memory_copy_trace.csv')
rows = list(csv.DictReader(open(f[0])))
print(rows[0].keys())
big = [r for r in rows if 'HOST_TO_DEVICE' in r.get('Direction', '') or 'H2D' in r.get('Direction', '')]
big = sorted(big, key=lambda r: int(r['Start_Timestamp']))
sel = []
for r in big:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if d > 200000:
        sel.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
print(len(sel), 'large H2D copies')
for i in range(max(0, len(sel) - 30), len(sel)):
    s, e = sel[i]
    gap = s - sel[i - 1][1] if i else 0
    print('copy %3d  dur %.1f us  gap before %.1f us' % (i, (e - s)/1e3, gap/1e3))
PY
