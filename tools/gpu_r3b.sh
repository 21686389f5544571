# Round 3: V.29 quad kernel iteration -- parity tests, timing, counters (mapping given as $1, default 4).  Output: gpurun_out/r3b/.
set -x
cd $GRAFT_REPO_ROOT
M=${1:-4}
W=${2:-v29}
R=$GRAFT_REPO_ROOT/gpurun_out/r3b
rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_${W}_gpu.py tests/test_modem_var_gpu.py tests/test_shim_modem_gpu.py tests/test_refstate_gpu.py -m gpu -q -x > $R/pytest_$W.log 2>&1; echo "pytest rc=$?" >> $R/pytest_$W.log
tail -3 $R/pytest_$W.log
timeout 600 python -m pytest tests/test_full_size_gpu.py -m gpu -q -k "$W" > $R/pytest_full.log 2>&1; echo "pytest rc=$?" >> $R/pytest_full.log
tail -3 $R/pytest_full.log
for m in 1 $M; do
  timeout 300 python tools/bench_paths.py --workload $W --no-cpu-baseline --modem-mapping $m > $R/${W}_map$m.json 2> $R/${W}_map$m.err; echo "map $m rc=$?"
  python3 -c "import json;d=json.load(open('$R/${W}_map$m.json'));print($m, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['min_launch_us'], d['roofline']['max_launch_us'])"
done
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/pmc$i -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $W --steps 40 --no-cpu-baseline --modem-mapping $M > $R/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > gpurun_out/r3b/pmc_summary.txt
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r3b/pmc*/')):
    for f in glob.glob(d+'*/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if '_quad_kernel' not in r['Kernel_Name'] and '_bank_kernel' not in r['Kernel_Name']: continue
            if 'tx_bank' in r['Kernel_Name'] or 'awgn' in r['Kernel_Name']: continue
            k = r['Kernel_Name'][:48]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
        for k, v in acc.items():
            print(k, {a: round(b/n[(k, a)]) for a, b in v.items()}, 'launches', max(n.values()))
            w = v.get('SQ_WAVES', 0)/max(1, n[(k, 'SQ_WAVES')])
            if w:
                print('  per wave per sample:', {a: round(b/n[(k, a)]/w/160, 1) for a, b in v.items()})
PY
cat gpurun_out/r3b/pmc_summary.txt
