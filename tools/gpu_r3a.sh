# Round 3, first GPU call: the four-lanes-per-channel V.29 kernel -- parity tests, A-B timing against the one-lane kernel,
# counters.  Output under gpurun_out/r3a/.
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_v29_gpu.py tests/test_modem_var_gpu.py tests/test_shim_modem_gpu.py tests/test_refstate_gpu.py tests/test_fax_front_end_gpu.py -m gpu -q -x > $R/pytest_v29.log 2>&1; echo "pytest rc=$?" >> $R/pytest_v29.log
tail -5 $R/pytest_v29.log
timeout 600 python -m pytest tests/test_full_size_gpu.py -m gpu -q -k "v29" > $R/pytest_full.log 2>&1; echo "pytest rc=$?" >> $R/pytest_full.log
tail -5 $R/pytest_full.log
for m in 1 4 8; do
  timeout 300 python tools/bench_paths.py --workload v29 --no-cpu-baseline --modem-mapping $m > $R/v29_map$m.json 2> $R/v29_map$m.err; echo "map $m rc=$?"
  python3 -c "import json;d=json.load(open('$R/v29_map$m.json'));print($m, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['min_launch_us'], d['config'])"
done
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/pmc$i -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload v29 --steps 40 --no-cpu-baseline --modem-mapping 4 > $R/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > gpurun_out/r3a/pmc_summary.txt
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r3a/pmc*/')):
    for f in glob.glob(d+'*/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if 'v29_' not in r['Kernel_Name']: continue
            k = r['Kernel_Name'][:48]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
        for k, v in acc.items():
            print(k, {a: round(b/n[(k, a)]) for a, b in v.items()}, 'launches', max(n.values()))
PY
cat gpurun_out/r3a/pmc_summary.txt
