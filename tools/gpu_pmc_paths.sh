# Counters of one tools/bench_paths.py workload: bash tools/gpu_pmc_paths.sh <workload> [extra bench_paths flags...].  Output: gpurun_out/pmcp/.
cd $GRAFT_REPO_ROOT
W=$1; shift
R=$GRAFT_REPO_ROOT/gpurun_out/pmcp
rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/pmc$i -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $W --steps 30 --no-cpu-baseline "$@" > $R/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - "$W $*" <<'PY' > gpurun_out/pmcp/pmc_summary.txt
import csv, glob, collections, sys
print('#', sys.argv[1])
for d in sorted(glob.glob('gpurun_out/pmcp/pmc*/')):
    for f in glob.glob(d+'*/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
        for k, v in acc.items():
            cnt = max(n[(k, a)] for a in v)
            if cnt < 20: continue
            print(k, {a: round(b/n[(k, a)]) for a, b in v.items()}, 'launches', cnt)
PY
cat gpurun_out/pmcp/pmc_summary.txt
