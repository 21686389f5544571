# Run on the GPU box: instruction-mix counters for the echo kernel (kernel-trace only, one --pmc pass per group).
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/pmc_echo
mkdir -p $R
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$tag -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --no-cpu-baseline --steps 20 > $R/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_echo/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "echo_bank_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "launches", len(v), "mean", sum(v)/len(v))
PY
