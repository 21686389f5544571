set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|FETCH_SIZE\|WRITE_SIZE" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/counters.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc$i -- $GRAFT_REPO_ROOT/tools/probe tone > $GRAFT_REPO_ROOT/gpurun_out/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
ls gpurun_out/pmc1/*/ | head
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc*/')):
    for f in glob.glob(d+'*/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = (r['Kernel_Name'][:40], r.get('Grid_Size'))
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
        for k, v in acc.items():
            print(d, k, {a: b for a, b in v.items()})
PY
