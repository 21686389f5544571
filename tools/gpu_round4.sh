# Round 4 measurements (run on the GPU box through gpurun): bash tools/gpu_round4.sh <what>
# Everything lands under gpurun_out/r4/; the summaries that are judged are copied into profiles/ (r4_*).
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r4
mkdir -p $R
export TMPDIR=/tmp
case "$1" in
probe)
  echo "### tools/probe2 r4 (eager)" > $R/probe.log
  timeout 300 ./tools/probe2 r4 >> $R/probe.log 2>&1
  echo "### tools/probe2 r4 graph" >> $R/probe.log
  timeout 300 ./tools/probe2 r4 graph >> $R/probe.log 2>&1
  grep -v "^      \|loader waves" $R/probe.log | tail -60
  timeout 900 python -m pytest tests/test_tone_gpu.py tests/test_mitel.py tests/test_feed_gpu.py -m gpu -q -x > $R/pytest_tone.log 2>&1; echo "pytest rc=$?" >> $R/pytest_tone.log
  tail -5 $R/pytest_tone.log
  cd /tmp; timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e > $R/bench_quick.json 2> $R/bench_quick.err; tail -c 900 $R/bench_quick.json
  ;;
bench)
  cd /tmp
  timeout 600 python $GRAFT_REPO_ROOT/bench.py > $R/bench.json 2> $R/bench.err
  timeout 600 python $GRAFT_REPO_ROOT/bench.py --workload echo > $R/bench_echo.json 2> $R/bench_echo.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-e2e --no-paths > $R/bench_stats.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/bench_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-paths > $R/bench_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/bench_write -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-paths > $R/bench_write.log 2>&1
  for w in mixed v29 v17 v27ter; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${w}_stats -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --no-cpu-baseline --no-e2e > $R/${w}_stats.log 2>&1
    cp $(ls $R/${w}_stats/*/*kernel_stats.csv | head -1) $R/${w}_kernel_stats.csv
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/echo_stats -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --no-cpu-baseline --no-e2e --echo-seconds 3 > $R/echo_stats.log 2>&1
  cp $(ls $R/echo_stats/*/*kernel_stats.csv | head -1) $R/echo_kernel_stats.csv
  cp $(ls $R/bench_stats/*/*kernel_stats.csv | head -1) $R/bench_kernel_stats.csv
  cd $GRAFT_REPO_ROOT
  python3 - <<'PY'
import csv, glob, collections, json
R = "gpurun_out/r4"
out = {}
for name in ("bench_fetch", "bench_write"):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (R, name)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "tone_fast_kernel" in k or "tone_bank_kernel" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(name, {})[k] = {c: {"launches": len(x), "mean": sum(x)/len(x)} for c, x in v.items()}
json.dump(out, open(R + "/counters_raw.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
  rm -rf $R/bench_stats $R/bench_fetch $R/bench_write $R/*_stats
  for w in mixed supertone fsk mct sigtone dtmf_tx v29_tx awgn v17 v27ter; do
    timeout 300 python tools/bench_paths.py --workload $w > $R/paths_$w.json 2> $R/paths_$w.err; echo "$w rc=$?"
  done
  tail -c 600 $R/bench.json
  ;;
valu)
  bash tools/gpu_valu.sh > $R/valu.log 2>&1
  cp gpurun_out/valu/valu_counters.json $R/ 2>/dev/null
  tail -30 $R/valu.log
  ;;
echo)
  timeout 900 python -m pytest tests/test_echo_gpu.py tests/test_shim_echo_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -k "echo" > $R/pytest_echo.log 2>&1; echo "pytest rc=$?" >> $R/pytest_echo.log
  tail -5 $R/pytest_echo.log
  cd /tmp
  timeout 300 python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --no-cpu-baseline --no-e2e --echo-seconds 3 > $R/echo_quick.json 2> $R/echo_quick.err; tail -c 1500 $R/echo_quick.json
  ;;
echo_pmc)
  cd /tmp
  SETS=${ECHO_PMC_SETS:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY|SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"}
  IFS='|' read -ra SETLIST <<< "$SETS"
  for set in "${SETLIST[@]}"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/pmc_$tag -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --steps 30 --no-cpu-baseline --no-e2e --echo-seconds 2 > $R/pmc_$tag.log 2>&1
  done
  cd $GRAFT_REPO_ROOT
  python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r4/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "echo_bank_kernel" in k or "echo_pair_kernel" in k:
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v)/len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("gpurun_out/r4/echo_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
  find $R -mindepth 1 -maxdepth 1 -type d -name 'pmc_*' -exec rm -rf {} +
  ;;
modem)
  timeout 1200 python -m pytest tests/test_v29_gpu.py tests/test_v17_gpu.py tests/test_v27ter_gpu.py tests/test_modem_qam_gpu.py tests/test_modem_var_gpu.py tests/test_full_size_gpu.py tests/test_feed_gpu.py -m gpu -q -x -k "not echo" > $R/pytest_modem.log 2>&1; echo "pytest rc=$?" >> $R/pytest_modem.log
  tail -3 $R/pytest_modem.log
  cd /tmp
  for w in v29 v17 v27ter; do
    timeout 300 python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --no-cpu-baseline --no-e2e > $R/${w}_quick.json 2> $R/${w}_quick.err
    grep -o '"avg_launch_us": [0-9.]*' $R/${w}_quick.json | head -1
    timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/pmc_$w -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --steps 40 --warmup 110 --no-cpu-baseline --no-e2e > $R/pmc_$w.log 2>&1
  done
  cd $GRAFT_REPO_ROOT
  python3 - <<'PY'
import csv, glob, collections, json
out = {}
for w in ("v29", "v17", "v27ter"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r4/pmc_%s/*/*counter_collection.csv" % w):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "quad_kernel" in k:
                acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[w] = {k: {c: sum(v)/len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("gpurun_out/r4/modem_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
  find $R -mindepth 1 -maxdepth 1 -type d -name 'pmc_*' -exec rm -rf {} +
  ;;
echo_occ)
  # the echo kernel at 1, 2, 3, 4, 6, 8, 9, 12 waves per SIMD (four lanes per channel, three waves a SIMD resident)
  cd /tmp
  for n in 16384 32768 49152 65536 98304 131072 147456 196608; do
    timeout 200 python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --channels $n --echo-lanes 4 --no-cpu-baseline --no-e2e --echo-seconds 2 > $R/echo_occ_$n.json 2> $R/echo_occ_$n.err
    echo "$n $(grep -o '"avg_launch_us": [0-9.]*' $R/echo_occ_$n.json | head -1)" | tee -a $R/echo_occ.log
  done
  ;;
echo_lanes)
  # where the lane mappings cross over, after the round-4 kernel work
  cd /tmp
  for n in 4096 8192 16384 32768; do
    for g in 4 8 16; do
      timeout 200 python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --channels $n --echo-lanes $g --no-cpu-baseline --no-e2e --echo-seconds 2 > $R/echo_lanes_${n}_$g.json 2> $R/echo_lanes_${n}_$g.err
      echo "$n lanes $g $(grep -o '"avg_launch_us": [0-9.]*' $R/echo_lanes_${n}_$g.json | head -1)" | tee -a $R/echo_lanes.log
    done
  done
  ;;
sched)
  # A-B of the maximum-ILP scheduler: every workload with the product library and with tools/experiments/libspangpu_ilp.so
  cd /tmp
  for lib in ${SCHED_LIBS:-std ilp}; do
    if [ $lib = ilp ]; then export SPANGPU_LIB=$GRAFT_REPO_ROOT/tools/experiments/${SCHED_LIB:-libspangpu_ilp.so}; else unset SPANGPU_LIB; fi
    timeout 200 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-paths > $R/sched_${lib}_dtmf.json 2>/dev/null
    echo "$lib dtmf $(grep -o '"avg_launch_us": [0-9.]*' $R/sched_${lib}_dtmf.json | head -1)" | tee -a $R/sched.log
    for w in ${SCHED_W:-mixed supertone fsk mct sigtone dtmf_tx v29_tx awgn echo v29 v17 v27ter}; do
      timeout 200 python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --no-cpu-baseline --no-e2e --echo-seconds 2 > $R/sched_${lib}_$w.json 2>/dev/null
      echo "$lib $w $(grep -o '"avg_launch_us": [0-9.]*' $R/sched_${lib}_$w.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $R/sched_${lib}_$w.json | head -1)" | tee -a $R/sched.log
    done
  done
  unset SPANGPU_LIB
  ;;
tests)
  python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
  timeout 1500 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
  tail -4 $R/pytest_gpu.log; tail -2 $R/smoke.log
  ;;
esac
