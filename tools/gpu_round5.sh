# Round 5 measurements (run on the GPU box through gpurun): bash tools/gpu_round5.sh <what> [...]
# Everything lands under gpurun_out/r5/; the summaries that are judged are copied into profiles/ (r5_*).
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r5
mkdir -p $R
export TMPDIR=/tmp
for what in "$@"; do
case "$what" in
modem_quick)
  # the three receivers at 16 384 x 160 by tools/bench_paths.py's clock (MQ_LIBS: alternative builds of the library to put beside it)
  for rep in 1 2; do
  for lib in product ${MQ_LIBS}; do
    for w in ${MQ_W:-v29 v17 v27ter}; do
      if [ $lib = product ]; then unset SPANGPU_LIB; else export SPANGPU_LIB=$GRAFT_REPO_ROOT/tools/experiments/libspangpu_$lib.so; fi
      timeout 200 python tools/bench_paths.py --workload $w --no-cpu-baseline --no-e2e ${MQ_ARGS} > $R/mq_${w}_${lib}_$rep.json 2> $R/mq_${w}_${lib}_$rep.err
      echo "$w $lib rep $rep: $(grep -o '"avg_launch_us": [0-9.]*' $R/mq_${w}_${lib}_$rep.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $R/mq_${w}_${lib}_$rep.json | head -1)"
    done
  done
  done
  unset SPANGPU_LIB
  ;;
modem_tests)
  timeout 900 python -m pytest tests/test_v29_gpu.py tests/test_v17_gpu.py tests/test_v27ter_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > $R/pytest_modem.log 2>&1; echo "pytest rc=$?"; tail -5 $R/pytest_modem.log
  ;;
modem_counters)
  # SQ counters of a receiver's kernel, the product build beside alternative builds (MQ_LIBS), data mode launches only
  cd /tmp
  for lib in product ${MQ_LIBS}; do
    for w in ${MQ_W:-v29}; do
      if [ $lib = product ]; then unset SPANGPU_LIB; else export SPANGPU_LIB=$GRAFT_REPO_ROOT/tools/experiments/libspangpu_$lib.so; fi
      rm -rf /tmp/mc_$lib_$w
      timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/mc_${lib}_$w -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --steps 40 --warmup 110 --no-cpu-baseline --no-e2e > $R/mc_${lib}_$w.log 2>&1
      python3 - /tmp/mc_${lib}_$w $lib $w <<'PY'
import csv, glob, collections, sys
d, lib, w = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "quad_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, m in acc.items():
    last = {c: sum(v[-40:])/len(v[-40:]) for c, v in m.items()}        # the timed region: data mode
    wv = last.get("SQ_WAVES", 1) or 1
    print(lib, w, "per wave: VALU %.0f SALU %.0f LDS %.0f  wave-cycles %.0f  active %.3f wait %.3f" % (last["SQ_INSTS_VALU"]/wv, last["SQ_INSTS_SALU"]/wv, last["SQ_INSTS_LDS"]/wv,
          4*last["SQ_WAVE_CYCLES"]/wv, last["SQ_ACTIVE_INST_ANY"]/last["SQ_WAVE_CYCLES"], last["SQ_WAIT_ANY"]/last["SQ_WAVE_CYCLES"]))
PY
      rm -rf /tmp/mc_${lib}_$w
    done
  done
  unset SPANGPU_LIB
  cd $GRAFT_REPO_ROOT
  ;;
modem_final_counters)
  # the three receivers' kernels after the round's second half: SQ counters (valu_counters.json) and the HBM bytes (hbm_traffic_raw.json)
  VALU_W="none" bash tools/gpu_valu.sh > $R/valu.log 2>&1
  cp gpurun_out/valu/valu_counters.json $R/ 2>/dev/null
  HBM_W="v29 v17 v27ter" HBM_NO_DTMF=1 bash tools/gpu_round5.sh hbm > $R/hbm.log 2>&1
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5/valu_counters.json"))["workloads"]
for k in ("v29", "v17", "v27ter"):
    v = d.get(k, {})
    print(k, v.get("kernel", "")[:50], v.get("valu_insts_per_wave_sample"), v.get("active_frac"), v.get("wait_frac"), v.get("source"))
PY
  tail -30 $R/hbm.log | head -40
  ;;
mq)
  # sub-banks on hardware queues of their own (tools/probe8.hip): the product kernel, K = 1, 2, 4, 8
  { echo "### tools/probe8 65536"; timeout 400 ./tools/probe8 65536 2000 5; } > $R/probe_mq.log 2>&1
  { echo "### tools/probe8 131072"; timeout 300 ./tools/probe8 131072 1000 3; } >> $R/probe_mq.log 2>&1
  cat $R/probe_mq.log
  ;;
mq_trace)
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/mq_trace -- $GRAFT_REPO_ROOT/tools/probe8 65536 200 1 trace > $R/mq_trace.log 2>&1
  cd $GRAFT_REPO_ROOT
  python3 tools/trace_overlap.py $(ls $R/mq_trace/*/*kernel_trace.csv | head -1) > $R/mq_trace_overlap.txt 2>&1
  cat $R/mq_trace_overlap.txt
  rm -rf $R/mq_trace
  ;;
hbm)
  # FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of the dominant kernel of every BASELINE path
  cd /tmp
  for w in ${HBM_W:-mixed v29 echo v17 v27ter}; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/hbm_${w}_$c -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --steps 40 --no-cpu-baseline --no-e2e --echo-seconds 2 > $R/hbm_${w}_$c.log 2>&1
      echo "$w $c rc=$?"
    done
  done
  [ -n "$HBM_NO_DTMF" ] || timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/hbm_dtmf_FETCH_SIZE -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-paths > $R/hbm_dtmf_FETCH_SIZE.log 2>&1
  [ -n "$HBM_NO_DTMF" ] || timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/hbm_dtmf_WRITE_SIZE -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-paths > $R/hbm_dtmf_WRITE_SIZE.log 2>&1
  cd $GRAFT_REPO_ROOT
  python3 tools/hbm_summary.py $R > $R/hbm_traffic_raw.json
  cat $R/hbm_traffic_raw.json
  find $R -mindepth 1 -maxdepth 1 -type d -name 'hbm_*' -exec rm -rf {} +
  ;;
calib)
  # what FETCH_SIZE / WRITE_SIZE report for 1 GiB moved in the access patterns of this repository's kernels (tools/probe_fetch.hip)
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/calib_$c -- $GRAFT_REPO_ROOT/tools/probe_fetch > $R/calib_$c.log 2>&1
  done
  cd $GRAFT_REPO_ROOT
  python3 - <<'PY' > $R/hbm_calibration.json
import csv, glob, collections, json
out = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/r5/calib_%s/*/*counter_collection.csv" % c):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(k, {})[c] = {"launches": len(v), "mean_KiB": sum(v)/len(v), "bytes_moved": 1 << 30,
                                          "reported_over_moved": sum(v)/len(v)*1024.0/(1 << 30)}
print(json.dumps(out, indent=1))
PY
  cat $R/hbm_calibration.json
  rm -rf $R/calib_FETCH_SIZE $R/calib_WRITE_SIZE
  ;;
mq_big)
  { echo "### tools/probe8 1048576"; timeout 400 ./tools/probe8 1048576 200 3; } > $R/probe_mq_1M.log 2>&1
  { echo "### tools/probe8 262144"; timeout 300 ./tools/probe8 262144 500 3; } >> $R/probe_mq_1M.log 2>&1
  cat $R/probe_mq_1M.log
  ;;
mixed_streams)
  cd /tmp
  for v in "" "--one-launch" "--separate-launches"; do
    echo "### mixed $v"
    timeout 200 python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload mixed --no-cpu-baseline --no-e2e $v > $R/mixed_try.json 2> $R/mixed_try.err
    grep -o '"avg_launch_us": [0-9.]*\|"ms_per_step": [0-9.]*' $R/mixed_try.json | head -3; tail -2 $R/mixed_try.err
  done
  ;;
tests)
  python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
  timeout 1500 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
  tail -4 $R/pytest_gpu.log; tail -2 $R/smoke.log
  ;;
tone_tests)
  timeout 900 python -m pytest tests/test_tone_gpu.py tests/test_mitel.py tests/test_feed_gpu.py tests/test_soak_gpu.py tests/test_shard_gpu.py tests/test_cadence_gpu.py -m gpu -q -x > $R/pytest_tone.log 2>&1; echo "pytest rc=$?" >> $R/pytest_tone.log
  tail -5 $R/pytest_tone.log
  ;;
echo)
  timeout 1200 python -m pytest tests/test_echo_gpu.py tests/test_shim_echo_gpu.py tests/test_refstate_gpu.py -m gpu -q -x -k "echo" > $R/pytest_echo.log 2>&1; echo "pytest rc=$?" >> $R/pytest_echo.log
  tail -5 $R/pytest_echo.log
  cd /tmp
  timeout 300 python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --no-cpu-baseline --no-e2e --echo-seconds 3 > $R/echo_quick.json 2> $R/echo_quick.err; grep -o '"avg_launch_us": [0-9.]*\|"ms_per_step": [0-9.]*' $R/echo_quick.json | head -3; tail -2 $R/echo_quick.err
  cd $GRAFT_REPO_ROOT
  ;;
misc_tests)
  timeout 1200 python -m pytest tests/test_shard_gpu.py tests/test_shim_prims_gpu.py tests/test_prim_gpu.py tests/test_feed_gpu.py tests/test_bench_spawn_gpu.py -m gpu -q -x > $R/pytest_misc.log 2>&1; echo "pytest rc=$?" >> $R/pytest_misc.log
  grep -v "^E2026\|^W2026" $R/pytest_misc.log | tail -30
  ;;
final_tests)
  python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
  timeout 1700 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
  grep -v "^E2026\|^W2026" $R/pytest_gpu.log | tail -6; tail -2 $R/smoke.log
  ;;
final_bench)
  cd /tmp
  timeout 900 python $GRAFT_REPO_ROOT/bench.py > $R/bench.json 2> $R/bench.err; tail -2 $R/bench.err
  timeout 600 python $GRAFT_REPO_ROOT/bench.py --workload echo > $R/bench_echo.json 2> $R/bench_echo.err; tail -2 $R/bench_echo.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-e2e --no-paths > $R/bench_stats.log 2>&1
  cp $(ls $R/bench_stats/*/*kernel_stats.csv | head -1) $R/bench_kernel_stats.csv
  for w in mixed v29 v17 v27ter; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${w}_stats -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload $w --no-cpu-baseline --no-e2e > $R/${w}_stats.log 2>&1
    cp $(ls $R/${w}_stats/*/*kernel_stats.csv | head -1) $R/${w}_kernel_stats.csv
  done
  # the mixed tick's kernels side by side on their streams (concurrent residency)
  python3 $GRAFT_REPO_ROOT/tools/trace_overlap.py $(ls $R/mixed_stats/*/*kernel_trace.csv | head -1) "tone_fast_kernel" > $R/mixed_trace_overlap.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/echo_stats -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --no-cpu-baseline --no-e2e --echo-seconds 3 > $R/echo_stats.log 2>&1
  cp $(ls $R/echo_stats/*/*kernel_stats.csv | head -1) $R/echo_kernel_stats.csv
  rm -rf $R/bench_stats $R/*_stats
  cd $GRAFT_REPO_ROOT
  python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("traffic"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
print("large", json.dumps(d.get("large_bank")))
for k, v in (d.get("paths") or {}).items():
    if isinstance(v, dict):
        r = v.get("roofline") or {}
        print(k, v.get("ms_per_step"), r.get("avg_launch_us"), r.get("frac"), r.get("one_launch_us"), r.get("traffic"), v.get("error"))
        c = v.get("cpu_baseline") or {}
        print("   cpu", c.get("value"), c.get("cores"), c.get("single_core"), c.get("scaling_efficiency"), (c.get("spot_check") or {}) if not isinstance(c.get("spot_check"), dict) else list(c.get("spot_check").items())[:3])
PY
  cat $R/mixed_trace_overlap.txt
  ;;
final_counters)
  VALU_W="echo mixed" bash tools/gpu_valu.sh > $R/valu.log 2>&1
  cp gpurun_out/valu/valu_counters.json $R/ 2>/dev/null
  bash tools/gpu_round5.sh hbm > $R/hbm.log 2>&1
  bash tools/gpu_round4.sh echo_pmc > $R/echo_pmc.log 2>&1; cp gpurun_out/r4/echo_pmc.json $R/echo_pmc.json 2>/dev/null
  tail -40 $R/hbm.log | head -5
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5/valu_counters.json"))["workloads"]
for k in ("dtmf", "mixed", "v29", "echo"):
    v = d.get(k, {})
    print(k, v.get("kernel", "")[:60], v.get("valu_insts_per_wave_sample"), v.get("active_frac"), v.get("wait_frac"), v.get("source"))
PY
  ;;
issue)
  timeout 300 ./tools/probe_issue > $R/probe_issue.log 2>&1; cat $R/probe_issue.log
  ;;
sched_variants)
  timeout 1700 python -m pytest tests/test_sched_variants_gpu.py -m gpu -q > $R/pytest_sched.log 2>&1; echo "pytest rc=$?" >> $R/pytest_sched.log
  grep -v "^E2026\|^W2026" $R/pytest_sched.log | tail -8
  ;;
mixed_trace)
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/mixed_tr -- python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload mixed --no-cpu-baseline --no-e2e > $R/mixed_tr.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/trace_overlap.py $(ls $R/mixed_tr/*/*kernel_trace.csv | head -1) "tone_fast_kernel" --merge > $R/mixed_trace_overlap.txt 2>&1
  cat $R/mixed_trace_overlap.txt
  rm -rf $R/mixed_tr
  cd $GRAFT_REPO_ROOT
  ;;
tile)
  cd $GRAFT_REPO_ROOT
  { echo "### tools/probe2 tile: the state in wave-major tiles of 16-byte lane vectors (the product); abl1024 = no state traffic"; timeout 600 ./tools/probe2 tile;
    echo "### tools/probe2_rows tile: the same kernel with the state as word-major rows (the tree before the tiles, built by hand for this A-B); abl1049600 there = abl1024";
    [ -x ./tools/probe2_rows ] && timeout 600 ./tools/probe2_rows tile;
    echo "### again: tiles"; timeout 600 ./tools/probe2 tile; } > $R/probe_tile_real.log 2>&1
  grep -v "^      \|loader waves\|last launch" $R/probe_tile_real.log | grep "abl0 \|###\|----" | cut -c1-118
  ;;
probe_geo)
  { echo "### tools/probe8 65536 (with the one-launch geometry variants)"; timeout 400 ./tools/probe8 65536 2000 5; } > $R/probe_mq_geo.log 2>&1
  grep "K=1" $R/probe_mq_geo.log
  ;;
echo_wide)
  timeout 1500 python -m pytest tests/test_echo_gpu.py tests/test_shim_echo_gpu.py tests/test_refstate_gpu.py tests/test_feed_gpu.py tests/test_shard_gpu.py tests/test_full_size_gpu.py tests/test_nccl_gpu.py tests/test_soak_gpu.py -m gpu -q -x > $R/pytest_echo_wide.log 2>&1; echo "pytest rc=$?" >> $R/pytest_echo_wide.log
  grep -v "^E2026\|^W2026" $R/pytest_echo_wide.log | tail -6
  cd /tmp
  for n in 8192 32768; do
    timeout 200 python $GRAFT_REPO_ROOT/tools/bench_paths.py --workload echo --channels $n --no-cpu-baseline --no-e2e --echo-seconds 3 > $R/echo_n.json 2> $R/echo_n.err
    echo "$n $(grep -o '"avg_launch_us": [0-9.]*' $R/echo_n.json | head -1)"
  done
  cd $GRAFT_REPO_ROOT
  ;;
echo_counters)
  VALU_W="echo" VALU_MODEMS="" bash tools/gpu_valu.sh > $R/valu.log 2>&1
  cp gpurun_out/valu/valu_counters.json $R/ 2>/dev/null
  HBM_W="echo" bash tools/gpu_round5.sh hbm > $R/hbm.log 2>&1
  bash tools/gpu_round4.sh echo_pmc > $R/echo_pmc.log 2>&1; cp gpurun_out/r4/echo_pmc.json $R/echo_pmc.json 2>/dev/null
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5/valu_counters.json"))["workloads"]
v = d.get("echo", {})
print("echo", v.get("kernel", "")[:60], v.get("valu_insts_per_wave_sample"), v.get("active_frac"), v.get("wait_frac"), v.get("source"))
PY
  ;;
bench_quick)
  cd /tmp; timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-paths > $R/bench_quick.json 2> $R/bench_quick.err; tail -c 1500 $R/bench_quick.json; tail -3 $R/bench_quick.err
  ;;
bench_nocpu)
  cd /tmp; timeout 600 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e > $R/bench_nocpu.json 2> $R/bench_nocpu.err; tail -3 $R/bench_nocpu.err
  python3 - <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/r5/bench_nocpu.json").read().strip().splitlines()[-1])
print("headline", d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
print("large", json.dumps(d.get("large_bank")))
for k, v in (d.get("paths") or {}).items():
    if isinstance(v, dict):
        print(k, v.get("ms_per_step"), (v.get("roofline") or {}).get("avg_launch_us"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("one_launch_us"), v.get("error"))
PY
  ;;
queues)
  cd /tmp
  for n in 131072 262144; do for q in 1 2; do
    timeout 200 python $GRAFT_REPO_ROOT/bench.py --channels $n --queues $q --distinct-frames 30 --no-cpu-baseline --no-e2e --no-paths > $R/bench_q.json 2> $R/bench_q.err
    echo "$n queues $q: $(grep -o '"avg_launch_us": [0-9.]*' $R/bench_q.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $R/bench_q.json | head -1)"; tail -1 $R/bench_q.err
  done; done
  ;;
*)
  echo "unknown stage $what"
  ;;
esac
done
