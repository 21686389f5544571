#!/bin/bash
# Round 6, VERDICT item 5: the two levers left on the headline kernel, measured on ONE box.
#   (a) the v_pk_fma_f32 BUILD VARIANT (tools/build_variant.sh fma worktree EXTRA=-DSPG_TONE_FMA -> tools/experiments/libspangpu_fma.so):
#       every tone parity test, the side-1 programs, the full-size DTMF bank and the soak under it (digits / hits / codes exact,
#       floats within 1e-5 of the vector's largest: tests/test_tone_gpu.py same_f32), and bench.py beside the product library's;
#   (b) frames per launch 1 / 2 / 3 (bench.py's frames_per_launch side key), product library.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD/gpurun_out/r6
mkdir -p $R
FMA=$PWD/tools/experiments/libspangpu_fma.so
python bench.py --no-cpu-baseline --no-e2e > $R/bench_exact.json 2> $R/bench_exact.err
SPANGPU_LIB=$FMA python bench.py --no-cpu-baseline --no-e2e > $R/bench_fma.json 2> $R/bench_fma.err
python bench.py --no-cpu-baseline --no-e2e > $R/bench_exact2.json 2>> $R/bench_exact.err
SPANGPU_LIB=$FMA python bench.py --no-cpu-baseline --no-e2e > $R/bench_fma2.json 2>> $R/bench_fma.err
python - <<'PY' | tee gpurun_out/r6/tone_levers.log
import json
def load(f):
    return json.loads(open('gpurun_out/r6/%s.json' % f).read().strip().splitlines()[-1])
for name in ("bench_exact", "bench_fma", "bench_exact2", "bench_fma2"):
    d = load(name)
    lb = d.get("large_bank") or {}
    print("%-13s 65536 ch: %.3f us/launch (frac %.4f)" % (name, d["roofline"]["avg_launch_us"], d["roofline"]["frac"]),
          " | ".join("%s: q1 %.1f us (%.3f) q2 %.1f us (%.3f)" % (k, v["queues_1"]["us_per_step"], v["queues_1"]["roofline_frac"], v["queues_2"]["us_per_step"], v["queues_2"]["roofline_frac"]) for k, v in lb.items() if "queues_1" in v))
d = load("bench_exact")
for k, v in (d.get("frames_per_launch") or {}).items():
    print("frames_per_launch %s:" % k, json.dumps(v))
d = load("bench_exact2")
for k, v in (d.get("frames_per_launch") or {}).items():
    print("frames_per_launch %s (second run):" % k, json.dumps(v))
PY
SPANGPU_LIB=$FMA SPANGPU_TEST_FMA=1 timeout 1500 python -m pytest tests/test_tone_gpu.py tests/test_mitel.py tests/test_bell_mf_side1.py tests/test_r2_mf_side1.py tests/test_soak_gpu.py tests/test_full_size_gpu.py -q -m gpu -k "not v29 and not v17 and not v27 and not echo and not modem and not fsk and not mct" > $R/pytest_fma.log 2>&1
tail -15 $R/pytest_fma.log
