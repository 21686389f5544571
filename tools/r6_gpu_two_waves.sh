#!/bin/bash
# VERDICT item 2's premise, measured: does a SECOND WAVE per SIMD hide what a lone wave of the four-lane receivers waits for?
# 16 384 channels x 4 lanes = 1 024 waves = one per SIMD; 32 768 channels = two per SIMD, same kernel, same code per wave.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for w in v29 v17; do
  for line in in_step contract; do
    for ch in 16384 32768 16384 32768; do
      python tools/bench_paths.py --workload $w --channels $ch --line $line --no-cpu-baseline --modem-mapping 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w $line channels=$ch launch_us=%.1f us_per_16384=%.1f kernel=%s data_mode=%s' % (r['avg_launch_us'], r['avg_launch_us']*16384.0/$ch, r['kernel'], d['config']['sampled_channels_in_data_mode_at_end']))"
    done
  done
done
