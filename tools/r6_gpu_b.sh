#!/bin/bash
# round 6, second GPU call: the new parity tests and what each condition of SURVEY 8(d)-4's lines costs the V.29 bank
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_modem_offset_gpu.py tests/test_modemtx_gpu.py -x -q -m gpu -k "offset or line" > gpurun_out/r6_tests_b.log 2>&1; tail -3 gpurun_out/r6_tests_b.log
python tools/bench_paths.py --workload v29 > gpurun_out/r6_paths_v29.json 2> gpurun_out/r6_paths_v29.err; tail -2 gpurun_out/r6_paths_v29.err
for parts in none carrier level snr carrier,level carrier,snr level,snr; do
  for st in 0 160; do
    LINE_PARTS=$parts python tools/bench_paths.py --workload v29 --no-cpu-baseline --stagger $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('parts=$parts stagger=$st avg_launch_us=%.1f ms_per_step=%.4f data_mode=%s' % (d['roofline']['avg_launch_us'], d['ms_per_step'], d['config']['sampled_channels_in_data_mode_at_end']))
" >> gpurun_out/r6_v29_line_parts.log
  done
done
cat gpurun_out/r6_v29_line_parts.log
