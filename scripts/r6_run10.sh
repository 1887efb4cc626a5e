#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu --durations=12 > gpurun_out/r6_gpu_suite.log 2>&1
echo "rc=$?" >> gpurun_out/r6_gpu_suite.log
tail -25 gpurun_out/r6_gpu_suite.log | cut -c1-300
python bench.py > gpurun_out/r6_bench_b.json 2> gpurun_out/r6_bench_b.err
echo "bench rc=$?"
python - <<'P'
import json
o=json.load(open('gpurun_out/r6_bench_b.json'))
print({k:o[k] for k in ('value','ms_per_step','ms_per_step_events_off','sustained','gpu_active_s')})
print(o['secondary']['ms_per_step'], o['secondary']['ms_per_step_events_off'], o['secondary']['kernel_launches_per_step'], o['secondary']['compute_frac'])
print(o['inference']['generate']['refine'])
print(o['inference']['generate']['value'], o['inference']['predict']['value'])
print(o['cpu_baseline']['generate'])
P
