#!/bin/bash
# round 6, GPU call 2: the whole GPU suite (with the new whole-network oracle tests) + the contract bench line
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu -s --durations=15 > gpurun_out/r6_gpu_suite.log 2>&1
echo "rc=$?" >> gpurun_out/r6_gpu_suite.log
grep -E "^d=|passed|failed|error|rc=" gpurun_out/r6_gpu_suite.log | tail -40
python bench.py > gpurun_out/r6_bench_a.json 2> gpurun_out/r6_bench_a.err
echo "bench rc=$?"
python - <<'P'
import json
o=json.load(open('gpurun_out/r6_bench_a.json'))
print({k:o[k] for k in ('value','ms_per_step','ms_per_step_events_off','sustained','gpu_active_s')})
print(o['secondary']['ms_per_step'], o['secondary']['kernel_launches_per_step'], o['secondary']['compute_frac'])
print(o['cpu_baseline'])
print(o['roofline'])
P
