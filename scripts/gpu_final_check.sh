#!/bin/bash
# the last GPU call of a round: smoke, the whole GPU suite, the contract bench line -- on the final tree
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_final_smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/r6_final_smoke.log | cut -c1-300
python -m pytest tests/ -q -m gpu --durations=8 > gpurun_out/r6_final_suite.log 2>&1; echo "suite rc=$?"
tail -14 gpurun_out/r6_final_suite.log | cut -c1-200
python bench.py > gpurun_out/r6_final_bench.json 2> gpurun_out/r6_final_bench.err; echo "bench rc=$?"
python - <<'P'
import json
o=json.load(open('gpurun_out/r6_final_bench.json'))
print({k:o[k] for k in ('value','ms_per_step','ms_per_step_events_off','gpu_active_s')}, o['sustained']['ms_per_step'])
print(o['roofline']['frac'], o['roofline']['traffic_source'][:60])
print(o['secondary']['ms_per_step'], o['inference']['generate']['refine']['host_refine_ms'], o['inference']['generate']['refine']['end_to_end_grids_per_s'])
P
