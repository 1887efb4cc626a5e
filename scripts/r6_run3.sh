#!/bin/bash
# round 6, GPU call 3: whole GPU suite, the wgrad timeline, the hipGraph probe, PMC passes of the U-Net step
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu -s --durations=15 > gpurun_out/r6_gpu_suite.log 2>&1
echo "rc=$?" >> gpurun_out/r6_gpu_suite.log
grep -E "^d=|passed|failed|FAILED|rc=" gpurun_out/r6_gpu_suite.log | tail -30
ICSG3D_LIB_PATH=icsg3d_amd/variants/libicsg3d_hip_wgtl.so timeout 600 python scripts/wgrad_timeline.py > gpurun_out/r6_wgrad_timeline.txt 2> gpurun_out/r6_wgrad_timeline.err
tail -25 gpurun_out/r6_wgrad_timeline.txt; tail -3 gpurun_out/r6_wgrad_timeline.err
timeout 600 python scripts/graph_probe.py > gpurun_out/r6_graph_probe.txt 2> gpurun_out/r6_graph_probe.err
cat gpurun_out/r6_graph_probe.txt; tail -5 gpurun_out/r6_graph_probe.err
timeout 1500 bash scripts/run_pmc.sh r6 > gpurun_out/r6_pmc.log 2>&1
cp gpurun_out/pmc_r6/summary.txt gpurun_out/r6_pmc_summary.txt; cp gpurun_out/pmc_r6/traffic.json gpurun_out/r6_pmc_traffic.json
grep -A8 "conv_wino_wgrad_kernel<true, true>" gpurun_out/r6_pmc_summary.txt | head -12
