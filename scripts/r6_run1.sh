#!/bin/bash
# round 6, GPU call 1: host facts + the whole-network oracle tests at the stated batch
mkdir -p gpurun_out
{ nproc; free -g; lscpu | grep -E "Model name|Socket|Thread|Core"; } > gpurun_out/r6_host.txt 2>&1
python -m pytest tests/test_gpu_fullsize_oracle.py -q -s -x > gpurun_out/r6_fullsize_oracle.log 2>&1
echo "rc=$?" >> gpurun_out/r6_fullsize_oracle.log
tail -40 gpurun_out/r6_fullsize_oracle.log
