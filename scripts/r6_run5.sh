#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize_oracle.py -q -s --durations=6 > gpurun_out/r6_fullsize_oracle.log 2>&1
echo "rc=$?" >> gpurun_out/r6_fullsize_oracle.log
grep -E "^d=|^F?d=|passed|failed|rc=|AssertionError" gpurun_out/r6_fullsize_oracle.log | cut -c1-1500
