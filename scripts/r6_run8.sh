#!/bin/bash
mkdir -p gpurun_out
timeout 600 python scripts/r6_diag3.py 32 32 > gpurun_out/r6_canary_b32.txt 2>&1; cat gpurun_out/r6_canary_b32.txt | cut -c1-1500
timeout 600 python scripts/r6_diag3.py 8 64 > gpurun_out/r6_canary_d64.txt 2>&1; cat gpurun_out/r6_canary_d64.txt | cut -c1-1500
timeout 600 python scripts/r6_diag3.py 2 16 > gpurun_out/r6_canary_d16.txt 2>&1; cat gpurun_out/r6_canary_d16.txt | cut -c1-1500
