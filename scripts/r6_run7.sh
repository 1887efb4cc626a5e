#!/bin/bash
mkdir -p gpurun_out
timeout 600 python scripts/r6_diag2.py 32 > gpurun_out/r6_diag2_b32.txt 2>&1; cat gpurun_out/r6_diag2_b32.txt | cut -c1-1200
timeout 600 python scripts/r6_diag2.py 8 > gpurun_out/r6_diag2_b8.txt 2>&1; cat gpurun_out/r6_diag2_b8.txt | cut -c1-600
