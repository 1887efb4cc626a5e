#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/r6_diag.py threads > gpurun_out/r6_diag_threads.txt 2>&1
head -8 gpurun_out/r6_diag_threads.txt
TAG=default timeout 300 python scripts/r6_diag.py e0 2>&1 | tail -1
TAG=nocondfold ICSG3D_NO_COND_FOLD=1 timeout 300 python scripts/r6_diag.py e0 2>&1 | tail -1
TAG=nothinc ICSG3D_NO_THIN_C=1 timeout 300 python scripts/r6_diag.py e0 2>&1 | tail -1
timeout 600 python scripts/r6_diag.py e0ref 2>&1 | tail -1
timeout 600 python scripts/prof_refine.py 32 > gpurun_out/r6_prof_refine.txt 2>&1
head -45 gpurun_out/r6_prof_refine.txt; tail -3 gpurun_out/r6_prof_refine.txt
