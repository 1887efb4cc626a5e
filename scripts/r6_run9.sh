#!/bin/bash
mkdir -p gpurun_out
TAG=default timeout 600 python scripts/r6_diag4.py 2>&1 | grep -E "^default|Error" | cut -c1-900
TAG=nofold ICSG3D_NO_COND_FOLD=1 timeout 600 python scripts/r6_diag4.py 2>&1 | grep -E "^nofold|Error" | cut -c1-900
