#!/bin/bash
# the whole GPU suite as an out-of-bounds hunt: guard bytes behind every device buffer, checked when a handle is destroyed
mkdir -p gpurun_out
ICSG3D_DEBUG_CANARY=1 python -m pytest tests/ -q -m gpu -s -k "not fullsize_oracle" > gpurun_out/r6_canary_suite.log 2>&1
echo "rc=$?"
grep -c "CANARY DIRTY" gpurun_out/r6_canary_suite.log; grep "CANARY DIRTY" gpurun_out/r6_canary_suite.log | sort | uniq -c | head -20
tail -3 gpurun_out/r6_canary_suite.log
