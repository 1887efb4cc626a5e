#!/bin/bash
# On the GPU box, from the repo root: full GPU test suite, the contract bench line, the rocprofv3 kernel
# statistics of the same command, and the PMC passes.  Outputs land in gpurun_out/refresh_<tag>/.
TAG=${1:-r1}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/refresh_$TAG
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*.db" -delete
cd $ROOT && bash scripts/run_pmc.sh $TAG
