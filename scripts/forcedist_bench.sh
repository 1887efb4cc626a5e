#!/bin/bash
# On the GPU box: the U-Net step without a communicator vs with a single-rank RCCL communicator forced on (the
# driver's launch line), local BN and SyncBN.  Outputs -> gpurun_out/forcedist/; copy into profiles/ to keep.
OUT=$GRAFT_REPO_ROOT/gpurun_out/forcedist; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_nocomm.json 2> $OUT/nocomm.err
ICSG3D_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --no-cpu-baseline --no-secondary > $OUT/bench_forcedist.json 2> $OUT/forcedist.err
ICSG3D_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --no-cpu-baseline --no-secondary --sync-bn > $OUT/bench_forcedist_syncbn.json 2> $OUT/forcedist_syncbn.err
for f in nocomm forcedist forcedist_syncbn; do python3 -c "
import json,sys
d=json.loads([l for l in open('$OUT/bench_$f.json') if l.startswith('{')][-1])
print('$f', d['value'], d['ms_per_step'], d.get('ms_per_step_events_off'))"; done
