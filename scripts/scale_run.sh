#!/bin/bash
# On a multi-GPU MI355X node, from the repo root: the weak-scaling curve of the contract line at N = 1, 2, 4, 8 (32 grids per
# GPU; `python bench.py --gpus N` starts its own ranks), the two-rank parity test, and a kernel trace of one 2-rank run that
# shows the RCCL buckets on the communication stream next to the backward kernels.  Nothing here has run yet: every box this
# repository has seen had ONE GPU.  Outputs -> gpurun_out/scale_<tag>/; copy scale.json and the trace summary into profiles/.
TAG=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/scale_$TAG
mkdir -p $OUT
cd $ROOT
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $NG" | tee $OUT/info.txt
python -m pytest tests/test_gpu_dp2.py -m gpu -q > $OUT/dp2_test.log 2>&1; tail -2 $OUT/dp2_test.log
for N in 1 2 4 8; do
  if [ $N -le $NG ]; then
    python bench.py --gpus $N --no-cpu-baseline --no-inference > $OUT/bench_n$N.json 2> $OUT/bench_n$N.err
    python bench.py --gpus $N --no-cpu-baseline --no-inference --sync-bn --no-secondary > $OUT/bench_syncbn_n$N.json 2> $OUT/bench_syncbn_n$N.err
  fi
done
python - "$OUT" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "bench_n*.json")) + glob.glob(os.path.join(out, "bench_syncbn_n*.json"))):
    lines = [l for l in open(f) if l.startswith("{")]
    if not lines:
        continue
    d = json.loads(lines[-1])
    key = ("syncbn" if "syncbn" in f else "localbn")
    rows.setdefault(key, {})[d["n_gpus"]] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "rccl_ranks": d.get("rccl_ranks"),
                                             "vae": (d.get("secondary") or {}).get("value")}
for key, r in rows.items():
    base = r.get(1, {}).get("value")
    for n, v in sorted(r.items()):
        v["speedup_vs_1"] = round(v["value"] / base, 3) if base else None
json.dump(rows, open(os.path.join(out, "scale.json"), "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
if [ 2 -le $NG ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_n2 -o prof -- python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29611 $ROOT/bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-inference > $OUT/prof_n2.log 2>&1
  find $OUT -name "*kernel_trace.csv" -size +64M -delete
fi
