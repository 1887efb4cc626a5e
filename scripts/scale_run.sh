#!/bin/bash
# On a multi-GPU MI355X node, from the repo root: the weak-scaling curve of the contract line at N = 1, 2, 4, 8 (32 grids per
# GPU; `python bench.py --gpus N` starts its own ranks), the two-rank parity test, and the per-launch-site rows of a 2-rank run
# (the RCCL buckets on the communication stream next to the backward kernels).  Nothing here has run yet: every box this
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
# (No rocprofv3 around the launcher: a profiled process that has initialised the GPU must not exec its ranks.  The engine's own
# profiler rows carry the RCCL buckets -- "rccl_allreduce_grads", HIP events on the communication stream: bench.py --dump-rows.)
if [ 2 -le $NG ]; then
  python bench.py --gpus 2 --no-cpu-baseline --no-secondary --no-inference --dump-rows $OUT/rows_n2.jsonl > $OUT/bench_rows_n2.json 2> $OUT/bench_rows_n2.err
fi
