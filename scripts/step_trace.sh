#!/bin/bash
# GPU box: the ordered kernel sequence of ONE steady-state step (name, grid, duration, gap to the previous kernel's end)
# usage: step_trace.sh <unet|vae> <out-name>     (writes gpurun_out/<out-name>.txt)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/st_$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$1" = "vae" ]; then SCRIPT=$ROOT/scripts/quick_bench_vae.py; else SCRIPT=$ROOT/scripts/quick_bench.py; fi
rocprofv3 --kernel-trace -f csv -d $OUT -o prof -- python3 $SCRIPT 32 32 6 > $OUT/run.log 2>&1
cd $ROOT
python3 - "$OUT" "$1" > gpurun_out/$2.txt <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marker = "adam_kernel"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# a steady-state step = the kernels between the 3rd and the 4th Adam launch of the engine being trained
a, b = idx[2] + 1, idx[3] + 1
prev_end = int(rows[a - 1]["End_Timestamp"])
tot = gap = 0.0
print("# kernels in the step: %d" % (b - a))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (s - prev_end) / 1e3
    print("%8.1f us  gap %6.1f  grid %8s x %4s  %s" % ((e - s) / 1e3, g, r["Grid_Size_X"], r["Workgroup_Size_X"], r["Kernel_Name"][:110]))
    tot += (e - s) / 1e3; gap += max(g, 0.0); prev_end = max(prev_end, e)
print("# sum of durations %.1f us, sum of positive gaps %.1f us, wall %.1f us" % (tot, gap, (int(rows[b-1]["End_Timestamp"]) - int(rows[a-1]["End_Timestamp"])) / 1e3))
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
