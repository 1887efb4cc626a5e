#!/bin/bash
# GPU box: per-launch kernel durations of a few U-Net steps, grouped by (kernel, grid) -- which launch of a shared
# kernel is the slow one.  usage: ktrace.sh <out-name> [filter-substring]   (writes gpurun_out/<out-name>.txt)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/kt_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d $OUT -o prof -- python3 $ROOT/scripts/quick_bench.py 32 32 3 > $OUT/run.log 2>&1
cd $ROOT
python3 - "$OUT" "$2" > gpurun_out/$1.txt <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if filt in r["Kernel_Name"]:
        g[(r["Kernel_Name"][:70], r["Grid_Size_X"], r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-72s grid %9s wg %4s n=%4d  med %9.1f us  min %9.1f" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
    if len(v) <= 48: print("      " + " ".join("%.0f" % x for x in v))
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
