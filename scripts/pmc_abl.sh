#!/bin/bash
# compare memory-path counters between ablation variants of the c18 forward kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_abl; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for ABL in 3 7; do
 for P in "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" "TCP_TCC_READ_REQ_LATENCY TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_READ_TAGCONFLICT_STALL_CYCLES" "TCP_TCR_TCP_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES TCP_TCP_LATENCY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  N=$(echo $P | cut -d" " -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $P -f csv -d $OUT/a${ABL}_$N -o x -- python3 $GRAFT_REPO_ROOT/scripts/conv_abl.py $ABL > /dev/null 2>&1
 done
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_abl"
for abl in (3, 7):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(out + "/a%d_*/**/*counter_collection.csv" % abl, recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_fwd_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print("ABL", abl, " ".join("%s=%.4g" % (k, v[0] / v[1]) for k, v in sorted(acc.items())))
PY
rm -rf $OUT/a*_*
