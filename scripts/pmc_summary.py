"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (averages per dispatch).
usage: pmc_summary.py <dir> [kernel-substring ...]"""
import csv, glob, sys, collections
d = sys.argv[1]
filt = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
meta = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = k.split("(")[0].replace("void ics::", "").replace("ics::", "")
        if filt and not any(s in short for s in filt):
            continue
        a = acc[short][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        meta[short] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("GRBM_GUI_ACTIVE", [0, 1]))[0]):
    c = {n: v[0] / v[1] for n, v in acc[k].items()}
    n = max(v[1] for v in acc[k].values())
    print("== %s  dispatches=%d vgpr/agpr/lds/grid=%s" % (k, n, meta[k]))
    print("   " + "  ".join("%s=%.4g" % kv for kv in sorted(c.items())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c:
        print("   mfma_busy/busy_cu = %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CU_CYCLES"]))
    if "SQ_WAVE_CYCLES" in c:
        for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if w in c:
                print("   %s/WAVE_CYCLES = %.3f" % (w, c[w] / c["SQ_WAVE_CYCLES"]))
