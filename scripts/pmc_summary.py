"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (averages per dispatch).
usage: pmc_summary.py <dir> [kernel-substring ...]     (env KSTATS=<rocprofv3 --stats kernel_stats.csv> adds each
kernel's average duration as avg_ms: bench.py compares it with its live HIP-event average to detect a stale profile)
Also writes <dir>/traffic.json: per bench.py kernel id, HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB
(gfx950: FETCH_SIZE counts 64-B requests for 128-B wide reads, MI355X_MICROARCH.md HBM/rocprofv3 section)."""
import csv, glob, json, re, sys, collections


def bench_id(short):
    """bench.py / the engine profiler label kernels by their exact template instantiation"""
    return short.strip()

import os
d = sys.argv[1]
filt = sys.argv[2:]
avg_ms = {}
if os.environ.get("KSTATS") and os.path.exists(os.environ["KSTATS"]):
    for r in csv.DictReader(open(os.environ["KSTATS"])):
        short = r["Name"].split("(")[0].replace("void ics::", "").replace("ics::", "")
        avg_ms[short.strip()] = float(r["AverageNs"]) / 1e6
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

# ---- traffic.json keyed by bench.py kernel ids (dispatch-weighted over the loader variants of one tile)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for k in acc:
    for n, v in acc[k].items():
        a = agg[bench_id(k)][n]
        a[0] += v[0]; a[1] += v[1]
out = {"source": "rocprofv3 --kernel-trace --pmc <one counter group per pass> (scripts/run_pmc.sh); "
                 "traffic = (2*FETCH_SIZE + WRITE_SIZE) KB per launch (gfx950 correction for wide reads)",
       "kernels": {}}
for k, cs in agg.items():
    c = {n: v[0] / v[1] for n, v in cs.items()}
    e = {"launches_sampled": max(v[1] for v in cs.values())}
    if k in avg_ms:
        e["avg_ms"] = round(avg_ms[k], 4)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["fetch_size_kb_avg"] = round(c["FETCH_SIZE"], 1)
        e["write_size_kb_avg"] = round(c["WRITE_SIZE"], 1)
        e["traffic_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"] > 0:
        e["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CU_CYCLES"] / 4, 4)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        e["lds_bank_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
        e["wave_wait_any_frac"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    out["kernels"][k] = e
json.dump(out, open(d + "/traffic.json", "w"), indent=1)
