"""Developer helper: max relative difference (to the tensor's max |value|) of every quantity tests/test_gpu_switches.py
compares, per switch -- what the tolerances in that test were chosen from."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_switches as T

ref = T._run({})
switches = sys.argv[1:] or ["ICSG3D_NO_FWD_SPLITK", "ICSG3D_NO_UPSPLIT", "ICSG3D_NO_THIN_C", "ICSG3D_NO_WINO", "ICSG3D_NO_WINO64",
                            "ICSG3D_NO_WINOG", "ICSG3D_DGRAD_BNFUSE_MIN=0", "ICSG3D_NO_BWD_FOLD"]
for sw in switches:
    name, _, val = sw.partition("=")
    alt = T._run({name: val or "1"})
    row = []
    for k, r in ref.items():
        scale = max(float(np.abs(r).max()), 1e-30)
        row.append("%s %.1e" % (k, float(np.abs(alt[k] - r).max()) / scale))
    print(sw, " | ".join(row), flush=True)
