"""Developer timing helper (not the contract bench): U-Net train step at B=32,d=32 with per-kernel rows."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icsg3d_amd.engine import UnetEngine
from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
eng = UnetEngine(in_channels=1, d=d, max_batch=B, lr=3e-6)
eng.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
X, lab, _ = synthetic_batch(B, d, 1, seed=0)
eng.upload_batch(X, lab)
print("metrics", eng.train_step_resident(True))
eng.sync()
t0 = time.perf_counter()
for _ in range(steps):
    eng.train_step_resident(False)
eng.sync()
dt = (time.perf_counter() - t0) / steps
print("ms/step %.2f  grids/s %.1f" % (dt * 1e3, B / dt))
eng.profile_enable(True)
for _ in range(2):
    eng.train_step_resident(False)
eng.sync()
rows = sorted(eng.profile_rows(), key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in rows)
print("total profiled ms/step %.2f" % (tot / 2))
for r in rows:
    tf = r["flop"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0
    gb = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0
    print("%-22s n=%3d  %8.3f ms/step  %6.1f TF/s  %7.1f GB/s" % (r["label"], r["launches"], r["ms"] / 2, tf, gb))
