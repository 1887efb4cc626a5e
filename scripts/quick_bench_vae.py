"""Developer timing helper (not the contract bench): DFC-VAE train step with per-kernel rows.
python scripts/quick_bench_vae.py [B d steps]"""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icsg3d_amd.engine import UnetEngine, VaeEngine
from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ue = UnetEngine(d=d, max_batch=B); ue.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
ve = VaeEngine(ue, d=d, max_batch=B); ve.set_weights(glorot_params(vae_param_shapes(1, d=d), 3))
X, _, cond = synthetic_batch(B, d, 1, seed=0)
eps = np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32)
ve.upload_batch(X, cond, eps)
print("metrics", ve.train_step_resident(True))
ve.sync(); t0 = time.perf_counter()
for _ in range(steps): ve.train_step_resident(False)
ve.sync(); dt = (time.perf_counter() - t0) / steps
print("ms/step %.3f  grids/s %.1f" % (dt * 1e3, B / dt))
ve.profile_enable(True); ue.profile_enable(True)
for _ in range(2): ve.train_step_resident(False)
ve.sync()
rows = sorted(list(ve.profile_rows()) + [dict(r, label="pm:" + r["label"]) for r in ue.profile_rows()], key=lambda r: -r["ms"])
print("total profiled ms/step %.2f  launches/step %d" % (sum(r["ms"] for r in rows) / 2, sum(r["launches"] for r in rows) / 2))
for r in rows:
    tf = r["flop"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0
    gb = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0
    print("%-60s n=%3d  %8.3f ms/step  %6.1f TF/s  %7.1f GB/s" % (r["label"][:60], r["launches"], r["ms"] / 2, tf, gb))
