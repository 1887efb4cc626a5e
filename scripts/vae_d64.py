import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icsg3d_amd.engine import UnetEngine, VaeEngine
from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
B, d = 8, 64
ue = UnetEngine(d=d, max_batch=B); ue.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
ve = VaeEngine(ue, d=d, max_batch=B); ve.set_weights(glorot_params(vae_param_shapes(1, d=d), 3))
X, _, cond = synthetic_batch(B, d, 1, seed=0)
eps = np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32)
ve.upload_batch(X, cond, eps)
m0 = ve.train_step_resident(True); print("vae d=64 metrics", m0)
ve.sync(); t0 = time.perf_counter()
for _ in range(3): ve.train_step_resident(False)
ve.sync(); dt = (time.perf_counter() - t0) / 3
m1 = ve.train_step_resident(True)
print("DFC-VAE d=64 B=8: %.1f ms/step %.1f grids/s; loss %.4f -> %.4f" % (dt * 1e3, B / dt, m0[0], m1[0]))
