"""Developer probe: the U-Net / DFC-VAE train step through the host-buffer entry points (ics_unet_train_step / ics_vae_train_step:
numpy in, metrics out, one synchronisation per step -- what train_on_batch does) next to the resident form bench.py times.
python scripts/pcie_inclusive.py"""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icsg3d_amd.engine import UnetEngine, VaeEngine
from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
B, d, K = 32, 32, 20
X, lab, cond = synthetic_batch(B, d, 1, seed=0)
eps = np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32)
ue = UnetEngine(d=d, max_batch=B, lr=3e-6); ue.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
for _ in range(3): ue.train_step(X, lab)
t0 = time.perf_counter()
for _ in range(K): ue.train_step(X, lab)
t_host = (time.perf_counter() - t0) / K
ue.upload_batch(X, lab)
for _ in range(3): ue.train_step_resident(False)
ue.sync(); t0 = time.perf_counter()
for _ in range(K): ue.train_step_resident(False)
ue.sync(); t_res = (time.perf_counter() - t0) / K
print("U-Net  host buffers %.3f ms/step (%.1f grids/s), resident %.3f ms/step (%.1f grids/s): +%.3f ms for %.1f MB in, 20 B out"
      % (t_host * 1e3, B / t_host, t_res * 1e3, B / t_res, (t_host - t_res) * 1e3, (X.nbytes + lab.nbytes) / 1e6))
pm = UnetEngine(d=d, max_batch=B); pm.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
ve = VaeEngine(pm, d=d, max_batch=B, lr=5e-4); ve.set_weights(glorot_params(vae_param_shapes(1, d=d), 3))
for _ in range(3): ve.train_step(X, cond, eps)
t0 = time.perf_counter()
for _ in range(K): ve.train_step(X, cond, eps)
t_host = (time.perf_counter() - t0) / K
ve.upload_batch(X, cond, eps)
for _ in range(3): ve.train_step_resident(False)
ve.sync(); t0 = time.perf_counter()
for _ in range(K): ve.train_step_resident(False)
ve.sync(); t_res = (time.perf_counter() - t0) / K
print("DFC-VAE host buffers %.3f ms/step (%.1f grids/s), resident %.3f ms/step (%.1f grids/s): +%.3f ms"
      % (t_host * 1e3, B / t_host, t_res * 1e3, B / t_res, (t_host - t_res) * 1e3))
