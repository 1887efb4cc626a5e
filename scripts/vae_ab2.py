"""Developer probe: does the DFC-VAE step time depend on what the shared perceptual U-Net engine did before?"""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icsg3d_amd.engine import UnetEngine, VaeEngine
from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
B, d = 32, 32
X, lab, cond = synthetic_batch(B, d, 1, seed=0)
eps = np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32)
def vae_time(ue, tag):
    ve = VaeEngine(ue, d=d, max_batch=B, lr=5e-4); ve.set_weights(glorot_params(vae_param_shapes(1, d=d), 3))
    ve.upload_batch(X, cond, eps)
    for _ in range(5): ve.train_step_resident(False)
    ve.sync(); t0 = time.perf_counter()
    for _ in range(30): ve.train_step_resident(False)
    ve.sync(); dt = (time.perf_counter() - t0) / 30
    print("%-40s %.3f ms/step" % (tag, dt * 1e3), flush=True)
    return ve
ue = UnetEngine(d=d, max_batch=B, lr=3e-6); ue.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
v1 = vae_time(ue, "fresh U-Net engine")
ue.upload_batch(X, lab)
v2 = vae_time(ue, "after upload_batch")
for _ in range(10): ue.train_step_resident(False)
ue.sync()
v3 = vae_time(ue, "after 10 U-Net train steps")
ue.profile_filter("conv_fwd:"); ue.profile_enable(True)
for _ in range(3): ue.train_step_resident(False)
ue.sync(); ue.profile_rows(); ue.profile_enable(False); ue.profile_filter("")
v4 = vae_time(ue, "after profiled U-Net steps")
