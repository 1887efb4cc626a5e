"""canary run: DFC-VAE + U-Net train steps at B = 32 with guard bytes behind every buffer"""
import os, sys
import numpy as np
os.environ["ICSG3D_DEBUG_CANARY"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icsg3d_amd.engine import UnetEngine, VaeEngine
from oracle import numpy_ref as R
B, d = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 32
X, lab, cond = R.synthetic_batch(B, d, 1, seed=0, dtype=np.float64)
eps = np.random.default_rng(2).standard_normal((B, 256))
ush, vsh = R.unet_param_shapes(1, 95), R.vae_param_shapes(1, d=d)
Pu, Pv = R.init_params(ush, 1), R.init_params(vsh, 3)
ue = UnetEngine(in_channels=1, d=d, max_batch=B); ue.set_weights(Pu)
ve = VaeEngine(ue, in_channels=1, d=d, max_batch=B, lr=5e-4); ve.set_weights(Pv)
print("after creation:", ve.check_canaries(), ue.check_canaries())
m = ve.train_step(X, cond.astype(np.float64), eps)
print("after a DFC-VAE train step:", m, "\n  vae:", ve.check_canaries(), "\n  unet:", ue.check_canaries())
m = ue.train_step(X, lab)
print("after a U-Net train step:", m, "\n  unet:", ue.check_canaries())
ue.predict(X[:16]); ve.test_step(X, cond.astype(np.float64), eps)
print("after predict / test_step:\n  vae:", ve.check_canaries(), "\n  unet:", ue.check_canaries())
