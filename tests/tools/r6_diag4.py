"""pinned oracle vs engine for e0 / e1 / e2 gradients at B = 32 (TAG = default | nofold via env), saved as npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.set_num_threads(32)
from oracle import numpy_ref as R, torch_ref as T
import test_gpu_fullsize_oracle as F
from icsg3d_amd.engine import UnetEngine, VaeEngine
B, d = 32, 32
X, _, cond, eps = F._inputs(B, d)
ush, vsh = R.unet_param_shapes(1, 95), R.vae_param_shapes(1, d=d)
Pu, Su, Pv, Sv = R.init_params(ush, 1), R.init_bn_state(ush), R.init_params(vsh, 3), R.init_bn_state(vsh)
ue = UnetEngine(in_channels=1, d=d, max_batch=B); ue.set_weights(Pu)
ve = VaeEngine(ue, in_channels=1, d=d, max_batch=B, lr=5e-4); ve.set_weights(Pv)
m = ve.train_step(X, cond, eps)
grads = {name: ve.get_grad(name, shape) for name, shape, tr in ve.tensor_infos() if tr}
vs = F._vae_shapes(B, d); ps = {n: F._ushape(n, B, d) for n in F.UNET_LAYERS[:8]}
kink = {n: ve.get_activation(n, s) for n, s in vs.items()}
kink_pm = {n: ue.get_activation(n, s) for n, s in ps.items()}
aff = {n: ve.get_bn_affine(n, vs[n][-1]) for n in ("e0", "e1", "e2", "e3")}
aff_pm = {n: ue.get_bn_affine(n, ps[n][-1]) for n in ("c2", "c4", "c6")}
m_ref, g_ref, stats, _, _, _ = T.vae_step_grads(Pv, Sv, Pu, Su, X, cond, eps, in_ch=1, d=d, kink=kink, kink_pm=kink_pm, affine=aff, affine_pm=aff_pm)
tag = os.environ.get("TAG", "default")
keep = [k for k in grads if k.split("/")[0] in ("e0", "e1", "e2")]
np.savez(os.path.join(ROOT, "gpurun_out", "r6_pinned_%s.npz" % tag), **{"eng__" + k.replace("/", "__"): grads[k] for k in keep},
         **{"ref__" + k.replace("/", "__"): g_ref[k] for k in keep})
for k in ("e1/beta", "e1/gamma", "e0/beta"):
    e, r = grads[k].astype(np.float64), g_ref[k]
    print(tag, k, "max|ref| %.3e" % np.abs(r).max(), "diff/max per channel:", " ".join("%+.1e" % v for v in (e - r) / np.abs(r).max()))
