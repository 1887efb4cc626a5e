import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import numpy_ref as R
from icsg3d_amd.engine import UnetEngine, VaeEngine
import test_gpu_unet as TU, test_gpu_vae as TV
def rel(a, b): return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
which = sys.argv[1] if len(sys.argv) > 1 else "unet"
B, d, C = 2, 16, 1
if which == "unet":
    orc, eng, X, lab = TU._setup(B, d, C, "tf_cpu", lr=1e-3)
    m = eng.train_step(X, lab)
    kink = {n: eng.get_activation(n, TU._layer_shape(n, B, d)) for n in TU.UNET_LAYERS}
    affine = {n: eng.get_bn_affine(n, TU._layer_shape(n, B, d)[-1]) for n in ("c2", "c4", "c6")}
    m_ref = orc.train_on_batch(X, lab, kink=kink, affine=affine)
    print("metrics", m, m_ref); print("flips", orc.kink_flips)
    for name, shape, tr in eng.tensor_infos():
        if tr:
            g = eng.get_grad(name, shape)
            print("%-14s grad %.2e |g|max %.3e" % (name, rel(g, orc.last_grads[name]), np.abs(orc.last_grads[name]).max()))
else:
    uo, vo, ue, ve, X, cond, eps = TV._setup(B, d, C)
    m = ve.train_step(X, cond, eps)
    kink = {n: ve.get_activation(n, s) for n, s in TV._vae_layer_shapes(B, d, C).items()}
    kink_pm = {n: ue.get_activation(n, s) for n, s in TV._pm_layer_shapes(B, d).items()}
    aff = {n: ve.get_bn_affine(n, TV._vae_layer_shapes(B, d, C)[n][-1]) for n in ("e0", "e1", "e2", "e3")}
    aff_pm = {n: ue.get_bn_affine(n, TV._pm_layer_shapes(B, d)[n][-1]) for n in ("c2", "c4", "c6")}
    m_r = vo.train_on_batch(X, cond, eps, kink=kink, kink_pm=kink_pm, affine=aff, affine_pm=aff_pm)
    print("metrics", m, m_r); print("flips", vo.kink_flips)
    for name, shape, tr in ve.tensor_infos():
        if tr:
            g = ve.get_grad(name, shape)
            print("%-18s grad %.2e |g|max %.3e |gpu|max %.3e" % (name, rel(g, vo.last_grads[name]), np.abs(vo.last_grads[name]).max(), np.abs(g).max()))
