"""which buffers of a DFC-VAE step at B = 3 depend on the handle's max_batch (3 vs 5)?  They must not."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import numpy_ref as R
from icsg3d_amd.engine import UnetEngine, VaeEngine
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
from fuzz_steps import vae_shapes, ushape, UNET_LAYERS, inputs

d, C, B = int(os.environ.get("D", 32)), int(os.environ.get("C", 1)), int(os.environ.get("B", 3))
X, _, cond, eps = inputs(B, d, C, 11)
res = {}
for maxB in [int(v) for v in os.environ.get("MAXB", "3,5").split(",")]:
    ue = UnetEngine(in_channels=C, d=d, max_batch=maxB, lr=1e-3)
    ue.set_weights(R.init_params(R.unet_param_shapes(C, 95), 7))
    ve = VaeEngine(ue, in_channels=C, d=d, max_batch=maxB, lr=5e-4)
    ve.set_weights(R.init_params(R.vae_param_shapes(C, d=d), 8))
    mode = os.environ.get("MODE", "train")
    m = ve.train_step(X, cond, eps) if mode == "train" else ve.test_step(X, cond, eps)
    vs = vae_shapes(B, d, C)
    out = {"metrics": np.asarray(m)}
    for n, s in vs.items():
        out["act:" + n] = ve.get_activation(n, s)
    for n in ("e0", "e1", "e2", "e3", "d0", "d1", "d2", "d3", "dout"):
        a = ve.get_bn_affine(n, vs[n][-1])
        out["aff:" + n] = np.stack([np.asarray(v) for v in a])
    for n in UNET_LAYERS[:8]:
        out["pm:" + n] = ue.get_activation(n, ushape(n, B, d))
    if mode == "train":
        for name, shape, tr in ve.tensor_infos():
            if tr:
                out["grad:" + name] = ve.get_grad(name, shape)
    res[maxB] = out
    if os.environ.get("ICSG3D_DEBUG_CANARY"):
        for nm, e in (("vae", ve), ("unet", ue)):
            try:
                print("canaries", nm, e.check_canaries(), flush=True)
            except Exception as ex:
                print("canaries", nm, "DIRTY:", str(ex)[:600], flush=True)
    print("max_batch %d: metrics %s" % (maxB, m), flush=True)
    ve.close(); ue.close()
ks = sorted(res)
a, b = res[ks[0]], res[ks[-1]]
for k in a:
    dlt = float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max())
    sc = float(np.abs(a[k]).max())
    if dlt > 0:
        print("%-22s max|diff| %.3e  (max|value| %.3e)" % (k, dlt, sc))
print("done")
