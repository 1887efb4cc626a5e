"""round 6 diagnostic: DFC-VAE forward activations of e0 / e1 at B = 32 with the condition fold on vs off (two engines in one
process), to localise the e0 / e1 gradient error the pinned oracle test shows only with the fold on."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icsg3d_amd.engine import UnetEngine, VaeEngine
from oracle import numpy_ref as R
B, d = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 32
X, _, cond = R.synthetic_batch(B, d, 1, seed=0, dtype=np.float64)
X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
eps = np.random.default_rng(2).standard_normal((B, 256))
ush, vsh = R.unet_param_shapes(1, 95), R.vae_param_shapes(1, d=d)
Pu, Pv = R.init_params(ush, 1), R.init_params(vsh, 3)
res = {}
for tag, env in (("fold", None), ("nofold", "ICSG3D_NO_COND_FOLD")):
    if env: os.environ[env] = "1"
    ue = UnetEngine(in_channels=1, d=d, max_batch=B); ue.set_weights(Pu)
    ve = VaeEngine(ue, in_channels=1, d=d, max_batch=B, lr=5e-4); ve.set_weights(Pv)
    if env: del os.environ[env]
    m = ve.train_step(X, cond.astype(np.float64), eps)
    res[tag] = {"m": m, "e0": ve.get_activation("e0", (B, d, d, d, 16)), "e1": ve.get_activation("e1", (B, d // 2, d // 2, d // 2, 32)),
                "aff0": ve.get_bn_affine("e0", 16), "aff1": ve.get_bn_affine("e1", 32),
                "e2:dA": ve.get_activation("e2:dA", (B, d // 4, d // 4, d // 4, 32)), "e2:dy": ve.get_activation("e2:dy", (B, d // 4, d // 4, d // 4, 64)),
                "e1:dy": ve.get_activation("e1:dy", (B, d // 2, d // 2, d // 2, 32)), "e1:dA": ve.get_activation("e1:dA", (B, d // 2, d // 2, d // 2, 16)),
                "e1:pooled": ve.get_activation("e1:pooled", (B, d // 4, d // 4, d // 4, 32)), "e0:dy": ve.get_activation("e0:dy", (B, d, d, d, 16)),
                "e3:dA": ve.get_activation("e3:dA", (B, d // 8, d // 8, d // 8, 64)),
                "g": {n: ve.get_grad(n, s) for n, s, tr in ve.tensor_infos() if tr and n.split("/")[0] in ("e0", "e1", "e2")}}
    ve.close(); ue.close()
a, b = res["fold"], res["nofold"]
print("metrics fold  ", a["m"]); print("metrics nofold", b["m"])
for k in ("e0", "e1"):
    da = np.abs(a[k].astype(np.float64) - b[k]); sc = np.abs(b[k]).max()
    print("%s activations: max |diff| / max|s| = %.2e (max|s| %.3g); mean |diff| %.2e" % (k, da.max() / sc, sc, da.mean()))
    per_b = da.reshape(B, -1).max(1) / sc
    print("   per sample max: " + " ".join("%.1e" % v for v in per_b))
    if k == "e0":
        S = d
        idx = np.arange(S); cls = np.where(idx == 0, 0, np.where(idx == S - 1, 2, 1))
        c3 = cls[:, None, None] * 9 + cls[None, :, None] * 3 + cls[None, None, :]
        dm = da.max(axis=(0, 4))
        print("   per border class max: " + " ".join("%d:%.1e" % (c, dm[c3 == c].max() / sc) for c in range(27)))
for k in ("e3:dA", "e2:dy", "e2:dA", "e1:pooled", "e1:dy", "e1:dA", "e0:dy"):
    da = np.abs(a[k].astype(np.float64) - b[k]); sc = np.abs(b[k]).max()
    nb = int((da > 1e-4 * sc).sum())
    print("%-10s max |diff| / max %.2e, mean |diff| / max %.2e, entries off by > 1e-4 of max: %d of %d; per sample max: %s"
          % (k, da.max() / sc, da.mean() / sc, nb, da.size, " ".join("%.0e" % v for v in da.reshape(B, -1).max(1) / sc)))
    if nb and nb < 200:
        idx = np.argwhere(da > 1e-4 * sc)
        print("    where:", idx[:12].tolist())
for k in ("aff0", "aff1"):
    print(k, "scale diff %.2e shift diff %.2e" % (np.abs(a[k][0] - b[k][0]).max() / np.abs(b[k][0]).max(), np.abs(a[k][1] - b[k][1]).max() / np.abs(b[k][1]).max()))
for k in a["g"]:
    print("grad %-12s fold vs nofold %.2e" % (k, np.abs(a["g"][k].astype(np.float64) - b["g"][k]).max() / max(np.abs(b["g"][k]).max(), 1e-30)))
