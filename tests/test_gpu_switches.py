"""Every fast path has an A/B switch (ICSG3D_NO_*) that routes the same layer through the general kernels.
The switches are read when an engine is created; each configuration runs in its own subprocess; all of them must give
the same forward results and train-step metrics as the default configuration (different summation orders:
<= 2e-5), which keeps the fallback kernels covered at the network's real layer shapes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from icsg3d_amd.engine import UnetEngine, VaeEngine
from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
B, d = 2, 16
PU = glorot_params(unet_param_shapes(1, 95), 1)
X, lab, cond = synthetic_batch(B, d, 1, seed=0, noise=1e-3)
eps = np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32)
u = UnetEngine(d=d, max_batch=B, lr=1e-4); u.set_weights(PU)
soft, sig = u.predict(X)
mu = u.train_step(X, lab)
pm = UnetEngine(d=d, max_batch=B); pm.set_weights(PU)
v = VaeEngine(pm, d=d, max_batch=B, lr=1e-4); v.set_weights(glorot_params(vae_param_shapes(1, d=d), 3))
zm, zlv, z = v.encode(X, cond, eps)
rec = v.decode(z, cond)
mv = v.train_step(X, cond, eps)
g = u.get_grad("soft/kernel", (1, 1, 1, 128, 95))
# gradients further down the backward pass (what the BatchNorm-backward fusions feed): sampled
deep = {}
for name, shape in (("c18/kernel", (3, 3, 3, 128, 128)), ("c18/bias", (128,)), ("c17/kernel", (3, 3, 3, 192, 128)),
                    ("c17/gamma", (128,)), ("c17/beta", (128,)), ("c17/bias", (128,)), ("c16/kernel", (3, 3, 3, 256, 128)),
                    ("c15/gamma", (256,)), ("c15/beta", (256,)), ("c15/bias", (256,)), ("c4/kernel", (3, 3, 3, 64, 128)),
                    ("c3/gamma", (64,)), ("c3/bias", (64,)), ("c1/kernel", (3, 3, 3, 1, 32))):
    a = u.get_grad(name, shape).ravel()
    deep[name.replace("/", "_")] = a[::max(1, a.size // 512)].tolist()
# ... and EVERY U-Net gradient tensor of this configuration against the fp64 oracle with this run's own ReLU / pool
# decisions pinned (as tests/test_gpu_unet.py does for the default configuration): a fallback path with a real bug in
# c1..c17 cannot hide inside the loose engine-vs-engine bound the BatchNorm amplification forces on the deep gradients
from oracle import numpy_ref as R
orc = R.UnetOracle(in_ch=1, seed=1, lr=1e-4)
orc.P = {k: np.asarray(PU[k], np.float64) for k in orc.P}
LAYERS = ["c1", "c2", "c3", "c4", "c5", "c6", "c9", "c10", "c13", "c14", "c15", "c16", "c17", "c18"]
RES = {"c1": 1, "c2": 1, "c3": 2, "c4": 2, "c5": 4, "c6": 4, "c9": 8, "c10": 8, "c13": 4, "c14": 4, "c15": 2, "c16": 2, "c17": 1, "c18": 1}
COUT = dict((n, c) for n, _, c in R.UNET_CONVS)
shp = lambda n: (B, d // RES[n], d // RES[n], d // RES[n], COUT[n])
kink = {n: u.get_activation(n, shp(n)) for n in LAYERS}
affine = {n: u.get_bn_affine(n, COUT[n]) for n in ("c2", "c4", "c6")}
orc.train_on_batch(X.astype(np.float64), lab, kink=kink, affine=affine)
oracle_err = {}
for name, shape, trainable in u.tensor_infos():
    if trainable:
        ref = orc.last_grads[name]
        oracle_err[name] = float(np.abs(u.get_grad(name, shape) - ref).max() / np.abs(ref).max())
print(json.dumps({"oracle_err": oracle_err, "oracle_flips": int(sum(orc.kink_flips.values())),
                  "soft": soft[:, ::5, ::5, ::5].ravel().tolist(), "sig": sig[:, ::5, ::5, ::5].ravel().tolist(),
                  "mu": np.asarray(mu).tolist(), "mv": np.asarray(mv).tolist(), "zm": zm.ravel()[::7].tolist(),
                  "rec": rec[:, ::5, ::5, ::5].ravel().tolist(), "g": g.ravel()[::37].tolist(), **deep}))
""" % ROOT


ORACLE_GRAD_TOL = 1e-4     # every gradient tensor of every configuration vs the fp64 oracle (one-step tests: 6e-5 measured 4e-5)


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    err, flips = res.pop("oracle_err"), res.pop("oracle_flips")
    worst = max(err, key=err.get)
    print("%s: worst gradient error vs the fp64 oracle %.2e (%s), %d decisions pinned" % (env_extra or "default", err[worst], worst, flips))
    assert err[worst] <= ORACLE_GRAD_TOL, (env_extra, worst, err[worst])
    assert flips <= 64, flips
    return {k: np.asarray(v) for k, v in res.items()}


@pytest.fixture(scope="module")
def default_run():
    return _run({})


@pytest.mark.parametrize("switch", ["ICSG3D_NO_REUSE", "ICSG3D_NO_WGRAD3", "ICSG3D_NO_FWD_SPLITK",
                                    "ICSG3D_NO_THIN_N", "ICSG3D_NO_UPSPLIT", "ICSG3D_NO_THIN_C", "ICSG3D_NO_BWD_FOLD",
                                    "ICSG3D_NO_WGRAD3S", "ICSG3D_NO_COND_FOLD", "ICSG3D_NO_WINO",
                                    "ICSG3D_NO_WINO64", "ICSG3D_NO_UP3", "ICSG3D_NO_WINO_WGRAD", "ICSG3D_NO_FUSED_HEAD",
                                    "ICSG3D_NO_FAST_BNBWD", "ICSG3D_NO_THIN1_2STAGE", "ICSG3D_NO_WINOG", "ICSG3D_NO_HEAD_BNFUSE", "ICSG3D_NO_UP3N", "ICSG3D_NO_POOL_PRESUM",
                                    # not a fallback: = 1 puts every upsampled-channel launch on the 32-voxel tile that
                                    # the bench-sized launches use (here: c15.up with ONE block row in y, the VAE's d1)
                                    "ICSG3D_UP3_BIG_MIN_WG",
                                    # not a fallback either: = 0 turns the BatchNorm-backward-in-backward-data fusion on at
                                    # this size (c18 -> c17, c16 -> c15, c4 -> c3; default: from 64 MB of activations on)
                                    "ICSG3D_DGRAD_BNFUSE_MIN=0", "ICSG3D_NO_DGRAD_BNFUSE",
                                    # round 5: the DFC-VAE step's two-stream schedule (perceptual y_true pass / weight
                                    # gradients on the second stream) against the serial one
                                    "ICSG3D_NO_PM_SIDE", "ICSG3D_NO_VAE_SIDE_WGRAD", "ICSG3D_NO_HEAD_LABELS"])
def test_fallback_path_matches_default(default_run, switch):
    name, _, val = switch.partition("=")
    alt = _run({name: val or "1"})
    # The sampled gradients of the deeper layers, engine vs engine: a switch that changes a FORWARD summation order moves
    # them by up to 7e-3 of their largest entry at this size (B = 2, d = 16: the BatchNorm layers at 2^3 .. 4^3 voxels see
    # 16 .. 128 values per channel and amplify 1e-7 forward differences into different ReLU / pool decisions;
    # scripts/switch_deltas.py prints the table), the switches that only re-associate the backward pass stay below 3e-6.
    # That bound is loose by necessity -- the tight check of the deep gradients is _run's: every gradient tensor of THIS
    # configuration against the fp64 oracle with its own decisions pinned, <= 1e-4.
    backward_only = name in ("ICSG3D_DGRAD_BNFUSE_MIN", "ICSG3D_NO_DGRAD_BNFUSE", "ICSG3D_NO_BWD_FOLD",
                             "ICSG3D_NO_HEAD_BNFUSE", "ICSG3D_NO_FAST_BNBWD", "ICSG3D_NO_POOL_PRESUM",
                             "ICSG3D_NO_PM_SIDE", "ICSG3D_NO_VAE_SIDE_WGRAD", "ICSG3D_NO_HEAD_LABELS")
    for k, ref in default_run.items():
        scale = max(float(np.abs(ref).max()), 1e-30)
        deep = k.startswith("c") and "_" in k
        tol = 2e-2 if (deep and not backward_only) else 2e-5
        assert float(np.abs(alt[k] - ref).max()) <= tol * scale, (switch, k)
