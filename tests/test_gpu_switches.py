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
print(json.dumps({"soft": soft[:, ::5, ::5, ::5].ravel().tolist(), "sig": sig[:, ::5, ::5, ::5].ravel().tolist(),
                  "mu": np.asarray(mu).tolist(), "mv": np.asarray(mv).tolist(), "zm": zm.ravel()[::7].tolist(),
                  "rec": rec[:, ::5, ::5, ::5].ravel().tolist(), "g": g.ravel()[::37].tolist(), **deep}))
""" % ROOT


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return {k: np.asarray(v) for k, v in json.loads(out.stdout.strip().splitlines()[-1]).items()}


@pytest.fixture(scope="module")
def default_run():
    return _run({})


@pytest.mark.parametrize("switch", ["ICSG3D_NO_REUSE", "ICSG3D_NO_WGRAD3", "ICSG3D_NO_FWD_SPLITK",
                                    "ICSG3D_NO_THIN_N", "ICSG3D_NO_UPSPLIT", "ICSG3D_NO_THIN_C", "ICSG3D_NO_BWD_FOLD",
                                    "ICSG3D_NO_WGRAD3S", "ICSG3D_SIDE_STREAM", "ICSG3D_NO_COND_FOLD", "ICSG3D_NO_WINO",
                                    "ICSG3D_NO_WINO64", "ICSG3D_NO_UP3", "ICSG3D_NO_WINO_WGRAD", "ICSG3D_NO_FUSED_HEAD",
                                    "ICSG3D_NO_FAST_BNBWD", "ICSG3D_NO_THIN1_2STAGE", "ICSG3D_NO_WINOG", "ICSG3D_NO_HEAD_BNFUSE", "ICSG3D_NO_UP3N", "ICSG3D_NO_TICKET", "ICSG3D_NO_POOL_PRESUM",
                                    # not a fallback: = 1 puts every upsampled-channel launch on the 32-voxel tile that
                                    # the bench-sized launches use (here: c15.up with ONE block row in y, the VAE's d1)
                                    "ICSG3D_UP3_BIG_MIN_WG",
                                    # not a fallback either: = 0 turns the BatchNorm-backward-in-backward-data fusion on at
                                    # this size (c18 -> c17, c16 -> c15, c4 -> c3; default: from 64 MB of activations on)
                                    "ICSG3D_DGRAD_BNFUSE_MIN=0", "ICSG3D_NO_DGRAD_BNFUSE"])
def test_fallback_path_matches_default(default_run, switch):
    name, _, val = switch.partition("=")
    alt = _run({name: val or "1"})
    # The sampled gradients of the deeper layers: a switch that changes a FORWARD summation order moves them by up to
    # 7e-3 of their largest entry at this size (B = 2, d = 16: the BatchNorm layers at 2^3 .. 4^3 voxels see 16 .. 128
    # values per channel and amplify 1e-7 forward differences; scripts/switch_deltas.py prints the table), the switches
    # that only re-associate the backward pass stay below 3e-6.
    backward_only = name in ("ICSG3D_DGRAD_BNFUSE_MIN", "ICSG3D_NO_DGRAD_BNFUSE", "ICSG3D_NO_BWD_FOLD",
                             "ICSG3D_NO_HEAD_BNFUSE", "ICSG3D_NO_TICKET", "ICSG3D_NO_FAST_BNBWD", "ICSG3D_NO_POOL_PRESUM")
    for k, ref in default_run.items():
        scale = max(float(np.abs(ref).max()), 1e-30)
        deep = k.startswith("c") and "_" in k
        tol = 2e-2 if (deep and not backward_only) else 2e-5
        assert float(np.abs(alt[k] - ref).max()) <= tol * scale, (switch, k)
