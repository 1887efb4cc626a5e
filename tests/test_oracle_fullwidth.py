"""CPU: the numpy oracle against the d=32 torch-fp64 fixture (the d=16 one is in test_oracle_golden.py; d=64 is
compared on the GPU box inside tests/test_gpu_fullwidth.py, where the oracle runs anyway)."""
import os

import numpy as np

from oracle import numpy_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))


def _checks(g):
    f = np.asarray(g, np.float64).ravel()
    idx = np.linspace(0, f.size - 1, 8).astype(int)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[idx]])


def test_numpy_vae_oracle_matches_torch_fixture_d32():
    fx = np.load(os.path.join(HERE, "golden", "unet_vae_d32_b2.npz"))
    B, d = int(fx["B"]), int(fx["d"])
    X, _, cond = R.synthetic_batch(B, d, 1, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    eps = np.random.default_rng(2).standard_normal((B, 256))
    uo = R.UnetOracle(in_ch=1, seed=1)
    vo = R.VaeOracle(uo, in_ch=1, d=d, seed=3)
    m = vo.train_on_batch(X, cond.astype(np.float64), eps)
    np.testing.assert_allclose(m, fx["vae_train_metrics"], rtol=1e-10)
    for k, g in vo.last_grads.items():
        ref = fx["vae_grad__" + k.replace("/", "__")]
        np.testing.assert_allclose(_checks(g), ref, rtol=1e-7, atol=1e-12 * max(np.abs(ref[1]), 1.0), err_msg=k)
