"""Pins the numpy oracle (oracle/numpy_ref.py) against tests/golden/unet_vae_d16_b2.npz, produced by
the independent torch-CPU implementation (tests/golden/make_golden.py).  The reference itself has no
golden vectors and cannot be imported here: parity w.r.t. Keras stays "unpinned"."""
import os

import numpy as np
import pytest

from oracle import numpy_ref as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_vae_d16_b2.npz"))


def checks(g):
    f = np.asarray(g, np.float64).ravel()
    idx = np.linspace(0, f.size - 1, 8).astype(int)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[idx]])


def inputs():
    B, d, C = int(G["B"]), int(G["d"]), int(G["C"])
    X, lab, cond = R.synthetic_batch(B, d, C, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    return B, d, C, X, lab, cond.astype(np.float64)


@pytest.mark.parametrize("ties", ["tf_cpu", "first"])
def test_unet_step_against_golden(ties):
    B, d, C, X, lab, _ = inputs()
    u = R.UnetOracle(in_ch=C, seed=1, lr=1e-3, pool_ties=ties)
    cache = {}
    soft, sig = u.forward(X, training=True, cache=cache)
    m = u.loss_and_metrics(soft, sig, lab)
    np.testing.assert_allclose(m[:3], G["unet_%s_metrics" % ties], rtol=1e-10)
    np.testing.assert_allclose(soft[0, ::5, ::5, ::5, ::7], G["unet_%s_soft_sample" % ties], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(sig[0, ::5, ::5, ::5, 0], G["unet_%s_sig_sample" % ties], rtol=1e-9)
    for n, _, _ in R.UNET_CONVS:
        np.testing.assert_allclose(cache[n]["mean"], G["unet_%s_bnmean__%s" % (ties, n)], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(cache[n]["var"], G["unet_%s_bnvar__%s" % (ties, n)], rtol=1e-9)
    grads = u.backward(lab, cache)
    for k, g in grads.items():
        ref = G["unet_%s_grad__%s" % (ties, k.replace("/", "__"))]
        got = checks(g)
        scale = ref[1] / g.size   # mean |g|
        np.testing.assert_allclose(got[:2], ref[:2], rtol=1e-7, atol=1e-9 * ref[1], err_msg=k)
        np.testing.assert_allclose(got[2:], ref[2:], rtol=1e-6, atol=1e-6 * scale, err_msg=k)


def test_vae_step_against_golden():
    B, d, C, X, lab, cond = inputs()
    u = R.UnetOracle(in_ch=C, seed=1)
    v = R.VaeOracle(u, in_ch=C, d=d, seed=3)
    eps = np.random.default_rng(2).standard_normal((B, 256))
    m_eval = v.test_on_batch(X, cond, eps)
    np.testing.assert_allclose(m_eval, G["vae_eval_metrics"], rtol=1e-9)
    metrics, recon, cache, pmc, taps = v.forward_losses(X, cond, eps, True)
    np.testing.assert_allclose(metrics, G["vae_train_metrics"], rtol=1e-9)
    np.testing.assert_allclose(recon[0, ::5, ::5, ::5, 0], G["vae_train_recon_sample"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(cache["_enc"]["zm"], G["vae_train_zmean"], rtol=1e-8, atol=1e-12)
    grads = v.backward(X, cache, pmc, taps, recon)
    gmax = max(np.abs(g).max() for g in grads.values())
    for k, g in grads.items():
        ref = G["vae_grad__%s" % k.replace("/", "__")]
        got = checks(g)
        # conv biases in front of BatchNorm have an exactly-zero true gradient (rounding noise only)
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-9 * gmax * g.size, err_msg=k)
