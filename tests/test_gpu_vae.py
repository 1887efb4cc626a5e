"""End-to-end parity of the HIP DFC-VAE engine against the fp64 oracle (LatticeDFCVAE graph and
loss, /root/reference/vae/lattice_vae.py:160-270) with eps injected."""
import numpy as np
import pytest

from oracle import numpy_ref as R

pytestmark = pytest.mark.gpu


def _setup(B=2, d=16, C=1, lr=5e-4):
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    uo = R.UnetOracle(in_ch=C, seed=1)
    vo = R.VaeOracle(uo, in_ch=C, d=d, seed=3, lr=lr)
    ue = UnetEngine(in_channels=C, d=d, max_batch=B)
    ue.set_weights(uo.P)
    ve = VaeEngine(ue, in_channels=C, d=d, max_batch=B, lr=lr)
    ve.set_weights(vo.P)
    X, _, cond = R.synthetic_batch(B, d, C, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    eps = np.random.default_rng(2).standard_normal((B, 256))
    return uo, vo, ue, ve, X, cond.astype(np.float64), eps


def test_vae_encode_decode_match_oracle(relerr):
    uo, vo, ue, ve, X, cond, eps = _setup()
    rng = np.random.default_rng(3)
    for k in list(vo.S):
        vo.S[k] = (rng.uniform(0.5, 1.5, vo.S[k].shape) if k.endswith("var")
                   else rng.uniform(-0.2, 0.2, vo.S[k].shape))
        ve.set_tensor(k, vo.S[k])
    zm_r, zlv_r, z_r = vo.predict_encoder(X, cond, eps)
    zm, zlv, z = ve.encode(X, cond, eps)
    assert relerr(zm, zm_r) <= 1e-5 and relerr(zlv, zlv_r) <= 1e-5 and relerr(z, z_r) <= 1e-5
    rec_r = vo.predict_decoder(z_r, cond)
    rec = ve.decode(z_r, cond)
    assert relerr(rec, rec_r) <= 1e-5     # "VAE reconstructions within 1e-5" (north_star)
    m_r = vo.test_on_batch(X, cond, eps)
    m = ve.test_step(X, cond, eps)
    np.testing.assert_allclose(m, m_r, rtol=2e-5)


def test_vae_train_step_matches_oracle(relerr):
    uo, vo, ue, ve, X, cond, eps = _setup()
    m_r = vo.train_on_batch(X, cond, eps)
    m = ve.train_step(X, cond, eps)
    np.testing.assert_allclose(m, m_r, rtol=2e-5)
    gscale = max(np.abs(g).max() for g in vo.last_grads.values())
    for name, shape, trainable in ve.tensor_infos():
        if trainable:
            g = ve.get_grad(name, shape)
            # conv biases in front of BatchNorm have exactly-zero true gradient: compare on the
            # scale of the largest gradient instead of their own (rounding-noise) scale
            err = np.abs(g - vo.last_grads[name]).max() / max(np.abs(vo.last_grads[name]).max(), 1e-6 * gscale)
            assert err <= 5e-4, (name, err)
        else:
            assert relerr(ve.get_tensor(name, shape), vo.S[name]) <= 1e-5, name
    # the frozen perceptual U-Net must be untouched (weights AND moving statistics)
    for name, shape, _ in ue.tensor_infos():
        ref = uo.P[name] if name in uo.P else uo.S[name]
        assert relerr(ue.get_tensor(name, shape), ref) <= 1e-7, name
