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


def _vae_layer_shapes(B, d, C):
    sh, S, f = {}, d, (16, 32, 64, 128)
    for i in range(4):
        sh["e%d" % i] = (B, S, S, S, f[i]); S //= 2
    sh["e4"] = (B, S, S, S, 4)
    sh["enc_dense"] = (B, 256)
    S = d // 8
    for i in range(4):
        sh["d%d" % i] = (B, S, S, S, f[3 - i])
        if i < 3:
            S *= 2
    sh["dout"] = (B, d, d, d, C)
    return sh


def _pm_layer_shapes(B, d):
    res = {"c1": 1, "c2": 1, "c3": 2, "c4": 2, "c5": 4, "c6": 4, "c9": 8, "c10": 8}
    cout = dict((n, c) for n, _, c in R.UNET_CONVS)
    return {n: (B, d // r, d // r, d // r, cout[n]) for n, r in res.items()}


# C = 4: the reference scripts' default input_shape (train_vae.py:90): e0 is a real 44-channel convolution (no analytic
# fold of the tiled condition, which exists for C = 1 only), dout has 4 output channels, the frozen U-Net's c1 4 inputs
@pytest.mark.parametrize("B,C", [(2, 1), (3, 1), (5, 1), (2, 4), (3, 4)])
def test_vae_train_step_matches_oracle(B, C, relerr):
    """One DFC-VAE train step: [Loss, PM, MSE, KLD], all gradients, BN statistics, frozen U-Net.
    Activation kinks are pinned to the engine's stored activations (oracle.apply_kink).  Odd batch sizes: ragged
    last tiles, split plans and block-to-sample mappings that do not divide evenly."""
    d = 16
    uo, vo, ue, ve, X, cond, eps = _setup(B, d, C)
    m = ve.train_step(X, cond, eps)
    kink = {n: ve.get_activation(n, s) for n, s in _vae_layer_shapes(B, d, C).items()}
    kink_pm = {n: ue.get_activation(n, s) for n, s in _pm_layer_shapes(B, d).items()}
    aff = {n: ve.get_bn_affine(n, _vae_layer_shapes(B, d, C)[n][-1]) for n in ("e0", "e1", "e2", "e3", "d0", "d1", "d2", "d3", "dout")}
    aff_pm = {n: ue.get_bn_affine(n, _pm_layer_shapes(B, d)[n][-1]) for n in ("c2", "c4", "c6")}
    m_r = vo.train_on_batch(X, cond, eps, kink=kink, kink_pm=kink_pm, affine=aff, affine_pm=aff_pm)
    print("kink flips:", {k: v for k, v in vo.kink_flips.items() if v})
    np.testing.assert_allclose(m, m_r, rtol=2e-5)
    gscale = max(np.abs(g).max() for g in vo.last_grads.values())
    worst = 0.0
    for name, shape, trainable in ve.tensor_infos():
        if trainable:
            g = ve.get_grad(name, shape)
            # conv biases in front of BatchNorm have exactly-zero true gradient: compare on the
            # scale of the largest gradient instead of their own (rounding-noise) scale
            scale = np.abs(vo.last_grads[name]).max()
            if name.endswith("/bias"):   # ... i.e. of the same layer's kernel gradient
                scale = max(scale, np.abs(vo.last_grads[name[:-4] + "kernel"]).max())
            err = np.abs(g - vo.last_grads[name]).max() / max(scale, 1e-6 * gscale)
            worst = max(worst, err)
            assert err <= 1e-4, (name, err)   # measured <= 4e-5
        else:
            assert relerr(ve.get_tensor(name, shape), vo.S[name]) <= 1e-5, name
    print("worst grad rel err", worst)
    # the frozen perceptual U-Net must be untouched (weights AND moving statistics)
    for name, shape, _ in ue.tensor_infos():
        ref = uo.P[name] if name in uo.P else uo.S[name]
        assert relerr(ue.get_tensor(name, shape), ref) <= 1e-7, name
