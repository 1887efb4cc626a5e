"""Full width at the reference scripts' DEFAULT input_shape (d, d, d, 4) (train_unet.py:84, train_vae.py:90, generate.py:123;
SURVEY F6): one U-Net and one DFC-VAE train step at d = 32, B = 2, C = 4 against the fp64 numpy oracle run live with the
engine's ReLU / LeakyReLU / pool / clip decisions pinned -- every gradient tensor.  At C = 4 the encoder's conv-0 is a real
44-channel convolution at 32^3 (no analytic fold of the tiled condition: that exists for C = 1 only), the U-Net's c1 has four
input channels (the padded-channel loaders instead of the single-channel stencil), decoder_output has four output channels.
No committed fixture (the torch-fp64 fixtures are C = 1): the oracle is the same restatement the C = 1 fixtures pin."""
import numpy as np
import pytest

from oracle import numpy_ref as R
from test_gpu_fullwidth import (COUT, GRAD_TOL, MAX_FLIP_FRAC, UNET_LAYERS, VAE_GRAD_TOL, _grad_err, _pm_shapes, _ushape,
                                _vae_shapes)

pytestmark = pytest.mark.gpu

B, d, C = 2, 32, 4


def _inputs():
    X, lab, cond = R.synthetic_batch(B, d, C, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    eps = np.random.default_rng(2).standard_normal((B, 256))
    return X, lab, cond.astype(np.float64), eps


def test_unet_step_full_width_four_channels():
    from icsg3d_amd.engine import UnetEngine
    X, lab, _, _ = _inputs()
    lr = 1e-3
    orc = R.UnetOracle(in_ch=C, seed=1, lr=lr)
    eng = UnetEngine(in_channels=C, d=d, max_batch=B, lr=lr)
    eng.set_weights(orc.P)
    m = eng.train_step(X, lab)
    grads = {name: eng.get_grad(name, shape) for name, shape, tr in eng.tensor_infos() if tr}
    kink = {n: eng.get_activation(n, _ushape(n, B, d)) for n in UNET_LAYERS}
    affine = {n: eng.get_bn_affine(n, COUT[n]) for n in ("c2", "c4", "c6")}
    dz_eng = eng.get_activation("head", (B, d, d, d, 96))
    clip_pin = {"sig": dz_eng[..., 95] != 0, "soft": np.any(dz_eng[..., :95] != 0, axis=-1)}
    m_ref = orc.train_on_batch(X, lab, kink=kink, affine=affine, clip_pin=clip_pin)
    np.testing.assert_allclose(m[:3], m_ref[:3], rtol=1e-5)
    flips = sum(orc.kink_flips.values())
    total = sum(int(np.prod(_ushape(n, B, d))) for n in UNET_LAYERS)
    assert flips <= max(8, MAX_FLIP_FRAC * total), (flips, total)
    worst = 0.0
    for name, g in grads.items():
        e = _grad_err(g, orc.last_grads[name], orc.last_grads, name)
        worst = max(worst, e)
        assert e <= GRAD_TOL, (name, e)
    assert grads["c1/kernel"].shape == (3, 3, 3, 4, 32)
    print("C=4 d=32: worst pinned U-Net gradient error %.2e, %d of %d decisions pinned" % (worst, flips, total))


def test_vae_step_full_width_four_channels():
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    X, _, cond, eps = _inputs()
    uo = R.UnetOracle(in_ch=C, seed=1)
    vo = R.VaeOracle(uo, in_ch=C, d=d, seed=3, lr=5e-4)
    ue = UnetEngine(in_channels=C, d=d, max_batch=B)
    ue.set_weights(uo.P)
    ve = VaeEngine(ue, in_channels=C, d=d, max_batch=B, lr=5e-4)
    ve.set_weights(vo.P)
    m = ve.train_step(X, cond, eps)
    grads = {name: ve.get_grad(name, shape) for name, shape, tr in ve.tensor_infos() if tr}
    vs, ps = _vae_shapes(B, d), _pm_shapes(B, d)
    vs["dout"] = (B, d, d, d, C)
    kink = {n: ve.get_activation(n, s) for n, s in vs.items()}
    kink_pm = {n: ue.get_activation(n, s) for n, s in ps.items()}
    aff = {n: ve.get_bn_affine(n, vs[n][-1]) for n in ("e0", "e1", "e2", "e3", "d0", "d1", "d2", "d3", "dout")}
    aff_pm = {n: ue.get_bn_affine(n, ps[n][-1]) for n in ("c2", "c4", "c6")}
    m_ref = vo.train_on_batch(X, cond, eps, kink=kink, kink_pm=kink_pm, affine=aff, affine_pm=aff_pm)
    np.testing.assert_allclose(m, m_ref, rtol=3e-5)
    assert grads["e0/kernel"].shape == (3, 3, 3, 44, 16) and grads["dout/kernel"].shape == (3, 3, 3, 16, 4)
    gscale = max(np.abs(g).max() for g in vo.last_grads.values())
    worst = 0.0
    for name, g in grads.items():
        scale = np.abs(vo.last_grads[name]).max()
        if name.endswith("/bias"):
            scale = max(scale, np.abs(vo.last_grads[name[:-4] + "kernel"]).max())
        e = float(np.abs(g - vo.last_grads[name]).max() / max(scale, 1e-6 * gscale))
        worst = max(worst, e)
        assert e <= VAE_GRAD_TOL[d], (name, e)
    print("C=4 d=32: worst pinned DFC-VAE gradient error %.2e" % worst)
