"""The reference's OWN loss / metric / sampling formulas as the expected values.

tests/golden/loss_golden.npz holds what the FunctionDefs of /root/reference/unet/unet.py:159-221 and
/root/reference/vae/lattice_vae.py:53-66,232-270 return on seeded inputs -- the reference's code, extracted with `ast` and
executed in the build container with a numpy namespace standing in for `keras.backend`
(tests/golden/make_loss_golden.py: "reference formulas, stand-in backend").  Here the oracle's restatements
(oracle/numpy_ref.py) and the product's numpy mirrors (icsg3d_amd/unet/unet.py, icsg3d_amd/vae/lattice_vae.py) are held
to those values; tests/test_gpu_metrics.py holds the device kernels to the same fixture."""
import os

import numpy as np
import pytest

from oracle import numpy_ref as R

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_golden.npz"))
CASES = ["generic", "confident", "saturated", "halves"]
NC = 95


def _case(name):
    lab = GOLD["unet/%s/labels" % name]
    p = GOLD["unet/%s/p" % name]
    return lab, (lab[..., None] == np.arange(NC)).astype(np.float64), p


@pytest.mark.parametrize("name", CASES)
def test_oracle_unet_loss_and_metrics_equal_reference_formulas(name):
    lab, y, p = _case(name)
    g = lambda k: GOLD["unet/%s/f64/%s" % (name, k)]
    np.testing.assert_allclose(R.wcce_loss(y, p, 95.0), g("wcce_w95"), rtol=1e-13)
    np.testing.assert_allclose(R.wcce_loss(y, p, np.linspace(0.5, 2.0, NC)), g("wcce_wvec"), rtol=1e-13)
    np.testing.assert_allclose(R.f1_m(y, p), g("f1_m"), rtol=1e-13, atol=0)
    np.testing.assert_allclose(R.wr_m(y, p), g("wr_m"), rtol=1e-13, atol=0)
    if name != "generic":
        assert g("f1_m") > 0.05 and g("wr_m") > 0.05            # the fixture is not the 0 == 0 regime


@pytest.mark.parametrize("name", CASES)
def test_product_mirrors_equal_reference_formulas(name):
    from icsg3d_amd.unet import unet as U
    lab, y, p = _case(name)
    g = lambda k: GOLD["unet/%s/f64/%s" % (name, k)]
    np.testing.assert_allclose(U.weighted_categorical_crossentropy(95)(y, p), g("wcce_w95"), rtol=1e-13)
    np.testing.assert_allclose(U.weighted_categorical_crossentropy(np.linspace(0.5, 2.0, NC))(y, p), g("wcce_wvec"),
                               rtol=1e-13)
    for fn in ("r_m", "p_m", "f1_m", "wr_m"):
        np.testing.assert_allclose(getattr(U, fn)(y, p), g(fn), rtol=1e-13, atol=0, err_msg=fn)


def test_round_half_even_corner_is_in_the_fixture():
    """'halves': a true-class probability of exactly 0.5 does NOT count as a true positive (K.round is half-to-even),
    0.5 + 2^-20 does, and 1.5 (un-normalised rows) is clipped to 1 first."""
    lab, y, p = _case("halves")
    pt = np.take_along_axis(p, lab[..., None].astype(np.int64), -1)[..., 0]
    exact_half = int((pt == 0.5).sum())
    assert exact_half > 20 and int((pt == 1.5).sum()) > 20 and int((pt > 1.5).sum()) > 20
    tp, possible, predicted = GOLD["unet/halves/f64/counts"]
    assert possible == lab.size and tp == lab.size - exact_half
    # f32 evaluation of the same formulas gives the same integer counts (the corner cases are exact in fp32)
    np.testing.assert_array_equal(GOLD["unet/halves/f32/counts"], GOLD["unet/halves/f64/counts"])


def test_oracle_vae_losses_equal_reference_formulas():
    zm, zlv, eps, x, rec = (GOLD["vae/" + k] for k in ("zm", "zlv", "eps", "x", "rec"))
    names = ["re_lu_2", "re_lu_4", "re_lu_6", "re_lu_8"]
    w = GOLD["vae/pm_weights"]
    B = x.shape[0]
    taps = lambda t: [(t.reshape(B, -1) @ GOLD["vae/map/" + n]).reshape(B, 2, 2, -1) for n in names]
    np.testing.assert_allclose(R.sampling(zm, zlv, eps), GOLD["vae/f64/z"], rtol=1e-14)
    mse, kld = R.mse_loss(x, rec), R.kld_loss(zm, zlv)
    pm = R.perceptual_from_taps(taps(x), taps(rec), w)
    np.testing.assert_allclose(mse, GOLD["vae/f64/mse"], rtol=1e-13)
    np.testing.assert_allclose(kld, GOLD["vae/f64/kld"], rtol=1e-13)
    np.testing.assert_allclose(pm, GOLD["vae/f64/pm"], rtol=1e-12)
    np.testing.assert_allclose(R.vae_dfc_loss(mse, pm, kld, 0.5, 3e-4), GOLD["vae/f64/loss"], rtol=1e-13)
    # the reference's fp32 evaluation of the same formulas sits within fp32 rounding of the fp64 one
    np.testing.assert_allclose(GOLD["vae/f32/loss"], GOLD["vae/f64/loss"], rtol=2e-6)
    from icsg3d_amd.vae.lattice_vae import sampling
    np.testing.assert_allclose(sampling([zm, zlv], eps), GOLD["vae/f64/z"], rtol=1e-14)
