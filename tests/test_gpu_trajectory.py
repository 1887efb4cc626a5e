"""K consecutive train steps, engine and fp64 oracle each from ITS OWN state: Adam at t >= 2 against
the oracle's parameters, not against Adam re-applied to the engine's own gradient, and no re-synchronisation of weights
between steps.  (unet/unet.py:370 fit_generator's inner loop; vae/lattice_vae.py:296 train_on_batch in the epoch loop.)

The ReLU / LeakyReLU / max-pool decisions of each step are pinned to the engine's (oracle.apply_kink refuses any that
differ away from the kink) exactly as in the one-step tests."""
import numpy as np
import pytest

from oracle import numpy_ref as R
from test_gpu_unet import UNET_LAYERS, _layer_shape
from test_gpu_vae import _pm_layer_shapes, _vae_layer_shapes

pytestmark = pytest.mark.gpu

K = 5
# What K un-synchronised Adam steps can and cannot hold (measured on MI355X, round 5):
#  * FOLLOWING oracle (teacher forcing): a second oracle takes the engine's state (parameters, moving statistics, Adam
#    m / v / t) before every step and makes the same step in fp64.  The engine is never re-synchronised; the oracle's
#    Adam at t = 2..K starts from the same state, so its moments after the step differ from the engine's by one step's
#    gradient error only: m, v tensor-relative <= MOMENT_TOL at EVERY step.  This is the precise check of "Adam at t >= 2,
#    moment persistence, repacked weights" against the oracle's formulas.
#  * FREE oracle: its own fp64 state from step 0 on.  Two trajectories separate (measured ~5x per step for the DFC-VAE:
#    e0's BatchNorm scale ~10, lr 5e-4), so its bounds are looser and say so: metrics 1e-4 per step, moments
#    FREE_MOMENT_TOL (measured U-Net 5.8e-4, DFC-VAE 2.9e-3), moving statistics 2e-3 / 2e-4 (measured 4.6e-4 U-Net c9, 16 rows per channel; 6.2e-5 DFC-VAE).
#  * the parameter UPDATE is lr_t m / (sqrt(v) + 1e-7): scale-free in g, so an element with |g| = 1e-3 max|g| carries its
#    gradient's 1e-5-of-max error as a 1e-2 RELATIVE error into the update, and below eps_hat = 1e-7 / sqrt(1 - beta2) =
#    3.2e-6 a gradient error is amplified by lr / eps_hat = 310: in ONE step from identical state the worst single
#    element of the U-Net's update is off by 77 % of lr (measured).  Updates and parameters are therefore compared
#    L2-relative per tensor (one step vs the following oracle <= STEP_UPDATE_TOL, measured 2.0e-3 / 3.1e-3; K free steps
#    <= UPDATE_TOL / PARAM_TOL); the worst element is printed, not asserted.  1e-4 on the parameters is not reachable at this learning rate by ANY fp32
#    implementation -- two fp32 summation orders of the same gradient differ by as much.
MOMENT_TOL = 1e-4
FREE_MOMENT_TOL = {"unet": 2e-3, "vae": 1e-2}
UPDATE_TOL = 2e-2         # free oracle, K steps, L2-relative per tensor
PARAM_TOL = 1e-3          # free oracle, K steps, L2-relative per tensor (single elements: see the header)
STEP_UPDATE_TOL = 5e-3    # following oracle, one step, L2-relative per tensor


def _moments(eng):
    """{tensor name: (m, v)} from the flat Adam buffers (trainable tensors in registration order)."""
    m, v, t = eng.get_optimizer_state()
    out, off = {}, 0
    for name, shape, trainable in eng.tensor_infos():
        if trainable:
            n = int(np.prod(shape))
            out[name] = (m[off:off + n].reshape(shape), v[off:off + n].reshape(shape))
            off += n
    assert off == m.size
    return out, t


def _follow(orc, eng):
    """Teacher forcing: the oracle takes the engine's complete training state."""
    mom, t = _moments(eng)
    for name, shape, trainable in eng.tensor_infos():
        v = eng.get_tensor(name, shape).astype(np.float64)
        if trainable:
            orc.P[name] = v
            orc.m[name], orc.v[name] = (a.astype(np.float64) for a in mom[name])
        else:
            orc.S[name] = v
    orc.t = t


def _check_step_update(eng, before, orc, skip=()):
    """One step's parameter update against the following oracle's: L2-relative per tensor (asserted) and the worst single
    element relative to the largest update (reported: Adam amplifies a gradient error at |g| < eps_hat by lr / eps_hat)."""
    worst, worst_el = 0.0, 0.0
    for name, shape, trainable in eng.tensor_infos():
        if trainable and name not in skip:
            moved = orc.P[name] - before[name]
            diff = (eng.get_tensor(name, shape) - before[name]) - moved
            e2 = float(np.sqrt((diff ** 2).sum() / (moved ** 2).sum()))
            worst, worst_el = max(worst, e2), max(worst_el, float(np.abs(diff).max() / np.abs(moved).max()))
            assert e2 <= STEP_UPDATE_TOL, (name, e2)
    return worst, worst_el


def _check_moments(eng, orc_m, orc_v, K, relerr, skip=(), tol=MOMENT_TOL):
    mom, t = _moments(eng)
    assert t == K
    worst = 0.0
    gm = max(np.abs(a).max() for a in orc_m.values())
    for name, (m, v) in mom.items():
        if name in skip:
            continue
        em = np.abs(m - orc_m[name]).max() / max(np.abs(orc_m[name]).max(), 1e-6 * gm)
        ev = np.abs(v - orc_v[name]).max() / max(np.abs(orc_v[name]).max(), 1e-12 * gm * gm)
        worst = max(worst, em, ev)
        assert em <= tol and ev <= 2 * tol, (name, em, ev)
    return worst


def test_unet_five_step_trajectory(relerr):
    from icsg3d_amd.engine import UnetEngine
    B, d, C, lr = 2, 16, 1, 1e-3
    orc = R.UnetOracle(in_ch=C, seed=1, lr=lr)
    eng = UnetEngine(in_channels=C, d=d, max_batch=B, lr=lr)
    eng.set_weights(orc.P)
    p0 = {k: v.copy() for k, v in orc.P.items()}
    fol = R.UnetOracle(in_ch=C, seed=1, lr=lr)
    rng = np.random.default_rng(5)
    worst_m = worst_f = worst_su = worst_el = 0.0
    for step in range(K):
        X, lab, _ = R.synthetic_batch(B, d, C, seed=step, dtype=np.float64)     # a new batch every step
        X = X + 1e-3 * rng.uniform(size=X.shape)
        _follow(fol, eng)
        before = {k: v.copy() for k, v in fol.P.items()}
        m = eng.train_step(X, lab)
        kink = {n: eng.get_activation(n, _layer_shape(n, B, d)) for n in UNET_LAYERS}
        affine = {n: eng.get_bn_affine(n, _layer_shape(n, B, d)[-1]) for n in ("c2", "c4", "c6")}
        m_fol = fol.train_on_batch(X, lab, kink=kink, affine=affine)
        np.testing.assert_allclose(m[:3], m_fol[:3], rtol=1e-5, err_msg="following oracle, step %d" % step)
        worst_f = max(worst_f, _check_moments(eng, fol.m, fol.v, step + 1, relerr))
        su = _check_step_update(eng, before, fol)
        worst_su, worst_el = max(worst_su, su[0]), max(worst_el, su[1])
        for name in fol.S:
            assert relerr(eng.get_tensor(name), fol.S[name]) <= 1e-5, (step, name)
        m_ref = orc.train_on_batch(X, lab, kink=kink, affine=affine)
        print("step", step, "metrics", m, "oracle", m_ref, "kink flips", sum(orc.kink_flips.values()))
        np.testing.assert_allclose(m[:3], m_ref[:3], rtol=1e-4, err_msg="step %d" % step)
        worst_m = max(worst_m, float(np.abs(m[:3] / m_ref[:3] - 1).max()))
    assert orc.t == K
    worst_mv = _check_moments(eng, orc.m, orc.v, K, relerr, tol=FREE_MOMENT_TOL["unet"])
    print("Adam moments: following oracle, worst of %d steps %.2e (one-step update: L2 %.2e, worst element %.2e); "
          "free oracle after %d steps %.2e" % (K, worst_f, worst_su, worst_el, K, worst_mv))
    worst_p = worst_u = 0.0
    for name, shape, trainable in eng.tensor_infos():
        got = eng.get_tensor(name, shape)
        if trainable:
            # the UPDATE K Adam steps made, to 2 % of its largest element (Adam divides by sqrt(v) + 1e-7: where |g| is below
            # eps_hat = 3.2e-6 a gradient error is amplified by lr / eps_hat, so the update is the honest scale for it) ...
            moved = orc.P[name] - p0[name]
            eu = float(np.sqrt((((got - p0[name]) - moved) ** 2).sum() / (moved ** 2).sum()))
            worst_u = max(worst_u, eu)
            assert eu <= UPDATE_TOL, (name, eu)
            # ... and the parameters themselves (biases and beta start at 0: they ARE the update)
            if np.abs(p0[name]).max() > 0:
                e = float(np.sqrt(((got - orc.P[name]) ** 2).sum() / (orc.P[name] ** 2).sum()))
                worst_p = max(worst_p, e)
                assert e <= PARAM_TOL, (name, e)
        else:
            assert relerr(got, orc.S[name]) <= 2e-3, name    # free oracle, c9 / c10 normalise over 16 rows here (following oracle above: 1e-5 at every step)
    print("5-step U-Net trajectory: worst metric rel err %.2e, Adam moments %.2e, parameters %.2e, update %.2e"
          % (worst_m, worst_mv, worst_p, worst_u))


def test_vae_five_step_trajectory(relerr):
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    B, d, C, lr = 2, 16, 1, 5e-4
    uo = R.UnetOracle(in_ch=C, seed=1)
    vo = R.VaeOracle(uo, in_ch=C, d=d, seed=3, lr=lr)
    ue = UnetEngine(in_channels=C, d=d, max_batch=B)
    ue.set_weights(uo.P)
    ve = VaeEngine(ue, in_channels=C, d=d, max_batch=B, lr=lr)
    ve.set_weights(vo.P)
    p0 = {k: v.copy() for k, v in vo.P.items()}
    fol = R.VaeOracle(uo, in_ch=C, d=d, seed=3, lr=lr)
    bn_bias = {n for n in vo.P if n.endswith("/bias") and (n[:-5] + "/gamma") in vo.P}
    rng = np.random.default_rng(5)
    eps_rng = np.random.default_rng(2)
    sh, shp = _vae_layer_shapes(B, d, C), _pm_layer_shapes(B, d)
    worst_m = worst_f = worst_su = worst_el = 0.0
    for step in range(K):
        X, _, cond = R.synthetic_batch(B, d, C, seed=step, dtype=np.float64)
        X = X + 1e-3 * rng.uniform(size=X.shape)
        cond = cond.astype(np.float64)
        eps = eps_rng.standard_normal((B, 256))
        _follow(fol, ve)
        before = {k: v.copy() for k, v in fol.P.items()}
        m = ve.train_step(X, cond, eps)
        kink = {n: ve.get_activation(n, s) for n, s in sh.items()}
        kink_pm = {n: ue.get_activation(n, s) for n, s in shp.items()}
        aff = {n: ve.get_bn_affine(n, sh[n][-1]) for n in ("e0", "e1", "e2", "e3", "d0", "d1", "d2", "d3", "dout")}
        aff_pm = {n: ue.get_bn_affine(n, shp[n][-1]) for n in ("c2", "c4", "c6")}
        m_fol = fol.train_on_batch(X, cond, eps, kink=kink, kink_pm=kink_pm, affine=aff, affine_pm=aff_pm)
        np.testing.assert_allclose(m, m_fol, rtol=2e-5, err_msg="following oracle, step %d" % step)
        worst_f = max(worst_f, _check_moments(ve, fol.m, fol.v, step + 1, relerr, skip=bn_bias))
        su = _check_step_update(ve, before, fol, skip=bn_bias)
        worst_su, worst_el = max(worst_su, su[0]), max(worst_el, su[1])
        for name in fol.S:
            assert relerr(ve.get_tensor(name), fol.S[name]) <= 1e-5, (step, name)
        # kink_tol 1e-3: e0's BatchNorm scale is ~10 here, so the 1e-6-level parameter differences that K un-synchronised
        # steps accumulate move its LeakyReLU input by up to a few 1e-4 -- the flips are counted and bounded below
        m_r = vo.train_on_batch(X, cond, eps, kink=kink, kink_pm=kink_pm, affine=aff, affine_pm=aff_pm, kink_tol=1e-3)
        flips = sum(vo.kink_flips.values())
        print("step", step, "metrics", m, "oracle", m_r, "kink flips", flips)
        assert flips <= 200, vo.kink_flips
        # measured: 3e-8, 6e-7, 7e-6, 2e-5, 1.1e-4 (KLD) -- the two un-synchronised trajectories separate ~5x per step
        # (lr 5e-4, e0's BatchNorm scale ~10, Adam's eps-normalised update, see the header); the last step gets 3e-4
        np.testing.assert_allclose(m, m_r, rtol=1e-4 if step < K - 1 else 3e-4, err_msg="step %d" % step)
        worst_m = max(worst_m, float(np.abs(m / m_r - 1).max()))
    # (conv biases in front of BatchNorm: true gradient exactly zero, the engine's moments are rounding noise -> skipped)
    worst_mv = _check_moments(ve, vo.m, vo.v, K, relerr, skip=bn_bias, tol=FREE_MOMENT_TOL["vae"])
    print("Adam moments: following oracle, worst of %d steps %.2e (one-step update: L2 %.2e, worst element %.2e); "
          "free oracle after %d steps %.2e" % (K, worst_f, worst_su, worst_el, K, worst_mv))
    worst_p = worst_u = 0.0
    for name, shape, trainable in ve.tensor_infos():
        got = ve.get_tensor(name, shape)
        if trainable:
            scale = np.abs(vo.P[name]).max()
            if name.endswith("/bias") and (name[:-5] + "/gamma") in vo.P:
                # a conv bias in front of BatchNorm has an exactly-zero true gradient; the engine's is rounding noise that Adam
                # normalises (eps_hat = 1e-7 / sqrt(1 - beta2) = 3.2e-6): held on the scale of the same layer's kernel
                scale = max(scale, np.abs(vo.P[name[:-4] + "kernel"]).max())
            e = float(np.sqrt(((got - vo.P[name]) ** 2).mean()) / scale)
            worst_p = max(worst_p, e)
            if np.abs(p0[name]).max() > 0 or (name.endswith("/bias") and (name[:-5] + "/gamma") in vo.P):
                assert e <= PARAM_TOL, (name, e)
            moved = vo.P[name] - p0[name]
            if not (name.endswith("/bias") and (name[:-5] + "/gamma") in vo.P):
                eu = float(np.sqrt((((got - p0[name]) - moved) ** 2).sum() / (moved ** 2).sum()))
                worst_u = max(worst_u, eu)
                assert eu <= UPDATE_TOL, (name, eu)
        else:
            assert relerr(got, vo.S[name]) <= 2e-4, name     # free oracle (following oracle above: 1e-5 at every step)
    for name, shape, _ in ue.tensor_infos():          # the frozen U-Net stays frozen over the whole trajectory (F9)
        ref = uo.P[name] if name in uo.P else uo.S[name]
        assert relerr(ue.get_tensor(name, shape), ref) <= 1e-7, name
    print("5-step DFC-VAE trajectory: worst metric rel err %.2e, Adam moments %.2e, parameters %.2e, update %.2e"
          % (worst_m, worst_mv, worst_p, worst_u))
