"""End-to-end parity of the HIP U-Net engine against the fp64 oracle (AtomUnet graph,
/root/reference/unet/unet.py:272-355; losses/metrics :159-221; Adam) through the C ABI."""
import numpy as np
import pytest

from oracle import numpy_ref as R

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-5       # north_star: outputs within 1e-5 tensor-relative of the CPU reference
GRAD_TOL = 6e-5      # fp32 rounding through 16 conv+BN layers, tensor-relative (measured <= 4e-5: 2x regressions show red)
STEP_TOL = 1e-5


def _setup(B, d, C, ties, seed=1, lr=1e-3):
    from icsg3d_amd.engine import UnetEngine
    orc = R.UnetOracle(in_ch=C, seed=seed, lr=lr, pool_ties=ties)
    eng = UnetEngine(in_channels=C, d=d, max_batch=B, lr=lr, pool_ties=ties)
    eng.set_weights({k: v for k, v in orc.P.items()})
    X, lab, _ = R.synthetic_batch(B, d, C, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    return orc, eng, X, lab


@pytest.mark.parametrize("B,d,C", [(2, 16, 1), (3, 8, 4)])
def test_unet_predict_matches_oracle(B, d, C, relerr):
    orc, eng, X, lab = _setup(B, d, C, "tf_cpu")
    # non-trivial moving statistics so eval-mode BN is exercised
    rng = np.random.default_rng(3)
    for k in list(orc.S):
        orc.S[k] = (rng.uniform(0.5, 1.5, orc.S[k].shape) if k.endswith("var")
                    else rng.uniform(-0.2, 0.2, orc.S[k].shape))
        eng.set_tensor(k, orc.S[k])
    soft_ref, sig_ref = orc.forward(X, training=False)
    soft, sig = eng.predict(X)
    assert relerr(soft, soft_ref) <= FWD_TOL
    assert relerr(sig, sig_ref) <= FWD_TOL
    # argmax labels: bit-exact wherever the top-2 margin exceeds 1e-4
    sp, mk = eng.predict_labels(X, 0.8)
    srt = np.sort(soft_ref, -1)
    clear = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert np.array_equal(sp[clear], soft_ref.argmax(-1)[clear])
    clear_s = np.abs(sig_ref[..., 0] - 0.8) > 1e-4
    assert np.array_equal(mk[clear_s], (sig_ref[..., 0] >= 0.8)[clear_s])
    # (Glorot heads: the mask is empty and f1 = wr = 0 here; tests/test_gpu_metrics.py repeats this with a confident head)
    m_ref = orc.test_on_batch(X, lab)
    m = eng.test_step(X, lab)
    np.testing.assert_allclose(m, m_ref, rtol=1e-4, atol=1e-6)


UNET_LAYERS = ["c1", "c2", "c3", "c4", "c5", "c6", "c9", "c10", "c13", "c14", "c15", "c16", "c17", "c18"]


def _layer_shape(name, B, d):
    res = {"c1": 1, "c2": 1, "c3": 2, "c4": 2, "c5": 4, "c6": 4, "c9": 8, "c10": 8, "c13": 4, "c14": 4,
           "c15": 2, "c16": 2, "c17": 1, "c18": 1}[name]
    cout = dict((n, c) for n, _, c in R.UNET_CONVS)[name]
    S = d // res
    return (B, S, S, S, cout)


# B = 4: at d = 16 the S = 4 layers (c5, c6, c14) then take conv_winog.hip's backward-weight GEMMs too (they need B % 4 == 0)
# fuse: the BatchNorm-backward-in-backward-data fusions (engine.hip dgrad_bnfuse_ok / skip_bnfuse_ok) forced on at this size --
# by default they start at 64 MB of activations, i.e. at the bench's batch -- under both max-pool tie rules (the deferred
# skip launch routes the pooled gradient through the forward's tie masks)
# C = 4: the reference scripts' default input_shape (train_unet.py:84) -- c1 is then a 4-channel convolution
@pytest.mark.parametrize("ties,B,fuse,C", [("tf_cpu", 2, False, 1), ("first", 2, False, 1), ("tf_cpu", 3, False, 1),
                                           ("tf_cpu", 4, False, 1), ("tf_cpu", 2, True, 1), ("first", 2, True, 1),
                                           ("tf_cpu", 4, True, 1), ("tf_cpu", 2, False, 4), ("tf_cpu", 3, True, 4)])
def test_unet_train_step_matches_oracle(ties, B, fuse, C, relerr, monkeypatch):
    """Gradients, BN moving statistics and the Adam update of one train step.

    ReLU'(0) is discontinuous, so the fp64 oracle is evaluated with the engine's own ReLU masks
    (exported activations); the oracle refuses masks that differ anywhere except within 1e-5 of the
    kink, so this pins everything but the sign of sub-rounding pre-activations."""
    d = 16
    lr = 1e-3
    if fuse:
        monkeypatch.setenv("ICSG3D_DGRAD_BNFUSE_MIN", "0")
    orc, eng, X, lab = _setup(B, d, C, ties, lr=lr)
    p0 = {k: v.copy() for k, v in orc.P.items()}
    eng.profile_enable(True)
    m = eng.train_step(X, lab)
    sites = {r["label"].split("|")[0] for r in eng.profile_rows()}
    eng.profile_enable(False)
    fused = {"bnfuse:c18", "bnfuse:c16", "bnfuse:c17.skip", "bnfuse:c2.pool", "bnfuse:c2", "bnfuse:c15.skip", "bnfuse:c4.pool"}
    assert (fused <= sites) if fuse else not (fused & sites), sorted(x for x in sites if x.startswith("bnfuse"))
    kink = {n: eng.get_activation(n, _layer_shape(n, B, d)) for n in UNET_LAYERS}
    affine = {n: eng.get_bn_affine(n, _layer_shape(n, B, d)[-1]) for n in ("c2", "c4", "c6")}
    m_ref = orc.train_on_batch(X, lab, kink=kink, affine=affine)
    print("relu masks flipped within rounding of 0:", sum(orc.kink_flips.values()), "of",
          sum(np.prod(_layer_shape(n, B, d)) for n in UNET_LAYERS))
    np.testing.assert_allclose(m[:3], m_ref[:3], rtol=1e-5)
    np.testing.assert_allclose(m[3:], m_ref[3:], rtol=1e-4, atol=1e-6)
    worst = 0.0
    for name, shape, trainable in eng.tensor_infos():
        if trainable:
            g = eng.get_grad(name, shape)
            e = relerr(g, orc.last_grads[name])
            worst = max(worst, e)
            assert e <= GRAD_TOL, (name, e)
            # Adam (keras 2.3.1 formula) applied to the engine's own gradient, in fp64
            p_exp, _, _ = R.adam_update(p0[name], g.astype(np.float64), 0.0, 0.0, 1, lr)
            assert relerr(eng.get_tensor(name, shape), p_exp) <= 2e-5, name   # fp32 Adam arithmetic
        else:
            assert relerr(eng.get_tensor(name, shape), orc.S[name]) <= STEP_TOL, name
    print("worst grad rel err", worst)
    # a second step exercises Adam's t=2 bias correction and the repacked weights
    eng.set_weights(orc.P)
    m2 = eng.train_step(X, lab)
    m_ref2 = orc.train_on_batch(X, lab)
    np.testing.assert_allclose(m2[:3], m_ref2[:3], rtol=2e-5)


def test_unet_step_is_deterministic():
    B, d, C = 2, 8, 1
    orc, eng, X, lab = _setup(B, d, C, "tf_cpu")
    w0 = eng.get_weights()
    m1 = eng.train_step(X, lab)
    p1 = eng.get_weights()
    eng.set_weights(w0)
    eng.reset_optimizer()
    m2 = eng.train_step(X, lab)
    p2 = eng.get_weights()
    assert np.array_equal(m1, m2)
    for k in p1:
        assert np.array_equal(p1[k], p2[k]), k


def test_profile_filter_restricts_launch_sites():
    """ics_net_profile_filter (bench.py's timed region): events only around the launch sites whose label starts with the
    prefix; an empty prefix brackets every launch again."""
    orc, eng, X, lab = _setup(2, 16, 1, "tf_cpu")
    eng.train_step(X, lab)
    eng.profile_filter("conv_wgrad:")
    eng.profile_enable(True)
    eng.train_step(X, lab)
    rows = eng.profile_rows()
    assert rows and all(r["label"].startswith("conv_wgrad:") for r in rows)
    assert sum(r["launches"] for r in rows) >= 14          # one per conv layer at least
    eng.profile_filter("")
    eng.profile_enable(True)
    eng.train_step(X, lab)
    labels = {r["label"].split("|")[0].split(":")[0] for r in eng.profile_rows()}
    eng.profile_enable(False)
    assert {"conv_fwd", "conv_wgrad", "conv_dgrad", "bn_act_bwd", "adam"} <= labels
