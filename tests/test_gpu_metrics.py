"""Non-vacuous parity of the device metrics / thresholds: f1_m, wr_m, the TP / predicted / possible
counts, round-half-even at p = 0.5 and the 0.8 mask, at values that are NOT zero.

Expected values: (1) the reference's own formulas (unet/unet.py:159-221) evaluated on seeded (labels, p) --
tests/golden/loss_golden.npz, "reference formulas, stand-in backend"; (2) the fp64 oracle on a confident head
(oracle/confident_head.py).  Everything goes through the C ABI."""
import os

import numpy as np
import pytest

from oracle import numpy_ref as R
from saturated import metric_counts, saturate_head

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_golden.npz"))
NC = 95


def _identity_head():
    ws = np.zeros((128, NC), np.float32)
    ws[np.arange(NC), np.arange(NC)] = 1.0
    wg = np.zeros(128, np.float32)
    wg[NC] = 1.0
    return ws, np.zeros(NC, np.float32), wg, np.zeros(1, np.float32)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", ["generic", "confident"])
def test_head_kernels_equal_reference_formulas(name, fused):
    """Logits = log p through an identity head: the device softmax returns the fixture's p, and loss / metrics / counts must
    equal what the reference's r_m / p_m / f1_m / wr_m / weighted_categorical_crossentropy(95) returned for it."""
    from icsg3d_amd.engine import unet_head
    lab = GOLD["unet/%s/labels" % name]
    p = GOLD["unet/%s/p" % name]
    M = lab.size
    x = np.zeros((M, 128), np.float32)
    x[:, :NC] = np.log(p.reshape(M, NC))
    zg = np.random.default_rng(3).normal(size=M) * 3
    x[:, NC] = zg
    out, m, sums = unet_head(x, *_identity_head(), lab, mode=1, fused=fused)
    g = lambda k: GOLD["unet/%s/f64/%s" % (name, k)]
    np.testing.assert_allclose(m[1], g("wcce_w95").mean(), rtol=1e-5)
    tp, possible, predicted = g("counts")
    assert np.abs(p - 0.5).min() > 1e-5                      # no probability within fp32 rounding of the threshold
    assert (sums["tp"], sums["predicted"], sums["voxels"]) == (tp, predicted, possible)
    np.testing.assert_allclose(m[3], g("f1_m"), rtol=1e-5, atol=0)
    np.testing.assert_allclose(m[4], g("wr_m"), rtol=1e-5, atol=0)
    if name == "confident":
        assert m[3] > 0.05 and m[4] > 0.05
    # the sigmoid head's loss: keras binary_crossentropy, target = labels != 0 (unet/data.py:87)
    t = (lab.reshape(M) != 0).astype(np.float64)
    np.testing.assert_allclose(m[2], R.bce_loss(t[:, None], R.sigmoid(zg.astype(np.float32).astype(np.float64))[:, None]).mean(),
                               rtol=1e-5)
    probs, _, _ = unet_head(x, *_identity_head(), lab, mode=0, fused=fused)
    np.testing.assert_allclose(probs[:, :NC], p.reshape(M, NC), rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("fused", [True, False])
def test_round_half_even_on_the_device(fused):
    """K.round is round-half-to-even (the 'halves' fixture: 0.5 does not count, 0.5 + 2^-20 does).  Two classes with equal
    logits and the rest at -200 give p = 0.5 EXACTLY in fp32; a true-class logit raised by 2^-9 gives p > 0.5."""
    from icsg3d_amd.engine import unet_head
    rng = np.random.default_rng(9)
    M = 256
    lab = rng.integers(0, NC, size=M).astype(np.uint8)
    other = (lab.astype(np.int64) + 1 + rng.integers(0, NC - 1, size=M)) % NC
    up = rng.uniform(size=M) < 0.4
    z = np.full((M, NC), -200.0)
    z[np.arange(M), lab] = np.where(up, 2.0 ** -9, 0.0)
    z[np.arange(M), other] = 0.0
    x = np.zeros((M, 128), np.float32)
    x[:, :NC] = z
    probs, _, _ = unet_head(x, *_identity_head(), lab, mode=0, fused=fused)
    pt = probs[np.arange(M), lab]
    assert np.all(pt[~up] == 0.5) and np.all(pt[up] > 0.5)            # exactly one half on the device
    _, m, sums = unet_head(x, *_identity_head(), lab, mode=1, fused=fused)
    p64 = R.softmax(z)
    y = R.one_hot(lab, NC)
    assert sums["tp"] == up.sum() == np.round(np.clip(y * p64, 0, 1)).sum()
    assert sums["predicted"] == up.sum()                              # the other class sits at exactly (or below) one half
    assert sums["wr_tp"] == (up & (lab != 0)).sum() and sums["wr_possible"] == (lab != 0).sum()
    np.testing.assert_allclose(m[3], R.f1_m(y, p64), rtol=1e-6)
    np.testing.assert_allclose(m[4], R.wr_m(y, p64), rtol=1e-6)


def _setup(B, d, C, training, seed=1):
    from icsg3d_amd.engine import UnetEngine
    orc = R.UnetOracle(in_ch=C, seed=seed, lr=1e-3)
    X, lab, _ = R.synthetic_batch(B, d, C, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    if not training:
        rng = np.random.default_rng(3)
        for k in list(orc.S):
            orc.S[k] = (rng.uniform(0.5, 1.5, orc.S[k].shape) if k.endswith("var") else rng.uniform(-0.2, 0.2, orc.S[k].shape))
    saturate_head(orc, X, lab, training)
    eng = UnetEngine(in_channels=C, d=d, max_batch=B, lr=1e-3)
    eng.set_weights(orc.P)
    for k, v in orc.S.items():
        eng.set_tensor(k, v)
    return orc, eng, X, lab


def _check_counts(sums, lab, soft_ref):
    """Counts equal as integers; a probability within 2e-5 of one half may round either way in fp32 (there are <= 3).
    Returns (reference counts, True when every count is equal)."""
    ref, near = metric_counts(lab, soft_ref, margin=2e-5)
    for k in ("tp", "predicted", "wr_tp", "wr_possible", "voxels"):
        assert abs(sums[k] - ref[k]) <= near, (k, sums[k], ref[k], near)
    assert near <= 3
    return ref, all(sums[k] == ref[k] for k in ref)


@pytest.mark.parametrize("C", [1, 4])
def test_unet_eval_metrics_and_mask_with_confident_head(C, relerr, monkeypatch):
    B, d = 2, 16
    orc, eng, X, lab = _setup(B, d, C, training=False)
    soft_ref, sig_ref = orc.forward(X, training=False)
    m_ref = orc.test_on_batch(X, lab)
    assert m_ref[3] > 0.05 and m_ref[4] > 0.05, m_ref               # not the 0 == 0 regime
    m = eng.test_step(X, lab)
    np.testing.assert_allclose(m[:3], m_ref[:3], rtol=1e-5)
    ref, exact = _check_counts(eng.metric_sums(), lab, soft_ref)
    assert ref["tp"] > 1000 and ref["wr_tp"] > 10
    np.testing.assert_allclose(m[3:], m_ref[3:], rtol=1e-5 if exact else 1e-3)
    soft, sig = eng.predict(X)
    assert relerr(soft, soft_ref) <= 1e-5 and relerr(sig, sig_ref) <= 1e-5
    sp, mk = eng.predict_labels(X, 0.8)
    want = sig_ref[..., 0] >= 0.8
    assert want.any() and mk.any() and 0.005 < want.mean() < 0.5    # the threshold separates something
    clear_s = np.abs(sig_ref[..., 0] - 0.8) > 1e-4
    assert np.array_equal(mk[clear_s].astype(bool), want[clear_s]) and clear_s.mean() > 0.999
    srt = np.sort(soft_ref, -1)
    clear = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert np.array_equal(sp[clear], soft_ref.argmax(-1)[clear]) and clear.mean() > 0.99
    assert len(np.unique(sp)) >= 4                                   # several species predicted, not one constant label
    # the labels come out of the fused head's registers (mode 2: no probability tensor): bit-identical to np.argmax / the
    # threshold of the probabilities the same engine returns, and to the stored-probabilities path (ICSG3D_NO_HEAD_LABELS)
    assert np.array_equal(sp, soft.argmax(-1)) and np.array_equal(mk.astype(bool), sig[..., 0] >= 0.8)
    from icsg3d_amd.engine import UnetEngine
    monkeypatch.setenv("ICSG3D_NO_HEAD_LABELS", "1")
    two_step = UnetEngine(in_channels=C, d=d, max_batch=B)
    monkeypatch.delenv("ICSG3D_NO_HEAD_LABELS")
    two_step.set_weights({**orc.P, **orc.S})
    two_step.profile_enable(True)
    sp2, mk2 = two_step.predict_labels(X, 0.8)
    assert "labels" in {r["label"].split("|")[0] for r in two_step.profile_rows()}
    eng.profile_enable(True)
    eng.predict_labels(X, 0.8)
    assert "labels" not in {r["label"].split("|")[0] for r in eng.profile_rows()}
    assert np.array_equal(sp, sp2) and np.array_equal(mk, mk2)


@pytest.mark.parametrize("C", [1, 4])
def test_unet_train_step_metrics_with_confident_head(C):
    """train_on_batch's logged metrics (training-mode BN) with TP > 0: f1 / wr and the counts behind them."""
    B, d = 2, 16
    orc, eng, X, lab = _setup(B, d, C, training=True)
    cache = {}
    soft_ref, _ = orc.forward(X, training=True, cache=cache)
    m_ref = orc.loss_and_metrics(soft_ref, cache["_head"]["sig"], lab)
    assert m_ref[3] > 0.05 and m_ref[4] > 0.05, m_ref
    m = eng.train_step(X, lab)
    np.testing.assert_allclose(m[:3], m_ref[:3], rtol=1e-5)
    ref, exact = _check_counts(eng.metric_sums(), lab, soft_ref)
    assert ref["tp"] > 1000 and ref["wr_tp"] > 10
    np.testing.assert_allclose(m[3:], m_ref[3:], rtol=1e-5 if exact else 1e-3)


@pytest.mark.parametrize("fused", [True, False])
def test_bce_from_logits_switch_on_the_device(fused):
    """ics_unet_config.bce_from_logits: TF 2.1's sigmoid_cross_entropy_with_logits short-circuit (confidence
    M) next to the default clipped-probability form.  Logits up to +-25: the two forms differ there (the clip bounds the
    per-voxel loss at 16.1 and zeroes the gradient), and each must equal its oracle."""
    from icsg3d_amd.engine import unet_head
    rng = np.random.default_rng(12)
    M = 512
    lab = np.where(rng.uniform(size=M) < 0.4, rng.integers(1, NC, size=M), 0).astype(np.uint8)
    t = (lab != 0).astype(np.float64)[:, None]
    zg = rng.normal(size=M) * 4
    zg[:40] = np.where(t[:40, 0] > 0, -25.0, 25.0)                   # confidently WRONG voxels: p clipped, loss 16.1 vs 25
    x = np.zeros((M, 128), np.float32)
    x[:, :NC] = rng.normal(size=(M, NC))
    x[:, NC] = zg
    z64 = x[:, NC].astype(np.float64)[:, None]
    dz_l, m_l, _ = unet_head(x, *_identity_head(), lab, mode=2, fused=fused, bce_from_logits=True)
    dz_c, m_c, _ = unet_head(x, *_identity_head(), lab, mode=2, fused=fused, bce_from_logits=False)
    ref_l = R.bce_logits_loss(t, z64).mean()
    ref_c = R.bce_loss(t, R.sigmoid(z64)).mean()
    assert ref_l > 1.2 * ref_c                                       # the forms really differ on this input
    np.testing.assert_allclose(m_l[2], ref_l, rtol=1e-5)
    # the clipped form AT saturation is an fp32 statement: the upper bound 1 - 1e-7 is 1 - 2^-23 in fp32, so a clipped
    # voxel's -log(1 - p) is 15.94 there and 16.12 in fp64 (Keras computes it in fp32 too).  Held to the same formula
    # evaluated in fp32 numpy; against the fp64 value it is within 0.5 %.
    f = np.float32
    p32 = (f(1) / (f(1) + np.exp(-x[:, NC:NC + 1]))).astype(f)
    pc = np.clip(p32, f(1e-7), f(1) - f(1e-7))
    t32 = t.astype(f)
    ref_c32 = (-(t32 * np.log(pc) + (f(1) - t32) * np.log(f(1) - pc))).astype(np.float64).mean()
    np.testing.assert_allclose(m_c[2], ref_c32, rtol=2e-5)
    np.testing.assert_allclose(m_c[2], ref_c, rtol=5e-3)
    np.testing.assert_allclose(m_l[1], m_c[1], rtol=1e-7)            # the softmax head does not care
    g_l = R.bce_logits_bwd(t, z64, np.full(M, 1.0 / M))[:, 0]
    # (tensor-relative: sigmoid(z) - t of a nearly-right confident voxel is a difference of two fp32 numbers close to 1)
    assert np.abs(dz_l[:, NC] - g_l).max() <= 1e-6 * np.abs(g_l).max()
    assert np.all(dz_c[:40, NC] == 0) and np.all(np.abs(dz_l[:40, NC] * M) > 0.99)     # clip kink vs |sigmoid - t| ~ 1
    np.testing.assert_array_equal(dz_l[:, :NC], dz_c[:, :NC])


def test_unet_train_step_with_bce_from_logits_matches_oracle(relerr):
    """One full train step under the logits form of the sigmoid head's loss: metrics and every gradient tensor."""
    from icsg3d_amd.engine import UnetEngine
    from test_gpu_unet import UNET_LAYERS, _layer_shape
    B, d, C = 2, 16, 1
    orc = R.UnetOracle(in_ch=C, seed=1, lr=1e-3, bce_from_logits=True)
    X, lab, _ = R.synthetic_batch(B, d, C, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    saturate_head(orc, X, lab, True, sharp_sig=8.0)                  # sigmoid logits up to ~20: saturation is exercised
    eng = UnetEngine(in_channels=C, d=d, max_batch=B, lr=1e-3, bce_from_logits=True)
    eng.set_weights(orc.P)
    m = eng.train_step(X, lab)
    kink = {n: eng.get_activation(n, _layer_shape(n, B, d)) for n in UNET_LAYERS}
    affine = {n: eng.get_bn_affine(n, _layer_shape(n, B, d)[-1]) for n in ("c2", "c4", "c6")}
    m_ref = orc.train_on_batch(X, lab, kink=kink, affine=affine)
    np.testing.assert_allclose(m[:3], m_ref[:3], rtol=2e-5)
    worst = 0.0
    for name, shape, trainable in eng.tensor_infos():
        if trainable:
            e = relerr(eng.get_grad(name, shape), orc.last_grads[name])
            worst = max(worst, e)
            assert e <= 1e-4, (name, e)
    print("bce_from_logits train step: worst gradient error %.2e" % worst)
