"""Finite-difference and semantic checks of the oracle's ops (Keras semantics of SURVEY App. B)."""
import numpy as np

from oracle import numpy_ref as R


def fd(f, x, h=1e-6):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    while not it.finished:
        i = it.multi_index
        old = x[i]
        x[i] = old + h; fp = f()
        x[i] = old - h; fm = f()
        x[i] = old
        g[i] = (fp - fm) / (2 * h)
        it.iternext()
    return g


def test_conv_bn_act_block_gradients():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 4, 4, 4, 3))
    P = {"b/kernel": rng.standard_normal((3, 3, 3, 3, 2)) * 0.3, "b/bias": rng.standard_normal(2),
         "b/gamma": rng.uniform(0.5, 1.5, 2), "b/beta": rng.standard_normal(2)}
    w = rng.standard_normal((2, 4, 4, 4, 2))
    for pre, post in (("relu", None), (None, "lrelu"), (None, "relu")):
        blk = R.Block("b", pre, True, post)

        def loss():
            return float((blk.fwd(x, P, {}, True, {}) * w).sum())
        cache, g = {}, {}
        blk.fwd(x, P, {}, True, cache)
        dx = blk.bwd(w, P, {}, cache, g)
        np.testing.assert_allclose(dx, fd(loss, x), rtol=1e-5, atol=1e-7)
        for k in P:
            np.testing.assert_allclose(g[k], fd(loss, P[k]), rtol=1e-5, atol=1e-6, err_msg=k)


def test_pool_and_upsample():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 4, 4, 4, 2))
    y = R.maxpool_fwd(x)
    assert y.shape == (1, 2, 2, 2, 2)
    dy = rng.standard_normal(y.shape)
    for ties in ("tf_cpu", "first"):   # no ties in random data: both route to the argmax
        dx = R.maxpool_bwd(x, y, dy, ties)
        assert np.isclose(dx.sum(), dy.sum()) and np.count_nonzero(dx) == dy.size
    # exact ties (ReLU-dead voxels behind BN): TF-CPU rule duplicates, "first" does not
    xt = np.zeros((1, 2, 2, 2, 1)); yt = R.maxpool_fwd(xt); dt = np.ones_like(yt)
    assert R.maxpool_bwd(xt, yt, dt, "tf_cpu").sum() == 8 and R.maxpool_bwd(xt, yt, dt, "first").sum() == 1
    u = R.upsample_fwd(y)
    assert u.shape == x.shape and np.array_equal(u[0, :2, :2, :2, 0], np.full((2, 2, 2), y[0, 0, 0, 0, 0]))
    assert np.allclose(R.upsample_bwd(np.ones_like(x)), 8.0)


def test_unet_losses_and_metrics():
    rng = np.random.default_rng(2)
    z = rng.standard_normal((2, 2, 2, 2, 5)) * 3
    lab = rng.integers(0, 5, (2, 2, 2, 2))
    y = R.one_hot(lab, 5)
    p = R.softmax(z)

    def loss():
        return float(R.wcce_loss(y, R.softmax(z), 5.0).mean())
    dz = R.softmax_bwd(p, R.wcce_bwd(y, p, 5.0, np.full(2, 0.5)))
    np.testing.assert_allclose(dz, fd(loss, z), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(dz, 5.0 * (p - y) / (2 * 8), rtol=1e-9)   # closed form used on the GPU
    # clip: a true-class probability below 1e-7 contributes log(1e-7) and no gradient
    z2 = np.zeros((1, 1, 1, 1, 3)); z2[..., 0] = 40.0
    y2 = R.one_hot(np.array([[[[1]]]]), 3)
    assert np.isclose(R.wcce_loss(y2, R.softmax(z2), 3.0)[0], -3.0 * np.log(1e-7))
    assert np.all(R.wcce_bwd(y2, R.softmax(z2), 3.0, np.ones(1)) == 0)
    # metrics: round-half-even => p == 0.5 is not a positive
    yy = np.array([[[[[1.0, 0.0]]]]]); pp = np.array([[[[[0.5, 0.5]]]]])
    assert R.f1_m(yy, pp) == 0.0
    pp = np.array([[[[[0.6, 0.4]]]]])
    assert np.isclose(R.f1_m(yy, pp), 2 * (1 * 1) / (1 + 1 + 1e-7), rtol=1e-6)
    assert R.wr_m(yy, pp) == 0.0      # class 0 carries zero weight


def test_adam_and_bn_moving_update():
    p, m, v = R.adam_update(np.array([1.0]), np.array([0.5]), 0.0, 0.0, 1, 1e-3)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.isclose(p[0], 1.0 - lr_t * 0.05 / (np.sqrt(0.001 * 0.25) + 1e-7))
    mm, mv = R.bn_moving_update(np.zeros(1), np.ones(1), np.array([2.0]), np.array([4.0]), 100)
    assert np.isclose(mm[0], 0.02) and np.isclose(mv[0], 0.99 + 0.01 * 4.0 * 100 / (100 - 1.001))


def test_vae_cond_tiling_and_shapes():
    u = R.UnetOracle(in_ch=4, seed=1)
    v = R.VaeOracle(u, in_ch=4, d=16, seed=3)
    assert v.P["e0/kernel"].shape == (3, 3, 3, 44, 16)        # C + C*cond
    cond = np.eye(10)[[3]]
    t = v.tile_cond(cond, 16)
    assert t.shape == (1, 16, 16, 16, 40) and t[0, 5, 6, 7, 13] == 1.0 and t[0, 0, 0, 0, 12] == 0.0
    assert dict(R.vae_param_shapes(1, d=64))["dec_dense/kernel"] == (266, 2048)  # seed (d/8)^3*4 (F12 extension)


def test_bce_logits_form_matches_clipped_form_away_from_saturation_and_its_gradient():
    """The two candidate semantics of Keras' "binary_crossentropy" on a Sigmoid output: equal to ~1e-7
    relative where p is not saturated, different at saturation; the logits form's gradient is sigmoid(z) - t (checked by
    finite differences)."""
    rng = np.random.default_rng(4)
    z = rng.normal(size=(2, 3, 3, 3, 1)) * 2
    t = (rng.uniform(size=z.shape) < 0.4).astype(np.float64)
    a, b = R.bce_logits_loss(t, z), R.bce_loss(t, R.sigmoid(z))
    np.testing.assert_allclose(a, b, rtol=1e-6)
    zs = np.array([[[[[30.0]]]], [[[[-30.0]]]]])
    ts = np.array([[[[[0.0]]]], [[[[1.0]]]]])
    assert np.all(R.bce_logits_loss(ts, zs) > 29) and np.all(R.bce_loss(ts, R.sigmoid(zs)) < 17)     # clip at 1e-7: -log(1e-7) = 16.1
    w = rng.normal(size=z.shape[:-1])
    g = R.bce_logits_bwd(t, z, w)
    np.testing.assert_allclose(g, fd(lambda: float((R.bce_logits_loss(t, z) * w).sum()), z), rtol=1e-6, atol=1e-9)


def test_numpy_oracle_bce_from_logits_matches_independent_torch_autograd():
    """The switch's oracle path is pinned the way the default path is: the fp64 numpy restatement against torch-CPU autograd
    (oracle/torch_ref.py, an independent implementation) -- metrics and every gradient tensor of one U-Net step, d = 8."""
    from oracle import torch_ref as T
    B, d = 2, 8
    orc = R.UnetOracle(in_ch=1, seed=1, lr=1e-3, pool_ties="first", bce_from_logits=True)
    X, lab, _ = R.synthetic_batch(B, d, 1, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    orc.P["sig/kernel"] = orc.P["sig/kernel"] * 3.0           # logits up to ~ +-10 (torch.logit(sigmoid(z)) stays finite in fp64)
    P0 = {k: v.copy() for k, v in orc.P.items()}
    S0 = {k: v.copy() for k, v in orc.S.items()}
    m_t, g_t, _, _, sig_t = T.unet_step_grads(P0, S0, X, lab, ties="first", bce_from_logits=True)
    m_c, _, _, _, _ = T.unet_step_grads(P0, S0, X, lab, ties="first", bce_from_logits=False)
    m = orc.train_on_batch(X, lab)
    np.testing.assert_allclose(m[:3], m_t, rtol=1e-10)
    zmax = np.abs(np.log(sig_t / (1 - sig_t))).max()
    assert 3 < zmax < 30 and abs(m_t[2] - m_c[2]) > 1e-12 * m_t[2], (zmax, m_t[2], m_c[2])
    for k, g in g_t.items():
        np.testing.assert_allclose(orc.last_grads[k], g, rtol=1e-7, atol=1e-12 * max(np.abs(g).max(), 1e-30) + 1e-14, err_msg=k)
