"""The torch fp64 oracle with the tested implementation's decisions pinned (oracle/torch_ref.py: Pins, _PinnedAct,
_PinnedClamp, maxpool(route=)) against the numpy oracle's apply_kink machinery -- two independent implementations of the same
device.  The pinned torch form is what the B = 32 / d = 64, B = 8 whole-network GPU tests run on the box's host cores
(tests/test_gpu_fullsize_oracle.py: numpy's single-threaded elementwise passes would take minutes at 1 M voxels x 128
channels); here both run at d = 16, B = 2 on decisions that DO differ from the fp64 ones.
Reference graphs: /root/reference/unet/unet.py:272-355, vae/lattice_vae.py:160-270."""
import numpy as np
import pytest

from oracle import numpy_ref as R
from oracle import torch_ref as T

UNET_LAYERS = [n for n, _, _ in R.UNET_CONVS]


def _impl_like(s64, rng, band=5e-5):
    """An 'implementation' activation: the fp64 value rounded to fp32, with the elements inside (0, band) pushed to the other
    side of the kink (what fp32 accumulation does to a handful of them at full size)."""
    s = s64.astype(np.float32).copy()
    near = (np.abs(s64) < band) & (s64 != 0) & (rng.uniform(size=s64.shape) < 0.5)
    s[near] = -s[near] if s64.min() < 0 else 0.0
    return s, int(near.sum())


def _affine(c, P, n):
    inv = P[n + "/gamma"] / np.sqrt(c["var"] + R.BN_EPS)
    return inv.astype(np.float32), (P[n + "/beta"] - c["mean"] * inv).astype(np.float32)


def test_pinned_unet_step_equals_numpy_apply_kink():
    B, d = 2, 16
    X, lab, _ = R.synthetic_batch(B, d, 1, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    orc = R.UnetOracle(in_ch=1, seed=1, lr=1e-3)
    P0, S0 = {k: v.copy() for k, v in orc.P.items()}, {k: v.copy() for k, v in orc.S.items()}
    cache = {}
    orc.forward(X, training=True, cache=cache)
    rng = np.random.default_rng(11)
    kink, moved = {}, 0
    for n in UNET_LAYERS:
        kink[n], k = _impl_like(cache[n]["s"], rng)
        moved += k
    affine = {n: _affine(cache[n], orc.P, n) for n in ("c2", "c4", "c6")}
    clip_pin = {"soft": np.ones((B, d, d, d), bool), "sig": np.ones((B, d, d, d), bool)}
    assert moved > 20, moved

    m_np = orc.train_on_batch(X, lab, kink=kink, affine=affine, clip_pin=clip_pin)
    m_t, g_t, stats, _, _ = T.unet_step_grads(P0, S0, X, lab, kink=kink, affine=affine, clip_pin=clip_pin)
    np.testing.assert_allclose(m_t, m_np[:3], rtol=1e-11)
    np.testing.assert_allclose(T.unet_step_grads.f1_wr, m_np[3:], rtol=1e-9, atol=1e-12)
    assert T.unet_step_grads.flips["kink"] == orc.kink_flips and sum(orc.kink_flips.values()) >= moved // 2
    worst = 0.0
    for k, g in orc.last_grads.items():
        e = np.abs(g_t[k] - g).max() / max(np.abs(g).max(), 1e-30)
        worst = max(worst, e)
        assert e <= 1e-8, (k, e)
    # ... and the pins matter: the un-pinned torch gradients differ from the pinned ones by far more than that
    _, g_free, _, _, _ = T.unet_step_grads(P0, S0, X, lab)
    moved_by = max(np.abs(g_free[k] - g_t[k]).max() / max(np.abs(g_t[k]).max(), 1e-30) for k in g_t)
    print("pinned torch vs pinned numpy: worst %.2e; pinning moves a gradient tensor by up to %.2e" % (worst, moved_by))
    assert moved_by > 100 * worst

    # a decision that differs AWAY from the kink is refused, as apply_kink refuses it
    bad = dict(kink)
    bad["c3"] = kink["c3"].copy()
    idx = np.unravel_index(np.argmax(cache["c3"]["s"]), cache["c3"]["s"].shape)
    bad["c3"][idx] = 0.0
    with pytest.raises(AssertionError, match="differ away from the kink"):
        T.unet_step_grads(P0, S0, X, lab, kink=bad, affine=affine)
    cp = {"sig": clip_pin["sig"].copy()}
    cp["sig"][0, 0, 0, 0] = False
    with pytest.raises(AssertionError, match="clip decisions differ away"):
        T.unet_step_grads(P0, S0, X, lab, clip_pin=cp)


def test_pinned_vae_step_equals_numpy_apply_kink():
    B, d = 2, 16
    X, _, cond = R.synthetic_batch(B, d, 1, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    cond = cond.astype(np.float64)
    eps = np.random.default_rng(2).standard_normal((B, 256))
    uo = R.UnetOracle(in_ch=1, seed=1)
    vo = R.VaeOracle(uo, in_ch=1, d=d, seed=3, lr=5e-4)
    Pv, Sv = {k: v.copy() for k, v in vo.P.items()}, {k: v.copy() for k, v in vo.S.items()}
    _, recon, cache, pmc, _ = vo.forward_losses(X, cond, eps, True)
    rng = np.random.default_rng(12)
    kink, kink_pm, moved = {}, {}, 0
    for blk in vo._all_blocks():
        c = cache[blk.name]
        if blk.pre_act is not None:                 # e4: the stored activation is the post-LeakyReLU value
            kink[blk.name], k = _impl_like(c["s"], rng)
        else:                                       # conv -> BN -> act: the decision is taken on BN(s); move s where BN(s) ~ 0
            s = c["s"].astype(np.float32).copy()
            inv = vo.P[blk.name + "/gamma"] / np.sqrt(c["var"] + R.BN_EPS)
            near = (np.abs(c["bn"]) < 5e-5) & (rng.uniform(size=s.shape) < 0.5)
            s64 = c["s"] - 2 * c["bn"] / inv        # mirror BN(s) through zero
            s[near] = s64[near].astype(np.float32)
            kink[blk.name], k = s, int(near.sum())
        moved += k
    a = cache["_enc"]["a"]
    kink["enc_dense"] = a.astype(np.float32)
    for n in UNET_LAYERS[:8]:
        kink_pm[n], k = _impl_like(pmc[n]["s"], rng)
        moved += k
    aff = {n: _affine(cache[n], vo.P, n) for n in ("e0", "e1", "e2", "e3")}
    aff_pm = {n: _affine(pmc[n], uo.P, n) for n in ("c2", "c4", "c6")}
    assert moved > 20, moved

    m_np = vo.train_on_batch(X, cond, eps, kink=kink, kink_pm=kink_pm, affine=aff, affine_pm=aff_pm)
    m_t, g_t, _, _, _, _ = T.vae_step_grads(Pv, Sv, uo.P, uo.S, X, cond, eps, in_ch=1, d=d, kink=kink, kink_pm=kink_pm,
                                            affine=aff, affine_pm=aff_pm)
    np.testing.assert_allclose(m_t, m_np, rtol=1e-10)
    assert T.vae_step_grads.flips == vo.kink_flips, (T.vae_step_grads.flips, vo.kink_flips)
    assert sum(vo.kink_flips.values()) > 10
    gscale = max(np.abs(g).max() for g in vo.last_grads.values())
    for k, g in vo.last_grads.items():
        e = np.abs(g_t[k] - g).max() / max(np.abs(g).max(), 1e-6 * gscale)
        assert e <= 1e-7, (k, e)


def test_gemm_form_of_the_fp64_convolution_is_the_same_convolution():
    """torch_ref.conv3d takes a GEMM form for large fp64 problems (9 dgemms per layer instead of torch's batch-parallel fp64
    conv3d: 168 s -> see the GPU test's printout for a B = 32 step): same sums as F.conv3d and as the numpy oracle's 27 tap
    products, forward and both gradients, on non-cubic volumes and odd channel counts."""
    import torch
    import torch.nn.functional as F
    torch.manual_seed(0)
    for (B, C, Co, D, H, W) in [(2, 5, 7, 6, 8, 10), (1, 64, 32, 8, 8, 8), (3, 1, 4, 4, 4, 4)]:
        x = torch.randn(B, C, D, H, W, dtype=torch.float64, requires_grad=True)
        w = torch.randn(Co, C, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        b = torch.randn(Co, dtype=torch.float64, requires_grad=True)
        y0 = F.conv3d(x, w, b, padding=1)
        g = torch.randn_like(y0)
        g0 = torch.autograd.grad(y0, (x, w, b), g)
        y1 = T._Conv3dGemm.apply(x, w, b)
        g1 = torch.autograd.grad(y1, (x, w, b), g)
        assert (y0 - y1).abs().max() <= 1e-12 * y0.abs().max()
        for a, c in zip(g0, g1):
            assert (a - c).abs().max() <= 1e-12 * a.abs().max()
        # ... and the numpy oracle's definition (NDHWC, kernel (3,3,3,Cin,Cout))
        yn = R.conv3d_fwd(T.to_n(x), np.transpose(w.detach().numpy(), (2, 3, 4, 1, 0)), b.detach().numpy())
        assert np.abs(T.to_n(y1) - yn).max() <= 1e-12 * np.abs(yn).max()
    # the switch: below the work threshold conv3d() IS F.conv3d, above it the GEMM form; 1x1x1 heads likewise
    x = torch.randn(1, 8, 4, 4, 4, dtype=torch.float64)
    w = torch.randn(8, 8, 3, 3, 3, dtype=torch.float64)
    b = torch.zeros(8, dtype=torch.float64)
    old = T.GEMM_CONV_MIN_WORK
    try:
        T.GEMM_CONV_MIN_WORK = 0
        y_g, h_g = T.conv3d(x, w, b), T.conv1(x, w[:, :, 0:1, 0:1, 0:1].contiguous(), b)
        T.GEMM_CONV_MIN_WORK = float("inf")
        y_f, h_f = T.conv3d(x, w, b), T.conv1(x, w[:, :, 0:1, 0:1, 0:1].contiguous(), b)
    finally:
        T.GEMM_CONV_MIN_WORK = old
    assert (y_g - y_f).abs().max() <= 1e-12 and (h_g - h_f).abs().max() <= 1e-12
