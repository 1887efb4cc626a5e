"""Parity of the HIP Conv3D kernels (fwd / bwd-data / bwd-weight) against the fp64 oracle, through the C ABI
(ics_op_conv3d_*).  Tolerance: tensor-relative 1e-5.  The ops take the path the engine takes
for the shape: Winograd F(2x2x2, 3x3x3) where csrc/conv_wino.hip serves it (3x3x3, S >= 8, Cin and Cout multiples of
32), the 27-tap implicit GEMM or a direct stencil otherwise; every Winograd-served case is ALSO run with
ICSG3D_NO_WINO=1, which keeps the implicit-GEMM kernels covered at the same shapes."""
import os

import numpy as np
import pytest

from oracle import numpy_ref as R

pytestmark = pytest.mark.gpu

TOL = 1e-5

# (B, S, Cin, Cout, k): vector path, multi N-tile, thin/scalar paths, Cout=1, 1x1x1, ragged M tiles
CASES = [
    (2, 8, 32, 64, 3), (1, 16, 64, 128, 3), (1, 8, 128, 256, 3), (2, 8, 192, 128, 3),
    (2, 8, 1, 32, 3), (2, 8, 11, 16, 3), (1, 4, 4, 128, 3), (2, 8, 16, 1, 3), (2, 8, 16, 32, 3),
    (2, 8, 128, 96, 1), (3, 4, 32, 32, 3), (1, 2, 128, 4, 3), (5, 1, 266, 256, 1), (3, 1, 32, 256, 1),
    (2, 4, 512, 512, 3), (2, 8, 32, 4, 3), (1, 16, 64, 1, 3), (3, 4, 8, 2, 3),   # thin-N direct kernels
    # resolutions of the benchmark (S=32) and of the d=64 extension: the dx-reuse kernels' line padding
    (1, 32, 64, 128, 3), (1, 64, 32, 32, 3), (1, 32, 128, 128, 3),
    (1, 64, 64, 128, 3),   # S = 64 through the dx-reuse backward-weight kernel (half-line chunks with halo rows)
    # Winograd edge cases: odd batch with every block on the border (S = 8), one block row per axis, wide Cout
    (3, 8, 32, 32, 3), (5, 8, 64, 32, 3), (1, 16, 32, 96, 3), (2, 16, 96, 64, 3), (1, 8, 256, 512, 3),
    # S = 4 with >= 64 input channels: the Winograd-domain batched GEMMs (conv_winog.hip); backward-weight there needs
    # B % 4 == 0 (the GEMMs' reduction length = 8 B tiles), other batches take the 27-tap kernel for that gradient
    (4, 4, 256, 512, 3), (3, 4, 128, 192, 3), (8, 4, 128, 64, 3), (4, 4, 64, 128, 3),
    # the reference scripts' default input_shape (d,d,d,4) (train_unet.py:84, train_vae.py:90): encoder conv-0 is a real
    # 44-channel convolution (4 + 4*10), U-Net c1 has Cin = 4, decoder_output has Cout = 4
    (2, 16, 44, 16, 3), (2, 32, 4, 32, 3), (2, 16, 16, 4, 3), (1, 32, 44, 16, 3),
]


def _wino(case):
    """True when the default path of this shape is the Winograd kernel (and the suite is not itself run under
    ICSG3D_NO_WINO, in which case there is nothing to A/B)."""
    B, S, Cin, Cout, k = case
    return k == 3 and S >= 8 and Cin % 32 == 0 and Cout % 32 == 0 and not os.environ.get("ICSG3D_NO_WINO")


def _winog(case):
    """True when the default path of this shape is conv_winog.hip (S = 4, Cin >= 64)."""
    B, S, Cin, Cout, k = case
    return (k == 3 and S == 4 and Cin % 32 == 0 and Cin >= 64 and Cout % 64 == 0 and not os.environ.get("ICSG3D_NO_WINO")
            and not os.environ.get("ICSG3D_NO_WINOG"))


def _data(B, S, Cin, Cout, k, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, S, S, S, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, k, Cin, Cout)) / np.sqrt(k ** 3 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dy = rng.standard_normal((B, S, S, S, Cout)).astype(np.float32)
    return x, w, b, dy


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_S%d_%dto%d_k%d" % c)
def test_conv_forward(case, relerr, monkeypatch):
    from icsg3d_amd import engine as E
    x, w, b, _ = _data(*case)
    ref = R.conv3d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    got = E.conv3d_forward(x, w, b, pre_act=0)
    assert relerr(got, ref) <= TOL
    got_relu = E.conv3d_forward(x, w, b, pre_act=1)
    assert relerr(got_relu, np.maximum(ref, 0)) <= TOL
    if _wino(case):
        monkeypatch.setenv("ICSG3D_NO_WINO", "1")
        direct = E.conv3d_forward(x, w, b, pre_act=0)
        assert relerr(direct, ref) <= TOL
        assert not np.array_equal(direct, got)           # really two different kernels
        monkeypatch.delenv("ICSG3D_NO_WINO")
        if case[3] % 64 == 0:                            # the 16-tile x 64-channel Winograd shape is the default there:
            monkeypatch.setenv("ICSG3D_NO_WINO64", "1")  # keep the 32 x 32 kernel covered at the same shapes
            w32 = E.conv3d_forward(x, w, b, pre_act=0)
            assert relerr(w32, ref) <= TOL and not np.array_equal(w32, got)
    if _winog(case):
        monkeypatch.setenv("ICSG3D_NO_WINOG", "1")
        direct = E.conv3d_forward(x, w, b, pre_act=0)
        assert relerr(direct, ref) <= TOL and not np.array_equal(direct, got)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_S%d_%dto%d_k%d" % c)
def test_conv_backward(case, relerr, monkeypatch):
    from icsg3d_amd import engine as E
    x, w, _, dy = _data(*case, seed=1)
    dx_ref, dw_ref, _ = R.conv3d_bwd(x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64))
    dx, dw = E.conv3d_backward(x, w, dy)
    assert relerr(dx, dx_ref) <= TOL
    assert relerr(dw, dw_ref) <= TOL
    if _wino(case):
        monkeypatch.setenv("ICSG3D_NO_WINO", "1")
        dx2, dw2 = E.conv3d_backward(x, w, dy)
        assert relerr(dx2, dx_ref) <= TOL and relerr(dw2, dw_ref) <= TOL
        assert not np.array_equal(dx2, dx) and not np.array_equal(dw2, dw)
        monkeypatch.delenv("ICSG3D_NO_WINO")
        if case[2] % 64 == 0:                            # backward-data: N = Cin
            monkeypatch.setenv("ICSG3D_NO_WINO64", "1")
            dx3, _ = E.conv3d_backward(x, w, dy)
            assert relerr(dx3, dx_ref) <= TOL and not np.array_equal(dx3, dx)
    if _winog(case):
        monkeypatch.setenv("ICSG3D_NO_WINOG", "1")
        dx2, dw2 = E.conv3d_backward(x, w, dy)
        assert relerr(dx2, dx_ref) <= TOL and relerr(dw2, dw_ref) <= TOL
        # backward-data is itself a convolution Cout -> Cin: served when THAT geometry qualifies
        assert np.array_equal(dx2, dx) == (not (case[3] >= 64 and case[3] % 32 == 0 and case[2] % 64 == 0))
        assert np.array_equal(dw2, dw) == (case[0] % 4 != 0)     # backward-weight: the Winograd GEMMs only when B % 4 == 0


def test_pointwise_gemm_large_m_matches_numpy():
    """taps = 1 over 65 536 rows: the plain-GEMM loaders with 128 x 128 tiles and operands requested two chunks ahead
    (the coarse-grid GEMMs of the up-split backward at the bench's batch; smaller problems pick 64 x 64 tiles and never
    reach that loop).  Forward, backward-data and backward-weight against fp64 numpy."""
    from icsg3d_amd import engine as E
    rng = np.random.default_rng(11)
    B, S, Cin, Cout = 2, 32, 128, 256                       # forward K = 128 (4 chunks), backward-data K = 256 (8), both 128-wide N tiles
    x = rng.standard_normal((B, S, S, S, Cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, 1, Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)
    dy = rng.standard_normal((B, S, S, S, Cout)).astype(np.float32)
    y = E.conv3d_forward(x, w)
    X = x.reshape(-1, Cin).astype(np.float64)
    W = w.reshape(Cin, Cout).astype(np.float64)
    DY = dy.reshape(-1, Cout).astype(np.float64)
    ref = X @ W
    assert np.abs(y.reshape(-1, Cout) - ref).max() <= 1e-5 * np.abs(ref).max()
    dx, dw = E.conv3d_backward(x, w, dy)
    rdx, rdw = DY @ W.T, X.T @ DY
    assert np.abs(dx.reshape(-1, Cin) - rdx).max() <= 1e-5 * np.abs(rdx).max()
    assert np.abs(dw.reshape(Cin, Cout) - rdw).max() <= 2e-5 * np.abs(rdw).max()


# ---- the window 2^29 <= B*S^3*C < 2^31: conv_wino.hip's 32-bit ELEMENT offsets still reach, conv_wino64.hip's 32-bit
# BYTE offsets do not.  The weight layout is decided ONCE from the full predicate (size bound included) and handed to
# the launch; deciding it from Cout % 64 alone once packed layout-1 weights for the layout-0 kernel here.
def test_conv_wino_size_window_forward_backward(relerr, monkeypatch):
    from icsg3d_amd import engine as E
    B, S, Cin, Cout = 16, 64, 128, 128
    assert (1 << 29) <= B * S ** 3 * max(Cin, Cout) < (1 << 31)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((B, S, S, S, Cin), dtype=np.float32)
    w = (rng.standard_normal((3, 3, 3, Cin, Cout)) / np.sqrt(27 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = E.conv3d_forward(x, w, b, pre_act=0)
    # fp64 oracle on the first and the last sample (the last one sits past the 2^31-byte mark)
    for i in (0, B - 1):
        ref = R.conv3d_fwd(x[i:i + 1].astype(np.float64), w.astype(np.float64), b.astype(np.float64))
        assert relerr(got[i:i + 1], ref) <= TOL, i
    if not os.environ.get("ICSG3D_NO_WINO"):
        monkeypatch.setenv("ICSG3D_NO_WINO", "1")
        direct = E.conv3d_forward(x, w, b, pre_act=0)
        monkeypatch.delenv("ICSG3D_NO_WINO")
        assert relerr(got, direct) <= 2 * TOL
        assert not np.array_equal(got, direct)
    del got
    # backward-data through the same window (dy -> dx with the tap-flipped, transposed weights)
    dy = rng.standard_normal((B, S, S, S, Cout), dtype=np.float32)
    dx, _ = E.conv3d_backward(x, w, dy)
    for i in (0, B - 1):
        dx_ref, _, _ = R.conv3d_bwd(x[i:i + 1].astype(np.float64), w.astype(np.float64), dy[i:i + 1].astype(np.float64))
        assert relerr(dx[i:i + 1], dx_ref) <= TOL, i


def test_engine_large_max_batch_matches_small(relerr):
    """An engine built for a batch inside the window serves SMALLER batches with the same weight images: one grid
    through UnetEngine(d=64, max_batch=16) must equal the same grid through max_batch=1 (different Winograd kernels:
    to rounding, not bit for bit)."""
    from icsg3d_amd.engine import UnetEngine
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes
    PU = glorot_params(unet_param_shapes(1, 95), 1)
    X, _, _ = synthetic_batch(1, 64, 1, seed=3)
    small = UnetEngine(d=64, max_batch=1)
    small.set_weights(PU)
    soft1, sig1 = small.predict(X)
    del small
    big = UnetEngine(d=64, max_batch=16)
    big.set_weights(PU)
    soft16, sig16 = big.predict(X)
    assert relerr(soft16, soft1) <= TOL and relerr(sig16, sig1) <= TOL
