"""The algebra csrc/conv_wino.hip relies on, checked in numpy against the oracle's direct convolution (CPU only).

Forward / backward-data:  Y = A^T [ (G g G^T) .* (B^T d B) ] A  per axis, F(2,3):
    B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
Backward-weight:          dL/dg = G^T [ sum over tiles (A dY) .* (B^T d) ]
and the two reformulations the kernels use: the z rows of B^T applied once per voxel before tiling (every row of B^T
has exactly two non-zeros), and A's last row built WITHOUT its negation, the sign restored in the G^T contraction."""
import numpy as np

from oracle import numpy_ref as R

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def _tiles(xpad, S):
    """4x4x4 input tiles of the zero-padded grid, one per 2x2x2 output tile: [b, tz, ty, tx, 4, 4, 4, C]."""
    T = S // 2
    out = np.empty((xpad.shape[0], T, T, T, 4, 4, 4, xpad.shape[-1]))
    for tz in range(T):
        for ty in range(T):
            for tx in range(T):
                out[:, tz, ty, tx] = xpad[:, 2 * tz:2 * tz + 4, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
    return out


def _data(B=2, S=8, Cin=3, Cout=4, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, S, S, S, Cin))
    w = rng.standard_normal((3, 3, 3, Cin, Cout))
    dy = rng.standard_normal((B, S, S, S, Cout))
    return x, w, dy


def test_f23_1d():
    rng = np.random.default_rng(1)
    d, g = rng.standard_normal(4), rng.standard_normal(3)
    y = AT @ ((G @ g) * (BT @ d))
    np.testing.assert_allclose(y, [d[0:3] @ g, d[1:4] @ g], rtol=1e-13)


def test_forward_64_frequency_gemms_equal_direct_conv():
    x, w, _ = _data()
    B, S = x.shape[0], x.shape[1]
    ref = R.conv3d_fwd(x, w, np.zeros(w.shape[-1]))
    xpad = np.pad(x, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))
    U = np.einsum("ai,bj,ck,ntyxijkC->ntyxabcC", BT, BT, BT, _tiles(xpad, S))      # B^T d B per axis
    W = np.einsum("ai,bj,ck,ijkCN->abcCN", G, G, G, w)                             # G g G^T
    M = np.einsum("ntyxabcC,abcCN->ntyxabcN", U, W)                                # 64 GEMMs over the channels
    Y = np.einsum("ia,jb,kc,ntyxabcN->ntyxijkN", AT, AT, AT, M)                    # A^T . A
    out = Y.transpose(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, S, S, S, -1)
    np.testing.assert_allclose(out, ref, rtol=1e-11, atol=1e-11)


def test_z_rows_applied_once_per_voxel():
    """Staging-time z combination: plane (tz, fz) = row fz of B^T over the 4 raw z planes of tile row tz; tiling in
    y/x afterwards gives the same U as transforming every tile separately."""
    x, _, _ = _data(B=1, S=8, Cin=2)
    S = 8
    xpad = np.pad(x, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))
    U = np.einsum("ai,bj,ck,ntyxijkC->ntyxabcC", BT, BT, BT, _tiles(xpad, S))
    for tz in range(S // 2):
        planes = np.einsum("ai,niyxC->nayxC", BT, xpad[:, 2 * tz:2 * tz + 4])      # [n, fz, 10, 10, C]
        assert all(np.count_nonzero(r) == 2 for r in BT)                            # two raw planes per combined one
        for ty in range(S // 2):
            for tx in range(S // 2):
                t = planes[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
                u = np.einsum("bj,ck,najkC->nabcC", BT, BT, t)
                np.testing.assert_allclose(u, U[:, tz, ty, tx], rtol=1e-12, atol=1e-12)


def test_backward_data_is_forward_with_flipped_transposed_weights():
    x, w, dy = _data(seed=2)
    dx_ref, _, _ = R.conv3d_bwd(x, w, dy)
    wf = w[::-1, ::-1, ::-1].transpose(0, 1, 2, 4, 3)                              # tap 26 - t, Cin <-> Cout
    np.testing.assert_allclose(R.conv3d_fwd(dy, wf, np.zeros(w.shape[3])), dx_ref, rtol=1e-11, atol=1e-11)


def test_backward_weight_in_the_winograd_domain_with_unnegated_rows():
    x, w, dy = _data(seed=3)
    B, S = x.shape[0], x.shape[1]
    _, dw_ref, _ = R.conv3d_bwd(x, w, dy)
    xpad = np.pad(x, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))
    U = np.einsum("ai,bj,ck,ntyxijkC->ntyxabcC", BT, BT, BT, _tiles(xpad, S))
    dyt = dy.reshape(B, S // 2, 2, S // 2, 2, S // 2, 2, -1).transpose(0, 1, 3, 5, 2, 4, 6, 7)   # [n,t,t,t,2,2,2,N]
    A = AT.T                                                                        # 4x2: [1 0; 1 1; 1 -1; 0 -1]
    # exact form
    V = np.einsum("ai,bj,ck,ntyxijkN->ntyxabcN", A, A, A, dyt)
    dW = np.einsum("ntyxabcC,ntyxabcN->abcCN", U, V)                                # 64 GEMMs over ALL tiles
    np.testing.assert_allclose(np.einsum("ai,bj,ck,abcCN->ijkCN", G, G, G, dW), dw_ref, rtol=1e-10, atol=1e-10)
    # the kernel's form: last row of A without its minus sign, sign restored in the contraction with G
    Ap = A.copy(); Ap[3] = [0, 1]
    Gs = G.copy(); Gs[3] = -G[3]
    Vp = np.einsum("ai,bj,ck,ntyxijkN->ntyxabcN", Ap, Ap, Ap, dyt)
    dWp = np.einsum("ntyxabcC,ntyxabcN->abcCN", U, Vp)
    np.testing.assert_allclose(np.einsum("ai,bj,ck,abcCN->ijkCN", Gs, Gs, Gs, dWp), dw_ref, rtol=1e-10, atol=1e-10)


def test_multiplication_counts():
    assert 4 ** 3 == 64 and 8 * 27 == 216          # per 2x2x2 output tile and (ci, co) pair
    assert abs(64 / 216 - 8 / 27) < 1e-15          # the same ratio as the up-split's parity classes
