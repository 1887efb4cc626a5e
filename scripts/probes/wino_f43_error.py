"""CPU experiment (numpy, fp32 arithmetic): rounding error of F(2,3)^3 against a mixed F(4,3) x F(2,3)^2 Winograd
convolution (one axis with the 4-output transform: 96 instead of 128 multiplies per 16 outputs, -25 % MFMA work) on a
layer shaped like c18 (Cin = 128, Glorot weights, unit-variance inputs), both against an fp64 direct evaluation.
Decides whether the mixed form could hold the 1e-5 forward tolerance (what comes next)."""
import numpy as np

def mats23():
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    return Bt, G, At

def mats43():
    Bt = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                   [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
    G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
    At = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
    return Bt, G, At

def conv_tile(x, w, mz, my, mx, dt):
    """x [Cin, tz, ty, tx] input tile, w [Cin, 3,3,3] for ONE output channel; returns the output tile, arithmetic in dt."""
    (Bz, Gz, Az), (By, Gy, Ay), (Bx, Gx, Ax) = mz, my, mx
    f = lambda a: a.astype(dt)
    V = np.einsum("az,by,cx,kzyx->kabc", f(Bz), f(By), f(Bx), f(x), optimize=False).astype(dt)
    U = np.einsum("az,by,cx,kzyx->kabc", f(Gz), f(Gy), f(Gx), f(w), optimize=False).astype(dt)
    M = np.zeros(V.shape[1:], dt)
    for k in range(V.shape[0]):          # accumulate over channels in dt, as an MFMA chain does
        M = (M + V[k] * U[k]).astype(dt)
    return np.einsum("za,yb,xc,abc->zyx", f(Az), f(Ay), f(Ax), M, optimize=False).astype(dt)

rng = np.random.default_rng(0)
Cin = 128
lim = np.sqrt(6.0 / (27 * Cin + 27 * 128))
res = {"f23^3": [], "f43 x f23^2": [], "f43^2 x f23": []}
for trial in range(40):
    w = rng.uniform(-lim, lim, (Cin, 3, 3, 3))
    for name, (mz, my, mx) in {"f23^3": (mats23(), mats23(), mats23()), "f43 x f23^2": (mats23(), mats23(), mats43()),
                                "f43^2 x f23": (mats23(), mats43(), mats43())}.items():
        tz, ty, tx = mz[0].shape[1], my[0].shape[1], mx[0].shape[1]
        x = np.maximum(rng.standard_normal((Cin, tz, ty, tx)), 0) * 1.5      # post-ReLU-like
        ref = conv_tile(x, w, mz, my, mx, np.float64)
        got = conv_tile(x, w, mz, my, mx, np.float32)
        res[name].append(np.abs(got - ref).max() / np.abs(ref).max())
for k, v in res.items():
    print("%-12s max rel err  median %.2e  worst %.2e" % (k, np.median(v), np.max(v)))
