"""
ORACLE (test infrastructure, NOT product code) -- numpy restatement of the ICSG3D hot path.

  ** parity unpinned **  The reference's arithmetic lives in Keras 2.3.1 / TensorFlow 2.1.x
  (requirements.txt:43,103), neither importable here; the reference ships no tests, golden
  vectors or weights.  This file restates the graphs and losses that
  /root/reference/unet/unet.py and /root/reference/vae/lattice_vae.py spell out, plus the
  Keras/TF op semantics of SURVEY.md Appendix B (recalled, each switchable where graded
  "medium").  It is pinned only against an independent torch-CPU implementation
  (oracle/torch_ref.py, fixtures in tests/golden/) and finite differences.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (icsg3d_amd/) never does.

All tensors are channels-last NDHWC: (B, D, H, W, C); conv kernels are (3,3,3,Cin,Cout)
(Keras layout, cross-correlation, zero "same" padding).  dtype defaults to float64.
"""
from __future__ import annotations

import numpy as np

K_EPSILON = 1e-7          # keras.backend.epsilon()
BN_EPS = 1e-3             # keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99        # keras BatchNormalization default momentum
LEAKY_ALPHA = 0.3         # keras LeakyReLU default alpha (vae/lattice_vae.py:175)
POOL_TIE_TOL = 1e-5       # TF CPU MaxPool3DGrad "equal to max" tolerance (see maxpool_bwd)


# --------------------------------------------------------------------------------------
# Conv3D  (keras.layers.Conv3D(kernel_size=3|1, padding="same"); unet/unet.py:276-352,
#          vae/lattice_vae.py:173,178,213,219)
# --------------------------------------------------------------------------------------
def _pad1(x):
    return np.pad(x, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))


def conv3d_fwd(x, w, b):
    """y[b,z,y,x,co] = b[co] + sum_{dz,dy,dx,ci} x[b,z+dz-1,y+dy-1,x+dx-1,ci] * w[dz,dy,dx,ci,co]."""
    k = w.shape[0]
    B, D, H, W, Cin = x.shape
    Cout = w.shape[-1]
    if k == 1:
        return (x.reshape(-1, Cin) @ w.reshape(Cin, Cout)).reshape(B, D, H, W, Cout) + b
    xp = _pad1(x)
    y = np.zeros((B * D * H * W, Cout), dtype=x.dtype)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                xs = xp[:, dz:dz + D, dy:dy + H, dx:dx + W, :].reshape(-1, Cin)
                y += xs @ w[dz, dy, dx]
    return y.reshape(B, D, H, W, Cout) + b


def conv3d_bwd(x, w, dy, need_dx=True):
    """Returns (dx, dw, db) of conv3d_fwd."""
    k = w.shape[0]
    B, D, H, W, Cin = x.shape
    Cout = w.shape[-1]
    dyf = dy.reshape(-1, Cout)
    db = dyf.sum(0)
    if k == 1:
        dw = (x.reshape(-1, Cin).T @ dyf).reshape(1, 1, 1, Cin, Cout)
        dx = (dyf @ w.reshape(Cin, Cout).T).reshape(x.shape) if need_dx else None
        return dx, dw, db
    xp = _pad1(x)
    dw = np.zeros_like(w)
    dxp = np.zeros_like(xp) if need_dx else None
    for dz in range(3):
        for dy_ in range(3):
            for dx_ in range(3):
                xs = xp[:, dz:dz + D, dy_:dy_ + H, dx_:dx_ + W, :].reshape(-1, Cin)
                dw[dz, dy_, dx_] = xs.T @ dyf
                if need_dx:
                    dxp[:, dz:dz + D, dy_:dy_ + H, dx_:dx_ + W, :] += (
                        dyf @ w[dz, dy_, dx_].T).reshape(B, D, H, W, Cin)
    dx = dxp[:, 1:-1, 1:-1, 1:-1, :] if need_dx else None
    return dx, dw, db


# --------------------------------------------------------------------------------------
# Activations
# --------------------------------------------------------------------------------------
def act_fwd(x, kind):
    """kind: None | "relu" | "lrelu" (alpha=0.3)."""
    if kind is None:
        return x
    if kind == "relu":
        return np.maximum(x, 0)
    if kind == "lrelu":
        return np.where(x > 0, x, LEAKY_ALPHA * x)
    raise ValueError(kind)


def act_bwd(x, dy, kind):
    """x is the activation INPUT (or, equivalently for these monotone acts, its output)."""
    if kind is None:
        return dy
    if kind == "relu":
        return dy * (x > 0)
    if kind == "lrelu":
        return dy * np.where(x > 0, 1.0, LEAKY_ALPHA)
    raise ValueError(kind)


# --------------------------------------------------------------------------------------
# BatchNormalization (axis=-1, eps=1e-3, momentum=0.99); SURVEY Appendix B
#   train: mean,var = tf.nn.moments(x, (0,1,2,3))  (biased var, two-pass)
#          y = x*inv + (beta - mean*inv), inv = gamma*rsqrt(var+eps)
#   moving update: moving = moving*0.99 + batch*0.01, the variance fed to it rescaled
#          var * n/(n-(1+eps))   (Keras 2.3.x quirk, confidence M -> switch bn_unbias)
# --------------------------------------------------------------------------------------
def bn_train_fwd(x, gamma, beta, eps=BN_EPS, moments=None):
    """moments: optional callable x -> (mean, biased var) of the GLOBAL batch (SyncBN under data parallelism:
    icsg3d_amd.dataparallel.syncbn_moments); default = this process's batch, like the single-process reference."""
    C = x.shape[-1]
    xf = x.reshape(-1, C)
    if moments is not None:
        mean, var = moments(x)
    else:
        mean = xf.mean(0)
        var = ((xf - mean) ** 2).mean(0)
    rstd = 1.0 / np.sqrt(var + eps)
    y = x * (gamma * rstd) + (beta - mean * gamma * rstd)
    return y, mean, var


def bn_eval_fwd(x, gamma, beta, mmean, mvar, eps=BN_EPS):
    inv = gamma / np.sqrt(mvar + eps)
    return x * inv + (beta - mmean * inv)


def bn_train_bwd(x, gamma, mean, var, dy, eps=BN_EPS, allsum=None):
    """Returns (dx, dgamma, dbeta) for training-mode BN (gradient flows through batch stats).
    allsum: optional callable (vector, count) -> (sum over ranks, total count) -- SyncBN: dx uses the sums over the
    GLOBAL batch, dgamma / dbeta stay this rank's sums (the gradient all-reduce adds the ranks later)."""
    C = x.shape[-1]
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    dyf = dy.reshape(-1, C)
    xh = xhat.reshape(-1, C)
    dbeta = dyf.sum(0)
    dgamma = (dyf * xh).sum(0)
    n = dyf.shape[0]
    s1, s2 = dbeta, dgamma
    if allsum is not None:
        both, n = allsum(np.concatenate([dbeta, dgamma]), n)
        s1, s2 = both[:C], both[C:]
    dx = (gamma * rstd) * (dy - s1 / n - xhat * (s2 / n))
    return dx, dgamma, dbeta


def bn_eval_bwd(gamma, mvar, dy, eps=BN_EPS):
    return dy * (gamma / np.sqrt(mvar + eps))


def bn_moving_update(mmean, mvar, mean, var, n, momentum=BN_MOMENTUM, eps=BN_EPS, unbias=True):
    v = var * (n / (n - (1.0 + eps))) if unbias else var
    return mmean * momentum + mean * (1 - momentum), mvar * momentum + v * (1 - momentum)


# --------------------------------------------------------------------------------------
# MaxPool3D(2) / UpSampling3D(2)
# --------------------------------------------------------------------------------------
def _windows(x):
    B, D, H, W, C = x.shape
    # (B, D/2, H/2, W/2, 8, C), window scan order (dz, dy, dx) with dx fastest
    xw = x.reshape(B, D // 2, 2, H // 2, 2, W // 2, 2, C).transpose(0, 1, 3, 5, 2, 4, 6, 7)
    return xw.reshape(B, D // 2, H // 2, W // 2, 8, C)


def _unwindows(xw):
    B, D2, H2, W2, _, C = xw.shape
    x = xw.reshape(B, D2, H2, W2, 2, 2, 2, C).transpose(0, 1, 4, 2, 5, 3, 6, 7)
    return x.reshape(B, D2 * 2, H2 * 2, W2 * 2, C)


def maxpool_fwd(x):
    return _windows(x).max(axis=4)


def maxpool_bwd(x, y, dy, ties="tf_cpu", route=None):
    """
    route: optional tensor to take the routing decision from instead of x (tests pass the tested
        implementation's own fp32 pool input, see apply_kink: the decision is a discontinuity).
    ties="tf_cpu": TensorFlow's CPU MaxPool3DGrad (pooling_ops_3d.cc, recalled; confidence M)
        routes dy to EVERY window element with |x - max| < 1e-5 (Eigen select, no argmax).
    ties="first": cuDNN-like, only the first maximal element in (dz,dy,dx) scan order.
    The two differ whenever ReLU-dead voxels tie inside a window (common in the U-Net, where
    pooling follows BN(ReLU(.)) ), and then only through the BN-backward sums.
    """
    xw = _windows(x if route is None else route)
    if route is not None:
        y = xw.max(axis=4)
    if ties == "tf_cpu":
        mask = (np.abs(xw - y[..., None, :]) < POOL_TIE_TOL).astype(x.dtype)
    elif ties == "first":
        am = xw.argmax(axis=4)  # numpy argmax returns the first maximal index
        mask = (np.arange(8).reshape(1, 1, 1, 1, 8, 1) == am[..., None, :]).astype(x.dtype)
    else:
        raise ValueError(ties)
    return _unwindows(mask * dy[..., None, :])


def upsample_fwd(x):
    return x.repeat(2, axis=1).repeat(2, axis=2).repeat(2, axis=3)


def upsample_bwd(dy):
    return _windows(dy).sum(axis=4)


# --------------------------------------------------------------------------------------
# Dense
# --------------------------------------------------------------------------------------
def dense_fwd(x, w, b):
    return x @ w + b


def dense_bwd(x, w, dy):
    return dy @ w.T, x.T @ dy, dy.sum(0)


# --------------------------------------------------------------------------------------
# U-Net heads, losses and metrics  (unet/unet.py:159-221, 339-352)
# --------------------------------------------------------------------------------------
def softmax(z):
    e = np.exp(z - z.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def one_hot(labels, n):
    return (labels[..., None] == np.arange(n)).astype(np.float64)


def wcce_loss(y_onehot, p, weights):
    """weighted_categorical_crossentropy (unet/unet.py:211-219) -> (B,) ; weights scalar or (C,)."""
    q = p / p.sum(-1, keepdims=True)
    qc = np.clip(q, K_EPSILON, 1 - K_EPSILON)
    loss = -(y_onehot * np.log(qc) * weights).sum(-1)
    return loss.mean(axis=(1, 2, 3))


def _pinned_inside(inside, v, pin, tol, what):
    """K.clip's zero-gradient region is a kink like ReLU'(0): `pin` (the decisions of the implementation under test)
    replaces `inside` -- but only where the clipped quantity is within `tol` (relative to the bound's distance from the
    nearer end of [0, 1]) of a bound; a decision that differs anywhere else is an error.  Returns (inside, flips)."""
    pin = np.asarray(pin, bool).reshape(inside.shape)
    diff = pin != inside.astype(bool)
    near = (np.abs(v - K_EPSILON) <= tol * K_EPSILON) | (np.abs((1 - v) - K_EPSILON) <= tol * K_EPSILON)
    if np.any(diff & ~near):
        raise AssertionError("%s: %d clip decisions differ away from the bounds" % (what, int(np.sum(diff & ~near))))
    return pin.astype(inside.dtype), int(diff.sum())


def wcce_bwd(y_onehot, p, weights, dloss, pin=None, pin_tol=0.5, flips=None):
    """d(sum_b dloss[b]*loss[b])/dp through the renormalisation and the clip (zero grad when clipped).
    pin: optional per-voxel decisions "the true class's probability is inside the clip range" of the implementation
    under test (see _pinned_inside)."""
    S = p.sum(-1, keepdims=True)
    q = p / S
    inside = ((q >= K_EPSILON) & (q <= 1 - K_EPSILON)).astype(p.dtype)
    if pin is not None:
        # only the true class's entry carries a gradient: pin that one
        qt = (q * y_onehot).sum(-1, keepdims=True)
        it = ((qt >= K_EPSILON) & (qt <= 1 - K_EPSILON)).astype(p.dtype)
        it, nf = _pinned_inside(it, qt, pin, pin_tol, "softmax clip")
        inside = np.where(y_onehot > 0, it, inside)
        if flips is not None:
            flips["soft_clip"] = nf
    nvox = np.prod(p.shape[1:4])
    dq = -(y_onehot * weights) / np.clip(q, K_EPSILON, 1 - K_EPSILON) * inside
    dq = dq * (dloss.reshape(-1, 1, 1, 1, 1) / nvox)
    # q = p/S : dp_j = dq_j/S - sum_i dq_i p_i / S^2
    return dq / S - (dq * p).sum(-1, keepdims=True) / (S * S)


def softmax_bwd(p, dp):
    return p * (dp - (dp * p).sum(-1, keepdims=True))


def bce_loss(t, p):
    """keras 'binary_crossentropy' (SURVEY App. B): mean over last axis -> (B,D,H,W); p clipped."""
    pc = np.clip(p, K_EPSILON, 1 - K_EPSILON)
    return (-(t * np.log(pc) + (1 - t) * np.log(1 - pc))).mean(-1)


def bce_logits_loss(t, z):
    """tf.nn.sigmoid_cross_entropy_with_logits, which tf.keras.backend.binary_crossentropy (TF 2.1) substitutes when the
    prediction's producer op is a Sigmoid (confidence M): max(z, 0) - z t + log1p(exp(-|z|)), mean over the
    last axis like bce_loss."""
    return (np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))).mean(-1)


def bce_logits_bwd(t, z, dl):
    """d bce_logits_loss / dz = sigmoid(z) - t: no clip, no kink."""
    return (sigmoid(z) - t) * dl[..., None] / z.shape[-1]


def bce_bwd(t, p, dl, pin=None, pin_tol=0.5, flips=None):
    """dl has the shape of bce_loss's output.  pin: optional clip decisions of the implementation under test."""
    pc = np.clip(p, K_EPSILON, 1 - K_EPSILON)
    inside = ((p >= K_EPSILON) & (p <= 1 - K_EPSILON)).astype(p.dtype)
    if pin is not None:
        inside, nf = _pinned_inside(inside, p, pin, pin_tol, "sigmoid clip")
        if flips is not None:
            flips["sig_clip"] = nf
    return (-(t / pc) + (1 - t) / (1 - pc)) * inside * dl[..., None] / p.shape[-1]


def f1_m(y, p):
    """unet/unet.py:159-193; K.round is round-half-to-even (np.round likewise)."""
    tp = np.round(np.clip(y * p, 0, 1)).sum()
    possible = np.round(np.clip(y, 0, 1)).sum()
    predicted = np.round(np.clip(p, 0, 1)).sum()
    precision = tp / (predicted + K_EPSILON)
    recall = tp / (possible + K_EPSILON)
    return 2 * ((precision * recall) / (precision + recall + K_EPSILON))


def wr_m(y, p):
    w = np.ones(y.shape[-1])
    w[0] = 0.0
    tp = np.round(np.clip(w * y * p, 0, 1)).sum()
    possible = np.round(np.clip(w * y, 0, 1)).sum()
    return tp / (possible + K_EPSILON)


# --------------------------------------------------------------------------------------
# Adam (keras.optimizers.Adam 2.3.1)
# --------------------------------------------------------------------------------------
def adam_update(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=K_EPSILON):
    """t is the 1-based step count. Returns (p, m, v)."""
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + eps)
    return p, m, v


# --------------------------------------------------------------------------------------
# Parameter initialisation: Glorot-uniform from PCG64(seed), bias 0, BN identity.
# The order of draws defines the synthetic weights used by tests and bench alike.
# --------------------------------------------------------------------------------------
def glorot(rng, shape):
    if len(shape) == 5:
        rf = shape[0] * shape[1] * shape[2]
        fan_in, fan_out = rf * shape[3], rf * shape[4]
    else:
        fan_in, fan_out = shape
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


# U-Net layer table: name, Cin (None = input channels), Cout   (unet/unet.py:276-336)
UNET_CONVS = [
    ("c1", None, 32), ("c2", 32, 64), ("c3", 64, 64), ("c4", 64, 128), ("c5", 128, 128),
    ("c6", 128, 256), ("c9", 256, 512), ("c10", 512, 512), ("c13", 768, 512), ("c14", 512, 256),
    ("c15", 384, 256), ("c16", 256, 128), ("c17", 192, 128), ("c18", 128, 128),
]
PM_TAPS = ("c2", "c4", "c6", "c10")   # re_lu_2/4/6/8  (vae/lattice_vae.py:100)


def unet_param_shapes(in_ch=1, num_classes=95):
    shapes = []
    for name, cin, cout in UNET_CONVS:
        cin = in_ch if cin is None else cin
        shapes += [(name + "/kernel", (3, 3, 3, cin, cout)), (name + "/bias", (cout,)),
                   (name + "/gamma", (cout,)), (name + "/beta", (cout,))]
    shapes += [("soft/kernel", (1, 1, 1, 128, num_classes)), ("soft/bias", (num_classes,)),
               ("sig/kernel", (1, 1, 1, 128, 1)), ("sig/bias", (1,))]
    return shapes


def init_params(shapes, seed, dtype=np.float64):
    rng = np.random.Generator(np.random.PCG64(seed))
    p = {}
    for name, shp in shapes:
        if name.endswith("/kernel"):
            p[name] = glorot(rng, shp).astype(dtype)
        elif name.endswith("/gamma"):
            p[name] = np.ones(shp, dtype)
        else:
            p[name] = np.zeros(shp, dtype)
    return p


def init_bn_state(shapes, dtype=np.float64):
    s = {}
    for name, shp in shapes:
        if name.endswith("/gamma"):
            base = name[:-len("/gamma")]
            s[base + "/moving_mean"] = np.zeros(shp, dtype)
            s[base + "/moving_var"] = np.ones(shp, dtype)
    return s


# --------------------------------------------------------------------------------------
# Generic conv block:   s = pre_act(conv(x)+b);  o = post_act(BN(s))
#   U-Net block  : pre_act=relu, BN, post_act=None      (unet/unet.py:276-278  Conv->ReLU->BN)
#   VAE block    : pre_act=None, BN, post_act=lrelu     (lattice_vae.py:173-175 Conv->BN->LeakyReLU)
#   decoder tail : pre_act=None, BN, post_act=relu      (lattice_vae.py:219-226)
#   e4           : pre_act=lrelu, no BN                 (lattice_vae.py:178-179)
# --------------------------------------------------------------------------------------
class Block:
    def __init__(self, name, pre_act, has_bn, post_act):
        self.name, self.pre_act, self.has_bn, self.post_act = name, pre_act, has_bn, post_act

    def fwd(self, x, P, S, training, cache):
        n = self.name
        y = conv3d_fwd(x, P[n + "/kernel"], P[n + "/bias"])
        s = act_fwd(y, self.pre_act)
        c = {"x": x, "s": s}
        if self.has_bn:
            if training:
                bn, mean, var = bn_train_fwd(s, P[n + "/gamma"], P[n + "/beta"], moments=cache.get("_sync_moments"))
                c["mean"], c["var"] = mean, var
            else:
                bn = bn_eval_fwd(s, P[n + "/gamma"], P[n + "/beta"],
                                 S[n + "/moving_mean"], S[n + "/moving_var"])
            c["bn"] = bn
            o = act_fwd(bn, self.post_act)
        else:
            o = s
        c["training"] = training
        cache[n] = c
        return o

    def bwd(self, do, P, S, cache, grads, need_dx=True, param_grads=True):
        n = self.name
        c = cache[n]
        # "kink_s"/"kink_bn": activations of the implementation under test; they decide act'(.)
        # where the fp64 value is within rounding of the kink (see apply_kink)
        if self.has_bn:
            dbn = act_bwd(c.get("kink_bn", c["bn"]), do, self.post_act)
            if c["training"]:
                ds, dg, dbt = bn_train_bwd(c["s"], P[n + "/gamma"], c["mean"], c["var"], dbn,
                                           allsum=cache.get("_sync_allsum"))
            else:
                ds = bn_eval_bwd(P[n + "/gamma"], S[n + "/moving_var"], dbn)
                xh = (c["s"] - S[n + "/moving_mean"]) / np.sqrt(S[n + "/moving_var"] + BN_EPS)
                C = dbn.shape[-1]
                dg, dbt = (dbn * xh).reshape(-1, C).sum(0), dbn.reshape(-1, C).sum(0)
            if param_grads:
                grads[n + "/gamma"], grads[n + "/beta"] = dg, dbt
        else:
            ds = do
        dy = act_bwd(c.get("kink_s", c["s"]), ds, self.pre_act)
        dx, dw, db = conv3d_bwd(c["x"], P[n + "/kernel"], dy, need_dx=need_dx)
        if param_grads:
            grads[n + "/kernel"], grads[n + "/bias"] = dw, db
        return dx

    def moving_update(self, S, cache, unbias=True):
        if not self.has_bn:
            return
        c = cache[self.name]
        nel = c["s"].size // c["s"].shape[-1]
        mm, mv = bn_moving_update(S[self.name + "/moving_mean"], S[self.name + "/moving_var"],
                                  c["mean"], c["var"], nel, unbias=unbias)
        S[self.name + "/moving_mean"], S[self.name + "/moving_var"] = mm, mv


def apply_kink(blocks, cache, P, kink, tol=1e-4, affine=None):
    """Pin activation-derivative decisions at the kinks of ReLU / LeakyReLU to the implementation
    under test.  act'(0) is a discontinuity: among ~1e6 pre-activations a few lie within fp32
    accumulation error of 0 and an fp32 implementation legitimately masks them differently from
    fp64; each flip moves a gradient sum by O(max|dy|) (~1e-3 relative here).  kink = {layer: the
    implementation's stored activations s}.  Raises if a sign differs where BOTH values are farther
    than `tol` (absolute; forward parity is 1e-5 of max|s|) from 0.  Returns flips per layer.
    affine = {layer: (scale32, shift32)}: the implementation's fp32 BatchNorm affine; with it the
    implementation's pool input o = act(fma(s, scale, shift)) is reproduced in fp32 and stored as
    cache[layer]["impl_o"], which maxpool_bwd then uses for the (equally discontinuous) routing."""
    flips = {}
    for blk in blocks:
        n = blk.name
        if n not in kink or n not in cache:
            continue
        c = cache[n]
        s_ref = c["s"]
        s_impl = np.asarray(kink[n], s_ref.dtype).reshape(s_ref.shape)
        pairs = []
        if blk.pre_act is not None:
            pairs.append((s_ref, s_impl))
            c["kink_s"] = s_impl
        o32 = None
        if affine is not None and n in affine:
            sc, sh = affine[n]
            o32 = (s_impl.astype(np.float64) * sc.astype(np.float64) + sh.astype(np.float64)).astype(np.float32)
            c["impl_o"] = act_fwd(o32, blk.post_act).astype(s_ref.dtype)
        if blk.has_bn and blk.post_act is not None:
            if not c["training"]:
                raise ValueError("kink pinning is for training-mode steps")
            if o32 is not None:
                # the implementation's own fp32 BatchNorm output decides (a value rebuilt with the oracle's statistics can
                # land on the other side of zero when it is within ~1e-7 of it)
                bn_impl = o32.astype(s_ref.dtype)
            else:
                inv = P[n + "/gamma"] / np.sqrt(c["var"] + BN_EPS)
                bn_impl = s_impl * inv + (P[n + "/beta"] - c["mean"] * inv)
            pairs.append((c["bn"], bn_impl))
            c["kink_bn"] = bn_impl
        cnt = 0
        for ref, impl in pairs:
            diff = (impl > 0) != (ref > 0)
            bad = diff & (np.minimum(np.abs(ref), np.abs(impl)) > tol)
            if bad.any():
                raise AssertionError("activation signs of %s differ away from the kink (%d elements)"
                                     % (n, int(bad.sum())))
            cnt += int(diff.sum())
        flips[n] = cnt
    return flips


# --------------------------------------------------------------------------------------
# AtomUnet graph (unet/unet.py:272-355)
# --------------------------------------------------------------------------------------
class UnetOracle:
    metric_names = ["Loss", "lsoft", "lsig", "f1", "wr"]   # unet/unet.py:250

    def __init__(self, in_ch=1, num_classes=95, seed=1, lr=1e-6, dtype=np.float64,
                 pool_ties="tf_cpu", bn_unbias=True, loss_weight=None, bce_from_logits=False):
        self.in_ch, self.num_classes, self.lr, self.dtype = in_ch, num_classes, lr, dtype
        self.bce_from_logits = bce_from_logits      # the sig head's loss: clipped probabilities (default) or TF's logits form
        self.pool_ties, self.bn_unbias = pool_ties, bn_unbias
        # unet.py:253 passes the INTEGER num_classes as "weights" -> scalar 95.0
        self.loss_weight = float(num_classes) if loss_weight is None else loss_weight
        self.shapes = unet_param_shapes(in_ch, num_classes)
        self.P = init_params(self.shapes, seed, dtype)
        self.S = init_bn_state(self.shapes, dtype)
        self.blocks = {n: Block(n, "relu", True, None) for n, _, _ in UNET_CONVS}
        self.t = 0
        self.m = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.P.items()}

    # -- trunk up to the last conv block; returns c18 output and the cache
    def _trunk(self, x, training, cache, upto=None):
        b, P, S = self.blocks, self.P, self.S
        f = lambda n, t: b[n].fwd(t, P, S, training, cache)
        c1 = f("c1", x); c2 = f("c2", c1); p1 = maxpool_fwd(c2)
        c3 = f("c3", p1); c4 = f("c4", c3); p2 = maxpool_fwd(c4)
        c5 = f("c5", p2); c6 = f("c6", c5); p3 = maxpool_fwd(c6)
        c9 = f("c9", p3); c10 = f("c10", c9)
        cache["_o"] = {"c2": c2, "c4": c4, "c6": c6, "p1": p1, "p2": p2, "p3": p3}
        if upto == "c10":
            return c10
        u1 = upsample_fwd(c10)
        c13 = f("c13", np.concatenate([c6, u1], -1)); c14 = f("c14", c13)
        u3 = upsample_fwd(c14)
        c15 = f("c15", np.concatenate([c4, u3], -1)); c16 = f("c16", c15)
        u4 = upsample_fwd(c16)
        c17 = f("c17", np.concatenate([c2, u4], -1)); c18 = f("c18", c17)
        return c18

    def forward(self, x, training=False, cache=None):
        cache = {} if cache is None else cache
        x = np.asarray(x, self.dtype)
        c18 = self._trunk(x, training, cache)
        zs = conv3d_fwd(c18, self.P["soft/kernel"], self.P["soft/bias"])
        zg = conv3d_fwd(c18, self.P["sig/kernel"], self.P["sig/bias"])
        soft, sig = softmax(zs), sigmoid(zg)
        cache["_head"] = {"c18": c18, "soft": soft, "sig": sig, "zs": zs, "zg": zg}
        return soft, sig

    predict = forward

    def loss_and_metrics(self, soft, sig, labels, zg=None):
        """zg: the sigmoid head's logits, needed when bce_from_logits (forward() leaves them in cache["_head"]["zg"])."""
        y = one_hot(labels, self.num_classes)
        t = (labels != 0).astype(self.dtype)[..., None]
        lsoft = wcce_loss(y, soft, self.loss_weight).mean()
        if self.bce_from_logits:
            zg = np.log(sig) - np.log1p(-sig) if zg is None else zg
            lsig = bce_logits_loss(t, zg).mean()
        else:
            lsig = bce_loss(t, sig).mean()
        return np.array([lsoft + lsig, lsoft, lsig, f1_m(y, soft), wr_m(y, soft)])

    def backward(self, labels, cache, clip_pin=None):
        """Gradients of Loss = mean_b(wcce) + mean(bce) w.r.t. every trainable parameter.
        clip_pin: optional {"soft": bool (B,D,H,W), "sig": bool (B,D,H,W)} K.clip decisions of the implementation under
        test (the clip's zero-gradient region is a kink, like ReLU'(0))."""
        P, S, b = self.P, self.S, self.blocks
        clip_pin = clip_pin or {}
        self.clip_flips = {}
        h = cache["_head"]
        soft, sig, c18 = h["soft"], h["sig"], h["c18"]
        B = soft.shape[0]
        y = one_hot(labels, self.num_classes)
        t = (labels != 0).astype(self.dtype)[..., None]
        g = {}
        pin_soft = clip_pin.get("soft")
        dsoft = wcce_bwd(y, soft, self.loss_weight, np.full(B, 1.0 / B),
                         pin=None if pin_soft is None else np.asarray(pin_soft)[..., None], flips=self.clip_flips)
        dzs = softmax_bwd(soft, dsoft)
        pin_sig = clip_pin.get("sig")
        if self.bce_from_logits:
            dzg = bce_logits_bwd(t, h["zg"], np.full(sig.shape[:-1], 1.0 / sig[..., 0].size))
        else:
            dsig = bce_bwd(t, sig, np.full(sig.shape[:-1], 1.0 / sig[..., 0].size),
                           pin=None if pin_sig is None else np.asarray(pin_sig)[..., None], flips=self.clip_flips)
            dzg = dsig * sig * (1 - sig)
        d1, g["soft/kernel"], g["soft/bias"] = conv3d_bwd(c18, P["soft/kernel"], dzs)
        d2, g["sig/kernel"], g["sig/bias"] = conv3d_bwd(c18, P["sig/kernel"], dzg)
        d = d1 + d2
        o = cache["_o"]
        bw = lambda n, dd, need_dx=True: b[n].bwd(dd, P, S, cache, g, need_dx=need_dx)
        d = bw("c18", d); d = bw("c17", d)
        dc2_skip, du4 = d[..., :64], d[..., 64:]
        d = upsample_bwd(du4)
        d = bw("c16", d); d = bw("c15", d)
        dc4_skip, du3 = d[..., :128], d[..., 128:]
        d = upsample_bwd(du3)
        d = bw("c14", d); d = bw("c13", d)
        dc6_skip, du1 = d[..., :256], d[..., 256:]
        d = upsample_bwd(du1)
        d = bw("c10", d); d = bw("c9", d)
        d = maxpool_bwd(o["c6"], o["p3"], d, self.pool_ties, cache["c6"].get("impl_o")) + dc6_skip
        d = bw("c6", d); d = bw("c5", d)
        d = maxpool_bwd(o["c4"], o["p2"], d, self.pool_ties, cache["c4"].get("impl_o")) + dc4_skip
        d = bw("c4", d); d = bw("c3", d)
        d = maxpool_bwd(o["c2"], o["p1"], d, self.pool_ties, cache["c2"].get("impl_o")) + dc2_skip
        d = bw("c2", d); bw("c1", d, need_dx=False)
        return g

    def apply_adam(self, grads):
        self.t += 1
        for k in self.P:
            self.P[k], self.m[k], self.v[k] = adam_update(
                self.P[k], grads[k], self.m[k], self.v[k], self.t, self.lr)

    def train_on_batch(self, x, labels, kink=None, kink_tol=1e-4, affine=None, clip_pin=None):
        """One Keras train_on_batch: returns [Loss, lsoft, lsig, f1, wr] (pre-update forward).
        kink: optional {layer: stored activations of the implementation under test} (apply_kink);
        clip_pin: optional K.clip decisions of the losses (backward)."""
        cache = {}
        soft, sig = self.forward(x, training=True, cache=cache)
        metrics = self.loss_and_metrics(soft, sig, labels, cache["_head"]["zg"])
        self.kink_flips = apply_kink(self.blocks.values(), cache, self.P, kink, kink_tol, affine) if kink else {}
        grads = self.backward(labels, cache, clip_pin=clip_pin)
        for blk in self.blocks.values():
            blk.moving_update(self.S, cache, self.bn_unbias)
        self.apply_adam(grads)
        self.last_grads = grads
        return metrics

    def test_on_batch(self, x, labels):
        cache = {}
        soft, sig = self.forward(x, training=False, cache=cache)
        return self.loss_and_metrics(soft, sig, labels, cache["_head"]["zg"])

    # -- perceptual sub-model (vae/lattice_vae.py:257-270): taps = ReLU outputs of c2,c4,c6,c10
    def pm_forward(self, x, training, cache):
        self._trunk(np.asarray(x, self.dtype), training, cache, upto="c10")
        return [cache[n]["s"] for n in PM_TAPS]

    def pm_backward(self, dtaps, cache):
        """Gradient w.r.t. the input given gradients w.r.t. the 4 tap tensors (weights frozen)."""
        P, S, b = self.P, self.S, self.blocks
        o = cache["_o"]

        def bw(n, do, dtap=None):
            # gradient entering via the tap is w.r.t. s (post-ReLU, pre-BN)
            blk, c = b[n], cache[n]
            if c["training"]:
                ds, _, _ = bn_train_bwd(c["s"], P[n + "/gamma"], c["mean"], c["var"], do)
            else:
                ds = bn_eval_bwd(P[n + "/gamma"], S[n + "/moving_var"], do)
            if dtap is not None:
                ds = ds + dtap
            dy = act_bwd(c.get("kink_s", c["s"]), ds, blk.pre_act)
            dx, _, _ = conv3d_bwd(c["x"], P[n + "/kernel"], dy)
            return dx

        z = np.zeros_like
        d = bw("c10", z(cache["c10"]["s"]), dtaps[3]); d = bw("c9", d)
        d = maxpool_bwd(o["c6"], o["p3"], d, self.pool_ties, cache["c6"].get("impl_o"))
        d = bw("c6", d, dtaps[2]); d = bw("c5", d)
        d = maxpool_bwd(o["c4"], o["p2"], d, self.pool_ties, cache["c4"].get("impl_o"))
        d = bw("c4", d, dtaps[1]); d = bw("c3", d)
        d = maxpool_bwd(o["c2"], o["p1"], d, self.pool_ties, cache["c2"].get("impl_o"))
        d = bw("c2", d, dtaps[0]); d = bw("c1", d)
        return d


# --------------------------------------------------------------------------------------
# LatticeDFCVAE graph (vae/lattice_vae.py:160-270)
# --------------------------------------------------------------------------------------
def vae_param_shapes(in_ch=1, cond=10, filters=(16, 32, 64, 128), latent=256, d=32):
    sh = []
    cin = in_ch + in_ch * cond            # K.tile quirk: cond channels = C*cond
    for i, f in enumerate(filters):
        n = "e%d" % i
        sh += [(n + "/kernel", (3, 3, 3, cin, f)), (n + "/bias", (f,)),
               (n + "/gamma", (f,)), (n + "/beta", (f,))]
        cin = f
    sh += [("e4/kernel", (3, 3, 3, cin, 4)), ("e4/bias", (4,))]
    flat = (d // 16) ** 3 * 4
    sh += [("enc_dense/kernel", (flat, latent)), ("enc_dense/bias", (latent,)),
           ("z_mean/kernel", (latent, latent)), ("z_mean/bias", (latent,)),
           ("z_log_var/kernel", (latent, latent)), ("z_log_var/bias", (latent,))]
    seed = (d // 8) ** 3 * 4              # Reshape((4,4,4,4)) at d=32; generalised for d=64 (F12)
    sh += [("dec_dense/kernel", (latent + cond, seed)), ("dec_dense/bias", (seed,))]
    cin = 4
    for i, f in enumerate(filters[::-1]):
        n = "d%d" % i
        sh += [(n + "/kernel", (3, 3, 3, cin, f)), (n + "/bias", (f,)),
               (n + "/gamma", (f,)), (n + "/beta", (f,))]
        cin = f
    sh += [("dout/kernel", (3, 3, 3, cin, in_ch)), ("dout/bias", (in_ch,)),
           ("dout/gamma", (in_ch,)), ("dout/beta", (in_ch,))]
    return sh


def sampling(zm, zlv, eps):
    """lattice_vae.py:53-66 with the K.random_normal draw injected."""
    return zm + np.exp(0.5 * zlv) * eps


def mse_loss(x, recon):
    """lattice_vae.py:232-233: keras `mse` over the FLATTENED tensors -> one scalar (mean over batch too)."""
    return ((np.asarray(x) - recon) ** 2).mean()


def kld_loss(zm, zlv):
    """lattice_vae.py:235-239 -> (B,)."""
    return -0.5 * (1 + zlv - zm ** 2 - np.exp(zlv)).sum(-1)


def perceptual_from_taps(h1, h2, weights):
    """lattice_vae.py:264-270: sum_l w_l * mean over the flattened features of (h1_l - h2_l)^2 -> (B,)."""
    pm = 0.0
    for a, b_, w in zip(h1, h2, weights):
        B = a.shape[0]
        pm = pm + w * ((a - b_).reshape(B, -1) ** 2).mean(-1)
    return pm


def vae_dfc_loss(mse, pm, kld, alpha, beta):
    """lattice_vae.py:247-253: K.mean(rs + alpha * pm + beta * kl) -- scalar + (B,) + (B,) then the batch mean."""
    return (mse + alpha * pm + beta * kld).mean()


class VaeOracle:
    metric_names = ["Loss", "PM", "MSE", "KLD"]    # lattice_vae.py:123

    def __init__(self, unet: UnetOracle, in_ch=1, cond=10, filters=(16, 32, 64, 128), latent=256,
                 d=32, alpha=0.5, beta=3e-4, lr=5e-4, seed=3, dtype=np.float64,
                 pm_layer_weights=(1.0, 1.0, 1.0, 1.0), bn_unbias=True):
        self.unet, self.in_ch, self.cond, self.latent, self.d = unet, in_ch, cond, latent, d
        self.filters = tuple(filters)
        self.alpha, self.beta, self.lr, self.dtype = alpha, beta, lr, dtype
        self.pm_w, self.bn_unbias = pm_layer_weights, bn_unbias
        self.shapes = vae_param_shapes(in_ch, cond, filters, latent, d)
        self.P = init_params(self.shapes, seed, dtype)
        self.S = init_bn_state(self.shapes, dtype)
        nf = len(filters)
        self.enc = [Block("e%d" % i, None, True, "lrelu") for i in range(nf)]
        self.e4 = Block("e4", "lrelu", False, None)
        self.dec = [Block("d%d" % i, None, True, "lrelu") for i in range(nf)]
        self.dout = Block("dout", None, True, "relu")
        self.t = 0
        self.m = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.P.items()}

    def _all_blocks(self):
        return self.enc + [self.e4] + self.dec + [self.dout]

    def tile_cond(self, cond, d):
        """Reshape((1,1,1,cond)) -> K.tile(n=input_shape) -> (B,d,d,d,C*cond) (lattice_vae.py:167-168)."""
        B = cond.shape[0]
        c = np.tile(cond.reshape(B, 1, 1, 1, self.cond), (1, d, d, d, self.in_ch))
        return c

    def encode(self, x, cond, eps, training, cache):
        P, S = self.P, self.S
        x = np.asarray(x, self.dtype); cond = np.asarray(cond, self.dtype)
        h = np.concatenate([x, self.tile_cond(cond, x.shape[1])], -1)
        cache["_pool"] = []
        for blk in self.enc:
            o = blk.fwd(h, P, S, training, cache)
            h = maxpool_fwd(o)
            cache["_pool"].append((o, h))
        h = self.e4.fwd(h, P, S, training, cache)
        flat = h.reshape(h.shape[0], -1)                       # Flatten: row-major (D,H,W,C)
        a = dense_fwd(flat, P["enc_dense/kernel"], P["enc_dense/bias"])
        hd = np.maximum(a, 0)
        zm = dense_fwd(hd, P["z_mean/kernel"], P["z_mean/bias"])
        zlv = dense_fwd(hd, P["z_log_var/kernel"], P["z_log_var/bias"])
        z = sampling(zm, zlv, np.asarray(eps, self.dtype))
        cache["_enc"] = {"flat": flat, "a": a, "hd": hd, "zm": zm, "zlv": zlv, "eps": eps,
                         "e4shape": h.shape}
        return zm, zlv, z

    def decode(self, z, cond, training, cache):
        P, S = self.P, self.S
        z = np.asarray(z, self.dtype); cond = np.asarray(cond, self.dtype)
        zc = np.concatenate([z, cond], -1)
        h = dense_fwd(zc, P["dec_dense/kernel"], P["dec_dense/bias"])
        s = self.d // 8
        h = h.reshape(-1, s, s, s, 4)
        cache["_dec"] = {"zc": zc}
        for i, blk in enumerate(self.dec):
            h = blk.fwd(h, P, S, training, cache)
            if i < len(self.dec) - 1:
                h = upsample_fwd(h)
        return self.dout.fwd(h, P, S, training, cache)

    def losses(self, x, recon, zm, zlv, training, pm_cache=None):
        """Returns ([Loss, PM, MSE, KLD] with PM/KLD batch-averaged, per-sample pm, taps)."""
        x = np.asarray(x, self.dtype)
        mse = mse_loss(x, recon)
        kld = kld_loss(zm, zlv)
        c1, c2 = {}, ({} if pm_cache is None else pm_cache)
        h1 = self.unet.pm_forward(x, training, c1)
        h2 = self.unet.pm_forward(recon, training, c2)
        pm = perceptual_from_taps(h1, h2, self.pm_w)
        loss = vae_dfc_loss(mse, pm, kld, self.alpha, self.beta)
        return np.array([loss, pm.mean(), mse, kld.mean()]), h1, h2

    def forward_losses(self, x, cond, eps, training):
        cache = {}
        zm, zlv, z = self.encode(x, cond, eps, training, cache)
        recon = self.decode(z, cond, training, cache)
        pmc = {}
        metrics, h1, h2 = self.losses(x, recon, zm, zlv, training, pmc)
        return metrics, recon, cache, pmc, (h1, h2)

    def backward(self, x, cache, pmc, taps, recon):
        P, S = self.P, self.S
        g = {}
        x = np.asarray(x, self.dtype)
        B = x.shape[0]
        h1, h2 = taps
        # d Loss / d recon : mse (global mean) + alpha * mean_b(pm_b)
        drec = 2.0 * (recon - x) / x.size
        dt = [self.alpha * w * (-2.0) * (a - b_) / (a[0].size * B)
              for a, b_, w in zip(h1, h2, self.pm_w)]
        drec = drec + self.unet.pm_backward(dt, pmc)
        d = self.dout.bwd(drec, P, S, cache, g)
        for i in reversed(range(len(self.dec))):
            if i < len(self.dec) - 1:
                d = upsample_bwd(d)
            d = self.dec[i].bwd(d, P, S, cache, g)
        dzc, g["dec_dense/kernel"], g["dec_dense/bias"] = dense_bwd(
            cache["_dec"]["zc"], P["dec_dense/kernel"], d.reshape(B, -1))
        dz = dzc[:, :self.latent]
        e = cache["_enc"]
        zm, zlv, eps = e["zm"], e["zlv"], np.asarray(e["eps"], self.dtype)
        # z = zm + exp(0.5 zlv) eps ; KLD term beta * mean_b(kld_b)
        dzm = dz + self.beta * zm / B
        dzlv = dz * eps * 0.5 * np.exp(0.5 * zlv) + self.beta * (-0.5) * (1 - np.exp(zlv)) / B
        dh1, g["z_mean/kernel"], g["z_mean/bias"] = dense_bwd(e["hd"], P["z_mean/kernel"], dzm)
        dh2, g["z_log_var/kernel"], g["z_log_var/bias"] = dense_bwd(e["hd"], P["z_log_var/kernel"], dzlv)
        da = (dh1 + dh2) * (e.get("kink_hd", e["a"]) > 0)
        dflat, g["enc_dense/kernel"], g["enc_dense/bias"] = dense_bwd(e["flat"], P["enc_dense/kernel"], da)
        d = self.e4.bwd(dflat.reshape(e["e4shape"]), P, S, cache, g)
        for i in reversed(range(len(self.enc))):
            o, p = cache["_pool"][i]
            d = maxpool_bwd(o, p, d, self.unet.pool_ties, cache["e%d" % i].get("impl_o"))
            d = self.enc[i].bwd(d, P, S, cache, g, need_dx=(i > 0))
        return g

    def apply_adam(self, grads):
        self.t += 1
        for k in self.P:
            self.P[k], self.m[k], self.v[k] = adam_update(
                self.P[k], grads[k], self.m[k], self.v[k], self.t, self.lr)

    def train_on_batch(self, x, cond, eps, kink=None, kink_pm=None, kink_tol=1e-4, affine=None, affine_pm=None):
        """model.train_on_batch([M,cond], M) (lattice_vae.py:296) with eps injected.
        The perceptual U-Net runs BN in batch-stat mode, weights and moving stats frozen (F9).
        kink / kink_pm: stored activations of the implementation under test for the VAE layers and
        for the perceptual U-Net's pass over the reconstruction (see apply_kink)."""
        metrics, recon, cache, pmc, taps = self.forward_losses(x, cond, eps, True)
        self.kink_flips = {}
        if kink:
            self.kink_flips.update(apply_kink(self._all_blocks(), cache, self.P, kink, kink_tol, affine))
            if "enc_dense" in kink:
                a = cache["_enc"]["a"]
                hd_impl = np.asarray(kink["enc_dense"], a.dtype).reshape(a.shape)
                diff = (hd_impl > 0) != (a > 0)
                if (diff & (np.minimum(np.abs(a), np.abs(hd_impl)) > kink_tol)).any():
                    raise AssertionError("enc_dense relu masks differ away from the kink")
                cache["_enc"]["kink_hd"] = hd_impl
                self.kink_flips["enc_dense"] = int(diff.sum())
        if kink_pm:
            fl = apply_kink(self.unet.blocks.values(), pmc, self.unet.P, kink_pm, kink_tol, affine_pm)
            self.kink_flips.update({"pm/" + k: v for k, v in fl.items()})
        grads = self.backward(x, cache, pmc, taps, recon)
        for blk in self._all_blocks():
            blk.moving_update(self.S, cache, self.bn_unbias)
        self.apply_adam(grads)
        self.last_grads = grads
        return metrics

    def test_on_batch(self, x, cond, eps):
        return self.forward_losses(x, cond, eps, False)[0]

    def predict_encoder(self, x, cond, eps):
        return self.encode(x, cond, eps, False, {})

    def predict_decoder(self, z, cond):
        return self.decode(z, cond, False, {})


# --------------------------------------------------------------------------------------
# Synthetic workload; shared definition for tests and bench (data only).
# --------------------------------------------------------------------------------------
def synthetic_batch(B, d=32, C=1, seed=0, dtype=np.float32):
    rng = np.random.Generator(np.random.PCG64(seed))
    zz, yy, xx = np.meshgrid(np.arange(d), np.arange(d), np.arange(d), indexing="ij")
    X = np.zeros((B, d, d, d, C), dtype)
    labels = np.zeros((B, d, d, d), np.uint8)
    for b in range(B):
        n = int(rng.integers(2, 9))
        dens = np.zeros((d, d, d))
        for k in range(n):
            c = rng.uniform(0, d, 3)
            sig = rng.uniform(1.5, 4.0) * d / 32.0
            amp = rng.uniform(0.5, 3.0)
            r2 = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2
            dens += amp * np.exp(-r2 / (2 * sig * sig))
            labels[b][r2 <= sig * sig] = 1 + (k * 13) % 94
        X[b, ..., 0] = np.maximum(dens, 0)
        if C > 1:
            g = np.stack([zz, yy, xx], -1) / float(d)
            X[b, ..., 1:4] = g[..., :C - 1]
    cond = np.eye(10, dtype=dtype)[np.arange(B) % 10]
    return X, labels, cond
