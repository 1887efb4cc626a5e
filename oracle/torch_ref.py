"""
ORACLE CROSS-CHECK + CPU BASELINE (test infrastructure, NOT product code).

An INDEPENDENT torch-CPU implementation (F.conv3d / F.batch_norm / F.max_pool3d / autograd) of the
same two graphs that oracle/numpy_ref.py restates from /root/reference/unet/unet.py:272-355 and
/root/reference/vae/lattice_vae.py:160-270.  Purposes:
  (1) pin the numpy restatement (layout transposes, padding, BN variance convention, gradient
      formulas) against a second implementation that shares no code with it -- the reference's own
      Keras/TF path cannot be imported here, so parity stays "unpinned" w.r.t. Keras;
  (2) generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py);
  (3) bench.py's cpu_baseline leg ("port"): fp32, all host cores, timed in a subprocess.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
K_EPS = 1e-7
LEAKY = 0.3


def to_t(a, dtype):          # NDHWC numpy -> NCDHW torch
    t = torch.as_tensor(np.ascontiguousarray(a), dtype=dtype)
    if t.ndim == 5:
        t = t.permute(0, 4, 1, 2, 3).contiguous()
    return t


def to_n(t):                 # NCDHW torch -> NDHWC numpy
    if t.ndim == 5:
        t = t.permute(0, 2, 3, 4, 1)
    return t.detach().contiguous().numpy()


def kernel_t(w, dtype):      # (kd,kh,kw,Cin,Cout) -> (Cout,Cin,kd,kh,kw)
    return torch.as_tensor(np.ascontiguousarray(np.transpose(w, (4, 3, 0, 1, 2))), dtype=dtype)


def kernel_grad_n(g):        # (Cout,Cin,kd,kh,kw) -> (kd,kh,kw,Cin,Cout)
    return np.ascontiguousarray(np.transpose(g.detach().numpy(), (2, 3, 4, 1, 0)))


# ------------------------------------------------------------------------------ fp64 convolution as GEMMs
# torch's fp64 conv3d on CPU parallelises over the batch only and took 168 s for one U-Net step at B = 32 on the GPU box's
# 128 cores (round 6, first run of tests/test_gpu_fullsize_oracle.py).  The same sums as 9 GEMMs per layer -- one per (dz, dy)
# tap pair, the three dx taps side by side in K -- run on MKL's threaded dgemm.  Same definition as F.conv3d (cross-
# correlation, zero "same" padding, unet/unet.py:276-336); tests/test_oracle_pinned_torch.py holds the two to 1e-12.
GEMM_CONV_MIN_WORK = 2e9      # multiply-adds from which conv3d() takes the GEMM form (fp64 only); 0 forces it, inf disables


def _tap_cols(xp, dz, dy, B, D, H, W, C, buf):
    """(B*D*H*W, 3C): for every output voxel the three x-neighbours of tap row (dz, dy) of the padded NDHWC tensor, gathered
    into `buf` (one buffer per convolution call: nine fresh 100 MB - 5 GB temporaries per layer were a quarter of the oracle's
    time on the GPU box -- page faults and unmaps, not arithmetic)."""
    v = xp[:, dz:dz + D, dy:dy + H]                       # (B, D, H, W + 2, C), C contiguous
    st = v.stride()
    buf.view(B, D, H, W, 3 * C).copy_(v.as_strided((B, D, H, W, 3 * C), (st[0], st[1], st[2], st[3], 1)))
    return buf


def _conv_gemm_fwd(x, w):
    """x (B, Cin, D, H, W), w (Cout, Cin, 3, 3, 3) -> (B, Cout, D, H, W) (a permuted view of an NDHWC result)."""
    B, C, D, H, W = x.shape
    Co = w.shape[0]
    xp = F.pad(x.permute(0, 2, 3, 4, 1), (0, 0, 1, 1, 1, 1, 1, 1)).contiguous()
    wk = w.permute(2, 3, 4, 1, 0).contiguous()            # (3, 3, 3, Cin, Cout)
    y = x.new_zeros(B * D * H * W, Co)
    buf = x.new_empty(B * D * H * W, 3 * C)
    for dz in range(3):
        for dy in range(3):
            y.addmm_(_tap_cols(xp, dz, dy, B, D, H, W, C, buf), wk[dz, dy].reshape(3 * C, Co))
    return y.view(B, D, H, W, Co).permute(0, 4, 1, 2, 3)


def _conv_gemm_wgrad(x, dy):
    B, C, D, H, W = x.shape
    Co = dy.shape[1]
    xp = F.pad(x.permute(0, 2, 3, 4, 1), (0, 0, 1, 1, 1, 1, 1, 1)).contiguous()
    dyf = dy.permute(0, 2, 3, 4, 1).reshape(B * D * H * W, Co)
    dw = x.new_empty(3, 3, 3 * C, Co)
    buf = x.new_empty(B * D * H * W, 3 * C)
    for dz in range(3):
        for dy_ in range(3):
            torch.mm(_tap_cols(xp, dz, dy_, B, D, H, W, C, buf).t(), dyf, out=dw[dz, dy_])
    return dw.view(3, 3, 3, C, Co).permute(4, 3, 0, 1, 2).contiguous()


class _Conv3dGemm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return _conv_gemm_fwd(x, w) + b.view(1, -1, 1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = _conv_gemm_fwd(dy, w.flip(2, 3, 4).transpose(0, 1)) if ctx.needs_input_grad[0] else None
        dw = _conv_gemm_wgrad(x, dy) if ctx.needs_input_grad[1] else None
        db = dy.sum((0, 2, 3, 4)) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def conv3d(x, w, b):
    """3x3x3 "same" convolution + bias: F.conv3d, or the GEMM form above for large fp64 problems."""
    work = float(x.shape[0]) * x.shape[2] * x.shape[3] * x.shape[4] * 27 * w.shape[0] * w.shape[1]
    if x.dtype == torch.float64 and work >= GEMM_CONV_MIN_WORK:
        return _Conv3dGemm.apply(x, w, b)
    return F.conv3d(x, w, b, padding=1)


def conv1(x, w, b):
    """1x1x1 convolution (the two heads, unet/unet.py:339-352): one matmul over the channel axis for large fp64 inputs."""
    if x.dtype == torch.float64 and x.numel() * w.shape[0] >= GEMM_CONV_MIN_WORK:
        B, C, D, H, W = x.shape
        y = x.permute(0, 2, 3, 4, 1).reshape(-1, C) @ w.view(w.shape[0], C).t() + b
        return y.view(B, D, H, W, -1).permute(0, 4, 1, 2, 3)
    return F.conv3d(x, w, b)


class Params:
    """Holds torch leaf tensors converted from the oracle's name->numpy dict."""

    def __init__(self, P, S, dtype=torch.float64, requires_grad=True):
        self.t, self.s, self.dtype = {}, {}, dtype
        for k, v in P.items():
            if k.endswith("/kernel") and v.ndim == 5:
                t = kernel_t(v, dtype)
            else:
                t = torch.as_tensor(np.ascontiguousarray(v), dtype=dtype)
            self.t[k] = t.requires_grad_(requires_grad)
        for k, v in S.items():
            self.s[k] = torch.as_tensor(np.ascontiguousarray(v), dtype=dtype)

    def grads_numpy(self):
        out = {}
        for k, t in self.t.items():
            if t.grad is None:
                continue
            out[k] = kernel_grad_n(t.grad) if (k.endswith("/kernel") and t.ndim == 5) else t.grad.numpy().copy()
        return out


def maxpool(x, ties, route=None):
    """route: optional tensor to take the routing decision from (the tested implementation's own pool input, Pins.route)."""
    y = F.max_pool3d(x, 2)
    if ties == "first":
        if route is not None:
            raise ValueError("pinned routing is implemented for the tf_cpu tie rule")
        return y
    # TF-CPU MaxPool3DGrad semantics: gradient to every element within 1e-5 of the window max
    r = x.detach() if route is None else route
    up = F.interpolate(F.max_pool3d(r, 2), scale_factor=2, mode="nearest")
    mask = ((r - up).abs() < 1e-5).to(x.dtype)
    routed = F.avg_pool3d(x * mask, 2) * 8.0
    return y.detach() + routed - routed.detach()


def act(x, kind):
    if kind is None:
        return x
    if kind == "relu":
        return F.relu(x)
    return F.leaky_relu(x, LEAKY)


# ------------------------------------------------------------------------------ pinned decisions (fp32 implementation under test)
class _PinnedAct(torch.autograd.Function):
    """ReLU / LeakyReLU whose DERIVATIVE takes the side of the kink from `impl` (the tested implementation's value at the same
    place) instead of from x: act'(0) is a discontinuity, and an fp32 implementation legitimately lands a handful of
    pre-activations on the other side of it than fp64 does (numpy_ref.apply_kink is the same device for the numpy oracle)."""

    @staticmethod
    def forward(ctx, x, impl, slope):
        ctx.save_for_backward(impl > 0)
        ctx.slope = slope
        return torch.where(x > 0, x, slope * x)

    @staticmethod
    def backward(ctx, dy):
        (pos,) = ctx.saved_tensors
        return torch.where(pos, dy, ctx.slope * dy), None, None


class _PinnedClamp(torch.autograd.Function):
    """K.clip(x, lo, hi) whose zero-gradient region is given (`inside`, 0/1) instead of derived from x."""

    @staticmethod
    def forward(ctx, x, inside, lo, hi):
        ctx.save_for_backward(inside)
        return torch.clamp(x, lo, hi)

    @staticmethod
    def backward(ctx, dy):
        (inside,) = ctx.saved_tensors
        return dy * inside, None, None, None


class Pins:
    """Decisions of the implementation under test, for a training step of one network.
    kink   {layer: its stored activation s = pre_act(conv + b), NDHWC float32}
    affine {layer: (scale32, shift32)} of the layers in front of a max-pool: the implementation's pool input
           o = act(fma(s, scale, shift)) is rebuilt in fp32 and decides the (equally discontinuous) routing.
    A decision may differ from the fp64 one only where BOTH values -- the oracle's PRE-activation and the implementation's
    stored value -- are within `tol` x max(1, max|pre-activation| of the layer) of the kink (forward parity is 1e-5 of the
    tensor's largest entry, so that is the scale on which an fp32 value may land on the other side); anything else raises.
    (Stricter than numpy_ref.apply_kink, which sees only post-activation values: for a ReLU layer one of the two is then
    always 0 and its test `min(|ref|, |impl|) > tol` cannot fire.)  worst[layer] = the largest such relative distance from
    the kink among the pinned decisions."""

    def __init__(self, kink=None, affine=None, tol=1e-4, dtype=torch.float64):
        self.kink, self.affine, self.tol, self.dtype = kink or {}, affine or {}, tol, dtype
        self.flips, self.route, self.worst = {}, {}, {}

    def impl(self, name, like):
        if name not in self.kink:
            return None
        t = to_t(np.asarray(self.kink[name]).reshape(tuple(like.permute(0, 2, 3, 4, 1).shape)), self.dtype)
        return t

    def count(self, name, ref, impl):
        """ref: the oracle's value in front of the activation; impl: the implementation's (pre- or post-activation: same sign)."""
        diff = (impl > 0) != (ref > 0)
        n = int(diff.sum())
        if n:
            dist = torch.maximum(ref.abs(), impl.abs())[diff] / max(1.0, float(ref.abs().max()))
            self.worst[name] = max(self.worst.get(name, 0.0), float(dist.max()))
            if self.worst[name] > self.tol:
                raise AssertionError("activation signs of %s differ away from the kink (%d elements, up to %.3g from it)"
                                     % (name, int((dist > self.tol).sum()), self.worst[name]))
        self.flips[name] = self.flips.get(name, 0) + n


def bn(x, p, name, training, stats_out=None):
    g, b = p.t[name + "/gamma"], p.t[name + "/beta"]
    if training:
        if x.dtype == torch.float64:
            # tf.nn.moments + the affine, spelled out: torch's fp64 native_batch_norm_backward on a channels-last-strided
            # tensor was a quarter of the oracle's time on the GPU box (316 ms per layer at B = 8); the same arithmetic as
            # elementwise passes differentiates through parallel kernels
            mean = x.mean(dim=(0, 2, 3, 4), keepdim=True)
            var = ((x - mean) ** 2).mean(dim=(0, 2, 3, 4), keepdim=True)
            y = (x - mean) * (g.view(1, -1, 1, 1, 1) * torch.rsqrt(var + BN_EPS)) + b.view(1, -1, 1, 1, 1)
            if stats_out is not None:
                n = x.numel() // x.shape[1]
                stats_out[name] = (mean.detach().reshape(-1).numpy().copy(), var.detach().reshape(-1).numpy().copy(), n)
            return y
        y = F.batch_norm(x, None, None, g, b, True, 0.0, BN_EPS)
        if stats_out is not None:
            xf = x.detach().transpose(0, 1).reshape(x.shape[1], -1)
            stats_out[name] = (xf.mean(1).numpy(), xf.var(1, unbiased=False).numpy(), xf.shape[1])
        return y
    return F.batch_norm(x, p.s[name + "/moving_mean"], p.s[name + "/moving_var"], g, b, False, 0.0, BN_EPS)


_SLOPE = {"relu": 0.0, "lrelu": LEAKY}


GRAD_CAPTURE = None      # diagnosis aid: a dict here receives {layer: dLoss/ds (NCDHW tensor)} for every block whose stored
                         # activation s carries a gradient (tests/tools/fuzz_steps.py FUZZ_DY=1 compares them with the engine's "layer:dy")


def block(x, p, name, pre, has_bn, post, training, taps=None, stats=None, pins=None):
    y = conv3d(x, p.t[name + "/kernel"], p.t[name + "/bias"])
    si = pins.impl(name, y) if pins is not None else None
    if si is None:
        s = act(y, pre)
    else:
        s = y
        if pre is not None:
            s = _PinnedAct.apply(y, si, _SLOPE[pre])
            pins.count(name, y.detach(), si)
    if taps is not None:
        taps[name] = s
    if GRAD_CAPTURE is not None and s.requires_grad:
        s.register_hook(lambda g, n=name: GRAD_CAPTURE.__setitem__(n, g.detach()))
    if not has_bn:
        return s
    o = bn(s, p, name, training, stats)
    o32 = None
    if si is not None and name in pins.affine:
        sc, sh = pins.affine[name]
        view = (1, -1, 1, 1, 1)
        o32 = (si * torch.as_tensor(np.asarray(sc, np.float64)).view(view)
               + torch.as_tensor(np.asarray(sh, np.float64)).view(view)).to(torch.float32)     # = fmaf(s, scale, shift) in fp32
        pins.route[name] = act(o32, post).to(s.dtype)
    if si is None or post is None:
        return act(o, post)
    if not training:
        raise ValueError("kink pinning is for training-mode steps")
    if o32 is not None:
        # the implementation's OWN BatchNorm output, bit for bit: the side of the kink it took.  (Without its fp32 affine
        # the output is rebuilt from its stored activation and the oracle's statistics -- the sign of a value within ~1e-7
        # of zero can then differ from the implementation's; at 2e7 activations per layer that happens: round 6, e1 at B = 32.)
        bn_impl = o32.to(s.dtype)
    else:
        sd = s.detach()
        mean = sd.mean(dim=(0, 2, 3, 4), keepdim=True)
        var = ((sd - mean) ** 2).mean(dim=(0, 2, 3, 4), keepdim=True)
        inv = p.t[name + "/gamma"].detach().view(1, -1, 1, 1, 1) / torch.sqrt(var + BN_EPS)
        bn_impl = si * inv + (p.t[name + "/beta"].detach().view(1, -1, 1, 1, 1) - mean * inv)
    pins.count(name, o.detach(), bn_impl)
    return _PinnedAct.apply(o, bn_impl, _SLOPE[post])


def unet_trunk(x, p, training, ties, upto=None, taps=None, stats=None, pins=None):
    f = lambda n, t: block(t, p, n, "relu", True, None, training, taps, stats, pins)
    rt = (lambda n: pins.route.get(n)) if pins is not None else (lambda n: None)
    c2 = f("c2", f("c1", x)); p1 = maxpool(c2, ties, rt("c2"))
    c4 = f("c4", f("c3", p1)); p2 = maxpool(c4, ties, rt("c4"))
    c6 = f("c6", f("c5", p2)); p3 = maxpool(c6, ties, rt("c6"))
    c10 = f("c10", f("c9", p3))
    if upto == "c10":
        return c10
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    c14 = f("c14", f("c13", torch.cat([c6, up(c10)], 1)))
    c16 = f("c16", f("c15", torch.cat([c4, up(c14)], 1)))
    return f("c18", f("c17", torch.cat([c2, up(c16)], 1)))


def unet_forward(x, p, training, ties="tf_cpu", stats=None, pins=None):
    c18 = unet_trunk(x, p, training, ties, stats=stats, pins=pins)
    zs = conv1(c18, p.t["soft/kernel"], p.t["soft/bias"])
    zg = conv1(c18, p.t["sig/kernel"], p.t["sig/bias"])
    return torch.softmax(zs, 1), torch.sigmoid(zg)


def _pin_inside(inside, v, pin, tol, what, flips, key):
    """numpy_ref._pinned_inside: `pin` replaces the clip's inside / outside decision, but only where the clipped quantity is
    within tol (relative to the bound's distance from the nearer end of [0, 1]) of a bound."""
    pin = torch.as_tensor(np.asarray(pin, bool)).reshape(inside.shape)
    diff = pin != inside
    near = ((v - K_EPS).abs() <= tol * K_EPS) | (((1 - v) - K_EPS).abs() <= tol * K_EPS)
    if bool((diff & ~near).any()):
        raise AssertionError("%s: %d clip decisions differ away from the bounds" % (what, int((diff & ~near).sum())))
    flips[key] = int(diff.sum())
    return pin


def unet_loss(soft, sig, labels, num_classes=95, weight=None, bce_from_logits=False, clip_pin=None, clip_flips=None):
    """[Loss, lsoft, lsig] as torch scalars (unet/unet.py:211-219,252-256).  bce_from_logits: the sigmoid head's loss in
    TF 2.1's short-circuit form, sigmoid_cross_entropy_with_logits on the head's logits (confidence M).
    clip_pin: optional {"soft": bool (B,D,H,W), "sig": bool (B,D,H,W)} -- the tested implementation's K.clip decisions
    ("the true class's probability / the sigmoid output is inside [1e-7, 1 - 1e-7]"), pinned like the ReLU masks."""
    weight = float(num_classes) if weight is None else weight
    clip_pin = clip_pin or {}
    clip_flips = {} if clip_flips is None else clip_flips
    lab = torch.as_tensor(labels.astype(np.int64))
    y = F.one_hot(lab, num_classes).permute(0, 4, 1, 2, 3).to(soft.dtype)
    t = (lab != 0).to(soft.dtype).unsqueeze(1)
    q = soft / soft.sum(1, keepdim=True)
    if clip_pin.get("soft") is not None:
        qd = q.detach()
        inside = (qd >= K_EPS) & (qd <= 1 - K_EPS)
        qt = (qd * y).sum(1, keepdim=True)                  # only the true class's entry carries a gradient: pin that one
        it = _pin_inside((qt >= K_EPS) & (qt <= 1 - K_EPS), qt, np.asarray(clip_pin["soft"])[:, None], 0.5,
                         "softmax clip", clip_flips, "soft_clip")
        inside = torch.where(y > 0, it, inside)
        qc = _PinnedClamp.apply(q, inside.to(q.dtype), K_EPS, 1 - K_EPS)
    else:
        qc = torch.clamp(q, K_EPS, 1 - K_EPS)
    lsoft = (-(y * torch.log(qc) * weight).sum(1)).mean(dim=(1, 2, 3)).mean()
    if bce_from_logits:
        lsig = F.binary_cross_entropy_with_logits(torch.logit(sig), t)       # logit(sigmoid(z)) = z
    else:
        if clip_pin.get("sig") is not None:
            sd = sig.detach()
            inside = _pin_inside((sd >= K_EPS) & (sd <= 1 - K_EPS), sd, np.asarray(clip_pin["sig"])[:, None], 0.5,
                                 "sigmoid clip", clip_flips, "sig_clip")
            pc = _PinnedClamp.apply(sig, inside.to(sig.dtype), K_EPS, 1 - K_EPS)
        else:
            pc = torch.clamp(sig, K_EPS, 1 - K_EPS)
        lsig = (-(t * torch.log(pc) + (1 - t) * torch.log(1 - pc))).mean()
    return lsoft + lsig, lsoft, lsig


def unet_metrics(soft, labels, num_classes=95):
    """f1_m / wr_m (unet/unet.py:159-193) on torch tensors: K.round is half-to-even (torch.round likewise)."""
    lab = torch.as_tensor(labels.astype(np.int64))
    p = soft.detach()
    y = F.one_hot(lab, num_classes).permute(0, 4, 1, 2, 3).to(p.dtype)
    yp = torch.round(torch.clamp(y * p, 0, 1))
    tp, possible, predicted = yp.sum(), y.sum(), torch.round(torch.clamp(p, 0, 1)).sum()
    precision, recall = tp / (predicted + K_EPS), tp / (possible + K_EPS)
    f1 = 2 * ((precision * recall) / (precision + recall + K_EPS))
    wr = yp[:, 1:].sum() / (y[:, 1:].sum() + K_EPS)
    unet_metrics.counts = {"tp": float(tp), "predicted": float(predicted), "wr_tp": float(yp[:, 1:].sum()),
                           "wr_possible": float(y[:, 1:].sum()), "voxels": float(possible)}
    # K.round at one half is a discontinuity: probabilities within 2e-5 of 0.5 (the fp32 forward tolerance on a probability)
    # may round either way in an fp32 implementation -- how many such entries each count holds
    near = (p - 0.5).abs() <= 2e-5
    unet_metrics.borderline = {"predicted": float(near.sum()), "tp": float((near & (y > 0)).sum()),
                               "wr_tp": float((near[:, 1:] & (y[:, 1:] > 0)).sum()), "wr_possible": 0.0}
    return float(f1), float(wr)


def unet_step_grads(P, S, x, labels, dtype=torch.float64, ties="tf_cpu", num_classes=95, bce_from_logits=False,
                    kink=None, affine=None, clip_pin=None, kink_tol=1e-4, want_outputs=True):
    """Returns (metrics[3], grads dict (numpy, Keras layouts), bn batch stats, soft, sig).
    kink / affine / clip_pin: decisions of the implementation under test (Pins, unet_loss); the flips are left in
    unet_step_grads.flips = {"kink": {layer: n}, "clip": {...}}, f1 / wr in unet_step_grads.f1_wr."""
    p = Params(P, S, dtype)
    stats = {}
    pins = Pins(kink, affine, kink_tol, dtype) if kink else None
    soft, sig = unet_forward(to_t(x, dtype), p, True, ties, stats, pins)
    cf = {}
    loss, lsoft, lsig = unet_loss(soft, sig, labels, num_classes, bce_from_logits=bce_from_logits, clip_pin=clip_pin,
                                  clip_flips=cf)
    loss.backward()
    unet_step_grads.flips = {"kink": pins.flips if pins else {}, "clip": cf}
    unet_step_grads.pin_worst = dict(pins.worst) if pins else {}
    unet_step_grads.f1_wr = unet_metrics(soft, labels, num_classes)
    return (np.array([loss.item(), lsoft.item(), lsig.item()]), p.grads_numpy(), stats,
            to_n(soft) if want_outputs else None, to_n(sig) if want_outputs else None)


# ------------------------------------------------------------------------------ VAE
def vae_forward(x, cond, eps, pv, training, in_ch, ncond, d, nf=4, stats=None, pins=None):
    B = x.shape[0]
    ct = cond.reshape(B, ncond, 1, 1, 1).repeat(1, in_ch, d, d, d)     # K.tile quirk: C*cond channels
    h = torch.cat([x, ct], 1)
    for i in range(nf):
        o = block(h, pv, "e%d" % i, None, True, "lrelu", training, None, stats, pins)
        h = maxpool(o, _POOL_TIES["mode"], pins.route.get("e%d" % i) if pins is not None else None)
    h = block(h, pv, "e4", "lrelu", False, None, training, pins=pins)
    flat = h.permute(0, 2, 3, 4, 1).reshape(B, -1)                      # Keras Flatten is (D,H,W,C)
    a = flat @ pv.t["enc_dense/kernel"] + pv.t["enc_dense/bias"]
    if pins is not None and "enc_dense" in pins.kink:
        hi = torch.as_tensor(np.asarray(pins.kink["enc_dense"]).reshape(tuple(a.shape)), dtype=a.dtype)
        hd = _PinnedAct.apply(a, hi, 0.0)
        pins.count("enc_dense", a.detach(), hi)
    else:
        hd = F.relu(a)
    zm = hd @ pv.t["z_mean/kernel"] + pv.t["z_mean/bias"]
    zlv = hd @ pv.t["z_log_var/kernel"] + pv.t["z_log_var/bias"]
    z = zm + torch.exp(0.5 * zlv) * eps
    recon = vae_decode(z, cond, pv, training, d, nf, stats, pins)
    return zm, zlv, z, recon


_POOL_TIES = {"mode": "tf_cpu"}


def vae_decode(z, cond, pv, training, d, nf=4, stats=None, pins=None):
    B = z.shape[0]
    h = torch.cat([z, cond], 1) @ pv.t["dec_dense/kernel"] + pv.t["dec_dense/bias"]
    s = d // 8
    h = h.reshape(B, s, s, s, 4).permute(0, 4, 1, 2, 3)                 # Keras Reshape is (D,H,W,C)
    for i in range(nf):
        h = block(h, pv, "d%d" % i, None, True, "lrelu", training, None, stats, pins)
        if i < nf - 1:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
    return block(h, pv, "dout", None, True, "relu", training, None, stats, pins)


def vae_losses(x, recon, zm, zlv, pu, training, alpha, beta, ties, pm_w=(1, 1, 1, 1), pins_pm=None):
    """pins_pm: decisions of the tested implementation's perceptual pass over the RECONSTRUCTION (the pass over x carries
    no gradient)."""
    mse = ((x - recon) ** 2).mean()
    kld = -0.5 * (1 + zlv - zm ** 2 - torch.exp(zlv)).sum(-1)
    t1, t2 = {}, {}
    with torch.no_grad():
        unet_trunk(x, pu, training, ties, upto="c10", taps=t1)
    unet_trunk(recon, pu, training, ties, upto="c10", taps=t2, pins=pins_pm)
    pm = 0.0
    B = x.shape[0]
    for n, w in zip(("c2", "c4", "c6", "c10"), pm_w):
        pm = pm + w * ((t1[n] - t2[n]).reshape(B, -1) ** 2).mean(-1)
    loss = (mse + alpha * pm + beta * kld).mean()
    return loss, pm.mean(), mse, kld.mean()


def vae_step_grads(Pv, Sv, Pu, Su, x, cond, eps, in_ch=1, ncond=10, d=32, alpha=0.5, beta=3e-4,
                   dtype=torch.float64, ties="tf_cpu", training=True, nf=4,
                   kink=None, kink_pm=None, affine=None, affine_pm=None, kink_tol=1e-4):
    """kink / affine: decisions of the tested implementation's VAE layers, kink_pm / affine_pm: of its perceptual U-Net pass
    over the reconstruction (Pins); flips land in vae_step_grads.flips."""
    _POOL_TIES["mode"] = ties
    pv = Params(Pv, Sv, dtype)
    pu = Params(Pu, Su, dtype, requires_grad=False)
    stats = {}
    xt = to_t(x, dtype)
    ct = torch.as_tensor(cond, dtype=dtype)
    et = torch.as_tensor(eps, dtype=dtype)
    pins = Pins(kink, affine, kink_tol, dtype) if kink else None
    pins_pm = Pins(kink_pm, affine_pm, kink_tol, dtype) if kink_pm else None
    zm, zlv, z, recon = vae_forward(xt, ct, et, pv, training, in_ch, ncond, d, nf, stats, pins)
    loss, pm, mse, kld = vae_losses(xt, recon, zm, zlv, pu, training, alpha, beta, ties, pins_pm=pins_pm)
    fl = dict(pins.flips) if pins else {}
    fl.update({"pm/" + k: v for k, v in (pins_pm.flips if pins_pm else {}).items()})
    vae_step_grads.flips = fl
    vae_step_grads.pin_worst = {**(pins.worst if pins else {}), **{"pm/" + k: v for k, v in (pins_pm.worst if pins_pm else {}).items()}}
    if training:
        loss.backward()
    return (np.array([loss.item(), pm.item(), mse.item(), kld.item()]),
            pv.grads_numpy() if training else {}, stats, to_n(recon), zm.detach().numpy(), zlv.detach().numpy())


# ------------------------------------------------------------------------------ CPU baseline
def time_unet_train_step(B=4, d=32, in_ch=1, steps=2, warmup=1, threads=None):
    """fp32 torch-CPU U-Net fwd+bwd+Adam on synthetic data; returns (grids_per_s, threads, secs/step)."""
    import time
    from . import numpy_ref as R
    if threads:
        torch.set_num_threads(threads)
    threads = torch.get_num_threads()
    shapes = R.unet_param_shapes(in_ch, 95)
    P = R.init_params(shapes, 1, np.float32)
    S = R.init_bn_state(shapes, np.float32)
    p = Params(P, S, torch.float32)
    for k, t in list(p.t.items()):           # channels_last_3d conv weights for the MKL-DNN path
        if t.ndim == 5:
            p.t[k] = t.detach().contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    opt = torch.optim.Adam(list(p.t.values()), lr=3e-6, eps=1e-7)
    X, labels, _ = R.synthetic_batch(B, d, in_ch, seed=0)
    xt = to_t(X, torch.float32).contiguous(memory_format=torch.channels_last_3d)
    times = []
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        soft, sig = unet_forward(xt, p, True, "first")
        loss, _, _ = unet_loss(soft, sig, labels)
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    time_unet_train_step.samples = list(times)
    sec = float(np.median(times))
    return B / sec, threads, sec


def time_vae_train_step(B=4, d=32, in_ch=1, steps=1, warmup=0, threads=None):
    """fp32 torch-CPU DFC-VAE train step (encoder + decoder + frozen perceptual U-Net c1..c10 on x and on the
    reconstruction, batch-statistics BN, backward, Adam) on synthetic data; returns (grids_per_s, threads, secs/step)."""
    import time
    from . import numpy_ref as R
    if threads:
        torch.set_num_threads(threads)
    threads = torch.get_num_threads()
    ush, vsh = R.unet_param_shapes(in_ch, 95), R.vae_param_shapes(in_ch, d=d)
    pu = Params(R.init_params(ush, 1, np.float32), R.init_bn_state(ush, np.float32), torch.float32, requires_grad=False)
    pv = Params(R.init_params(vsh, 3, np.float32), R.init_bn_state(vsh, np.float32), torch.float32)
    for p in (pu, pv):
        for k, t in list(p.t.items()):
            if t.ndim == 5:
                p.t[k] = t.detach().contiguous(memory_format=torch.channels_last_3d).requires_grad_(p is pv)
    opt = torch.optim.Adam(list(pv.t.values()), lr=5e-4, eps=1e-7)
    X, _, cond = R.synthetic_batch(B, d, in_ch, seed=0)
    xt = to_t(X, torch.float32).contiguous(memory_format=torch.channels_last_3d)
    ct = torch.as_tensor(cond, dtype=torch.float32)
    et = torch.as_tensor(np.random.default_rng(2).standard_normal((B, 256)), dtype=torch.float32)
    _POOL_TIES["mode"] = "first"
    times = []
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        zm, zlv, z, recon = vae_forward(xt, ct, et, pv, True, in_ch, 10, d)
        loss, _, _, _ = vae_losses(xt, recon, zm, zlv, pu, True, 0.5, 3e-4, "first")
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    _POOL_TIES["mode"] = "tf_cpu"
    time_vae_train_step.samples = list(times)
    sec = float(np.median(times))
    return B / sec, threads, sec


def time_unet_predict(B=16, d=32, in_ch=1, steps=1, warmup=0, threads=None):
    """fp32 torch-CPU U-Net forward (eval-mode BN, both heads) on B synthetic grids -- BASELINE configs[0]'s CPU-runnable
    case; returns (grids_per_s, threads, secs/call)."""
    import time
    from . import numpy_ref as R
    if threads:
        torch.set_num_threads(threads)
    threads = torch.get_num_threads()
    shapes = R.unet_param_shapes(in_ch, 95)
    p = Params(R.init_params(shapes, 1, np.float32), R.init_bn_state(shapes, np.float32), torch.float32, requires_grad=False)
    for k, t in list(p.t.items()):
        if t.ndim == 5:
            p.t[k] = t.detach().contiguous(memory_format=torch.channels_last_3d)
    X, _, _ = R.synthetic_batch(B, d, in_ch, seed=0)
    xt = to_t(X, torch.float32).contiguous(memory_format=torch.channels_last_3d)
    times = []
    with torch.no_grad():
        for it in range(warmup + steps):
            t0 = time.perf_counter()
            unet_forward(xt, p, False, "first")
            dt = time.perf_counter() - t0
            if it >= warmup:
                times.append(dt)
    time_unet_predict.samples = list(times)
    sec = float(np.median(times))
    return B / sec, threads, sec


def time_generate_tail(B=32, d=32, in_ch=1, threads=None, steps=1, warmup=0, full_samples=0):
    """fp32 torch-CPU generate.py:204-236 tail on B synthetic latent vectors: decoder forward -> U-Net forward (eval BN) ->
    argmax / threshold -> connected components (> 3 voxels) -> majority vote + centroids (oracle/watershed_ref.py's
    scipy.ndimage-based convex-branch pass, one sample at a time as the reference does).  The threshold is the 90 %
    quantile of the sigmoid output of a 2-sample call made BEFORE the timed region, as in bench.py's GPU block (random
    weights never reach 0.8).  Like the GPU figure it stops at the components of the convex branch; the hull test and the
    recursive split are timed separately (bench.py inference.generate.refine.cpu_baseline).
    Returns (grids_per_s, threads, median secs/call)."""
    import time
    from . import numpy_ref as R
    from . import watershed_ref as W
    if threads:
        torch.set_num_threads(threads)
    threads = torch.get_num_threads()
    ush, vsh = R.unet_param_shapes(in_ch, 95), R.vae_param_shapes(in_ch, d=d)
    pu = Params(R.init_params(ush, 1, np.float32), R.init_bn_state(ush, np.float32), torch.float32, requires_grad=False)
    pv = Params(R.init_params(vsh, 3, np.float32), R.init_bn_state(vsh, np.float32), torch.float32, requires_grad=False)
    for p in (pu, pv):
        for k, t in list(p.t.items()):
            if t.ndim == 5:
                p.t[k] = t.detach().contiguous(memory_format=torch.channels_last_3d)
    rng = np.random.default_rng(7)
    z = torch.as_tensor(rng.standard_normal((B, 256)), dtype=torch.float32)
    cond = torch.as_tensor(np.eye(10, dtype=np.float32)[np.arange(B) % 10])
    times = []
    with torch.no_grad():
        rec = vae_decode(z[:2], cond[:2], pv, False, d)
        _, sig = unet_forward(rec.contiguous(memory_format=torch.channels_last_3d), pu, False, "first")
        thr = float(np.quantile(sig[:, 0].numpy(), 0.9))
        for it in range(warmup + steps):
            t0 = time.perf_counter()
            rec = vae_decode(z, cond, pv, False, d)
            soft, sig = unet_forward(rec.contiguous(memory_format=torch.channels_last_3d), pu, False, "first")
            species = soft.argmax(1).numpy().astype(np.uint8)
            mask = (sig[:, 0].numpy() >= thr).astype(np.uint8)
            for b in range(B):
                W.watershed_clustering_convex(species[b], mask[b])
            if it >= warmup:
                times.append(time.perf_counter() - t0)
    time_generate_tail.samples = list(times)
    sec = float(np.median(times))
    # ... and what generate.py does with those volumes next (watershed.py:190-203 in full: convex-hull test of every kept
    # component, marker watershed + recursion where one fails) -- the CPU counterpart of bench.py's
    # inference.generate.refine block.  Pure-Python heap flood (oracle/watershed_ref.py), so a BOUNDED sample: the first
    # `full_samples` grids, one at a time as the reference does.
    time_generate_tail.full = None
    if full_samples:
        nf = min(int(full_samples), B)
        t0 = time.perf_counter()
        failed = 0
        for b in range(nf):
            try:
                W.watershed_clustering(None, species[b], mask[b], degenerate="solid")
            except Exception:
                failed += 1
        dt = time.perf_counter() - t0
        time_generate_tail.full = {"grids": nf, "seconds": dt, "s_per_grid": dt / nf, "failed": failed}
    return B / sec, threads, sec
