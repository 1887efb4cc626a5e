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


def maxpool(x, ties):
    y = F.max_pool3d(x, 2)
    if ties == "first":
        return y
    # TF-CPU MaxPool3DGrad semantics: gradient to every element within 1e-5 of the window max
    up = F.interpolate(y.detach(), scale_factor=2, mode="nearest")
    mask = ((x.detach() - up).abs() < 1e-5).to(x.dtype)
    routed = F.avg_pool3d(x * mask, 2) * 8.0
    return y.detach() + routed - routed.detach()


def act(x, kind):
    if kind is None:
        return x
    if kind == "relu":
        return F.relu(x)
    return F.leaky_relu(x, LEAKY)


def bn(x, p, name, training, stats_out=None):
    g, b = p.t[name + "/gamma"], p.t[name + "/beta"]
    if training:
        y = F.batch_norm(x, None, None, g, b, True, 0.0, BN_EPS)
        if stats_out is not None:
            xf = x.detach().transpose(0, 1).reshape(x.shape[1], -1)
            stats_out[name] = (xf.mean(1).numpy(), xf.var(1, unbiased=False).numpy(), xf.shape[1])
        return y
    return F.batch_norm(x, p.s[name + "/moving_mean"], p.s[name + "/moving_var"], g, b, False, 0.0, BN_EPS)


def block(x, p, name, pre, has_bn, post, training, taps=None, stats=None):
    s = act(F.conv3d(x, p.t[name + "/kernel"], p.t[name + "/bias"], padding=1), pre)
    if taps is not None:
        taps[name] = s
    if not has_bn:
        return s
    return act(bn(s, p, name, training, stats), post)


def unet_trunk(x, p, training, ties, upto=None, taps=None, stats=None):
    f = lambda n, t: block(t, p, n, "relu", True, None, training, taps, stats)
    c2 = f("c2", f("c1", x)); p1 = maxpool(c2, ties)
    c4 = f("c4", f("c3", p1)); p2 = maxpool(c4, ties)
    c6 = f("c6", f("c5", p2)); p3 = maxpool(c6, ties)
    c10 = f("c10", f("c9", p3))
    if upto == "c10":
        return c10
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    c14 = f("c14", f("c13", torch.cat([c6, up(c10)], 1)))
    c16 = f("c16", f("c15", torch.cat([c4, up(c14)], 1)))
    return f("c18", f("c17", torch.cat([c2, up(c16)], 1)))


def unet_forward(x, p, training, ties="tf_cpu", stats=None):
    c18 = unet_trunk(x, p, training, ties, stats=stats)
    zs = F.conv3d(c18, p.t["soft/kernel"], p.t["soft/bias"])
    zg = F.conv3d(c18, p.t["sig/kernel"], p.t["sig/bias"])
    return torch.softmax(zs, 1), torch.sigmoid(zg)


def unet_loss(soft, sig, labels, num_classes=95, weight=None, bce_from_logits=False):
    """[Loss, lsoft, lsig] as torch scalars (unet/unet.py:211-219,252-256).  bce_from_logits: the sigmoid head's loss in
    TF 2.1's short-circuit form, sigmoid_cross_entropy_with_logits on the head's logits (confidence M)."""
    weight = float(num_classes) if weight is None else weight
    lab = torch.as_tensor(labels.astype(np.int64))
    y = F.one_hot(lab, num_classes).permute(0, 4, 1, 2, 3).to(soft.dtype)
    t = (lab != 0).to(soft.dtype).unsqueeze(1)
    q = soft / soft.sum(1, keepdim=True)
    qc = torch.clamp(q, K_EPS, 1 - K_EPS)
    lsoft = (-(y * torch.log(qc) * weight).sum(1)).mean(dim=(1, 2, 3)).mean()
    if bce_from_logits:
        lsig = F.binary_cross_entropy_with_logits(torch.logit(sig), t)       # logit(sigmoid(z)) = z
    else:
        pc = torch.clamp(sig, K_EPS, 1 - K_EPS)
        lsig = (-(t * torch.log(pc) + (1 - t) * torch.log(1 - pc))).mean()
    return lsoft + lsig, lsoft, lsig


def unet_step_grads(P, S, x, labels, dtype=torch.float64, ties="tf_cpu", num_classes=95, bce_from_logits=False):
    """Returns (metrics[3], grads dict (numpy, Keras layouts), bn batch stats)."""
    p = Params(P, S, dtype)
    stats = {}
    soft, sig = unet_forward(to_t(x, dtype), p, True, ties, stats)
    loss, lsoft, lsig = unet_loss(soft, sig, labels, num_classes, bce_from_logits=bce_from_logits)
    loss.backward()
    return (np.array([loss.item(), lsoft.item(), lsig.item()]), p.grads_numpy(), stats,
            to_n(soft), to_n(sig))


# ------------------------------------------------------------------------------ VAE
def vae_forward(x, cond, eps, pv, training, in_ch, ncond, d, nf=4, stats=None):
    B = x.shape[0]
    ct = cond.reshape(B, ncond, 1, 1, 1).repeat(1, in_ch, d, d, d)     # K.tile quirk: C*cond channels
    h = torch.cat([x, ct], 1)
    for i in range(nf):
        h = maxpool_first(block(h, pv, "e%d" % i, None, True, "lrelu", training, None, stats), pv)
    h = block(h, pv, "e4", "lrelu", False, None, training)
    flat = h.permute(0, 2, 3, 4, 1).reshape(B, -1)                      # Keras Flatten is (D,H,W,C)
    hd = F.relu(flat @ pv.t["enc_dense/kernel"] + pv.t["enc_dense/bias"])
    zm = hd @ pv.t["z_mean/kernel"] + pv.t["z_mean/bias"]
    zlv = hd @ pv.t["z_log_var/kernel"] + pv.t["z_log_var/bias"]
    z = zm + torch.exp(0.5 * zlv) * eps
    recon = vae_decode(z, cond, pv, training, d, nf, stats)
    return zm, zlv, z, recon


_POOL_TIES = {"mode": "tf_cpu"}


def maxpool_first(x, pv):
    return maxpool(x, _POOL_TIES["mode"])


def vae_decode(z, cond, pv, training, d, nf=4, stats=None):
    B = z.shape[0]
    h = torch.cat([z, cond], 1) @ pv.t["dec_dense/kernel"] + pv.t["dec_dense/bias"]
    s = d // 8
    h = h.reshape(B, s, s, s, 4).permute(0, 4, 1, 2, 3)                 # Keras Reshape is (D,H,W,C)
    for i in range(nf):
        h = block(h, pv, "d%d" % i, None, True, "lrelu", training, None, stats)
        if i < nf - 1:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
    return block(h, pv, "dout", None, True, "relu", training, None, stats)


def vae_losses(x, recon, zm, zlv, pu, training, alpha, beta, ties, pm_w=(1, 1, 1, 1)):
    mse = ((x - recon) ** 2).mean()
    kld = -0.5 * (1 + zlv - zm ** 2 - torch.exp(zlv)).sum(-1)
    t1, t2 = {}, {}
    unet_trunk(x, pu, training, ties, upto="c10", taps=t1)
    unet_trunk(recon, pu, training, ties, upto="c10", taps=t2)
    pm = 0.0
    B = x.shape[0]
    for n, w in zip(("c2", "c4", "c6", "c10"), pm_w):
        pm = pm + w * ((t1[n] - t2[n]).reshape(B, -1) ** 2).mean(-1)
    loss = (mse + alpha * pm + beta * kld).mean()
    return loss, pm.mean(), mse, kld.mean()


def vae_step_grads(Pv, Sv, Pu, Su, x, cond, eps, in_ch=1, ncond=10, d=32, alpha=0.5, beta=3e-4,
                   dtype=torch.float64, ties="tf_cpu", training=True, nf=4):
    _POOL_TIES["mode"] = ties
    pv = Params(Pv, Sv, dtype)
    pu = Params(Pu, Su, dtype, requires_grad=False)
    stats = {}
    xt = to_t(x, dtype)
    ct = torch.as_tensor(cond, dtype=dtype)
    et = torch.as_tensor(eps, dtype=dtype)
    zm, zlv, z, recon = vae_forward(xt, ct, et, pv, training, in_ch, ncond, d, nf, stats)
    loss, pm, mse, kld = vae_losses(xt, recon, zm, zlv, pu, training, alpha, beta, ties)
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
    sec = float(np.mean(times))
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
    sec = float(np.mean(times))
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
    sec = float(np.mean(times))
    return B / sec, threads, sec


def time_generate_tail(B=32, d=32, in_ch=1, threads=None):
    """fp32 torch-CPU generate.py:204-236 tail on B synthetic latent vectors: decoder forward -> U-Net forward (eval BN) ->
    argmax / threshold -> connected components (> 3 voxels) -> majority vote + centroids (oracle/watershed_ref.py's
    scipy.ndimage-based convex-branch pass, one sample at a time as the reference does).  The threshold is the 90 %
    quantile of the sigmoid output, as in bench.py's GPU block (random weights never reach 0.8).
    Returns (grids_per_s, threads, secs/call)."""
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
    with torch.no_grad():
        t0 = time.perf_counter()
        rec = vae_decode(z, cond, pv, False, d)
        soft, sig = unet_forward(rec.contiguous(memory_format=torch.channels_last_3d), pu, False, "first")
        species = soft.argmax(1).numpy().astype(np.uint8)
        sg = sig[:, 0].numpy()
        mask = (sg >= np.quantile(sg, 0.9)).astype(np.uint8)
        for b in range(B):
            W.watershed_clustering_convex(species[b], mask[b])
        sec = time.perf_counter() - t0
    return B / sec, threads, sec
