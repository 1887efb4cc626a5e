"""Randomised whole-network parity sweep on the GPU (not a test: a hunt).  Every trial builds a U-Net engine (and, every
second trial, a DFC-VAE engine on top of it) at a random (d, C, max_batch), then runs a few CONSECUTIVE train steps at
random batch sizes <= max_batch -- a handle used at 5 grids after 8 after 1 is where stale workspace, split plans and
BatchNorm partials of a previous batch size would show -- and checks each step against oracle/torch_ref.py in fp64 from the
engine's own weights before the step (metrics, every gradient tensor; the engine's ReLU / pool / clip decisions pinned as in
tests/test_gpu_fullsize_oracle.py).  Prints one line per step and a summary; exit code 1 if any bound is exceeded.

    python scripts/fuzz_steps.py [trials=12] [seed=0]
Reference: /root/reference/unet/unet.py:272-355,370, vae/lattice_vae.py:160-270,296."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import numpy_ref as R          # noqa: E402
from oracle import torch_ref as T          # noqa: E402

GRAD_TOL, HEAD_TOL, VAE_GRAD_TOL, FWD_TOL, VAE_FWD_TOL, KINK_TOL = 6e-5, 5e-4, 1e-4, 1e-5, 3e-5, 1e-4
UNET_LAYERS = [n for n, _, _ in R.UNET_CONVS]
RES = {"c1": 1, "c2": 1, "c3": 2, "c4": 2, "c5": 4, "c6": 4, "c9": 8, "c10": 8, "c13": 4, "c14": 4,
       "c15": 2, "c16": 2, "c17": 1, "c18": 1}
COUT = dict((n, c) for n, _, c in R.UNET_CONVS)


def ushape(n, B, d):
    S = d // RES[n]
    return (B, S, S, S, COUT[n])


def vae_shapes(B, d, C):
    sh, S, f = {}, d, (16, 32, 64, 128)
    for i in range(4):
        sh["e%d" % i] = (B, S, S, S, f[i]); S //= 2
    sh["e4"] = (B, S, S, S, 4)
    sh["enc_dense"] = (B, 256)
    S = d // 8
    for i in range(4):
        sh["d%d" % i] = (B, S, S, S, f[3 - i])
        if i < 3:
            S *= 2
    sh["dout"] = (B, d, d, d, C)
    return sh


def grad_err(g, ref, ref_all, name, floor=0.0):
    scale = np.abs(ref).max()
    if name.endswith("/bias") and name[:-4] + "kernel" in ref_all:
        scale = max(scale, np.abs(ref_all[name[:-4] + "kernel"]).max())
    return float(np.abs(np.asarray(g, np.float64) - ref).max() / max(scale, floor, 1e-300))


def engine_state(eng):
    P, S = {}, {}
    for name, shape, tr in eng.tensor_infos():
        v = eng.get_tensor(name, shape).astype(np.float64)
        (S if name.endswith(("/moving_mean", "/moving_var")) else P)[name] = v
    return P, S


def inputs(B, d, C, seed):
    X, lab, cond = R.synthetic_batch(B, d, C, seed=seed, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(seed + 5).uniform(size=X.shape)
    eps = np.random.default_rng(seed + 2).standard_normal((B, 256))
    return X, lab, cond.astype(np.float64), eps


def unet_step(eng, B, d, C, seed):
    X, lab, _, _ = inputs(B, d, C, seed)
    P, S = engine_state(eng)
    m = eng.train_step(X, lab)
    grads = {name: eng.get_grad(name, shape) for name, shape, tr in eng.tensor_infos() if tr}
    kink = {n: eng.get_activation(n, ushape(n, B, d)) for n in UNET_LAYERS}
    affine = {n: eng.get_bn_affine(n, COUT[n]) for n in ("c2", "c4", "c6")}
    dz = eng.get_activation("head", (B, d, d, d, 96))
    clip_pin = {"sig": dz[..., 95] != 0, "soft": np.any(dz[..., :95] != 0, axis=-1)}
    m_ref, g_ref, _, _, _ = T.unet_step_grads(P, S, X, lab, kink=kink, affine=affine, clip_pin=clip_pin, want_outputs=False,
                                              kink_tol=KINK_TOL)
    fwd = float(np.abs(np.asarray(m[:3], np.float64) - m_ref).max() / np.abs(m_ref).max())
    errs = {n: grad_err(g, g_ref[n], g_ref, n) for n, g in grads.items()}
    body = max(e for n, e in errs.items() if n.split("/")[0] not in ("soft", "sig"))
    head = max(e for n, e in errs.items() if n.split("/")[0] in ("soft", "sig"))
    flips = sum(T.unet_step_grads.flips["kink"].values())
    bad = fwd > FWD_TOL or body > GRAD_TOL or head > HEAD_TOL
    worst = max(errs, key=errs.get)
    return bad, "unet d=%d C=%d B=%d: loss err %.1e, grads %.1e (head %.1e; worst %s), %d decisions pinned" % (
        d, C, B, fwd, body, head, worst, flips)


def vae_step(ve, ue, B, d, C, seed):
    X, _, cond, eps = inputs(B, d, C, seed)
    Pv, Sv = engine_state(ve)
    Pu, Su = engine_state(ue)
    m = ve.train_step(X, cond, eps)
    grads = {name: ve.get_grad(name, shape) for name, shape, tr in ve.tensor_infos() if tr}
    vs = vae_shapes(B, d, C)
    ps = {n: ushape(n, B, d) for n in UNET_LAYERS[:8]}
    kink = {n: ve.get_activation(n, s) for n, s in vs.items()}
    kink_pm = {n: ue.get_activation(n, s) for n, s in ps.items()}
    aff = {n: ve.get_bn_affine(n, vs[n][-1]) for n in ("e0", "e1", "e2", "e3", "d0", "d1", "d2", "d3", "dout")}
    aff_pm = {n: ue.get_bn_affine(n, ps[n][-1]) for n in ("c2", "c4", "c6")}
    m_ref, g_ref, _, _, _, _ = T.vae_step_grads(Pv, Sv, Pu, Su, X, cond, eps, in_ch=C, d=d, kink=kink, kink_pm=kink_pm,
                                                affine=aff, affine_pm=aff_pm, kink_tol=KINK_TOL)
    fwd = float((np.abs(np.asarray(m, np.float64) - m_ref) / np.maximum(np.abs(m_ref), 1e-300)).max())
    gscale = max(np.abs(g).max() for g in g_ref.values())
    errs = {n: grad_err(g, g_ref[n], g_ref, n, floor=1e-6 * gscale) for n, g in grads.items()}
    worst = max(errs, key=errs.get)
    bad = fwd > VAE_FWD_TOL or errs[worst] > VAE_GRAD_TOL
    return bad, "vae  d=%d C=%d B=%d: metric err %.1e, grads %.1e (worst %s), %d decisions pinned" % (
        d, C, B, fwd, errs[worst], worst, sum(T.vae_step_grads.flips.values()))


def main():
    import torch
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    torch.set_num_threads(min(32, os.cpu_count() or 32))
    nbad = nstep = 0
    t0 = time.time()
    if os.environ.get("FUZZ_CASES"):       # explicit cases "vae,32,4,5,3;unet,16,1,8,5": kind, d, C, max_batch, B (first step of a fresh engine)
        for case in os.environ["FUZZ_CASES"].split(";"):
            kind, d, C, maxB, B = case.split(",")
            d, C, maxB, B = int(d), int(C), int(maxB), int(B)
            ue = UnetEngine(in_channels=C, d=d, max_batch=maxB, lr=1e-3)
            ue.set_weights(R.init_params(R.unet_param_shapes(C, 95), 7))
            try:
                if kind == "vae":
                    ve = VaeEngine(ue, in_channels=C, d=d, max_batch=maxB, lr=5e-4)
                    ve.set_weights(R.init_params(R.vae_param_shapes(C, d=d), 8))
                    bad, line = vae_step(ve, ue, B, d, C, 11)
                    ve.close()
                else:
                    bad, line = unet_step(ue, B, d, C, 11)
            except AssertionError as e:
                bad, line = True, "%s d=%d C=%d max_batch=%d B=%d: %s" % (kind, d, C, maxB, B, str(e)[:160])
            ue.close()
            nbad += bad
            print(("  FAIL " if bad else "  ok   ") + line + "  [max_batch %d]" % maxB, flush=True)
        sys.exit(1 if nbad else 0)
    for t in range(trials):
        d = int(rng.choice([16, 16, 32]))
        C = int(rng.choice([1, 4]))
        maxB = int(rng.integers(2, 41 if d == 16 else 10))
        seed = int(rng.integers(1, 1 << 20))
        ush = R.unet_param_shapes(C, 95)
        ue = UnetEngine(in_channels=C, d=d, max_batch=maxB, lr=float(rng.choice([1e-3, 1e-4])))
        ue.set_weights(R.init_params(ush, seed))
        with_vae = t % 2 == 1
        if with_vae:
            ve = VaeEngine(ue, in_channels=C, d=d, max_batch=maxB, lr=5e-4)
            ve.set_weights(R.init_params(R.vae_param_shapes(C, d=d), seed + 1))
        batches = [maxB] + [int(b) for b in rng.integers(1, maxB + 1, size=3)]
        rng.shuffle(batches)
        print("trial %d: d=%d C=%d max_batch=%d batches %s%s" % (t, d, C, maxB, batches, " + DFC-VAE" if with_vae else ""),
              flush=True)
        for i, B in enumerate(batches):
            try:
                if with_vae and i % 2 == 0:
                    bad, line = vae_step(ve, ue, B, d, C, seed + 10 * i)
                else:
                    bad, line = unet_step(ue, B, d, C, seed + 10 * i)
            except AssertionError as e:      # the oracle refuses a decision far from its kink: the forward pass differs
                bad, line = True, "d=%d C=%d B=%d: %s" % (d, C, B, str(e)[:200])
            nbad += bad
            nstep += 1
            print(("  FAIL " if bad else "  ok   ") + line, flush=True)
        if with_vae:
            ve.close()
        ue.close()
    print("fuzz_steps: %d steps in %d trials, %d outside the bounds (%.0f s)" % (nstep, trials, nbad, time.time() - t0))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
