"""Randomised whole-network parity sweep on the GPU (not a test: a hunt).  Every trial builds a U-Net engine (and, every
second trial, a DFC-VAE engine on top of it) at a random (d, C, max_batch), then runs a few CONSECUTIVE train steps at
random batch sizes <= max_batch -- a handle used at 5 grids after 8 after 1 is where stale workspace, split plans and
BatchNorm partials of a previous batch size would show -- and checks each step against oracle/torch_ref.py in fp64 from the
engine's own weights before the step (metrics, every gradient tensor; the engine's ReLU / pool / clip decisions pinned as in
tests/test_gpu_fullsize_oracle.py).  Prints one line per step and a summary; exit code 1 if any bound is exceeded.

Every trial runs with guard bytes behind the device buffers (ICSG3D_DEBUG_CANARY=1) and a third of them with one or two of
the ICSG3D_NO_* A/B switches.

    python tests/tools/fuzz_steps.py [trials=12] [seed=0]        (FUZZ_ONLY=3,7: only those trials of the same sequence)
Reference: /root/reference/unet/unet.py:272-355,370, vae/lattice_vae.py:160-270,296."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import numpy_ref as R          # noqa: E402
from oracle import torch_ref as T          # noqa: E402

GRAD_TOL, HEAD_TOL, VAE_GRAD_TOL, FWD_TOL, VAE_FWD_TOL, KINK_TOL = 6e-5, 5e-4, 1e-4, 1e-5, 3e-5, 1e-4
UNET_LAYERS = [n for n, _, _ in R.UNET_CONVS]
# the A/B switches of tests/test_gpu_switches.py: a third of the trials run with one or two of them (the general kernels
# behind a fast path have their own split plans and workspaces)
SWITCHES = ["ICSG3D_NO_REUSE", "ICSG3D_NO_WGRAD3", "ICSG3D_NO_FWD_SPLITK", "ICSG3D_NO_THIN_N", "ICSG3D_NO_UPSPLIT",
            "ICSG3D_NO_THIN_C", "ICSG3D_NO_BWD_FOLD", "ICSG3D_NO_WGRAD3S", "ICSG3D_NO_COND_FOLD", "ICSG3D_NO_WINO",
            "ICSG3D_NO_WINO64", "ICSG3D_NO_UP3", "ICSG3D_NO_WINO_WGRAD", "ICSG3D_NO_FUSED_HEAD", "ICSG3D_NO_FAST_BNBWD",
            "ICSG3D_NO_THIN1_2STAGE", "ICSG3D_NO_WINOG", "ICSG3D_NO_HEAD_BNFUSE", "ICSG3D_NO_UP3N", "ICSG3D_NO_POOL_PRESUM",
            "ICSG3D_DGRAD_BNFUSE_MIN=0", "ICSG3D_NO_DGRAD_BNFUSE", "ICSG3D_NO_PM_SIDE", "ICSG3D_NO_VAE_SIDE_WGRAD",
            "ICSG3D_NO_HEAD_LABELS"]
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


def unet_step(eng, B, d, C, seed, resident=False):
    X, lab, _, _ = inputs(B, d, C, seed)
    P, S = engine_state(eng)
    if resident:                      # the benchmark's form: upload, then a step on the resident batch
        eng.upload_batch(X, lab)
        m = eng.train_step_resident(True)
    else:
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
    line = "unet d=%d C=%d B=%d: loss err %.1e, grads %.1e (head %.1e; worst %s), %d decisions pinned" % (
        d, C, B, fwd, body, head, worst, flips)
    if bad and fwd <= FWD_TOL:
        # Is it the engine or fp32?  The same oracle in fp32 (same pins, torch's own kernels) against itself in fp64: a
        # gradient the engine misses by no more than a few times what plain fp32 arithmetic misses it by is conditioning
        # (a head that has become confident: every dz is a difference of nearly equal numbers), not a defect.
        import torch
        _, g32, _, _, _ = T.unet_step_grads(P, S, X, lab, kink=kink, affine=affine, clip_pin=clip_pin, want_outputs=False,
                                            kink_tol=1.0, dtype=torch.float32)
        e32 = {n: grad_err(g32[n], g_ref[n], g_ref, n) for n in grads}
        over = {n: (errs[n], e32[n]) for n in errs
                if errs[n] > (HEAD_TOL if n.split("/")[0] in ("soft", "sig") else GRAD_TOL)}
        cond = all(e <= 4 * f for e, f in over.values())
        line += "; over the bound (engine, fp32 torch): %s -> %s" % (
            {n: "%.1e / %.1e" % v for n, v in over.items()}, "fp32 conditioning" if cond else "ENGINE")
        bad = not cond
    return bad, line


def vae_step(ve, ue, B, d, C, seed, resident=False):
    X, _, cond, eps = inputs(B, d, C, seed)
    Pv, Sv = engine_state(ve)
    Pu, Su = engine_state(ue)
    if resident:
        ve.upload_batch(X, cond, eps)
        m = ve.train_step_resident(True)
    else:
        m = ve.train_step(X, cond, eps)
    grads = {name: ve.get_grad(name, shape) for name, shape, tr in ve.tensor_infos() if tr}
    vs = vae_shapes(B, d, C)
    ps = {n: ushape(n, B, d) for n in UNET_LAYERS[:8]}
    kink = {n: ve.get_activation(n, s) for n, s in vs.items()}
    kink_pm = {n: ue.get_activation(n, s) for n, s in ps.items()}
    aff = {n: ve.get_bn_affine(n, vs[n][-1]) for n in ("e0", "e1", "e2", "e3", "d0", "d1", "d2", "d3", "dout")}
    aff_pm = {n: ue.get_bn_affine(n, ps[n][-1]) for n in ("c2", "c4", "c6")}
    cap = {} if os.environ.get("FUZZ_DY") else None
    T.GRAD_CAPTURE = cap
    m_ref, g_ref, _, _, _, _ = T.vae_step_grads(Pv, Sv, Pu, Su, X, cond, eps, in_ch=C, d=d, kink=kink, kink_pm=kink_pm,
                                                affine=aff, affine_pm=aff_pm, kink_tol=KINK_TOL)
    T.GRAD_CAPTURE = None
    if cap is not None:      # where along the backward chain does the engine leave the oracle?  dLoss/ds per layer, in backward order
        order = ["c10", "c9", "c6", "c5", "c4", "c3", "c2", "c1", "dout", "d3", "d2", "d1", "d0", "e4", "e3", "e2", "e1", "e0"]
        for n in order:
            if n not in cap:
                continue
            ref = T.to_n(cap[n])
            got = (ue if n.startswith("c") else ve).get_activation(n + ":dy", ref.shape)
            print("    dy %-5s engine vs fp64: %.1e  (max|dy| %.2e)" % (n, np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300),
                                                                    np.abs(ref).max()), flush=True)
    fwd = float((np.abs(np.asarray(m, np.float64) - m_ref) / np.maximum(np.abs(m_ref), 1e-300)).max())
    gscale = max(np.abs(g).max() for g in g_ref.values())
    errs = {n: grad_err(g, g_ref[n], g_ref, n, floor=1e-6 * gscale) for n, g in grads.items()}
    worst = max(errs, key=errs.get)
    if cap is not None:
        print("    per-tensor: %s" % {k: "%.1e" % v for k, v in errs.items()}, flush=True)
    bad = fwd > VAE_FWD_TOL or errs[worst] > VAE_GRAD_TOL
    line = "vae  d=%d C=%d B=%d: metric err %.1e, grads %.1e (worst %s), %d decisions pinned" % (
        d, C, B, fwd, errs[worst], worst, sum(T.vae_step_grads.flips.values()))
    if bad and fwd <= VAE_FWD_TOL:
        import torch
        _, g32, _, _, _, _ = T.vae_step_grads(Pv, Sv, Pu, Su, X, cond, eps, in_ch=C, d=d, kink=kink, kink_pm=kink_pm,
                                              affine=aff, affine_pm=aff_pm, kink_tol=1.0, dtype=torch.float32)
        over = {n: (errs[n], grad_err(g32[n], g_ref[n], g_ref, n, floor=1e-6 * gscale)) for n in errs if errs[n] > VAE_GRAD_TOL}
        cond_ = all(e <= 4 * f for e, f in over.values())
        line += "; over the bound (engine, fp32 torch): %s -> %s" % (
            {n: "%.1e / %.1e" % v for n, v in over.items()}, "fp32 conditioning" if cond_ else "ENGINE")
        bad = not cond_
    return bad, line


def infer_checks(ue, ve, Bi, d, C, seed):
    """Inference-mode entry points with the moving statistics the train steps left, on Bi grids (Bi may exceed max_batch:
    the host classes stream chunks): model.predict / test_on_batch of the U-Net; encoder / decoder / test_on_batch of the VAE."""
    import torch
    X, lab, cond, eps = inputs(Bi, d, C, seed)
    Pu, Su = engine_state(ue)
    pu = T.Params(Pu, Su, torch.float64, requires_grad=False)
    with torch.no_grad():
        soft_t, sig_t = T.unet_forward(T.to_t(X, torch.float64), pu, False, "tf_cpu")
    soft, sig = ue.predict(X)
    e_soft = float(np.abs(soft - T.to_n(soft_t)).max())
    e_sig = float(np.abs(sig - T.to_n(sig_t)).max())
    out = ["predict B=%d: soft %.1e sig %.1e" % (Bi, e_soft, e_sig)]
    bad = e_soft > 1e-5 or e_sig > 1e-5
    b = min(Bi, ue.max_batch)
    with torch.no_grad():
        loss, lsoft, lsig = T.unet_loss(soft_t[:b], sig_t[:b], lab[:b])
    m = ue.test_step(X[:b], lab[:b])
    ref = np.array([loss.item(), lsoft.item(), lsig.item()])
    e_m = float((np.abs(m[:3] - ref) / np.abs(ref)).max())
    out.append("test_step B=%d: %.1e" % (b, e_m))
    bad = bad or e_m > 2e-5
    margin_p = margin_s = 1e-4
    if bad:
        # engine or fp32?  the same forward and loss by torch in fp32 against fp64 (moving statistics a few steps old do not
        # normalise: activations grow through 14 layers, the heads saturate, and K.clip's upper bound 1 - 1e-7 is 1 - 1.19e-7 in fp32)
        p32 = T.Params(Pu, Su, torch.float32, requires_grad=False)
        with torch.no_grad():
            s32, g32 = T.unet_forward(T.to_t(X, torch.float32), p32, False, "tf_cpu")
            l32 = [v.item() for v in T.unet_loss(s32[:b], g32[:b], lab[:b])]
        f_soft = float(np.abs(T.to_n(s32) - T.to_n(soft_t)).max())
        f_sig = float(np.abs(T.to_n(g32) - T.to_n(sig_t)).max())
        f_m = float((np.abs(np.array(l32) - ref) / np.abs(ref)).max())
        e_m32 = float((np.abs(m[:3] - np.array(l32)) / np.abs(l32)).max())
        limited = (e_soft <= max(1e-5, 4 * f_soft) and e_sig <= max(1e-5, 4 * f_sig) and (e_m <= 2e-5 or e_m32 <= 2e-5 or e_m <= 4 * f_m))
        out.append("fp32 torch vs fp64: soft %.1e sig %.1e loss %.1e; engine vs fp32 torch loss %.1e -> %s"
                   % (f_soft, f_sig, f_m, e_m32, "fp32 conditioning" if limited else "ENGINE"))
        bad = not limited
        margin_p, margin_s = max(1e-4, 4 * f_soft), max(1e-4, 4 * f_sig)     # a label may differ where fp32 cannot tell
    sp, mk = ue.predict_labels(X, thresh=0.8)
    pr, sg = T.to_n(soft_t), T.to_n(sig_t)[..., 0]
    top2 = np.partition(pr, -2, axis=-1)[..., -2:]
    sure = (top2[..., 1] - top2[..., 0]) > margin_p
    n_sp = int((sp[sure] != pr.argmax(-1)[sure]).sum())
    sure_m = np.abs(sg - 0.8) > margin_s
    n_mk = int(((mk != 0)[sure_m] != (sg >= 0.8)[sure_m]).sum())
    out.append("labels: %d / %d differ outside the margins (%.0e / %.0e)" % (n_sp, n_mk, margin_p, margin_s))
    bad = bad or n_sp or n_mk
    if ve is not None:
        Pv, Sv = engine_state(ve)
        b = min(Bi, ve.max_batch)
        m_ref, _, _, recon, zm, zlv = T.vae_step_grads(Pv, Sv, Pu, Su, X[:b], cond[:b], eps[:b], in_ch=C, d=d, training=False)
        zm_e, zlv_e, z_e = ve.encode(X[:b], cond[:b], eps[:b])
        rec_e = ve.decode(z_e, cond[:b])
        m = ve.test_step(X[:b], cond[:b], eps[:b])
        e = [float(np.abs(zm_e - zm).max() / max(np.abs(zm).max(), 1e-30)), float(np.abs(zlv_e - zlv).max() / max(np.abs(zlv).max(), 1e-30)),
             float(np.abs(rec_e - recon).max() / max(np.abs(recon).max(), 1e-30)),
             float((np.abs(np.asarray(m, np.float64) - m_ref) / np.maximum(np.abs(m_ref), 1e-300)).max())]
        out.append("vae B=%d: z_mean %.1e z_log_var %.1e recon %.1e test_step %.1e" % (b, *e))
        bad = bad or max(e[:3]) > 1e-5 or e[3] > 3e-5
    return bool(bad), "infer d=%d C=%d: " % (d, C) + "; ".join(out)


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
    os.environ["ICSG3D_DEBUG_CANARY"] = "1"      # guard bytes behind every device buffer, checked at the end of each trial
    for t in range(trials):
        d = int(rng.choice([16, 16, 16, 32, 32, 64]))
        C = int(rng.choice([1, 4]))
        maxB = int(rng.integers(2, {16: 41, 32: 10, 64: 4}[d]))
        sw = [str(v) for v in rng.choice(SWITCHES, size=int(rng.integers(1, 3)), replace=False)] if rng.random() < 0.34 else []
        seed = int(rng.integers(1, 1 << 20))
        ush = R.unet_param_shapes(C, 95)
        lr = float(rng.choice([1e-3, 1e-4]))
        batches = [maxB] + [int(b) for b in rng.integers(1, maxB + 1, size=3)]
        rng.shuffle(batches)
        Bi = min(int(rng.integers(1, 2 * maxB + 2)), {16: 24, 32: 6, 64: 2}[d])    # grids of the inference checks
        comm, res = 0, [False] * 4
        if not os.environ.get("FUZZ_PLAIN"):      # (FUZZ_PLAIN=1: the draw sequence of the runs in profiles/r6_fuzz.txt before these two existed)
            comm = int(rng.integers(0, 4))        # 1: a single-rank RCCL communicator (bucketed all-reduce path); 2: + SyncBN
            res = [bool(v) for v in rng.integers(0, 2, size=4)]      # which steps take the resident form
        if os.environ.get("FUZZ_ONLY") and str(t) not in os.environ["FUZZ_ONLY"].split(","):
            continue
        if "FUZZ_SW" in os.environ:                     # override the drawn switches: "none" or a comma list
            sw = [v for v in os.environ["FUZZ_SW"].split(",") if v and v != "none"]
        for k in SWITCHES:
            os.environ.pop(k.split("=")[0], None)
        for k in sw:                                  # read per handle at creation
            os.environ[k.split("=")[0]] = k.split("=")[1] if "=" in k else "1"
        ue = UnetEngine(in_channels=C, d=d, max_batch=maxB, lr=lr)
        ue.set_weights(R.init_params(ush, seed))
        with_vae = t % 2 == 1
        if with_vae:
            ve = VaeEngine(ue, in_channels=C, d=d, max_batch=maxB, lr=5e-4)
            ve.set_weights(R.init_params(R.vae_param_shapes(C, d=d), seed + 1))
        if comm in (1, 2):
            from icsg3d_amd.engine import comm_unique_id
            for e in ([ue, ve] if with_vae else [ue]):
                e.comm_init(0, 1, comm_unique_id()); e.broadcast_state(0); e.set_sync_bn(comm == 2)
        print("trial %d: d=%d C=%d max_batch=%d lr=%g batches %s%s %s%s resident %s" % (
            t, d, C, maxB, lr, batches, " + DFC-VAE" if with_vae else "", " ".join(sw),
            {1: " comm", 2: " comm+syncbn"}.get(comm, ""), [int(v) for v in res]), flush=True)
        for i, B in enumerate(batches):
            try:
                if with_vae and i % 2 == 0:
                    bad, line = vae_step(ve, ue, B, d, C, seed + 10 * i, res[i])
                else:
                    bad, line = unet_step(ue, B, d, C, seed + 10 * i, res[i])
            except AssertionError as e:      # the oracle refuses a decision far from its kink: the forward pass differs
                bad, line = True, "d=%d C=%d B=%d: %s" % (d, C, B, str(e)[:200])
            nbad += bad
            nstep += 1
            print(("  FAIL " if bad else "  ok   ") + line, flush=True)
        try:
            bad, line = infer_checks(ue, ve if with_vae else None, Bi, d, C, seed + 77)
        except AssertionError as e:
            bad, line = True, "infer d=%d C=%d B=%d: %s" % (d, C, Bi, str(e)[:200])
        nbad += bad
        nstep += 1
        print(("  FAIL " if bad else "  ok   ") + line, flush=True)
        for nm, e in (("vae", ve if with_vae else None), ("unet", ue)):
            if e is not None:
                dirty, what = e.check_canaries()
                if dirty:
                    nbad += 1
                    print("  FAIL %s handle: %d guards written: %s" % (nm, dirty, what[:300]), flush=True)
        if with_vae:
            ve.close()
        ue.close()
    print("fuzz_steps: %d steps in %d trials, %d outside the bounds (%.0f s)" % (nstep, trials, nbad, time.time() - t0))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
