"""round 6 diagnostics on the GPU box (one-off): (a) where the fp64 torch oracle's time goes at B = 8 and how it scales with
threads, (b) the DFC-VAE e0/kernel gradient at B = 32 by row group, default vs ICSG3D_NO_COND_FOLD / _NO_THIN_C."""
import os, subprocess, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
what = sys.argv[1]

if what == "threads":
    import torch
    from oracle import numpy_ref as R, torch_ref as T
    B, d = 8, 32
    X, lab, _ = R.synthetic_batch(B, d, 1, seed=0, dtype=np.float64)
    sh = R.unet_param_shapes(1, 95)
    P, S = R.init_params(sh, 1), R.init_bn_state(sh)
    p32 = T.Params(P, S, torch.float32, requires_grad=False)
    taps = {}
    with torch.no_grad():
        T.unet_trunk(T.to_t(X, torch.float32), p32, True, "tf_cpu", taps=taps)
    kink = {n: T.to_n(t) for n, t in taps.items()}
    for nt in (16, 32, 64, 128):
        torch.set_num_threads(nt)
        t0 = time.time(); T.unet_step_grads(P, S, X, lab, kink=kink, kink_tol=1e-3, want_outputs=False); dt = time.time() - t0
        t0 = time.time(); T.unet_step_grads(P, S, X, lab, want_outputs=False); dt2 = time.time() - t0
        print("threads %3d: pinned step %.1f s, un-pinned %.1f s" % (nt, dt, dt2), flush=True)
    torch.set_num_threads(32)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        T.unet_step_grads(P, S, X, lab, kink=kink, kink_tol=1e-3, want_outputs=False)
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=22, max_name_column_width=44))

elif what == "e0":
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    from oracle import numpy_ref as R
    B, d = 32, 32
    X, _, cond = R.synthetic_batch(B, d, 1, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    eps = np.random.default_rng(2).standard_normal((B, 256))
    ush, vsh = R.unet_param_shapes(1, 95), R.vae_param_shapes(1, d=d)
    Pu, Pv = R.init_params(ush, 1), R.init_params(vsh, 3)
    ue = UnetEngine(in_channels=1, d=d, max_batch=B); ue.set_weights(Pu)
    ve = VaeEngine(ue, in_channels=1, d=d, max_batch=B, lr=5e-4); ve.set_weights(Pv)
    m = ve.train_step(X, cond.astype(np.float64), eps)
    g = {n: ve.get_grad(n, s) for n, s, tr in ve.tensor_infos() if tr and n.split("/")[0] in ("e0", "e1", "d3", "dout")}
    np.savez(os.path.join(ROOT, "gpurun_out", "r6_e0_%s.npz" % os.environ.get("TAG", "default")), metrics=m,
             **{k.replace("/", "__"): v for k, v in g.items()})
    print(os.environ.get("TAG", "default"), m)

elif what == "e0ref":
    from oracle import numpy_ref as R, torch_ref as T
    B, d = 32, 32
    X, _, cond = R.synthetic_batch(B, d, 1, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    eps = np.random.default_rng(2).standard_normal((B, 256))
    ush, vsh = R.unet_param_shapes(1, 95), R.vae_param_shapes(1, d=d)
    Pu, Su, Pv, Sv = R.init_params(ush, 1), R.init_bn_state(ush), R.init_params(vsh, 3), R.init_bn_state(vsh)
    m, g, _, _, _, _ = T.vae_step_grads(Pv, Sv, Pu, Su, X, cond.astype(np.float64), eps, in_ch=1, d=d)
    np.savez(os.path.join(ROOT, "gpurun_out", "r6_e0_ref.npz"), metrics=m,
             **{k.replace("/", "__"): v for k, v in g.items() if k.split("/")[0] in ("e0", "e1", "d3", "dout")})
    print("ref", m)
