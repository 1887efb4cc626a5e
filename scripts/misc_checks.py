"""GPU sanity checks outside the pytest suite: RCCL single-rank communicator, VAE B=32 timing, d=64."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icsg3d_amd.engine import UnetEngine, VaeEngine, comm_unique_id
from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes

what = sys.argv[1:] or ["comm", "vae", "d64"]
if "comm" in what:
    B, d = 2, 16
    X, lab, _ = synthetic_batch(B, d, 1, seed=0, noise=1e-3)
    P = glorot_params(unet_param_shapes(1, 95), 1)
    a = UnetEngine(d=d, max_batch=B, lr=1e-3); a.set_weights(P)
    b = UnetEngine(d=d, max_batch=B, lr=1e-3); b.set_weights(P)
    b.comm_init(0, 1, comm_unique_id())
    ma, mb = a.train_step(X, lab), b.train_step(X, lab)
    wa, wb = a.get_weights(), b.get_weights()
    same = all(np.array_equal(wa[k], wb[k]) for k in wa)
    print("comm(nranks=1): metrics equal", np.array_equal(ma, mb), "weights equal", same, "max-reduce", b.allreduce_max(3.5))
if "vae" in what:
    B, d = 32, 32
    ue = UnetEngine(d=d, max_batch=B); ue.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
    ve = VaeEngine(ue, d=d, max_batch=B); ve.set_weights(glorot_params(vae_param_shapes(1), 3))
    X, _, cond = synthetic_batch(B, d, 1, seed=0)
    eps = np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32)
    ve.upload_batch(X, cond, eps)
    print("vae metrics", ve.train_step_resident(True))
    ve.sync(); t0 = time.perf_counter()
    for _ in range(5): ve.train_step_resident(False)
    ve.sync(); dt = (time.perf_counter() - t0) / 5
    print("DFC-VAE step B=32 d=32: %.2f ms  %.1f grids/s" % (dt * 1e3, B / dt))
    ve.profile_enable(True); ue.profile_enable(True)
    for _ in range(2): ve.train_step_resident(False)
    ve.sync()
    urows = sorted(ue.profile_rows(), key=lambda r: -r["ms"])
    print("profiled perceptual U-Net ms/step %.2f" % (sum(r["ms"] for r in urows) / 2))
    for r in urows[:30]:
        print("  PM %-44s n=%3d %8.3f ms/step" % (r["label"], r["launches"], r["ms"] / 2))
    rows = sorted(ve.profile_rows(), key=lambda r: -r["ms"])
    print("profiled VAE-engine ms/step %.2f (perceptual U-Net launches are not in these rows)" % (sum(r["ms"] for r in rows) / 2))
    for r in rows[:40]:
        print("  %-44s n=%3d %8.3f ms/step" % (r["label"], r["launches"], r["ms"] / 2))
if "d64" in what:
    B, d = 2, 64
    e = UnetEngine(d=d, max_batch=B, lr=3e-6); e.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
    X, lab, _ = synthetic_batch(B, d, 1, seed=0)
    e.upload_batch(X, lab)
    print("d=64 metrics", e.train_step_resident(True))
    e.sync(); t0 = time.perf_counter()
    for _ in range(2): e.train_step_resident(False)
    e.sync(); dt = (time.perf_counter() - t0) / 2
    print("U-Net d=64 B=2: %.1f ms/step, %.2f grids/s (%.1f TFLOP/s)" % (dt * 1e3, B / dt, 3 * 1007.09e9 * B / dt / 1e12))
