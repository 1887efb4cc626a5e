"""Developer probe: cProfile of icsg3d_amd.watershed.refine_atoms on the random-weight masks bench.py's generate block uses
(every sample has a non-convex component there: the worst case of the host recursion).  python scripts/prof_refine.py"""
import sys, time, cProfile, pstats, numpy as np
sys.path.insert(0, '.')
from icsg3d_amd.engine import UnetEngine, VaeEngine
from icsg3d_amd.synthetic import glorot_params, unet_param_shapes, vae_param_shapes
from icsg3d_amd.watershed import refine_atoms
B, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 32
ue = UnetEngine(d=d, max_batch=B); ue.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
ve = VaeEngine(ue, d=d, max_batch=B); ve.set_weights(glorot_params(vae_param_shapes(1, d=d), 3))
rng = np.random.default_rng(7)
z = rng.standard_normal((B, 256)).astype(np.float32); cnd = np.eye(10, dtype=np.float32)[np.arange(B) % 10]
thr = float(np.quantile(ue.predict(ve.decode(z[:2], cnd[:2]))[1], 0.9))
out = ve.decode_to_atoms(ue, z, cnd, thresh=thr, max_atoms=4096, want_regions=True)
refine_atoms(ve.decode_to_atoms(ue, z, cnd, thresh=thr, max_atoms=4096, want_regions=True), degenerate="solid")
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); refine_atoms(out, degenerate="solid"); dt = time.perf_counter() - t0
pr.disable()
print("refine %d samples: %.1f ms, split %s, atoms %s" % (B, dt * 1e3, out["split"].sum(), out["n_atoms"].tolist()))
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
import os
for dev in ("1", "0"):          # the split on the device (one GPU lane per box) against the host-thread form
    os.environ["ICSG3D_WS_DEVICE"] = dev
    o2 = ve.decode_to_atoms(ue, z, cnd, thresh=thr, max_atoms=4096, want_regions=True)
    t0 = time.perf_counter(); refine_atoms(o2, degenerate="solid"); dt = time.perf_counter() - t0
    print("ICSG3D_WS_DEVICE=%s: refine %d samples %.1f ms" % (dev, B, dt * 1e3))
