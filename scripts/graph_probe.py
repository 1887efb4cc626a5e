"""hipGraph replay of the train steps, re-measured: round 1 measured 69.2 vs 69.0 ms on a U-Net step whose
kernels were 2.5x longer; the DFC-VAE step of round 6 is ~200 launches of which ~150 are small.  ics_net_graph_probe captures
the resident step from the engine's stream (two streams, fork / join events included) and replays it.
  python scripts/graph_probe.py [B d] > profiles/r6_graph_probe.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icsg3d_amd.engine import UnetEngine, VaeEngine  # noqa: E402
from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes  # noqa: E402

B, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 32)
X, lab, cond = synthetic_batch(B, d, 1, seed=0)
unet = UnetEngine(in_channels=1, d=d, max_batch=B, lr=3e-6)
unet.set_weights(glorot_params(unet_param_shapes(1, 95), seed=1))
unet.upload_batch(X, lab)
vae = VaeEngine(unet, in_channels=1, d=d, max_batch=B, lr=5e-4)
vae.set_weights(glorot_params(vae_param_shapes(1, d=d), seed=3))
vae.upload_batch(X, cond, np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32))
for name, eng in (("DFC-VAE train step", vae), ("U-Net train step", unet)):
    for rep in range(3):
        e, g, n = eng.graph_probe(20)
        print("%-20s B=%d d=%d: eager %.3f ms/step, hipGraph replay %.3f ms/step (%+.1f %%), %d graph nodes"
              % (name, B, d, e, g, 100 * (g / e - 1), n))
m = vae.train_step_resident(True)
assert np.all(np.isfinite(m)), m
print("after the replays the engines still step: [Loss, PM, MSE, KLD] =", m)
