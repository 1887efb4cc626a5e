"""Developer probe (round 5): the DFC-VAE step after a U-Net training phase in the same process
(bench.py's contract order) ran 10 % slower than in a fresh process under the two-stream schedule.  Modes a..g bisect it
(profiling first / plain steps first / separate perceptual engine / metrics read-back / predict first / tap buffers
allocated early / serial schedule); the cause was two of the library's streams sharing one of the runtime's four default
hardware queues (GPU_MAX_HW_QUEUES=8 python scripts/vae_stream_bisect.py b  is fast).  usage: vae_stream_bisect.py <mode>"""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icsg3d_amd.engine import UnetEngine, VaeEngine
from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
B, d = 32, 32
X, lab, cond = synthetic_batch(B, d, 1, seed=0)
eps = np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32)
PU = glorot_params(unet_param_shapes(1, 95), 1)
def vae_time(ue, tag, n=30):
    ve = VaeEngine(ue, d=d, max_batch=B, lr=5e-4); ve.set_weights(glorot_params(vae_param_shapes(1, d=d), 3))
    ve.upload_batch(X, cond, eps)
    for _ in range(5): ve.train_step_resident(False)
    ve.sync(); t0 = time.perf_counter()
    for _ in range(n): ve.train_step_resident(False)
    ve.sync(); dt = (time.perf_counter() - t0) / n
    print("%-60s %.3f ms/step" % (tag, dt * 1e3), flush=True)
    return ve
mode = sys.argv[1]
ue = UnetEngine(d=d, max_batch=B, lr=3e-6); ue.set_weights(PU); ue.upload_batch(X, lab)
if mode == "a":      # all-events profiling of U-Net steps first (bench.py's untimed pass)
    ue.profile_filter(""); ue.profile_enable(True)
    for _ in range(2): ue.train_step_resident(False)
    ue.sync(); ue.profile_rows(); ue.profile_enable(False)
    vae_time(ue, "after 2 all-events U-Net steps")
elif mode == "b":    # many plain U-Net steps (heat)
    for _ in range(60): ue.train_step_resident(False)
    ue.sync()
    vae_time(ue, "after 60 plain U-Net steps")
    time.sleep(3)
    vae_time(ue, "... and 3 s idle later (second VAE engine)")
elif mode == "c":    # separate perceptual engine after the same U-Net phase
    for _ in range(60): ue.train_step_resident(False)
    ue.sync()
    pm = UnetEngine(d=d, max_batch=B); pm.set_weights(PU)
    vae_time(pm, "separate perceptual engine after 60 U-Net steps")
elif mode == "d":    # metrics read-back once (bench does it)
    ue.train_step_resident(True)
    vae_time(ue, "after one U-Net step with metrics")
elif mode == "e":    # forward only before the VAE exists (allocates the pack table, nothing else)
    ue.predict(X[:2])
    vae_time(ue, "after one predict")
elif mode == "f":    # tap buffers allocated first (a VAE engine created and closed), then U-Net training, then the VAE
    v0 = VaeEngine(ue, d=d, max_batch=B); v0.close()
    for _ in range(10): ue.train_step_resident(False)
    ue.sync()
    vae_time(ue, "tap buffers allocated before the U-Net trained")
elif mode == "g":    # the slow order, serial schedule
    import os
    for _ in range(10): ue.train_step_resident(False)
    ue.sync()
    os.environ["ICSG3D_NO_PM_SIDE"] = "1"; os.environ["ICSG3D_NO_VAE_SIDE_WGRAD"] = "1"
    vae_time(ue, "slow order, serial schedule")
