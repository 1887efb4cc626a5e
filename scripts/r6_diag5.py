"""profile of the pinned fp64 oracle step at 32 threads on the GPU box (B = 8, d = 32)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import numpy_ref as R, torch_ref as T
torch.set_num_threads(32)
B, d = 8, 32
X, lab, _ = R.synthetic_batch(B, d, 1, seed=0, dtype=np.float64)
sh = R.unet_param_shapes(1, 95)
P, S = R.init_params(sh, 1), R.init_bn_state(sh)
p32 = T.Params(P, S, torch.float32, requires_grad=False)
taps = {}
with torch.no_grad():
    T.unet_trunk(T.to_t(X, torch.float32), p32, True, "tf_cpu", taps=taps)
kink = {n: T.to_n(t) for n, t in taps.items()}
t0 = time.time(); T.unet_step_grads(P, S, X, lab, kink=kink, kink_tol=1e-3, want_outputs=False); print("pinned step %.1f s" % (time.time() - t0))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=False, with_stack=False) as prof:
    T.unet_step_grads(P, S, X, lab, kink=kink, kink_tol=1e-3, want_outputs=False)
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=50))
