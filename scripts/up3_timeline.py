"""Where a conv_up3 workgroup spends its time (variant build -DICS_UP3_TIMELINE): prologue / main loop / epilogue.
ICSG3D_LIB_PATH=icsg3d_amd/variants/libicsg3d_hip_tl3.so python scripts/up3_timeline.py"""
import ctypes as C, os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icsg3d_amd.engine import UnetEngine
from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes
from icsg3d_amd import _lib
lib = _lib.load()
B, d = 32, 32
eng = UnetEngine(d=d, max_batch=B); eng.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
X, lab, _ = synthetic_batch(B, d, 1, seed=0)
eng.upload_batch(X, lab); eng.train_step_resident(False); eng.sync()
# the last conv_up3 launch of a step is c17.up (16 384 workgroups)
nwg = 16384
buf = np.zeros(nwg * 16, np.uint64)
lib.ics_debug_up3_timeline.argtypes = [C.c_void_p, C.c_int]
assert lib.ics_debug_up3_timeline(buf.ctypes.data, nwg * 16) == 0
r = buf.reshape(nwg, 16).astype(np.int64)
t0, t1, t2, t3 = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
for name, dd in (("prologue", t1 - t0), ("main loop", t2 - t1), ("epilogue", t3 - t2), ("total", t3 - t0)):
    dd = dd * 10.0 / 1e3
    print("  %-10s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f" % (name, dd.mean(), np.percentile(dd, 10), np.percentile(dd, 50), np.percentile(dd, 90)))
cu = (r[:, 15] & 0xf) * 65536 + ((r[:, 14] >> 8) & 0xff)
gaps = []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]; o = idx[np.argsort(t0[idx])]
    gaps.extend(((t0[o][1:] - t3[o][:-1]) * 10.0 / 1e3).tolist())
gaps = np.array(gaps)
print("  CUs %d; gap between workgroups on a CU: mean %.2f us p50 %.2f" % (len(np.unique(cu)), gaps.mean(), np.percentile(gaps, 50)))
print("  first entry -> last exit %.1f us" % ((t3.max() - t0.min()) * 10.0 / 1e3))
