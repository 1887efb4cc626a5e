"""A/B of the A-tile LDS row stride (make EXTRA=-DICS_KLDA=n): forward / backward-data timings of the layer shapes
whose 32-row MFMA tiles span several x-lines (S <= 16), plus c18 as the S = 32 control."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icsg3d_amd import _lib
lib = _lib.load()
B = 32
def run(S, cin, cout, mode, iters=6):
    ms = C.c_float(0)
    _lib.check(lib.ics_op_conv3d_bench(B, S, cin, cout, 27, mode, 0, iters, C.byref(ms)))
    return ms.value, 2.0 * B * S ** 3 * 27 * cin * cout / (ms.value * 1e-3) / 1e12
for name, S, ci, co in [("c18", 32, 128, 128), ("c16", 16, 256, 128), ("c15s", 16, 128, 256), ("c14", 8, 512, 256),
                        ("c13s", 8, 256, 512), ("c6", 8, 128, 256), ("c10", 4, 512, 512), ("c4", 16, 64, 128)]:
    print(name, "  ".join("%s %.3f ms %.1f TF" % ((tag,) + run(S, ci, co, mode)) for mode, tag in ((0, "fwd"), (1, "dgrad"))))
