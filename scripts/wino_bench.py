"""Winograd kernel A/B on the bench shapes (HIP events, ics_op_conv3d_bench takes the engine's path for the shape):
python scripts/wino_bench.py   -- run once plain, once with ICSG3D_NO_WINO64=1 to compare the two kernel shapes."""
import ctypes as C, os, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icsg3d_amd import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
FEAT = int(os.environ.get("WINO_FEAT", "0"))   # 8: BatchNorm-affine source, 16: BatchNorm statistics, 32: bias (forward only)
MODES = [int(m) for m in os.environ.get("WINO_MODES", "0,1,2").split(",")]
def run(S, cin, cout, mode, iters=10):
    ms = C.c_float(0)
    _lib.check(lib.ics_op_conv3d_bench(B, S, cin, cout, 27, mode, FEAT if mode == 0 else 0, iters, C.byref(ms)))
    fl = 2.0 * B * S ** 3 * 27 * cin * cout
    return ms.value, fl / (ms.value * 1e-3) / 1e12
layers = [("c18", 32, 128, 128), ("c17s", 32, 64, 128), ("c2", 32, 32, 64), ("c16", 16, 256, 128), ("c15s", 16, 128, 256),
          ("c4", 16, 64, 128), ("c14", 8, 512, 256), ("c13s", 8, 256, 512), ("c6", 8, 128, 256)]
tag = ("NO_WINO64" if os.environ.get("ICSG3D_NO_WINO64") else "wino64") + (" feat=%d" % FEAT if FEAT else "")
tot = [0.0, 0.0, 0.0]
for name, S, ci, co in layers:
    r = ["%-5s" % name]
    for mode, t in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
        if mode not in MODES:
            continue
        ms, tf = run(S, ci, co, mode)
        tot[mode] += ms
        r.append("%s %.3f ms %5.1f TF (exec %.1f)" % (t, ms, tf, tf * 64 / 216))
    print(tag, "  ".join(r))
print(tag, "sum fwd %.3f dgrad %.3f wgrad %.3f ms" % tuple(tot))
