"""Per-layer conv micro-benchmark (HIP events): python scripts/conv_bench.py [B]"""
import ctypes as C, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icsg3d_amd import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def run(S, cin, cout, mode, abl=0, taps=27, iters=5):
    ms = C.c_float(0)
    _lib.check(lib.ics_op_conv3d_bench(B, S, cin, cout, taps, mode, abl, iters, C.byref(ms)))
    fl = 2.0 * B * S ** 3 * taps * cin * cout
    return ms.value, fl / (ms.value * 1e-3) / 1e12
layers = [("c18", 32, 128, 128), ("c17", 32, 192, 128), ("c15", 16, 384, 256), ("c13", 8, 768, 512), ("c2", 32, 32, 64), ("c10", 4, 512, 512)]
for name, S, ci, co in layers:
    r = ["%s" % name]
    for mode, tag in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
        ms, tf = run(S, ci, co, mode)
        r.append("%s %.3f ms %.1f TF" % (tag, ms, tf))
    print("  ".join(r))
for abl in (0, 1, 2):
    ms, tf = run(32, 128, 128, 2, abl)
    print("c18 wgrad ablate=%d: %.3f ms  %.1f TF/s" % (abl, ms, tf))
