"""GPU box: what the 64x64-tile plain GEMM path delivers on the shapes a Winograd-domain evaluation of the S = 4 layers
would launch: 64 frequencies x [tiles = 8 B] x [Cin] x [Cout] == one [64 * 8 B][Cin] x [Cin][Cout] GEMM in block count
and per-block work (python scripts/gemm_probe.py)."""
import ctypes as C, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icsg3d_amd import _lib
lib = _lib.load()
def run(B, S, cin, cout, mode, taps, iters=10):
    ms = C.c_float(0)
    _lib.check(lib.ics_op_conv3d_bench(B, S, cin, cout, taps, mode, 0, iters, C.byref(ms)))
    return ms.value
for name, cin, cout in (("c9", 256, 512), ("c10", 512, 512)):
    for mode, tag in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
        d = run(32, 4, cin, cout, mode, 27)
        # 64 frequency GEMMs over 256 tiles = rows 64 * 256 = B' * 4^3 with B' = 256
        g = run(256, 4, cin, cout, mode, 1)
        fl = 2.0 * 64 * 256 * cin * cout
        print("%s %s: direct 27-tap %.3f ms | 64-frequency GEMM equivalent %.3f ms (%.1f TF/s executed)" % (name, tag, d, g, fl / (g * 1e-3) / 1e12))
