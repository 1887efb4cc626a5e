"""Where a conv_wino_wgrad_kernel workgroup spends its time (the backward-weight kernel had never had the
per-workgroup timeline the forward kernel got).  Variant build with -DICS_WG_TIMELINE:

  scripts/variants.sh conv_wino "wgtl:-DICS_WG_TIMELINE"
  ICSG3D_LIB_PATH=icsg3d_amd/variants/libicsg3d_hip_wgtl.so python scripts/wgrad_timeline.py > profiles/r6_wgrad_timeline.txt

Per layer shape of the U-Net step (B = 32): launch time (HIP events, kernel + split reduction), and per workgroup from the
100 MHz wall clock: prologue (entry -> first block staged, first operands built), main loop (per wave: the spread between the
first and the last wave to leave it = wave skew), epilogue (G^T contraction through LDS + the split write-out), the
main loop's time per tile block against the block's pure MFMA time, and how the workgroups of the launch line up on the chip
(first entry -> last exit against the mean workgroup time: tail / quantisation)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icsg3d_amd import _lib  # noqa: E402

lib = _lib.load()
lib.ics_debug_wgrad_timeline.argtypes = [C.c_void_p, C.c_int]
B = 32
TICK = 10.0          # ns per wall_clock64 tick (100 MHz)
MFMA_NS_PER_BLOCK = 8 * 8 * 2 * 64 / 2.4   # 8 k-steps x 8 MFMAs (32x32x2: 64 cycles) x 2 waves per SIMD at 2.4 GHz = 3413 ns
SHAPES = [("c18", 32, 128, 128), ("c17.skip", 32, 64, 128), ("c2", 32, 32, 64), ("c16", 16, 256, 128), ("c15.skip", 16, 128, 256),
          ("c4", 16, 64, 128), ("c3", 16, 64, 64), ("c14", 8, 512, 256), ("c13.skip", 8, 256, 512), ("c6", 8, 128, 256),
          ("c5", 8, 128, 128)]


def one(name, S, cin, cout, feat):
    ms = C.c_float(0)
    _lib.check(lib.ics_op_conv3d_bench(B, S, cin, cout, 27, 2, feat, 5, C.byref(ms)))
    nblocks = B * (S // 4) * (S // 4) * (S // 8)
    buf = np.zeros(8192 * 16, np.uint64)
    assert lib.ics_debug_wgrad_timeline(buf.ctypes.data, buf.size) == 0
    r = buf.reshape(8192, 16).astype(np.int64)
    per_split = int(r[0, 11])
    nwg = nblocks // per_split * (cin // 32) * (cout // 32)
    r = r[:nwg]
    t0, t1, t_end = r[:, 0], r[:, 1], r[:, 10]
    w_done = r[:, 2:10]
    first, last = w_done.min(1), w_done.max(1)
    us = lambda d: d * TICK / 1e3   # noqa: E731
    total = us(t_end - t0)
    pro, loop_first, skew, epi = us(t1 - t0), us(first - t1), us(last - first), us(t_end - last)
    per_block = us(last - t1) * 1e3 / per_split        # ns per tile block, to the LAST wave
    span = us(t_end.max() - t0.min())
    flops = 2.0 * B * S ** 3 * 27 * cin * cout * 64.0 / 216.0
    cu = (r[:, 15] & 0xf) * 65536 + ((r[:, 14] >> 8) & 0xff) + ((r[:, 14] >> 13) & 7) * 4096 + ((r[:, 14] >> 12) & 1) * 256
    ncu = len(np.unique(cu))
    print("%-9s S=%2d %3d->%3d %s: launch %.3f ms (kernel + split reduction; executed %.0f TFLOP/s over the launch), %d workgroups "
          "x %d tile blocks on %d CUs; first entry -> last exit %.1f us"
          % (name, S, cin, cout, "affine source" if feat else "plain source ", ms.value, flops / (ms.value * 1e-3) / 1e12, nwg,
             per_split, ncu, span))
    for label, d in (("prologue", pro), ("main loop (first wave out)", loop_first), ("wave skew (first -> last wave)", skew),
                     ("epilogue + split write-out", epi), ("workgroup total", total)):
        print("    %-32s mean %8.2f us  p10 %8.2f  p50 %8.2f  p90 %8.2f   (%4.1f %% of the workgroup)"
              % (label, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), 100 * d.mean() / total.mean()))
    out = 100.0 * (pro.mean() + skew.mean() + epi.mean()) / total.mean()
    print("    per tile block %.0f ns against %.0f ns of MFMA time = %.3f of the matrix pipe inside the main loop; outside the "
          "main loop %.1f %% of a workgroup; chip-level: sum(workgroup) / CUs = %.1f us of the %.1f us span (%.1f %% idle tail / "
          "start-up)" % (per_block.mean(), MFMA_NS_PER_BLOCK, MFMA_NS_PER_BLOCK / per_block.mean(), out,
                         total.sum() / max(ncu, 1), span, 100 * (1 - total.sum() / max(ncu, 1) / span)))
    return ms.value


if __name__ == "__main__":
    tot = 0.0
    for name, S, cin, cout in SHAPES:
        tot += one(name, S, cin, cout, 8)         # the engine's launches read a BatchNorm-affine source
    print("sum over the %d shapes: %.3f ms per U-Net step" % (len(SHAPES), tot))
    one("c18", 32, 128, 128, 0)
