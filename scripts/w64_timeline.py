"""Where a conv_wino64 workgroup spends its time (variant build with -DICS_W64_TIMELINE, see scripts/variants.sh):
ICSG3D_LIB_PATH=icsg3d_amd/variants/libicsg3d_hip_tl.so python scripts/w64_timeline.py [S cin cout mode feat]
Stamps are the 100 MHz wall clock: prologue / main loop / epilogue per workgroup, and the gap between consecutive
workgroups on the same CU."""
import ctypes as C, os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icsg3d_amd import _lib
lib = _lib.load()
S, cin, cout, mode, feat = [int(a) for a in (sys.argv[1:6] + ["32", "128", "128", "0", "24"][len(sys.argv) - 1:])]
B = 32
ms = C.c_float(0)
_lib.check(lib.ics_op_conv3d_bench(B, S, cin, cout, 27, mode, feat if mode == 0 else 0, 3, C.byref(ms)))
nwg = min(32768, B * (S // 4) * (S // 4) * (S // 8) * (cout // 64))
buf = np.zeros(nwg * 16, np.uint64)
lib.ics_debug_w64_timeline.argtypes = [C.c_void_p, C.c_int]
assert lib.ics_debug_w64_timeline(buf.ctypes.data, nwg * 16) == 0
r = buf.reshape(nwg, 16).astype(np.int64)
t0, t1, t2, t3 = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
tick = 10.0   # ns
print("S=%d %d->%d mode %d feat %d: %.3f ms, %d workgroups" % (S, cin, cout, mode, feat, ms.value, nwg))
e = [t2] + [r[:, i] for i in range(4, 10)] + [t3]
names = ["wait other waves", "pass0 transform+write", "pass0 read+store", "pass1 barrier", "pass1 transform+write", "pass1 read+store", "stats/exit"]
rows = [("prologue", t1 - t0), ("main loop", t2 - t1), ("epilogue", t3 - t2)] + [("  " + n, e[i + 1] - e[i]) for i, n in enumerate(names)] + [("total", t3 - t0)]
for name, d in rows:
    d = d * tick / 1e3
    print("  %-24s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f" % (name, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
cu = (r[:, 15] & 0xf) * 65536 + ((r[:, 14] >> 8) & 0xff)
gaps, per = [], collections.Counter()
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    o = idx[np.argsort(t0[idx])]
    per[len(o)] += 1
    gaps.extend(((t0[o][1:] - t3[o][:-1]) * tick / 1e3).tolist())
gaps = np.array(gaps)
print("  CUs seen %d; workgroups per CU %s" % (len(np.unique(cu)), dict(per)))
print("  gap between workgroups on a CU: mean %.2f us  p10 %.2f  p50 %.2f  p90 %.2f" % (gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 50), np.percentile(gaps, 90)))
span = (t3.max() - t0.min()) * tick / 1e3
print("  first entry -> last exit %.1f us; sum(total)/CUs = %.1f us" % (span, (t3 - t0).sum() * tick / 1e3 / len(np.unique(cu))))
