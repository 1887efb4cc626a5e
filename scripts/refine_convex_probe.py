"""Developer probe: segment_atoms + refine_atoms on 32 grids of nine ball-shaped atoms each (every component convex: what a
trained segmentation produces), with and without the device-side convexity bounds.  python scripts/refine_convex_probe.py"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from icsg3d_amd.watershed import refine_atoms, segment_atoms
rng = np.random.default_rng(0)
B, d = 32, 32
zz, yy, xx = np.mgrid[:d, :d, :d]
masks = np.zeros((B, d, d, d), np.uint8)
for b in range(B):
    for k in range(9):
        c = np.array([5 + 11 * (k // 9 % 3), 5 + 11 * (k // 3 % 3), 5 + 11 * (k % 3)]) + rng.uniform(-1, 1, 3)
        r = rng.uniform(2.2, 3.6)
        masks[b][((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= r * r] = 1
species = np.where(masks != 0, rng.integers(1, 95, size=masks.shape), 0).astype(np.uint8)
segment_atoms(masks[:2], species[:2])
for use in (True, False, True, False):
    t0 = time.perf_counter(); out = segment_atoms(masks, species, max_atoms=64); t1 = time.perf_counter()
    if not use: out["bounds"] = None
    refine_atoms(out); t2 = time.perf_counter()
    print("bounds %-5s segment_atoms %.1f ms  refine %.1f ms  atoms/grid %.1f split %d failed %d" % (use, (t1 - t0) * 1e3, (t2 - t1) * 1e3, out["n_atoms"].mean(), out["split"].sum(), out["failed"].sum()))
