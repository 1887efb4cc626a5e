"""Randomised sweep of the generate tail's post-processing (segment_atoms on the device -> refine_atoms: convexity decisions,
marker-watershed splits on the library's host threads, recursion) against oracle/watershed_ref.py's
restatement of watershed_clustering (/root/reference/watershed.py:40-203; skimage restated: parity unpinned).  Volumes: unions
of random balls / ellipsoids that touch and overlap (non-convex components, splits, recursion levels), thin bridges, salt
noise; batches of 1..8 grids at 16^3 / 32^3 / 64^3.  Integer work: region volumes, species votes and centroids must be EQUAL.

    python tests/tools/fuzz_segment.py [trials=40] [seed=0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import watershed_ref as W          # noqa: E402


def volume(rng, d):
    zz, yy, xx = np.mgrid[:d, :d, :d]
    m = np.zeros((d, d, d), bool)
    n = int(rng.integers(2, 4 + d // 4))
    centres = rng.uniform(2, d - 2, size=(n, 3))
    for i in range(n):
        if i and rng.random() < 0.5:          # next to an earlier blob: touching / overlapping pairs and chains
            j = int(rng.integers(0, i))
            step = rng.normal(size=3)
            centres[i] = np.clip(centres[j] + step / np.linalg.norm(step) * rng.uniform(3, 8), 1, d - 2)
        r = rng.uniform(1.5, 2.5 + d / 10, size=3) if rng.random() < 0.4 else np.full(3, rng.uniform(1.5, 2.5 + d / 10))
        c = centres[i]
        m |= ((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 <= 1.0
    if rng.random() < 0.5:                    # a one-voxel bridge between two blobs
        a, b = centres[int(rng.integers(0, n))], centres[int(rng.integers(0, n))]
        for t in np.linspace(0, 1, 3 * d):
            p = np.round(a + t * (b - a)).astype(int)
            m[tuple(np.clip(p, 0, d - 1))] = True
    if rng.random() < 0.5:
        m |= rng.uniform(size=m.shape) < rng.choice([0.002, 0.01, 0.05])
    return m.astype(np.uint8)


def main():
    from icsg3d_amd.watershed import refine_atoms, segment_atoms
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    nbad = ngrid = nsplit = 0
    t0 = time.time()
    for t in range(trials):
        d = int(rng.choice([16, 32, 32, 32, 64]))
        B = int(rng.integers(1, 9 if d < 64 else 4))
        masks = np.stack([volume(rng, d) for _ in range(B)])
        species = np.where(masks != 0, rng.integers(1, 95, size=masks.shape), 0).astype(np.uint8)
        out = refine_atoms(segment_atoms(masks, species, max_atoms=1024), degenerate="solid")
        bad = []
        for b in range(B):
            a_ref, mu_ref, R_ref = W.watershed_clustering(None, species[b], masks[b].astype(np.int32), degenerate="solid")
            a, mu = out["atoms"][b]
            ok = (np.array_equal(out["regions"][b], R_ref.astype(np.int32)) and list(a) == list(a_ref)
                  and np.array_equal(np.array(mu).reshape(len(a), 3), np.array(mu_ref).reshape(len(a_ref), 3)))
            if not ok:
                bad.append(b)
        ngrid += B
        nsplit += int(np.sum(out["split"]))
        nbad += len(bad)
        print("  %s trial %d: d=%d B=%d atoms %s split %s%s" % ("FAIL" if bad else "ok  ", t, d, B, [len(out["atoms"][b][0]) for b in range(B)],
                                                              [int(v) for v in out["split"]], " grids %s differ" % bad if bad else ""),
              flush=True)
    print("fuzz_segment: %d grids in %d trials (%d took the non-convex path), %d differ from the oracle (%.0f s)"
          % (ngrid, trials, nsplit, nbad, time.time() - t0))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
