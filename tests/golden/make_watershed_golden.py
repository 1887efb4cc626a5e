"""Generates tests/golden/watershed_golden.npz in THIS container by running the REFERENCE's own `centroids` and
`majority_vote` (/root/reference/watershed.py:153-187).  The module cannot be imported (skimage and the plotting
stack at its top are absent), so the two FunctionDefs are pulled out of the file with `ast`, compiled and executed
with only numpy and itertools.product in scope -- it is the reference's code that runs, nothing of its text is written
to the repo.  The fixture is data: per case a binary mask and a species volume (inputs), the region matrix R that
`segment_nuclei` (watershed.py:52-56,85-92) produces when every component with more than 3 voxels takes its convex
branch -- built here with scipy.ndimage.label (6-connectivity, raster-order numbering, the same partition and order
as skimage's measure.label(connectivity=1)) -- and the (atoms, means) the reference returns for (species, R).

    python tests/golden/make_watershed_golden.py
"""
import ast
import os
from itertools import product

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/watershed.py"
D = 32          # centroids() hard-codes 32 (watershed.py:175-177)


def reference_functions():
    tree = ast.parse(open(REF).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("majority_vote", "centroids")]
    assert len(fns) == 2
    ns = {"np": np, "product": product}
    exec(compile(ast.Module(body=fns, type_ignores=[]), REF, "exec"), ns)
    return ns["centroids"]


def regions_of(mask, min_voxels=3):
    """R of segment_nuclei when every kept component is convex: components of > min_voxels voxels, renumbered 1..n in
    label order (watershed.py:52-56,85-92)."""
    lab, n = ndimage.label(mask != 0)                 # default structure = 6-connectivity
    sizes = np.bincount(lab.ravel(), minlength=n + 1)
    R = np.zeros(lab.shape, np.int32)
    k = 0
    for cl in range(1, n + 1):
        if sizes[cl] > min_voxels:
            k += 1
            R[lab == cl] = k
    return R, n, k


def cases():
    rng = np.random.default_rng(11)
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(D), np.arange(D), indexing="ij")
    out = {}
    # Gaussian blobs, species = id of the nearest centre with 15 % label noise and 10 % zeros inside the mask
    dens = np.zeros((D, D, D))
    near = np.zeros((D, D, D), np.int64)
    best = np.full((D, D, D), np.inf)
    ids = [8, 26, 57, 8, 94, 1, 38]
    for k, sp in enumerate(ids):
        c = rng.uniform(3, D - 3, 3)
        sig = rng.uniform(1.5, 3.5)
        r2 = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2
        dens += np.exp(-r2 / (2 * sig * sig))
        near = np.where(r2 < best, sp, near)
        best = np.minimum(best, r2)
    mask = (dens > 0.45).astype(np.uint8)
    species = near.copy()
    flip = rng.uniform(size=species.shape) < 0.15
    species[flip] = rng.integers(1, 95, size=int(flip.sum()))
    species[rng.uniform(size=species.shape) < 0.10] = 0
    out["blobs"] = (mask, species.astype(np.uint8))
    # salt noise: hundreds of tiny components, most of them at or below the size filter
    mask = (rng.uniform(size=(D, D, D)) < 0.22).astype(np.uint8)
    out["noise"] = (mask, rng.integers(0, 95, size=(D, D, D)).astype(np.uint8))
    # hand-made: tie in the vote (larger id wins), an all-zero-species region (skipped), sizes 3 / 4 around the filter,
    # diagonal neighbours (separate under 6-connectivity), regions on the volume border, a ring
    mask = np.zeros((D, D, D), np.uint8)
    species = np.zeros((D, D, D), np.uint8)
    mask[0:2, 0:2, 0:4] = 1; species[0:2, 0:2, 0:2] = 7; species[0:2, 0:2, 2:4] = 21          # 8 vs 8 voxels
    mask[5:8, 5:8, 5:8] = 1                                                                    # species all zero
    mask[10, 10, 10:13] = 1; species[10, 10, 10:13] = 3                                        # 3 voxels: dropped
    mask[12, 12, 10:14] = 1; species[12, 12, 10:14] = 4                                        # 4 voxels: kept
    mask[20:22, 20:22, 20:22] = 1; species[20:22, 20:22, 20:22] = 50
    mask[22:24, 22:24, 22:24] = 1; species[22:24, 22:24, 22:24] = 51                           # touches the former at a corner only
    mask[30:32, 30:32, 28:32] = 1; species[30:32, 30:32, 28:32] = 94; species[31, 31, 31] = 0  # far corner
    mask[26, 2:9, 2:9] = 1; mask[26, 4:7, 4:7] = 0; species[26, 2:9, 2:9] = 60; species[26, 2, 2:9] = 61   # ring
    mask[15:17, 0:32, 16] = 1; species[15, 0:32, 16] = 9; species[16, 0:32, 16] = 10           # 32 vs 32: tie again
    out["handmade"] = (mask, species)
    out["full"] = (np.ones((D, D, D), np.uint8), np.full((D, D, D), 13, np.uint8))
    out["empty"] = (np.zeros((D, D, D), np.uint8), rng.integers(0, 95, size=(D, D, D)).astype(np.uint8))
    return out


def main():
    centroids = reference_functions()
    arrays = {}
    for name, (mask, species) in cases().items():
        R, ncomp, nkept = regions_of(mask)
        atoms, means = centroids(species.astype(np.int64), R)
        arrays[name + "/mask"] = mask
        arrays[name + "/species"] = species
        arrays[name + "/R"] = R
        arrays[name + "/counts"] = np.array([ncomp, nkept], np.int32)
        arrays[name + "/atoms"] = np.array(atoms, np.int32)
        arrays[name + "/means"] = np.array(means, np.float64).reshape(len(atoms), 3)
        print("%-9s %5d components, %4d kept, %4d atoms" % (name, ncomp, nkept, len(atoms)))
    np.savez_compressed(os.path.join(HERE, "watershed_golden.npz"), **arrays)


if __name__ == "__main__":
    main()
