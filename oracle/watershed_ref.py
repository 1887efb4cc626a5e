"""TEST INFRASTRUCTURE (never imported by the product path): CPU restatement of the integer part of
/root/reference/watershed.py that csrc/segment.hip computes on the device.

Pinned: `centroids` / `majority_vote` below are checked against tests/golden/watershed_golden.npz, whose expected
outputs were produced by the reference's own two functions (tests/golden/make_watershed_golden.py).  `regions`
restates segment_nuclei's labelling with scipy.ndimage.label in place of skimage.measure.label (absent here): same
6-connectivity, same raster-order numbering."""
import numpy as np
from scipy import ndimage


def regions(mask, min_voxels=3):
    """watershed.py:52-56 + the convex branch :85-92: components of the binary mask under 6-connectivity
    (`measure.label(binary, connectivity=1)`), those with `count > 3` kept and renumbered 1..n in label order.
    Returns (R int32, n_components, n_kept)."""
    lab, n = ndimage.label(np.asarray(mask) != 0)
    sizes = np.bincount(lab.ravel(), minlength=n + 1)
    keep = np.zeros(n + 1, np.int32)
    kept = [cl for cl in range(1, n + 1) if sizes[cl] > min_voxels]
    keep[kept] = np.arange(1, len(kept) + 1)
    return keep[lab].astype(np.int32), n, len(kept)


def majority_vote(seg_img, R, cl):
    """watershed.py:153-163: most frequent non-zero species inside region cl; the reference sorts (value, count) pairs
    by count with a stable sort over ascending values and takes the last: equal counts -> the larger value."""
    vals = seg_img[R == cl]
    vals = vals[vals != 0]
    if vals.size == 0:
        return 0
    u, c = np.unique(vals, return_counts=True)
    return int(u[np.nonzero(c == c.max())[0][-1]])


def centroids(seg_img, R):
    """watershed.py:165-187.  classes = np.unique(R)[1:] drops the smallest value whatever it is: a volume without a
    single background voxel loses its first region (kept as is).  Mean over ALL voxels of a region, in index units."""
    classes = np.unique(R)[1:]
    atoms, means = [], []
    for cl in classes:
        sp = majority_vote(seg_img, R, cl)
        if sp != 0:
            idx = np.argwhere(R == cl)
            means.append(idx.astype(np.float64).mean(axis=0))
            atoms.append(sp)
    return atoms, means


def watershed_clustering_convex(species, mask, min_voxels=3):
    R, ncomp, nkept = regions(mask, min_voxels)
    atoms, means = centroids(np.asarray(species).astype(np.int64), R)
    return atoms, means, R, ncomp, nkept
