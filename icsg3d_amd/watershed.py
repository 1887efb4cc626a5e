"""Device-side counterpart of the integer part of /root/reference/watershed.py (`watershed_clustering`, :190-203):
6-connected component labelling of the binary mask with the reference's size filter (`segment_nuclei` step 1,
:52-56), the region matrix R for components that take the convex branch (:85-92), and `centroids` +
`majority_vote` (:153-187).  Compute is libicsg3d_hip.so (csrc/segment.hip); there is no CPU fallback.

NOT implemented (skimage is absent from the image, so neither can be pinned): the convex-hull test
(`morphology.convex_hull_image`, :80) and the marker watershed (`segmentation.watershed`, :96-150) that the
reference applies to components whose convexity is below 0.8.  Every kept component is treated as convex; the
per-region voxel counts and bounding boxes returned here are the inputs a host-side implementation of those two
steps needs."""
from __future__ import annotations

import numpy as np

from . import _lib as L

STAT_FIELDS = ("species", "voxels", "sum0", "sum1", "sum2", "lo0", "lo1", "lo2", "hi0", "hi1", "hi2")


def _atoms_from_stats(counts, stats, voxels_per_sample):
    """[(atoms, means)] per sample, exactly what `centroids(seg_img, R)` returns (watershed.py:165-187): regions in
    ascending label order, skipped when the majority vote is 0, mean = coordinate mean over ALL voxels of the region.
    The reference takes `np.unique(R)[1:]` as the region list, i.e. drops the smallest value present whatever it is:
    a volume WITHOUT a single background voxel loses its first region.  Kept."""
    out = []
    for b in range(stats.shape[0]):
        st = stats[b, :counts[b, 1]]
        if len(st) and int(st[:, 1].sum()) == voxels_per_sample:
            st = st[1:]
        keep = st[:, 0] != 0
        atoms = [int(v) for v in st[keep, 0]]
        means = [st[i, 2:5].astype(np.float64) / np.float64(st[i, 1]) for i in np.nonzero(keep)[0]]
        out.append((atoms, means))
    return out


def segment_atoms(mask, species, min_voxels=3, max_atoms=512, num_species=95, want_regions=True):
    """mask / species: (B,d,d,d) arrays (non-zero mask = foreground; species = class ids < num_species).
    Returns dict(regions int32 (B,d,d,d) | None, n_components (B,), n_atoms (B,), stats int32 (B,max_atoms,11),
    atoms [(species list, mean list)] per sample)."""
    mask = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
    species = np.ascontiguousarray(species, dtype=np.uint8)
    if mask.ndim != 4 or mask.shape != species.shape or len(set(mask.shape[1:])) != 1:
        raise ValueError("mask and species must both be (B,d,d,d), got %s / %s" % (mask.shape, species.shape))
    B, d = mask.shape[0], mask.shape[1]
    regions = np.empty(mask.shape, np.int32) if want_regions else None
    counts = np.zeros((B, 2), np.int32)
    stats = np.zeros((B, max_atoms, len(STAT_FIELDS)), np.int32)
    L.check(L.load().ics_op_segment_atoms(L.u8ptr(mask), L.u8ptr(species), B, d, int(min_voxels), int(max_atoms),
                                          int(num_species), L.i32ptr(regions), L.i32ptr(counts), L.i32ptr(stats)))
    return {"regions": regions, "n_components": counts[:, 0].copy(), "n_atoms": counts[:, 1].copy(), "stats": stats,
            "atoms": _atoms_from_stats(counts, stats, d ** 3)}


def watershed_clustering(M, S, Sb, max_iters=5, return_ws=False, verbose=False):
    """Signature of the reference's entry point (watershed.py:190): (atoms, means[, R]) for ONE sample, computed on
    the device.  `M` (density) and `max_iters` only feed the marker watershed, which is not implemented (see module
    docstring): components are never split."""
    S = np.asarray(S).squeeze()
    Sb = np.asarray(Sb).squeeze()
    r = segment_atoms(Sb[None], S[None], want_regions=return_ws)
    atoms, means = r["atoms"][0]
    if return_ws:
        return np.array(atoms), np.array(means), r["regions"][0].astype(np.float64)
    return np.array(atoms), np.array(means)
