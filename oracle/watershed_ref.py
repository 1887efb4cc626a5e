"""TEST INFRASTRUCTURE (never imported by the product path): CPU restatement of /root/reference/watershed.py
(`watershed_clustering`, :190-203 and everything under it) that csrc/segment.hip + icsg3d_amd/watershed.py compute.

Pinned: `centroids` / `majority_vote` below are checked against tests/golden/watershed_golden.npz, whose expected
outputs were produced by the reference's own two functions (tests/golden/make_watershed_golden.py).  `regions`
restates segment_nuclei's labelling with scipy.ndimage.label in place of skimage.measure.label (absent here): same
6-connectivity, same raster-order numbering.

**PARITY UNPINNED: skimage absent.**  `segment_nuclei` (watershed.py:40-150) calls four scikit-image 0.17.2 routines
(requirements.txt:95) -- measure.label, morphology.erosion / dilation with ball(1), morphology.convex_hull_image and
segmentation.watershed -- and scikit-image is not installed in this image and not installable (no network).  They are
restated below from the PUBLISHED algorithms of that release, recalled, not executed:
  * `label_equal`: components of EQUAL value (skimage labels an integer image, not a mask), numbered in raster order of
    their first voxel; connectivity 1 = 6 neighbours, connectivity 3 (the default `None`) = 26.
  * `erode_ball1` / `dilate_ball1`: skimage 0.17.2 forwards to scipy.ndimage.grey_erosion / grey_dilation with the
    footprint ball(1) (the 7-voxel cross) and scipy's default border mode 'reflect': an out-of-volume neighbour of a
    border voxel is the border voxel itself, i.e. it is ignored.
  * `convex_hull_image`: scipy.spatial.ConvexHull (Qhull, as skimage uses) over the voxel coordinates offset by
    +-0.5 along each axis ("offset_coordinates"), a grid point is inside iff every hull inequality is < 1e-10.
  * `watershed_flood`: skimage's priority flood: a binary heap ordered by (image value, age); every marker voxel enters
    with age 0 in raster order, every flooded voxel with the next value of one global counter; a popped voxel labels
    and pushes its unlabelled 6-neighbours in the order -z, -y, -x, +x, +y, +z.  Entries with equal (value, age) --
    only the age-0 markers -- leave the heap in the order skimage's array heap (heap_general.pxi: strict `smaller`
    in sift-up and sift-down, left child preferred) produces, which the restatement emulates operation by operation;
    `tie="fifo"` is the stable alternative (kept to measure how much the tie rule matters: see the tests).
What IS pinned independently of skimage: every step is checked against scipy.ndimage (labelling, grey morphology) and
against brute-force definitions in tests/test_oracle_watershed.py; the device path is held to THIS file bit for bit."""
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


# ======================================================================================================================
# segment_nuclei (watershed.py:40-150) -- PARITY UNPINNED: skimage absent (see the module docstring)
# ======================================================================================================================
CROSS = ndimage.generate_binary_structure(3, 1)          # morphology.ball(1): the 7-voxel cross
FULL = ndimage.generate_binary_structure(3, 3)           # 26-connectivity


def label_equal(vol, connectivity=1):
    """skimage.measure.label(vol, connectivity=...) for an integer volume: maximal connected sets of voxels with the
    SAME non-zero value, numbered 1.. in raster order of their first voxel.  Returns (labels int32, n)."""
    vol = np.asarray(vol)
    st = CROSS if connectivity == 1 else FULL
    out = np.zeros(vol.shape, np.int32)
    firsts = []                                           # (first raster index, value, per-value label)
    per_value = {}
    for v in np.unique(vol):
        if v == 0:
            continue
        lab, n = ndimage.label(vol == v, structure=st)
        per_value[v] = lab
        flat = lab.ravel()
        first = np.full(n + 1, flat.size, np.int64)
        np.minimum.at(first, flat, np.arange(flat.size))
        firsts += [(int(first[k]), v, k) for k in range(1, n + 1)]
    firsts.sort()
    for new, (_, v, k) in enumerate(firsts, start=1):
        out[per_value[v] == k] = new
    return out, len(firsts)


def erode_ball1(a):
    """morphology.erosion(a, ball(1)) of skimage 0.17.2 = scipy.ndimage.grey_erosion(a, footprint=ball(1)), border mode
    'reflect' (watershed.py:30-35)."""
    return ndimage.grey_erosion(np.asarray(a), footprint=CROSS)


def dilate_ball1(a):
    """morphology.dilation(a, ball(1)) = scipy.ndimage.grey_dilation with the (symmetric) footprint (watershed.py:26-28)."""
    return ndimage.grey_dilation(np.asarray(a), footprint=CROSS)


def convex_hull_image(img, tolerance=1e-10, degenerate="raise"):
    """skimage.morphology.convex_hull_image(img) for a 3-D volume (offset_coordinates=True, tolerance=1e-10).
    scikit-image 0.17.2 reduces the coordinates to their hull vertices with an unguarded ConvexHull(coords): coplanar /
    collinear inputs raise QhullError there, the error leaves watershed_clustering and generate.py:246-248 skips the
    sample.  degenerate="raise" (default) reproduces that (the QhullError propagates); "solid" skips the pre-reduction
    instead (the offset points are always full-dimensional) -- lenient, not the reference."""
    from scipy.spatial import ConvexHull, QhullError
    img = np.asarray(img)
    if np.count_nonzero(img) == 0:
        return np.zeros(img.shape, bool)
    coords = np.transpose(np.nonzero(img)).astype(np.float64)
    try:
        h0 = ConvexHull(coords)
        coords = h0.points[h0.vertices]
    except (QhullError, ValueError):
        if degenerate != "solid":
            raise
    offsets = np.zeros((6, 3))
    for k, (axis, off) in enumerate((a, o) for a in range(3) for o in (-0.5, 0.5)):
        offsets[k, axis] = off
    pts = np.unique((coords[:, None, :] + offsets).reshape(-1, 3), axis=0)
    hull = ConvexHull(pts)
    grid = np.reshape(np.mgrid[tuple(map(slice, img.shape))], (3, -1)).astype(np.float64)
    inside = np.ones(grid.shape[1], bool)
    for eq in hull.equations:
        inside &= (eq[:3] @ grid + eq[3]) < tolerance
    return inside.reshape(img.shape)


class _Heap:
    """skimage/_shared/heap_general.pxi: array heap of (value, age, index); `smaller` compares value, then age, and is
    strict -- equal keys never swap in sift-up, the left child wins equal-key comparisons in sift-down."""
    __slots__ = ("a",)

    def __init__(self):
        self.a = []

    @staticmethod
    def _smaller(x, y):
        return x[0] < y[0] if x[0] != y[0] else x[1] < y[1]

    def push(self, item):
        a = self.a
        a.append(item)
        child = len(a) - 1
        while child > 0:
            parent = (child + 1) // 2 - 1
            if self._smaller(a[child], a[parent]):
                a[child], a[parent] = a[parent], a[child]
                child = parent
            else:
                break

    def pop(self):
        a = self.a
        top = a[0]
        last = a.pop()
        n = len(a)
        if n == 0:
            return top
        a[0] = last
        i = 0
        while True:
            l, r = 2 * i + 1, 2 * i + 2
            smallest = i
            if l < n:
                if self._smaller(a[l], a[i]):
                    smallest = l
                if r < n and self._smaller(a[r], a[smallest]):
                    smallest = r
            else:
                break
            if smallest == i:
                break
            a[i], a[smallest] = a[smallest], a[i]
            i = smallest
        return top

    def __len__(self):
        return len(self.a)


def watershed_flood(image, markers, tie="heap"):
    """skimage.segmentation.watershed(image, markers) with the defaults (connectivity 1, no mask, compactness 0, no
    watershed line): returns the label volume; voxels no marker reaches stay 0."""
    image = np.asarray(image, np.float64)
    out = np.array(markers, np.int32)
    D, H, W = out.shape
    flat_img, flat = image.ravel(), out.ravel()
    nbr = ((-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0))
    heap = _Heap()
    if tie == "fifo":
        import heapq
        q, seq = [], 0
        for idx in np.flatnonzero(flat):
            heapq.heappush(q, (flat_img[idx], 0, seq, int(idx))); seq += 1
        age = 1
        while q:
            _, _, _, idx = heapq.heappop(q)
            z, y, x = idx // (H * W), (idx // W) % H, idx % W
            for dz, dy, dx in nbr:
                zz, yy, xx = z + dz, y + dy, x + dx
                if not (0 <= zz < D and 0 <= yy < H and 0 <= xx < W):
                    continue
                n = (zz * H + yy) * W + xx
                if flat[n]:
                    continue
                age += 1
                flat[n] = flat[idx]
                heapq.heappush(q, (flat_img[n], age, seq, n)); seq += 1
        return out
    for idx in np.flatnonzero(flat):
        heap.push((flat_img[idx], 0, int(idx)))
    age = 1
    while len(heap):
        _, _, idx = heap.pop()
        z, y, x = idx // (H * W), (idx // W) % H, idx % W
        for dz, dy, dx in nbr:
            zz, yy, xx = z + dz, y + dy, x + dx
            if not (0 <= zz < D and 0 <= yy < H and 0 <= xx < W):
                continue
            n = (zz * H + yy) * W + xx
            if flat[n]:
                continue
            age += 1
            flat[n] = flat[idx]
            heap.push((flat_img[n], age, n))
    return out


def bbox_of(mask):
    """regionprops(...)[0].bbox: (z0, y0, x0, z1, y1, x1), half-open."""
    idx = np.argwhere(mask)
    lo, hi = idx.min(0), idx.max(0) + 1
    return (int(lo[0]), int(lo[1]), int(lo[2]), int(hi[0]), int(hi[1]), int(hi[2]))


def split_component(binary_bbox, cl, tie="heap"):
    """watershed.py:95-110 for one non-convex component cropped to its bounding box (values {0, cl}): eroded cores as
    markers, `markers[unknown == 1] = 0` -- the reference compares the shell against 1, not against cl, so the shell is
    opened for flooding ONLY for the component whose label is 1; for every other label all voxels stay markers and the
    flood has nothing to do: the result is the eroded cores --, the flood, `wss[wss == 1] = 0`.
    Returns wss BEFORE the max_class shift (labels 2.. or 0)."""
    fg = erode_ball1(binary_bbox)
    bg = dilate_ball1(binary_bbox)
    unknown = bg - fg
    markers, _ = label_equal(fg, connectivity=3)          # measure.label(fg): default = full connectivity
    markers = markers + 1
    markers[unknown == 1] = 0
    wss = watershed_flood(binary_bbox, markers, tie=tie)
    wss[wss == 1] = 0
    return wss


def segment_nuclei(binary, wmin=8, it=1, max_iters=5, min_convexity=0.8, tie="heap", trace=None, degenerate="raise"):
    """watershed.py:40-150 (species / intensity only ride along in the reference and never influence R).
    Returns R float64 like the reference.  trace (list) collects (it, cl, count, convexity, branch) tuples."""
    R = np.zeros(binary.shape)
    binary = np.asarray(binary).astype(int)
    labels, _ = label_equal(binary, connectivity=1)
    seg_classes, seg_counts = np.unique(labels, return_counts=True)
    seg_classes = np.array([seg_classes[i] for i in range(len(seg_classes)) if seg_counts[i] > 3])
    seg_classes = seg_classes[seg_classes != 0]
    for cl in seg_classes:
        binary_cl = np.where(labels == cl, labels, 0)
        bb = bbox_of(binary_cl != 0)
        sl = (slice(bb[0], bb[3]), slice(bb[1], bb[4]), slice(bb[2], bb[5]))
        binary_bbox = binary_cl[sl]
        chull = convex_hull_image(binary_bbox, degenerate=degenerate)
        convexity = np.count_nonzero(binary_bbox) / np.count_nonzero(chull)
        if convexity >= min_convexity:
            max_class = np.max(R)
            R[sl] = np.where(binary_bbox == cl, max_class + 1, R[sl])
            if trace is not None:
                trace.append((it, int(cl), int(np.count_nonzero(binary_bbox)), float(convexity), "convex"))
            continue
        wss = split_component(binary_bbox, cl, tie=tie)
        max_class = np.max(R)
        wss = wss + max_class
        wss[wss == max_class] = 0
        nclasses = len(np.unique(wss)) - 1
        if int(np.count_nonzero(wss) / wmin) > nclasses and it < max_iters:
            if trace is not None:
                trace.append((it, int(cl), int(np.count_nonzero(binary_bbox)), float(convexity), "recurse"))
            Rp = segment_nuclei(wss, it=it + 1, max_iters=max_iters, min_convexity=min_convexity, tie=tie, trace=trace,
                                degenerate=degenerate)
            max_class = np.max(R)
            Rp = Rp + max_class
            Rp[Rp == max_class] = 0
            R[sl] = np.where(Rp != 0, Rp, R[sl])
        else:
            if trace is not None:
                trace.append((it, int(cl), int(np.count_nonzero(binary_bbox)), float(convexity), "split"))
            R[sl] = np.where(wss != 0, wss, R[sl])
    return R


def watershed_clustering(M, S, Sb, max_iters=5, tie="heap", trace=None, degenerate="raise"):
    """watershed.py:190-203: (atoms, means, R).  A flat kept component raises scipy's QhullError, as in the reference
    stack (convex_hull_image above), unless degenerate="solid"."""
    S = np.asarray(S).squeeze()
    Sb = np.asarray(Sb).squeeze()
    R = segment_nuclei(Sb, max_iters=max_iters, tie=tie, trace=trace, degenerate=degenerate)
    atoms, means = centroids(S.astype(np.int64), R)
    return atoms, means, R
