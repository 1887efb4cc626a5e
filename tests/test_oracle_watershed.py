"""oracle/watershed_ref.py against tests/golden/watershed_golden.npz, whose (atoms, means) come from the reference's
own centroids / majority_vote (tests/golden/make_watershed_golden.py)."""
import os

import numpy as np
import pytest

from oracle import watershed_ref as W

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "watershed_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})


@pytest.mark.parametrize("case", CASES)
def test_oracle_regions_and_centroids_match_reference_outputs(case):
    mask, species = GOLD[case + "/mask"], GOLD[case + "/species"]
    atoms, means, R, ncomp, nkept = W.watershed_clustering_convex(species, mask)
    assert np.array_equal(R, GOLD[case + "/R"])
    assert [ncomp, nkept] == list(GOLD[case + "/counts"])
    assert list(atoms) == list(GOLD[case + "/atoms"])
    got = np.array(means, np.float64).reshape(len(atoms), 3)
    assert np.array_equal(got, GOLD[case + "/means"])          # integer sums / counts in float64: exact


def test_vote_tie_goes_to_the_larger_species_and_full_volume_quirk():
    mask, species = GOLD["handmade/mask"], GOLD["handmade/species"]
    atoms = list(GOLD["handmade/atoms"])
    assert atoms[0] == 21 and 10 in atoms and 9 not in atoms   # 8 vs 8 voxels of 7 / 21; 32 vs 32 of 9 / 10
    assert len(GOLD["full/atoms"]) == 0                        # np.unique(R)[1:] drops the only region (no background)


# ======================================================================================================================
# segment_nuclei (watershed.py:40-150): the restated skimage routines -- PARITY UNPINNED (skimage absent); what can be
# checked here is that each restatement does what its definition says (brute force / scipy.ndimage) and that the
# recursion behaves as the reference's control flow prescribes.
# ======================================================================================================================
def _balls(d, specs):
    zz, yy, xx = np.mgrid[:d, :d, :d]
    m = np.zeros((d, d, d), bool)
    for (c, r) in specs:
        m |= (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2 <= r * r
    return m.astype(np.int32)


def _brute_label(vol, conn26):
    """flood fill in raster order: components of equal non-zero value"""
    D, H, Wd = vol.shape
    lab = np.zeros(vol.shape, np.int32)
    n = 0
    nb = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)
          if (dz, dy, dx) != (0, 0, 0) and (conn26 or abs(dz) + abs(dy) + abs(dx) == 1)]
    for z in range(D):
        for y in range(H):
            for x in range(Wd):
                if vol[z, y, x] == 0 or lab[z, y, x]:
                    continue
                n += 1
                stack = [(z, y, x)]
                lab[z, y, x] = n
                while stack:
                    a, b, c = stack.pop()
                    for dz, dy, dx in nb:
                        p = (a + dz, b + dy, c + dx)
                        if 0 <= p[0] < D and 0 <= p[1] < H and 0 <= p[2] < Wd and not lab[p] and vol[p] == vol[a, b, c]:
                            lab[p] = n
                            stack.append(p)
    return lab, n


@pytest.mark.parametrize("conn", [1, 3])
def test_label_equal_is_equal_value_components_in_raster_order(conn):
    rng = np.random.default_rng(conn)
    vol = rng.integers(0, 4, size=(7, 9, 6)).astype(np.int32)
    lab, n = W.label_equal(vol, connectivity=conn)
    ref, nref = _brute_label(vol, conn == 3)
    assert n == nref and np.array_equal(lab, ref)


def test_ball1_morphology_ignores_out_of_box_neighbours():
    rng = np.random.default_rng(5)
    a = (rng.uniform(size=(6, 5, 7)) < 0.6).astype(np.int32) * 3
    er, di = W.erode_ball1(a), W.dilate_ball1(a)
    D, H, Wd = a.shape
    for z in range(D):
        for y in range(H):
            for x in range(Wd):
                nb = [a[z, y, x]] + [a[p] for p in ((z - 1, y, x), (z + 1, y, x), (z, y - 1, x), (z, y + 1, x), (z, y, x - 1),
                                                    (z, y, x + 1)) if 0 <= p[0] < D and 0 <= p[1] < H and 0 <= p[2] < Wd]
                assert er[z, y, x] == min(nb) and di[z, y, x] == max(nb)


def test_convex_hull_image_solid_shapes():
    cube = np.ones((4, 5, 3), np.int32)
    assert W.convex_hull_image(cube).all()
    ball = _balls(13, [((6, 6, 6), 5)])
    hull = W.convex_hull_image(ball)
    assert hull[ball != 0].all()
    assert np.count_nonzero(ball) / np.count_nonzero(hull) > 0.8            # 0.84: the +-0.5 offsets give every voxel an extent
    two = _balls(16, [((5, 5, 4), 3), ((5, 5, 11), 3)])                     # two balls joined by nothing: hull fills the gap
    two[5, 5, 4:12] = 1
    hull2 = W.convex_hull_image(two)
    assert hull2[two != 0].all() and np.count_nonzero(two) / np.count_nonzero(hull2) < 0.8
    # coplanar / collinear voxels: scikit-image 0.17.2's unguarded ConvexHull(coords) raises QhullError (the reference's
    # generate.py then skips the sample); "solid" is the lenient alternative
    from scipy.spatial import QhullError
    flat = np.zeros((3, 6, 6), np.int32)
    flat[1, 1:5, 1:5] = 1
    line = np.zeros((3, 3, 6), np.int32)
    line[1, 1, 1:5] = 1
    for v in (flat, line):
        with pytest.raises(QhullError):
            W.convex_hull_image(v)
    assert W.convex_hull_image(flat, degenerate="solid")[1, 1:5, 1:5].all()
    from icsg3d_amd.watershed import DegenerateComponent, convex_hull_volume
    for v in (flat, line):
        with pytest.raises(DegenerateComponent):
            convex_hull_volume(v)
        assert convex_hull_volume(v, degenerate="solid") == np.count_nonzero(W.convex_hull_image(v, degenerate="solid"))


def test_watershed_flood_hand_example_and_tie_rules():
    # 1 x 1 x 9 line: image 0 0 1 1 1 1 1 0 0, markers 1 . . 2 . 3 . . 1 : the background (level 0) floods first and
    # takes the level-1 voxels next to it; the two cores then share what is left
    img = np.array([0, 0, 1, 1, 1, 1, 1, 0, 0]).reshape(1, 1, 9)
    mk = np.array([1, 0, 0, 2, 0, 3, 0, 0, 1]).reshape(1, 1, 9)
    out = W.watershed_flood(img, mk).ravel()
    assert list(out) == [1, 1, 1, 2, 2, 3, 1, 1, 1]
    assert list(W.watershed_flood(img, mk, tie="fifo").ravel()) == [1, 1, 1, 2, 2, 3, 1, 1, 1]
    # every voxel reachable from a marker gets a marker's label, whatever the tie rule
    rng = np.random.default_rng(2)
    img = (rng.uniform(size=(6, 7, 8)) < 0.5).astype(np.int32)
    mk = np.zeros(img.shape, np.int32)
    mk[0, 0, 0], mk[3, 3, 4], mk[5, 6, 7] = 1, 2, 3
    for tie in ("heap", "fifo"):
        o = W.watershed_flood(img, mk, tie=tie)
        assert o.min() >= 1 and set(np.unique(o)) <= {1, 2, 3} and (o[mk != 0] == mk[mk != 0]).all()


def test_heap_emulation_pops_equal_keys_in_array_heap_order():
    h = W._Heap()
    for k in "abc":
        h.push((1.0, 0, k))
    assert [h.pop()[2] for _ in range(3)] == ["a", "c", "b"]                # not FIFO: the last element moves to the root


def test_segment_nuclei_control_flow():
    d = 32
    one = _balls(d, [((10, 10, 10), 4)])
    tr = []
    R = W.segment_nuclei(one, trace=tr)
    assert [t[4] for t in tr] == ["convex"] and set(np.unique(R)) == {0.0, 1.0} and np.array_equal(R != 0, one != 0)
    # two touching balls as the FIRST component (label 1): not convex -> split (the shell opens for label 1 only)
    m = _balls(d, [((6, 6, 6), 4), ((6, 6, 12), 4), ((20, 20, 10), 4), ((20, 20, 16), 4), ((20, 8, 24), 3)])
    tr = []
    R = W.segment_nuclei(m, trace=tr)
    top = [t for t in tr if t[0] == 1]
    assert [t[4] for t in top] == ["recurse", "recurse", "convex"] and top[0][3] < 0.8
    assert max(t[0] for t in tr) >= 2                                       # the recursion ran
    assert ((R != 0) <= (m != 0)).all()                                     # regions only inside the mask
    # specks of <= 3 voxels never become regions
    sp = np.zeros((d, d, d), np.int32)
    sp[1, 1, 1:4] = 1
    assert not W.segment_nuclei(sp).any()
    # max_iters = 1: no recursion, the first split is final
    tr1 = []
    W.segment_nuclei(m, max_iters=1, trace=tr1)
    assert all(t[0] == 1 for t in tr1) and "recurse" not in [t[4] for t in tr1]


def test_product_convexity_test_matches_oracle_hull():
    """icsg3d_amd.watershed.convex_hull_volume (host side of the product: Qhull through scipy, as skimage does) against
    the oracle's hull image -- no GPU needed."""
    from icsg3d_amd.watershed import convex_hull_volume
    rng = np.random.default_rng(9)
    for _ in range(6):
        specs = [((rng.integers(4, 12), rng.integers(4, 12), rng.integers(4, 12)), rng.integers(2, 5)) for _ in range(2)]
        box = _balls(16, specs)
        assert convex_hull_volume(box) == np.count_nonzero(W.convex_hull_image(box))


def test_convexity_prefilter_decides_exactly_like_the_hull():
    """icsg3d_amd.watershed.convexity_at_least: the flat test (exact integers) fails exactly the sets Qhull refuses, the
    26-direction polytope never counts fewer grid points than the hull, and the decision `convexity >= threshold` is the
    hull's for every threshold -- on balls, unions of balls, random subsets and thin / flat sets -- no GPU needed."""
    from scipy.spatial import QhullError
    from icsg3d_amd.watershed import DegenerateComponent, convexity_at_least, convexity_many, dop_count, fill_count, is_flat
    rng = np.random.default_rng(21)
    cases = []
    for _ in range(30):
        specs = [((rng.integers(3, 9), rng.integers(3, 9), rng.integers(3, 9)), rng.integers(1, 4)) for _ in range(rng.integers(1, 4))]
        cases.append(_balls(12, specs))
    for _ in range(30):                                   # sparse random subsets: ragged, mostly non-convex
        dims = tuple(int(v) for v in rng.integers(2, 7, size=3))
        cases.append((rng.uniform(size=dims) < rng.uniform(0.2, 0.9)).astype(np.int32))
    for _ in range(20):                                   # plates, lines, L-shapes in a plane, two parallel layers
        a = np.zeros((5, 6, 6), np.int32)
        kind = rng.integers(0, 4)
        if kind == 0:
            a[2, 1:rng.integers(3, 6), 1:rng.integers(3, 6)] = 1
        elif kind == 1:
            a[2, 2, 1:rng.integers(4, 6)] = 1
        elif kind == 2:
            a[2, 1:4, 1] = 1; a[2, 1, 1:5] = 1
        else:
            a[1:3, 1:4, 1:4] = 1
        cases.append(a)
    decided_cheaply = flat = decided_nonconvex = 0
    solid = []
    for img in cases:
        n = int(np.count_nonzero(img))
        if n < 4:
            continue
        pts = np.argwhere(img != 0)
        try:
            hull = np.count_nonzero(W.convex_hull_image(img))
            qhull_refuses = False
        except QhullError:
            qhull_refuses = True
        assert is_flat(pts) == qhull_refuses, pts.tolist()
        if qhull_refuses:
            flat += 1
            with pytest.raises(DegenerateComponent):
                convexity_at_least(img, 0.8)
            hull = np.count_nonzero(W.convex_hull_image(img, degenerate="solid"))
        assert dop_count(img) >= hull >= fill_count(img) >= n      # outer polytope >= hull >= axis-line fill >= the set
        for thr in (0.5, 0.8, 0.9, 1.0):
            ok, val = convexity_at_least(img, thr, degenerate="solid")
            assert ok == (n / hull >= thr), (thr, n, hull, val)
            decided_cheaply += int(ok and val != n / hull)
            decided_nonconvex += int((not ok) and val != n / hull)
        solid.append((img, n, hull))
    assert flat >= 10 and decided_cheaply >= 20           # both branches were exercised
    assert decided_nonconvex >= 10                        # ... and the upper bound settled some non-convex ones without a hull
    # the batched form (thread pool for the hulls the bounds leave open) takes the same decisions, in order
    many = convexity_many([c[0] for c in solid], 0.8, degenerate="solid")
    assert [m[0] for m in many] == [n / hull >= 0.8 for _, n, hull in solid]
    assert [m[0] for m in convexity_many([c[0] for c in solid], 0.8, degenerate="solid", pool=False)] == [m[0] for m in many]


def test_breadth_first_host_logic_against_the_oracle_without_a_gpu(monkeypatch):
    """icsg3d_amd.watershed.segment_nuclei_batch / _assemble / refine bookkeeping are HOST logic: with the two device
    primitives replaced by the oracle's own routines (label_equal, split_component) the breadth-first batch must return
    the oracle's depth-first R, visiting order and failures -- on CPU."""
    from scipy.spatial import QhullError
    import icsg3d_amd.watershed as P

    def label_boxes(vols, connectivity=1, max_labels=1024, want_bounds=False, min_voxels=3, hull_threshold=0.0):
        out = []
        for v in vols:
            lab, n = W.label_equal(v, connectivity=connectivity)
            stats = np.zeros((n, 7), np.int32)
            bounds = np.zeros((n, 5), np.int64)           # H (exact hull count) left 0: the hulls go through Qhull here
            for cl in range(1, n + 1):
                m = lab == cl
                stats[cl - 1] = (int(m.sum()),) + W.bbox_of(m)
                if stats[cl - 1, 0] > min_voxels:      # the bounds the C++ op returns, from the numpy definitions
                    bounds[cl - 1, :4] = (int(m.sum()), P.dop_count(m), P.fill_count(m), int(P.is_flat(np.argwhere(m))))
            out.append((lab, n, stats, bounds) if want_bounds else (lab, n, stats))
        return out

    monkeypatch.setattr(P, "label_boxes", label_boxes)
    monkeypatch.setattr(P, "watershed_split", lambda boxes, cls, tie="heap": [W.split_component(b, c, tie=tie).astype(np.int32)
                                                                             for b, c in zip(boxes, cls)])
    rng = np.random.default_rng(31)
    vols = [_balls(24, [((7, 7, 5), 4), ((7, 7, 11), 4), ((7, 13, 8), 4), ((17, 17, 17), 4)]),
            _balls(24, [((8, 8, 8), 3), ((16, 15, 10), 3)]),
            (rng.uniform(size=(16, 16, 16)) < 0.3).astype(np.int32)]
    for mode in ("solid", "raise"):
        traces = [[] for _ in vols]
        Rs, errors = P.segment_nuclei_batch(vols, traces=traces, degenerate=mode)
        for i, v in enumerate(vols):
            tr = []
            try:
                R_ref = W.segment_nuclei(v, trace=tr, degenerate=mode)
            except QhullError:
                assert errors[i] is not None and Rs[i] is None
                continue
            assert errors[i] is None and np.array_equal(Rs[i], R_ref), (mode, i)
            assert [(t[0], t[1], t[2], t[4]) for t in traces[i]] == [(t[0], t[1], t[2], t[4]) for t in tr]
    assert any(t[4] in ("recurse", "split") for t in traces[0])
    # the single-volume wrapper and the depth-first form give the same thing
    assert np.array_equal(P.segment_nuclei(vols[0]), P._segment_nuclei_recursive(vols[0]))


def test_component_bounds_op_equals_the_numpy_definitions():
    """ics_op_component_bounds (host threads inside the library, no device work): {voxels, P, F, flat} of every component of
    labelled boxes equal icsg3d_amd.watershed's numpy definitions dop_count / fill_count / is_flat -- which the test above
    holds to Qhull's hull (P >= hull >= F, flat <=> Qhull refuses) -- and H, the EXACT hull count (integer gift wrapping on
    doubled coordinates), equals np.count_nonzero(convex_hull_image(component)) as Qhull + the reference's 1e-10 tolerance
    give it, on every component the bounds leave undecided: ragged random volumes, smooth blobs, unions of balls with holes,
    plates and lines."""
    import icsg3d_amd.watershed as P
    from scipy import ndimage
    rng = np.random.default_rng(8)
    vols = [(rng.uniform(size=(16, 12, 9)) < 0.3).astype(np.int32), (rng.uniform(size=(20, 20, 20)) < 0.5).astype(np.int32),
            _balls(24, [((7, 7, 5), 4), ((7, 7, 11), 4), ((17, 17, 17), 5)]), np.zeros((4, 4, 4), np.int32)]
    plate = np.zeros((8, 8, 8), np.int32); plate[3, 1:6, 1:7] = 1; plate[6, 2, 1:7] = 1
    vols.append(plate)
    for _ in range(3):                                    # thresholded smooth noise: what a random-weight U-Net's mask looks like
        f = ndimage.gaussian_filter(rng.standard_normal((32, 32, 32)), 1.2)
        vols.append((f >= np.quantile(f, 0.9)).astype(np.int32))
    holes = _balls(24, [((8, 8, 8), 5), ((8, 8, 15), 5), ((15, 12, 11), 4)]) * (rng.uniform(size=(24, 24, 24)) > 0.06)
    vols.append(holes.astype(np.int32))
    labs, stats = [], []
    for v in vols:
        lab, n = W.label_equal(v, connectivity=1)
        st = np.zeros((n, 7), np.int32)
        for cl in range(1, n + 1):
            m = lab == cl
            st[cl - 1] = (int(m.sum()),) + W.bbox_of(m)
        labs.append(lab.astype(np.int32)); stats.append(st)
    for thr in (0.0, 0.8, 0.5):
        bounds = P.component_bounds(labs, stats, min_voxels=3, hull_threshold=thr)
        checked = flats = exact = 0
        for lab, st, B in zip(labs, stats, bounds):
            for cl in range(1, len(st) + 1):
                m = lab == cl
                if m.sum() <= 3:
                    assert list(B[cl - 1]) == [0, 0, 0, 0, 0]
                    continue
                n_, P_, F_ = int(m.sum()), P.dop_count(m), P.fill_count(m)
                assert list(B[cl - 1][:4]) == [n_, P_, F_, int(P.is_flat(np.argwhere(m)))], (cl, list(B[cl - 1]))
                checked += 1
                flats += int(B[cl - 1][3])
                undecided = thr > 0 and n_ / P_ < thr <= n_ / F_
                if undecided:                              # the exact count is there, and it is Qhull's
                    z0, y0, x0, z1, y1, x1 = st[cl - 1, 1:7]
                    assert B[cl - 1][4] == P.convex_hull_volume(m[z0:z1, y0:y1, x0:x1], degenerate="solid"), (cl, list(B[cl - 1]))
                    exact += 1
                else:
                    assert B[cl - 1][4] == 0
        assert checked > 100 and flats >= 2
        assert exact == 0 if thr == 0.0 else exact > 20, (thr, exact)
    # the decisions taken from these rows are convexity_at_least's
    B08 = P.component_bounds(labs, stats, min_voxels=3, hull_threshold=0.8)
    for lab, st, B in zip(labs[5:7], stats[5:7], B08[5:7]):
        boxes, rows = [], []
        for cl in range(1, len(st) + 1):
            if st[cl - 1, 0] > 3:
                z0, y0, x0, z1, y1, x1 = st[cl - 1, 1:7]
                boxes.append(lab[z0:z1, y0:y1, x0:x1] == cl); rows.append(B[cl - 1])
        got = P.convexity_many(boxes, 0.8, degenerate="solid", bounds=rows)
        assert [g[0] for g in got] == [P.convexity_at_least(b, 0.8, degenerate="solid")[0] for b in boxes]
