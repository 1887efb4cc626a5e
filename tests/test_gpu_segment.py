"""Device connected components, region statistics and the marker-watershed split (csrc/segment.hip)
through the C ABI: bit-exact against (1) tests/golden/watershed_golden.npz -- (atoms, means) produced by the
reference's own centroids / majority_vote (/root/reference/watershed.py:153-187) -- and (2) oracle/watershed_ref.py on
random volumes, batches and the 64^3 grid, including touching-blob cases that force a split and a recursion of
segment_nuclei (watershed.py:40-150; the scikit-image routines behind it are RESTATED in the oracle: parity unpinned,
skimage absent).  Integer work: every comparison is exact, centroids included (integer sums / counts divided in float64
on both sides)."""
import os

import numpy as np
import pytest

from oracle import watershed_ref as W

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "watershed_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})
FLAT_CASES = {"handmade", "noise"}      # hold a kept component (> 3 voxels) whose voxels are coplanar / collinear


@pytest.mark.parametrize("case", CASES)
def test_segment_atoms_matches_reference_outputs(case):
    from icsg3d_amd.watershed import segment_atoms, watershed_clustering
    mask, species = GOLD[case + "/mask"], GOLD[case + "/species"]
    r = segment_atoms(mask[None], species[None], min_voxels=3, max_atoms=512)
    assert np.array_equal(r["regions"][0], GOLD[case + "/R"])
    assert [int(r["n_components"][0]), int(r["n_atoms"][0])] == list(GOLD[case + "/counts"])
    atoms, means = r["atoms"][0]
    assert atoms == list(GOLD[case + "/atoms"])
    assert np.array_equal(np.array(means, np.float64).reshape(len(atoms), 3), GOLD[case + "/means"])
    # the reference-shaped entry point (watershed.py:190): the full path, convexity test and splits included.
    # A kept component that is flat (coplanar / collinear voxels: the 4-voxel line of 'hand', many in 'noise') makes
    # scikit-image 0.17.2's convex_hull_image raise, and the reference skips the sample (generate.py:246-248): so do we
    from scipy.spatial import QhullError
    from icsg3d_amd.watershed import DegenerateComponent
    trace = []
    try:
        W.watershed_clustering(None, species, mask, trace=trace)
        flat = False
    except QhullError:
        flat = True
    if flat:
        with pytest.raises(DegenerateComponent):
            watershed_clustering(np.zeros_like(mask, dtype=np.float32), species, mask, return_ws=True)
    assert flat == (case in FLAT_CASES), case
    mode = "solid" if flat else "raise"
    trace = []
    a2, m2, R2 = watershed_clustering(np.zeros_like(mask, dtype=np.float32), species, mask, return_ws=True, degenerate=mode)
    a3, m3, R3 = W.watershed_clustering(None, species, mask, trace=trace, degenerate=mode)
    assert np.array_equal(R2, R3) and list(a2) == list(a3)
    assert np.array_equal(np.array(m2).reshape(len(a2), 3), np.array(m3).reshape(len(a3), 3))
    # where every kept component took the convex branch the result must STILL be the reference-run golden one
    # (R of the fixture = all components convex; atoms / means = the reference's own centroids on it)
    if all(t[4] == "convex" for t in trace):
        assert np.array_equal(R2, GOLD[case + "/R"]) and list(a2) == list(GOLD[case + "/atoms"])
        assert np.array_equal(np.array(m2, np.float64).reshape(len(a2), 3), GOLD[case + "/means"])


@pytest.mark.parametrize("B,d,p", [(3, 32, 0.30), (2, 64, 0.26), (5, 16, 0.45), (1, 64, 0.7)])
def test_segment_atoms_matches_oracle_on_random_volumes(B, d, p):
    from icsg3d_amd.watershed import segment_atoms
    rng = np.random.default_rng(100 + d)
    mask = (rng.uniform(size=(B, d, d, d)) < p).astype(np.uint8)
    species = rng.integers(0, 95, size=(B, d, d, d)).astype(np.uint8)
    mask[-1, :, :, : d // 2] = 0                       # samples with different component counts
    r = segment_atoms(mask, species, min_voxels=3, max_atoms=4096)
    for b in range(B):
        R, ncomp, nkept = W.regions(mask[b])
        assert np.array_equal(r["regions"][b], R)
        assert (int(r["n_components"][b]), int(r["n_atoms"][b])) == (ncomp, nkept)
        st = r["stats"][b, :nkept]
        assert np.array_equal(st[:, 1], np.bincount(R.ravel(), minlength=nkept + 1)[1:])
        # bounding boxes (half-open, as skimage's regionprops reports them) and votes of a few regions
        for a in list(range(min(nkept, 5))) + ([nkept - 1] if nkept else []):
            idx = np.argwhere(R == a + 1)
            assert list(st[a, 5:8]) == list(idx.min(0)) and list(st[a, 8:11]) == list(idx.max(0) + 1)
            assert int(st[a, 0]) == W.majority_vote(species[b].astype(np.int64), R, a + 1)
            assert list(st[a, 2:5]) == list(idx.sum(0))


def test_segment_atoms_rejects_overflow_and_bad_shapes():
    from icsg3d_amd import _lib
    from icsg3d_amd.watershed import segment_atoms
    rng = np.random.default_rng(3)
    mask = (rng.uniform(size=(1, 32, 32, 32)) < 0.3).astype(np.uint8)
    r = segment_atoms(mask, mask, max_atoms=4)         # too many components: the SAMPLE fails, the call does not
    assert r["failed"][0] and r["atoms"][0] == ([], []) and int(r["n_components"][0]) > 4
    assert _lib is not None
    with pytest.raises(ValueError):
        segment_atoms(mask[0], mask[0])


def test_decode_to_atoms_continues_the_fused_tail_on_the_device():
    """ics_vae_decode_to_unet_atoms = ics_vae_decode_to_unet_labels + the component pass on the SAME device buffers:
    labels identical to the two-call path, atoms identical to the oracle run on those labels."""
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    from icsg3d_amd.synthetic import glorot_params, unet_param_shapes, vae_param_shapes
    B, d, C = 3, 32, 4
    ue = UnetEngine(in_channels=C, d=d, max_batch=B)
    ue.set_weights(glorot_params(unet_param_shapes(C, 95), 1))
    ve = VaeEngine(ue, in_channels=C, d=d, max_batch=B)
    ve.set_weights(glorot_params(vae_param_shapes(C, d=d), 3))
    rng = np.random.default_rng(0)
    z = rng.standard_normal((B, 256)).astype(np.float32)
    cond = np.eye(10, dtype=np.float32)[[1, 4, 7]]
    # a data-dependent threshold so that random-weight networks still produce foreground components
    thr = float(np.quantile(ue.predict(ve.decode(z, cond))[1], 0.9))
    ref = ve.decode_to_labels(ue, z, cond, thresh=thr)
    out = ve.decode_to_atoms(ue, z, cond, thresh=thr, max_atoms=4096, want_regions=True)
    for k in ("species", "mask", "density", "coord_minmax"):
        assert np.array_equal(out[k], ref[k]), k
    assert out["mask"].sum() > 0
    for b in range(B):
        atoms, means, R, ncomp, nkept = W.watershed_clustering_convex(out["species"][b], out["mask"][b])
        assert np.array_equal(out["regions"][b], R)
        assert (int(out["n_components"][b]), int(out["n_atoms"][b])) == (ncomp, nkept)
        a, m = out["atoms"][b]
        assert a == list(atoms)
        assert np.array_equal(np.array(m).reshape(len(a), 3), np.array(means).reshape(len(atoms), 3))


# ======================================================================================================================
# segment_nuclei's non-convex branch and recursion (watershed.py:40-150) -- parity unpinned: skimage absent
# ======================================================================================================================
def _balls(d, specs):
    zz, yy, xx = np.mgrid[:d, :d, :d]
    m = np.zeros((d, d, d), bool)
    for (c, r) in specs:
        m |= (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2 <= r * r
    return m.astype(np.int32)


TOUCHING = [((6, 6, 6), 4), ((6, 6, 12), 4), ((20, 20, 10), 4), ((20, 20, 16), 4), ((20, 8, 24), 3)]


@pytest.mark.parametrize("conn", [1, 3])
def test_label_boxes_matches_oracle(conn):
    from icsg3d_amd.watershed import label_boxes
    rng = np.random.default_rng(40 + conn)
    vols = [rng.integers(0, 4, size=(7, 9, 6)).astype(np.int32), (rng.uniform(size=(32, 32, 32)) < 0.3).astype(np.int32) * 5,
            rng.integers(0, 3, size=(1, 1, 17)).astype(np.int32), np.zeros((3, 3, 3), np.int32),
            (rng.uniform(size=(64, 20, 33)) < 0.45).astype(np.int32)]
    res = label_boxes(vols, connectivity=conn, max_labels=64)      # grows max_labels by itself
    for v, (lab, n, st) in zip(vols, res):
        ref, nref = W.label_equal(v, connectivity=conn)
        assert n == nref and np.array_equal(lab, ref)
        assert np.array_equal(st[:, 0], np.bincount(ref.ravel(), minlength=n + 1)[1:])
        for a in list(range(min(n, 4))) + ([n - 1] if n else []):
            idx = np.argwhere(ref == a + 1)
            assert list(st[a, 1:4]) == list(idx.min(0)) and list(st[a, 4:7]) == list(idx.max(0) + 1)


@pytest.mark.parametrize("where", ["host", "device"])
@pytest.mark.parametrize("tie", ["heap", "fifo"])
def test_watershed_split_matches_oracle(tie, where, monkeypatch):
    """Both forms of ics_op_watershed_split: the host-thread flood that ships since round 6 (the flood is a chain of
    dependent heap operations; a CPU core walks it ~30x faster than one GPU lane) and the kernel (ICSG3D_WS_DEVICE=1)."""
    from icsg3d_amd.watershed import watershed_split
    monkeypatch.setenv("ICSG3D_WS_DEVICE", "1" if where == "device" else "0")
    rng = np.random.default_rng(7)
    boxes, cls = [], []
    two = _balls(16, [((6, 6, 4), 4), ((6, 6, 10), 4)])
    z, y, x = np.nonzero(two)
    two = two[z.min():z.max() + 1, y.min():y.max() + 1, x.min():x.max() + 1]
    for cl in (1, 5):                                   # label 1: the shell opens and the flood runs; label 5: eroded cores
        boxes.append(two * cl); cls.append(cl)
    three = _balls(24, [((8, 8, 5), 4), ((8, 8, 11), 4), ((8, 13, 8), 4)])
    boxes.append(three); cls.append(1)
    for _ in range(3):                                  # random blobs: ragged extents, thin necks, several cores
        dims = tuple(int(v) for v in rng.integers(5, 14, size=3))
        b = (rng.uniform(size=dims) < 0.75).astype(np.int32)
        boxes.append(b); cls.append(1)
    big = (rng.uniform(size=(30, 31, 29)) < 0.85).astype(np.int32)    # 26970 voxels: the heap outgrows any LDS budget
    boxes.append(big); cls.append(1)
    got = watershed_split(boxes, cls, tie=tie)
    for b, cl, g in zip(boxes, cls, got):
        ref = W.split_component(b, cl, tie=tie)
        assert np.array_equal(g, ref), (b.shape, cl)
    # the two tie rules are different algorithms: on the random blobs they disagree (oracle: 40 of 40 such boxes), so
    # matching the oracle under BOTH rules means the device reproduces the heap's pop order, not just some flood
    if tie == "fifo":
        other = watershed_split(boxes, cls, tie="heap")
        assert any(not np.array_equal(a, b) for a, b in zip(got, other))
    # host and device forms agree bit for bit (many boxes: the host form deals them to threads)
    monkeypatch.setenv("ICSG3D_WS_DEVICE", "0" if where == "device" else "1")
    for a, b in zip(got, watershed_split(boxes * 5, cls * 5, tie=tie)):
        assert np.array_equal(a, b)


def test_segment_nuclei_and_clustering_match_oracle_with_splits_and_recursion():
    from icsg3d_amd.watershed import segment_nuclei, watershed_clustering
    d = 32
    m = _balls(d, TOUCHING)
    rng = np.random.default_rng(1)
    species = np.where(m != 0, rng.integers(1, 95, size=m.shape), 0).astype(np.uint8)
    tr, tr_ref = [], []
    R = segment_nuclei(m, trace=tr)
    R_ref = W.segment_nuclei(m, trace=tr_ref)
    assert np.array_equal(R, R_ref)
    assert [(t[0], t[1], t[2], t[4]) for t in tr] == [(t[0], t[1], t[2], t[4]) for t in tr_ref]
    assert "recurse" in [t[4] for t in tr] and max(t[0] for t in tr) >= 2        # a split AND a recursion happened
    a, mu, Rw = watershed_clustering(None, species, m, return_ws=True)
    a_ref, mu_ref, _ = W.watershed_clustering(None, species, m)
    assert np.array_equal(Rw, R_ref) and list(a) == list(a_ref)
    assert np.array_equal(np.array(mu).reshape(len(a), 3), np.array(mu_ref).reshape(len(a_ref), 3))
    # max_iters = 1 (--clus_iters 1): the first split is final
    assert np.array_equal(segment_nuclei(m, max_iters=1), W.segment_nuclei(m, max_iters=1))
    # a first component of <= 3 voxels takes label 1 away from the first real component: no shell opens anywhere
    m2 = m.copy()
    m2[0, 0, 0:2] = 1
    assert np.array_equal(segment_nuclei(m2), W.segment_nuclei(m2))
    # random volume at 64^3: many components, some non-convex
    # (it holds flat kept components: the reference stack fails such a sample -- both sides raise; "solid" compares the rest)
    from scipy.spatial import QhullError
    from icsg3d_amd.watershed import DegenerateComponent
    m3 = (np.random.default_rng(5).uniform(size=(64, 64, 64)) < 0.2).astype(np.int32)
    with pytest.raises(DegenerateComponent):
        segment_nuclei(m3)
    with pytest.raises(QhullError):
        W.segment_nuclei(m3)
    assert np.array_equal(segment_nuclei(m3, degenerate="solid"), W.segment_nuclei(m3, degenerate="solid"))


def test_refine_atoms_continues_the_batch_result():
    from icsg3d_amd.watershed import refine_atoms, segment_atoms
    d = 32
    plate = _balls(d, [((10, 10, 10), 3)])
    plate[25, 20:22, 20:22] = 1                          # a 2 x 2 plate: 4 coplanar voxels, kept by the size filter
    masks = np.stack([_balls(d, TOUCHING), _balls(d, [((10, 10, 10), 3), ((22, 20, 12), 3)]), plate]).astype(np.uint8)
    rng = np.random.default_rng(3)
    species = np.where(masks != 0, rng.integers(1, 95, size=masks.shape), 0).astype(np.uint8)
    out = segment_atoms(masks, species, max_atoms=64)    # carries mask and species for refine_atoms
    refine_atoms(out)
    assert list(out["split"]) == [True, False, False]
    # the flat component fails the SAMPLE as in the reference stack (generate.py:246-248), not the batch
    assert list(out["failed"]) == [False, False, True] and out["atoms"][2] == ([], [])
    from scipy.spatial import QhullError
    with pytest.raises(QhullError):
        W.watershed_clustering(None, species[2], masks[2])
    lenient = refine_atoms(segment_atoms(masks, species, max_atoms=64), degenerate="solid")
    assert not lenient["failed"].any() and len(lenient["atoms"][2][0]) == 2
    with pytest.raises(ValueError, match="needs out"):
        refine_atoms({k: v for k, v in segment_atoms(masks, species, max_atoms=64).items() if k != "mask"})
    for b in range(2):
        a_ref, mu_ref, R_ref = W.watershed_clustering(None, species[b], masks[b])
        a, mu = out["atoms"][b]
        assert np.array_equal(out["regions"][b], R_ref.astype(np.int32)) and list(a) == list(a_ref)
        assert np.array_equal(np.array(mu).reshape(len(a), 3), np.array(mu_ref).reshape(len(a_ref), 3))


@pytest.mark.parametrize("B,d,p", [(3, 32, 0.30), (2, 16, 0.45), (1, 64, 0.22)])
def test_device_convexity_bounds_equal_the_host_integers(B, d, p):
    """ics_op_segment_atoms' convexity_bounds: per region the 26-direction polytope count and the second moments, the
    integers refine_atoms decides most components from without a hull -- equal to the host functions
    (icsg3d_amd.watershed.dop_count / is_flat, themselves held to Qhull's hull in tests/test_oracle_watershed.py)."""
    from icsg3d_amd.watershed import _scatter_is_singular, dop_count, is_flat, segment_atoms
    rng = np.random.default_rng(200 + d)
    mask = (rng.uniform(size=(B, d, d, d)) < p).astype(np.uint8)
    mask[0, :3] = 0
    mask[0, 1, 2:6, 2:6] = 1                          # a 4 x 4 plate (flat, kept) ...
    mask[0, 1, 10, 8:14] = 1                          # ... and a 6-voxel line, isolated in an emptied slab
    species = rng.integers(0, 95, size=mask.shape).astype(np.uint8)
    r = segment_atoms(mask, species, min_voxels=3, max_atoms=4096)
    assert not r["failed"].any()
    nflat = ncheap = 0
    for b in range(B):
        R = r["regions"][b]
        n = int(r["n_atoms"][b])
        for a in list(range(min(n, 40))) + list(range(max(n - 5, 0), n)):
            st, bd = r["stats"][b, a], r["bounds"][b, a]
            z0, y0, x0, z1, y1, x1 = (int(v) for v in st[5:11])
            box = R[z0:z1, y0:y1, x0:x1] == a + 1
            assert int(bd[0]) == dop_count(box), (b, a)
            pts = np.argwhere(R == a + 1)
            assert [int(v) for v in bd[1:7]] == [int((pts[:, i] * pts[:, j]).sum()) for i, j in
                                                 ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))]
            flat = _scatter_is_singular(int(st[1]), [int(v) for v in st[2:5]], [int(v) for v in bd[1:7]])
            assert flat == is_flat(pts), (b, a)
            nflat += int(flat)
            ncheap += int(int(st[1]) / int(bd[0]) >= 0.8)
    assert nflat >= 2 and ncheap >= 1


def test_refine_atoms_takes_the_same_decisions_with_and_without_device_bounds():
    from icsg3d_amd.watershed import refine_atoms, segment_atoms
    d = 32
    masks = np.stack([_balls(d, TOUCHING), _balls(d, [((10, 10, 10), 3), ((22, 20, 12), 3), ((8, 24, 24), 4)]),
                      _balls(d, [((16, 16, 16), 6)])]).astype(np.uint8)
    rng = np.random.default_rng(4)
    species = np.where(masks != 0, rng.integers(1, 95, size=masks.shape), 0).astype(np.uint8)
    with_b = refine_atoms(segment_atoms(masks, species, max_atoms=64))
    no_b = segment_atoms(masks, species, max_atoms=64)
    no_b["bounds"] = None
    refine_atoms(no_b)
    assert list(with_b["split"]) == list(no_b["split"]) == [True, False, False]
    for b in range(3):
        assert np.array_equal(with_b["regions"][b], no_b["regions"][b])
        assert with_b["atoms"][b][0] == no_b["atoms"][b][0]


def test_breadth_first_batch_equals_the_depth_first_recursion():
    """segment_nuclei_batch (one label launch and one flood launch per recursion LEVEL for all samples) against the
    reference's control flow (`_segment_nuclei_recursive`: depth first, one round trip per level and parent): identical R,
    identical visiting order (traces), identical failures."""
    from icsg3d_amd.watershed import DegenerateComponent, _segment_nuclei_recursive, segment_nuclei_batch
    d = 32
    rng = np.random.default_rng(17)
    vols = [_balls(d, TOUCHING),
            _balls(d, [((10, 10, 10), 3), ((22, 20, 12), 3)]),
            _balls(d, [((8, 8, 6), 4), ((8, 8, 12), 4), ((8, 14, 9), 4), ((22, 22, 22), 5), ((22, 22, 15), 4)]),
            (rng.uniform(size=(d, d, d)) < 0.24).astype(np.int32),          # ragged: many non-convex components, some flat
            np.zeros((d, d, d), np.int32)]
    for mode in ("solid", "raise"):
        traces = [[] for _ in vols]
        Rs, errors = segment_nuclei_batch(vols, traces=traces, degenerate=mode)
        for i, v in enumerate(vols):
            tr = []
            try:
                R_ref = _segment_nuclei_recursive(v, trace=tr, degenerate=mode)
            except DegenerateComponent:
                assert errors[i] is not None and Rs[i] is None, i
                continue
            assert errors[i] is None and np.array_equal(Rs[i], R_ref), (mode, i)
            assert [(t[0], t[1], t[2], t[4]) for t in traces[i]] == [(t[0], t[1], t[2], t[4]) for t in tr], (mode, i)
    assert any(t[4] == "recurse" for t in traces[0]) and any(e is not None for e in errors)      # both paths were taken
    # max_iters reaches the batch the same way
    Rs1, _ = segment_nuclei_batch(vols[:3], max_iters=1, degenerate="solid")
    for i in range(3):
        assert np.array_equal(Rs1[i], _segment_nuclei_recursive(vols[i], max_iters=1, degenerate="solid"))
