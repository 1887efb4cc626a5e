"""Device connected components + region statistics (csrc/segment.hip) through the C ABI:
bit-exact against (1) tests/golden/watershed_golden.npz -- (atoms, means) produced by the reference's own
centroids / majority_vote (/root/reference/watershed.py:153-187) -- and (2) oracle/watershed_ref.py
(scipy.ndimage.label, 6-connectivity) on random volumes, batches and the 64^3 grid.  Integer work: every comparison
is exact, centroids included (integer sums / counts divided in float64 on both sides)."""
import os

import numpy as np
import pytest

from oracle import watershed_ref as W

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "watershed_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})


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
    # the reference-shaped entry point (watershed.py:190)
    a2, m2, R2 = watershed_clustering(np.zeros_like(mask, dtype=np.float32), species, mask, return_ws=True)
    assert list(a2) == atoms and np.array_equal(R2, GOLD[case + "/R"])


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
    with pytest.raises(_lib.IcsError, match="max_atoms"):
        segment_atoms(mask, mask, max_atoms=4)
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
