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
