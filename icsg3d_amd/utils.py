"""The one piece of /root/reference/utils.py the training scripts need: data_split (utils.py:36-61).
Everything else there (voxelisation, lattice parameters, pymatgen) is out of scope.

data_split is index work, so it is reproduced bit-exactly -- including the reference's quirks:
  * the cap `n` counts UN-rotated base ids and is applied BEFORE the shuffle (utils.py:44);
  * the shuffle is Python's `random` module seeded through the GLOBAL generator (utils.py:46-48);
  * every base id is expanded to `n_rot` names "<id>_rot_<k>.npy" whether or not the file exists
    (utils.py:53-60);
  * the stem is taken with str.strip(".npy"), which strips the CHARACTERS '.', 'n', 'p', 'y' from both
    ends, not the suffix (utils.py:55,59) -- "mp-123.npy" -> "mp-123", "nacl.npy" -> "acl".
tests/golden/data_split_golden.json holds id lists produced by the reference function itself
(tests/golden/make_data_split_golden.py); tests/test_host_logic.py compares against them."""
import os
import random

import numpy as np


def _rotated_names(base_name, n_rot):
    stem = base_name.strip(".npy")          # character strip, as the reference does
    return [stem + "_rot_" + str(k) + ".npy" for k in range(n_rot)]


def data_split(path, n=None, frac=0.80, n_rot=10, shuffle=True, seed=28):
    """Train/validation split of the matrices under <path>/density_matrices -> (training_ids, validation_ids)."""
    listing = sorted(f for f in os.listdir(path + "/density_matrices") if f.endswith(".npy"))
    base = [f for f in listing if "_rot_" not in f][:n]
    if shuffle:
        if seed is not None:
            random.seed(seed)               # the global generator, like the reference (callers see the reseed)
        random.shuffle(base)
    cut = int(frac * len(base))
    split = []
    for part in (base[:cut], base[cut:]):
        ids = []
        for name in part:
            ids.append(name)
            ids.extend(_rotated_names(name, n_rot))
        split.append(ids)
    training_ids, validation_ids = split
    assert not set(training_ids) & set(validation_ids)
    return training_ids, validation_ids


def to_lattice_params_from_minmax(coord_minmax, eps_frac=0.25, d=32):
    """to_lattice_params (/root/reference/utils.py:160-178) given only what it reads of the coordinate
    channels: per sample and channel the {min, max} over the grid, as returned by the fused inference tail
    (ics_vae_decode_to_unet_labels).  coord_minmax: (B, 3, 2) -> lattice params (B, 3)."""
    mm = np.asarray(coord_minmax, dtype=np.float64)
    lp = mm[:, :, 1] - mm[:, :, 0]
    lp = lp / (1 + 2 * eps_frac)
    lp = lp / (1 - 1.0 / d)
    lp -= lp / d
    return lp


def to_lattice_params(p, eps_frac=0.25, d=32):
    """Same from the coordinate grids themselves, (B, d, d, d, 3) (utils.py:160-178)."""
    p = np.asarray(p)
    mm = np.stack([p[..., :3].min(axis=(1, 2, 3)), p[..., :3].max(axis=(1, 2, 3))], axis=-1)
    return to_lattice_params_from_minmax(mm, eps_frac, d)


def to_voxel_params(lp, eps=0.25, d=32):
    """Voxel edge lengths from lattice params (utils.py:181-190)."""
    lp = np.asarray(lp, dtype=np.float64)
    return (lp + 2 * lp * eps) / d
