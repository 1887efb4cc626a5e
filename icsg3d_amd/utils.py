"""The one piece of /root/reference/utils.py the training scripts need: data_split (utils.py:36-61).
Everything else there (voxelisation, lattice parameters, pymatgen) is out of scope."""
import os

import numpy as np


def data_split(path, n, frac=0.8, n_rot=10, seed=28):
    """Deterministic train/validation split over the UN-rotated ids found in
    <path>/density_matrices (sorted, shuffled with `seed`), each expanded to its `n_rot` rotated
    copies "<id>_rot_<k>.npy" as create_matrices.py names them.  n caps the number of base ids * n_rot."""
    folder = os.path.join(path, "density_matrices")
    base = sorted(f[:-4] for f in os.listdir(folder) if f.endswith(".npy") and "_rot_" not in f)
    rng = np.random.RandomState(seed)
    rng.shuffle(base)
    n_base = max(1, min(len(base), int(n / max(n_rot, 1)) if n_rot else n))
    base = base[:n_base]
    cut = int(len(base) * frac)

    def expand(ids):
        out = []
        for i in ids:
            out.append(i + ".npy")
            for k in range(n_rot):
                f = "%s_rot_%d.npy" % (i, k)
                if os.path.exists(os.path.join(folder, f)):
                    out.append(f)
        return out

    train, val = expand(base[:cut]), expand(base[cut:])
    assert not set(train) & set(val)
    return train, val
