#!/usr/bin/env python3
"""Hot-path portion of /root/reference/generate.py on the MI355X engine: encode a base compound, sample
z ~ N(z_mean, var), decode, segment with the U-Net, argmax species + 0.8 binary threshold, then connected
components + majority vote + centroids -- one device-resident chain per batch (generate.py:196-236 ->
ics_vae_decode_to_unet_atoms) -- and watershed_clustering's convexity test + recursive marker watershed for the
non-convex components (icsg3d_amd/watershed.py; the scikit-image routines behind it are restated, parity unpinned).
Written under output/results/<formula>__v=<var>/: densities, species, binary masks, voxel parameters and per-sample
atom coordinates (.npy).  The steps after that in the reference (CIF writing with pymatgen, CGCNN property
prediction, generate.py:247-300) are out of scope.

Flags: the reference's, same names and defaults (generate.py:52-102): --name --base --batch_size --nsamples --var
--eps_frac --clus_iters --alpha --beta --gamma --target --ncond --d.  --alpha/--beta/--gamma only feed
to_pymatgen_structure (not built): accepted, recorded in run.json; --clus_iters is watershed_clustering's max_iters.
--eps_frac goes where the reference puts it: the shift of the atom coordinates (generate.py:237-241); like the
reference, to_lattice_params / to_voxel_params keep their default 0.25 (generate.py:211,214).
Added: --channels (the reference hard-codes 4) and --synthetic (a Gaussian-blob base compound, condition bin
--cond_bin; no dataset exists offline).
"""
import argparse
import json
import os

import numpy as np

from icsg3d_amd.synthetic import synthetic_batch
from icsg3d_amd.utils import to_lattice_params_from_minmax, to_voxel_params
from icsg3d_amd.unet.unet import AtomUnet
from icsg3d_amd.vae.lattice_vae import LatticeDFCVAE


def base_compound_from_csv(csv_path, base, target, ncond):
    """generate.py:112-137,190-191: the base compound's task id, formula, target value and qcut condition bin."""
    import pandas as pd
    df = pd.read_csv(csv_path)
    df["interval"] = pd.qcut(df[target], ncond, np.arange(ncond))
    if base.startswith("mp-"):
        formula = df[df["task_id"] == base]["pretty_formula"].values[0]
        task_id = base
    else:
        formula = df[df["pretty_formula"] == base]["pretty_formula"].values[0]
        task_id = df[df["pretty_formula"] == base]["task_id"].values[0]
    row = df[df["task_id"] == task_id]
    return task_id, formula, float(row[target].values[0]), int(row["interval"].values[0])


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--name", metavar="name", type=str, help="Name of data folder")
    p.add_argument("--base", metavar="base", type=str, help="Base Compound", default="LaFeO3")
    p.add_argument("--batch_size", metavar="batch_size", type=int, help="Batch size", default=100)
    p.add_argument("--nsamples", metavar="nsamples", type=int, help="Number of samples", default=100)
    p.add_argument("--var", metavar="var", type=float, help="Variance of sampling", default=0.5)
    p.add_argument("--eps_frac", metavar="eps_frac", type=float, help="Eps of lattice vector", default=0.25)
    p.add_argument("--clus_iters", metavar="clus_iters", type=int, help="Iterations of Clustering", default=5)
    p.add_argument("--alpha", metavar="alpha", type=int, help="alpha", default=90)
    p.add_argument("--beta", metavar="beta", type=int, help="beta", default=90)
    p.add_argument("--gamma", metavar="gamma", type=int, help="gamma", default=90)
    p.add_argument("--target", metavar="target", type=str, default="formation_energy_per_atom")
    p.add_argument("--ncond", metavar="ncond", type=int, help="Number of condition bins", default=10)
    p.add_argument("--d", "--dim", dest="d", metavar="d", type=int, help="Number of map voxels", default=32)
    p.add_argument("--channels", type=int, default=4, help="input channels (density + 3 coordinate grids)")
    p.add_argument("--synthetic", action="store_true", help="Gaussian-blob base compound instead of data/<name>")
    p.add_argument("--cond_bin", type=int, default=0, help="--synthetic: condition bin of the base compound")
    a = p.parse_args()

    mode, d, C, bs, eps = a.name, a.d, a.channels, a.batch_size, a.eps_frac
    path = os.path.join("data", mode, "matrices")
    vae_weights = os.path.join("saved_models", "vae", mode, "vae_weights_" + mode + ".best.hdf5")
    unet_weights = os.path.join("saved_models", "unet", mode, "unet_weights_" + mode + ".best.hdf5")
    perceptual_model = os.path.join("saved_models", "unet", mode, "unet_weights_" + mode + ".best.h5")

    if a.synthetic:
        base_compound, base_formula, base_target_value, cond_bin = "synthetic", "synthetic", None, a.cond_bin
        M_base = synthetic_batch(1, d, C, seed=0)[0]
    else:
        base_compound, base_formula, base_target_value, cond_bin = base_compound_from_csv(
            os.path.join("data", mode, mode + ".csv"), a.base, a.target, a.ncond)
        M_base = np.load(os.path.join(path, "density_matrices", base_compound + ".npy")).reshape(1, d, d, d, 1)
        if C > 1:
            C_base = np.load(os.path.join(path, "coordinate_grids", base_compound + ".npy")).reshape(1, d, d, d, 3)
            M_base = np.concatenate([M_base, C_base], axis=-1)
    cond = np.zeros((1, a.ncond), np.float32)
    cond[0, cond_bin] = 1.0

    out_dir = os.path.join("output", "results", base_formula + "_" + "_v=" + str(a.var))
    for sub in ("densities", "species", "binary", "voxel_params", "coords"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    with open(os.path.join(out_dir, "run.json"), "w") as f:
        json.dump(dict(vars(a), base_compound=base_compound, base_formula=base_formula, base_target=base_target_value,
                       cond_bin=cond_bin), f)

    vae = LatticeDFCVAE(input_shape=(d, d, d, C), perceptual_model=perceptual_model, cond_shape=a.ncond)
    vae._set_model(vae_weights, batch_size=bs)
    unet = AtomUnet(weights=unet_weights, input_shape=(d, d, d, C), max_batch=bs)

    z_mu_base, z_logvar_base, z_base = vae.encoder.predict([M_base, cond])
    n_atoms = []
    for batch in range(int(a.nsamples / bs)):
        print("Batch", batch)
        z_samples = np.random.normal(z_mu_base, a.var, size=(bs, vae.latent_dim))
        # generate.py:204-236 as one device-resident chain: decoder -> U-Net -> argmax / 0.8 threshold -> connected
        # components (> 3 voxels) -> majority vote + centroids.  The reconstruction and the label volumes stay in HBM;
        # back come 2 bytes per voxel, the density channel, the coordinate channels' min / max (all to_lattice_params
        # reads, generate.py:211-217) and one row of integers per atom.
        # then, per sample, the convexity test and the recursive marker watershed of the non-convex components
        # (watershed.py:76-150, --clus_iters): icsg3d_amd/watershed.py.  A sample that fails is skipped like in the
        # reference ("Failed", continue: generate.py:228-236)
        out = vae.decode_segment_atoms(z_samples, np.tile(cond, (bs, 1)), unet, thresh=0.8, min_voxels=3, max_atoms=2048,
                                       split=True, max_iters=a.clus_iters)
        l_prime = to_lattice_params_from_minmax(out["coord_minmax"], d=d)
        dv_pred = to_voxel_params(l_prime, d=d)
        for i in range(bs):
            k = batch * bs + i
            if out["failed"][i]:
                print("Failed", k)
                continue
            species_sample, mu = out["atoms"][i]
            mu = np.array(mu, np.float64).reshape(len(species_sample), 3)
            mu = mu * dv_pred[i] - (eps * l_prime[i]) + (dv_pred[i] / 2.0)              # generate.py:237-241
            coords = np.concatenate([np.array(species_sample, np.float64).reshape(-1, 1), mu], axis=-1)
            n_atoms.append(len(species_sample))
            np.save(os.path.join(out_dir, "densities", "%d.npy" % k), out["density"][i])
            np.save(os.path.join(out_dir, "species", "%d.npy" % k), out["species"][i])
            np.save(os.path.join(out_dir, "binary", "%d.npy" % k), out["mask"][i])
            np.save(os.path.join(out_dir, "voxel_params", "%d.npy" % k), dv_pred[i])
            np.save(os.path.join(out_dir, "coords", "%d.npy" % k), coords)
    print("wrote", out_dir, "- atoms per sample: mean %.1f" % (np.mean(n_atoms) if n_atoms else 0.0))
