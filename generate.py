#!/usr/bin/env python3
"""Hot-path portion of /root/reference/generate.py:171-225 on the MI355X engine: encode a base compound,
sample z ~ N(z_mean, var), decode, segment with the U-Net, argmax species + 0.8 binary threshold.
Outputs densities / species / binary masks as .npy under output/results/<base>__v=<var>/ -- the steps
after that in the reference (watershed, CIF writing, CGCNN property prediction) need pymatgen/skimage
and are out of scope.  Flags keep the reference's names/defaults (generate.py:52-102).
"""
import argparse
import os

import numpy as np

from icsg3d_amd.synthetic import synthetic_batch
from icsg3d_amd.utils import to_lattice_params_from_minmax, to_voxel_params
from icsg3d_amd.unet.unet import AtomUnet
from icsg3d_amd.vae.lattice_vae import LatticeDFCVAE

if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--name", type=str, help="Name of data folder")
    p.add_argument("--base", type=str, default="synthetic", help="base compound id (or 'synthetic')")
    p.add_argument("--batch_size", type=int, default=100)
    p.add_argument("--nsamples", type=int, default=100)
    p.add_argument("--var", type=float, default=0.5)
    p.add_argument("--ncond", type=int, default=10)
    p.add_argument("--cond_bin", type=int, default=0, help="condition bin of the base compound")
    p.add_argument("--d", type=int, default=32)
    p.add_argument("--channels", type=int, default=4)
    a = p.parse_args()

    d, C, bs = a.d, a.channels, a.batch_size
    path = os.path.join("data", a.name, "matrices")
    vae_weights = os.path.join("saved_models", "vae", a.name, "vae_weights_" + a.name + ".best.h5")
    unet_weights = os.path.join("saved_models", "unet", a.name, "unet_weights_" + a.name + ".best.h5")
    out_dir = os.path.join("output", "results", a.base + "__v=" + str(a.var))
    for sub in ("densities", "species", "binary", "voxel_params"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)

    vae = LatticeDFCVAE(input_shape=(d, d, d, C), perceptual_model=unet_weights, cond_shape=a.ncond)
    vae._set_model(vae_weights, batch_size=bs)
    unet = AtomUnet(weights=unet_weights, input_shape=(d, d, d, C), max_batch=bs)

    if a.base == "synthetic":
        M_base = synthetic_batch(1, d, C, seed=0)[0]
    else:
        M_base = np.load(os.path.join(path, "density_matrices", a.base + ".npy")).reshape(1, d, d, d, 1)
        if C > 1:
            C_base = np.load(os.path.join(path, "coordinate_grids", a.base + ".npy")).reshape(1, d, d, d, 3)
            M_base = np.concatenate([M_base, C_base], axis=-1)
    cond = np.zeros((1, a.ncond), np.float32)
    cond[0, a.cond_bin] = 1.0

    z_mu_base, z_logvar_base, z_base = vae.encoder.predict([M_base, cond])
    for batch in range(int(a.nsamples / bs)):
        print("Batch", batch)
        z_samples = np.random.normal(z_mu_base, a.var, size=(bs, vae.latent_dim))
        # generate.py:208-225 as one device-resident chain: decoder -> U-Net -> argmax / 0.8 threshold.  The
        # reconstruction stays in HBM; back come 2 bytes per voxel, the density channel (watershed input) and
        # the coordinate channels' min / max, which is all to_lattice_params reads (generate.py:211-217).
        out = vae.decode_segment(z_samples, np.tile(cond, (bs, 1)), unet, thresh=0.8)
        dv_pred = to_voxel_params(to_lattice_params_from_minmax(out["coord_minmax"], d=d), d=d)
        for i in range(bs):
            k = batch * bs + i
            np.save(os.path.join(out_dir, "densities", "%d.npy" % k), out["density"][i])
            np.save(os.path.join(out_dir, "species", "%d.npy" % k), out["species"][i])
            np.save(os.path.join(out_dir, "binary", "%d.npy" % k), out["mask"][i])
            np.save(os.path.join(out_dir, "voxel_params", "%d.npy" % k), dv_pred[i])
    print("wrote", out_dir)
