"""Weight containers.  The reference writes Keras HDF5 (unet/unet.py:361-379, lattice_vae.py:339-341);
h5py is not available here, so the same paths hold an .npz archive of named arrays (Keras kernel
layouts; BatchNorm moving statistics included).  Keras-HDF5 import/export is a "next" row."""
from __future__ import annotations

import os

import numpy as np


def save_npz(path, weights, meta=None):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    payload = {k.replace("/", "__"): np.asarray(v) for k, v in weights.items()}
    if meta:
        for k, v in meta.items():
            payload["_meta_" + k] = np.asarray(v)
    with open(path, "wb") as f:       # file object: numpy must not append ".npz" to the reference's path
        np.savez(f, **payload)


def load_npz(path):
    with np.load(path, allow_pickle=False) as z:
        weights = {k.replace("__", "/"): z[k] for k in z.files if not k.startswith("_meta_")}
        meta = {k[len("_meta_"):]: z[k] for k in z.files if k.startswith("_meta_")}
    return weights, meta
