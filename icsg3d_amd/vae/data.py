"""VAE batch generators: contract of /root/reference/vae/data.py:21-100 -- (M (B,d,d,d,C), cond
one-hot (B,n_bins)), condition bins from pandas.qcut of a CSV property column."""
import os
import re

import numpy as np

from ..synthetic import synthetic_batch


class VAEDataGenerator:
    def __init__(self, list_IDs, data_path, batch_size=2, dim=(32, 32, 32), n_channels=4, n_classes=95,
                 shuffle=False, property_csv="property.csv", n_bins=10, target="formation_energy_per_atom",
                 return_S=False):
        import pandas as pd
        self.dim, self.batch_size, self.list_IDs = tuple(dim), batch_size, list(list_IDs)
        self.n_channels, self.n_classes, self.shuffle, self.data_path = n_channels, n_classes, shuffle, data_path
        self.property_df = pd.read_csv(property_csv)
        self.n_bins = n_bins
        self.n_atoms = self.property_df["nsites"].max() + 1 if "nsites" in self.property_df else None
        self.property_df["bin"] = pd.qcut(self.property_df[target], n_bins, np.arange(n_bins)).astype(int)
        self._bin = dict(zip(self.property_df["task_id"], self.property_df["bin"]))
        self.return_S = return_S
        self.on_epoch_end()

    def __len__(self):
        return int(np.floor(len(self.list_IDs) / self.batch_size))

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def on_epoch_end(self):
        self.indexes = np.arange(len(self.list_IDs))
        if self.shuffle:
            np.random.shuffle(self.indexes)

    def __getitem__(self, index):
        self.indexes_temp = self.indexes[index * self.batch_size:(index + 1) * self.batch_size]
        self.list_IDs_temp = [self.list_IDs[k] for k in self.indexes_temp]
        M = np.empty((self.batch_size, *self.dim, self.n_channels), np.float32)
        cond = np.zeros((self.batch_size, self.n_bins), np.float32)
        S = np.empty((self.batch_size, *self.dim), np.uint8) if self.return_S else None
        for i, ID in enumerate(self.list_IDs_temp):
            M[i] = self.create_lattice_meshgrid(ID, self.n_channels)
            cond[i] = self.property_to_categorical(ID)
            if self.return_S:
                S[i] = np.load(os.path.join(self.data_path, "species_matrices", ID)).reshape(self.dim)
        if self.return_S:
            return M, [cond, S, (S != 0).astype(np.float32)[..., None]]
        return M, cond

    def property_to_categorical(self, ID):
        cif_id = re.split(r"_|\.", ID)[0]
        out = np.zeros(self.n_bins, np.float32)
        out[int(self._bin[cif_id])] = 1.0
        return out

    def create_lattice_meshgrid(self, ID, channels=4):
        M = np.load(os.path.join(self.data_path, "density_matrices", ID)).reshape(*self.dim, 1)
        if channels == 1:
            return M
        p = np.load(os.path.join(self.data_path, "coordinate_grids", ID)).reshape(*self.dim, 3)
        return np.concatenate((M, p), axis=-1)


class SyntheticVAEGenerator:
    def __init__(self, n_samples, batch_size=2, dim=(32, 32, 32), n_channels=1, n_bins=10, seed=0, shuffle=False):
        self.batch_size, self.dim, self.n_channels, self.n_bins = batch_size, tuple(dim), n_channels, n_bins
        self.list_IDs = ["synthetic_%06d" % i for i in range(n_samples)]
        self.seed, self.shuffle = seed, shuffle
        self.on_epoch_end()

    def __len__(self):
        return len(self.list_IDs) // self.batch_size

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def on_epoch_end(self):
        self.indexes = np.arange(len(self.list_IDs))

    def __getitem__(self, index):
        self.list_IDs_temp = self.list_IDs[index * self.batch_size:(index + 1) * self.batch_size]
        X, _, cond = synthetic_batch(self.batch_size, self.dim[0], self.n_channels, seed=self.seed + index, noise=1e-3)
        return X, cond[:, :self.n_bins]
