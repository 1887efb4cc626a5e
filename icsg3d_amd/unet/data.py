"""U-Net batch generators: same `__len__/__getitem__/on_epoch_end` contract as
/root/reference/unet/data.py:20-100 (X (B,d,d,d,C), [y, b]) over the same on-disk layout
data/<name>/matrices/{density_matrices,species_matrices,coordinate_grids}/<id>.npy.
y is handed over as uint8 class ids (the engine's native label format) unless one_hot=True."""
import os

import numpy as np

from ..synthetic import synthetic_batch


class UnetDataGenerator:
    def __init__(self, list_IDs, data_path, batch_size=2, dim=(32, 32, 32), n_channels=7, n_classes=95,
                 shuffle=False, one_hot=False):
        self.dim, self.batch_size, self.list_IDs = tuple(dim), batch_size, list(list_IDs)
        self.n_channels, self.n_classes, self.shuffle = n_channels, n_classes, shuffle
        self.data_path, self.one_hot = data_path, one_hot
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
        idx = self.indexes[index * self.batch_size:(index + 1) * self.batch_size]
        self.list_IDs_temp = [self.list_IDs[k] for k in idx]
        X = np.empty((self.batch_size, *self.dim, self.n_channels), np.float32)
        y = np.empty((self.batch_size, *self.dim), np.uint8)
        for i, ID in enumerate(self.list_IDs_temp):
            X[i] = self.create_lattice_meshgrid(ID, self.n_channels)
            y[i] = np.load(os.path.join(self.data_path, "species_matrices", ID)).reshape(self.dim)
        b = (y != 0).astype(np.float32)[..., None]
        if self.one_hot:
            y = np.eye(self.n_classes, dtype=np.float32)[y]
        return X, [y, b]

    def create_lattice_meshgrid(self, ID, channels=4):
        M = np.load(os.path.join(self.data_path, "density_matrices", ID)).reshape(*self.dim, 1)
        if channels == 1:
            return M
        p = np.load(os.path.join(self.data_path, "coordinate_grids", ID)).reshape(*self.dim, 3)
        return np.concatenate((M, p), axis=-1)


class SyntheticUnetGenerator:
    """Same contract, Gaussian-blob grids of SURVEY 8(d) instead of files (no dataset offline)."""

    def __init__(self, n_samples, batch_size=2, dim=(32, 32, 32), n_channels=1, n_classes=95, seed=0, shuffle=False):
        self.batch_size, self.dim, self.n_channels, self.n_classes = batch_size, tuple(dim), n_channels, n_classes
        self.list_IDs = ["synthetic_%06d" % i for i in range(n_samples)]
        self.seed, self.shuffle = seed, shuffle
        self.on_epoch_end()

    def __len__(self):
        return len(self.list_IDs) // self.batch_size

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def on_epoch_end(self):
        self.indexes = np.arange(len(self.list_IDs))
        if self.shuffle:
            np.random.shuffle(self.indexes)

    def __getitem__(self, index):
        X, lab, _ = synthetic_batch(self.batch_size, self.dim[0], self.n_channels, seed=self.seed + index, noise=1e-3)
        return X, [lab, (lab != 0).astype(np.float32)[..., None]]
