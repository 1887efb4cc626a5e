"""Class weights for the U-Net loss (mirror of /root/reference/unet/get_weights.py:19-33): inverse
frequency of each species over the un-rotated training matrices; np.ones(n) when no path is given
(the form `custom_objects` uses).  Note SURVEY F11: the reference's training loss never uses them."""
import os

import numpy as np


def get_weights(path="", training_ids=(), n_classes=95):
    if not path:
        return np.ones(n_classes)
    counts = np.zeros(n_classes)
    ids = set(training_ids)
    folder = os.path.join(path, "species_matrices")
    for fname in os.listdir(folder):
        if not fname.endswith(".npy") or "_rot_" in fname or fname not in ids:
            continue
        values, n = np.unique(np.load(os.path.join(folder, fname)), return_counts=True)
        for v, c in zip(values, n):
            counts[int(v)] += c
    with np.errstate(divide="ignore"):
        w = counts.sum() / counts
    w[np.isinf(w)] = 0
    return w
