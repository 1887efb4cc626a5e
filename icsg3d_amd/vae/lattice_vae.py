"""LatticeDFCVAE on the MI355X engine -- same class/method surface as
/root/reference/vae/lattice_vae.py (Conditional deep-feature-consistent VAE).

`LatticeDFCVAE(...)`, `._set_model`, `.train`, `.sample_vae`, `.save_`, `.encoder.predict`,
`.decoder.predict`, `.model.predict / train_on_batch / test_on_batch / load_weights / save_weights /
save` keep the reference's names, argument meaning and return shapes (lattice_vae.py:69-357).
`sampling` is the reparameterisation of lattice_vae.py:53-66; the N(0,1) draw is made on the host
(np.random) and handed to the engine, so every `predict` is stochastic exactly like the reference
 unless `eps=` is passed.
"""
from __future__ import annotations

import os
import time

import numpy as np

from ..checkpoint import load_weights as _load_weight_file
from ..checkpoint import save_weights as _save_weight_file
from ..dataparallel import DataParallelMixin
from ..engine import UnetEngine, VaeEngine
from ..synthetic import bn_state_defaults, glorot_params, unet_param_shapes, vae_param_shapes
from ..unet.unet import custom_objects


class Adam:
    """Stand-in for keras.optimizers.Adam in constructor signatures (only `lr` is consumed)."""

    def __init__(self, lr=0.001, **_):
        self.lr = lr


def sampling(args, epsilon=None):
    z_mean, z_log_var = args
    if epsilon is None:
        epsilon = np.random.normal(size=np.shape(z_mean))
    return z_mean + np.exp(0.5 * z_log_var) * epsilon


def _to_categorical(y, num_classes):
    y = np.asarray(y, dtype=int).ravel()
    out = np.zeros((len(y), num_classes), np.float32)
    out[np.arange(len(y)), y] = 1.0
    return out


class _Encoder:
    def __init__(self, o):
        self._o = o

    def predict(self, inputs, eps=None, batch_size=None):
        M, cond = inputs
        M = np.asarray(M)
        if eps is None:
            eps = np.random.normal(size=(len(M), self._o.latent_dim))
        return self._o._engine(min(len(M), batch_size or 32)).encode(M, cond, eps)     # (z_mean, z_log_var, z)


class _Decoder:
    def __init__(self, o):
        self._o = o

    def predict(self, inputs, batch_size=None):
        z, cond = inputs
        return self._o._engine(min(len(z), batch_size or 32)).decode(z, cond)


class _VaeModel:
    """Stands where `self.model` (encoder o decoder, lattice_vae.py:131-145) stands."""

    def __init__(self, o):
        self._o = o

    def predict(self, inputs, eps=None):
        M, cond = inputs
        _, _, z = self._o.encoder.predict([M, cond], eps=eps)    # the SAMPLED z feeds the decoder (F8)
        return self._o.decoder.predict([z, cond])

    def _eps(self, n, eps):
        return np.random.normal(size=(n, self._o.latent_dim)) if eps is None else eps

    def train_on_batch(self, inputs, target=None, eps=None):
        M, cond = inputs
        with self._o._dp_watch("LatticeDFCVAE train_on_batch"):
            return [float(v) for v in self._o._engine(len(M), grow=True).train_step(M, cond, self._eps(len(M), eps))]

    def test_on_batch(self, inputs, target=None, eps=None):
        M, cond = inputs
        with self._o._dp_watch("LatticeDFCVAE test_on_batch"):
            return [float(v) for v in self._o._engine(len(M), grow=True).test_step(M, cond, self._eps(len(M), eps))]

    def load_weights(self, path):
        o = self._o
        exp = dict(vae_param_shapes(o.channels, o.cond_shape, tuple(o.filters), o.latent_dim, o.input_shape[0]))
        o._set_weights(_load_weight_file(path, "vae", expected_shapes=exp))

    def save_weights(self, path):
        _save_weight_file(path, self._o._get_weights(), "vae")

    def save(self, path):
        _save_weight_file(path, self._o._get_weights(), "vae", full_model=True)


class LatticeDFCVAE(DataParallelMixin):
    """Conditional VAE with a deep-feature-consistent (perceptual U-Net) loss; parameters as
    lattice_vae.py:89-105.  `perceptual_model` is a U-Net checkpoint written by
    `AtomUnet.model.save` (or an AtomUnet / dict of weights); its weights stay frozen and its
    BatchNorm runs on batch statistics inside train_on_batch."""

    def __init__(self, input_shape=(32, 32, 32, 4), kernel_size=(3, 3, 3), pool_size=(2, 2, 2),
                 filters=(16, 32, 64, 128), latent_dim=256, beta=3e-4, alpha=0.5, optimizer=None,
                 perceptual_model="saved_models/unet.h5",
                 pm_layers=("re_lu_2", "re_lu_4", "re_lu_6", "re_lu_8"),
                 pm_layer_weights=(1.0, 1.0, 1.0, 1.0), cond_shape=10, custom_objects=custom_objects,
                 output_dir="output", pool_ties="tf_cpu"):
        if tuple(kernel_size) != (3, 3, 3) or tuple(pool_size) != (2, 2, 2) or len(filters) != 4:
            raise ValueError("the engine implements the reference configuration: 3x3x3 kernels, 2x2x2 pools, 4 filter stages")
        if list(pm_layers) != ["re_lu_2", "re_lu_4", "re_lu_6", "re_lu_8"]:
            raise ValueError("perceptual taps are fixed to re_lu_2/4/6/8 (= ReLU outputs of c2,c4,c6,c10)")
        self.input_shape = tuple(input_shape)
        self.kernel_size, self.pool_size, self.filters = kernel_size, pool_size, list(filters)
        self.latent_dim = latent_dim
        self.channels = self.input_shape[-1]
        self.optimizer = optimizer if optimizer is not None else Adam(5e-4)
        self.lr = float(getattr(self.optimizer, "lr", self.optimizer))
        self.beta, self.alpha = beta, alpha
        self.batch_size = None
        self.cond_shape = cond_shape
        self.losses = []
        self.sdir = output_dir
        self.pool_ties = pool_ties
        self.pm_layers, self.pm_layer_weights = list(pm_layers), list(pm_layer_weights)
        self.metric_names = ["Loss", "PM", "MSE", "KLD"]
        # perceptual U-Net weights (load_model(perceptual_model, custom_objects), lattice_vae.py:120)
        if isinstance(perceptual_model, dict):
            self._pm_weights = perceptual_model
        elif hasattr(perceptual_model, "_get_weights"):
            self._pm_weights = perceptual_model._get_weights()
        else:   # load_model(perceptual_model, custom_objects): the U-Net's .h5 (Keras HDF5, or a round-1 .npz)
            self._pm_weights = _load_weight_file(perceptual_model, "unet")
        self._eng = self._pm_eng = None
        self._host_weights = None
        self.encoder = self.decoder = self.model = None

    # ---- engine management: sized on first use; inference streams in chunks of max_batch and never re-creates
    # the engines; a larger TRAINING batch does, carrying weights, BN statistics and the Adam state across.
    def _build(self, mb):
        d, C = self.input_shape[0], self.channels
        self._pm_eng = UnetEngine(in_channels=C, d=d, max_batch=mb, pool_ties=self.pool_ties,
                                  num_classes=int(self._pm_weights["soft/bias"].shape[0]))
        self._pm_eng.set_weights(self._pm_weights)
        self._eng = VaeEngine(self._pm_eng, in_channels=C, cond_shape=self.cond_shape,
                              latent_dim=self.latent_dim, filters=self.filters, d=d, max_batch=mb,
                              lr=self.lr, alpha=self.alpha, beta=self.beta,
                              pm_layer_weights=self.pm_layer_weights)

    def _engine(self, batch, grow=False):
        batch = max(int(batch), 1)
        if self._eng is None:
            carry = dict(self._host_weights)
            self._build(max(batch, self.batch_size or 0))
            self._eng.set_weights(carry)
            self._dp_attach(self._eng)
        elif grow and batch > self._eng.max_batch:
            carry, opt = self._eng.get_weights(), self._eng.get_optimizer_state()
            self._eng.close(); self._pm_eng.close()
            self._build(batch)
            self._eng.set_weights(carry)
            self._eng.set_optimizer_state(*opt)
            self._dp_attach(self._eng)
        return self._eng

    def _get_weights(self):
        return self._eng.get_weights() if self._eng is not None else dict(self._host_weights)

    def _set_weights(self, w):
        if self._eng is not None:
            self._eng.set_weights(w)
        else:
            self._host_weights.update({k: np.asarray(v, np.float32) for k, v in w.items()})

    # ---- reference methods
    def _set_model(self, weights=None, batch_size=20):
        shapes = vae_param_shapes(self.channels, self.cond_shape, tuple(self.filters), self.latent_dim,
                                  self.input_shape[0])
        self._host_weights = glorot_params(shapes, seed=int(np.random.randint(0, 2 ** 31 - 1)))
        self._host_weights.update(bn_state_defaults(shapes))
        self.encoder, self.decoder, self.model = _Encoder(self), _Decoder(self), _VaeModel(self)
        self.batch_size = batch_size
        if weights and os.path.exists(weights):
            self.model.load_weights(weights)
            self.filepath = weights
        elif weights:
            self.filepath = weights
        else:
            self.filepath = "saved_models/lattice_dfc_vae_weights.best.hdf5"

    def train(self, train_gen, val_gen, epochs, weights=None):
        best_loss = np.inf
        self.train_batch_size, self.val_batch_size = train_gen.batch_size, val_gen.batch_size
        self.batch_size = self.train_batch_size
        self.num_epochs = epochs
        train_steps = int(len(train_gen.list_IDs) / self.train_batch_size)
        val_steps = int(len(val_gen.list_IDs) / self.val_batch_size)
        print("Data size %d,    batch_size %d    steps per epoch %d"
              % (len(train_gen.list_IDs), self.train_batch_size, train_steps))
        self._set_model(weights, batch_size=self.train_batch_size)
        self.losses = np.empty((self.num_epochs, 2))
        for e in range(self.num_epochs):
            print("Epoch %s:" % e)
            t0 = time.time()
            # each batch is read from disk ONCE (the target is the input) and one batch ahead of the GPU step
            tm = np.mean([self.model.train_on_batch([M, cond], M) for M, cond in self._batches(train_gen, train_steps)], axis=0)
            vm = np.mean([self.model.test_on_batch([M, cond], M) for M, cond in self._batches(val_gen, val_steps)], axis=0) \
                if val_steps else tm
            s = "Time: %.3f s   " % (time.time() - t0)
            s += "".join("Train %s: %.3f    " % (n, v) for n, v in zip(self.metric_names, tm))
            s += "".join("Val %s: %.3f    " % (n, v) for n, v in zip(self.metric_names, vm))
            print(s)
            self.losses[e] = [tm[0], vm[0]]
            if vm[0] < best_loss:
                best_loss = vm[0]
                if self._dp_is_writer():                 # data parallel: vm is rank-reduced, rank 0 writes
                    print("Saving Model")
                    self.model.save_weights(self.filepath)
        self._dp_barrier()
        if os.path.exists(self.filepath):
            self.model.load_weights(self.filepath)
        if self._dp_is_writer():
            self.model.save(os.path.splitext(self.filepath)[0] + ".h5")
            print("Model saved")
        self._dp_barrier()

    @staticmethod
    def _batches(gen, steps):
        from ..prefetch import prefetched

        class _First:
            def __len__(self):
                return steps

            def __getitem__(self, i):
                return gen[i]
        return prefetched(_First(), workers=1, max_queue_size=2, prepare=lambda it: (it[0], it[1]))

    def decode_segment(self, z, cond, unet, thresh=0.8):
        """generate.py:204-225 as ONE device-resident chain: decoder.predict -> unet.model.predict -> argmax /
        (sig >= thresh).  `unet` is an AtomUnet.  Returns dict(species, mask uint8 (B,d,d,d); density float32
        (B,d,d,d); coord_minmax (B,3,2)) -- see icsg3d_amd.utils.to_lattice_params_from_minmax."""
        z = np.asarray(z)
        eng = self._engine(min(len(z), self.batch_size or 32))
        return eng.decode_to_labels(unet._engine(min(len(z), eng.max_batch)), z, cond, thresh)

    def decode_segment_atoms(self, z, cond, unet, thresh=0.8, min_voxels=3, max_atoms=512, want_regions=False,
                             split=True, max_iters=5):
        """decode_segment continued through `watershed_clustering` (generate.py:228-236 -> watershed.py:40-203):
        connected components, size filter, majority vote and centroids on the device for the whole batch; with
        split=True the convexity test and the recursive marker watershed of the non-convex components follow
        (icsg3d_amd.watershed.refine_atoms; max_iters = the reference's --clus_iters).  Adds n_components, n_atoms,
        stats, atoms = [(species list, mean list)], failed (B,), split (B,)."""
        z = np.asarray(z)
        eng = self._engine(min(len(z), self.batch_size or 32))
        out = eng.decode_to_atoms(unet._engine(min(len(z), eng.max_batch)), z, cond, thresh, min_voxels, max_atoms,
                                  want_regions=want_regions or split)
        if split:
            from ..watershed import refine_atoms
            refine_atoms(out, max_iters=max_iters, num_species=unet.num_classes if hasattr(unet, "num_classes") else 95)
        return out

    def save_(self, weights, model="saved_models/vae.h5"):
        self.model.load_weights(weights)
        self.model.save(model)

    def sample_vae(self, n_samples, cond=None, var=1.0):
        if cond is None:
            cond = np.random.randint(low=0, high=self.cond_shape, size=n_samples)
        cond_tensor = _to_categorical(cond, self.cond_shape)
        if len(cond_tensor) != n_samples:      # scalar cond: the reference tiles it (lattice_vae.py:349-350)
            cond_tensor = np.tile(cond_tensor, (n_samples, 1))[:n_samples]
        z_sample = np.random.normal(0, var, size=(n_samples, self.latent_dim))
        return z_sample, self.decoder.predict([z_sample, cond_tensor])
