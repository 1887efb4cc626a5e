"""AtomUnet on the MI355X engine -- same class/method surface as /root/reference/unet/unet.py.

`AtomUnet(...)`, `.model.predict / train_on_batch / test_on_batch / load_weights / save_weights /
save / fit_generator / predict_generator`, `.train_generator`, `.predict_generator`, `.save_`, and
the module-level `f1_m`, `wr_m`, `r_m`, `p_m`, `weighted_categorical_crossentropy`, `TrainingPlot`,
`custom_objects` keep the reference's names, argument meaning and return shapes
(unet/unet.py:159-221,224-399).  Where the reference builds a Keras graph, this builds a
`UnetEngine` (icsg3d_amd/engine.py -> include/icsg3d.h -> HIP kernels).
"""
from __future__ import annotations

import os

import numpy as np

from ..checkpoint import load_weights as _load_weight_file
from ..checkpoint import save_weights as _save_weight_file
from ..dataparallel import DataParallelMixin
from ..engine import UnetEngine
from ..synthetic import bn_state_defaults, glorot_params, unet_param_shapes
from .get_weights import get_weights

K_EPSILON = 1e-7


# ------------------------------------------------------------------ metrics / loss (numpy forms)
def r_m(y_true, y_pred):
    tp = np.sum(np.round(np.clip(y_true * y_pred, 0, 1)))
    possible = np.sum(np.round(np.clip(y_true, 0, 1)))
    return tp / (possible + K_EPSILON)


def wr_m(y_true, y_pred):
    """Weighted recall: class 0 removed (unet/unet.py:169-179; 95 classes hard-coded there)."""
    w = np.ones(y_true.shape[-1])
    w[0] = 0.0
    tp = np.sum(np.round(np.clip(w * y_true * y_pred, 0, 1)))
    possible = np.sum(np.round(np.clip(w * y_true, 0, 1)))
    return tp / (possible + K_EPSILON)


def p_m(y_true, y_pred):
    tp = np.sum(np.round(np.clip(y_true * y_pred, 0, 1)))
    predicted = np.sum(np.round(np.clip(y_pred, 0, 1)))
    return tp / (predicted + K_EPSILON)


def f1_m(y_true, y_pred):
    precision, recall = p_m(y_true, y_pred), r_m(y_true, y_pred)
    return 2 * ((precision * recall) / (precision + recall + K_EPSILON))


def weighted_categorical_crossentropy(weights):
    """Returns loss(y_true, y_pred) -> (B,) exactly as unet/unet.py:196-221 (numpy evaluation)."""
    weights = np.asarray(weights, np.float64)

    def loss(y_true, y_pred):
        y_pred = y_pred / np.sum(y_pred, axis=-1, keepdims=True)
        y_pred = np.clip(y_pred, K_EPSILON, 1 - K_EPSILON)
        return np.mean(-np.sum(y_true * np.log(y_pred) * weights, axis=-1), axis=(1, 2, 3))

    return loss


def _to_labels(y, num_classes):
    """Accept the reference generator's one-hot (B,d,d,d,C) (unet/data.py:89) or uint8 class ids."""
    y = np.asarray(y)
    if y.ndim == 5 and y.shape[-1] == num_classes:
        return np.argmax(y, axis=-1).astype(np.uint8)
    if y.ndim == 5 and y.shape[-1] == 1:
        y = y[..., 0]
    return y.astype(np.uint8)


class TrainingPlot:
    """Loss-curve callback (unet/unet.py:39-157).  Records the per-epoch logs and writes
    unet_loss.png when matplotlib is importable; the segmentation slice plots are plotting-only
    and out of scope."""

    def __init__(self, val_gen, sdir):
        self.val_gen, self.sdir = val_gen, sdir
        self.min_val_loss = np.inf
        self.losses, self.val_losses, self.logs = [], [], []

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.logs.append(logs)
        self.losses.append(logs.get("loss"))
        self.val_losses.append(logs.get("val_loss"))
        if self.val_losses[-1] is not None and self.val_losses[-1] < self.min_val_loss:
            self.min_val_loss = self.val_losses[-1]
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            n = np.arange(len(self.losses))
            plt.figure()
            plt.plot(n, self.losses, label="train_loss")
            plt.plot(n, self.val_losses, label="val_loss")
            plt.title("Training Loss [Epoch {}]".format(epoch))
            plt.xlabel("Epoch #"); plt.ylabel("Loss"); plt.legend()
            os.makedirs(self.sdir, exist_ok=True)
            plt.savefig(os.path.join(self.sdir, "unet_loss.png"))
            plt.close()
        except Exception:
            pass


class _UnetModel:
    """Stands where the reference's `keras.models.Model` stands (`AtomUnet.model`)."""

    metrics_names = ["loss", "soft_loss", "sig_loss", "soft_f1_m", "soft_wr_m"]   # unet/unet.py:249-250

    def __init__(self, owner):
        self._o = owner

    # -- inference (generate.py:220, eval.py:166, view_results.py:136).  Keras' predict streams the input in
    # batches of `batch_size` (default 32); the engine does the same in chunks of its max_batch and is never
    # re-created (and never loses optimizer state) because of an inference call.
    def predict(self, X, batch_size=None, verbose=0):
        X = np.asarray(X)
        return list(self._o._engine(min(len(X), batch_size or 32)).predict(X))

    def predict_labels(self, X, thresh=0.8, batch_size=None):
        """Fused generate.py:220-225 tail: uint8 argmax species and (sig >= thresh) mask."""
        X = np.asarray(X)
        return self._o._engine(min(len(X), batch_size or 32)).predict_labels(X, thresh)

    def predict_generator(self, gen):
        soft, sig = [], []
        for i in range(len(gen)):
            X = gen[i][0]
            s, g = self.predict(X)
            soft.append(s); sig.append(g)
        return [np.concatenate(soft), np.concatenate(sig)]

    # -- training steps
    def train_on_batch(self, X, y):
        labels = _to_labels(y[0] if isinstance(y, (list, tuple)) else y, self._o.num_classes)
        with self._o._dp_watch("AtomUnet train_on_batch"):
            return [float(v) for v in self._o._engine(len(X), grow=True).train_step(X, labels)]

    def test_on_batch(self, X, y):
        labels = _to_labels(y[0] if isinstance(y, (list, tuple)) else y, self._o.num_classes)
        with self._o._dp_watch("AtomUnet test_on_batch"):
            return [float(v) for v in self._o._engine(len(X), grow=True).test_step(X, labels)]

    def fit_generator(self, generator, validation_data=None, epochs=1, callbacks=None, workers=4,
                      use_multiprocessing=False, verbose=1, max_queue_size=10):
        """Epoch loop of unet/unet.py:370-377: mean of per-batch metrics as the epoch log.  `workers` loader
        threads fill a bounded queue with ready batches (labels already reduced to uint8 class ids) while
        the GPU runs the previous step -- Keras' OrderedEnqueuer with use_multiprocessing=False."""
        from ..prefetch import prefetched
        nc = self._o.num_classes

        def prep(item):
            X, y = item
            return np.asarray(X, np.float32), _to_labels(y[0] if isinstance(y, (list, tuple)) else y, nc)

        history = []
        for e in range(epochs):
            tm = np.mean([[float(v) for v in self._o._engine(len(X), grow=True).train_step(X, lab)]
                          for X, lab in prefetched(generator, workers, max_queue_size, prep)], axis=0)
            logs = dict(zip(self.metrics_names, tm))
            if validation_data is not None and len(validation_data):
                vm = np.mean([[float(v) for v in self._o._engine(len(X), grow=True).test_step(X, lab)]
                              for X, lab in prefetched(validation_data, workers, max_queue_size, prep)], axis=0)
                logs.update({"val_" + k: v for k, v in zip(self.metrics_names, vm)})
            if verbose:
                print("Epoch %d/%d  " % (e + 1, epochs) + "  ".join("%s: %.4f" % kv for kv in logs.items()))
            for cb in callbacks or []:
                cb.on_epoch_end(e, logs)
            for g in (generator, validation_data):
                if g is not None and hasattr(g, "on_epoch_end"):
                    g.on_epoch_end()
            history.append(logs)
        return history

    # -- checkpoints (unet/unet.py:261-264,378-379,387-390)
    def get_weights_dict(self):
        return self._o._get_weights()

    def load_weights(self, path):
        """Keras HDF5 (save_weights or full-model files, e.g. the published models/unet/*.h5) or a round-1
        .npz; tensor shapes are validated against this model's input_shape / num_classes."""
        o = self._o
        exp = dict(unet_param_shapes(o.input_shape[-1], o.num_classes))
        self._o._set_weights(_load_weight_file(path, "unet", expected_shapes=exp))

    def save_weights(self, path):
        _save_weight_file(path, self._o._get_weights(), "unet")

    def save(self, path):
        """model.save(.h5) (unet/unet.py:379,389): the weight tree under /model_weights plus the `model_config` /
        `training_config` attributes Keras 2.3.1 writes for this graph (checkpoint.unet_model_config: every layer of
        unet_3d_multiclass with its auto-generated name and inbound nodes), which is what `load_model(path,
        custom_objects)` -- how the reference's LatticeDFCVAE opens its perceptual U-Net (vae/lattice_vae.py:120) -- needs
        to rebuild the network and find re_lu_2/4/6/8.  Not verifiable here: Keras itself is absent from the image."""
        from ..checkpoint import unet_model_config, unet_training_config
        o = self._o
        _save_weight_file(path, o._get_weights(), "unet", full_model=True,
                          model_config=unet_model_config(o.input_shape, o.num_classes),
                          training_config=unet_training_config(getattr(o, "lr", 1e-6)))


class _BestCheckpoint:
    """ModelCheckpoint(filepath, monitor="val_loss", save_best_only=True, mode="min") (/root/reference/unet/unet.py:361-367).
    The reference does not set save_weights_only, so Keras writes the FULL model (model_config + training_config +
    model_weights/...) into the ".best.hdf5" path; so does this (model.save), and load_weights reads either layout, as
    Keras' does (unet.py:378)."""

    def __init__(self, model, filepath):
        self.model, self.filepath, self.best = model, filepath, np.inf

    def on_epoch_end(self, epoch, logs):
        v = logs.get("val_loss", logs.get("loss"))
        if v is not None and v < self.best:
            if self.model._o._dp_is_writer():
                print("Epoch %05d: val_loss improved from %.5f to %.5f, saving model to %s"
                      % (epoch + 1, self.best, v, self.filepath))
            self.best = v
            if self.model._o._dp_is_writer():        # data parallel: same decision on every rank, rank 0 writes
                self.model.save(self.filepath)


class AtomUnet(DataParallelMixin):
    """U-Net for semantic segmentation of electron-density maps (unet/unet.py:224-391).

    num_classes, class_weights, weights, input_shape, lr: as the reference.  class_weights is
    stored and, as in the reference, not used by the loss: the compiled loss weight is
    the scalar float(num_classes).  Extra keyword `pool_ties`: "tf_cpu" (TensorFlow-CPU
    MaxPool3DGrad tie rule, default) or "first"; `bce_from_logits`: False (default) = "binary_crossentropy" on clipped
    probabilities, True = TF 2.1's short-circuit to sigmoid_cross_entropy_with_logits for a Sigmoid output (SURVEY
    App. B: which of the two Keras 2.3.1 / TF 2.1 takes cannot be checked offline; they differ only at saturation).
    """

    def __init__(self, num_classes=95, class_weights=None, weights=None, input_shape=(32, 32, 32, 4),
                 lr=1e-6, pool_ties="tf_cpu", max_batch=None, bce_from_logits=False):
        self.class_weights = class_weights
        self.bce_from_logits = bool(bce_from_logits)
        self.input_shape = tuple(input_shape)
        self.lr = lr
        self.num_classes = num_classes
        self.pool_ties = pool_ties
        self._eng = None
        self._max_batch = max_batch
        shapes = unet_param_shapes(self.input_shape[-1], num_classes)
        self._host_weights = glorot_params(shapes, seed=int(np.random.randint(0, 2 ** 31 - 1)))
        self._host_weights.update(bn_state_defaults(shapes))
        self.model = _UnetModel(self)
        self.metrics = {"soft": [f1_m, wr_m]}
        self.metric_names = ["Loss", "lsoft", "lsig", "f1", "wr"]
        if weights and os.path.exists(weights):
            self.model.load_weights(weights)
            print("loaded weights")
            self.filepath = weights
        elif weights:
            self.filepath = weights
        else:
            self.filepath = "./saved_models/unet_%d_channel_weights.best.hdf5" % self.input_shape[-1]

    # ---- engine management: Keras models take any batch size; the engine is sized on first use.  Inference
    # never re-creates it (the engine streams larger inputs in chunks of max_batch); a TRAINING batch larger
    # than max_batch does, and then weights, BN moving statistics AND the Adam state (moments, step count)
    # move to the new engine -- keras.optimizers.Adam keeps its state for the life of the model.
    def _engine(self, batch, grow=False):
        batch = max(int(batch), 1)
        if self._eng is None:
            carry = dict(self._host_weights)
            self._eng = UnetEngine(in_channels=self.input_shape[-1], num_classes=self.num_classes,
                                   d=self.input_shape[0], max_batch=max(batch, self._max_batch or 0), lr=self.lr,
                                   pool_ties=self.pool_ties, bce_from_logits=self.bce_from_logits)
            self._eng.set_weights(carry)
            self._dp_attach(self._eng)
        elif grow and batch > self._eng.max_batch:
            carry, opt = self._eng.get_weights(), self._eng.get_optimizer_state()
            self._eng.close()
            self._eng = UnetEngine(in_channels=self.input_shape[-1], num_classes=self.num_classes,
                                   d=self.input_shape[0], max_batch=batch, lr=self.lr, pool_ties=self.pool_ties,
                                   bce_from_logits=self.bce_from_logits)
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
    def train_generator(self, train_gen, val_gen, epochs=100, output_dir="output/unet/"):
        print("Training...")
        callbacks = [_BestCheckpoint(self.model, self.filepath)]
        if self._dp_is_writer():
            callbacks.append(TrainingPlot(val_gen, output_dir))
        self.model.fit_generator(generator=train_gen, validation_data=val_gen, use_multiprocessing=False,
                                 workers=4, epochs=epochs, callbacks=callbacks)
        self._dp_barrier()                               # rank 0 has finished writing the best checkpoint
        if os.path.exists(self.filepath):
            self.model.load_weights(self.filepath)
        if self._dp_is_writer():
            self.model.save(os.path.splitext(self.filepath)[0] + ".h5")
            print("Model saved")
        self._dp_barrier()

    def predict_generator(self, test_gen):
        return self.model.predict_generator(test_gen)

    def save_(self, weights, model="saved_models/unet.h5"):
        self.model.load_weights(weights)
        self.model.save(model)


# Weights for the perceptual model (unet/unet.py:393-399)
class_weights = get_weights()
custom_objects = {"loss": weighted_categorical_crossentropy(class_weights), "f1_m": f1_m, "wr_m": wr_m}
