"""Weight files.  The reference writes Keras HDF5: `ModelCheckpoint` / `model.save_weights` / `model.save`
(/root/reference/unet/unet.py:261-264,361-379, vae/lattice_vae.py:149-151,339-341) and publishes its U-Net
as `models/unet/*.h5` (.gitattributes:1-2).  `save_weights` / `load_weights` here speak that format
through icsg3d_amd/hdf5_min.py (pure Python, no h5py):

  Keras 2.3.1 layout   root attrs  layer_names, backend, keras_version
  (save_weights)       /<layer>    attr weight_names = [b"<layer>/kernel:0", ...]
                       /<layer>/<weight_name>          float32 datasets, Keras layouts
  (model.save)         the same tree under /model_weights (+ model_config / training_config attrs)

Engine tensor names map onto Keras' auto-generated layer names in creation order:
  U-Net  c1,c2,c3,c4,c5,c6,c9,c10,c13..c18 <-> conv3d_1..14 + batch_normalization_1..14; soft; sig
  VAE    nested models `encoder` (conv3d_1..5, batch_normalization_1..4, dense_1, z_mean, z_log_var) and
         `decoder` (dense_2, conv3d_6..9, decoder_output, batch_normalization_5..9)
Import does not trust the numbers (they depend on what else the saving process had built): conv / BN /
dense layers are matched by their ORDER among layers of the same kind, the named layers by name, and
every tensor's shape is checked.  `.npz` archives (round-1 checkpoints) are still read and written.
"""
from __future__ import annotations

import json
import os
import re

import numpy as np

from .hdf5_min import Hdf5Error, Hdf5File, Hdf5Writer, is_hdf5

UNET_ORDER = ["c1", "c2", "c3", "c4", "c5", "c6", "c9", "c10", "c13", "c14", "c15", "c16", "c17", "c18"]
_BN_VARS = ["gamma", "beta", "moving_mean", "moving_variance"]


# ------------------------------------------------------------------------------------------ npz
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


# ------------------------------------------------------------------------------------------ Keras naming
def _ours(var):
    return "moving_var" if var == "moving_variance" else var


def unet_keras_layers(weights):
    """engine name -> array  ==>  ordered [(keras_layer, [(keras_weight_name, array), ...])] as Keras 2.3.1
    lists them for AtomUnet.model (conv3d_k, batch_normalization_k interleaved; heads last)."""
    layers = []
    for i, n in enumerate(UNET_ORDER, 1):
        cv, bn = "conv3d_%d" % i, "batch_normalization_%d" % i
        layers.append((cv, [("%s/kernel:0" % cv, weights[n + "/kernel"]), ("%s/bias:0" % cv, weights[n + "/bias"])]))
        layers.append((bn, [("%s/%s:0" % (bn, v), weights[n + "/" + _ours(v)]) for v in _BN_VARS]))
    for h in ("soft", "sig"):
        layers.append((h, [("%s/kernel:0" % h, weights[h + "/kernel"]), ("%s/bias:0" % h, weights[h + "/bias"])]))
    return layers


def vae_keras_layers(weights):
    """The VAE's outer model has two weighted layers, the nested models `encoder` and `decoder`; inside each
    Keras lists all trainable weights first, then the BatchNorm moving statistics."""
    def conv(k, n):
        return [("conv3d_%d/kernel:0" % k, weights[n + "/kernel"]), ("conv3d_%d/bias:0" % k, weights[n + "/bias"])]

    def bn_t(k, n):
        return [("batch_normalization_%d/gamma:0" % k, weights[n + "/gamma"]),
                ("batch_normalization_%d/beta:0" % k, weights[n + "/beta"])]

    def bn_s(k, n):
        return [("batch_normalization_%d/moving_mean:0" % k, weights[n + "/moving_mean"]),
                ("batch_normalization_%d/moving_variance:0" % k, weights[n + "/moving_var"])]

    def dense(name, n):
        return [("%s/kernel:0" % name, weights[n + "/kernel"]), ("%s/bias:0" % name, weights[n + "/bias"])]

    enc, enc_s = [], []
    for i in range(4):
        enc += conv(i + 1, "e%d" % i) + bn_t(i + 1, "e%d" % i)
        enc_s += bn_s(i + 1, "e%d" % i)
    enc += conv(5, "e4") + dense("dense_1", "enc_dense") + dense("z_mean", "z_mean") + dense("z_log_var", "z_log_var")
    dec, dec_s = dense("dense_2", "dec_dense"), []
    for i in range(4):
        dec += conv(6 + i, "d%d" % i) + bn_t(5 + i, "d%d" % i)
        dec_s += bn_s(5 + i, "d%d" % i)
    dec += [("decoder_output/kernel:0", weights["dout/kernel"]), ("decoder_output/bias:0", weights["dout/bias"])]
    dec += bn_t(9, "dout")
    dec_s += bn_s(9, "dout")
    return [("encoder", enc + enc_s), ("decoder", dec + dec_s)]


def _by_kind(entries):
    """[(weight_name, array)] -> {kind: [ {var: array}, ... in layer order ]}, named layers under their name."""
    order, table = [], {}
    for wname, arr in entries:
        wname = wname.decode() if isinstance(wname, bytes) else str(wname)
        parts = wname.split("/")
        layer, var = parts[-2] if len(parts) >= 2 else "", parts[-1].split(":")[0]
        if layer not in table:
            table[layer] = {}
            order.append(layer)
        table[layer][var] = np.asarray(arr)
    kinds = {}
    for layer in order:
        # auto-numbered Keras layers; TF1 appends further "_<k>" suffixes to a variable scope whose name is already
        # taken in the graph (the VAE's conv3d_1 after a load_model'ed U-Net owns conv3d_1: "conv3d_1_1")
        m = re.match(r"^(conv3d|batch_normalization|dense)((?:_\d+)*)$", layer)
        if m:
            nums = tuple(int(v) for v in m.group(2).split("_")[1:])
            kinds.setdefault(m.group(1), []).append((nums, layer, table[layer]))
        else:
            kinds.setdefault("named", {})[layer] = table[layer]
    for k in ("conv3d", "batch_normalization", "dense"):
        # Keras numbers layers in creation order and the file lists them in that order; sorting by the numeric suffixes
        # (first number first) is the same order and survives a writer that lists them differently
        kinds[k] = [t for _, _, t in sorted(kinds.get(k, []), key=lambda e: e[0])]
    kinds.setdefault("named", {})
    return kinds


def _take(dst, name, table, kernel_bias=True):
    if kernel_bias:
        dst[name + "/kernel"], dst[name + "/bias"] = table["kernel"], table["bias"]
    else:
        for v in _BN_VARS:
            dst[name + "/" + _ours(v)] = table[v]


def unet_from_keras(layers):
    entries = [e for _, ws in layers for e in ws]
    k = _by_kind(entries)
    if len(k["conv3d"]) != 14 or len(k["batch_normalization"]) != 14 or not {"soft", "sig"} <= set(k["named"]):
        raise Hdf5Error("not an AtomUnet weight file: found %d conv3d, %d batch_normalization layers and %s"
                        % (len(k["conv3d"]), len(k["batch_normalization"]), sorted(k["named"])))
    out = {}
    for n, cv, bn in zip(UNET_ORDER, k["conv3d"], k["batch_normalization"]):
        _take(out, n, cv)
        _take(out, n, bn, kernel_bias=False)
    _take(out, "soft", k["named"]["soft"])
    _take(out, "sig", k["named"]["sig"])
    return out


def vae_from_keras(layers):
    groups = dict(layers)
    if "encoder" not in groups or "decoder" not in groups:
        raise Hdf5Error("not a LatticeDFCVAE weight file: layers %s" % [n for n, _ in layers])
    e, d = _by_kind(groups["encoder"]), _by_kind(groups["decoder"])
    if (len(e["conv3d"]), len(e["batch_normalization"]), len(e["dense"])) != (5, 4, 1) or \
            (len(d["conv3d"]), len(d["batch_normalization"]), len(d["dense"])) != (4, 5, 1):
        raise Hdf5Error("unexpected LatticeDFCVAE layer counts in the weight file")
    out = {}
    for i in range(4):
        _take(out, "e%d" % i, e["conv3d"][i])
        _take(out, "e%d" % i, e["batch_normalization"][i], kernel_bias=False)
    _take(out, "e4", e["conv3d"][4])
    _take(out, "enc_dense", e["dense"][0])
    _take(out, "z_mean", e["named"]["z_mean"])
    _take(out, "z_log_var", e["named"]["z_log_var"])
    _take(out, "dec_dense", d["dense"][0])
    for i in range(4):
        _take(out, "d%d" % i, d["conv3d"][i])
        _take(out, "d%d" % i, d["batch_normalization"][i], kernel_bias=False)
    _take(out, "dout", d["named"]["decoder_output"])
    _take(out, "dout", d["batch_normalization"][4], kernel_bias=False)
    return out


# ------------------------------------------------------------------------------------------ HDF5 files
def _attr_list(attrs, name):
    """Keras splits attributes above 64 KB into name0, name1, ... (saving.py save_attributes_to_hdf5_group)."""
    if name in attrs:
        return [v for v in np.atleast_1d(attrs[name])]
    out, i = [], 0
    while "%s%d" % (name, i) in attrs:
        out += [v for v in np.atleast_1d(attrs["%s%d" % (name, i)])]
        i += 1
    return out


def read_keras_h5(path):
    """-> ordered [(layer_name, [(weight_name, float32 array), ...])] for layers that hold weights; accepts
    save_weights files and full-model files (weights under /model_weights), like Keras' load_weights."""
    with Hdf5File(path) as f:
        g = f
        if "layer_names" not in f.attrs and "layer_names0" not in f.attrs and "model_weights" in f:
            g = f["model_weights"]
        names = _attr_list(g.attrs, "layer_names")
        if not names:
            raise Hdf5Error("%s: no layer_names attribute -- not a Keras weight file" % path)
        layers = []
        for ln in names:
            ln = ln.decode() if isinstance(ln, bytes) else str(ln)
            lg = g[ln]
            ws = []
            for wn in _attr_list(lg.attrs, "weight_names"):
                wn = wn.decode() if isinstance(wn, bytes) else str(wn)
                ws.append((wn, np.asarray(lg[wn].read(), dtype=np.float32)))
            if ws:
                layers.append((ln, ws))
        return layers


def write_keras_h5(path, layers, full_model=False, model_name="model"):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    w = Hdf5Writer()
    g = w.root.create_group("model_weights") if full_model else w.root
    for node in ([w.root, g] if full_model else [w.root]):
        node.attrs["backend"] = b"tensorflow"
        node.attrs["keras_version"] = b"2.3.1"
    if full_model:
        # enough for Keras' load_weights (which only walks /model_weights); load_model would need the real
        # layer graph JSON, which only Keras can write
        w.root.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {"name": model_name},
                                                   "written_by": "icsg3d_amd"}).encode()
    g.attrs["layer_names"] = np.array([ln.encode() for ln, _ in layers])
    for ln, ws in layers:
        lg = g.create_group(ln)
        lg.attrs["weight_names"] = np.array([wn.encode() for wn, _ in ws])
        for wn, arr in ws:
            lg.create_dataset(wn, np.asarray(arr, np.float32))
    w.write(path)


def _check_shapes(weights, expected_shapes, what):
    for name, shape in expected_shapes.items():
        if name not in weights:
            raise ValueError("%s: tensor %s missing from the weight file" % (what, name))
        if tuple(weights[name].shape) != tuple(shape):
            raise ValueError("%s: %s has shape %s in the file, the model needs %s (input_shape / num_classes "
                             "mismatch?)" % (what, name, tuple(weights[name].shape), tuple(shape)))


def save_weights(path, weights, kind, meta=None, full_model=False):
    """kind: "unet" | "vae".  `.npz` paths keep the archive format; everything else (the reference's
    `.hdf5` / `.h5` names) is Keras HDF5."""
    if path.endswith(".npz"):
        return save_npz(path, weights, meta)
    layers = unet_keras_layers(weights) if kind == "unet" else vae_keras_layers(weights)
    write_keras_h5(path, layers, full_model=full_model, model_name=kind)


def load_weights(path, kind, expected_shapes=None):
    """-> {engine tensor name: float32 array}.  Sniffs the container: HDF5 (Keras) or npz (round-1 files)."""
    with open(path, "rb") as f:
        magic = f.read(8)
    if is_hdf5(path):
        layers = read_keras_h5(path)
        weights = unet_from_keras(layers) if kind == "unet" else vae_from_keras(layers)
    elif magic[:2] == b"PK":
        weights, _ = load_npz(path)
    else:
        raise ValueError("%s is neither a Keras HDF5 file nor an .npz archive (magic %r)" % (path, magic))
    if expected_shapes is not None:
        _check_shapes(weights, expected_shapes, path)
    return weights
