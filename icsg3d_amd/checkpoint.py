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


# ------------------------------------------------------------------------------------------ model.save: the layer graph
_GLOROT = {"class_name": "VarianceScaling", "config": {"scale": 1.0, "mode": "fan_avg", "distribution": "uniform", "seed": None}}
_ZEROS, _ONES = {"class_name": "Zeros", "config": {}}, {"class_name": "Ones", "config": {}}


def _conv3d_cfg(name, filters, k, activation="linear"):
    return {"name": name, "trainable": True, "dtype": "float32", "filters": int(filters), "kernel_size": [k, k, k],
            "strides": [1, 1, 1], "padding": "same", "data_format": "channels_last", "dilation_rate": [1, 1, 1],
            "activation": activation, "use_bias": True, "kernel_initializer": _GLOROT, "bias_initializer": _ZEROS,
            "kernel_regularizer": None, "bias_regularizer": None, "activity_regularizer": None,
            "kernel_constraint": None, "bias_constraint": None}


def _bn_cfg(name):
    return {"name": name, "trainable": True, "dtype": "float32", "axis": -1, "momentum": 0.99, "epsilon": 0.001,
            "center": True, "scale": True, "beta_initializer": _ZEROS, "gamma_initializer": _ONES,
            "moving_mean_initializer": _ZEROS, "moving_variance_initializer": _ONES, "beta_regularizer": None,
            "gamma_regularizer": None, "beta_constraint": None, "gamma_constraint": None}


def unet_model_config(input_shape=(32, 32, 32, 4), num_classes=95):
    """The functional-model JSON Keras 2.3.1 stores as the `model_config` attribute of `model.save` for
    AtomUnet.unet_3d_multiclass (/root/reference/unet/unet.py:272-355): every layer in creation order with the
    auto-generated names Keras gives them in a fresh process (conv3d_k / re_lu_k / batch_normalization_k, k = 1..14;
    max_pooling3d_1..3, up_sampling3d_1..3, concatenate_1..3; `unet_input`, `soft`, `sig` are named in the source) and the
    inbound nodes the source wires.  This is what lets `load_model(path, custom_objects)` -- how the reference's
    LatticeDFCVAE opens its perceptual U-Net (vae/lattice_vae.py:120), which then looks the taps up as re_lu_2/4/6/8
    (:100,260-261) -- rebuild the network from a file written here.  The graph is fixed and known, so the JSON is emitted
    from this table; a real Keras `load_model` could not be run in this image (Keras / TF absent)."""
    layers = []

    def add(name, cls, cfg, inbound):
        layers.append({"name": name, "class_name": cls, "config": cfg,
                       "inbound_nodes": [[[src, 0, 0, {}] for src in inbound]] if inbound else []})
        return name

    x = add("unet_input", "InputLayer", {"batch_input_shape": [None] + [int(v) for v in input_shape], "dtype": "float32",
                                         "sparse": False, "name": "unet_input"}, [])
    count = {"conv": 0, "pool": 0, "up": 0, "cat": 0}

    def block(src, filters):
        count["conv"] += 1
        k = count["conv"]
        c = add("conv3d_%d" % k, "Conv3D", _conv3d_cfg("conv3d_%d" % k, filters, 3), [src])
        r = add("re_lu_%d" % k, "ReLU", {"name": "re_lu_%d" % k, "trainable": True, "dtype": "float32", "max_value": None,
                                         "negative_slope": 0.0, "threshold": 0.0}, [c])
        return add("batch_normalization_%d" % k, "BatchNormalization", _bn_cfg("batch_normalization_%d" % k), [r])

    def pool(src):
        count["pool"] += 1
        n = "max_pooling3d_%d" % count["pool"]
        return add(n, "MaxPooling3D", {"name": n, "trainable": True, "dtype": "float32", "pool_size": [2, 2, 2],
                                       "padding": "valid", "strides": [2, 2, 2], "data_format": "channels_last"}, [src])

    def up(src):
        count["up"] += 1
        n = "up_sampling3d_%d" % count["up"]
        return add(n, "UpSampling3D", {"name": n, "trainable": True, "dtype": "float32", "size": [2, 2, 2],
                                       "data_format": "channels_last"}, [src])

    def cat(a, b):
        count["cat"] += 1
        n = "concatenate_%d" % count["cat"]
        return add(n, "Concatenate", {"name": n, "trainable": True, "dtype": "float32", "axis": -1}, [a, b])

    c1 = block(x, 32); c2 = block(c1, 64); p1 = pool(c2)
    c3 = block(p1, 64); c4 = block(c3, 128); p2 = pool(c4)
    c5 = block(p2, 128); c6 = block(c5, 256); p3 = pool(c6)
    c9 = block(p3, 512); c10 = block(c9, 512); u1 = up(c10)
    c13 = block(cat(c6, u1), 512); c14 = block(c13, 256); u3 = up(c14)
    c15 = block(cat(c4, u3), 256); c16 = block(c15, 128); u4 = up(c16)
    c17 = block(cat(c2, u4), 128); c18 = block(c17, 128)
    add("soft", "Conv3D", _conv3d_cfg("soft", num_classes, 1, "softmax"), [c18])
    add("sig", "Conv3D", _conv3d_cfg("sig", 1, 1, "sigmoid"), [c18])
    return {"class_name": "Model",
            "config": {"name": "unet", "layers": layers, "input_layers": [["unet_input", 0, 0]],
                       "output_layers": [["soft", 0, 0], ["sig", 0, 0]]},
            "keras_version": "2.3.1", "backend": "tensorflow"}


def unet_training_config(lr=1e-6):
    """`training_config` of the compiled AtomUnet (unet/unet.py:243-259): the weighted CCE is the closure `loss`
    (:211-221, the key the reference's custom_objects uses, :393-399), metrics f1_m / wr_m on `soft`."""
    return {"optimizer_config": {"class_name": "Adam",
                                 "config": {"learning_rate": float(lr), "beta_1": 0.9, "beta_2": 0.999, "decay": 0.0,
                                            "epsilon": 1e-07, "amsgrad": False}},
            "loss": {"soft": "loss", "sig": "binary_crossentropy"}, "metrics": {"soft": ["f1_m", "wr_m"]},
            "weighted_metrics": None, "sample_weight_mode": None, "loss_weights": None}


def write_keras_h5(path, layers, full_model=False, model_name="model", model_config=None, training_config=None):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    w = Hdf5Writer()
    g = w.root.create_group("model_weights") if full_model else w.root
    for node in ([w.root, g] if full_model else [w.root]):
        node.attrs["backend"] = b"tensorflow"
        node.attrs["keras_version"] = b"2.3.1"
    if full_model:
        # model_config: the layer graph as Keras serialises it (unet_model_config) where this package knows the graph;
        # otherwise a stub that is enough for Keras' load_weights (which only walks /model_weights)
        cfg = model_config if model_config is not None else {"class_name": "Model", "config": {"name": model_name},
                                                             "written_by": "icsg3d_amd"}
        w.root.attrs["model_config"] = json.dumps(cfg).encode()
        if training_config is not None:
            w.root.attrs["training_config"] = json.dumps(training_config).encode()
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


def save_weights(path, weights, kind, meta=None, full_model=False, model_config=None, training_config=None):
    """kind: "unet" | "vae".  `.npz` paths keep the archive format; everything else (the reference's
    `.hdf5` / `.h5` names) is Keras HDF5."""
    if path.endswith(".npz"):
        return save_npz(path, weights, meta)
    layers = unet_keras_layers(weights) if kind == "unet" else vae_keras_layers(weights)
    write_keras_h5(path, layers, full_model=full_model, model_name=kind, model_config=model_config,
                   training_config=training_config)


def read_model_config(path):
    """(model_config, training_config | None) of a `model.save` file, parsed."""
    with Hdf5File(path) as f:
        def js(name):
            v = f.attrs.get(name)
            if v is None:
                return None
            v = np.asarray(v).ravel()[0] if isinstance(v, np.ndarray) else v
            return json.loads(v.decode() if isinstance(v, (bytes, np.bytes_)) else str(v))
        return js("model_config"), js("training_config")


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
