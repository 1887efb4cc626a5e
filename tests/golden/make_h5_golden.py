"""Generates tests/golden/keras_*.h5 in THIS container with the REAL HDF5 library (libhdf5 1.10 from the
image's conda tree, driven through ctypes by tests/h5ref.py exactly the way h5py drives it for Keras 2.3.1's
save_weights / model.save).  They pin icsg3d_amd/hdf5_min.py's reader and the Keras-name mapping of
icsg3d_amd/checkpoint.py.  Tensors are tiny stand-ins (the real U-Net is 125 MB): `tiny_weights` regenerates
them from the seed, so the fixtures hold no information beyond the container format and the naming.

    python tests/golden/make_h5_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import h5ref  # noqa: E402

UNET = ["c1", "c2", "c3", "c4", "c5", "c6", "c9", "c10", "c13", "c14", "c15", "c16", "c17", "c18"]


def tiny_weights(kind, seed):
    """engine-name -> small float32 arrays with the right RANKS (kernel 5-D, dense 2-D, vectors 1-D)"""
    rng = np.random.default_rng(seed)
    r = lambda *s: rng.standard_normal(s).astype(np.float32)
    w = {}
    if kind == "unet":
        for i, n in enumerate(UNET):
            w[n + "/kernel"], w[n + "/bias"] = r(3, 3, 3, 2, 3 + i % 2), r(3 + i % 2)
            for v in ("gamma", "beta", "moving_mean", "moving_var"):
                w[n + "/" + v] = r(3 + i % 2)
        w["soft/kernel"], w["soft/bias"], w["sig/kernel"], w["sig/bias"] = r(1, 1, 1, 3, 5), r(5), r(1, 1, 1, 3, 1), r(1)
    else:
        for n in ["e0", "e1", "e2", "e3", "d0", "d1", "d2", "d3", "dout"]:
            w[n + "/kernel"], w[n + "/bias"] = r(3, 3, 3, 2, 2), r(2)
            for v in ("gamma", "beta", "moving_mean", "moving_var"):
                w[n + "/" + v] = r(2)
        w["e4/kernel"], w["e4/bias"] = r(3, 3, 3, 2, 4), r(4)
        for n in ("enc_dense", "z_mean", "z_log_var", "dec_dense"):
            w[n + "/kernel"], w[n + "/bias"] = r(6, 7), r(7)
    return w


def keras_unet_layers(w, first=1, with_weightless=True, scope=""):
    """Keras' model.layers order for AtomUnet (unet/unet.py:272-355): conv, re_lu, batch_normalization per block,
    pools / upsampling / concatenate in between (weightless: empty weight_names), heads last.  `first` shifts the
    auto-numbering, as happens when the saving process had built other layers before.  `scope`: suffix TF1 appends
    to a VARIABLE scope whose name is already taken in the graph ("conv3d_1" -> "conv3d_1_1"): the layer (group) name
    stays, the weight names carry the suffix."""
    layers = [("input_1", [])] if with_weightless else []
    for i, n in enumerate(UNET):
        k = first + i
        cv, bn = "conv3d_%d" % k, "batch_normalization_%d" % k
        cvs, bns = cv + scope, bn + scope
        layers.append((cv, [(cvs + "/kernel:0", w[n + "/kernel"]), (cvs + "/bias:0", w[n + "/bias"])]))
        if with_weightless:
            layers.append(("re_lu_%d" % k, []))
        layers.append((bn, [(bns + "/gamma:0", w[n + "/gamma"]), (bns + "/beta:0", w[n + "/beta"]),
                            (bns + "/moving_mean:0", w[n + "/moving_mean"]),
                            (bns + "/moving_variance:0", w[n + "/moving_var"])]))
        if with_weightless and n in ("c2", "c4", "c6"):
            layers.append(("max_pooling3d_%d" % (UNET.index(n) // 2 + 1), []))
        if with_weightless and n in ("c10", "c14", "c16"):
            layers += [("up_sampling3d_%d" % k, []), ("concatenate_%d" % k, [])]
    for h in ("soft", "sig"):
        layers.append((h, [(h + "/kernel:0", w[h + "/kernel"]), (h + "/bias:0", w[h + "/bias"])]))
    return layers


def keras_vae_layers(w, conv0=1, bn0=1, dense0=1, scope=""):
    """outer model layers [input, input, encoder, decoder]; nested models list trainable weights, then BN
    moving statistics (Keras 2.3.1 Network.weights).  `scope`: the suffix the conv3d_* / batch_normalization_*
    variable scopes get when the perceptual U-Net was rebuilt by load_model first (vae/lattice_vae.py:120): its
    layers keep their SAVED names conv3d_1..14, Keras' own counter still starts the encoder at conv3d_1, and TF1
    uniquifies the second scope of that name to conv3d_1_1."""
    def sc(name):
        return name + scope if name.startswith(("conv3d_", "batch_normalization_")) else name

    def kb(name, n):
        return [(sc(name) + "/kernel:0", w[n + "/kernel"]), (sc(name) + "/bias:0", w[n + "/bias"])]

    def gb(name, n):
        return [(sc(name) + "/gamma:0", w[n + "/gamma"]), (sc(name) + "/beta:0", w[n + "/beta"])]

    def mv(name, n):
        return [(sc(name) + "/moving_mean:0", w[n + "/moving_mean"]), (sc(name) + "/moving_variance:0", w[n + "/moving_var"])]

    enc, enc_s = [], []
    for i in range(4):
        enc += kb("conv3d_%d" % (conv0 + i), "e%d" % i) + gb("batch_normalization_%d" % (bn0 + i), "e%d" % i)
        enc_s += mv("batch_normalization_%d" % (bn0 + i), "e%d" % i)
    enc += kb("conv3d_%d" % (conv0 + 4), "e4") + kb("dense_%d" % dense0, "enc_dense") + kb("z_mean", "z_mean") + \
        kb("z_log_var", "z_log_var")
    dec, dec_s = kb("dense_%d" % (dense0 + 1), "dec_dense"), []
    for i in range(4):
        dec += kb("conv3d_%d" % (conv0 + 5 + i), "d%d" % i) + gb("batch_normalization_%d" % (bn0 + 4 + i), "d%d" % i)
        dec_s += mv("batch_normalization_%d" % (bn0 + 4 + i), "d%d" % i)
    dec += kb("decoder_output", "dout") + gb("batch_normalization_%d" % (bn0 + 8), "dout")
    dec_s += mv("batch_normalization_%d" % (bn0 + 8), "dout")
    return [("input_3", []), ("input_4", []), ("encoder", enc + enc_s), ("decoder", dec + dec_s)]


def main():
    h = h5ref.H5()
    wu, wv = tiny_weights("unet", 11), tiny_weights("vae", 12)
    # 1. save_weights of a U-Net built first in its process
    h.write_keras(os.path.join(HERE, "keras_unet_weights.h5"), keras_unet_layers(wu, 1))
    # 2. model.save / ModelCheckpoint (full model: weights under /model_weights), auto-numbering shifted by an
    #    earlier model in the process, chunked datasets
    h.write_keras(os.path.join(HERE, "keras_unet_fullmodel_shifted_chunked.h5"), keras_unet_layers(wu, 15),
                  full_model=True, chunked=True)
    # 3. the VAE's nested encoder / decoder models, numbering shifted by a perceptual U-Net built before
    h.write_keras(os.path.join(HERE, "keras_vae_weights.h5"), keras_vae_layers(wv, conv0=15, bn0=15, dense0=1))
    # 4./5. TF1 scope reuse: weight names with a "_1" scope suffix (see keras_vae_layers)
    h.write_keras(os.path.join(HERE, "keras_vae_weights_scoped.h5"), keras_vae_layers(wv, scope="_1"))
    h.write_keras(os.path.join(HERE, "keras_unet_weights_scoped.h5"), keras_unet_layers(wu, 1, scope="_1"))
    for f in sorted(os.listdir(HERE)):
        if f.startswith("keras_"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
