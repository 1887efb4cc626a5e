"""Keras-HDF5 weight files without h5py: icsg3d_amd/hdf5_min.py + checkpoint.py.

Pinned against the REAL HDF5 library: tests/golden/keras_*.h5 were written by libhdf5 (through ctypes, the way
h5py does for Keras 2.3.1: tests/golden/make_h5_golden.py) and are read here with the pure-Python reader; when the
image has libhdf5 / h5ls, files written by the pure-Python writer are read back with the real library too."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

from icsg3d_amd import checkpoint as K
from icsg3d_amd import hdf5_min as H
import make_h5_golden as G


def _eq(a, b):
    assert set(a) == set(b)
    for k in a:
        assert a[k].dtype == np.float32 and np.array_equal(a[k], b[k]), k


def test_reads_libhdf5_written_unet_weights():
    w = G.tiny_weights("unet", 11)
    _eq(K.load_weights(os.path.join(HERE, "golden", "keras_unet_weights.h5"), "unet"), w)


def test_reads_full_model_with_shifted_numbering_and_chunked_datasets():
    """ModelCheckpoint without save_weights_only writes a full model into the "weights" path
    (unet/unet.py:361-367); layer numbers depend on what the process built before: matched by order."""
    w = G.tiny_weights("unet", 11)
    path = os.path.join(HERE, "golden", "keras_unet_fullmodel_shifted_chunked.h5")
    with H.Hdf5File(path) as f:
        assert "model_weights" in f and f.attrs["keras_version"] == b"2.3.1"
        assert f["model_weights"].attrs["layer_names"][1] == b"conv3d_15"
    _eq(K.load_weights(path, "unet"), w)


def test_reads_nested_vae_models():
    w = G.tiny_weights("vae", 12)
    _eq(K.load_weights(os.path.join(HERE, "golden", "keras_vae_weights.h5"), "vae"), w)


@pytest.mark.parametrize("kind,seed", [("unet", 3), ("vae", 4)])
def test_writer_reader_round_trip_bit_equal(tmp_path, kind, seed):
    w = G.tiny_weights(kind, seed)
    for full in (False, True):
        p = str(tmp_path / ("%s_%d.h5" % (kind, full)))
        K.save_weights(p, w, kind, full_model=full)
        assert H.is_hdf5(p)
        _eq(K.load_weights(p, kind), w)


def test_tf1_scope_suffixed_weight_names():
    """a VAE saved by a process that had load_model'ed the perceptual U-Net first carries variable scopes
    conv3d_1_1 / batch_normalization_1_1 ... (libhdf5-written fixtures; the U-Net case for symmetry)."""
    _eq(K.load_weights(os.path.join(HERE, "golden", "keras_vae_weights_scoped.h5"), "vae"), G.tiny_weights("vae", 12))
    _eq(K.load_weights(os.path.join(HERE, "golden", "keras_unet_weights_scoped.h5"), "unet"), G.tiny_weights("unet", 11))


def test_shape_validation_and_container_sniffing(tmp_path):
    w = G.tiny_weights("unet", 3)
    p = str(tmp_path / "u.hdf5")
    K.save_weights(p, w, "unet")
    bad = {k: v.shape for k, v in w.items()}
    bad["c1/kernel"] = (3, 3, 3, 4, 32)
    with pytest.raises(ValueError, match="c1/kernel has shape"):
        K.load_weights(p, "unet", expected_shapes=bad)
    with pytest.raises(H.Hdf5Error, match="not a LatticeDFCVAE"):
        K.load_weights(p, "vae")
    # round-1 checkpoints: an .npz archive behind the reference's .hdf5 name still loads
    q = str(tmp_path / "old.hdf5")
    K.save_npz(q, w)
    _eq({k: v.astype(np.float32) for k, v in K.load_weights(q, "unet").items()}, w)
    r = str(tmp_path / "junk.h5")
    open(r, "wb").write(b"not a weight file")
    with pytest.raises(ValueError, match="neither a Keras HDF5"):
        K.load_weights(r, "unet")


def _h5():
    import h5ref
    if h5ref.find_lib() is None:
        pytest.skip("libhdf5 not present in this image")
    return h5ref.H5()


def test_real_libhdf5_reads_our_files(tmp_path):
    h = _h5()
    w = G.tiny_weights("vae", 5)
    p = str(tmp_path / "vae.h5")
    K.save_weights(p, w, "vae")
    assert h.read_str_attr(p, "/", "layer_names") == [b"encoder", b"decoder"]
    assert h.read_str_attr(p, "/", "keras_version") == [b"2.3.1"]
    names = h.read_str_attr(p, "/encoder", "weight_names")
    assert names[0] == b"conv3d_1/kernel:0" and names[-1] == b"batch_normalization_4/moving_variance:0"
    assert np.array_equal(h.read_dataset(p, "/encoder/conv3d_1/kernel:0"), w["e0/kernel"])
    assert np.array_equal(h.read_dataset(p, "/decoder/decoder_output/bias:0"), w["dout/bias"])
    assert np.array_equal(h.read_dataset(p, "/encoder/z_log_var/kernel:0"), w["z_log_var/kernel"])
    h5ls = shutil.which("h5ls") or "/opt/conda/bin/h5ls"
    if os.path.exists(h5ls):
        out = subprocess.run([h5ls, "-r", p], capture_output=True, text=True)
        assert out.returncode == 0 and "/decoder/dense_2/kernel:0" in out.stdout, out.stderr


def test_full_size_unet_through_real_libhdf5(tmp_path):
    """The real tensor shapes (31 M parameters, 125 MB): libhdf5 writes the Keras file, our reader loads it; our
    writer saves it again, libhdf5 reads it back -- bit-equal both ways."""
    h = _h5()
    from icsg3d_amd.synthetic import glorot_params, unet_param_shapes
    shapes = unet_param_shapes(4, 95)
    w = glorot_params(shapes, seed=7)
    rng = np.random.default_rng(1)
    for n in K.UNET_ORDER:
        c = w[n + "/bias"].shape[0]
        w[n + "/moving_mean"], w[n + "/moving_var"] = rng.standard_normal(c).astype(np.float32), rng.uniform(0.5, 2, c).astype(np.float32)
    p = str(tmp_path / "ref.h5")
    h.write_keras(p, G.keras_unet_layers(w, 1))
    exp = dict(shapes)
    got = K.load_weights(p, "unet", expected_shapes=exp)
    _eq(got, w)
    q = str(tmp_path / "ours.hdf5")
    K.save_weights(q, got, "unet", full_model=True)
    assert np.array_equal(h.read_dataset(q, "/model_weights/conv3d_9/conv3d_9/kernel:0"), w["c13/kernel"])
    assert np.array_equal(h.read_dataset(q, "/model_weights/batch_normalization_14/batch_normalization_14/moving_variance:0"),
                          w["c18/moving_var"])
    assert np.array_equal(h.read_dataset(q, "/model_weights/soft/soft/kernel:0"), w["soft/kernel"])


def test_unet_model_save_carries_the_keras_layer_graph(tmp_path):
    """AtomUnet.model.save writes the functional-model JSON Keras 2.3.1 stores for unet_3d_multiclass
    (/root/reference/unet/unet.py:272-355,378-379), so that load_model (vae/lattice_vae.py:120) can rebuild it and find
    the perceptual taps re_lu_2/4/6/8 (:100).  Parse it back from the file, rebuild the layer list, run shape
    inference over it and check names, shapes and inbound nodes.  (A real Keras load_model cannot be run here.)"""
    from icsg3d_amd import checkpoint as ck
    from icsg3d_amd.synthetic import bn_state_defaults, glorot_params, unet_param_shapes
    C, ncls, d = 4, 95, 32
    shapes = unet_param_shapes(C, ncls)
    W = dict(glorot_params(shapes, 3))
    W.update(bn_state_defaults(shapes))
    path = str(tmp_path / "unet.h5")
    ck.save_weights(path, W, "unet", full_model=True, model_config=ck.unet_model_config((d, d, d, C), ncls),
                    training_config=ck.unet_training_config(3e-6))
    cfg, tr = ck.read_model_config(path)
    assert cfg["class_name"] == "Model" and cfg["keras_version"] == "2.3.1" and cfg["config"]["name"] == "unet"
    layers = cfg["config"]["layers"]
    names = [l["name"] for l in layers]
    assert len(names) == len(set(names))
    kinds = {}
    for l in layers:
        kinds.setdefault(l["class_name"], []).append(l["name"])
    assert kinds["InputLayer"] == ["unet_input"]
    assert kinds["Conv3D"] == ["conv3d_%d" % k for k in range(1, 15)] + ["soft", "sig"]
    assert kinds["ReLU"] == ["re_lu_%d" % k for k in range(1, 15)]
    assert kinds["BatchNormalization"] == ["batch_normalization_%d" % k for k in range(1, 15)]
    assert kinds["MaxPooling3D"] == ["max_pooling3d_%d" % k for k in (1, 2, 3)]
    assert kinds["UpSampling3D"] == ["up_sampling3d_%d" % k for k in (1, 2, 3)]
    assert kinds["Concatenate"] == ["concatenate_%d" % k for k in (1, 2, 3)]
    assert cfg["config"]["input_layers"] == [["unet_input", 0, 0]]
    assert cfg["config"]["output_layers"] == [["soft", 0, 0], ["sig", 0, 0]]
    # rebuild: every inbound layer exists BEFORE its consumer (creation order), shapes by inference
    by = {l["name"]: l for l in layers}
    shape = {}
    for l in layers:
        src = [n[0] for n in l["inbound_nodes"][0]] if l["inbound_nodes"] else []
        assert all(s in shape for s in src), l["name"]
        c = l["config"]
        if l["class_name"] == "InputLayer":
            shape[l["name"]] = tuple(c["batch_input_shape"][1:])
        elif l["class_name"] == "Conv3D":
            assert c["padding"] == "same" and c["strides"] == [1, 1, 1] and c["use_bias"] and c["data_format"] == "channels_last"
            shape[l["name"]] = shape[src[0]][:3] + (c["filters"],)
        elif l["class_name"] in ("ReLU", "BatchNormalization"):
            shape[l["name"]] = shape[src[0]]
        elif l["class_name"] == "MaxPooling3D":
            assert c["pool_size"] == [2, 2, 2] and c["strides"] == [2, 2, 2] and c["padding"] == "valid"
            shape[l["name"]] = tuple(v // 2 for v in shape[src[0]][:3]) + shape[src[0]][3:]
        elif l["class_name"] == "UpSampling3D":
            shape[l["name"]] = tuple(v * 2 for v in shape[src[0]][:3]) + shape[src[0]][3:]
        elif l["class_name"] == "Concatenate":
            assert c["axis"] == -1 and shape[src[0]][:3] == shape[src[1]][:3]
            shape[l["name"]] = shape[src[0]][:3] + (shape[src[0]][3] + shape[src[1]][3],)
    assert shape["soft"] == (d, d, d, ncls) and shape["sig"] == (d, d, d, 1)
    assert by["soft"]["config"]["activation"] == "softmax" and by["sig"]["config"]["activation"] == "sigmoid"
    assert by["soft"]["config"]["kernel_size"] == [1, 1, 1]
    # the graph's wiring: Conv -> ReLU -> BN blocks (unet.py:276-278), [skip, upsampled] concat order (:312,322,332)
    for k in range(1, 15):
        assert by["re_lu_%d" % k]["inbound_nodes"] == [[["conv3d_%d" % k, 0, 0, {}]]]
        assert by["batch_normalization_%d" % k]["inbound_nodes"] == [[["re_lu_%d" % k, 0, 0, {}]]]
        assert by["batch_normalization_%d" % k]["config"]["epsilon"] == 0.001
    assert [n[0] for n in by["concatenate_1"]["inbound_nodes"][0]] == ["batch_normalization_6", "up_sampling3d_1"]
    assert [n[0] for n in by["concatenate_2"]["inbound_nodes"][0]] == ["batch_normalization_4", "up_sampling3d_2"]
    assert [n[0] for n in by["concatenate_3"]["inbound_nodes"][0]] == ["batch_normalization_2", "up_sampling3d_3"]
    assert by["up_sampling3d_1"]["inbound_nodes"][0][0][0] == "batch_normalization_8"
    # the perceptual taps (lattice_vae.py:100): re_lu_2/4/6/8 = the ReLUs after c2, c4, c6, c10, i.e. the 2nd, 4th, 6th and
    # 8th convolution, with 64 / 128 / 256 / 512 channels at 32 / 16 / 8 / 4 voxels
    assert [shape["re_lu_%d" % k] for k in (2, 4, 6, 8)] == [(32, 32, 32, 64), (16, 16, 16, 128), (8, 8, 8, 256), (4, 4, 4, 512)]
    # every weight in /model_weights belongs to a layer of the graph with the shape the graph implies
    for ln, ws in ck.read_keras_h5(path):
        assert ln in by
        for wn, arr in ws:
            if wn.endswith("kernel:0"):
                k = by[ln]["config"]["kernel_size"][0]
                cin = shape[by[ln]["inbound_nodes"][0][0][0]][3]
                assert arr.shape == (k, k, k, cin, by[ln]["config"]["filters"])
    # training_config: the closure `loss` (the key of the reference's custom_objects, unet.py:393-399) and the metrics
    assert tr["loss"] == {"soft": "loss", "sig": "binary_crossentropy"} and tr["metrics"] == {"soft": ["f1_m", "wr_m"]}
    assert tr["optimizer_config"]["class_name"] == "Adam" and abs(tr["optimizer_config"]["config"]["learning_rate"] - 3e-6) < 1e-12
    # and the file still loads as weights, bit for bit
    back = ck.load_weights(path, "unet", expected_shapes=dict(shapes))
    assert all(np.array_equal(back[k], W[k]) for k in W)
