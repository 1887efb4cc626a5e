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
