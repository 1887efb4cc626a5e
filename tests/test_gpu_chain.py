"""Fused inference tail (generate.py:204-225): decoder -> U-Net -> argmax / threshold on the device, plus the
coordinate channels' min/max for to_lattice_params (utils.py:160-178); Keras-HDF5 checkpoints and engine re-sizing
through the class API."""
import os

import numpy as np
import pytest

from oracle import numpy_ref as R
from saturated import saturate_head

pytestmark = pytest.mark.gpu


def test_decode_to_unet_labels_matches_two_step_path_and_oracle():
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    from icsg3d_amd.utils import to_lattice_params, to_lattice_params_from_minmax
    B, d, C = 3, 16, 4
    uo = R.UnetOracle(in_ch=C, seed=1)
    vo = R.VaeOracle(uo, in_ch=C, d=d, seed=3)
    rng = np.random.default_rng(3)
    for o in (uo, vo):                       # non-trivial moving statistics: eval-mode BN everywhere
        for k in list(o.S):
            o.S[k] = rng.uniform(0.5, 1.5, o.S[k].shape) if k.endswith("var") else rng.uniform(-0.2, 0.2, o.S[k].shape)
    z = rng.standard_normal((B, 256))
    cond = np.eye(10)[[1, 4, 7]]
    # a confident segmentation head (oracle/confident_head.py): with Glorot heads sig never reaches 0.8 and the mask comparison
    # below would be all-zeros == all-zeros.  Pseudo-labels: the densest 3 % of each decoded grid.
    rec0 = vo.predict_decoder(z, cond)
    dens = rec0[..., 0]
    lab = np.zeros(dens.shape, np.uint8)
    for b_ in range(B):
        lab[b_][dens[b_] > np.quantile(dens[b_], 0.97)] = 1 + 13 * b_
    saturate_head(uo, rec0, lab, training=False)
    pm = UnetEngine(in_channels=C, d=d, max_batch=2); pm.set_weights({**uo.P, **uo.S})
    ve = VaeEngine(pm, in_channels=C, d=d, max_batch=2); ve.set_weights({**vo.P, **vo.S})
    seg = UnetEngine(in_channels=C, d=d, max_batch=4); seg.set_weights({**uo.P, **uo.S})   # a separate U-Net handle
    out = ve.decode_to_labels(seg, z, cond, thresh=0.8)        # B=3 streams through max_batch=2 in two chunks
    # (a) bit-identical to the host round trip it replaces
    rec = ve.decode(z, cond)
    sp, mk = seg.predict_labels(rec, 0.8)
    assert np.array_equal(out["species"], sp) and np.array_equal(out["mask"], mk)
    assert np.array_equal(out["density"], rec[..., 0])
    assert np.array_equal(out["coord_minmax"][:, :, 0], rec[..., 1:4].min(axis=(1, 2, 3)))
    assert np.array_equal(out["coord_minmax"][:, :, 1], rec[..., 1:4].max(axis=(1, 2, 3)))
    np.testing.assert_allclose(to_lattice_params_from_minmax(out["coord_minmax"], d=d), to_lattice_params(rec[..., 1:], d=d),
                               rtol=1e-12)
    # (b) against the fp64 oracle chain: labels bit-exact wherever the decision margin exceeds 1e-4
    rec_ref = vo.predict_decoder(z, cond)
    soft_ref, sig_ref = uo.forward(rec_ref, training=False)
    srt = np.sort(soft_ref, -1)
    clear = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert clear.mean() > 0.5
    assert np.array_equal(out["species"][clear], soft_ref.argmax(-1)[clear])
    clear_s = np.abs(sig_ref[..., 0] - 0.8) > 1e-4
    want = sig_ref[..., 0] >= 0.8
    assert want.any() and out["mask"].any() and 0.003 < want.mean() < 0.5 and clear_s.mean() > 0.999
    assert np.array_equal(out["mask"][clear_s].astype(bool), want[clear_s])
    assert len(np.unique(out["species"])) >= 3
    assert np.abs(out["density"] - rec_ref[..., 0]).max() <= 1e-5 * np.abs(rec_ref).max()


def test_class_api_hdf5_checkpoints_resize_and_generate_tail(tmp_path):
    from icsg3d_amd.hdf5_min import is_hdf5
    from icsg3d_amd.synthetic import synthetic_batch
    from icsg3d_amd.unet.unet import AtomUnet
    from icsg3d_amd.vae.lattice_vae import LatticeDFCVAE
    d, C = 16, 4
    X, lab, cond = synthetic_batch(6, d, C, seed=1, noise=1e-3)
    unet = AtomUnet(input_shape=(d, d, d, C), lr=1e-3, max_batch=2)
    y = [np.eye(95, dtype=np.float32)[lab], (lab != 0)[..., None].astype(np.float32)]   # reference generator layout
    unet.model.train_on_batch(X[:2], [y[0][:2], y[1][:2]])
    eng = unet._eng
    # inference larger than max_batch streams through the SAME engine (no re-creation, optimizer state kept)
    soft, sig = unet.model.predict(X)
    assert unet._eng is eng and soft.shape == (6, d, d, d, 95)
    _, _, t1 = eng.get_optimizer_state()
    # a larger TRAINING batch re-creates the engine and carries Adam's state across
    unet.model.train_on_batch(X[:4], [y[0][:4], y[1][:4]])
    assert unet._eng is not eng and unet._eng.max_batch == 4
    m, v, t2 = unet._eng.get_optimizer_state()
    assert t1 == 1 and t2 == 2 and np.abs(m).max() > 0
    soft, sig = unet.model.predict(X)
    # Keras-HDF5 checkpoints behind the reference's file names
    wpath, mpath = str(tmp_path / "unet_weights.best.hdf5"), str(tmp_path / "unet.h5")
    unet.model.save_weights(wpath); unet.model.save(mpath)
    assert is_hdf5(wpath) and is_hdf5(mpath)
    clone = AtomUnet(input_shape=(d, d, d, C), weights=mpath, max_batch=2)
    s2, g2 = clone.model.predict(X[:2])
    assert np.array_equal(s2, soft[:2]) and np.array_equal(g2, sig[:2])
    with pytest.raises(ValueError, match="c1/kernel has shape"):
        AtomUnet(input_shape=(d, d, d, 1), weights=wpath)
    # VAE: perceptual_model from the U-Net's .h5 (load_model(perceptual_model), lattice_vae.py:120)
    vae = LatticeDFCVAE(input_shape=(d, d, d, C), perceptual_model=mpath)
    vae._set_model(None, batch_size=2)
    eps = np.random.default_rng(0).standard_normal((2, 256))
    vae.model.train_on_batch([X[:2], cond[:2]], X[:2], eps=eps)
    vpath = str(tmp_path / "vae.hdf5")
    vae.model.save_weights(vpath)
    vae2 = LatticeDFCVAE(input_shape=(d, d, d, C), perceptual_model=unet)
    vae2._set_model(vpath, batch_size=2)
    z = np.random.default_rng(1).standard_normal((3, 256))
    r1, r2 = vae.decoder.predict([z, cond[:3]]), vae2.decoder.predict([z, cond[:3]])
    assert np.array_equal(r1, r2)
    out = vae2.decode_segment(z, cond[:3], clone, thresh=0.8)
    sp, mk = clone.model.predict_labels(r2, 0.8)
    assert np.array_equal(out["species"], sp) and np.array_equal(out["mask"], mk)
