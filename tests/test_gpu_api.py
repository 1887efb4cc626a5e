"""The reference-shaped Python surface (AtomUnet / LatticeDFCVAE) driven the way
train_unet.py / train_vae.py / generate.py drive the reference, on synthetic generators."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_atomunet_train_save_predict(tmp_path):
    from icsg3d_amd.unet.data import SyntheticUnetGenerator
    from icsg3d_amd.unet.unet import AtomUnet, f1_m
    d, C, B = 16, 1, 2
    weights = str(tmp_path / "saved_models" / "unet" / "t" / "unet_weights_t.best.hdf5")
    np.random.seed(0)
    unet = AtomUnet(num_classes=95, input_shape=(d, d, d, C), weights=weights, lr=3e-4)
    tg = SyntheticUnetGenerator(6, batch_size=B, dim=(d, d, d), n_channels=C)
    vg = SyntheticUnetGenerator(2, batch_size=B, dim=(d, d, d), n_channels=C, seed=100)
    unet.train_generator(tg, vg, epochs=2, output_dir=str(tmp_path / "out"))
    assert os.path.exists(weights) and os.path.exists(os.path.splitext(weights)[0] + ".h5")
    X, (y, b) = vg[0]
    soft, sig = unet.model.predict(X)
    assert soft.shape == (B, d, d, d, 95) and sig.shape == (B, d, d, d, 1)
    np.testing.assert_allclose(soft.sum(-1), 1.0, atol=1e-5)
    # reload into a fresh object (generate.py:176 path): identical predictions
    unet2 = AtomUnet(weights=os.path.splitext(weights)[0] + ".h5", input_shape=(d, d, d, C))
    soft2, sig2 = unet2.model.predict(X)
    assert np.array_equal(soft, soft2) and np.array_equal(sig, sig2)
    # one-hot targets (reference generator format) and uint8 ids give the same test metrics
    onehot = np.eye(95, dtype=np.float32)[y]
    m1 = unet2.model.test_on_batch(X, [onehot, b])
    m2 = unet2.model.test_on_batch(X, [y, b])
    assert m1 == m2
    # f1 metric column equals the reference's formula evaluated on the predictions
    assert abs(m1[3] - f1_m(onehot, soft2)) < 1e-5
    sp, mk = unet2.model.predict_labels(X, 0.8)
    assert np.array_equal(sp, soft2.argmax(-1)) and np.array_equal(mk, (sig2[..., 0] >= 0.8))


def test_lattice_vae_train_sample_roundtrip(tmp_path):
    from icsg3d_amd.unet.unet import AtomUnet
    from icsg3d_amd.vae.data import SyntheticVAEGenerator
    from icsg3d_amd.vae.lattice_vae import LatticeDFCVAE
    d, C, B = 16, 1, 2
    np.random.seed(1)
    unet = AtomUnet(input_shape=(d, d, d, C))
    pm_path = str(tmp_path / "unet.h5")
    unet.model.save(pm_path)
    vae = LatticeDFCVAE(input_shape=(d, d, d, C), perceptual_model=pm_path, output_dir=str(tmp_path))
    tg = SyntheticVAEGenerator(4, batch_size=B, dim=(d, d, d), n_channels=C)
    vg = SyntheticVAEGenerator(2, batch_size=B, dim=(d, d, d), n_channels=C, seed=50)
    wpath = str(tmp_path / "vae_weights.best.hdf5")
    vae.train(tg, vg, epochs=2, weights=wpath)
    assert os.path.exists(wpath) and np.all(np.isfinite(vae.losses))
    M, cond = vg[0]
    zm, zlv, z = vae.encoder.predict([M, cond])
    assert zm.shape == (B, 256) and not np.allclose(z, zm)          # sampled z
    rec = vae.decoder.predict([zm, cond])
    assert rec.shape == (B, d, d, d, C) and rec.min() >= 0.0         # decoder tail is ReLU
    zs, out = vae.sample_vae(3, var=0.5)
    assert out.shape == (3, d, d, d, C)
    # generate.py:172-173 path: fresh object + _set_model(weights)
    vae2 = LatticeDFCVAE(input_shape=(d, d, d, C), perceptual_model=pm_path)
    vae2._set_model(wpath, batch_size=B)
    assert np.array_equal(vae2.decoder.predict([zm, cond]), rec)


def test_training_scripts_run_on_synthetic_data(tmp_path):
    """train_unet.py then train_vae.py (which loads the U-Net checkpoint as perceptual model), CLI as
    in the reference README, on --synthetic grids."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    common = ["--name", "t", "--synthetic", "8", "--channels", "1", "--d", "16", "--epochs", "1", "--batch_size", "2"]
    subprocess.run([sys.executable, os.path.join(root, "train_unet.py")] + common, cwd=tmp_path, env=env, check=True)
    assert os.path.exists(tmp_path / "saved_models" / "unet" / "t" / "unet_weights_t.best.h5")
    subprocess.run([sys.executable, os.path.join(root, "train_vae.py")] + common, cwd=tmp_path, env=env, check=True)
    assert os.path.exists(tmp_path / "saved_models" / "vae" / "t" / "vae_weights_t.best.h5")
    # the reference's own flags (generate.py:52-102) are accepted; --synthetic stands in for the base compound
    subprocess.run([sys.executable, os.path.join(root, "generate.py"), "--name", "t", "--synthetic", "--channels", "1",
                    "--d", "16", "--batch_size", "2", "--nsamples", "4", "--eps_frac", "0.3", "--clus_iters", "3",
                    "--alpha", "90", "--beta", "90", "--gamma", "120", "--target", "band_gap", "--ncond", "10"],
                   cwd=tmp_path, env=env, check=True)
    res = tmp_path / "output" / "results" / "synthetic__v=0.5"
    sp = np.load(res / "species" / "3.npy")
    assert sp.shape == (16, 16, 16) and sp.dtype == np.uint8
    co = np.load(res / "coords" / "3.npy")
    assert co.ndim == 2 and co.shape[1] == 4


def test_batches_below_max_batch_match_a_right_sized_engine():
    """The last batch of an epoch is smaller than max_batch: workspaces and split-K plans are sized at
    max_batch, results must not depend on it (bit-equal to an engine created for exactly that batch)."""
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
    d = 16
    PU, PV = glorot_params(unet_param_shapes(1, 95), 1), glorot_params(vae_param_shapes(1, d=d), 3)

    def fresh(eng, P):   # weights, BatchNorm moving statistics and optimizer state back to their initial values
        eng.set_weights(P)
        for name, shape, trainable in eng.tensor_infos():
            if name.endswith("moving_mean"):
                eng.set_tensor(name, np.zeros(shape, np.float32))
            if name.endswith("moving_var"):
                eng.set_tensor(name, np.ones(shape, np.float32))
        eng.reset_optimizer()

    big_u = UnetEngine(d=d, max_batch=8, lr=1e-4)
    big_pm = UnetEngine(d=d, max_batch=8)
    big_v = VaeEngine(big_pm, d=d, max_batch=8, lr=1e-4)
    for B in (1, 3, 8):
        X, lab, cond = synthetic_batch(B, d, 1, seed=B, noise=1e-3)
        eps = np.random.default_rng(B).standard_normal((B, 256)).astype(np.float32)
        u = UnetEngine(d=d, max_batch=B, lr=1e-4)
        fresh(u, PU); fresh(big_u, PU)
        np.testing.assert_array_equal(big_u.predict(X)[0], u.predict(X)[0])
        np.testing.assert_array_equal(big_u.train_step(X, lab), u.train_step(X, lab))
        wa, wb = big_u.get_weights(), u.get_weights()
        assert all(np.array_equal(wa[k], wb[k]) for k in wa)
        pm = UnetEngine(d=d, max_batch=B)
        v = VaeEngine(pm, d=d, max_batch=B, lr=1e-4)
        fresh(pm, PU); fresh(big_pm, PU); fresh(v, PV); fresh(big_v, PV)
        np.testing.assert_array_equal(big_v.train_step(X, cond, eps), v.train_step(X, cond, eps))
        wa, wb = big_v.get_weights(), v.get_weights()
        assert all(np.array_equal(wa[k], wb[k]) for k in wa)


def test_launcher_counts_this_box_like_the_runtime():
    """icsg3d_amd.launcher.visible_gpus (KFD topology + render nodes + *_VISIBLE_DEVICES, no HIP in the caller) agrees with
    hipGetDeviceCount on the box the suite runs on."""
    from icsg3d_amd import _lib, launcher
    assert launcher.visible_gpus() == _lib.device_count() >= 1


@pytest.mark.parametrize("B,d", [(3, 16), (2, 32), (5, 32)])
def test_no_kernel_writes_past_its_buffers(B, d, monkeypatch):
    """ICSG3D_DEBUG_CANARY=1: 8 KB of guard bytes behind every device buffer of a handle; after train / test / predict steps
    of both engines and the fused inference tail at EVERY batch size the handle accepts (odd batches at d = 16: every tile
    remainder path) none of them was touched.  (Round 4 found such a write by accident; this looks for them.  Round 6's
    tests/tools/fuzz_steps.py found the next one: a DFC-VAE train step on 3 grids with a 5-grid handle at d = 32 wrote 16 KB past the
    bias-gradient partials of e0 and d3 -- the BatchNorm-backward pass takes 1536 blocks for 3 grids and 1280 for 5, and the
    buffer was sized for max_batch alone.  The (5, 32) case is that handle.)"""
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
    monkeypatch.setenv("ICSG3D_DEBUG_CANARY", "1")
    X, lab, cond = synthetic_batch(B, d, 1, seed=0, noise=1e-3)
    eps = np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32)
    ue = UnetEngine(in_channels=1, d=d, max_batch=B, lr=1e-3); ue.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
    ve = VaeEngine(ue, in_channels=1, d=d, max_batch=B, lr=5e-4); ve.set_weights(glorot_params(vae_param_shapes(1, d=d), 3))
    assert ue.check_canaries()[0] == 0 and ve.check_canaries()[0] == 0
    for b in range(B, 0, -1):
        ue.train_step(X[:b], lab[:b]); ue.test_step(X[:b], lab[:b]); ue.predict(X[:b])
        ve.train_step(X[:b], cond[:b], eps[:b]); ve.test_step(X[:b], cond[:b], eps[:b])
        z = ve.encode(X[:b], cond[:b], eps[:b])[2]
        ve.decode_to_atoms(ue, z, cond[:b], thresh=0.5, max_atoms=64)
    for eng in (ue, ve):
        dirty, what = eng.check_canaries()
        assert dirty == 0, what
    ve.close(); ue.close()


def test_graph_probe_times_both_forms_and_leaves_the_engine_training():
    """ics_net_graph_probe (measurement aid): the resident train step eagerly and as a replayed hipGraph."""
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
    B, d = 2, 16
    X, lab, cond = synthetic_batch(B, d, 1, seed=0, noise=1e-3)
    ue = UnetEngine(in_channels=1, d=d, max_batch=B, lr=1e-3); ue.set_weights(glorot_params(unet_param_shapes(1, 95), 1))
    ve = VaeEngine(ue, in_channels=1, d=d, max_batch=B, lr=5e-4); ve.set_weights(glorot_params(vae_param_shapes(1, d=d), 3))
    ue.upload_batch(X, lab)
    ve.upload_batch(X, cond, np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32))
    for eng in (ve, ue):
        eager, graph, nodes = eng.graph_probe(3)
        assert eager > 0 and graph > 0 and nodes > 50
    assert np.all(np.isfinite(ve.train_step_resident(True))) and np.all(np.isfinite(ue.train_step_resident(True)))
    ve.close(); ue.close()
