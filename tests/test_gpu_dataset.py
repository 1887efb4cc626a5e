"""SURVEY 8(f) rank 3 on the GPU: the training scripts driven from the reference's ON-DISK dataset format instead of
--synthetic (/root/reference/train_unet.py:95-111, train_vae.py:104-134, unet/data.py:64-100, vae/data.py:66-100):
data/<name>/matrices/{density_matrices,species_matrices,coordinate_grids}/<id>[_rot_k].npy + data/<name>/<name>.csv
with pd.qcut condition bins.  The directory listing is one of tests/golden/data_split_golden.json's cases, so the ids
the scripts train on must equal the lists the REFERENCE's own data_split produced for it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = 16


def _write_dataset(root, name, files, seed=0):
    from icsg3d_amd.synthetic import synthetic_batch
    mats = os.path.join(root, "data", name, "matrices")
    for sub in ("density_matrices", "species_matrices", "coordinate_grids"):
        os.makedirs(os.path.join(mats, sub))
    base_ids = sorted({f.split("_rot_")[0].replace(".npy", "") for f in files})
    for k, f in enumerate(files):
        X, lab, _ = synthetic_batch(1, D, 4, seed=seed + k, noise=1e-3)
        # the reference stores flat or (d,d,d) arrays and reshapes on load (unet/data.py:88-99): store them flat
        np.save(os.path.join(mats, "density_matrices", f), X[0, ..., 0].astype(np.float64).ravel())
        np.save(os.path.join(mats, "coordinate_grids", f), X[0, ..., 1:4].astype(np.float64))
        np.save(os.path.join(mats, "species_matrices", f), lab[0].astype(np.int64))
    rng = np.random.default_rng(seed)
    with open(os.path.join(root, "data", name, name + ".csv"), "w") as fh:
        fh.write("task_id,pretty_formula,formation_energy_per_atom,nsites\n")
        for i in base_ids:
            fh.write("%s,X%s,%.6f,%d\n" % (i, i.replace("-", ""), rng.normal(), rng.integers(2, 9)))
    return base_ids


def test_training_scripts_run_from_the_on_disk_dataset(tmp_path):
    from icsg3d_amd.hdf5_min import is_hdf5
    gold = {c["name"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "data_split_golden.json")))}
    case = gold["mp_ids_all"]                      # 23 compounds x (1 + 2 rotations), n = None, frac 0.8, n_rot 2
    _write_dataset(str(tmp_path), "x", case["files"])
    env = dict(os.environ, PYTHONPATH=ROOT)
    common = ["--name", "x", "--d", str(D), "--epochs", "1", "--batch_size", "3", "--nrot", "2", "--split", "0.8"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_unet.py"), "--lr", "1e-4"] + common,
                       cwd=tmp_path, env=env, check=True, capture_output=True, text=True)
    ids = json.load(open(tmp_path / "output" / "unet" / "x" / "split_ids.json"))
    assert ids["train"] == case["train"] and ids["val"] == case["val"]          # the reference's own split, in order
    for f in ("unet_weights_x.best.hdf5", "unet_weights_x.best.h5", "class_weights.npy"):
        assert os.path.exists(tmp_path / "saved_models" / "unet" / "x" / f), f
    assert is_hdf5(str(tmp_path / "saved_models" / "unet" / "x" / "unet_weights_x.best.hdf5"))
    assert is_hdf5(str(tmp_path / "saved_models" / "unet" / "x" / "unet_weights_x.best.h5"))
    # ModelCheckpoint without save_weights_only (/root/reference/unet/unet.py:361-367) writes a FULL model into the
    # "weights" path: model_config at the root, the tensors under model_weights/
    from icsg3d_amd.hdf5_min import Hdf5File
    with Hdf5File(str(tmp_path / "saved_models" / "unet" / "x" / "unet_weights_x.best.hdf5")) as f:
        assert "model_config" in f.attrs and "model_weights" in f.keys(), (list(f.attrs), list(f.keys()))
    assert "nan" not in r.stdout.lower() and "val_loss improved" in r.stdout
    cw = np.load(tmp_path / "saved_models" / "unet" / "x" / "class_weights.npy")
    assert cw.shape == (95,) and cw[0] == 0.0 and np.all(np.isfinite(cw))       # train_unet.py:107-111

    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_vae.py"), "--ncond", "4"] + common,
                       cwd=tmp_path, env=env, check=True, capture_output=True, text=True)
    ids = json.load(open(tmp_path / "output" / "vae" / "x" / "split_ids.json"))
    nt, nv = len(case["train"]) // 3 * 3, len(case["val"]) // 3 * 3             # trimmed to whole batches (train_vae.py:108-111)
    assert ids["train"] == case["train"][:nt] and ids["val"] == case["val"][:nv]
    for f in ("vae_weights_x.best.hdf5", "vae_weights_x.best.h5"):
        assert is_hdf5(str(tmp_path / "saved_models" / "vae" / "x" / f)), f
    line = [ln for ln in r.stdout.splitlines() if "Train Loss" in ln][-1]
    vals = [float(t) for t in line.replace(":", " ").split() if t.replace(".", "", 1).replace("-", "", 1).isdigit()]
    assert len(vals) >= 8 and np.all(np.isfinite(vals)), line

    # generate.py on the same dataset with the reference's flags: base compound by task id, qcut condition from the CSV
    r = subprocess.run([sys.executable, os.path.join(ROOT, "generate.py"), "--name", "x", "--base", "mp-3", "--d", str(D),
                        "--batch_size", "2", "--nsamples", "2", "--ncond", "4", "--eps_frac", "0.25", "--clus_iters", "5",
                        "--alpha", "90", "--beta", "90", "--gamma", "90", "--target", "formation_energy_per_atom"],
                       cwd=tmp_path, env=env, check=True, capture_output=True, text=True)
    res = tmp_path / "output" / "results" / "Xmp3__v=0.5"
    assert np.load(res / "species" / "1.npy").shape == (D, D, D)
    assert np.load(res / "coords" / "1.npy").shape[1] == 4


def test_generators_feed_the_engine_what_the_reference_contract_says(tmp_path):
    """UnetDataGenerator / VAEDataGenerator over files: shapes, dtypes, label / mask consistency, qcut bins, and a
    train step per generator batch through the class API."""
    from icsg3d_amd.unet.data import UnetDataGenerator
    from icsg3d_amd.unet.unet import AtomUnet
    from icsg3d_amd.vae.data import VAEDataGenerator
    files = ["mp-%d%s.npy" % (i, r) for i in range(1, 7) for r in ("", "_rot_0")]
    _write_dataset(str(tmp_path), "y", files, seed=5)
    path = str(tmp_path / "data" / "y" / "matrices")
    g = UnetDataGenerator(files, data_path=path, batch_size=4, dim=(D, D, D), n_channels=4, shuffle=False)
    assert len(g) == 3
    X, (y, b) = g[1]
    assert X.shape == (4, D, D, D, 4) and X.dtype == np.float32 and y.shape == (4, D, D, D) and y.dtype == np.uint8
    assert np.array_equal(b[..., 0] != 0, y != 0) and g.list_IDs_temp == files[4:8]
    M0 = np.load(os.path.join(path, "density_matrices", files[4])).reshape(D, D, D)
    assert np.array_equal(X[0, ..., 0], M0.astype(np.float32))
    unet = AtomUnet(input_shape=(D, D, D, 4), lr=1e-4, max_batch=4)
    m = unet.model.train_on_batch(X, [y, b])
    assert np.all(np.isfinite(m))
    v = VAEDataGenerator(files, data_path=path, property_csv=str(tmp_path / "data" / "y" / "y.csv"), batch_size=4,
                         dim=(D, D, D), n_channels=4, n_bins=3)
    M, cond = v[0]
    assert M.shape == (4, D, D, D, 4) and cond.shape == (4, 3) and np.array_equal(cond.sum(1), np.ones(4))
    assert np.array_equal(cond[0], cond[1])          # a compound and its rotation share the bin (vae/data.py:94-100)
    import pandas as pd
    df = pd.read_csv(tmp_path / "data" / "y" / "y.csv")
    bins = pd.qcut(df["formation_energy_per_atom"], 3, np.arange(3)).astype(int)
    assert int(np.argmax(cond[0])) == int(bins[df["task_id"] == "mp-1"].values[0])
