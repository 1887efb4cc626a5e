"""Host-side logic of the product package that needs no GPU: synthetic workload definition,
checkpoints, label handling, numpy metric mirrors, data generators, DP sharding."""
import os

import numpy as np
import pytest

from icsg3d_amd import synthetic
from icsg3d_amd.checkpoint import load_npz, save_npz
from icsg3d_amd.dataparallel import shard_range
from icsg3d_amd.unet import unet as U
from icsg3d_amd.unet.data import SyntheticUnetGenerator, UnetDataGenerator
from icsg3d_amd.unet.get_weights import get_weights
from icsg3d_amd.vae.data import VAEDataGenerator
from oracle import numpy_ref as R


def test_synthetic_definition_matches_oracle():
    assert synthetic.unet_param_shapes(4, 95) == R.unet_param_shapes(4, 95)
    assert synthetic.vae_param_shapes(1, 10, (16, 32, 64, 128), 256, 32) == R.vae_param_shapes(1)
    P = synthetic.glorot_params(synthetic.unet_param_shapes(1, 95), 1)
    Q = R.init_params(R.unet_param_shapes(1, 95), 1, np.float32)
    assert all(np.array_equal(P[k], Q[k]) for k in Q)
    X, lab, cond = synthetic.synthetic_batch(3, 16, 4, seed=7)
    X2, lab2, cond2 = R.synthetic_batch(3, 16, 4, seed=7)
    assert np.array_equal(X, X2) and np.array_equal(lab, lab2) and np.array_equal(cond, cond2)
    assert X.shape == (3, 16, 16, 16, 4) and lab.max() <= 94 and X[..., 0].min() >= 0


def test_checkpoint_roundtrip_keeps_reference_paths(tmp_path):
    w = {"c1/kernel": np.random.rand(3, 3, 3, 1, 4).astype(np.float32), "c1/moving_var": np.ones(4, np.float32)}
    path = str(tmp_path / "saved_models" / "unet" / "x" / "unet_weights_x.best.hdf5")
    save_npz(path, w, {"num_classes": 95})
    assert os.path.exists(path)                      # exactly the reference's file name, no ".npz" appended
    w2, meta = load_npz(path)
    assert set(w2) == set(w) and all(np.array_equal(w[k], w2[k]) for k in w) and int(meta["num_classes"]) == 95


def test_label_handling_and_metric_mirrors():
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 95, (2, 4, 4, 4)).astype(np.uint8)
    onehot = np.eye(95, dtype=np.float32)[lab]
    assert np.array_equal(U._to_labels(onehot, 95), lab)
    assert np.array_equal(U._to_labels(lab[..., None].astype(np.float64), 95), lab)
    p = R.softmax(rng.standard_normal((2, 4, 4, 4, 95)) * 4)
    y = R.one_hot(lab, 95)
    assert np.isclose(U.f1_m(y, p), R.f1_m(y, p)) and np.isclose(U.wr_m(y, p), R.wr_m(y, p))
    np.testing.assert_allclose(U.weighted_categorical_crossentropy(95)(y, p), R.wcce_loss(y, p, 95.0))
    assert set(U.custom_objects) == {"loss", "f1_m", "wr_m"} and np.array_equal(U.class_weights, np.ones(95))


def test_file_generators_follow_the_reference_layout(tmp_path):
    root = tmp_path / "matrices"
    for sub in ("density_matrices", "species_matrices", "coordinate_grids"):
        os.makedirs(root / sub)
    ids = []
    rng = np.random.default_rng(1)
    for i in range(4):
        name = "mp-%d.npy" % i
        ids.append(name)
        np.save(root / "density_matrices" / name, rng.random((32, 32, 32)))
        np.save(root / "species_matrices" / name, rng.integers(0, 95, (32, 32, 32)))
        np.save(root / "coordinate_grids" / name, rng.random((32, 32, 32, 3)))
    g = UnetDataGenerator(ids, str(root), batch_size=2, n_channels=4)
    assert len(g) == 2
    X, (y, b) = g[1]
    assert X.shape == (2, 32, 32, 32, 4) and y.shape == (2, 32, 32, 32) and y.dtype == np.uint8
    assert b.shape == (2, 32, 32, 32, 1) and np.array_equal(b[..., 0], (y != 0))
    g1 = UnetDataGenerator(ids, str(root), batch_size=2, n_channels=4, one_hot=True)
    assert g1[0][1][0].shape == (2, 32, 32, 32, 95)       # reference format (unet/data.py:89)
    w = get_weights(str(root), ids, 95)
    assert w.shape == (95,) and np.all(np.isfinite(w)) and np.array_equal(get_weights(), np.ones(95))
    import pandas as pd
    csv = tmp_path / "p.csv"
    pd.DataFrame({"task_id": ["mp-%d" % i for i in range(4)], "formation_energy_per_atom": [0.1, 0.5, -1.0, 2.0],
                  "nsites": [2, 3, 4, 5]}).to_csv(csv, index=False)
    vg = VAEDataGenerator(ids, str(root), batch_size=2, n_channels=4, property_csv=str(csv), n_bins=2)
    M, cond = vg[0]
    assert M.shape == (2, 32, 32, 32, 4) and cond.shape == (2, 2) and np.all(cond.sum(1) == 1)
    assert vg.list_IDs_temp == ids[:2]


def test_synthetic_generator_contract():
    g = SyntheticUnetGenerator(5, batch_size=2, dim=(16, 16, 16), n_channels=1)
    assert len(g) == 2 and g.batch_size == 2 and len(g.list_IDs) == 5
    X, (y, b) = g[0]
    assert X.shape == (2, 16, 16, 16, 1) and y.dtype == np.uint8 and b.shape == (2, 16, 16, 16, 1)


def test_shard_range_partitions_the_global_batch():
    for gb, world in ((256, 8), (10, 4), (3, 8)):
        spans = [shard_range(gb, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 8, 8)


def test_data_split_matches_reference_golden(tmp_path):
    """Bit-exact against id lists produced by the reference's own data_split (utils.py:36-61) run in the
    build container (tests/golden/make_data_split_golden.py): cap before shuffle, Python `random` seeded
    through the global generator, unconditional _rot_k expansion, the str.strip(".npy") character quirk."""
    import json
    import random
    from icsg3d_amd.utils import data_split
    here = os.path.dirname(os.path.abspath(__file__))
    cases = json.load(open(os.path.join(here, "golden", "data_split_golden.json")))
    assert len(cases) >= 7
    for case in cases:
        root = tmp_path / case["name"]
        os.makedirs(root / "density_matrices")
        for f in case["files"]:
            open(root / "density_matrices" / f, "w").close()
        tr, va = data_split(str(root), **case["kwargs"])
        assert tr == case["train"], case["name"]
        assert va == case["val"], case["name"]
        if case["kwargs"].get("shuffle", True):
            # the reference reseeds the GLOBAL generator; callers that draw from `random` afterwards see it
            after = random.random()
            random.seed(case["kwargs"].get("seed", 28))
            random.shuffle([f for f in sorted(case["files"]) if f.endswith(".npy") and "_rot_" not in f][:case["kwargs"].get("n")])
            assert random.random() == after, case["name"]
    # defaults are the reference's: n=None, frac=0.8, n_rot=10, shuffle=True, seed=28
    import inspect
    sig = inspect.signature(data_split)
    assert [(k, v.default) for k, v in sig.parameters.items()][1:] == [
        ("n", None), ("frac", 0.8), ("n_rot", 10), ("shuffle", True), ("seed", 28)]


def test_shard_ids_equal_whole_batches():
    from icsg3d_amd.dataparallel import DataParallelMixin, from_env, shard_ids
    ids = ["g%03d" % i for i in range(103)]
    shards = [shard_ids(ids, r, 4, 5) for r in range(4)]
    assert {len(s) for s in shards} == {25}                       # 103 // (4*5) = 5 batches of 5 on every rank
    flat = [x for s in shards for x in s]
    assert len(set(flat)) == 100 and set(flat) <= set(ids)        # disjoint
    assert shards[1][:3] == ["g001", "g005", "g009"]              # dealt round-robin
    assert shard_ids(ids, 0, 1, 10) == ids[:100]
    assert shard_ids(ids[:7], 1, 4, 5) == []
    # without a launcher there is no process group, and the mixin is inert: the single process writes
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE") if k in os.environ}
    try:
        assert from_env() is None
    finally:
        os.environ.update(env)
    m = DataParallelMixin()
    assert m._dp_is_writer() and m._dp_barrier() is None and m._dp_attach(object()) is None
    m._dp = (None, 3, 8, False, False)
    assert not m._dp_is_writer()
