"""Data-parallel path on the one GPU this box has: a single-rank RCCL communicator drives everything a multi-rank
run executes (ncclCommInitRank, state broadcast, BN moving-statistics average, gradient buckets on the comm stream
behind events, metric-sum all-reduce, SyncBN all-gather / all-reduce) -- with one rank every collective is the
identity, so results must equal the engine without a communicator BIT FOR BIT.  Multi-rank arithmetic is covered on
CPU by tests/test_dataparallel_gloo.py; the driver runs the 2/4/8-GPU bench."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unet_pair(B=2, d=16):
    from icsg3d_amd.engine import UnetEngine, comm_unique_id
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes
    P = glorot_params(unet_param_shapes(1, 95), 1)
    X, lab, cond = synthetic_batch(B, d, 1, seed=0, noise=1e-3)
    a = UnetEngine(in_channels=1, d=d, max_batch=B, lr=1e-3); a.set_weights(P)
    b = UnetEngine(in_channels=1, d=d, max_batch=B, lr=1e-3); b.set_weights(P)
    b.comm_init(0, 1, comm_unique_id())
    b.broadcast_state(0)
    return a, b, X, lab, cond


# fuse: the BatchNorm-backward-in-backward-data fusions forced on at this small size (default: from 64 MB of activations
# on, i.e. only at bench sizes) -- their weight-gradient fix-ups must be final before the gradient buckets leave
@pytest.mark.parametrize("sync_bn,fuse", [(False, False), (True, False), (False, True)])
def test_single_rank_communicator_is_bit_identical(sync_bn, fuse, monkeypatch):
    if fuse:
        monkeypatch.setenv("ICSG3D_DGRAD_BNFUSE_MIN", "0")
    a, b, X, lab, _ = _unet_pair()
    b.set_sync_bn(sync_bn)
    assert b.comm_info()["nranks"] == 1 and a.comm_info()["nranks"] == 0
    b.profile_enable(True)
    steps = 3
    ma = [a.train_step(X, lab) for _ in range(steps)]
    mb = [b.train_step(X, lab) for _ in range(steps)]
    for x, y in zip(ma, mb):
        if sync_bn:      # statistics merged through the fp64 rank-merge: same values to fp32 rounding
            np.testing.assert_allclose(x, y, rtol=2e-6)
        else:
            assert np.array_equal(x, y)
    wa, wb = a.get_weights(), b.get_weights()
    for k in wa:
        if sync_bn:
            assert np.abs(wa[k] - wb[k]).max() <= 1e-5 * max(np.abs(wa[k]).max(), 1e-12), k
        else:
            assert np.array_equal(wa[k], wb[k]), k
    info = b.comm_info()
    assert info["buckets_last_step"] >= 3, info            # head..c14 | c13 | c10+c9 | rest
    rows = {r["label"]: r for r in b.profile_rows()}
    assert rows["rccl_allreduce_grads"]["launches"] == steps * info["buckets_last_step"]
    assert abs(rows["rccl_allreduce_grads"]["bytes"] - steps * 4.0 * b.num_params()) < 1.0
    if not sync_bn:
        assert rows["rccl_allreduce_bn_moving"]["launches"] == steps
    # optimizer state travels through the ABI (engine re-creation keeps Adam's moments and step count)
    m, v, t = b.get_optimizer_state()
    assert t == steps and np.abs(m).max() > 0 and v.min() >= 0
    ma2, va2, ta2 = a.get_optimizer_state()
    if not sync_bn:
        assert ta2 == t and np.array_equal(m, ma2) and np.array_equal(v, va2)


def test_vae_single_rank_communicator_is_bit_identical():
    from icsg3d_amd.engine import UnetEngine, VaeEngine, comm_unique_id
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
    B, d = 2, 16
    PU = glorot_params(unet_param_shapes(1, 95), 1)
    PV = glorot_params(vae_param_shapes(1, d=d), 3)
    X, _, cond = synthetic_batch(B, d, 1, seed=0, noise=1e-3)
    eps = np.random.default_rng(2).standard_normal((B, 256)).astype(np.float32)
    out = []
    for with_comm in (False, True):
        pm = UnetEngine(in_channels=1, d=d, max_batch=B); pm.set_weights(PU)
        ve = VaeEngine(pm, in_channels=1, d=d, max_batch=B); ve.set_weights(PV)
        if with_comm:
            ve.comm_init(0, 1, comm_unique_id()); ve.broadcast_state(0)
        ms = [ve.train_step(X, cond, eps) for _ in range(2)]
        out.append((ms, ve.get_weights(), ve.comm_info()))
    (m0, w0, _), (m1, w1, info) = out
    assert all(np.array_equal(x, y) for x, y in zip(m0, m1))
    assert all(np.array_equal(w0[k], w1[k]) for k in w0)
    assert info["buckets_last_step"] >= 1


def test_broadcast_state_overwrites_replica_state():
    a, b, X, lab, _ = _unet_pair()
    for _ in range(2):
        b.train_step(X, lab)
    w, (m, v, t) = b.get_weights(), b.get_optimizer_state()
    b.broadcast_state(0)                                    # root = self: a no-op on the values
    w2, (m2, v2, t2) = b.get_weights(), b.get_optimizer_state()
    assert t2 == t == 2 and np.array_equal(m, m2) and np.array_equal(v, v2)
    assert all(np.array_equal(w[k], w2[k]) for k in w)


@pytest.mark.timeout(900)
def test_bench_through_torchrun_single_rank():
    """bench.py under `python -m torch.distributed.run --nproc-per-node 1` with ICSG3D_BENCH_FORCE_DIST=1: gloo
    rendezvous, ncclUniqueId hand-off, communicator, broadcast, buckets -- the launch line the driver uses for N > 1.
    The launcher starts before anything touches the GPU."""
    env = dict(os.environ, ICSG3D_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--batch", "4", "--no-cpu-baseline", "--soak-seconds", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=850)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["value"] > 0 and "RCCL buckets" in r["config"]["grad_allreduce"]
    assert r["secondary"]["value"] > 0 and r["roofline"]["frac"] > 0
    # the sustained figure next to the 3-step one: the same step for >= 1 s, on the host clock and on the GPU's own
    su = r["sustained"]
    assert su["seconds"] >= 0.9 and su["steps"] > 3 and abs(su["gpu_active_s"] - su["seconds"]) < 0.2 * su["seconds"]


def test_class_api_data_parallel_single_rank(tmp_path):
    """AtomUnet.enable_data_parallel: the communicator is attached when the engine is created (and again when a
    larger training batch re-creates it); with one rank the losses equal the plain model's bit for bit."""
    import torch.distributed as dist
    from icsg3d_amd.synthetic import synthetic_batch
    from icsg3d_amd.unet.unet import AtomUnet
    X, lab, _ = synthetic_batch(4, 16, 1, seed=0, noise=1e-3)
    y = [lab, (lab != 0).astype(np.float32)[..., None]]
    np.random.seed(11); a = AtomUnet(input_shape=(16, 16, 16, 1), lr=1e-3, weights=str(tmp_path / "a.hdf5"))
    np.random.seed(11); b = AtomUnet(input_shape=(16, 16, 16, 1), lr=1e-3, weights=str(tmp_path / "b.hdf5"))
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    try:
        b.enable_data_parallel(dist, 0, 1, force=True)
        for n in (2, 2, 4):                                   # the third step grows the engine: comm re-attached
            ma = a.model.train_on_batch(X[:n], y[0][:n])
            mb = b.model.train_on_batch(X[:n], y[0][:n])
            assert ma == mb, (n, ma, mb)
            assert b._eng.comm_info()["nranks"] == 1 and a._eng.comm_info()["nranks"] == 0
        assert b._eng.max_batch == 4
        assert b._dp_is_writer()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_train_scripts_through_torchrun_single_rank(tmp_path):
    """train_unet.py then train_vae.py under the launcher, data-parallel path forced with one rank: sharded
    synthetic ids, communicator through the class API, rank-0 checkpoints in Keras HDF5."""
    env = dict(os.environ, ICSG3D_FORCE_DP="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
              "127.0.0.1", "--master-port", "29541"]
    common = ["--name", "dp", "--synthetic", "8", "--channels", "1", "--dim", "16", "--epochs", "1", "--batch_size", "2"]
    p = subprocess.run(launch + [os.path.join(ROOT, "train_unet.py")] + common, capture_output=True, text=True,
                       env=env, cwd=str(tmp_path), timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "val_loss improved" in p.stdout and "Model saved" in p.stdout, p.stdout[-2000:]
    from icsg3d_amd.hdf5_min import is_hdf5
    for f in ("unet_weights_dp.best.hdf5", "unet_weights_dp.best.h5"):
        assert is_hdf5(str(tmp_path / "saved_models" / "unet" / "dp" / f)), f
    p = subprocess.run(launch + [os.path.join(ROOT, "train_vae.py")] + common + ["--ncond", "10"],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "Saving Model" in p.stdout and "Model saved" in p.stdout, p.stdout[-2000:]
    assert is_hdf5(str(tmp_path / "saved_models" / "vae" / "dp" / "vae_weights_dp.best.h5"))
