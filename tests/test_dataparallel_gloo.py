"""world_size-2 gloo test of the data-parallel logic (one process per GPU in production; here two CPU
processes).  Each rank runs the train step of its shard (the oracle stands in for the engine), the
gradients are averaged with the same flat all-reduce arithmetic the RCCL path uses, every rank
applies Adam -- parameters must stay identical across ranks and equal the single-process result
computed from both shards (local-BN data parallelism)."""
import os
import socket

import numpy as np
import pytest

import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dp_host_ref import allreduce_mean_host, syncbn_moments
from icsg3d_amd.dataparallel import exchange_unique_id, max_over_ranks, shard_range
from oracle import numpy_ref as R

D, C, GB = 8, 1, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_grads(lo, hi):
    X, lab, _ = R.synthetic_batch(GB, D, C, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    u = R.UnetOracle(in_ch=C, seed=1, lr=1e-3)
    cache = {}
    soft, sig = u.forward(X[lo:hi], training=True, cache=cache)
    return u, u.backward(lab[lo:hi], cache)


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        uid = exchange_unique_id(dist, rank, lambda: bytes(range(128)))
        assert uid == bytes(range(128))
        lo, hi = shard_range(GB, rank, world)
        u, g = _shard_grads(lo, hi)
        g = allreduce_mean_host(dist, g)
        u.apply_adam(g)
        t = max_over_ranks(dist, 1.0 + rank)
        assert t == float(world)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **{k.replace("/", "__"): v for k, v in u.P.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_averaging(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert all(np.array_equal(r0[k], r1[k]) for k in r0.files)          # replicas stay in lock-step
    # single-process expectation: mean of the two shard gradients, then Adam
    u, g0 = _shard_grads(*shard_range(GB, 0, 2))
    _, g1 = _shard_grads(*shard_range(GB, 1, 2))
    u.apply_adam({k: 0.5 * (g0[k] + g1[k]) for k in g0})
    for k, v in u.P.items():
        np.testing.assert_allclose(r0[k.replace("/", "__")], v, rtol=1e-12, atol=1e-15, err_msg=k)


# ---------------------------------------------------------------------------------------------------------------
# SyncBN: two ranks x B/2 grids with the per-channel statistics exchanged (forward: all-gather of (n, mean, M2)
# merged in rank order; backward: all-reduce of (sum d, sum d*xhat)) must reproduce ONE process at batch B -- the
# single-process reference (unet/unet.py:370).  The oracle stands in for the engine; the exchange arithmetic is
# the one csrc/elementwise.hip implements (bn_local_merge / bn_sync_finalize / bn_bwd_sync_c).
def _syncbn_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        X, lab, _ = R.synthetic_batch(GB, D, C, seed=0, dtype=np.float64)
        X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
        lo, hi = shard_range(GB, rank, world)
        u = R.UnetOracle(in_ch=C, seed=1, lr=1e-3)

        def moments(x):
            mean, var, _ = syncbn_moments(dist, x)
            return mean, var

        def allsum(vec, n):
            t = torch.from_numpy(np.concatenate([vec, [float(n)]]))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            a = t.numpy()
            return a[:-1], a[-1]

        cache = {"_sync_moments": moments, "_sync_allsum": allsum}
        soft, sig = u.forward(X[lo:hi], training=True, cache=cache)
        g = allreduce_mean_host(dist, u.backward(lab[lo:hi], cache))
        # loss / metric NUMERATORS and DENOMINATORS are what gets reduced, not the per-rank ratios
        y = R.one_hot(lab[lo:hi], 95)
        nums = np.array([R.wcce_loss(y, soft, 95.0).sum() * soft[0, ..., 0].size,     # sum over voxels of wcce
                         float(y[..., 0].size)])
        t = torch.from_numpy(nums); dist.all_reduce(t, op=dist.ReduceOp.SUM)
        np.savez(os.path.join(out_dir, "sync%d.npz" % rank), lsoft=t.numpy()[0] / t.numpy()[1],
                 bn_mean=cache["c13"]["mean"], bn_var=cache["c13"]["var"],
                 **{k.replace("/", "__"): v for k, v in g.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_syncbn_sharded_equals_full_batch(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_syncbn_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "sync0.npz"), np.load(tmp_path / "sync1.npz")
    X, lab, _ = R.synthetic_batch(GB, D, C, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    u = R.UnetOracle(in_ch=C, seed=1, lr=1e-3)
    cache = {}
    soft, sig = u.forward(X, training=True, cache=cache)            # one process, the whole batch
    g = u.backward(lab, cache)
    np.testing.assert_allclose(r0["bn_mean"], cache["c13"]["mean"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(r0["bn_var"], cache["c13"]["var"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(r0["lsoft"], R.wcce_loss(R.one_hot(lab, 95), soft, 95.0).mean(), rtol=1e-12)
    for k, v in g.items():
        kk = k.replace("/", "__")
        assert np.array_equal(r0[kk], r1[kk]), k
        scale = max(np.abs(v).max(), 1e-30)
        assert np.abs(r0[kk] - v).max() <= 1e-9 * scale, (k, np.abs(r0[kk] - v).max() / scale)
    # and WITHOUT the exchange the two-shard average differs (local BN is a different function of the weights)
    _, g0 = _shard_grads(*shard_range(GB, 0, 2))
    _, g1 = _shard_grads(*shard_range(GB, 1, 2))
    k = "c13/kernel"
    assert np.abs(0.5 * (g0[k] + g1[k]) - g[k]).max() > 1e-6 * np.abs(g[k]).max()
