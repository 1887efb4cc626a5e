"""world_size-2 gloo test of the data-parallel logic (one process per GPU in production; here two CPU
processes).  Each rank runs the train step of its shard (the oracle stands in for the engine), the
gradients are averaged with the same flat all-reduce arithmetic the RCCL path uses, every rank
applies Adam -- parameters must stay identical across ranks and equal the single-process result
computed from both shards (local-BN data parallelism)."""
import os
import socket

import numpy as np
import pytest

from icsg3d_amd.dataparallel import allreduce_mean_host, exchange_unique_id, max_over_ranks, shard_range
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
