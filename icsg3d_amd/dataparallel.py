"""Data-parallel host logic (new -- the reference is single-process).

One process per GPU.  The control plane (rendezvous, barriers, timing reduction, exchange of the
128-byte ncclUniqueId) runs over torch.distributed (gloo); the data path lives inside libicsg3d_hip.so
(csrc/engine.hip): the flat fp32 gradient buffer is all-reduced with RCCL in ~4 buckets, last layer first,
on a second HIP stream while the backward pass continues, and the 1/N scale is fused into Adam.
Default "local BN": each replica normalises with its own batch statistics (= the reference at its local
batch) and the BN moving statistics are averaged over the ranks every step; `sync_bn=True` exchanges the
per-channel statistics instead, so N x B grids normalise exactly like one process at N*B.
`init_engine_comm` also makes the replicas identical (rank 0's parameters, BN statistics and Adam state are
broadcast): the class API draws its initial weights from an unseeded RNG.
"""
from __future__ import annotations


def shard_range(global_batch: int, rank: int, world: int):
    """Contiguous, balanced shard [lo, hi) of a global batch (first `rem` ranks get one extra)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def exchange_unique_id(dist, rank: int, make_uid):
    """rank 0 creates the ncclUniqueId (bytes); everyone receives it over the control plane."""
    box = [make_uid() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = box[0]
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise RuntimeError("bad ncclUniqueId payload")
    return bytes(uid)


def init_engine_comm(engine, dist, rank: int, world: int, sync_bn: bool = False, force: bool = False):
    """Attach an RCCL communicator to an engine and synchronise the replicas: rank 0's parameters, BN moving
    statistics and Adam state overwrite everyone's.  No-op for world == 1 unless `force` (single-rank
    communicator: exercises the whole data-parallel path on a 1-GPU box)."""
    if world <= 1 and not force:
        return
    info = engine.comm_info()
    if info["nranks"] > 0:
        # idempotent: a second enable_data_parallel / a re-enable after train() finds the communicator in place
        if (info["nranks"], info["rank"]) != (world, rank):
            raise RuntimeError("engine already holds a communicator for rank %d of %d, asked for rank %d of %d"
                               % (info["rank"], info["nranks"], rank, world))
        engine.set_sync_bn(sync_bn)
        return
    from .engine import comm_unique_id
    engine.comm_init(rank, world, exchange_unique_id(dist, rank, comm_unique_id))
    engine.set_sync_bn(sync_bn)
    engine.broadcast_state(0)


def max_over_ranks(dist, value: float) -> float:
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class _Watched:
    def __init__(self, wd, where):
        self.wd, self.where = wd, where

    def __enter__(self):
        if self.wd is not None:
            self.wd.beat(self.where)

    def __exit__(self, *exc):
        if self.wd is not None:
            self.wd.pause(self.where + " (returned)")
        return False


class DataParallelMixin:
    """Data parallelism through the class API (AtomUnet / LatticeDFCVAE).  `enable_data_parallel` records the
    process group; the communicator is attached whenever the model (re)creates its engine -- all ranks do that
    at the same step because they see the same batch sizes.  Every rank runs the same number of steps (the
    caller shards the id lists to equal lengths: `shard_ids`); the loss/metric values a step returns are
    already reduced over the ranks, so every rank takes the same checkpoint decisions and only rank 0 writes."""

    _dp = None
    _wd = None

    def enable_data_parallel(self, dist, rank: int, world: int, sync_bn: bool = False, force: bool = False):
        self._dp = (dist, int(rank), int(world), bool(sync_bn), bool(force))
        if self._wd is None:
            # hang protection (icsg3d_amd/watchdog.py): a step that does not return within ICSG3D_WATCHDOG_S seconds ends
            # this rank with a message instead of holding every GPU of the job in a collective for ever
            from .watchdog import StepWatchdog
            self._wd = StepWatchdog(rank=int(rank))
            self._wd.pause("between steps")
        if getattr(self, "_eng", None) is not None:
            self._dp_attach(self._eng)
        return self

    def _dp_attach(self, engine):
        if self._dp is not None:
            dist, rank, world, sync_bn, force = self._dp
            if world > 1:
                # creating an engine is a collective (communicator + broadcast).  This check is itself a collective
                # (all_gather_object), so it catches exactly ONE case: every rank rebuilds its engine at the same
                # step but for DIFFERENT batch sizes -- they would otherwise average gradients of differently-shaped
                # batches or hang later in a mismatched collective.  A rank that rebuilds ALONE still blocks (here,
                # in the gather, instead of in ncclCommInitRank): that cannot be detected without a timeout, and the
                # callers avoid it by construction (shard_ids: every rank sees the same batch sizes at every step).
                sizes = [None] * world
                dist.all_gather_object(sizes, int(engine.max_batch))
                if len(set(sizes)) != 1:
                    raise RuntimeError("data parallel: ranks built engines for different batch sizes %s; shard the "
                                       "ids with shard_ids() so that every rank sees the same batches" % sizes)
            init_engine_comm(engine, dist, rank, world, sync_bn=sync_bn, force=force)

    def _dp_watch(self, where: str):
        """Context manager around ONE engine call of a data-parallel model: the watchdog runs while the call is in flight
        (that is where a collective can hang) and is paused between calls -- loading the next batch, a validation plot or
        the rest of the host program are not steps, and a process that merely stops training must not be ended."""
        return _Watched(self._wd, where)

    def _dp_is_writer(self) -> bool:
        return self._dp is None or self._dp[1] == 0

    def _dp_barrier(self):
        if self._dp is not None and self._dp[2] > 1:
            self._dp[0].barrier()


def shard_ids(ids, rank: int, world: int, batch_size: int):
    """This rank's share of an id list: ids[rank::world], every rank cut to the same whole number of
    batches (a rank with one batch more would wait for ever in the gradient all-reduce)."""
    n_batches = len(ids) // (world * batch_size)
    return list(ids[rank::world])[:n_batches * batch_size]


def from_env():
    """(dist, rank, world, local_rank) when launched under torch.distributed.run with WORLD_SIZE > 1 -- the
    control plane is initialised over gloo -- else None.  ICSG3D_FORCE_DP=1 takes the data-parallel path with a
    single rank too (a 1-GPU box can then exercise it end to end; pass force=True to enable_data_parallel)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ or (world <= 1 and os.environ.get("ICSG3D_FORCE_DP") != "1"):
        return None
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from . import _lib
    _lib.check(_lib.load().ics_set_device(local_rank))       # one process per GPU
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo")
    return dist, int(os.environ["RANK"]), world, local_rank
