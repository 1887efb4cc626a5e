"""Data-parallel host logic (new -- the reference is single-process).

One process per GPU.  The control plane (rendezvous, barriers, timing reduction, exchange of the
128-byte ncclUniqueId) runs over torch.distributed (gloo); the data path is ONE RCCL all-reduce of
the engine's flat fp32 gradient buffer per step, issued inside libicsg3d_hip.so on the engine's own
stream (ics_net_comm_init / adam_step in csrc/engine.hip), followed by the 1/N scale fused into Adam.
Each replica normalises BatchNorm with its own 32-grid batch statistics ("local BN": every replica
is the reference at B=32; see DESIGN.md for the SyncBN caveat).
"""
from __future__ import annotations

import numpy as np


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


def init_engine_comm(engine, dist, rank: int, world: int):
    """Attach an RCCL communicator to an engine (no-op for world == 1)."""
    if world <= 1:
        return
    from .engine import comm_unique_id
    engine.comm_init(rank, world, exchange_unique_id(dist, rank, comm_unique_id))


def allreduce_mean_host(dist, arrays: dict):
    """Host-side reference of the gradient exchange (tests / debugging): mean over ranks of every
    array, via one flat all-reduce in a fixed key order -- the arithmetic the RCCL path performs."""
    import torch
    keys = sorted(arrays)
    flat = np.concatenate([np.asarray(arrays[k], np.float64).ravel() for k in keys])
    t = torch.from_numpy(flat)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    flat = t.numpy() / dist.get_world_size()
    out, o = {}, 0
    for k in keys:
        n = int(np.prod(np.shape(arrays[k])))
        out[k] = flat[o:o + n].reshape(np.shape(arrays[k]))
        o += n
    return out


def max_over_ranks(dist, value: float) -> float:
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
