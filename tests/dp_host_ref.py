"""Host-side references of the data-parallel exchanges (test infrastructure): the arithmetic the RCCL path inside
libicsg3d_hip.so performs, written with torch.distributed (gloo) so that world-size-2 CPU tests can run it against the
oracle.  Not part of the product package."""
import numpy as np


def syncbn_moments(dist, x):
    """Host reference of the SyncBN exchange (tests): per-channel (mean, biased var) of the GLOBAL batch from
    each rank's shard x (..., C): all-gather (n, mean, M2) and merge in rank order (Chan et al.) -- the same
    arithmetic bn_local_merge / bn_sync_finalize perform on the device."""
    import torch
    xf = np.asarray(x, np.float64).reshape(-1, np.shape(x)[-1])
    n = float(xf.shape[0])
    mean = xf.mean(0)
    m2 = ((xf - mean) ** 2).sum(0)
    local = torch.from_numpy(np.concatenate([[n], mean, m2]))
    parts = [torch.zeros_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, local)
    C = xf.shape[1]
    N, mu, M2 = 0.0, np.zeros(C), np.zeros(C)
    for p in parts:
        p = p.numpy()
        nb, mb, qb = p[0], p[1:1 + C], p[1 + C:]
        nt = N + nb
        dlt = mb - mu
        M2 = M2 + qb + dlt * dlt * N * nb / nt
        mu = mu + dlt * nb / nt
        N = nt
    return mu, M2 / N, N


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


