"""Worker of tests/test_gpu_dp2.py: one process per GPU under torch.distributed.run.  Each rank trains on its half of
a 2B batch through the RCCL path; rank 0 also runs the whole 2B batch on a communicator-free engine and compares."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from icsg3d_amd import _lib
    _lib.check(_lib.load().ics_set_device(int(os.environ.get("LOCAL_RANK", "0"))))
    import torch.distributed as dist
    dist.init_process_group("gloo")
    # hang protection for the first run on real multi-GPU hardware: a collective that never completes ends this rank (exit 3)
    # after ICSG3D_WATCHDOG_S seconds without a beat instead of holding the lease (icsg3d_amd/watchdog.py)
    from icsg3d_amd.watchdog import StepWatchdog
    wd = StepWatchdog(rank=rank)
    wd.beat("start")
    from icsg3d_amd.dataparallel import init_engine_comm
    from icsg3d_amd.engine import UnetEngine
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes
    B, d = 2, 16
    P = glorot_params(unet_param_shapes(1, 95), 1)
    X, lab, _ = synthetic_batch(B * world, d, 1, seed=0, noise=1e-3)
    lo = rank * B
    out = {}
    for sync_bn in (True, False):
        eng = UnetEngine(in_channels=1, d=d, max_batch=B, lr=1e-3)
        # replicas start DIFFERENT on purpose: broadcast_state must make them rank 0's
        eng.set_weights(P if rank == 0 else glorot_params(unet_param_shapes(1, 95), 77))
        wd.beat("U-Net comm init (sync_bn=%s)" % sync_bn)
        init_engine_comm(eng, dist, rank, world, sync_bn=sync_bn)
        m = []
        for i in range(2):
            wd.beat("U-Net train step %d (sync_bn=%s: %s)" % (i, sync_bn, "two communicators in flight" if sync_bn else "buckets only"))
            m.append(eng.train_step(X[lo:lo + B], lab[lo:lo + B]))
        wd.beat("U-Net replica comparison (sync_bn=%s)" % sync_bn)
        w = eng.get_weights()
        g = {n: eng.get_grad(n, s) for n, s, tr in eng.tensor_infos() if tr}
        # every rank must hold the same averaged gradients, the same weights and the same BN moving statistics
        for k in sorted(w):
            mine = np.ascontiguousarray(w[k]).tobytes()
            box = [None] * world
            dist.all_gather_object(box, mine)
            assert all(b == box[0] for b in box), "replicas diverged in %s (sync_bn=%s)" % (k, sync_bn)
        if rank == 0:
            ref = UnetEngine(in_channels=1, d=d, max_batch=B * world, lr=1e-3)
            ref.set_weights(P)
            mr = [ref.train_step(X, lab) for _ in range(2)]
            gr = {n: ref.get_grad(n, s) for n, s, tr in ref.tensor_infos() if tr}
            if sync_bn:
                # N x B grids with exchanged statistics normalise exactly like one process at N*B: metrics of both
                # steps and the gradients of the second step agree to fp32 rounding
                err_m = max(float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)) for a, b in zip(m, mr))
                err_g = max(float(np.abs(g[k] - gr[k]).max() / max(np.abs(gr[k]).max(), 1e-12)) for k in g)
                # SyncBN + overlapped buckets = the two-communicator path (statistics on the split communicator while a
                # gradient bucket is in flight on the first): the buckets must really have been issued
                out["sync_bn"] = {"metrics_err": err_m, "grad_err": err_g, "buckets": eng.comm_info()["buckets_last_step"]}
            else:
                # local BN: each replica = the reference at its own batch; only the loss numerators are comparable
                out["local_bn"] = {"loss": float(m[0][0]), "loss_full_batch": float(mr[0][0]),
                                   "buckets": eng.comm_info()["buckets_last_step"]}
            ref.close()
        eng.close()
        dist.barrier()
    # ---- DFC-VAE: the same two statements for the second engine (its step runs on two streams; under SyncBN the perceptual
    # pass stays on the main stream; the gradient buckets wait for the side stream's weight gradients)
    from icsg3d_amd.engine import VaeEngine
    from icsg3d_amd.synthetic import vae_param_shapes
    PV = glorot_params(vae_param_shapes(1, d=d), 3)
    Xv, _, cond = synthetic_batch(B * world, d, 1, seed=1, noise=1e-3)
    eps = np.random.default_rng(2).standard_normal((B * world, 256)).astype(np.float32)
    for sync_bn in (True, False):
        pm = UnetEngine(in_channels=1, d=d, max_batch=B)
        pm.set_weights(P)
        ve = VaeEngine(pm, in_channels=1, d=d, max_batch=B, lr=5e-4)
        ve.set_weights(PV if rank == 0 else glorot_params(vae_param_shapes(1, d=d), 78))
        wd.beat("DFC-VAE comm init (sync_bn=%s)" % sync_bn)
        init_engine_comm(ve, dist, rank, world, sync_bn=sync_bn)
        m = []
        for i in range(2):
            wd.beat("DFC-VAE train step %d (sync_bn=%s)" % (i, sync_bn))
            m.append(ve.train_step(Xv[lo:lo + B], cond[lo:lo + B], eps[lo:lo + B]))
        wd.beat("DFC-VAE replica comparison (sync_bn=%s)" % sync_bn)
        w = ve.get_weights()
        g = {n: ve.get_grad(n, s) for n, s, tr in ve.tensor_infos() if tr}
        for k in sorted(w):
            box = [None] * world
            dist.all_gather_object(box, np.ascontiguousarray(w[k]).tobytes())
            assert all(b == box[0] for b in box), "VAE replicas diverged in %s (sync_bn=%s)" % (k, sync_bn)
        if rank == 0 and sync_bn:
            pmr = UnetEngine(in_channels=1, d=d, max_batch=B * world)
            pmr.set_weights(P)
            ref = VaeEngine(pmr, in_channels=1, d=d, max_batch=B * world, lr=5e-4)
            ref.set_weights(PV)
            mr = [ref.train_step(Xv, cond, eps) for _ in range(2)]
            gr = {n: ref.get_grad(n, s) for n, s, tr in ref.tensor_infos() if tr}
            gs = max(float(np.abs(v).max()) for v in gr.values())
            out["vae_sync_bn"] = {
                "metrics_err": max(float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)) for a, b in zip(m, mr)),
                # (conv biases in front of BatchNorm have an exactly-zero true gradient: their own scale is rounding noise)
                "grad_err": max(float(np.abs(g[k] - gr[k]).max() / max(np.abs(gr[k]).max(), 1e-4 * gs)) for k in g)}
            ref.close(); pmr.close()
        ve.close(); pm.close()
        dist.barrier()
    if rank == 0:
        print("DP2_RESULT " + json.dumps(out))
    wd.beat("teardown")
    dist.destroy_process_group()
    wd.stop()


if __name__ == "__main__":
    main()
