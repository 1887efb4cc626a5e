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
        init_engine_comm(eng, dist, rank, world, sync_bn=sync_bn)
        m = [eng.train_step(X[lo:lo + B], lab[lo:lo + B]) for _ in range(2)]
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
                out["sync_bn"] = {"metrics_err": err_m, "grad_err": err_g}
            else:
                # local BN: each replica = the reference at its own batch; only the loss numerators are comparable
                out["local_bn"] = {"loss": float(m[0][0]), "loss_full_batch": float(mr[0][0]),
                                   "buckets": eng.comm_info()["buckets_last_step"]}
            ref.close()
        eng.close()
        dist.barrier()
    if rank == 0:
        print("DP2_RESULT " + json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
