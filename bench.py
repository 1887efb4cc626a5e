#!/usr/bin/env python3
"""bench.py -- ICSG3D hot-path benchmark on MI355X (contract: see DESIGN.md "Measurement").

One "step" = one U-Net training step (forward, loss, backward, BN moving-stat update, Adam) on a
batch of 32 synthetic 32^3 x 1 voxel grids per GPU, inputs resident in HBM when the timed region
starts (BASELINE.json configs[1]; weak scaling over N GPUs with one RCCL all-reduce of the flat
gradient buffer per step).  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_HBM_GBS = 8000.0         # HBM3E spec
# algorithmic work of one U-Net train step per grid (C=1, d=32)
UNET_FLOP_PER_GRID = 377.66e9          # fwd 125.886 GFLOP x 3
UNET_BYTES_PER_GRID = 499.6e6          # fused-minimum activation traffic
UNET_PARAM_BYTES_PER_STEP = 1.25e9     # 124.6 MB x (1 fwd + 2 bwd + 7 Adam)


def cpu_baseline(sample_grids=4):
    """oracle/torch_ref.py (fp32 torch-CPU restatement of the same train step, all host cores) in a
    subprocess -- the checker timed as a baseline, never the product path."""
    code = ("import json,sys; sys.path.insert(0, %r); from oracle import torch_ref as T; "
            "v,c,s = T.time_unet_train_step(B=%d, d=32, in_ch=1, steps=2, warmup=1); "
            "print(json.dumps({'value': v, 'cores': c, 'sec_per_step': s}))" % (ROOT, sample_grids))
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        r = json.loads(out.stdout.strip().splitlines()[-1])
        return {"value": round(r["value"], 4), "unit": "voxel-grids/s", "cores": int(r["cores"]), "kind": "port",
                "sample": "oracle/torch_ref.py fp32 U-Net fwd+bwd+Adam on %d synthetic 32^3 grids/step, "
                          "1 warm-up + 2 timed steps (%.1f s/step), torch-CPU channels_last_3d; the reference's "
                          "Keras/TF path is not installable here" % (sample_grids, r["sec_per_step"])}
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "voxel-grids/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "cpu baseline failed: %s" % e}


def run_vae_or_joint(args, rank, local_rank, world, use_dist, dist, json_fd):
    """Secondary workloads (not the driver's default line): --workload vae = BASELINE configs[2] (DFC-VAE step with
    the frozen perceptual U-Net), --workload joint = one U-Net step + one DFC-VAE step per iteration on the same
    grids (configs[3]/[4] shape: both engines resident, both gradient all-reduces per iteration)."""
    from icsg3d_amd.engine import UnetEngine, VaeEngine, comm_unique_id
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
    B, d, C = args.batch, args.d, 1
    joint = args.workload == "joint"
    PU = glorot_params(unet_param_shapes(C, 95), seed=1)
    pm = UnetEngine(in_channels=C, d=d, max_batch=B); pm.set_weights(PU)
    vae = VaeEngine(pm, in_channels=C, d=d, max_batch=B, lr=5e-4)
    vae.set_weights(glorot_params(vae_param_shapes(C, d=d), seed=3))
    X, labels, cond = synthetic_batch(B, d, C, seed=rank)
    eps = np.random.default_rng(2 + rank).standard_normal((B, 256)).astype(np.float32)
    vae.upload_batch(X, cond, eps)
    engines = [vae]
    unet = None
    if joint:
        unet = UnetEngine(in_channels=C, d=d, max_batch=B, lr=3e-6); unet.set_weights(PU)
        unet.upload_batch(X, labels)
        engines.append(unet)
    if use_dist:
        for e in engines:
            uid = [comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            e.comm_init(rank, world, uid[0])

    def step(m=False):
        if unet is not None:
            unet.train_step_resident(False)
        return vae.train_step_resident(m)

    def barrier():
        for e in engines:
            e.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    for e in engines + [pm]:
        e.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rows = [r for e in engines + [pm] for r in e.profile_rows()]
    metrics = step(True)
    if not np.all(np.isfinite(metrics)):
        raise SystemExit("non-finite training metrics: %s" % metrics)
    if rank == 0:
        by_kernel = {}
        for r in rows:
            kid = r["label"].split("|")[1] if "|" in r["label"] and r["label"].split("|")[1] else r["label"]
            a = by_kernel.setdefault(kid, {"ms": 0.0, "flop": 0.0, "launches": 0})
            a["ms"] += r["ms"]; a["flop"] += r["flop"]; a["launches"] += r["launches"]
        dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
        achieved = dom["flop"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        exec_flop = sum(r["flop"] for r in rows) / args.steps
        ms_per_step = elapsed / args.steps * 1e3
        out = {"metric": "voxel-grids/s (fwd+bwd) for %d^3 %s at batch %d per GPU"
                         % (d, "U-Net step + DFC-VAE step" if joint else "DFC-VAE step", B),
               "value": round(world * B * args.steps / elapsed, 2), "unit": "voxel-grids/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s, %d x %d^3 x 1 grids per GPU" % (args.workload, B, d),
                          "global_batch": world * B, "grid": d, "parallelism": "dp%d" % world},
               "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 2),
                            "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": None,
                            "launches": dom["launches"]},
               "roofline_step": {"executed_tflop_per_step": round(exec_flop / 1e12, 3),
                                 "compute_frac": round(exec_flop / (ms_per_step * 1e-3) / (PEAK_FP32_TFLOPS * 1e12), 4)},
               "cpu_baseline": None}
        line = json.dumps(out) + "\n"
        if json_fd is not None:
            sys.stdout.flush(); os.write(json_fd, line.encode())
        else:
            sys.stdout.write(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="grids per GPU")
    ap.add_argument("--d", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=("unet", "vae", "joint"), default="unet",
                    help="unet = the contract line (BASELINE configs[1]); vae / joint = secondary measurements")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # the engine binds system ROCm; load it before anything else can pull in another HIP runtime
    from icsg3d_amd import _lib
    from icsg3d_amd.engine import UnetEngine, comm_unique_id
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes
    lib = _lib.load()
    _lib.check(lib.ics_set_device(local_rank))

    # ICSG3D_BENCH_FORCE_DIST=1 takes the multi-process path (gloo rendezvous, ncclUniqueId hand-off, RCCL
    # communicator, all-reduce inside Adam) even with one rank: the only way to exercise it on a 1-GPU box
    use_dist = world > 1 or (os.environ.get("ICSG3D_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    dist = None
    json_fd = None
    if use_dist:
        # RCCL prints its version banner on STDOUT from ncclCommInitRank; the contract is ONE JSON line there.
        # Everything else this process writes to fd 1 goes to stderr; the JSON line is written to the saved fd.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        import torch  # noqa: F401  (control plane only: gloo rendezvous, barrier, max-reduce)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    if args.workload != "unet":
        return run_vae_or_joint(args, rank, local_rank, world, use_dist, dist, json_fd)

    B, d, C = args.batch, args.d, 1
    eng = UnetEngine(in_channels=C, num_classes=95, d=d, max_batch=B, lr=3e-6)
    eng.set_weights(glorot_params(unet_param_shapes(C, 95), seed=1))
    X, labels, _ = synthetic_batch(B, d, C, seed=rank)   # each rank its own shard of the global batch
    eng.upload_batch(X, labels)

    if use_dist:
        import torch
        uid = [comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(rank, world, uid[0])

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        eng.train_step_resident(False)
    barrier()
    eng.profile_enable(True)     # HIP events around every launch on the engine's stream
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step_resident(False)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rows = eng.profile_rows()
    metrics = eng.train_step_resident(True)   # untimed: sanity that the job is still finite
    if not np.all(np.isfinite(metrics)):
        raise SystemExit("non-finite training metrics: %s" % metrics)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        # dominant kernel = the instantiation with the largest share of device time
        by_kernel = {}
        for r in rows:
            kid = r["label"].split("|")[1] if "|" in r["label"] and r["label"].split("|")[1] else r["label"]
            a = by_kernel.setdefault(kid, {"ms": 0.0, "flop": 0.0, "launches": 0, "bytes": 0.0})
            a["ms"] += r["ms"]; a["flop"] += r["flop"]; a["launches"] += r["launches"]; a["bytes"] += r["bytes"]
        dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
        achieved = dom["flop"] / (dom["ms"] * 1e-3) / 1e12
        scale = (d / 32.0) ** 3
        step_flop = UNET_FLOP_PER_GRID * scale * B
        exec_flop = sum(r["flop"] for r in rows) / args.steps
        step_bytes = UNET_BYTES_PER_GRID * scale * B + UNET_PARAM_BYTES_PER_STEP
        # HBM bytes per launch of that kernel from the committed PMC passes (rocprofv3 --pmc cannot run
        # inside this process); null when the profile does not cover the dominant kernel
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
                traffic = json.load(f)["kernels"][dom_name]["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "voxel-grids/s (fwd+bwd) for 32^3 U-Net at batch 32 per GPU",
            "value": round(value, 2), "unit": "voxel-grids/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "AtomUnet fwd+bwd+Adam train step, %d x %d^3 x 1 grids per GPU "
                                   "(BASELINE.json configs[1]), Glorot weights PCG64(1)" % (B, d),
                       "global_batch": world * B, "grid": d, "parallelism": "dp%d" % world,
                       "bn": "local per-replica batch statistics"},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 2),
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC: 2*FETCH_SIZE+WRITE_SIZE, profiles/r1_pmc_traffic.json)",
                         "algorithmic_bytes_per_launch": int(dom["bytes"] / max(dom["launches"], 1)),
                         "launches": dom["launches"],
                         "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
                         "share_of_device_time": round(dom["ms"] / sum(v["ms"] for v in by_kernel.values()), 4)},
            # executed = the FLOPs of the GEMMs actually launched (the [skip | upsampled] convs run their
            # upsampled channels on the low-res grid: 8/27 of the direct count, an exact reassociation);
            # direct = the textbook 27-tap count of the reference graph (fwd x 3)
            "roofline_step": {"executed_tflop_per_step": round(exec_flop / 1e12, 3),
                              "compute_frac": round(exec_flop / (ms_per_step * 1e-3) / (PEAK_FP32_TFLOPS * 1e12), 4),
                              "direct_conv_tflop_per_step": round(step_flop / 1e12, 3),
                              "direct_conv_equivalent_tflops": round(step_flop / (ms_per_step * 1e-3) / 1e12, 2),
                              "hbm_frac": round(step_bytes / (ms_per_step * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                              "algorithmic_gb_per_step": round(step_bytes / 1e9, 3)},
            "kernels": {k: {"ms_per_step": round(v["ms"] / args.steps, 3),
                            "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 and v["flop"] > 0 else None}
                        for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])[:8]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
