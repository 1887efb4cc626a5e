#!/usr/bin/env python3
"""bench.py -- ICSG3D hot-path benchmark on MI355X (contract: see DESIGN.md "Measurement").

One "step" = one U-Net training step (forward, loss, backward, BN moving-stat update, Adam) on a
batch of 32 synthetic 32^3 x 1 voxel grids per GPU, inputs resident in HBM when the timed region
starts (BASELINE.json configs[1]; weak scaling over N GPUs, gradients all-reduced with RCCL in buckets
overlapped with the backward pass).  The same JSON line carries a `secondary` block for the other half
of BASELINE.json's metric, the DFC-VAE step (configs[2]), timed right after the U-Net region.
Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...        (no launcher: starts the line above as a child process before touching the GPU,
                                       relays its JSON line and exit status; never prints an n_gpus: 1 line for N > 1)
  python bench.py --workload predict  (BASELINE configs[0]: U-Net forward-only, 16 grids)
  python bench.py --workload generate (generate.py:204-236: decoder -> U-Net -> labels -> atoms on the device)
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
WINO_EXEC = 64.0 / 216.0      # executed / algorithmic multiplies of the Winograd F(2x2x2, 3x3x3) kernels


def wino(kernel_name):
    """executed / algorithmic FLOPs of a kernel row: the Winograd kernels' rows carry the 27-tap count."""
    return WINO_EXEC if "conv_wino" in kernel_name else 1.0
PEAK_HBM_GBS = 8000.0         # HBM3E spec
# algorithmic work of one U-Net train step per grid (C=1, d=32)
UNET_FLOP_PER_GRID = 377.66e9          # fwd 125.886 GFLOP x 3
UNET_BYTES_PER_GRID = 499.6e6          # fused-minimum activation traffic
UNET_PARAM_BYTES_PER_STEP = 1.25e9     # 124.6 MB x (1 fwd + 2 bwd + 7 Adam)
VAE_FLOP_PER_GRID = 33.73e9            # VAE 3 x 2.126 + perceptual 3 x 9.116 GFLOP
PMC_TRAFFIC_FILES = ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json")


def cpu_baseline(batch=32, d=32):
    """oracle/torch_ref.py (fp32 torch-CPU restatement of the same train steps, all host cores) in a subprocess -- the
    checker timed as a baseline, never the product path.  The sample is the GPU figure's own batch.  Every leg gets one
    untimed warm-up call of its own kind first (oneDNN primitives, allocator, thread pools, scipy import: ADVICE r5), then
    the MEDIAN of 3 timed U-Net train steps at `batch` grids (`value`) and of 3 timed DFC-VAE train steps (`vae`), and of 2
    timed calls for the two inference legs; min / max are reported next to each median.
    ~60 s + ~20 s + ~15 s on the GPU box's 128 cores; bounded by a 1500 s timeout."""
    code = ("import json,sys; sys.path.insert(0, %r); from oracle import torch_ref as T; "
            "r = {}; "
            "T.time_unet_train_step(B=8, d=%d, in_ch=1, steps=1, warmup=0); "
            "v,c,s = T.time_unet_train_step(B=%d, d=%d, in_ch=1, steps=3, warmup=0); r['u'] = T.time_unet_train_step.samples; "
            "T.time_vae_train_step(B=8, d=%d, in_ch=1, steps=1, warmup=0); "
            "vv,c,sv = T.time_vae_train_step(B=%d, d=%d, in_ch=1, steps=3, warmup=0); r['v'] = T.time_vae_train_step.samples; "
            "vp,c,sp = T.time_unet_predict(B=16, d=%d, in_ch=1, steps=2, warmup=1); r['p'] = T.time_unet_predict.samples; "
            "vg,c,sg = T.time_generate_tail(B=%d, d=%d, in_ch=1, steps=2, warmup=1, full_samples=8); r['g'] = T.time_generate_tail.samples; r['gf'] = T.time_generate_tail.full; "
            "r.update({'v_': v, 's': s, 'vv': vv, 'sv': sv, 'vp': vp, 'sp': sp, 'vg': vg, 'sg': sg, 'cores': c}); "
            "print(json.dumps(r))"
            % (ROOT, d, batch, d, d, batch, d, d, batch, d))
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
        r = json.loads(out.stdout.strip().splitlines()[-1])

        def leg(value, sec, samples, key):
            return {"value": round(value, 4), "unit": "voxel-grids/s", key: round(sec, 2), "samples": len(samples),
                    "s_min": round(min(samples), 2), "s_max": round(max(samples), 2)}
        u = leg(r["v_"], r["s"], r["u"], "s_per_step")
        return {"value": u["value"], "unit": "voxel-grids/s", "cores": int(r["cores"]), "kind": "port",
                "s_per_step": u["s_per_step"], "samples": u["samples"], "s_min": u["s_min"], "s_max": u["s_max"],
                "vae": leg(r["vv"], r["sv"], r["v"], "s_per_step"),
                "predict": leg(r["vp"], r["sp"], r["p"], "s_per_call"),
                "generate": dict(leg(r["vg"], r["sg"], r["g"], "s_per_call"),
                                 # the host step the GPU block reports as `refine`: hull test of every kept component +
                                 # recursive marker watershed, oracle/watershed_ref.py one grid at a time (pure-Python heap)
                                 refine=None if not r.get("gf") else {
                                     "grids": r["gf"]["grids"], "s_per_grid": round(r["gf"]["s_per_grid"], 4),
                                     "grids_per_s": round(1.0 / r["gf"]["s_per_grid"], 2), "failed": r["gf"]["failed"],
                                     "what": "oracle/watershed_ref.watershed_clustering (convex-hull test + marker watershed "
                                             "+ recursion, watershed.py:190-203) on the first grids of the same call's masks"}),
                "sample": "oracle/torch_ref.py fp32, torch-CPU channels_last_3d, all host cores; each leg after one untimed "
                          "warm-up call of its own kind: MEDIAN of 3 timed U-Net fwd+bwd+Adam steps on %d synthetic %d^3 grids "
                          "(%.1f s, min %.1f / max %.1f; `value`, the batch the GPU figure is quoted on), median of 3 timed "
                          "DFC-VAE train steps on %d grids (%.1f s; `vae`), median of 2 U-Net forwards on 16 grids (%.1f s; "
                          "`predict`, BASELINE configs[0]) and of 2 generate tails on %d latent vectors (decoder + U-Net forward "
                          "+ argmax / threshold + scipy.ndimage components, threshold fixed beforehand, %.1f s; `generate`).  "
                          "The reference's Keras/TF path is not installable here"
                          % (batch, d, r["s"], min(r["u"]), max(r["u"]), batch, r["sv"], r["sp"], batch, r["sg"])}
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "voxel-grids/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "cpu baseline failed: %s" % e}


def by_kernel(rows):
    """engine profile rows -> {kernel instantiation (the name rocprofv3 prints): ms / flop / bytes / launches}"""
    acc = {}
    for r in rows:
        kid = r["label"].split("|")[1] if "|" in r["label"] and r["label"].split("|")[1] else r["label"]
        a = acc.setdefault(kid, {"ms": 0.0, "flop": 0.0, "launches": 0, "bytes": 0.0})
        a["ms"] += r["ms"]; a["flop"] += r["flop"]; a["launches"] += r["launches"]; a["bytes"] += r["bytes"]
    return acc


def pmc_traffic(kernel, live_avg_ms):
    """HBM bytes per launch of `kernel` from the committed PMC passes (rocprofv3 --pmc cannot run inside this
    process).  The file records the kernel's average duration in the profiled run; if this run's live
    HIP-event average disagrees by more than 15 % the kernel has changed since and the figure is stale -> null
    with the reason, never a silently outdated number."""
    for name in PMC_TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as f:
                e = json.load(f)["kernels"].get(kernel)
        except Exception as ex:
            return None, "unreadable %s: %s" % (name, ex)
        if e is None or "traffic_bytes_per_launch" not in e:
            return None, "profiles/%s has no PMC pass for this kernel" % name
        ref_ms = e.get("avg_ms")
        if ref_ms and abs(live_avg_ms - ref_ms) > 0.15 * ref_ms:
            return None, "stale: profiles/%s measured %.3f ms/launch, this run %.3f" % (name, ref_ms, live_avg_ms)
        note = "profiles/%s (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes)" % name
        if not ref_ms:
            note += "; file carries no avg_ms, staleness unchecked"
        return e["traffic_bytes_per_launch"], note
    return None, "no PMC traffic profile committed"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="grids per GPU")
    ap.add_argument("--d", "--dim", dest="d", type=int, default=32,
                    help="grid edge (torch.distributed.run's own parser rejects --d as an ambiguous abbreviation: the "
                         "self-launch forwards it as --dim)")
    ap.add_argument("--soak-seconds", type=float, default=10.0,
                    help="after the timed region: the same U-Net step for at least this long, reported as `sustained` "
                         "(never `value`); 0 skips it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the DFC-VAE block (profiling runs)")
    ap.add_argument("--workload", choices=("unet", "vae", "joint", "predict", "generate"), default="unet",
                    help="unet = the contract line (BASELINE configs[1] + the configs[2] secondary block + the two inference "
                         "blocks); vae / joint = profiling runs of the DFC-VAE step alone / U-Net + DFC-VAE step per iteration; "
                         "predict = configs[0] (U-Net forward-only, 16 grids); generate = decoder -> U-Net -> labels -> atoms")
    ap.add_argument("--no-inference", action="store_true", help="skip the predict / generate blocks of the contract line")
    ap.add_argument("--sync-bn", action="store_true", help="data parallel: global-batch BatchNorm statistics")
    ap.add_argument("--dump-rows", type=str, default=None, help="write the per-launch-site profile rows (JSON) here")
    args = ap.parse_args()

    # --gpus N > 1 without a launcher: become the launcher BEFORE any HIP call (icsg3d_amd/launcher.py); a mismatch
    # between --gpus and WORLD_SIZE, or fewer than N visible GPUs, exits non-zero -- never an n_gpus: 1 line for --gpus 8
    from icsg3d_amd.launcher import ensure_ranks
    ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # hang protection: no beat for ICSG3D_WATCHDOG_S (120) seconds -> this rank says where it is and exits 3; under
    # torch.distributed.run that ends the job with a non-zero status instead of holding the node (icsg3d_amd/watchdog.py)
    from icsg3d_amd.watchdog import StepWatchdog
    wd = StepWatchdog()
    wd.beat("library load / device %d" % local_rank)

    # the engine binds system ROCm; load it before anything else can pull in another HIP runtime
    from icsg3d_amd import _lib
    from icsg3d_amd.dataparallel import init_engine_comm
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
    lib = _lib.load()
    _lib.check(lib.ics_set_device(local_rank))

    # ICSG3D_BENCH_FORCE_DIST=1 takes the multi-process path (gloo rendezvous, ncclUniqueId hand-off, RCCL
    # communicator, state broadcast, bucketed all-reduce on the comm stream) even with one rank: the only way
    # to exercise it on a 1-GPU box
    force = os.environ.get("ICSG3D_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    use_dist = world > 1 or force
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
        wd.beat("gloo rendezvous")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    B, d, C = args.batch, args.d, 1
    X, labels, cond = synthetic_batch(B, d, C, seed=rank)   # each rank its own shard of the global batch
    PU = glorot_params(unet_param_shapes(C, 95), seed=1)

    def max_over_ranks(v):
        if dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(step, engines, profiled, step_profile=None):
        """W warm-up steps, then EXACTLY K steps between barrier + device sync on both sides; max over ranks.
        step_profile: the step of the untimed all-events pass when it must differ (joint: the two engines serialised, so
        that a kernel's event bracket does not time a kernel of the other stream it shares the chip with)."""
        def barrier(where="barrier"):
            wd.beat(where + ": device sync")
            for e in engines:
                e.sync()
            if dist is not None:
                wd.beat(where + ": gloo barrier")
                dist.barrier()
            wd.beat(where + ": done")
        wd.beat("warm-up steps")
        for _ in range(args.warmup):
            step()
        barrier("after warm-up")
        # untimed profiling pass: HIP events around EVERY launch (on the stream it is launched on) -> the per-kernel
        # table and the dominant kernel with its launch sites
        for e in profiled:
            e.profile_filter("")
            e.profile_enable(True)
        for _ in range(2):
            (step_profile or step)()
        barrier("after the all-events pass")
        rows_all = [r for e in profiled for r in e.profile_rows()]
        for r in rows_all:
            r["ms"] /= 2.0; r["flop"] /= 2.0; r["bytes"] /= 2.0; r["launches"] //= 2     # per step
        gem = {}
        for r in rows_all:
            if r["flop"] > 0 and "|" in r["label"]:
                gem.setdefault(r["label"].split("|", 1)[1], []).append(r)
        prefix = ""
        if gem:
            dom_rows = max(gem.values(), key=lambda rs: sum(r["ms"] for r in rs))
            # exactly the launch sites the dominant kernel ran at (';'-separated label prefixes, each up to its '|')
            prefix = ";".join(sorted({r["label"].split("|", 1)[0] + "|" for r in dom_rows}))
        # timed region: events only around the dominant kernel's launch sites (its live average duration is the
        # roofline's denominator); every other launch runs as in a training job
        for e in profiled:
            e.profile_filter(prefix)
            e.profile_enable(True)
        barrier("before the timed steps")
        t0 = time.perf_counter()
        for e in engines:
            e.timer_start()              # an event on each engine's stream: the GPU's own clock over the same K steps
        for _ in range(args.steps):
            step()
        wd.beat("timed steps enqueued: waiting for the device")
        gpu_ms = max(e.timer_stop() for e in engines)
        barrier("after the timed steps")
        elapsed = max_over_ranks(time.perf_counter() - t0)
        timed.gpu_active_s = gpu_ms * 1e-3
        rows = [r for e in profiled for r in e.profile_rows()]
        for e in profiled:
            e.profile_enable(False)
            e.profile_filter("")
        for r in rows_all:          # per step -> the K steps' totals the tables below divide down again
            r["ms"] *= args.steps; r["flop"] *= args.steps; r["bytes"] *= args.steps; r["launches"] *= args.steps
        if args.dump_rows and rank == 0:
            with open(args.dump_rows, "a") as f:
                f.write(json.dumps({"steps": args.steps, "rows": rows}) + "\n")
        # the same K steps without the per-launch events (what a training job sees); the library counts every kernel
        # it enqueues (ics_kernel_launches): the per-step difference is the number rocprofv3 --kernel-trace shows
        barrier("before the events-off steps")
        n0 = lib.ics_kernel_launches()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier("after the events-off steps")
        elapsed_plain = max_over_ranks(time.perf_counter() - t0)
        timed.kernel_launches_per_step = (lib.ics_kernel_launches() - n0) / float(args.steps)
        return elapsed, elapsed_plain, rows, rows_all

    def soak(step, engines, ms_per_step):
        """the timed region is 20 steps = 0.56 s -- too short for the driver's GPU sampler to see and too
        short to say what the chip does after ten seconds at 100 TFLOP/s.  The same step, events off, for >= --soak-seconds,
        between the same barriers, on the host clock and on the GPU's own; reported as `sustained`, never as `value`."""
        n = max(args.steps, int(1.03 * args.soak_seconds / max(ms_per_step * 1e-3, 1e-6)) + 2)   # 3 % margin: at least soak_seconds
        wd.beat("sustained soak: %d steps" % n)
        for e in engines:
            e.sync()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for e in engines:
            e.timer_start()
        for i in range(n):
            step()
            if i % 64 == 63:
                wd.beat("sustained soak: step %d of %d enqueued" % (i + 1, n))
        gpu_ms = max(e.timer_stop() for e in engines)
        for e in engines:
            e.sync()
        if dist is not None:
            wd.beat("sustained soak: gloo barrier")
            dist.barrier()
        sec = max_over_ranks(time.perf_counter() - t0)
        wd.beat("sustained soak done")
        return {"steps": n, "seconds": round(sec, 3), "ms_per_step": round(sec / n * 1e3, 3),
                "gpu_active_s": round(gpu_ms * 1e-3, 4), "value": round(world * B * n / sec, 2), "unit": "voxel-grids/s",
                "note": "the same train step, no events, run back to back for >= %.0f s after the timed region "
                        "(sustained clocks; visible to a coarse GPU-utilisation sampler); not `value`" % args.soak_seconds}

    out = None
    unet = None
    if args.workload in ("unet", "joint"):
        unet = UnetEngine(in_channels=C, num_classes=95, d=d, max_batch=B, lr=3e-6)
        unet.set_weights(PU)
        unet.upload_batch(X, labels)
        if use_dist:
            init_engine_comm(unet, dist, rank, world, sync_bn=args.sync_bn, force=force)
            got = unet.comm_info()["nranks"]
            if got != world:
                raise SystemExit("RCCL communicator has %d ranks, --gpus %d" % (got, world))

    if args.workload == "unet":
        elapsed, elapsed_plain, rows_live, rows = timed(lambda: unet.train_step_resident(False), [unet], [unet])
        unet_launches = timed.kernel_launches_per_step
        gpu_active_s = timed.gpu_active_s
        sustained = soak(lambda: unet.train_step_resident(False), [unet], elapsed_plain / args.steps * 1e3) \
            if args.soak_seconds > 0 else None
        metrics = unet.train_step_resident(True)   # untimed: sanity that the job is still finite
        if not np.all(np.isfinite(metrics)):
            raise SystemExit("non-finite training metrics: %s" % metrics)
        comm = unet.comm_info()
        if rank == 0:
            ms_per_step = elapsed / args.steps * 1e3
            value = world * B * args.steps / elapsed
            kern = by_kernel(rows)
            gemms = {k: v for k, v in kern.items() if v["flop"] > 0}
            dom_name, dom_all = max(gemms.items(), key=lambda kv: kv[1]["ms"])   # largest share of device time (profiling pass)
            dom = by_kernel(rows_live).get(dom_name, dom_all)   # its launches inside the TIMED region (events on its sites only)
            achieved = dom["flop"] / (dom["ms"] * 1e-3) / 1e12
            avg_ms = dom["ms"] / max(dom["launches"], 1)
            traffic, traffic_note = pmc_traffic(dom_name, avg_ms)
            scale = (d / 32.0) ** 3
            step_flop = UNET_FLOP_PER_GRID * scale * B
            # the profile rows of the Winograd kernels carry the ALGORITHMIC (27-tap) FLOPs of the convolution they
            # compute; the MFMAs they execute are 64/216 of that (F(2x2x2, 3x3x3))
            exec_flop = sum(v["flop"] * wino(k) for k, v in kern.items()) / args.steps
            step_bytes = UNET_BYTES_PER_GRID * scale * B + UNET_PARAM_BYTES_PER_STEP
            out = {
                # BASELINE.json's metric names both nets; `value` is configs[1] (the U-Net step, the configuration the
                # contract's N=1 line is quoted on), the DFC-VAE step of configs[2] is `secondary`, both per batch `unet_plus_vae`
                "metric": "voxel-grids/s (fwd+bwd) for 32^3 U-Net+VAE at batch 32: value = U-Net train step "
                          "(BASELINE configs[1]); DFC-VAE step (configs[2]) in secondary; both per batch in unet_plus_vae",
                "value": round(value, 2), "unit": "voxel-grids/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                # ranks of the RCCL communicator the gradient all-reduce ran on (0: single process, no communicator)
                "rccl_ranks": comm["nranks"],
                # the K timed steps on the GPU's own clock (HIP events on the engine's stream around the same region):
                # ms_per_step must agree with gpu_active_s / steps, or the host clock timed something else
                "gpu_active_s": round(gpu_active_s, 5),
                "sustained": sustained,
                "config": {"workload": "AtomUnet fwd+bwd+Adam train step, %d x %d^3 x 1 grids per GPU "
                                       "(BASELINE.json configs[1]; the metric's VAE half is timed right after as "
                                       "`secondary` = configs[2], and `unet_plus_vae` is one U-Net step + one DFC-VAE step "
                                       "per batch), Glorot weights PCG64(1)" % (B, d),
                           "global_batch": world * B, "grid": d, "parallelism": "dp%d" % world,
                           "conv": "3x3x3 layers: Winograd F(2x2x2,3x3x3) on fp32 MFMA (fwd, bwd-data, bwd-weight); "
                                   "upsampled channels: 27 three-point-transform products per low-res voxel forward, tap-pooled GEMMs on the coarse grid backward; rest: 27-tap implicit GEMM",
                           "bn": "sync (global-batch statistics)" if args.sync_bn and use_dist
                                 else "local per-replica batch statistics, moving statistics averaged over ranks",
                           "grad_allreduce": ("%d RCCL buckets per step on a second stream, overlapped with the "
                                              "backward pass" % comm["buckets_last_step"]) if comm["nranks"] else "none"},
                # value / ms_per_step: the K timed steps, with HIP events around the launch sites of the dominant kernel only
                # (the roofline's denominator is measured over exactly those steps; the per-kernel table comes from an
                # untimed pass with events around every launch); the same K steps with no events at all:
                "ms_per_step_events_off": round(elapsed_plain / args.steps * 1e3, 3),
                "value_events_off": round(world * B * args.steps / elapsed_plain, 2),
                # device kernels the library enqueues per step (counted at every launch site; = the ics:: rows of
                # rocprofv3 --kernel-trace for the same command); profiler_brackets = timed launch SITES, several
                # kernels can sit inside one
                "kernel_launches_per_step": round(unet_launches, 1),
                "profiler_brackets_per_step": round(sum(v["launches"] for v in kern.values()) / args.steps, 1),
                # achieved / frac: the multiply-adds the kernel ISSUES to the matrix cores per second over the fp32 MFMA
                # peak (<= 1 by construction: the matrix-core utilisation).  The Winograd kernels' profile rows carry
                # the 27-tap count of the convolution they compute (2*S^3*27*Cin*Cout); they execute 64/216
                # of it, so their rate in the reference graph's own FLOPs (algorithmic_equivalent_tflops) can exceed the peak
                "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved * wino(dom_name), 2),
                             "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(achieved * wino(dom_name) / PEAK_FP32_TFLOPS, 4),
                             "executed_over_algorithmic": round(wino(dom_name), 4),
                             "algorithmic_equivalent_tflops": round(achieved, 2),
                             "note": ("Winograd F(2x2x2,3x3x3): 64 multiplies per 2x2x2 output tile instead of 216; achieved / "
                                      "frac count the MFMA work executed, algorithmic_equivalent_tflops the direct "
                                      "convolution's 27-tap FLOPs per second") if wino(dom_name) != 1.0 else None,
                             "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_note,
                             "algorithmic_bytes_per_launch": int(dom["bytes"] / max(dom["launches"], 1)),
                             "launches": dom["launches"], "avg_launch_ms": round(avg_ms, 4),
                             "share_of_device_time": round(dom_all["ms"] / sum(v["ms"] for v in kern.values()), 4)},
                # executed = the FLOPs of the GEMMs actually launched (the [skip | upsampled] convs run their
                # upsampled channels on the low-res grid: 8/27 of the direct count, an exact reassociation);
                # direct = the textbook 27-tap count of the reference graph (fwd x 3)
                "roofline_step": {"executed_tflop_per_step": round(exec_flop / 1e12, 3),
                                  "compute_frac": round(exec_flop / (ms_per_step * 1e-3) / (PEAK_FP32_TFLOPS * 1e12), 4),
                                  "direct_conv_tflop_per_step": round(step_flop / 1e12, 3),
                                  "direct_conv_equivalent_tflops": round(step_flop / (ms_per_step * 1e-3) / 1e12, 2),
                                  "hbm_frac": round(step_bytes / (ms_per_step * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                                  "algorithmic_gb_per_step": round(step_bytes / 1e9, 3),
                                  "gemm_ms_per_step": round(sum(v["ms"] for v in gemms.values()) / args.steps, 3),
                                  "non_gemm_ms_per_step": round(sum(v["ms"] for k, v in kern.items() if k not in gemms) / args.steps, 3)},
                # per kernel: tflops = MFMA work executed per second (Winograd rows x 64/216), as roofline.achieved
                "kernels": {k: {"ms_per_step": round(v["ms"] / args.steps, 3),
                                "tflops": round(v["flop"] * wino(k) / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 and v["flop"] > 0 else None}
                            for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:10]},
            }

    # ---- inference configurations (N = 1): BASELINE configs[0] and the generate.py tail
    def simple_timed(call, engine, profiled):
        """W warm-up calls, an all-events profiling pass of 2 calls, then K timed calls (host clock around device syncs +
        the engine's own device timer).  Returns (seconds, gpu seconds, per-call profile rows)."""
        for _ in range(args.warmup):
            call()
        engine.sync()
        for e in profiled:
            e.profile_filter("")
            e.profile_enable(True)
        for _ in range(2):
            call()
        engine.sync()
        rows = [r for e in profiled for r in e.profile_rows()]
        for e in profiled:
            e.profile_enable(False)
        for r in rows:
            r["ms"] /= 2.0; r["flop"] /= 2.0; r["bytes"] /= 2.0; r["launches"] //= 2
        engine.sync()
        t0 = time.perf_counter()
        engine.timer_start()
        for _ in range(args.steps):
            call()
        gpu_ms = engine.timer_stop()
        engine.sync()
        return time.perf_counter() - t0, gpu_ms * 1e-3, rows

    def kernel_table(rows, top=6):
        kern = by_kernel(rows)
        gemms = {k: v for k, v in kern.items() if v["flop"] > 0}
        dom_name, dom = max(gemms.items(), key=lambda kv: kv[1]["ms"])
        exec_flop = sum(v["flop"] * wino(k) for k, v in kern.items())
        table = {k: {"ms_per_call": round(v["ms"], 3),
                     "tflops": round(v["flop"] * wino(k) / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 and v["flop"] > 0 else None}
                 for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:top]}
        return dom_name, dom, exec_flop, table

    def predict_block(eng):
        """BASELINE configs[0]: model.predict on 16 synthetic grids, eval-mode BatchNorm, inputs resident in HBM, the two
        output tensors left in HBM (`value`); and the same forward ending in the uint8 argmax / 0.8-threshold volumes of
        generate.py:221-225 (`labels_only`: 2 bytes per voxel of output instead of 384)."""
        Bp = min(16, eng.max_batch)
        Xp, labp, _ = synthetic_batch(Bp, d, C, seed=100)
        eng.upload_batch(Xp, labp)
        sec, gpu_s, rows = simple_timed(lambda: eng.predict_resident(False), eng, [eng])
        sec_l, gpu_l, _ = simple_timed(lambda: eng.predict_resident(True, 0.8), eng, [eng])
        dom_name, dom, exec_flop, table = kernel_table(rows)
        scale = (d / 32.0) ** 3
        ms = sec / args.steps * 1e3
        flop = 125.886e9 * scale * Bp                                   # U-Net forward per grid
        by_full = (95.75e6 + 70.78e6) * scale * Bp + 124.6e6            # sum in + sum out per grid + weights
        by_lab = by_full - (96 * 4 - 2) * (32 ** 3) * scale * Bp        # the head's 96 floats per voxel -> 2 bytes
        ex = dom["flop"] * wino(dom_name) / (dom["ms"] * 1e-3) / 1e12
        return {"workload": "AtomUnet.model.predict, %d x %d^3 x 1 grids, eval-mode BatchNorm (BASELINE.json configs[0]); "
                            "inputs and outputs resident in HBM" % (Bp, d),
                "value": round(Bp * args.steps / sec, 2), "unit": "voxel-grids/s", "ms_per_call": round(ms, 3),
                "gpu_active_s": round(gpu_s, 5), "steps": args.steps, "warmup": args.warmup, "batch": Bp,
                "labels_only": {"value": round(Bp * args.steps / sec_l, 2), "ms_per_call": round(sec_l / args.steps * 1e3, 3),
                                "hbm_frac": round(by_lab / (sec_l / args.steps) / (PEAK_HBM_GBS * 1e9), 4),
                                "algorithmic_gb_per_call": round(by_lab / 1e9, 3)},
                "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(ex, 2), "peak": PEAK_FP32_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(ex / PEAK_FP32_TFLOPS, 4),
                             "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
                             "traffic": None},
                "roofline_call": {"executed_tflop_per_call": round(exec_flop / 1e12, 3),
                                  "compute_frac": round(exec_flop / (ms * 1e-3) / (PEAK_FP32_TFLOPS * 1e12), 4),
                                  "direct_conv_tflop_per_call": round(flop / 1e12, 3),
                                  "direct_conv_equivalent_tflops": round(flop / (ms * 1e-3) / 1e12, 2),
                                  "hbm_frac": round(by_full / (ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                                  "algorithmic_gb_per_call": round(by_full / 1e9, 3)},
                "kernels": table}

    def generate_block(eng):
        """generate.py:204-236 on the device: decoder.predict -> unet.model.predict -> argmax / threshold -> connected
        components (> 3 voxels) -> majority vote + centroids, one C-ABI call per batch (ics_vae_decode_to_unet_atoms).
        z and cond go up (1 KB per grid); species, mask, the density channel and one row of integers per atom come back
        (the arrays generate.py saves).  The convexity test / recursive split of the non-convex components is host work
        (icsg3d_amd/watershed.py, Qhull) and is not in this figure.  Random weights never reach sig >= 0.8, so the threshold
        is the 90 % quantile of this batch's own sigmoid output -- components exist, the device pass has work to do."""
        Bg = eng.max_batch
        gvae = VaeEngine(eng, in_channels=C, d=d, max_batch=Bg)
        gvae.set_weights(glorot_params(vae_param_shapes(C, d=d), seed=3))
        rng = np.random.default_rng(7)
        z = rng.standard_normal((Bg, 256)).astype(np.float32)
        cnd = np.eye(10, dtype=np.float32)[np.arange(Bg) % 10]
        thr = float(np.quantile(eng.predict(gvae.decode(z[:2], cnd[:2]))[1], 0.9))
        res = {}

        def call():
            res["out"] = gvae.decode_to_atoms(eng, z, cnd, thresh=thr, max_atoms=4096)
        sec, gpu_s, rows = simple_timed(call, gvae, [gvae, eng])
        dom_name, dom, exec_flop, table = kernel_table(rows)
        scale = (d / 32.0) ** 3
        ms = sec / args.steps * 1e3
        flop = (125.886e9 + 1.62e9) * scale * Bg                        # U-Net fwd + decoder fwd per grid
        ex = dom["flop"] * wino(dom_name) / (dom["ms"] * 1e-3) / 1e12
        out_g = res["out"]
        # what generate.py does next on the host (icsg3d_amd/watershed.refine_atoms): the convexity test of every kept
        # component (an exact 26-direction bound first, Qhull only where that is inconclusive) and, for the samples with
        # a non-convex component, the recursive marker watershed (device flood, host recursion).  Timed once, untimed warm-up.
        from icsg3d_amd.watershed import refine_atoms
        refine_atoms(gvae.decode_to_atoms(eng, z, cnd, thresh=thr, max_atoms=4096, want_regions=True), degenerate="solid")
        t0 = time.perf_counter()
        out_r = gvae.decode_to_atoms(eng, z, cnd, thresh=thr, max_atoms=4096, want_regions=True)
        t1 = time.perf_counter()
        refine_atoms(out_r, degenerate="solid")
        t2 = time.perf_counter()
        refine = {"device_call_with_regions_ms": round((t1 - t0) * 1e3, 2), "host_refine_ms": round((t2 - t1) * 1e3, 2),
                  "samples_split": int(out_r["split"].sum()), "components_tested": int(out_r["n_atoms"].sum()),
                  "end_to_end_grids_per_s": round(Bg / (t2 - t0), 1),
                  "note": "random-weight masks: degenerate='solid' so that flat fragments do not fail the sample (the "
                          "reference stack would skip it, generate.py:246-248)"}
        # ... and the same host step on what a trained segmentation produces: nine ball-shaped atoms per grid, every component
        # convex -- decided from the device's integers (polytope count, second moments), no hull, no recursion
        from icsg3d_amd.watershed import segment_atoms
        zz, yy, xx = np.mgrid[:d, :d, :d]
        balls = np.zeros((Bg, d, d, d), np.uint8)
        for b_ in range(Bg):
            for k in range(9):
                c = (np.array([5 + 11 * (k // 9 % 3), 5 + 11 * (k // 3 % 3), 5 + 11 * (k % 3)]) + rng.uniform(-1, 1, 3)) * d / 32.0
                r = rng.uniform(2.2, 3.6) * d / 32.0
                balls[b_][((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= r * r] = 1
        sp_b = np.where(balls != 0, rng.integers(1, 95, size=balls.shape), 0).astype(np.uint8)
        refine_atoms(segment_atoms(balls[:2], sp_b[:2], max_atoms=64))
        t0 = time.perf_counter()
        ob = segment_atoms(balls, sp_b, max_atoms=64)
        t1 = time.perf_counter()
        refine_atoms(ob)
        t2 = time.perf_counter()
        refine["convex_case"] = {"host_arrays_device_pass_ms": round((t1 - t0) * 1e3, 2), "host_refine_ms": round((t2 - t1) * 1e3, 2),
                                 "atoms_per_grid": round(float(ob["n_atoms"].mean()), 1), "samples_split": int(ob["split"].sum()),
                                 "note": "synthetic masks of nine balls per grid (all convex), ics_op_segment_atoms + refine_atoms"}
        gvae.close()
        return {"workload": "generate.py:204-236 tail: decoder -> U-Net -> argmax / threshold -> components -> atoms, "
                            "%d x %d^3 x 1 grids per call; host <-> device copies of the call included" % (Bg, d),
                "value": round(Bg * args.steps / sec, 2), "unit": "voxel-grids/s", "ms_per_call": round(ms, 3),
                "gpu_active_s": round(gpu_s, 5), "steps": args.steps, "warmup": args.warmup, "batch": Bg,
                "threshold": round(thr, 4), "mask_fraction": round(float(out_g["mask"].mean()), 4),
                "atoms_per_grid": round(float(np.mean(out_g["n_atoms"])), 1),
                "components_per_grid": round(float(np.mean(out_g["n_components"])), 1),
                "refine": refine,
                "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(ex, 2), "peak": PEAK_FP32_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(ex / PEAK_FP32_TFLOPS, 4), "traffic": None},
                "roofline_call": {"executed_tflop_per_call": round(exec_flop / 1e12, 3),
                                  "compute_frac": round(exec_flop / (ms * 1e-3) / (PEAK_FP32_TFLOPS * 1e12), 4),
                                  "direct_conv_tflop_per_call": round(flop / 1e12, 3)},
                "kernels": table}

    if args.workload in ("predict", "generate"):
        if world != 1:
            raise SystemExit("--workload %s is a single-GPU configuration" % args.workload)
        eng = UnetEngine(in_channels=C, num_classes=95, d=d, max_batch=B)
        eng.set_weights(PU)
        blk = predict_block(eng) if args.workload == "predict" else generate_block(eng)
        out = {"metric": "voxel-grids/s (%s) for %d^3 grids" % ("U-Net forward-only" if args.workload == "predict"
                                                                  else "generate tail: decoder + U-Net forward + atoms", d),
               "n_gpus": 1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "config": {"workload": blk["workload"], "global_batch": blk["batch"], "grid": d,
                                               "parallelism": "dp1"}}
        out.update({k: v for k, v in blk.items() if k != "workload"})
        out["ms_per_step"] = out["ms_per_call"]
        if not args.no_cpu_baseline:
            wd.pause("CPU baseline subprocess (no collective in it)")
            cb = cpu_baseline(B, d)
            out["cpu_baseline"] = {"value": cb.get(args.workload, {}).get("value"), "unit": "voxel-grids/s", "cores": cb["cores"],
                                   "kind": "port", "sample": cb["sample"]}
        else:
            out["cpu_baseline"] = None

    # ---- DFC-VAE step (BASELINE configs[2]): encoder + decoder + frozen perceptual U-Net (batch-statistics BN)
    if args.workload in ("unet", "vae", "joint") and not (args.workload == "unet" and args.no_secondary):
        pm = unet if args.workload == "unet" else UnetEngine(in_channels=C, d=d, max_batch=B)
        if pm is not unet:
            pm.set_weights(PU)
        vae = VaeEngine(pm, in_channels=C, d=d, max_batch=B, lr=5e-4)
        vae.set_weights(glorot_params(vae_param_shapes(C, d=d), seed=3))
        eps = np.random.default_rng(2 + rank).standard_normal((B, 256)).astype(np.float32)
        vae.upload_batch(X, cond, eps)
        if use_dist:
            init_engine_comm(vae, dist, rank, world, sync_bn=args.sync_bn, force=force)
        if args.workload == "joint":
            # the two engines would own a stream each and, left alone, share the chip -- BOTH slow down (35.0 ms per pair
            # against 27.9 + 5.4 one after the other); chaining the streams with events costs more still (40.9).  The VAE
            # engine enqueues on the U-Net engine's stream instead: the steps alternate in program order.
            vae.share_stream(unet)

            def step():
                unet.train_step_resident(False)
                vae.train_step_resident(False)
            engines, profiled = [unet, vae], [unet, vae, pm]

            def step_serial():           # the all-events pass: one engine at a time, or a bracket on one stream times
                unet.train_step_resident(False); unet.sync()     # a kernel slowed down by the other stream's kernels
                vae.train_step_resident(False); vae.sync()
        else:
            step = lambda: vae.train_step_resident(False)  # noqa: E731
            engines, profiled = [vae], [vae, pm]
            step_serial = None
        elapsed_v, elapsed_v_plain, _rows_v_live, rows_v = timed(step, engines, profiled, step_serial)
        gpu_active_v = timed.gpu_active_s
        mv = vae.train_step_resident(True)
        if not np.all(np.isfinite(mv)):
            raise SystemExit("non-finite DFC-VAE metrics: %s" % mv)
        if rank == 0:
            kern = by_kernel(rows_v)
            gemms = {k: v for k, v in kern.items() if v["flop"] > 0}
            dom_name, dom = max(gemms.items(), key=lambda kv: kv[1]["ms"])
            ms_v = elapsed_v / args.steps * 1e3
            exec_flop = sum(v["flop"] * wino(k) for k, v in kern.items()) / args.steps
            blk = {"workload": ("U-Net step + DFC-VAE step per iteration" if args.workload == "joint" else
                                "LatticeDFCVAE train step (encoder + decoder + frozen perceptual U-Net c1..c10 x2 fwd "
                                "+ bwd-data, Adam), %d x %d^3 x 1 grids per GPU (BASELINE.json configs[2])" % (B, d)),
                   "value": round(world * B * args.steps / elapsed_v, 2), "unit": "voxel-grids/s",
                   "ms_per_step": round(ms_v, 3), "ms_per_step_events_off": round(elapsed_v_plain / args.steps * 1e3, 3),
                   "gpu_active_s": round(gpu_active_v, 5),
                   "steps": args.steps, "warmup": args.warmup,
                   "executed_tflop_per_step": round(exec_flop / 1e12, 3),
                   "compute_frac": round(exec_flop / (ms_v * 1e-3) / (PEAK_FP32_TFLOPS * 1e12), 4),
                   "algorithmic_tflop_per_step": round(VAE_FLOP_PER_GRID * (d / 32.0) ** 3 * B / 1e12, 3),
                   "kernel_launches_per_step": round(timed.kernel_launches_per_step, 1),
                   "profiler_brackets_per_step": round(sum(v["launches"] for v in kern.values()) / args.steps, 1),
                   # tflops = MFMA work executed per second (Winograd rows x 64/216), as roofline.achieved
                   "dominant_kernel": {"kernel": dom_name,
                                       "tflops": round(dom["flop"] * wino(dom_name) / (dom["ms"] * 1e-3) / 1e12, 2),
                                       "algorithmic_equivalent_tflops": round(dom["flop"] / (dom["ms"] * 1e-3) / 1e12, 2),
                                       "ms_per_step": round(dom["ms"] / args.steps, 3)},
                   "kernels": {k: {"ms_per_step": round(v["ms"] / args.steps, 3),
                                   "tflops": round(v["flop"] * wino(k) / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 and v["flop"] > 0 else None}
                               for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:8]}}
            if out is not None:
                out["secondary"] = blk
                # one U-Net step + one DFC-VAE step per batch, back to back (BASELINE metric "U-Net+VAE")
                out["unet_plus_vae"] = {"ms_per_batch": round(out["ms_per_step"] + ms_v, 3),
                                        "value": round(world * B / ((out["ms_per_step"] + ms_v) * 1e-3), 2),
                                        "unit": "voxel-grids/s"}
                # a reader of `metric` + `value` alone must not take `value` for the U-Net+VAE pair
                out["metric"] += " (= %.1f voxel-grids/s, U-Net step + DFC-VAE step per batch)" % out["unet_plus_vae"]["value"]
            else:
                out = {"metric": "voxel-grids/s (fwd+bwd) for %d^3 %s at batch %d per GPU" % (d, args.workload, B),
                       "n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                       "data": "synthetic", "config": {"workload": blk["workload"], "global_batch": world * B, "grid": d,
                                                       "parallelism": "dp%d" % world}}
                out.update({k: blk[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup")})
                ex = blk["dominant_kernel"]["tflops"]
                out["roofline"] = {"bound": "mfma", "kernel": dom_name, "achieved": round(ex, 2),
                                   "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(ex / PEAK_FP32_TFLOPS, 4),
                                   "algorithmic_equivalent_tflops": blk["dominant_kernel"]["algorithmic_equivalent_tflops"],
                                   "traffic": None,
                                   "source": ("untimed all-events pass with the two engines SERIALISED (in the timed region "
                                              "they run on two streams and share the chip: a per-kernel bracket there times "
                                              "contention, not the kernel); step-level compute_frac is the timed figure")
                                             if args.workload == "joint" else "untimed all-events pass"}
                out["detail"] = blk
                out["cpu_baseline"] = None

    if args.workload == "unet" and world == 1 and not args.no_inference:
        out["inference"] = {"predict": predict_block(unet), "generate": generate_block(unet)}
    if rank == 0:
        if args.workload == "unet" and world == 1 and not args.no_cpu_baseline:
            wd.pause("CPU baseline subprocess (no collective in it)")
            out["cpu_baseline"] = cpu_baseline(B, d)
            if "secondary" in out and out["cpu_baseline"].get("vae"):
                out["secondary"]["cpu_baseline"] = out["cpu_baseline"]["vae"]
            if "inference" in out and out["cpu_baseline"].get("predict"):
                out["inference"]["predict"]["cpu_baseline"] = out["cpu_baseline"]["predict"]
                out["inference"]["generate"]["cpu_baseline"] = out["cpu_baseline"].get("generate")
                if (out["cpu_baseline"].get("generate") or {}).get("refine"):       # like for like with the refine block
                    out["inference"]["generate"]["refine"]["cpu_baseline"] = out["cpu_baseline"]["generate"]["refine"]
        line = json.dumps(out) + "\n"
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, line.encode())
        else:
            sys.stdout.write(line)
    if dist is not None:
        wd.beat("final gloo barrier")
        dist.barrier()
        dist.destroy_process_group()
    wd.stop()


if __name__ == "__main__":
    main()
