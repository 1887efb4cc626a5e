"""Self-launch of one process per GPU (new -- the reference is single-process).

`python bench.py --gpus 8` (or train_unet.py / train_vae.py --gpus 8) started WITHOUT torch.distributed.run must not
quietly run one rank and report it as eight.  `ensure_ranks(n)` is called before anything touches the GPU (the GPU count
comes from sysfs, not from torch / HIP: visible_gpus): when the
process is not already a rank of an n-rank job it starts

    python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port P <script> <argv>

as a CHILD process (never exec: a process that has initialised HIP must not be replaced, and the parent stays around to
relay the child's stdout -- the one JSON line of bench.py -- and its exit status), then exits with the child's status.
If fewer than n GPUs are visible it exits non-zero with a message instead of measuring something else.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"
DRI_DIR = "/dev/dri"


def _kfd_gpus(nodes_dir=KFD_NODES, dri_dir=DRI_DIR):
    """GPU agents the way ROCr enumerates them, without loading ROCr: the KFD topology nodes with simd_count > 0 (CPUs have 0),
    in node order, whose DRM render node this process can open (a container that was handed one GPU of an eight-GPU host
    still sees all eight topology nodes; ROCr skips the ones it cannot open, and so do we).  Opening a render node does not
    create a KFD process.  Returns [{"unique_id": int}] or None when there is no KFD topology to read."""
    try:
        names = sorted((n for n in os.listdir(nodes_dir) if n.isdigit()), key=int)
    except OSError:
        return None
    gpus = []
    for n in names:
        props = {}
        try:
            with open(os.path.join(nodes_dir, n, "properties")) as f:
                for line in f:
                    kv = line.split()
                    if len(kv) == 2:
                        props[kv[0]] = kv[1]
        except OSError:
            continue
        if int(props.get("simd_count", "0")) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1"))
        if minor < 0:
            continue
        try:
            os.close(os.open(os.path.join(dri_dir, "renderD%d" % minor), os.O_RDWR))
        except OSError:
            continue
        gpus.append({"unique_id": int(props.get("unique_id", "0"))})
    return gpus


def _filter_visible(gpus, spec):
    """ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES semantics: a comma-separated list of indices into the current list (or
    GPU-<uuid> strings, ROCr only); the list ends at the first entry that does not name a device (-1 hides the rest)."""
    out = []
    for tok in spec.split(","):
        tok = tok.strip()
        if tok.upper().startswith("GPU-"):
            want = tok[4:].lower().lstrip("0")
            hit = [g for g in gpus if ("%x" % g["unique_id"]) == want]
            if not hit:
                break
            out.append(hit[0])
            continue
        try:
            i = int(tok)
        except ValueError:
            break
        if i < 0 or i >= len(gpus):
            break
        out.append(gpus[i])
    return out


def _count_in_child() -> int:
    """Fallback when there is no KFD topology to read: ask torch in a short-lived CHILD process (subprocess, never exec), so
    that whatever runtime the count initialises dies with the child and this process stays GPU-free."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def visible_gpus(nodes_dir=KFD_NODES, dri_dir=DRI_DIR, env=None) -> int:
    """Number of HIP devices a rank will see, counted WITHOUT importing torch or touching HIP / HSA in this process
    (ADVICE r5 / VERDICT r5 next 5a: torch.cuda.device_count() falls back to hipGetDeviceCount when amdsmi discovery fails,
    and the launcher parent would then sit on every GPU while its ranks run): KFD topology nodes with SIMDs whose render
    node opens, filtered by ROCR_VISIBLE_DEVICES, then by HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES as its alias), which
    index into what the previous level left."""
    env = os.environ if env is None else env
    gpus = _kfd_gpus(nodes_dir, dri_dir)
    if gpus is None:
        return _count_in_child()
    if env.get("ROCR_VISIBLE_DEVICES") is not None:
        gpus = _filter_visible(gpus, env["ROCR_VISIBLE_DEVICES"])
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if env.get(var) is not None:
            gpus = _filter_visible(gpus, env[var])
            break
    return len(gpus)


def free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def forward_argv(argv):
    """The script's own arguments as torch.distributed.run must see them: its argparse takes `--d` for an ambiguous
    abbreviation of its own options (--duplicate-stdout-filters, ...) and exits before any rank starts, so
    `--d X` / `--d=X` travel under their alias `--dim` (bench.py, train_unet.py, train_vae.py accept both)."""
    out = []
    for a in argv:
        if a == "--d":
            out.append("--dim")
        elif a.startswith("--d="):
            out.append("--dim=" + a[4:])
        else:
            out.append(a)
    return out


def launch_command(script: str, argv, n: int, port: int):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + forward_argv(argv)


def under_launcher(n: int) -> bool:
    """True when this process already is one rank of an n-rank job (RANK / WORLD_SIZE set by torch.distributed.run)."""
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return False
    world = int(os.environ["WORLD_SIZE"])
    if world != n:
        raise SystemExit("launched with WORLD_SIZE=%d but --gpus %d: pass the same number to both" % (world, n))
    return True


def ensure_ranks(n: int, script: str, argv, count_gpus=visible_gpus, run=subprocess.run) -> None:
    """Returns when this process should go on (n == 1, or it is a rank of an n-rank job); otherwise spawns the ranks,
    relays their output and exits with their status.  `count_gpus` / `run` are injectable for the CPU test."""
    if n < 1:
        raise SystemExit("--gpus must be >= 1")
    if n == 1 and int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    if under_launcher(n):
        return
    have = count_gpus()
    if have < n and count_gpus is visible_gpus:
        # the sysfs count is the cheap, HIP-free answer; before REFUSING a run on its word, ask the runtime itself -- in a
        # short-lived child process, so that this parent still never initialises HIP
        have = max(have, _count_in_child())
    if have < n:
        raise SystemExit("--gpus %d requested but %d GPU(s) visible: refusing to run a smaller job under that name"
                         % (n, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = launch_command(script, argv, n, free_port())
    sys.stderr.write("[launcher] %s\n" % " ".join(cmd))
    sys.stderr.flush()
    proc = run(cmd, env=env)                                  # child inherits stdout / stderr: its JSON line IS ours
    raise SystemExit(proc.returncode)
