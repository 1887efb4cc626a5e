"""Self-launch of one process per GPU (new -- the reference is single-process).

`python bench.py --gpus 8` (or train_unet.py / train_vae.py --gpus 8) started WITHOUT torch.distributed.run must not
quietly run one rank and report it as eight.  `ensure_ranks(n)` is called before anything touches the GPU: when the
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


def visible_gpus() -> int:
    """Number of HIP devices WITHOUT initialising the runtime in this process (torch.cuda.device_count() reads the
    driver's device list on this image; a HIP call here would make this process a GPU process before it forks ranks)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(script: str, argv, n: int, port: int):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)


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
