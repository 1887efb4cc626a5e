"""Two real ranks over RCCL: skipped on boxes with fewer than two GPUs -- which includes every box this
suite has run on so far; the single-rank communicator tests (test_gpu_dp.py) and the gloo arithmetic tests
(test_dataparallel_gloo.py) cover what one GPU can.  With two devices: replicas that start different end identical
(broadcast), stay identical (bucketed all-reduce, BN moving-statistics average), and SyncBN-sharded 2 x B equals one
process at 2B."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_two_ranks_match_one_process_at_twice_the_batch():
    from icsg3d_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29557", os.path.join(ROOT, "tests", "dp2_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=850)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("DP2_RESULT ")][-1]
    r = json.loads(line[len("DP2_RESULT "):])
    assert r["sync_bn"]["metrics_err"] <= 2e-5 and r["sync_bn"]["grad_err"] <= 1e-4, r
    assert r["local_bn"]["buckets"] >= 3, r
    assert r["sync_bn"]["buckets"] >= 3, r        # SyncBN collectives ran next to overlapped gradient buckets (two communicators)
    assert r["vae_sync_bn"]["metrics_err"] <= 5e-5 and r["vae_sync_bn"]["grad_err"] <= 2e-4, r
