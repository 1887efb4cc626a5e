"""`python bench.py --gpus N` without torch.distributed.run must start N ranks itself or fail loudly -- never print an
n_gpus: 1 line under the name of an N-GPU job.  icsg3d_amd/launcher.py, exercised on CPU: the decision
logic with injected GPU counts, and one real self-launch of two gloo ranks around a stub rank body."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from icsg3d_amd import launcher

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def clean_env(monkeypatch):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)


def test_single_gpu_and_already_launched_ranks_go_on(clean_env, monkeypatch):
    boom = lambda *a, **k: pytest.fail("must not spawn")
    launcher.ensure_ranks(1, "bench.py", [], count_gpus=boom, run=boom)
    monkeypatch.setenv("RANK", "1"); monkeypatch.setenv("WORLD_SIZE", "4")
    launcher.ensure_ranks(4, "bench.py", [], count_gpus=boom, run=boom)          # a rank of the right job: go on
    with pytest.raises(SystemExit, match="WORLD_SIZE=4 but --gpus 8"):
        launcher.ensure_ranks(8, "bench.py", [], count_gpus=boom, run=boom)
    with pytest.raises(SystemExit, match="WORLD_SIZE=4 but --gpus 1"):
        launcher.ensure_ranks(1, "bench.py", [], count_gpus=boom, run=boom)


def test_refuses_when_fewer_gpus_are_visible(clean_env):
    with pytest.raises(SystemExit, match="--gpus 8 requested but 1 GPU"):
        launcher.ensure_ranks(8, "bench.py", ["--gpus", "8"], count_gpus=lambda: 1, run=lambda *a, **k: pytest.fail("spawned"))


def test_spawns_the_contract_command_and_relays_the_status(clean_env):
    seen = {}

    class P:
        returncode = 7

    def run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return P()
    with pytest.raises(SystemExit) as e:
        launcher.ensure_ranks(8, "/x/bench.py", ["--gpus", "8", "--steps", "3"], count_gpus=lambda: 8, run=run)
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_real_self_launch_of_two_ranks_prints_one_line(tmp_path, clean_env):
    """The whole mechanism on CPU: a stub rank body (gloo rendezvous, rank 0 prints the JSON line) behind ensure_ranks."""
    script = tmp_path / "stub_bench.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        from icsg3d_amd.launcher import ensure_ranks
        ensure_ranks(2, os.path.abspath(__file__), sys.argv[1:], count_gpus=lambda: 2)
        import torch.distributed as dist
        dist.init_process_group("gloo")
        import torch
        t = torch.ones(1); dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t.item()), "argv": sys.argv[1:]}), flush=True)
        dist.barrier(); dist.destroy_process_group()
    """ % ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, str(script), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"n_gpus": 2, "sum": 2.0, "argv": ["--gpus", "2"]}


def test_bench_refuses_to_measure_one_gpu_under_the_name_of_eight(clean_env):
    """On this box (no GPU, or one): `python bench.py --gpus 8` exits non-zero and prints NO JSON line."""
    if launcher.visible_gpus() >= 8:
        pytest.skip("an 8-GPU node: the launch would go ahead")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "--gpus 8 requested" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
