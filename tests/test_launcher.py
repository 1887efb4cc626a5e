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
    # `--d 64` goes through torch.distributed.run's own argparse, which rejects --d as an ambiguous abbreviation of its
    # options: the launcher forwards it as --dim -- through the REAL two-rank launch
    p = subprocess.run([sys.executable, str(script), "--gpus", "2", "--d", "64", "--d=16"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"n_gpus": 2, "sum": 2.0, "argv": ["--gpus", "2", "--dim", "64", "--dim=16"]}


def test_scripts_accept_the_forwarded_spelling():
    """bench.py / train_unet.py / train_vae.py / generate.py all take --dim for --d."""
    import re
    for f in ("bench.py", "train_unet.py", "train_vae.py", "generate.py"):
        src = open(os.path.join(ROOT, f)).read()
        assert re.search(r'add_argument\("--d", "--dim"', src), f
    assert launcher.forward_argv(["--d", "64", "--steps", "3", "--d=32", "--data", "x"]) == \
        ["--dim", "64", "--steps", "3", "--dim=32", "--data", "x"]


def _fake_kfd(tmp_path, nodes, openable):
    """A KFD topology tree + /dev/dri: nodes = [(simd_count, drm_render_minor, unique_id)], openable = minors that exist."""
    nd, dri = tmp_path / "nodes", tmp_path / "dri"
    dri.mkdir(parents=True)
    for i, (simd, minor, uid) in enumerate(nodes):
        (nd / str(i)).mkdir(parents=True)
        (nd / str(i) / "properties").write_text(
            "cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\nunique_id %d\ngfx_target_version 90500\n"
            % (0 if simd else 64, simd, minor, uid))
    for m in openable:
        (dri / ("renderD%d" % m)).write_text("")
    return str(nd), str(dri)


def test_gpu_count_comes_from_sysfs_without_torch_or_hip(tmp_path):
    """the launcher parent counts GPUs from the KFD topology (simd_count > 0, render node openable),
    honouring ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES -- and imports neither torch nor a HIP library to do it."""
    # two CPU nodes + eight GPUs, of which this "container" was handed the render nodes of four
    nodes = [(0, -1, 0), (0, -1, 0)] + [(1024, 128 + i, 0xabc0 + i) for i in range(8)]
    nd, dri = _fake_kfd(tmp_path, nodes, openable=[128, 129, 130, 133])
    count = lambda env: launcher.visible_gpus(nd, dri, env)          # noqa: E731
    assert count({}) == 4
    assert count({"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert count({"HIP_VISIBLE_DEVICES": "3,1,7"}) == 2              # the list ends at the first index out of range
    assert count({"HIP_VISIBLE_DEVICES": "1,-1,2"}) == 1
    assert count({"HIP_VISIBLE_DEVICES": ""}) == 0
    assert count({"CUDA_VISIBLE_DEVICES": "0"}) == 1
    assert count({"ROCR_VISIBLE_DEVICES": "1,2,3"}) == 3
    assert count({"ROCR_VISIBLE_DEVICES": "1,2,3", "HIP_VISIBLE_DEVICES": "2,0"}) == 2    # indices into ROCr's list
    assert count({"ROCR_VISIBLE_DEVICES": "GPU-abc2,GPU-000000000000abc5"}) == 2          # uuid = unique_id in hex
    assert count({"ROCR_VISIBLE_DEVICES": "GPU-abc7"}) == 0                               # its render node is not ours
    # no GPU node at all (this build container): zero, not an exception
    nd0, dri0 = _fake_kfd(tmp_path / "cpu_only", [(0, -1, 0)], openable=[])
    assert launcher.visible_gpus(nd0, dri0, {}) == 0
    # the counting path itself stays off torch / HIP: run it in a fresh interpreter and look at what got imported
    code = ("import sys; sys.path.insert(0, %r); from icsg3d_amd import launcher; n = launcher.visible_gpus(%r, %r, {}); "
            "bad = [m for m in sys.modules if m.split('.')[0] in ('torch', 'ctypes')]; print(n, bad)" % (ROOT, nd, dri))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.stdout.split()[0] == "4" and out.stdout.strip().endswith("[]"), (out.stdout, out.stderr[-500:])


def test_watchdog_ends_a_hung_rank_and_the_job(tmp_path, clean_env):
    """a rank that makes no progress says where it is and exits 3 (os._exit, never a re-exec);
    torch.distributed.run then ends the other rank and the launcher parent relays a non-zero status."""
    from icsg3d_amd.watchdog import StepWatchdog
    hit = []
    wd = StepWatchdog(timeout=0.3, rank=5, exit_fn=hit.append, poll=0.05)
    import time
    for _ in range(8):                       # beats keep it quiet well past the timeout
        wd.beat("step")
        time.sleep(0.1)
    assert hit == []
    wd.pause("cpu baseline"); time.sleep(0.6)
    assert hit == []                         # a paused phase is not a hang
    wd.beat("all-reduce of bucket 2"); time.sleep(0.8)
    assert hit == [3]
    wd.stop()

    # through the class API (DataParallelMixin): the watchdog runs only while an engine call is in flight -- a process that
    # stops training (this very pytest process after test_gpu_dp.py, round 6's first GPU run) must not be ended
    from icsg3d_amd.dataparallel import DataParallelMixin
    hit2 = []
    m = DataParallelMixin()
    m._wd = StepWatchdog(timeout=0.3, rank=0, exit_fn=hit2.append, poll=0.05)
    m._wd.pause("between steps")
    with m._dp_watch("train_on_batch"):
        time.sleep(0.1)
    time.sleep(0.7)
    assert hit2 == []                        # idle between calls: quiet
    with m._dp_watch("train_on_batch"):
        time.sleep(0.7)                      # a call that does not return in time
    assert hit2 == [3]
    m._wd.stop()

    script = tmp_path / "stub_hang.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, %r)
        from icsg3d_amd.launcher import ensure_ranks
        ensure_ranks(2, os.path.abspath(__file__), sys.argv[1:], count_gpus=lambda: 2)
        from icsg3d_amd.watchdog import StepWatchdog
        import torch.distributed as dist
        dist.init_process_group("gloo")
        wd = StepWatchdog(timeout=2.0)
        wd.beat("step 0")
        if dist.get_rank() == 1:
            time.sleep(600)                  # the rank that never arrives
        wd.beat("gradient all-reduce of step 1")
        dist.barrier()                       # rank 0 hangs here, as in a collective whose peer died
        print("{}")
    """ % ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    t0 = time.time()
    p = subprocess.run([sys.executable, str(script), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "[watchdog] rank" in p.stderr and "no progress for" in p.stderr, p.stderr[-2000:]
    assert "gradient all-reduce of step 1" in p.stderr or "step 0" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 120


def test_bench_refuses_to_measure_one_gpu_under_the_name_of_eight(clean_env):
    """On this box (no GPU, or one): `python bench.py --gpus 8` exits non-zero and prints NO JSON line."""
    if launcher.visible_gpus() >= 8:            # (sysfs count: no torch, no HIP in this process)
        pytest.skip("an 8-GPU node: the launch would go ahead")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "--gpus 8 requested" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
