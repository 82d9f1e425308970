"""bench.py --gpus N must run N ranks however it is started (the reference's multi-GPU mechanism is
nn.DataParallel inside one process, dmc/train.py:230-237,598-600; here it is one process per GPU).
The rehearsal mode (--dry-run 1: gloo on the CPU, no frame coded) exercises exactly the launch logic."""
import json
import os
import subprocess
import sys

import pytest

from deepvideocodec_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _result(stdout):
    lines = [json.loads(l) for l in stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, stdout
    return lines[0]


@pytest.mark.timeout(300)
def test_plain_start_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: two ranks appear, n_gpus == 2, every GOP is coded once."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "1", "--steps", "2",
                        "--warmup", "1", "--gops-in-flight", "3"], capture_output=True, text=True, env=_clean_env(), timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _result(r.stdout)
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["ranks"] == [0, 1]
    assert out["gops"] == 2 * 3 * 3   # ranks x (warmup + steps) x GOPs in flight


@pytest.mark.timeout(300)
def test_under_torch_distributed_run_and_mismatch():
    """The driver's form (torch.distributed.run ... bench.py --gpus N) is used as is; a --gpus that
    disagrees with the launcher's world size is an error, not a silently wrong n_gpus."""
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
            "127.0.0.1", "--master-port", str(launch.free_port()), os.path.join(ROOT, "bench.py"), "--dry-run", "1",
            "--steps", "1", "--warmup", "0"]
    r = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, env=_clean_env(), timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _result(r.stdout)["n_gpus"] == 2
    base[base.index("--master-port") + 1] = str(launch.free_port())
    r = subprocess.run(base + ["--gpus", "4"], capture_output=True, text=True, env=_clean_env(), timeout=280)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_resolve_world_single_process(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    assert launch.resolve_world(None, "x.py", []) == 1
    assert launch.resolve_world(1, "x.py", []) == 1
    with pytest.raises(launch.WorldError):          # more ranks than GPUs (none here): refuse, do not shrink
        launch.resolve_world(2, "x.py", [], need_devices=True)
    monkeypatch.setenv("WORLD_SIZE", "8")
    with pytest.raises(launch.WorldError):          # rank variables missing
        launch.resolve_world(8, "x.py", [])


def test_host_budget_sizes_threads_per_rank(monkeypatch):
    """Cores a rank may count on = CPUs allowed / ranks of the node (capped by an OMP_NUM_THREADS >= 2), and the
    GOP threads / coder pool derived from it: eight ranks on a small host shrink instead of oversubscribing it."""
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(32)), raising=False)
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    b = launch.host_budget()
    assert b == {"cpus_allowed": 32, "local_world": 8, "cores_per_rank": 4}
    assert launch.default_gops_in_flight(4) == 2 and launch.coder_pool_workers(4) == 2
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    monkeypatch.setenv("OMP_NUM_THREADS", "16")
    assert launch.host_budget()["cores_per_rank"] == 16
    assert launch.default_gops_in_flight(16) == 3 and launch.coder_pool_workers(16) == 6
    monkeypatch.setenv("OMP_NUM_THREADS", "1")          # torch.distributed.run's default export: ignored
    assert launch.host_budget()["cores_per_rank"] == 32
    assert launch.default_gops_in_flight(1) == 1 and launch.coder_pool_workers(1) == 2
