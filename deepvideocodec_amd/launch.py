"""One process per GPU: how the benchmarks get their N ranks.

The reference's only multi-GPU mechanism is a single-process nn.DataParallel wrapper
(dmc/train.py:230-237, enabled at :598-600); here every GPU gets its own process under
torch.distributed (RCCL).  A benchmark script calls `resolve_world()` first thing in main():

  * started by `python -m torch.distributed.run --nproc-per-node N script.py --gpus N` (the rank
    environment is present): nothing to do, the caller goes on as rank RANK of WORLD_SIZE;
  * started plainly as `python script.py --gpus N` with N > 1: this process never touches the GPU.
    It starts `torch.distributed.run` with N ranks as a child process, relays the child's output,
    checks that the single JSON result line reports N ranks, and exits with the child's status.
    (A process that has initialised the GPU must not replace itself with another program, so the
    launch is a child process and happens before any HIP call.)

A `--gpus` that disagrees with WORLD_SIZE is an error rather than a silently wrong `n_gpus`.
"""
import json
import os
import socket
import subprocess
import sys
from typing import List, Optional


class WorldError(RuntimeError):
    pass


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def resolve_world(gpus: Optional[int], script: str, argv: List[str], need_devices: bool = True) -> int:
    """Returns the world size this process runs in.  Does not return in the launching parent
    (it exits with the child's status).  `gpus` None means "whatever the launcher set up"."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        world = int(env_world)
        if gpus is not None and gpus != world:
            raise WorldError(f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks")
        for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            if k not in os.environ:
                raise WorldError(f"WORLD_SIZE is set but {k} is not: start the ranks with torch.distributed.run")
        return world
    if gpus is None or gpus == 1:
        return 1
    if gpus < 1:
        raise WorldError(f"--gpus {gpus}")
    if need_devices:
        import torch
        have = torch.cuda.device_count()   # (this parent only ever starts a child process: it never launches GPU work)
        if have < gpus:
            raise WorldError(f"--gpus {gpus} but only {have} GPU(s) are visible")
    sys.exit(_launch(gpus, script, argv))


def _launch(gpus: int, script: str, argv: List[str]) -> int:
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL needs dmabuf IPC on this driver
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    results = []
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        s = line.strip()
        if s.startswith("{") and '"metric"' in s:
            try:
                results.append(json.loads(s))
            except ValueError:
                pass
    rc = proc.wait()
    if rc != 0:
        print(f"launch: torch.distributed.run exited with status {rc}", file=sys.stderr)
        return rc
    if len(results) != 1:
        print(f"launch: expected one JSON result line from rank 0, saw {len(results)}", file=sys.stderr)
        return 3
    if results[0].get("n_gpus") != gpus:
        print(f"launch: asked for {gpus} ranks, the result reports n_gpus={results[0].get('n_gpus')}", file=sys.stderr)
        return 4
    return 0


def host_budget() -> dict:
    """Host cores this rank may count on: the CPUs the process is allowed to run on (capped by OMP_NUM_THREADS
    where the environment sets one as the job's CPU share), divided by the ranks of this node."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    per_rank = max(1, allowed // local_world)
    try:
        omp = int(os.environ["OMP_NUM_THREADS"])
        if omp >= 2:      # (torch.distributed.run exports 1 when the variable is unset: not a statement about the host)
            per_rank = min(per_rank, omp)
    except (KeyError, ValueError):
        pass
    return {"cpus_allowed": allowed, "local_world": local_world, "cores_per_rank": per_rank}


def default_gops_in_flight(cores_per_rank: int) -> int:
    """Independent GOPs coded concurrently per GPU (one host thread + stream each, sharing a range-coder pool):
    three hide the decoder's host round trips completely and keep about two host cores busy (bench.py's `host`
    block); fewer where a rank has fewer than six cores to itself."""
    return 3 if cores_per_rank >= 6 else 2 if cores_per_rank >= 3 else 1


def coder_pool_workers(cores_per_rank: int) -> int:
    """Threads of the range-coder pool of one rank: a frame has six symbol streams, so six at most."""
    return max(2, min(6, cores_per_rank - 2))


def init_process_group(world: int, backend: str, device=None):
    """torch.distributed over RCCL ("nccl" on ROCm) or gloo (CPU rehearsal); checks that the group
    really has `world` ranks."""
    import torch.distributed as dist
    if world == 1:
        return None
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    if dist.get_world_size() != world:
        raise WorldError(f"process group has {dist.get_world_size()} ranks, expected {world}")
    return dist
