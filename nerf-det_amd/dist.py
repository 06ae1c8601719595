"""One-process-per-GPU plumbing: scenes are independent units, so inference shards by scene
with no data-path collective; the only exchanges are the result gather at the end (``mmdet.apis.multi_gpu_test`` /
``collect_results_gpu`` semantics, tools/test.py:131-136) and, in training, DDP's gradient all-reduce plus the scalar
``reduce_mean`` of imvoxel_head_v2.py:175.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def launched() -> bool:
    """True under a launcher (``torch.distributed.run``, :func:`launch_local_ranks`): the env:// rendezvous variables are set."""
    return all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))


def init_dist(backend: str = "nccl") -> tuple:
    """tools/train.py:98-102 ``init_dist('pytorch')``: env:// rendezvous, device = LOCAL_RANK.  Under a launcher the process group
    is created whatever the world size -- also for a single rank, so that the RCCL code path of a 1-GPU run is the 8-GPU one."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local)
    if (world > 1 or launched()) and not dist.is_initialized():
        kw = dict(device_id=torch.device("cuda", local)) if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return rank, world, local


def get_dist_info() -> tuple:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Scene indices of this rank: ``DistributedSampler(shuffle=False)`` -- round-robin, padded by wrapping so every
    rank runs the same number of steps (the padding is dropped again by :func:`collect_results`)."""
    per = (n + world - 1) // world
    idx = list(range(n))
    pad = per * world - n
    if pad and idx:   # DistributedSampler: repeat the index list as often as needed (n < world included), never leave the range
        idx += (idx * ((pad + n - 1) // n))[:pad]
    return idx[rank:per * world:world]


def collect_results(part: Sequence, size: int) -> List:
    """All ranks' per-scene results merged back into dataset order on rank 0 (``collect_results_gpu``):
    interleave the rank-local lists, drop the sampler padding.  Returns ``None`` on other ranks."""
    rank, world = get_dist_info()
    if world == 1:
        return list(part)[:size]
    parts = [None] * world
    dist.all_gather_object(parts, list(part))
    if rank != 0:
        return None
    ordered = []
    for items in zip(*parts):
        ordered.extend(items)
    return ordered[:size]


def multi_gpu_test(model: Callable, scenes: Sequence, to_device: Callable = lambda b: b) -> List:
    """Each rank runs ``model(return_loss=False, **scene)`` on its shard; rank 0 gets every result in order."""
    rank, world = get_dist_info()
    out = []
    with torch.no_grad():
        for i in shard_indices(len(scenes), rank, world):
            out.extend(model(return_loss=False, **to_device(scenes[i])))
    return collect_results(out, len(scenes))


def max_over_ranks(seconds: float, device=None) -> float:
    """bench.py timing: the step time of the job is the slowest rank's."""
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_affinity(local_rank: int, local_world: int, n_cpus: int = 0) -> dict:
    """Host share of one rank when ``local_world`` ranks run on one node: a contiguous block of logical CPUs (``cpus``) and the thread cap for
    torch's intra-op pool and the BLAS / OpenMP pools (``threads``).  Eight ranks each enqueue ~2.4 ms of launches per scene and run the numpy
    camera products of hostmath.py; left alone, every rank's pools size themselves for the whole machine (256 logical CPUs on the MI355X hosts)
    and the spinning workers of one rank delay the host syncs of the others (host-side findings)."""
    if not n_cpus:
        n_cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    local_world = max(1, local_world)
    per = max(1, n_cpus // local_world)
    first = (local_rank % local_world) * per
    return dict(cpus=list(range(first, min(first + per, n_cpus))) or [local_rank % n_cpus], threads=max(1, min(16, per)))


def apply_rank_affinity(local_rank: int, local_world: int) -> dict:
    """Pin this process to its block of the CPUs it is allowed to run on and cap its thread pools (call before the first heavy torch op)."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    share = rank_affinity(local_rank, local_world, len(allowed))
    cpus = [allowed[i] for i in share["cpus"] if i < len(allowed)]
    if hasattr(os, "sched_setaffinity") and cpus and local_world > 1:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    torch.set_num_threads(share["threads"])
    return dict(cpus=cpus, threads=share["threads"])


def launch_local_ranks(script: str, argv: Sequence[str], n: int, timeout: float = 0.0) -> int:
    """``python script --gpus N`` without an external launcher: start the N ranks of one node as fresh child processes (the parent never
    touches the GPU), 127.0.0.1 rendezvous on a free port, rank 0 inherits stdout.  The first rank that exits non-zero takes the others
    down with it (they would wait in a collective forever); returns that exit code.  The port is picked by binding port 0 and closed again
    before rank 0 binds it: if somebody else took it in between (rank 0 reports EADDRINUSE), the launch is repeated on another port, three
    times at most.  ``timeout`` > 0: ranks still running after that many seconds are killed and 124 is returned."""
    import threading
    rc = 0
    for attempt in range(3):
        port = _free_port()
        procs, tail = [], []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL,
                                          stderr=subprocess.PIPE if r == 0 else None))

        def tee(stream):     # rank 0's stderr passes through; its last lines are kept to recognise a lost port
            for line in iter(stream.readline, b""):
                sys.stderr.buffer.write(line)
                sys.stderr.buffer.flush()
                tail.append(line)
                del tail[:-80]
        th = threading.Thread(target=tee, args=(procs[0].stderr,), daemon=True)
        th.start()
        rc, t0, live = 0, time.monotonic(), list(procs)
        try:
            while live:
                for p in list(live):
                    code = p.poll()
                    if code is None:
                        continue
                    live.remove(p)
                    if code != 0:
                        rc = rc or code
                        for q in live:
                            q.terminate()
                if timeout and live and time.monotonic() - t0 > timeout:
                    rc = rc or 124
                    for q in live:
                        q.kill()
                time.sleep(0.05)
        finally:
            for p in live:
                p.kill()
        th.join(timeout=2.0)
        lost_port = rc != 0 and any(b"EADDRINUSE" in l or b"ddress already in use" in l for l in tail)
        if not lost_port:
            break
    return rc
