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


def launch_local_ranks(script: str, argv: Sequence[str], n: int) -> int:
    """``python script --gpus N`` without an external launcher: start the N ranks of one node as fresh child processes (the parent never
    touches the GPU), 127.0.0.1 rendezvous on a free port, rank 0 inherits stdout.  The first rank that exits non-zero takes the others
    down with it (they would wait in a collective forever); returns that exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            p.kill()
    return rc
