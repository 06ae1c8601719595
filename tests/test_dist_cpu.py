"""world_size-2 gloo tests (CPU) of the multi-process path: scene sharding + ordered result gather (inference),
loss-scalar averaging and reduce_mean (training), max-over-ranks timing (bench)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from nerfdet_amd import dist as D
    from nerfdet_amd.detector import BaseDetector
    from nerfdet_amd.head import _reduce_mean
    r, w, _ = D.init_dist("gloo")
    assert (r, w) == (rank, world)

    class FakeDet:  # stands in for nerfdet: one result dict per scene, tagged with the scene id
        def __call__(self, return_loss, scene_id):
            return [dict(scene=scene_id, rank=rank)]
    scenes = [dict(scene_id=i) for i in range(7)]  # odd count: exercises the sampler padding
    res = D.multi_gpu_test(FakeDet(), scenes)
    t = D.max_over_ranks(1.0 + rank)
    n_pos = _reduce_mean(torch.tensor(float(3 + 4 * rank)))
    loss, log = BaseDetector._parse_losses(dict(loss_cls=torch.tensor(1.0 + rank), loss_bbox=torch.tensor([2.0, 4.0]) * (rank + 1),
                                                acc=torch.tensor(10.0 * rank)))
    q.put((rank, res, t, float(n_pos), float(loss), log, D.shard_indices(7, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_scene_sharding_and_reductions():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        item = q.get(timeout=90)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    res0, t0, npos0, loss0, log0, idx0 = got[0]
    res1, t1, npos1, loss1, log1, idx1 = got[1]
    assert idx0 == [0, 2, 4, 6] and idx1 == [1, 3, 5, 0]          # round-robin, rank 1 padded by wrapping
    assert res1 is None and [r["scene"] for r in res0] == list(range(7))  # dataset order restored, padding dropped
    assert [r["rank"] for r in res0] == [0, 1, 0, 1, 0, 1, 0]
    assert t0 == t1 == 2.0                                          # slowest rank
    assert npos0 == npos1 == 5.0                                    # mean of 3 and 7 (imvoxel_head_v2.py:175)
    assert loss0 == 1.0 + 3.0 and loss1 == 2.0 + 6.0                # local loss = sum of the keys containing 'loss'
    assert log0 == log1 and abs(log0["loss_cls"] - 1.5) < 1e-6 and abs(log0["loss_bbox"] - 4.5) < 1e-6 and abs(log0["acc"] - 5.0) < 1e-6


def test_shard_and_collect_single_process():
    from nerfdet_amd import dist as D
    assert D.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.shard_indices(5, 2, 4) == [2, 1] and D.shard_indices(5, 1, 4) == [1, 0]  # n=5, world=4: per=2, pad 3 by wrapping
    # fewer scenes than ranks: the padding wraps around the 3 scenes and never leaves the range (DistributedSampler)
    shards = [D.shard_indices(3, r, 8) for r in range(8)]
    assert shards == [[0], [1], [2], [0], [1], [2], [0], [1]]
    assert D.shard_indices(0, 0, 4) == []
    assert D.collect_results([1, 2, 3], 2) == [1, 2]
    assert D.max_over_ranks(0.25) == 0.25


@pytest.mark.timeout(180)
def test_bench_launches_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` with no launcher in the environment spawns its two ranks itself (fresh children, the parent
    never initialises a device), they rendezvous, and ONE JSON line comes back with n_gpus=2 and the slowest rank's time.
    ``--dry-run`` swaps the GPU step for a sleep so that the plumbing runs on the CPU."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["dry_run"] is True
    assert d["ms_per_step"] >= 4.0          # rank 1 sleeps 4 ms per step, rank 0 2 ms: the job's time is the slowest rank's
    assert abs(d["value"] - 2 * 5 / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]
    # a rank that dies takes the job down with a non-zero exit code instead of leaving the others in the barrier
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "tiny"],
                         env=dict(env, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""), capture_output=True, text=True, timeout=150)
    assert bad.returncode != 0 and "bench.py needs a GPU" in bad.stderr
