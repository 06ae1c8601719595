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


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.timeout(400)
@pytest.mark.parametrize("script", ["bench.py", os.path.join("tools", "bench_train.py")])
def test_eight_rank_dry_run_of_both_benchmarks(script):
    """The driver's N = 8 launch, rehearsed on the CPU: ``python <script> --gpus 8 --dry-run`` starts its eight ranks
    through nerfdet_amd.dist.launch_local_ranks, they rendezvous on 127.0.0.1 over gloo, every rank takes its own share of the host
    (rank_affinity), and ONE JSON line with n_gpus = 8 comes back.  bench_train's dry run goes through the product's wrap_ddp /
    build_optimizer / train_one_step and checks that the ranks hold identical parameters afterwards."""
    import json
    import subprocess
    import sys
    for attempt in range(2):      # eight interpreters start at once on what may be an 8-CPU container: one retry for a rendezvous that timed out
        out = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run"],
                             env=_clean_env(OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=360)
        if out.returncode == 0:
            break
        print(f"attempt {attempt}: rc {out.returncode}\n{out.stderr[-3000:]}")
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["dry_run"] is True and d["value"] > 0
    if "ddp_buckets_bytes" in d:
        assert sum(d["ddp_buckets_bytes"]) == 4 * (64 * 256 + 256 + 256 * 1024 + 1024 + 1024 * 256 + 256 + 4 * 4 + 4)
        assert len(d["ddp_buckets_bytes"]) >= 2 and d["rank_threads"] >= 1


@pytest.mark.timeout(200)
def test_a_rank_dying_in_the_barrier_ends_the_job_in_bounded_time():
    """Rank 3 of 4 exits while the others wait in the closing barrier: the launcher takes them down and returns its exit code within seconds
    (without that the job would sit in the collective until the driver's limit)."""
    import subprocess
    import sys
    import time
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "0", "--dry-run"],
                         env=_clean_env(NDET_DRYRUN_DIE_RANK="3", OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=180)
    took = time.monotonic() - t0
    assert out.returncode == 3, (out.returncode, out.stderr[-1500:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert took < 120, took


def test_rank_affinity_partitions_the_host():
    from nerfdet_amd import dist as D
    shares = [D.rank_affinity(r, 8, 256) for r in range(8)]
    assert [s["cpus"][0] for s in shares] == [32 * r for r in range(8)] and all(len(s["cpus"]) == 32 and s["threads"] == 16 for s in shares)
    seen = [c for s in shares for c in s["cpus"]]
    assert sorted(seen) == list(range(256))                       # disjoint, complete
    assert D.rank_affinity(0, 1, 16) == dict(cpus=list(range(16)), threads=16)
    assert D.rank_affinity(5, 8, 4)["threads"] == 1 and len(D.rank_affinity(5, 8, 4)["cpus"]) == 1     # fewer CPUs than ranks: still one each


def test_launcher_retries_when_the_port_was_taken(tmp_path, monkeypatch):
    """The free port is found by bind(0) and closed before rank 0 binds it; if somebody else took it meanwhile, rank 0 reports EADDRINUSE and
    the launch is repeated on another port."""
    import sys
    from nerfdet_amd import dist as D
    script = tmp_path / "rank.py"
    marker = tmp_path / "attempts"
    script.write_text(
        "import os, sys\n"
        f"p = {str(marker)!r}\n"
        "n = int(open(p).read()) if os.path.exists(p) else 0\n"
        "if os.environ['RANK'] == '0':\n"
        "    open(p, 'w').write(str(n + 1))\n"
        "    if n == 0:\n"
        "        sys.stderr.write('RuntimeError: The server socket has failed to listen on any local network address. EADDRINUSE\\n')\n"
        "        sys.exit(1)\n"
        "print('ok')\n")
    assert D.launch_local_ranks(str(script), [], 1) == 0
    assert marker.read_text() == "2"
