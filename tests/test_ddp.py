"""The one collective of the path (north_star: "RCCL all-reduce over xGMI for DDP gradients only"), exercised for real:
two ranks, one scene each (``samples_per_gpu=1``), ``DistributedDataParallel(find_unused_parameters=True)`` as
tools/train.py:98-102 + config:185-186 set it up.  After one backward every rank must hold the MEAN of the two single-scene
gradients, dead parameters (``cov.*``, ``fpn_convs.1-3`` ...) must stay gradient-free without hanging the reducer, and the logged
losses must be the rank average.

CPU (gloo): the HIP ops are stood in for by the oracle (tests/cpu_detector.py).  GPU: the real autograd Functions over the HIP
kernels, two processes sharing the one card of the test box, gloo carrying the device tensors (RCCL refuses two ranks on one
device; the reducer, bucket and hook logic is the same)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WATCH = ["backbone.layer2.0.conv1.weight", "backbone.layer4.2.conv3.weight", "neck.lateral_convs.0.conv.weight", "neck.fpn_convs.0.conv.weight",
         "mapping.0.weight", "mapping.0.bias", "nerf_mlp.mlp.base.hidden_layers.0.weight", "nerf_mlp.mlp.sigma_layer.output_layer.weight",
         "nerf_mlp.mlp.rgb_layer.output_layer.weight", "neck_3d.down_layer_0.0.conv1.weight", "neck_3d.out_block_2.0.weight",
         "bbox_head.cls_conv.weight", "bbox_head.reg_conv.weight", "bbox_head.centerness_conv.weight"]
DEAD = ["cov.0.weight", "cov.4.bias", "mean_mapping.0.weight", "cov_mapping.0.bias", "mapping_2d.0.weight", "neck.fpn_convs.1.conv.weight",
        "neck.fpn_convs.3.conv.bias"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(device):
    from nerfdet_amd.config import _wrap
    from nerfdet_amd.presets import nerfdet_cfg
    from nerfdet_amd.registry import build_detector
    torch.manual_seed(0)
    cfg = _wrap(nerfdet_cfg(50, n_voxels=(16, 16, 8), voxel_size=(0.4, 0.4, 0.4), depth_supervise=True))
    cfg["model"]["N_rand"], cfg["model"]["N_samples"] = 64, 12
    det = build_detector(cfg["model"], train_cfg=cfg["train_cfg"], test_cfg=cfg["test_cfg"])
    with torch.no_grad():
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
    return det.to(device).train()


def _scene(rank, device):
    from nerfdet_amd.synth import batch_to, train_scene
    return batch_to(train_scene(5, (64, 96), t_views=2, n_boxes=4, seed=10 + rank), device)


def _grads(det):
    named = dict(det.named_parameters())
    return {k: (None if named[k].grad is None else named[k].grad.detach().float().cpu().clone()) for k in WATCH + DEAD}


def _single_rank_reference(device, world, cpu_ops):
    """Per-scene gradients and losses from a plain (non-DDP) model, one scene after the other.  The head normalises its losses by
    ``reduce_mean(n_pos)`` (imvoxel_head_v2.py:174-175), the positives averaged over the ranks: a first pass records every scene's
    count, the second pass hands the head their mean -- what the all-reduce gives each rank of the real job."""
    import nerfdet_amd.head as H
    import nerfdet_amd.rays as R
    det = _build(device)
    seen, out = [], []
    saved = H._reduce_mean
    try:
        for mode in ("record", "replay"):
            H._reduce_mean = (lambda t: (seen.append(float(t)), t)[1]) if mode == "record" else (lambda t: t.new_tensor(sum(seen) / len(seen)))
            for r in range(world):
                det.zero_grad(set_to_none=True)
                if cpu_ops is not None:
                    cpu_ops["rng"] = np.random.RandomState(234)
                else:
                    R.rng = np.random.RandomState(234)
                torch.manual_seed(1)
                res = det.train_step(_scene(r, device))
                if mode == "replay":
                    res["loss"].backward()
                    out.append((_grads(det), res["log_vars"]))
    finally:
        H._reduce_mean = saved
    assert len(seen) == world and seen[0] != seen[1], "the scenes must differ in their positive counts for this to test anything"
    return out


def _worker(rank, world, port, dev_type, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import contextlib
    import torch.distributed as dist
    import nerfdet_amd.rays as R
    from nerfdet_amd import dist as D
    from nerfdet_amd.train import build_optimizer, train_one_step, wrap_ddp
    device = torch.device("cuda", 0) if dev_type == "cuda" else torch.device("cpu")
    if dev_type == "cpu":
        torch.set_num_threads(2)
    D.init_dist("gloo")
    ctx = contextlib.nullcontext()
    if dev_type == "cpu":
        from cpu_detector import oracle_backed_cpu_ops
        ctx = oracle_backed_cpu_ops()
    with ctx as holder:
        det = _build(device)
        ddp = wrap_ddp(det, device if dev_type == "cuda" else None)
        if holder is not None:
            holder["rng"] = np.random.RandomState(234)
        else:
            R.rng = np.random.RandomState(234)
        torch.manual_seed(1)
        res = ddp.train_step(_scene(rank, device))
        res["loss"].backward()
        grads, logs = _grads(det), res["log_vars"]
        # a full optimizer step through the same wrapper must work too (clip + AdamW with the backbone lr multiplier)
        if holder is not None:
            holder["rng"] = np.random.RandomState(234)
        opt = build_optimizer(ddp)
        assert len(opt.param_groups) == 2 and abs(opt.param_groups[1]["lr"] - 2e-5) < 1e-12
        step = train_one_step(ddp, _scene(rank, device), opt)
        w0 = det.mapping[0].weight.detach().float().cpu().clone()
    # numpy, pickled by value: torch tensors would travel as shared-memory handles that die with this process
    q.put((rank, {k: (None if v is None else v.numpy()) for k, v in grads.items()}, logs, float(step["grad_norm"]), w0.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(dev_type, tol):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dev_type, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            rank, grads, logs, norm, w0 = q.get(timeout=400)
            got[rank] = ({k: (None if v is None else torch.from_numpy(v)) for k, v in grads.items()}, logs, norm, torch.from_numpy(w0))
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    # single-process reference, same scenes
    if dev_type == "cpu":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from cpu_detector import oracle_backed_cpu_ops
        with oracle_backed_cpu_ops() as holder:
            ref = _single_rank_reference(torch.device("cpu"), world, holder)
    else:
        ref = _single_rank_reference(torch.device("cuda", 0), world, None)
    worst = 0.0
    for k in WATCH:
        mean = (ref[0][0][k] + ref[1][0][k]) / 2
        assert float(mean.abs().max()) > 0, k
        for r in range(world):
            g = got[r][0][k]
            assert g is not None, k
            err = float((g - mean).abs().max()) / max(float(mean.abs().max()), 1e-12)
            worst = max(worst, err)
            assert err <= tol, f"{k}: rank {r} gradient differs from the mean of the single-rank gradients by {err:.2e}"
    for k in DEAD:                                   # never touched by forward: no gradient, and no reducer hang either
        assert all(got[r][0][k] is None or float(got[r][0][k].abs().max()) == 0 for r in range(world)), k
        assert ref[0][0][k] is None
    for key in ("loss_cls", "loss_bbox", "loss_centerness", "loss_nvs", "loss_depth", "loss"):
        want = (ref[0][1][key] + ref[1][1][key]) / 2
        assert abs(got[0][1][key] - want) <= 1e-3 * max(1.0, abs(want)) and got[0][1][key] == got[1][1][key], key
    assert abs(got[0][2] - got[1][2]) <= 1e-4 * got[0][2]      # same clipped-gradient norm on both ranks
    assert torch.equal(got[0][3], got[1][3]) or float((got[0][3] - got[1][3]).abs().max()) < 1e-7   # replicas stay in step
    print(f"DDP 2 ranks ({dev_type}): worst gradient difference {worst:.2e} of the gradient scale (tolerance {tol:.0e})")


@pytest.mark.timeout(900)
def test_ddp_two_ranks_gloo_cpu_gradients_are_the_rank_mean():
    _run("cpu", 2e-4)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_ddp_two_ranks_hip_autograd_functions_gradients_are_the_rank_mean(device):
    # two repeated single-process steps differ by <= 1.1e-6 of the gradient scale (float atomics; measured by tests/rccl_single_rank.py on
    # MI355X, round 3); the rank mean adds one fp32 rounding per element
    _run("cuda", 5e-5)


def _child_env(port):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_rccl_single_rank_bringup(device):
    """RCCL itself, on the one GPU there is: a fresh child process joins an "nccl" process group of world size 1 and runs DDP's bucketed
    gradient all-reduce, the head's reduce_mean and the loss logging over it (tests/rccl_single_rank.py).  Gradients through DDP/RCCL
    must equal the plain step's within 3x the measured run-to-run noise of the atomics (floor 2e-4 of the gradient scale)."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank.py")], env=_child_env(_free_port()), capture_output=True, text=True,
                         timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["ok"] and rep["backend"] == "nccl" and rep["dead_ok"] and rep["finite"]
    assert abs(rep["loss_plain"] - rep["loss_ddp"]) <= 1e-4 * max(1.0, abs(rep["loss_plain"]))
    for k in WATCH:
        assert rep["err"][k] <= max(3 * rep["noise"][k], 2e-4), f"{k}: DDP-over-RCCL gradient differs by {rep['err'][k]:.2e} (noise {rep['noise'][k]:.2e})"
    print("RCCL single-rank bring-up:", {k: (f"{rep['err'][k]:.1e}", f"{rep['noise'][k]:.1e}") for k in WATCH}, "rccl", rep["rccl"])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_under_torch_distributed_run_uses_rccl(device):
    """The driver's multi-GPU launch line with one rank: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 1 ...`` -- bench.py joins the nccl group, its barrier and max-over-ranks run on RCCL."""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "tiny", "--no-cpu-baseline",
           "--no-serving"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["collectives"].startswith("nccl") and line["value"] > 0 and line["median_ms"] > 0


def test_ddp_bucket_plan_of_the_shipped_model():
    """the bucket plan DDP builds for the 433 MB of fp32 gradients at train.wrap_ddp's 128 MB cap, computed with DDP's own
    assignment routine on the real parameter list (DDP closes a bucket once it has reached the cap, so a bucket may exceed it by its last
    tensor -- the 113 MB weight of the neck's 1024-channel 3x3x3 layer): 4 all-reduces of 7 / 184 / 136 / 106 MB, in the order the
    gradients become ready (head and 3D neck first), where the default 25 MB cap gives 10.  Printed by tools/bench_train.py on rank 0."""
    from nerfdet_amd.presets import build_nerfdet
    from nerfdet_amd.train import ddp_bucket_plan
    det = build_nerfdet(50)
    plan = ddp_bucket_plan(det)
    total = sum(p.numel() * 4 for p in det.parameters() if p.requires_grad)
    assert sum(plan) == total and 400e6 < total < 470e6, total
    mb = [round(b / 1e6) for b in plan]
    assert len(plan) == 4 and mb[0] <= 8 and all(100 <= b <= 190 for b in mb[1:]), mb
    assert len(ddp_bucket_plan(det, 25)) == 10                       # what the default cap would have cost
