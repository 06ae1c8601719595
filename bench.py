#!/usr/bin/env python3
"""Headline benchmark: scenes/s of the NeRF-Det volumetric path on BASELINE.json configs[1]
(50 views 240x320 -> 60x80x256 FPN features, 40x40x16 voxels, fp32, 1 scene per step per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU.  Scenes are independent units, so N ranks run N scene streams with no data-path
collective (weak scaling); the only collectives are the timing barrier and the max-over-ranks of the time.
Inputs are resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)

WORKLOADS = {
    # name: n_views, img (H,W), C, n_voxels, voxel_size
    "cfg2": dict(n_views=50, img_hw=(240, 320), channels=256, n_voxels=(40, 40, 16), voxel_size=(0.16, 0.16, 0.2)),
    "cfg5": dict(n_views=101, img_hw=(320, 480), channels=256, n_voxels=(80, 80, 32), voxel_size=(0.16, 0.16, 0.2)),
    "tiny": dict(n_views=6, img_hw=(64, 96), channels=64, n_voxels=(12, 12, 6), voxel_size=(0.5, 0.5, 0.5)),
}


def k1_algorithmic_bytes(w):
    """read every feature row once + write (C + count) per voxel (count is int64 here)."""
    n = w["n_voxels"][0] * w["n_voxels"][1] * w["n_voxels"][2]
    hf, wf = w["img_hw"][0] // 4, w["img_hw"][1] // 4
    return w["n_views"] * w["channels"] * hf * wf * 4 + (w["channels"] * 4 + 8) * n


def synth_scene(w, seed, device):
    """SURVEY.md 8(d) generator: ring cameras, N(0,1) features, U[0,1) de-normalised images."""
    from nerfdet_amd.synth import ring_scene_meta
    g = torch.Generator().manual_seed(seed)
    meta = ring_scene_meta(w["n_views"], w["img_hw"])
    hf, wf = w["img_hw"][0] // 4, w["img_hw"][1] // 4
    feats = torch.randn(w["n_views"], w["channels"], hf, wf, generator=g)
    rgb = torch.rand(w["n_views"], 3, *w["img_hw"], generator=g)
    return meta, feats, rgb


def build_modules(w, device):
    from nerfdet_amd.nerf_mlp import VanillaNeRFRadianceField
    torch.manual_seed(0)
    cm = w["channels"] // 8
    mapping = torch.nn.Sequential(torch.nn.Linear(w["channels"], cm))
    mlp = VanillaNeRFRadianceField(4, 256, 3, 2 * (cm + 3), 1, 128)
    with torch.no_grad():
        mapping[0].bias.normal_(0, 0.5)
    return mapping.to(device).eval(), mlp.to(device).eval()


def cpu_baseline(w, mapping, mlp, repeats=3):
    """The oracle (PyTorch-CPU restatement of the reference, materialised volume and all) on the host cores."""
    from oracle import nerfdet_oracle as O
    cores = os.cpu_count() or 1
    cores = min(cores, 32)  # measured on the GPU box: 16-32 threads is the knee, 256 is 50x slower
    torch.set_num_threads(cores)
    meta, feats, rgb = synth_scene(w, 0, "cpu")
    wt, bs = mapping[0].weight.detach().cpu(), mapping[0].bias.detach().cpu()
    sd = {k: v.detach().cpu() for k, v in mlp.state_dict().items()}
    times = []
    with torch.no_grad():
        for i in range(repeats + 1):
            t0 = time.perf_counter()
            O.extract_volume(feats, rgb, meta, w["n_voxels"], w["voxel_size"], wt, bs, sd)
            times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return dict(value=1.0 / t, unit="scenes/s", cores=cores, kind="port",
                sample=f"{repeats} scenes after 1 warm-up, median; oracle.extract_volume = reference steps 2-11 "
                       f"(projection, backproject x2, mean/var, density MLP, gating) at the same shape, fp32")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from nerfdet_amd import ops
    from nerfdet_amd.volume import extract_volume
    w = WORKLOADS[args.workload]
    meta, feats, rgb = synth_scene(w, rank, device)
    feats = feats.to(device).contiguous(memory_format=torch.channels_last)
    rgb = rgb.to(device)
    mapping, mlp = build_modules(w, device)

    k1_events = []

    def step(record):
        with torch.no_grad():
            # same as extract_volume, with event pairs around the dominant kernel
            if not record:
                return extract_volume(feats, rgb, meta, w["n_voxels"], w["voxel_size"], mapping, mlp)
            orig = ops.backproject_aggregate

            def timed(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig(*a, **k)
                e1.record()
                k1_events.append((e0, e1))
                return r
            import nerfdet_amd.volume as V
            V.ops.backproject_aggregate = timed
            try:
                return extract_volume(feats, rgb, meta, w["n_voxels"], w["voxel_size"], mapping, mlp)
            finally:
                V.ops.backproject_aggregate = orig

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step(False)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    k1_ms = sorted(e0.elapsed_time(e1) for e0, e1 in k1_events)
    k1_avg_ms = sum(k1_ms) / len(k1_ms)
    abytes = k1_algorithmic_bytes(w)
    achieved = abytes / (k1_avg_ms * 1e-3) / 1e9

    if rank == 0:
        res = {
            "metric": "scenes/sec (50-view 240x320, 40x40x16 voxels)",
            "value": world * args.steps / dt,
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: hot path steps 2-11 of extract_feat (FPN features resident -> gated "
                                   f"voxel volume + view count), {w['n_views']} views {w['img_hw'][0]}x{w['img_hw'][1]}, "
                                   f"{'x'.join(map(str, w['n_voxels']))} voxels, 1 scene/step/GPU",
                       "scenes_per_step": world, "parallelism": f"scene replicas x{world}"},
            "roofline": {"kernel": "k_backproject_aggregate", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes": abytes, "avg_launch_ms": k1_avg_ms, "median_launch_ms": k1_ms[len(k1_ms) // 2]},
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(w, mapping, mlp)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
