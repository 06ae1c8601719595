"""Training-step timing at BASELINE configs[2] shapes (SURVEY.md 8d "cfg3"): per rank one scene per step -- 50 sampled views of which
10 become NeRF targets -> 40 source views 240x320, 40x40x16 voxels, 2048 rays x 64 samples, all five losses, backward, gradient
clipping, AdamW.  ``--gpus N`` starts N ranks (one process per GPU, DistributedDataParallel over RCCL, nerfdet_amd/train.py);
the step time is the slowest rank's.  Not the headline metric (bench.py is).  Reports ms/step, scenes/s over all ranks and the
event-timed forward launches of the packed ray sampler K4 (algorithmic bytes per SURVEY.md 8d) against the HBM roofline.

    python tools/bench_train.py [--gpus N] [--steps K] [--warmup W] [--arith f32|bf16x3|bf16] [--depth-supervise 0|1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def launch(args):
    from nerfdet_amd.dist import launch_local_ranks      # the first failing rank terminates the others instead of leaving them in a collective
    return launch_local_ranks(__file__, sys.argv[1:], args.gpus)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arith", default=None)
    ap.add_argument("--depth-supervise", type=int, default=1)
    ap.add_argument("--views", type=int, default=40)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args))
    from nerfdet_amd import dist as D, rays, trace
    from nerfdet_amd.presets import build_nerfdet
    from nerfdet_amd.synth import batch_to, train_scene
    from nerfdet_amd.train import build_optimizer, train_one_step, wrap_ddp
    import nerfdet_amd.conv3d as C3
    rank, world, local = D.init_dist("nccl")
    dev = torch.device("cuda", local)
    if args.arith:
        C3.set_arithmetic(args.arith)
    torch.manual_seed(0)
    det = build_nerfdet(50, depth_supervise=bool(args.depth_supervise))
    with torch.no_grad():
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
    det.to(dev).train()
    model = wrap_ddp(det, dev) if world > 1 else det
    data = batch_to(train_scene(args.views, (240, 320), t_views=10, n_boxes=8, seed=rank), dev)
    opt = build_optimizer(model)
    rec = trace.Recorder()
    for _ in range(args.warmup):
        out = train_one_step(model, data, opt)
    trace.recorder = rec
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = train_one_step(model, data, opt)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = D.max_over_ranks((time.perf_counter() - t0) / args.steps, dev)
    trace.recorder = None
    if rank == 0:
        spans = rec.span_ms()
        k4 = spans.get("k_ray_stats_packed") or spans.get("k_ray_view_stats") or []
        line = dict(workload=f"cfg3 shapes, {world} GPU(s) x 1 scene/step, {C3.ARITHMETIC} convolutions: {args.views} source views 240x320, 2048 rays x 64 samples, "
                             f"{'5' if args.depth_supervise else '4'} losses + backward + clip + AdamW" + (", DDP over RCCL" if world > 1 else ""),
                    n_gpus=world, ms_per_train_step=dt * 1e3, scenes_per_s=world / dt, log_vars=out["log_vars"], grad_norm=out.get("grad_norm"),
                    peak_mem_GB=torch.cuda.max_memory_allocated() / 1e9)
        if k4:
            ms = sorted(m for m, _ in k4)
            b = k4[0][1]["bytes"]
            line["roofline_k4_forward"] = dict(kernel="k_ray_stats_packed (K4: Projector.compute + compute_mask_points fused)", bound="hbm",
                                               algorithmic_bytes=b, median_launch_ms=ms[len(ms) // 2], achieved=b / ms[len(ms) // 2] / 1e6,
                                               peak=8000.0, unit="GB/s", frac=b / ms[len(ms) // 2] / 1e6 / 8000.0)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
