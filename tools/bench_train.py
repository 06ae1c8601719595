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


def dry_run(args):
    """The launcher / rendezvous / DDP plumbing of this script without a GPU (tests/test_ddp.py): gloo, the package's own ``wrap_ddp`` +
    ``build_optimizer`` + ``train_one_step`` around a small stand-in detector, per-rank CPU share, one JSON line from rank 0."""
    from nerfdet_amd import dist as D
    from nerfdet_amd.detector import BaseDetector
    from nerfdet_amd.train import build_optimizer, ddp_bucket_plan, train_one_step, wrap_ddp
    rank, world, local = D.init_dist("gloo")
    share = D.apply_rank_affinity(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))

    class Tiny(BaseDetector):          # train_step / _parse_losses are the product's; the network is a stand-in
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Linear(64, 256)
            self.neck_3d = torch.nn.Sequential(torch.nn.Linear(256, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 256))
            self.dead = torch.nn.Linear(4, 4)      # never used: find_unused_parameters=True must cope

        def forward(self, img, img_metas, return_loss=True):
            y = self.neck_3d(torch.relu(self.backbone(img)))
            return dict(loss_a=y.pow(2).mean(), loss_b=y.abs().mean())

    torch.manual_seed(0)
    det = Tiny()
    grouped = torch.distributed.is_initialized()
    model = wrap_ddp(det, None, bucket_cap_mb=1) if grouped else det
    opt = build_optimizer(model)
    g = torch.Generator().manual_seed(rank)
    data = dict(img=torch.randn(8, 64, generator=g), img_metas=[{}])
    for _ in range(args.warmup):
        out = train_one_step(model, data, opt)
    if grouped:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = train_one_step(model, data, opt)
    if grouped:
        torch.distributed.barrier()
    dt = D.max_over_ranks((time.perf_counter() - t0) / args.steps)
    w0 = det.backbone.weight.detach().clone()
    if grouped:     # every rank must hold the same parameters after the all-reduced steps
        ws = [torch.empty_like(w0) for _ in range(world)]
        torch.distributed.all_gather(ws, w0)
        assert all(torch.allclose(ws[0], w, rtol=0, atol=1e-7) for w in ws), "ranks diverged: the gradients were not all-reduced"
    if rank == 0:
        print(json.dumps(dict(metric="dry-run", value=world / dt, unit="steps/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt * 1e3,
                              dry_run=True, log_vars=out["log_vars"], ddp_buckets_bytes=ddp_bucket_plan(det, 1), rank_threads=share["threads"],
                              rank_cpus=len(share["cpus"]))))
    if grouped:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arith", default=None)
    ap.add_argument("--depth-supervise", type=int, default=1)
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--lazy-log", type=int, default=1, help="1: a step's logged scalars are read from pinned memory after the next step has been queued "
                                                            "(train.StepLog); 0: host floats at the end of every step (one device drain per step)")
    ap.add_argument("--implicit-min-taps", type=int, default=None, help=argparse.SUPPRESS)      # measurement: conv_train.IMPLICIT_MIN_TAPS
    ap.add_argument("--threaded-draw", type=int, default=0, help=argparse.SUPPRESS)             # measurement: rays.THREADED_DRAW
    ap.add_argument("--train-f16x2", type=int, default=1, help="0: the training convolutions on the six-product bf16x3 kernels")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args))
    if args.dry_run:
        return dry_run(args)
    from nerfdet_amd import dist as D, rays, trace
    from nerfdet_amd.presets import build_nerfdet
    from nerfdet_amd.synth import batch_to, train_scene
    from nerfdet_amd.train import build_optimizer, train_one_step, wrap_ddp
    import nerfdet_amd.conv3d as C3
    rank, world, local = D.init_dist("nccl")
    D.apply_rank_affinity(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    dev = torch.device("cuda", local)
    if args.arith:
        C3.set_arithmetic(args.arith)
    C3.TRAIN_F16X2 = bool(args.train_f16x2)
    rays.THREADED_DRAW = bool(args.threaded_draw)
    if args.implicit_min_taps is not None:
        import nerfdet_amd.conv_train as CT
        CT.IMPLICIT_MIN_TAPS = args.implicit_min_taps
    torch.manual_seed(0)
    det = build_nerfdet(50, depth_supervise=bool(args.depth_supervise))
    with torch.no_grad():
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
    det.to(dev).train()
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()      # also a single rank under a launcher
    model = wrap_ddp(det, dev) if grouped else det
    data = batch_to(train_scene(args.views, (240, 320), t_views=10, n_boxes=8, seed=rank), dev)
    opt = build_optimizer(model)
    state = {"step": 0}
    # ~450 convolution launches per step (forward, data and weight gradients): their event pairs are sampled on every 4th timed step
    rec = trace.Recorder(sample=lambda name: (not name.startswith(("k_conv", "k_bottleneck", "k_point_mlp", "f32:", "bf16x3:", "bf16:", "f16x2:"))) or state["step"] % 4 == 0)
    for _ in range(args.warmup):
        out = train_one_step(model, data, opt)
    import gc
    gc.collect()
    gc.freeze()          # the model's long-lived Python objects out of the cyclic collector's reach for the timed steps (bench.py does the same)
    trace.recorder = rec
    if grouped:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ticks = [t0]
    prev = None
    for i in range(args.steps):
        state["step"] = i
        cur = train_one_step(model, data, opt, lazy_log=bool(args.lazy_log))     # lazy: step i's scalars are read after step i + 1 is queued
        if args.lazy_log:
            if prev is not None:
                prev["log"].get()
            prev = cur
        else:
            out = cur
        ticks.append(time.perf_counter())
    torch.cuda.synchronize()
    if args.lazy_log:
        vals = prev["log"].get()
        out = dict(grad_norm=vals.pop("grad_norm", None), log_vars=vals)
    if grouped:
        torch.distributed.barrier()
    dt = D.max_over_ranks((time.perf_counter() - t0) / args.steps, dev)
    trace.recorder = None
    # the same step with its scalars read on the host at the end of EVERY step (what mmdet's _parse_losses does with .item()): a few steps after
    # the timed region, reported next to the figure above so that nobody has to guess what the lazy read is worth
    dt_sync = None
    if args.lazy_log:
        n_sync = max(2, min(4, args.steps))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_sync):
            train_one_step(model, data, opt)
        torch.cuda.synchronize()
        dt_sync = D.max_over_ranks((time.perf_counter() - t1) / n_sync, dev)
    if rank == 0:
        per_step = sorted((b - a) * 1e3 for a, b in zip(ticks, ticks[1:]))
        pct = lambda q: per_step[min(len(per_step) - 1, int(round(q * (len(per_step) - 1))))]
        n_conv_steps = len([i for i in range(args.steps) if i % 4 == 0])
        spans = rec.span_ms()
        arith = C3.train_arithmetic()        # fp16 pairs (3 products) unless conv3d.TRAIN_F16X2 is off (then bf16x3: 6; the frozen prefix stays "/f16x2")
        conv_peak = {"f16x2": 2500.0 / 3.0, "bf16x3": 2500.0 / 6.0, "bf16": 2500.0, "f32": 157.3}[arith]
        peak_of = lambda name: 2500.0 / 3.0 if name.endswith("/f16x2") else (2500.0 / 6.0 if arith == "f16x2" else conv_peak)
        peak_note = {"f16x2": "dense fp16 MFMA peak 2500 TFLOP/s / 3 issued products per algorithmic multiply-add (6 for the launches pinned to bf16x3)",
                     "bf16x3": "dense bf16 MFMA peak 2500 TFLOP/s / 6 issued products per algorithmic multiply-add (3 for the fp16-pair launches of the "
                               "frozen prefix)", "bf16": "dense bf16 MFMA peak", "f32": "dense fp32-input MFMA peak"}[arith]
        conv = {k: v for k, v in spans.items() if v and v[0][1].get("kind") == "conv"}
        by_kernel = {k: [sum(i["flops"] for _, i in v), sum(ms for ms, _ in v), len(v)] for k, v in conv.items()}
        line = dict(metric="training steps/sec (cfg3 shapes)", value=world / dt, unit="scenes/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=dt * 1e3, median_ms=pct(0.5), p10_ms=pct(0.1), p90_ms=pct(0.9), higher_is_better=True, scaling="weak", dtype=arith + (" (frozen prefix: f16x2)" if C3.ARITHMETIC == "f16x2" and arith != "f16x2" else ""),
                    data="synthetic",
                    config=dict(workload=f"cfg3 shapes, {world} GPU(s) x 1 scene/step, {arith} convolutions: {args.views} source views 240x320 + 10 NeRF target "
                                         f"views, 40x40x16 voxels, 2048 rays x 64 samples, {'5' if args.depth_supervise else '4'} losses + backward + clip + AdamW"
                                         + (", DDP over RCCL" if grouped else ""), scenes_per_step=world),
                    log_vars=out["log_vars"], grad_norm=out.get("grad_norm"), peak_mem_GB=torch.cuda.max_memory_allocated() / 1e9,
                    host_sync=("none inside the timed region's steps: each step's logged scalars are read from pinned memory after the next step has been "
                               "queued (train.StepLog), the rays with depth come from the loader (batch key depth_rays)" if args.lazy_log else
                               "the logged scalars are read on the host at the end of every step"))
        if dt_sync is not None:
            line["ms_per_step_host_read_every_step"] = dt_sync * 1e3
        if grouped:
            from nerfdet_amd.train import ddp_bucket_plan
            line["ddp_buckets_bytes"] = ddp_bucket_plan(det)
        if by_kernel:
            dom, (df, dms, dn) = max(by_kernel.items(), key=lambda kv: kv[1][1])
            tf = df / (dms * 1e-3) / 1e12
            line["roofline"] = dict(kernel=f"{dom} (the convolution instantiation with the largest share of the training step: forward, data-gradient and "
                                           f"weight-gradient launches all run on it)", bound="mfma", achieved=tf, peak=peak_of(dom), unit="TFLOP/s", frac=tf / peak_of(dom),
                                    traffic=None, peak_note=peak_note, launches_per_step=dn / n_conv_steps, avg_launch_ms=dms / dn,
                                    total_ms_per_step=dms / n_conv_steps, sampled_steps=n_conv_steps)
            cf, cms, cn = (sum(v[i] for v in by_kernel.values()) for i in range(3))
            all_peak = cf / sum(v[0] / peak_of(k) for k, v in by_kernel.items())      # FLOP-weighted harmonic mean of the launches' peaks
            line["roofline_all_convolutions"] = dict(bound="mfma", achieved=cf / (cms * 1e-3) / 1e12, peak=all_peak, unit="TFLOP/s",
                                                     frac=cf / (cms * 1e-3) / 1e12 / all_peak, launches_per_step=cn / n_conv_steps,
                                                     total_ms_per_step=cms / n_conv_steps, algorithmic_flops_per_step=cf / n_conv_steps,
                                                     per_kernel={k: dict(launches_per_step=v[2] / n_conv_steps, avg_launch_ms=v[1] / v[2],
                                                                         tflops=v[0] / (v[1] * 1e-3) / 1e12) for k, v in sorted(by_kernel.items())})
        k4 = spans.get("k_ray_stats_packed") or spans.get("k_ray_view_stats") or []
        if k4:
            ms = sorted(m for m, _ in k4)
            b = k4[0][1]["bytes"]
            med = ms[len(ms) // 2]
            line["roofline_k4_forward"] = dict(kernel="k_ray_stats_packed (K4: Projector.compute + compute_mask_points fused)", bound="hbm",
                                               algorithmic_bytes=b, median_launch_ms=med, achieved=b / med / 1e6, peak=8000.0, unit="GB/s",
                                               frac=b / med / 1e6 / 8000.0)
        k4b = spans.get("k_ray_stats_packed_bwd") or spans.get("k_ray_view_stats_bwd") or []
        if k4b:
            ms = sorted(m for m, _ in k4b)
            med = ms[len(ms) // 2]
            info = k4b[0][1]
            line["roofline_k4_backward"] = dict(kernel="k_ray_stats_packed<true> (K4 backward: scatter of d(mean), d(var) into the mapped feature maps)",
                                                bound="hbm", algorithmic_bytes=info["bytes"], median_launch_ms=med, achieved=info["bytes"] / med / 1e6,
                                                peak=8000.0, unit="GB/s", frac=info["bytes"] / med / 1e6 / 8000.0, atomics_upper_bound=info["atomics_max"],
                                                note="the scatter is bound by float-atomic throughput (~1.3 TB/s of 4-byte adds measured on MI355X), not by "
                                                     "the algorithmic bytes")
        print(json.dumps(line))
    if grouped:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
