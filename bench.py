#!/usr/bin/env python3
"""Headline benchmark: scenes/s of NeRF-Det inference on BASELINE.json configs[1]
(nerfdet_res50_2x_low_res, 50 views 240x320, 40x40x16 voxels, fp32, 1 scene per step per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step is one full ``nerfdet.forward_test`` on one synthetic scene: ResNet-50 + FPN (PyTorch-ROCm / MIOpen), the
hand-written HIP hot path (projection, gather, multi-view aggregation, density MLP gating), the 3D neck + head,
box decoding and HIP NMS, results copied to the host as the reference does.  Nothing is skipped or cached
between steps.  Inputs are resident in HBM when the timed region starts.

One process per GPU.  Scenes are independent units, so N ranks run N scene streams with no data-path collective
(weak scaling); the only collectives are the timing barrier and the max-over-ranks of the elapsed time.
Rank 0 prints ONE JSON line; `roofline` is for the dominant hand-written kernel of the step (the convolution instantiation
with the largest share: its algorithmic FLOPs / its event-timed launches, against the matrix-core peak of its arithmetic),
`roofline_all_convolutions` aggregates every convolution launch, `roofline_projection` is the fused backproject+aggregate
kernel of the volumetric path against HBM; all timed with events on the stream the kernels are launched on, inside the
timed region.
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4-copy ceiling)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32-input MFMA peak (v_mfma_f32_16x16x4_f32), = fp32 vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)

WORKLOADS = {
    "cfg2": dict(n_views=50, img_hw=(240, 320), channels=256, n_voxels=(40, 40, 16), voxel_size=(0.16, 0.16, 0.2), depth=50),
    "cfg1": dict(n_views=10, img_hw=(240, 320), channels=256, n_voxels=(40, 40, 16), voxel_size=(0.16, 0.16, 0.2), depth=50),
    "cfg5": dict(n_views=101, img_hw=(240, 320), channels=256, n_voxels=(80, 80, 32), voxel_size=(0.08, 0.08, 0.1), depth=50),
    "tiny": dict(n_views=6, img_hw=(64, 96), channels=256, n_voxels=(16, 16, 8), voxel_size=(0.4, 0.4, 0.4), depth=50),
}


def k1_algorithmic_bytes(w):
    """every FPN feature row read once + (C fp32 + int64 count) written per voxel."""
    n = w["n_voxels"][0] * w["n_voxels"][1] * w["n_voxels"][2]
    hf, wf = w["img_hw"][0] // 4, w["img_hw"][1] // 4
    return w["n_views"] * w["channels"] * hf * wf * 4 + (w["channels"] * 4 + 8) * n


def k1_measured_traffic(workload):
    """HBM-side bytes per K1 launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE doubled as the
    gfx950 note in MI355X_MICROARCH.md prescribes, + WRITE_SIZE); None when no profile matches the workload."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")), reverse=True):  # newest round first
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("workload") == workload:
                return int(d["k1"]["traffic_bytes"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def synth_batch(w, seed):
    """ring cameras, img ~ N(0,1), denorm_images ~ U[0,1), one dummy NeRF target view
    (rays are unused at inference unless render_testing)."""
    from nerfdet_amd.synth import ring_scene_meta
    g = torch.Generator().manual_seed(seed)
    h, wd = w["img_hw"]
    meta = ring_scene_meta(w["n_views"], w["img_hw"])
    return dict(img=torch.randn(1, w["n_views"], 3, h, wd, generator=g), img_metas=[meta],
                denorm_images=torch.rand(1, w["n_views"], 3, h, wd, generator=g),
                lightpos=torch.zeros(1, 1, 4, 3), raydirs=torch.ones(1, 1, 4, 3), gt_images=torch.zeros(1, 1, 4, 3),
                gt_depths=[], nerf_sizes=[torch.tensor([[2, 2, 3]])])


def build_model(w):
    """Seeded random-init nerfdet (no checkpoints offline).  A raw random init is a degenerate workload (density 0
    everywhere, every score below score_thr -> NMS sees nothing), so the weights are nudged into a trained-like
    regime: O(1) FPN features, positive densities, a few hundred NMS candidates.  Costs are data independent
    except for NMS."""
    from nerfdet_amd.presets import build_nerfdet
    torch.manual_seed(0)
    det = build_nerfdet(w["depth"], n_voxels=w["n_voxels"], voxel_size=w["voxel_size"])
    with torch.no_grad():
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(2.0)
        det.bbox_head.cls_conv.weight.normal_(0, 0.3)
        det.bbox_head.cls_conv.bias.fill_(-2.0)
        det.bbox_head.centerness_conv.weight.normal_(0, 0.1)
        det.bbox_head.reg_conv.weight.normal_(0, 0.05)
        det.mapping[0].bias.normal_(0, 0.3)
    return det.eval()


def to_device(batch, device):
    out = {}
    for k, v in batch.items():
        out[k] = v.to(device) if isinstance(v, torch.Tensor) else v
    return out


def cpu_baseline(w, det_cpu, batch, scenes=2):
    """The reference's algorithm on the host cores: the oracle (PyTorch-CPU restatement: materialised per-view
    volume, Python per-view loops, sequential NMS) for the volumetric path, 3D neck, head and NMS, plus the same
    ResNet-50+FPN modules run by PyTorch-CPU."""
    from oracle import nerfdet_oracle as O
    cores = min(os.cpu_count() or 1, 32)  # measured on the GPU box: 16-32 threads is the knee (256 is 50x slower)
    torch.set_num_threads(cores)
    meta = batch["img_metas"][0]
    tc = det_cpu.bbox_head.test_cfg
    sd_mlp = det_cpu.nerf_mlp.state_dict()
    sd_n3 = dict(det_cpu.neck_3d.state_dict())
    sd_head = det_cpu.bbox_head.state_dict()
    times = []
    with torch.no_grad():
        for _ in range(scenes + 1):
            t0 = time.perf_counter()
            feats = det_cpu.neck(det_cpu.backbone(batch["img"][0]))[0]
            ov = O.extract_volume(feats, batch["denorm_images"][0], meta, w["n_voxels"], w["voxel_size"],
                                  det_cpu.mapping[0].weight, det_cpu.mapping[0].bias, sd_mlp)
            n3 = O.neck3d_forward(sd_n3, ov["volume"].unsqueeze(0))
            ctr, reg, cls = O.head_forward(sd_head, n3)
            O.head_get_bboxes(ctr, reg, cls, ov["valid"].unsqueeze(0).float(), meta["lidar2img"]["origin"], w["voxel_size"],
                              tc.nms_pre, tc.score_thr, tc.iou_thr)
            times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return dict(value=1.0 / t, unit="scenes/s", cores=cores, kind="port",
                sample=f"{scenes} full scenes after 1 warm-up (median {t:.2f} s/scene): ResNet-50+FPN (PyTorch-CPU) + oracle "
                       f"volumetric path + 3D neck + head + sequential NMS, same shapes and weights, fp32")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the static part of the step from hipGraphs (nerfdet_amd/graphed.py); measured equal to eager "
                         "launches within 1 %% on MI355X: the step is GPU-bound, launch-ahead already hides the gaps")
    ap.add_argument("--conv-arithmetic", default=None, choices=["f32", "bf16x3"],
                    help="convolution kernel family (default: the package default, nerfdet_amd.conv3d.ARITHMETIC)")
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

    import nerfdet_amd.volume as V
    w = WORKLOADS[args.workload]
    det = build_model(w)
    batch_cpu = synth_batch(w, rank)
    det_gpu = det.to(device)
    batch = to_device(batch_cpu, device)

    # event pairs around the dominant hand-written kernel and around the stages, all on torch's current stream,
    # which is the stream the C ABI launches on
    k1_events, stage_events = [], {"backbone_fpn": [], "volumetric_hot_path": [], "neck3d": [], "head_nms": []}
    orig_k1 = V.ops.backproject_aggregate
    record = {"on": False, "step": 0}

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def timed_k1(*a, **k):
        if not record["on"]:
            return orig_k1(*a, **k)
        e0 = ev()
        r = orig_k1(*a, **k)
        k1_events.append((e0, ev()))
        return r
    V.ops.backproject_aggregate = timed_k1

    # every launch of the MFMA convolution (the dominant kernel of the step: 3D neck/head + ResNet/FPN bottlenecks)
    import nerfdet_amd.conv3d as C3
    if args.conv_arithmetic:
        C3.set_arithmetic(args.conv_arithmetic)
    conv_events = []

    def conv_hook(flops, thunk, kernel_name=""):
        # 150 event records per step cost ~2 % of the step: sample every 4th timed step (still inside the timed region)
        if not record["on"] or record["step"] % 4 != 0:
            return thunk()
        e0 = ev()
        r = thunk()
        conv_events.append((flops, e0, ev(), kernel_name))
        return r
    C3.launch_hook = conv_hook

    from nerfdet_amd.graphed import GraphedForwardTest
    graphed = GraphedForwardTest(det_gpu)

    def k1_hook(fn):
        if not record["on"]:
            return fn()
        e0 = ev()
        r = fn()
        k1_events.append((e0, ev()))
        return r
    graphed.k1_hook = k1_hook

    def step_graph():
        """nerfdet.forward_test with the static part replayed from two hipGraphs; the aggregation kernel is launched
        eagerly between them, bracketed by events."""
        return graphed(return_loss=False, **batch)

    def step_eager():
        """= nerfdet.forward_test (simple_test), with event markers between its stages when recording."""
        with torch.no_grad():
            if not record["on"]:
                return det_gpu(return_loss=False, **batch)
            rb = det_gpu._ray_batch(batch)
            e0 = ev()
            x, b, stride = det_gpu.extract_2d(batch["img"])
            e1 = ev()
            # as nerfdet.extract_feat does: per-scene constants computed on the host behind the backbone queue, async upload
            geom = V.scene_geometry(batch["img_metas"][0], det_gpu.n_voxels, det_gpu.voxel_size, stride, device)
            out = V.extract_volume(x, rb["denorm_images"][0], batch["img_metas"][0], det_gpu.n_voxels, det_gpu.voxel_size,
                                   det_gpu.mapping, det_gpu.nerf_mlp, stride=stride, channels_last_out=True, geometry=geom)
            e2 = ev()
            x3 = det_gpu.neck_3d(out["volume"].unsqueeze(0))
            e3 = ev()
            from nerfdet_amd.boxes import DepthInstance3DBoxes, bbox3d2result
            batch["img_metas"][0].setdefault("box_type_3d", DepthInstance3DBoxes)
            boxes = det_gpu.bbox_head.simple_test_fused(x3, out["valid"].unsqueeze(0).float(), batch["img_metas"])
            res = [bbox3d2result(*bx) for bx in boxes]
            e4 = ev()
            for name, a, c in (("backbone_fpn", e0, e1), ("volumetric_hot_path", e1, e2), ("neck3d", e2, e3), ("head_nms", e3, e4)):
                stage_events[name].append((a, c))
            return res

    step = step_graph if args.graph else step_eager

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        res = step()
    record["on"] = True
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        record["step"] = i
        res = step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if args.graph:  # stage breakdown from a few eager steps outside the timed region (events cannot sit inside a graph)
        n_k1 = len(k1_events)
        for _ in range(5):
            record["step"] = 0   # sample the per-launch conv events on each of these
            step_eager()
        torch.cuda.synchronize()
        del k1_events[n_k1:]
    k1_ms = sorted(a.elapsed_time(b) for a, b in k1_events)
    k1_avg_ms = sum(k1_ms) / len(k1_ms)
    if C3.ARITHMETIC == "bf16x3":
        conv_kernel = ("k_conv_split (implicit-GEMM convolution on the bf16 matrix cores, fp32 operands split exactly into 3 bf16 terms, "
                       "6 MFMA products per multiply, fp32 accumulate: 3D neck + head, ResNet/FPN; all tile instantiations, split-K "
                       "reduce launches included in the event spans)")
        conv_peak = MFMA_BF16_PEAK_TFLOPS / 6.0
        conv_peak_note = ("achieved = algorithmic fp32 convolution FLOPs / time; peak = dense bf16 MFMA peak 2500 TFLOP/s / 6 issued "
                          "products per algorithmic multiply-add")
    else:
        conv_kernel = ("k_conv3d_igemm (fp32-MFMA implicit-GEMM convolution: 3D neck + head, ResNet/FPN bottlenecks; both tile "
                       "instantiations, split-K reduce launches included in the event spans)")
        conv_peak = MFMA_F32_PEAK_TFLOPS
        conv_peak_note = "dense fp32-input MFMA peak"
    abytes = k1_algorithmic_bytes(w)
    k1_traffic, k1_traffic_src = k1_measured_traffic(args.workload)
    achieved = abytes / (k1_avg_ms * 1e-3) / 1e9
    stages = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in stage_events.items()}
    conv_ms = [a.elapsed_time(b) for _, a, b, _ in conv_events]
    conv_flops = sum(f for f, _, _, _ in conv_events)
    conv_tflops = conv_flops / (sum(conv_ms) * 1e-3) / 1e12
    n_conv_steps = max(1, len({i for i in range(args.steps) if i % 4 == 0}) if not args.graph else 5)
    # the dominant single kernel: the instantiation with the largest share of the step
    by_kernel = {}
    for (f, _, _, name), ms in zip(conv_events, conv_ms):
        g = by_kernel.setdefault(name, [0.0, 0.0, 0])
        g[0] += f; g[1] += ms; g[2] += 1
    dom_name, (dom_flops, dom_ms, dom_n) = max(by_kernel.items(), key=lambda kv: kv[1][1])
    dom_tflops = dom_flops / (dom_ms * 1e-3) / 1e12

    if rank == 0:
        out = {
            "metric": f"scenes/sec ({w['n_views']}-view {w['img_hw'][0]}x{w['img_hw'][1]}, {'x'.join(map(str, w['n_voxels']))} voxels)",
            "value": world * args.steps / dt,
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if C3.ARITHMETIC == "f32" else "f32 (convolutions: fp32 operands as exact 3-term bf16 sums on the bf16 MFMA, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: nerfdet_res{w['depth']}_2x_low_res forward_test, {w['n_views']} views "
                                   f"{w['img_hw'][0]}x{w['img_hw'][1]}, {'x'.join(map(str, w['n_voxels']))} voxels, fp32, "
                                   f"1 scene/step/GPU, random-init weights",
                       "scenes_per_step": world, "parallelism": f"scene replicas x{world} (no data-path collective)"},
            "roofline": {"kernel": f"{dom_name} (the convolution instantiation with the largest share of the step; event spans include the "
                                   f"split-K reduce launch where a layer splits K)",
                         "bound": "mfma", "achieved": dom_tflops, "peak": conv_peak, "unit": "TFLOP/s",
                         "frac": dom_tflops / conv_peak, "traffic": None, "peak_note": conv_peak_note,
                         "algorithmic_flops_per_launch": dom_flops / dom_n, "launches_per_step": dom_n / n_conv_steps,
                         "avg_launch_ms": dom_ms / dom_n, "total_ms_per_step": dom_ms / n_conv_steps, "sampled_steps": n_conv_steps},
            "roofline_all_convolutions": {"kernel": conv_kernel,
                         "bound": "mfma", "achieved": conv_tflops, "peak": conv_peak, "unit": "TFLOP/s",
                         "frac": conv_tflops / conv_peak, "traffic": None, "peak_note": conv_peak_note,
                         "algorithmic_flops_per_step": conv_flops / n_conv_steps, "launches_per_step": len(conv_events) / n_conv_steps,
                         "avg_launch_ms": sum(conv_ms) / len(conv_ms), "total_ms_per_step": sum(conv_ms) / n_conv_steps,
                         "sampled_steps": n_conv_steps,
                         "per_kernel": {k: {"launches_per_step": v[2] / n_conv_steps, "avg_launch_ms": v[1] / v[2],
                                            "tflops": v[0] / (v[1] * 1e-3) / 1e12} for k, v in sorted(by_kernel.items())}},
            "roofline_projection": {"kernel": "k_backproject_aggregate (fused backproject + view mean/count + alpha gating)",
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": k1_traffic,
                         "traffic_source": f"{k1_traffic_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                         "algorithmic_bytes": abytes,
                         "avg_launch_ms": k1_avg_ms, "median_launch_ms": k1_ms[len(k1_ms) // 2]},
            "execution": "hipGraph replay (2 graphs) + eager K1 + eager post-processing" if args.graph else "eager launches",
            "stages_ms": stages,
            "detections_last_step": int(len(res[0]["scores_3d"])),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w, build_model(w), batch_cpu)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
