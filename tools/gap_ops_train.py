"""Scratch (GPU box): torch.profiler over two training steps; for every GPU-idle gap > 0.8 ms inside a step, the host-side ops that ran during it."""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd.presets import build_nerfdet
from nerfdet_amd.synth import batch_to, train_scene
from nerfdet_amd.train import build_optimizer, train_one_step

dev = torch.device("cuda")
torch.manual_seed(0)
model = build_nerfdet(50, depth_supervise=True)
with torch.no_grad():
    model.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
    model.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
model.to(dev).train()
opt = build_optimizer(model)
data = batch_to(train_scene(40, (240, 320), t_views=10, n_boxes=8, seed=0), dev)
for _ in range(4):
    train_one_step(model, data, opt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=False) as prof:
    for _ in range(2):
        train_one_step(model, data, opt)
    torch.cuda.synchronize()
ev = prof.events()
from torch.autograd import DeviceType
kern = sorted([e for e in ev if e.device_type == DeviceType.CUDA], key=lambda e: e.time_range.start)
cpu = sorted([e for e in ev if e.device_type == DeviceType.CPU], key=lambda e: e.time_range.start)
print(len(kern), "device events,", len(cpu), "host events")
pe = kern[0].time_range.end
for k in kern[1:]:
    g = k.time_range.start - pe
    if g > 800:
        ops = [c for c in cpu if c.time_range.end > pe and c.time_range.start < k.time_range.start]
        # outermost ops only (not contained in another listed op)
        ops.sort(key=lambda c: (c.time_range.start, -c.time_range.end))
        top, last_end = [], -1
        for c in ops:
            if c.time_range.start >= last_end:
                top.append(c); last_end = c.time_range.end
        desc = ", ".join(f"{c.name[:40]}({(min(c.time_range.end, k.time_range.start) - max(c.time_range.start, pe)) / 1e3:.2f}ms)" for c in top
                         if min(c.time_range.end, k.time_range.start) - max(c.time_range.start, pe) > 100)
        print(f"gap {g / 1e3:.2f} ms before {k.name[:50]}: {desc[:1500]}")
    pe = max(pe, k.time_range.end)
