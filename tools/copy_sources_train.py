"""Scratch (GPU box): who issues the elementwise copies of a training step (aten::copy_ grouped by its enclosing ops)."""
import os, sys, collections
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_one_step(model, data, opt)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ("aten::copy_", "aten::mul", "aten::add", "aten::add_", "aten::fill_", "aten::clamp_min_", "aten::threshold_backward", "aten::sum", "aten::cat"):
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 3:
            chain.append(p.name[:38]); p = p.cpu_parent
        key = (e.name, " < ".join(chain), str(e.input_shapes)[:60])
        agg[key][0] += 1
        agg[key][1] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
for (name, chain, shp), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / 1e3:7.3f} ms {n:4d} x {name:24s} {shp:60s} <- {chain}")
