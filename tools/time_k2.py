"""Scratch (GPU box): time K2 (density features) at the cfg2 shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import ops
from oracle import nerfdet_oracle as O
dev = torch.device("cuda")
meta = O.ring_scene_meta(50, (240, 320))
proj, rgbp = ops.compute_projection(meta, 4, dev), ops.compute_projection(meta, 1, dev)
pts = ops.get_points((40, 40, 16), (0.16, 0.16, 0.2), meta["lidar2img"]["origin"], dev)
mapped = torch.randn(50, 60, 80, 32, device=dev).permute(0, 3, 1, 2)
bias = torch.randn(32, device=dev); rgb = torch.rand(50, 3, 240, 320, device=dev)
ts = []
for i in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = ops.density_features(mapped, bias, rgb, pts, proj, rgbp); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("K2 us", sorted(ts)[len(ts) // 2], "checksum", float(out.double().sum()))
