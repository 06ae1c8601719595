"""Scratch (GPU box): torch.profiler key_averages over two training steps: which aten / autograd ops own the GPU time outside the convolutions."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        train_one_step(model, data, opt)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=False).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=50, max_shapes_column_width=90))
