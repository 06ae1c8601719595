"""Scratch (GPU box): which library ops keep channels-last memory on logical NCHW / NCDHW views, and where conv_forward has to copy."""
import os, sys, collections
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda")
x = torch.randn(4, 30, 40, 128, device=dev).permute(0, 3, 1, 2).requires_grad_(True)
bn = torch.nn.BatchNorm2d(128).to(dev)
for mode in ("eval", "train"):
    getattr(bn, mode)()
    y = bn(x)
    print("bn2d", mode, "out channels_last:", y.is_contiguous(memory_format=torch.channels_last), y.stride())
    g, = torch.autograd.grad(y.sum() * 1.0 + (y * y).sum(), x)
    print("   grad channels_last:", g.is_contiguous(memory_format=torch.channels_last), g.stride())
x3 = torch.randn(16, 12, 8, 64, device=dev).permute(3, 0, 1, 2).unsqueeze(0).requires_grad_(True)
bn3 = torch.nn.BatchNorm3d(64).to(dev).train()
y3 = bn3(x3)
print("bn3d train out channels_last_3d:", y3.is_contiguous(memory_format=torch.channels_last_3d), y3.stride())
r3 = F.relu(y3)
print("relu out:", r3.is_contiguous(memory_format=torch.channels_last_3d))
g3, = torch.autograd.grad((r3 * r3).sum(), x3)
print("   grad:", g3.is_contiguous(memory_format=torch.channels_last_3d), g3.stride())

# count real copies inside conv_forward during one training step
from nerfdet_amd import conv_train
from nerfdet_amd.presets import build_nerfdet
from nerfdet_amd.synth import batch_to, train_scene
from nerfdet_amd.train import build_optimizer, train_one_step
cnt = collections.Counter()
orig_apply = conv_train.ConvS1.apply
class Probe(conv_train.ConvS1):
    pass
orig_fwd, orig_bwd = conv_train.ConvS1.forward, conv_train.ConvS1.backward
def cf(conv, x):
    three_d = isinstance(conv, (torch.nn.Conv3d, torch.nn.ConvTranspose3d))
    xb = x[0].permute(1, 2, 3, 0) if three_d else x.permute(0, 2, 3, 1)
    cnt[("fwd-input", "3d" if three_d else "2d", xb.is_contiguous())] += 1
    return real_cf(conv, x)
real_cf = conv_train.conv_forward
import nerfdet_amd.neck3d as N3, nerfdet_amd.backbone as BB
N3.conv_forward = cf; BB.conv_forward = cf
def bwd(ctx, g):
    cnt[("bwd-grad", g.dim(), g.is_contiguous())] += 1
    return orig_bwd(ctx, g)
conv_train.ConvS1.backward = staticmethod(bwd)
torch.manual_seed(0)
model = build_nerfdet(50, depth_supervise=True).to(dev).train()
opt = build_optimizer(model)
data = batch_to(train_scene(40, (240, 320), t_views=10, n_boxes=8, seed=0), dev)
train_one_step(model, data, opt)
cnt.clear()
train_one_step(model, data, opt)
for k, v in sorted(cnt.items(), key=str):
    print(k, v)
