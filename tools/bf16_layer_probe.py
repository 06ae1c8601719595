"""scratch: one convolution layer, forward / data gradient / weight gradient in bf16 mode against fp32-class mode and against fp64."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import conv3d, conv_train
torch.manual_seed(0)
dev = torch.device("cuda")
def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))
for (cin, cout, grid, k, s) in [(256, 256, (20, 20, 8), 3, 1), (256, 512, (20, 20, 8), 3, 2), (512, 128, (20, 20, 8), 3, 1), (256, 256, (24, 24, 8), 1, 1)]:
    x = torch.randn(*grid, cin, device=dev)
    w = torch.randn(cout, cin, k, k, k, device=dev) / (cin * k ** 3) ** 0.5
    xd, wd = x.double().cpu().permute(3, 0, 1, 2).unsqueeze(0).requires_grad_(True), w.double().cpu().requires_grad_(True)
    yd = F.conv3d(xd, wd, stride=s, padding=k // 2)
    g = torch.randn(*yd.shape[2:], cout, device=dev)
    yd.backward(g.double().cpu().permute(3, 0, 1, 2).unsqueeze(0))
    ref = dict(y=yd[0].permute(1, 2, 3, 0), dx=xd.grad[0].permute(1, 2, 3, 0), dw=wd.grad)
    for mode in ("bf16x3", "bf16"):
        prev = conv3d.set_arithmetic(mode)
        try:
            xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y = conv_train.ConvS1.apply(xx, ww, s)
            y.backward(g)
            print(f"{cin}->{cout} k{k} s{s} {mode:7s}", " ".join(f"{n}: err {rel(t, ref[n])[0]:.2e} cos {rel(t, ref[n])[1]:.6f}" for n, t in (("y", y), ("dx", xx.grad), ("dw", ww.grad))))
        finally:
            conv3d.set_arithmetic(prev)
