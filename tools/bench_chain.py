"""GPU box: conv2 -> conv3 of the stage-1 / stage-2 bottlenecks, chained kernel vs the two launches (cfg2: 50 views 240x320)."""
import os, sys
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import conv3d as C

dev = torch.device("cuda")


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    if len(sys.argv) > 1:
        C.set_arithmetic(sys.argv[1])
    views = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    for name, cin, mid, cout, hw, stride in (("l1 64->64->256", 64, 64, 256, (60, 80), 1), ("l2 128->128->512", 128, 128, 512, (30, 40), 1),
                                             ("l2.0 128->128->512 s2", 128, 128, 512, (60, 80), 2)):
        torch.manual_seed(0)
        c2, c3 = nn.Conv2d(cin, mid, 3, stride, 1, bias=False).to(dev), nn.Conv2d(mid, cout, 1, bias=False).to(dev)
        b2, b3 = nn.BatchNorm2d(mid).to(dev).eval(), nn.BatchNorm2d(cout).to(dev).eval()
        x = torch.randn(views, *hw, cin, device=dev)
        pk2, pk3 = C.packed([c2], b2), C.packed([c3], b3)
        oh, ow = (hw[0] - 1) // stride + 1, (hw[1] - 1) // stride + 1
        res = torch.randn(views, oh, ow, cout, device=dev)
        with torch.no_grad():
            t2 = timeit(lambda: C.conv2d_nhwc(x, pk2, relu=1))
            y = C.conv2d_nhwc(x, pk2, relu=1)
            t3 = timeit(lambda: C.conv2d_nhwc(y, pk3, residual=res, relu=1))
            tc = timeit(lambda: C.conv2d_chain_nhwc(x, pk2, pk3, residual=res, relu=1))
            d = float((C.conv2d_chain_nhwc(x, pk2, pk3, residual=res, relu=1) - C.conv2d_nhwc(y, pk3, residual=res, relu=1)).abs().max())
        m = views * oh * ow
        gf = 2 * m * mid * (cin * 9 + cout) / 1e9
        mb = 4 * (x.numel() + res.numel() * 2) / 1e6
        print(f"{name:26s} conv2 {t2:7.1f} us + conv3 {t3:7.1f} us = {t2 + t3:7.1f} | chained {tc:7.1f} us  {gf / tc * 1e3:6.1f} TF  {mb / tc:5.2f} TB/s  maxdiff {d:.2e}", flush=True)


main()
