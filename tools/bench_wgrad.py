"""GPU box: weight gradient of the training-time layer shapes, staged form (tap copies + GEMM) vs implicit (k_wgrad_split)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import conv_train, conv3d as C

dev = torch.device("cuda")


def timeit(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


SHAPES = [("neck 3x3x3 256->256 @40x40x16", (40, 40, 16), 256, 256, (3, 3, 3), 1), ("neck 3x3x3 256->128", (40, 40, 16), 256, 128, (3, 3, 3), 1),
          ("neck s2 256->512", (40, 40, 16), 256, 512, (3, 3, 3), 2), ("neck 512->512 @20x20x8", (20, 20, 8), 512, 512, (3, 3, 3), 1),
          ("neck 1024->1024 @10x10x4", (10, 10, 4), 1024, 1024, (3, 3, 3), 1), ("head 128->25", (40, 40, 16), 128, 25, (3, 3, 3), 1),
          ("fpn out 3x3 256->256 @40x60x80", (40, 60, 80), 256, 256, (3, 3), 1), ("l2 3x3 128->128 @40x30x40", (40, 30, 40), 128, 128, (3, 3), 1),
          ("l2 1x1 512->128", (40, 30, 40), 512, 128, (1, 1), 1), ("l2 1x1 128->512", (40, 30, 40), 128, 512, (1, 1), 1),
          ("l3 1x1 1024->256", (40, 15, 20), 1024, 256, (1, 1), 1), ("l3 3x3 256->256", (40, 15, 20), 256, 256, (3, 3), 1),
          ("l4 1x1 2048->512", (40, 8, 10), 2048, 512, (1, 1), 1), ("l4 3x3 512->512", (40, 8, 10), 512, 512, (3, 3), 1)]
for name, dims, cin, cout, kernel, stride in SHAPES:
    two_d = len(kernel) == 2
    k3 = ((1,) + kernel) if two_d else kernel
    s3 = (1, stride, stride) if two_d else (stride,) * 3
    od = tuple((n + 2 * (k // 2) - k) // s + 1 for n, k, s in zip(dims, k3, s3))
    x = torch.randn(*dims, cin, device=dev)
    g = torch.randn(*od, cout, device=dev)
    ts = timeit(lambda: conv_train.weight_grad(x, g, kernel, stride, implicit=False))
    ti = timeit(lambda: conv_train.weight_grad(x, g, kernel, stride, implicit=True))
    gf = 2 * od[0] * od[1] * od[2] * cin * cout * k3[0] * k3[1] * k3[2] / 1e9
    print(f"{name:36s} {gf:7.1f} GF  staged {ts:8.1f} us ({gf / ts * 1e3:6.1f} TF)   implicit {ti:8.1f} us ({gf / ti * 1e3:6.1f} TF)", flush=True)
