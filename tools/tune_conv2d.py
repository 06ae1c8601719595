"""Scratch tuner (GPU box): the ResNet-50 / FPN layer shapes at cfg2 (50 x 240 x 320) through conv2d_nhwc, per tile / split."""
import os, sys, itertools
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd.conv3d import conv2d_nhwc, packed

LAYERS = [  # name, cin, cout, (n,h,w), k, stride, residual
    ("l1.conv1 1x1 256->64", 256, 64, (50, 60, 80), 1, 1, False),
    ("l1.conv2 3x3 64->64", 64, 64, (50, 60, 80), 3, 1, False),
    ("l1.conv3 1x1 64->256 +res", 64, 256, (50, 60, 80), 1, 1, True),
    ("l2.conv1 1x1 512->128", 512, 128, (50, 30, 40), 1, 1, False),
    ("l2.conv2 3x3 128->128", 128, 128, (50, 30, 40), 3, 1, False),
    ("l2.conv3 1x1 128->512 +res", 128, 512, (50, 30, 40), 1, 1, True),
    ("l3.conv1 1x1 1024->256", 1024, 256, (50, 15, 20), 1, 1, False),
    ("l3.conv2 3x3 256->256", 256, 256, (50, 15, 20), 3, 1, False),
    ("l3.conv3 1x1 256->1024 +res", 256, 1024, (50, 15, 20), 1, 1, True),
    ("l4.conv1 1x1 2048->512", 2048, 512, (50, 8, 10), 1, 1, False),
    ("l4.conv2 3x3 512->512", 512, 512, (50, 8, 10), 3, 1, False),
    ("l4.conv3 1x1 512->2048 +res", 512, 2048, (50, 8, 10), 1, 1, True),
    ("fpn.lat0 1x1 256->256", 256, 256, (50, 60, 80), 1, 1, False),
    ("fpn.out0 3x3 256->256", 256, 256, (50, 60, 80), 3, 1, False),
]

def layers_for(n, h, w):
    """ResNet bottleneck + FPN layer shapes for n views of h x w pixels (stage i runs at h/4/2^i x w/4/2^i)."""
    out = []
    for i, (cin, mid) in enumerate(((256, 64), (512, 128), (1024, 256), (2048, 512))):
        hh, ww = -(-h // (4 << i)), -(-w // (4 << i))
        out += [(f"l{i+1}.conv1 1x1 {cin}->{mid}", cin, mid, (n, hh, ww), 1, 1, False), (f"l{i+1}.conv2 3x3 {mid}->{mid}", mid, mid, (n, hh, ww), 3, 1, False),
                (f"l{i+1}.conv3 1x1 {mid}->{cin} +res", mid, cin, (n, hh, ww), 1, 1, True)]
    out += [("fpn.lat0 1x1 256->256", 256, 256, (n, h // 4, w // 4), 1, 1, False), ("fpn.out0 3x3 256->256", 256, 256, (n, h // 4, w // 4), 3, 1, False)]
    return out


def first_block_layers(n, h, w):
    """The shapes only the first block of a stage (and the FPN laterals that share their keys) has: conv1 at the previous stage's
    resolution, the strided 1x1 downsample."""
    out = []
    for i, (cin, mid) in enumerate(((64, 64), (256, 128), (512, 256), (1024, 512))):
        hi, wi = (-(-h // 4), -(-w // 4)) if i == 0 else (-(-h // (4 << (i - 1))), -(-w // (4 << (i - 1))))
        out.append((f"l{i+1}.0.conv1 1x1 {cin}->{mid}", cin, mid, (n, hi, wi), 1, 1, False))
        out.append((f"l{i+1}.0.ds 1x1 s{1 if i == 0 else 2} {cin}->{4 * mid}", cin, 4 * mid, (n, hi, wi), 1, 1 if i == 0 else 2, False))
    out.append(("fpn.lat3 1x1 2048->256", 2048, 256, (n, -(-h // 32), -(-w // 32)), 1, 1, False))
    return out


def main():
    from nerfdet_amd import conv3d as C3
    global LAYERS
    if len(sys.argv) > 1:
        C3.set_arithmetic(sys.argv[1])
    if len(sys.argv) > 4:          # tune_conv2d.py <arithmetic> <n_views> <H> <W> [first]
        LAYERS = (first_block_layers if len(sys.argv) > 5 else layers_for)(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    tiles = (64, 128, 12864, 128256, 129256, 129257, 129064, 3128, 3256, 3257, 3258, 100064, 100128, 112864) if C3.ARITHMETIC in ("bf16x3", "bf16", "f16x2") else (64, 128)
    if os.environ.get("TUNE_TILES"):
        tiles = tuple(int(t) for t in os.environ["TUNE_TILES"].split(","))
    if os.environ.get("TUNE_LAYERS"):
        LAYERS = [l for l in LAYERS if any(k in l[0] for k in os.environ["TUNE_LAYERS"].split(","))]
    print("arithmetic", C3.ARITHMETIC, flush=True)
    dev = torch.device("cuda")
    tot_best = tot_auto = 0.0
    for name, cin, cout, nhw, k, s, use_res in LAYERS:
        conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(dev)
        bn = nn.BatchNorm2d(cout).to(dev).eval()
        pk = packed([conv], bn)
        x = torch.randn(*nhw, cin, device=dev)
        oh, ow = (nhw[1] + 2 * (k // 2) - k) // s + 1, (nhw[2] + 2 * (k // 2) - k) // s + 1
        res = torch.randn(nhw[0], oh, ow, cout, device=dev) if use_res else None
        flops = 2 * nhw[0] * oh * ow * cout * cin * k * k
        byts = 4 * (x.numel() + nhw[0] * oh * ow * cout * (2 if use_res else 1) + cin * cout * k * k)
        best = None
        def run(**kw):
            ts = []
            for i in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); conv2d_nhwc(x, pk, residual=res, relu=1, **kw); e1.record(); torch.cuda.synchronize()
                if i >= 2: ts.append(e0.elapsed_time(e1))
            return sorted(ts)[2]
        for tile, splits in itertools.product(tiles, (1, 2, 3, 4, 8, 16)):
            if splits > k * k * (cin // 32): continue
            try:
                t = run(tile=tile, splits=splits)
            except Exception:
                continue
            if os.environ.get("TUNE_VERBOSE"): print(f"    {name:30s} tile {tile:6d} s={splits:2d} {t*1e3:8.1f} us", flush=True)
            if best is None or t < best[0]: best = (t, tile, splits)
        print("TUNED_JSON", __import__("json").dumps(dict(key=[nhw[0]*oh*ow, cout, k*k*(cin//32), 0], tile=best[1], splits=best[2], us=best[0]*1e3, name=name)), flush=True)
        ta = run()
        tot_best += best[0]; tot_auto += ta
        print(f"{name:30s} best tile={best[1]:3d} s={best[2]} {best[0]*1e3:7.1f} us {flops/best[0]/1e9:6.1f} TF {byts/best[0]/1e9:6.2f} TB/s | auto {ta*1e3:7.1f} us", flush=True)
    print("sum best", tot_best, "ms  sum auto", tot_auto, "ms")

if __name__ == "__main__":
    main()
