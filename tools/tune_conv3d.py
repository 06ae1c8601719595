"""Scratch tuner (GPU box): times the MFMA conv3d on every layer shape of the cfg2 neck/head across tile / split-K choices."""
import os, sys, itertools
import torch
from torch import nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfdet_amd.conv3d import conv3d_ndhwc, packed

LAYERS = [  # name, cin, cout, grid(in), k, stride, transposed
    ("fpn-like 256->256 @(50*60)x80x1 3x3x1?", 256, 256, (60, 80, 50), 3, 1, False),
    ("down0.conv 256->256 @40x40x16", 256, 256, (40, 40, 16), 3, 1, False),
    ("out0 256->128 @40x40x16", 256, 128, (40, 40, 16), 3, 1, False),
    ("down1.conv1 256->512 s2", 256, 512, (40, 40, 16), 3, 2, False),
    ("down1.conv2 512->512 @20x20x8", 512, 512, (20, 20, 8), 3, 1, False),
    ("down1.ds 1x1 s2 256->512", 256, 512, (40, 40, 16), 1, 2, False),
    ("out1 512->128 @20x20x8", 512, 128, (20, 20, 8), 3, 1, False),
    ("down2.conv1 512->1024 s2", 512, 1024, (20, 20, 8), 3, 2, False),
    ("down2.conv2 1024->1024 @10x10x4", 1024, 1024, (10, 10, 4), 3, 1, False),
    ("out2 1024->128 @10x10x4", 1024, 128, (10, 10, 4), 3, 1, False),
    ("up2.convT 1024->512", 1024, 512, (10, 10, 4), 2, 2, True),
    ("up1.convT 512->256", 512, 256, (20, 20, 8), 2, 2, True),
    ("head 128->25 @40x40x16", 128, 25, (40, 40, 16), 3, 1, False),
]

def layers_for(gx, gy, gz):
    """FastIndoorImVoxelNeck + head layer shapes on a gx x gy x gz voxel grid."""
    g0, g1, g2 = (gx, gy, gz), (gx // 2, gy // 2, gz // 2), (gx // 4, gy // 4, gz // 4)
    return [("down0.conv 256->256", 256, 256, g0, 3, 1, False), ("out0 256->128", 256, 128, g0, 3, 1, False), ("down1.conv1 256->512 s2", 256, 512, g0, 3, 2, False),
            ("down1.conv2 512->512", 512, 512, g1, 3, 1, False), ("down1.ds 1x1 s2 256->512", 256, 512, g0, 1, 2, False), ("out1 512->128", 512, 128, g1, 3, 1, False),
            ("down2.conv1 512->1024 s2", 512, 1024, g1, 3, 2, False), ("down2.conv2 1024->1024", 1024, 1024, g2, 3, 1, False), ("out2 1024->128", 1024, 128, g2, 3, 1, False),
            ("up2.convT 1024->512", 1024, 512, g2, 2, 2, True), ("up1.convT 512->256", 512, 256, g1, 2, 2, True), ("head0 128->25", 128, 25, g0, 3, 1, False),
            ("head1 128->25", 128, 25, g1, 3, 1, False), ("head2 128->25", 128, 25, g2, 3, 1, False)]


def main():
    from nerfdet_amd import conv3d as C3
    global LAYERS
    if len(sys.argv) > 1:
        C3.set_arithmetic(sys.argv[1])
    if len(sys.argv) > 4:          # tune_conv3d.py <arithmetic> <X> <Y> <Z>
        LAYERS = layers_for(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    tiles = (64, 128, 12864, 128256, 129256, 129257, 129064, 3128, 3256, 3257, 3258, 100064, 100128, 112864) if C3.ARITHMETIC in ("bf16x3", "bf16", "f16x2") else (64, 128)
    if os.environ.get("TUNE_TILES"):
        tiles = tuple(int(t) for t in os.environ["TUNE_TILES"].split(","))
    if os.environ.get("TUNE_LAYERS"):
        LAYERS = [l for l in LAYERS if any(k in l[0] for k in os.environ["TUNE_LAYERS"].split(","))]
    print("arithmetic", C3.ARITHMETIC, flush=True)
    dev = torch.device("cuda")
    for name, cin, cout, grid, k, s, tr in LAYERS:
        conv = (nn.ConvTranspose3d(cin, cout, 2, 2, bias=False) if tr else nn.Conv3d(cin, cout, k, s, k // 2, bias=False)).to(dev)
        bn = nn.BatchNorm3d(cout).to(dev).eval()
        pk = packed([conv], bn)
        x = torch.randn(*grid, cin, device=dev)
        od = [2 * g for g in grid] if tr else [(g + 2 * (k // 2) - k) // s + 1 for g in grid]
        flops = 2 * od[0] * od[1] * od[2] * cout * cin * (1 if tr else k ** 3)
        best = None
        for tile, splits in itertools.product(tiles, (1,) if tr else (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)):
            if splits > k ** 3 * (cin // 32):
                continue
            try:
                for _ in range(2):
                    conv3d_ndhwc(x, pk, relu=1, splits=splits, tile=tile)
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); conv3d_ndhwc(x, pk, relu=1, splits=splits, tile=tile); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                t = sorted(ts)[2]
            except Exception as e:
                print(name, tile, splits, "failed", repr(e)[:100]); continue
            if best is None or t < best[0]:
                best = (t, tile, splits)
            print(f"  {name:36s} tile={tile:3d} splits={splits} {t*1e3:8.1f} us {flops/t/1e9:7.1f} TF", flush=True)
        auto = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); conv3d_ndhwc(x, pk, relu=1); e1.record(); torch.cuda.synchronize(); auto.append(e0.elapsed_time(e1))
        m_ = (grid[0]*grid[1]*grid[2]) if tr else od[0]*od[1]*od[2]
        print("TUNED_JSON", __import__("json").dumps(dict(key=[m_, cout, (1 if tr else k**3) * (cin // 32), int(tr)], tile=best[1], splits=best[2], us=best[0]*1e3, name=name)), flush=True)
        print(f"BEST {name:36s} tile={best[1]} splits={best[2]} {best[0]*1e3:8.1f} us {flops/best[0]/1e9:7.1f} TF | auto {sorted(auto)[2]*1e3:8.1f} us  ({flops/1e9:.1f} GF)", flush=True)

if __name__ == "__main__":
    main()
