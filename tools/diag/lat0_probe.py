"""Diagnostic (GPU box): the FPN lateral 0 (1x1 256->256 over 50 x 60 x 80 with the 2x-upsampled coarser lateral as residual) and the other
HBM-bound 1x1 layers in their REAL epilogue configuration, per tile.  The tuner (tools/tune_conv2d.py) times them without the residual."""
import os, sys, torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerfdet_amd import conv3d as C3
C3.set_arithmetic(sys.argv[1] if len(sys.argv) > 1 else "f16x2")
dev = torch.device("cuda")

def timed(fn, n=9):
    ts = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]

CASES = [  # name, cin, cout, (n,h,w), residual kind (None / "same" / "up2"), bias-only (FPN) or BN
    ("fpn.lat0 +up2 residual", 256, 256, (50, 60, 80), "up2"),
    ("fpn.lat0 no residual", 256, 256, (50, 60, 80), None),
    ("fpn.lat1 +up2 residual", 512, 256, (50, 30, 40), "up2"),
    ("l1.conv3 64->256 +res", 64, 256, (50, 60, 80), "same"),
    ("l2.0.conv1 256->128", 256, 128, (50, 60, 80), None),
    ("l2.conv3 128->512 +res", 128, 512, (50, 30, 40), "same"),
    ("l3.conv3 256->1024 +res", 256, 1024, (50, 15, 20), "same"),
    ("l4.conv3 512->2048 +res", 512, 2048, (50, 8, 10), "same"),
    ("l2.0.ds 256->512", 256, 512, (50, 30, 40), None),
]
for name, cin, cout, nhw, rk in CASES:
    conv = nn.Conv2d(cin, cout, 1, bias=True).to(dev)
    pk = C3.packed([conv], None)
    x = torch.randn(*nhw, cin, device=dev)
    C3.amax_of(x)
    res = None
    if rk == "up2": res = torch.randn(nhw[0], (nhw[1] + 1) // 2, (nhw[2] + 1) // 2, cout, device=dev)
    if rk == "same": res = torch.randn(nhw[0], nhw[1], nhw[2], cout, device=dev)
    byts = 4 * (x.numel() + nhw[0] * nhw[1] * nhw[2] * cout + (0 if res is None else res.numel()))
    ref = None
    for tile in (tuple(int(t) for t in os.environ['PROBE_TILES'].split(',')) if os.environ.get('PROBE_TILES') else (0, 64, 128, 12864, 128256, 129256, 129064, 112864)):
        try:
            f = lambda: C3.conv2d_nhwc(x, pk, residual=res, relu=0, tile=tile, splits=1 if tile else 0, residual_up2=(rk == "up2"))
            y = f()
            if ref is None: ref = y
            err = float((y - ref).abs().max())
            t = timed(f)
        except Exception as e:
            print(f"  {name:28s} tile {tile:6d}  -- {str(e)[:70]}", flush=True); continue
        print(f"  {name:28s} tile {tile:6d} {t*1e3:7.1f} us {byts/t/1e9:6.2f} TB/s  maxdiff vs auto {err:.1e}", flush=True)
