"""scratch: one conv3d_ndhwc shape a few times (for rocprofv3 --pmc passes).  argv: cin cout X Y Z k tile splits [arithmetic]"""
import os, sys, torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import conv3d as C3
cin, cout, X, Y, Z, k, tile, splits = (int(v) for v in sys.argv[1:9])
C3.set_arithmetic(sys.argv[9] if len(sys.argv) > 9 else "bf16x3")
dev = torch.device("cuda")
conv = nn.Conv3d(cin, cout, k, 1, k // 2, bias=False).to(dev); bn = nn.BatchNorm3d(cout).to(dev).eval()
pk = C3.packed([conv], bn)
x = torch.randn(X, Y, Z, cin, device=dev)
for _ in range(4):
    y = C3.conv3d_ndhwc(x, pk, relu=1, splits=splits, tile=tile)
torch.cuda.synchronize()
