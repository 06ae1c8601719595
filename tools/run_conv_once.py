"""Scratch: run one big conv a few times (for rocprofv3 --pmc passes).  argv: [arith] [shape: neck|fpn] [tile] [splits]"""
import os, sys, torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import conv3d as C3
arith = sys.argv[1] if len(sys.argv) > 1 else "f32"
shape = sys.argv[2] if len(sys.argv) > 2 else "neck"
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 128
splits = int(sys.argv[4]) if len(sys.argv) > 4 else 1
C3.set_arithmetic(arith)
dev = torch.device("cuda")
conv = nn.Conv3d(256, 256, 3, 1, 1, bias=False).to(dev); bn = nn.BatchNorm3d(256).to(dev).eval()
pk = C3.packed([conv], bn)
x = torch.randn(*((40, 40, 16) if shape == "neck" else (60, 80, 50)), 256, device=dev)
for _ in range(4):
    y = C3.conv3d_ndhwc(x, pk, relu=1, splits=splits, tile=tile)
torch.cuda.synchronize()
