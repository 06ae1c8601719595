"""Scratch: run the big neck conv a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd.conv3d import conv3d_ndhwc, packed
dev = torch.device("cuda")
conv = nn.Conv3d(256, 256, 3, 1, 1, bias=False).to(dev); bn = nn.BatchNorm3d(256).to(dev).eval()
pk = packed([conv], bn)
x = torch.randn(40, 40, 16, 256, device=dev)
for _ in range(4):
    y = conv3d_ndhwc(x, pk, relu=1, splits=1, tile=128)
torch.cuda.synchronize()
