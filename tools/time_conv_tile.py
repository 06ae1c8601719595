"""Scratch (GPU box): time the 849-GFLOP 3x3x3 256->256 layer on one tile of the bf16x3 kernel through the C ABI.
   python tools/time_conv_tile.py [tile]   # 64 128 12864 128256 3128 3256"""
import os, sys, ctypes, torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import conv3d as C3, _lib
from ctypes import c_void_p
C3.set_arithmetic("bf16x3")
dev = torch.device("cuda")
conv = nn.Conv3d(256, 256, 3, 1, 1, bias=False).to(dev); bn = nn.BatchNorm3d(256).to(dev).eval()
pk = C3.packed([conv], bn)
x = torch.randn(60, 80, 50, 256, device=dev)
out = torch.empty(60, 80, 50, 256, device=dev)
planes = C3.split_planes(pk)
ws = torch.zeros(128, dtype=torch.int64, device=dev)
lib = _lib.load()
i3 = lambda *v: (ctypes.c_int * 3)(*v)
P = lambda t: c_void_p(t.data_ptr())
st = c_void_p(torch.cuda.current_stream().cuda_stream)
TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ts = []
for i in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.ndet_conv_ndhwc_split(P(x), P(planes), P(out), 60, 80, 50, 256, 256, i3(3,3,3), i3(1,1,1), i3(1,1,1), 0, P(pk["scale"]), P(pk["shift"]), None, 0, 1, 1, TILE, P(ws), st)
    e1.record(); torch.cuda.synchronize(); assert rc == 0
    ts.append(e0.elapsed_time(e1))
print("TILE", TILE, "ms", sorted(ts)[2], "TF", 2*240000*256*6912/sorted(ts)[2]/1e9, flush=True)

