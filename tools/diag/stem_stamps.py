"""GPU box: where a tile of the stem kernel spends its time.  Needs a diagnostic build of stem_kernels.hip that accumulates wall_clock64() differences
per workgroup and exports ndet_st_set_stamps (the stamps were a throw-away patch of round 4: split + store 0.6 us, multiply + stage 5.5 us, pool +
store 2.1 us per tile in the bf16x3 form; HISTORY.md R4).  Kept as the reader of that buffer."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerfdet_amd import conv3d as C, _lib
from torch import nn
dev = torch.device("cuda")
conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev); bn = nn.BatchNorm2d(64).eval().to(dev)
x = torch.randn(50, 3, 240, 320, device=dev)
raw = ctypes.CDLL(_lib.LIB_PATH); raw.ndet_st_set_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(512 * 4, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3): C.stem_conv_bn_relu_maxpool(x, conv, bn)
    torch.cuda.synchronize(); raw.ndet_st_set_stamps(ctypes.c_void_p(buf.data_ptr()))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); C.stem_conv_bn_relu_maxpool(x, conv, bn); b.record(); torch.cuda.synchronize()
s = buf.view(512, 4).cpu().double()
n = s[:, 3].sum()
print(f"launch {a.elapsed_time(b) * 1e3:.1f} us; tiles {int(n)}; per tile (us): split+store -> barrier A {float(s[:,0].sum()/n)*0.01:.2f}, multiply + stage -> barrier B {float(s[:,1].sum()/n)*0.01:.2f}, pool + store {float(s[:,2].sum()/n)*0.01:.2f}")
