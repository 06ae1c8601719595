"""GPU box: per-phase clock stamps of the whole-bottleneck kernel.  Needs a library whose bottleneck_kernels.hip was compiled with -DBT_STAMPS
(hipcc ... -DBT_STAMPS -c bottleneck_kernels.hip, linked with the other objects as the Makefile does) in place of nerf-det_amd/lib/libnerfdet_hip.so.
HISTORY.md R4.3 holds the numbers this produced."""
import os, sys, ctypes, torch
sys.path.insert(0, "/root/repo")
from nerfdet_amd import conv3d as C, _lib
from nerfdet_amd.backbone import Bottleneck
lib = _lib.load()
dev = torch.device("cuda")
torch.manual_seed(0)
blk = Bottleneck(256, 64, 1, None).eval().to(dev)
x = torch.relu(torch.randn(50, 60, 80, 256, device=dev))
nwg = 50 * 15 * 5
buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.ndet_bt_set_stamps.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(3):
        y = blk.forward_nhwc(x)
    torch.cuda.synchronize()
    assert raw.ndet_bt_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    y = blk.forward_nhwc(x)
    torch.cuda.synchronize()
    raw.ndet_bt_set_stamps(ctypes.c_void_p(0))
s = buf.view(nwg, 8).cpu().double()
t0 = s[:, 0].min()
names = ["A conv1", "B bn1+split", "C conv2", "D bn2+split", "E conv3+store", "amax commit"]
d = (s[:, 1:7] - s[:, 0:6]) * 0.01   # 100 MHz -> us
print("workgroups", nwg, " kernel span us", float((s[:, 6].max() - t0) * 0.01))
for i, n in enumerate(names):
    print(f"{n:16s} mean {float(d[:, i].mean()):7.2f} us   median {float(d[:, i].median()):7.2f}   p90 {float(d[:, i].quantile(0.9)):7.2f}")
life = (s[:, 6] - s[:, 0]) * 0.01
print(f"workgroup life   mean {float(life.mean()):7.2f} us   median {float(life.median()):7.2f}")
start = (s[:, 0] - t0) * 0.01
print("start times: first 10 sorted", [round(float(v), 1) for v in start.sort()[0][:10]], " last", round(float(start.max()), 1))
conc = life.sum() / ((s[:, 6].max() - t0) * 0.01)
print("average workgroups in flight", float(conc), "= per CU", float(conc) / 256)
