"""Scratch (GPU box): host time between the D2H sync that ends a scene and the first launch of the next one."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerfdet_amd import backbone as BB
w = bench.WORKLOADS["cfg2"]
dev = torch.device("cuda")
det = bench.build_model(w).to(dev)
batch = bench.to_device(bench.synth_batch(w, 0), dev)
marks = {}
orig_cpu = torch.Tensor.cpu
def cpu(self, *a, **k):
    r = orig_cpu(self, *a, **k)
    if self.is_cuda:
        marks["sync"] = time.perf_counter()
    return r
torch.Tensor.cpu = cpu
orig_stem = BB.stem_conv_bn_relu_maxpool
def stem(*a, **k):
    marks["stem"] = time.perf_counter()
    return orig_stem(*a, **k)
BB.stem_conv_bn_relu_maxpool = stem
rows = []
with torch.no_grad():
    for i in range(30):
        t_call = time.perf_counter()
        det(return_loss=False, **batch)
        t_ret = time.perf_counter()
        if i >= 10:
            rows.append((marks["stem"] - t_call, t_ret - marks["sync"]))
        prev_ret = t_ret
a = sum(r[0] for r in rows) / len(rows) * 1e6
b = sum(r[1] for r in rows) / len(rows) * 1e6
print(f"call entry -> stem launch: {a:.0f} us; sync return -> forward_test return: {b:.0f} us")
