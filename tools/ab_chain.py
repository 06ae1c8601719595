"""GPU box: the same cfg2 forward_test with and without the chained bottleneck tail (conv3d.CHAIN_BOTTLENECKS), alternating, same process."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerfdet_amd import conv3d

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
ATTR = sys.argv[2] if len(sys.argv) > 2 else "CHAIN_BOTTLENECKS"     # ab_chain.py cfg2 DIRECT_EPILOGUE: any boolean switch of nerfdet_amd.conv3d
dev = torch.device("cuda")
det = bench.build_model(w).to(dev)
batch = bench.to_device(bench.synth_batch(w, 0), dev)


def run(n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        det(return_loss=False, **batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    for flag in (True, False):
        setattr(conv3d, ATTR, flag)
        run(5)
    for rep in range(4):
        for flag in (True, False):
            setattr(conv3d, ATTR, flag)
            print(f"rep {rep} {ATTR}={flag}: {run(20):.3f} ms/step", flush=True)
