"""Scratch (GPU box): cProfile of the host side of forward_test at cfg2 (what delays the first launch of a step)."""
import cProfile, io, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.WORKLOADS["cfg2"]
dev = torch.device("cuda")
det = bench.build_model(w).to(dev)
batch = bench.to_device(bench.synth_batch(w, 0), dev)
with torch.no_grad():
    for _ in range(5):
        det(return_loss=False, **batch)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        det(return_loss=False, **batch)
    torch.cuda.synchronize()
    pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(38)
    print(s.getvalue()[:7000])
