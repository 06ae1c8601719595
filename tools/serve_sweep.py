"""Scratch (GPU box): throughput with 1..4 scenes in flight (bench.serve_in_flight)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = bench.WORKLOADS[wl]
dev = torch.device("cuda")
det = bench.build_model(w).to(dev)
batch = bench.to_device(bench.synth_batch(w, 0), dev)
with torch.no_grad():
    for _ in range(5):
        det(return_loss=False, **batch)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        det(return_loss=False, **batch)
    torch.cuda.synchronize()
    print(f"sequential: {20 / (time.perf_counter() - t):.2f} scenes/s")
for n in (1, 2, 3, 4):
    print(f"{n} in flight: {bench.serve_in_flight(det, batch, 24, n):.2f} scenes/s", flush=True)
