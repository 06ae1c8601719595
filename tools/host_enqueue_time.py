"""GPU box: how long the HOST needs to enqueue one cfg2 scene (every launch of forward_test_async, no waiting), next to the GPU time of the
scene.  If the two are close the step is launch-bound and the host, not a kernel, sets the scenes/s."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
dev = torch.device("cuda")
det = bench.build_model(w).to(dev)
batch = bench.to_device(bench.synth_batch(w, 0), dev)
kw = {k: v for k, v in batch.items() if k not in ("img", "img_metas")}
with torch.no_grad():
    for _ in range(5):
        det(return_loss=False, **batch)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fin = det.forward_test_async(batch["img"], batch["img_metas"], **kw)
        t1 = time.perf_counter()
        fin()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(f"host enqueue per scene: median {host[10]:.2f} ms (p10 {host[2]:.2f}, p90 {host[18]:.2f}); scene wall time from an idle GPU: median {total[10]:.2f} ms; "
      f"host cores {os.cpu_count()}, torch threads {torch.get_num_threads()}")
