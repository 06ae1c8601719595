"""scratch (GPU box): backbone + FPN of the cfg2 views on one stream vs. the views dealt to two / three streams (per-view independent work)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

w = bench.WORKLOADS["cfg2"]
dev = torch.device("cuda")
det = bench.build_model(w).to(dev)
batch = bench.to_device(bench.synth_batch(w, 0), dev)
img = batch["img"][0]          # (50, 3, 240, 320)


def one(x):
    return det.neck(det.backbone(x))[0]


def split(n):
    cur = torch.cuda.current_stream(dev)
    streams = STREAMS[:n]
    ev = torch.cuda.Event(); ev.record(cur)
    outs = []
    parts = torch.chunk(img, n, dim=0)
    for s, p in zip(streams, parts):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            outs.append(one(p))
    for s in streams:
        cur.wait_stream(s)
    return outs


STREAMS = [torch.cuda.Stream(dev) for _ in range(4)]
with torch.no_grad():
    for mode in ("one", 2, 3, "one", 2, 3):
        f = (lambda: one(img)) if mode == "one" else (lambda: split(mode))
        for _ in range(4):
            f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        print(f"backbone+FPN, {mode} stream(s): median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}", flush=True)
