"""Scratch (GPU box): ResNet+FPN on 50 views in one stream vs two halves of 25 views on two streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.WORKLOADS["cfg2"]
dev = torch.device("cuda")
det = bench.build_model(w).to(dev)
batch = bench.to_device(bench.synth_batch(w, 0), dev)
img = batch["img"].reshape(-1, 3, 240, 320)
s1 = torch.cuda.Stream()


def one():
    return det.neck(det.backbone(img))[0]


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    a = det.neck(det.backbone(img[:25]))[0]
    with torch.cuda.stream(s1):
        b = det.neck(det.backbone(img[25:]))[0]
    cur.wait_stream(s1)
    return a, b


def interleaved(n_parts=2):
    """Launch the halves alternately layer by layer is not possible from outside; instead issue half B first on the side stream, then A."""
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        b = det.neck(det.backbone(img[25:]))[0]
    a = det.neck(det.backbone(img[:25]))[0]
    cur.wait_stream(s1)
    return a, b


def timeit(f, n=20):
    with torch.no_grad():
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print(f"one stream, 50 views: {timeit(one):.3f} ms")
print(f"two streams, 25 + 25: {timeit(two):.3f} ms")
print(f"two streams, side first: {timeit(interleaved):.3f} ms")
print(f"one stream, 50 views: {timeit(one):.3f} ms")
