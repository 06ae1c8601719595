"""GPU box: the input contract at cfg2 size -- 300 decoded 240x320 uint8 BGR frames resident on the device,
50 source views (+ 1 NeRF target view) assembled into the batch forward_test takes: ms per scene, bytes moved, and the numpy oracle
(the reference's per-sample flow restated, tests' golden-pinned) timed beside it on a smaller sample."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import pipeline as P

dev = torch.device("cuda")
rng = np.random.RandomState(0)
n_frames, h, w, n_images = 300, 240, 320, 50
frames = torch.from_numpy(rng.randint(0, 256, (n_frames, h, w, 3), dtype=np.uint8)).to(dev)
poses = [np.eye(4, dtype=np.float32) + 0.01 * rng.randn(4, 4).astype(np.float32) for _ in range(n_frames)]
info = dict(extrinsics=poses, intrinsics=np.array([[577.0, 0, 160, 0], [0, 577.0, 120, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32),
            annos=dict(axis_align_matrix=np.eye(4, dtype=np.float32)))
cams = P.scene_cameras(info)
for target, tag in ((0, "inference (no NeRF target views)"), (1, "with one NeRF target view (rays, colours)")):
    pipe = P.MultiViewPipeline(n_images, margin=10, loading="sequence", nerf_target_views=target)
    for _ in range(3):
        batch = pipe(frames, cams, (968, 1296, 3))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        batch = pipe(frames, cams, (968, 1296, 3))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 20 * 1e3
    out_bytes = sum(v.numel() * v.element_size() for v in batch.values() if isinstance(v, torch.Tensor))
    in_bytes = n_images * h * w * 3
    print(f"{tag}: {ms:.3f} ms per scene; {in_bytes / 1e6:.1f} MB of frames in, {out_bytes / 1e6:.1f} MB of tensors out "
          f"({(in_bytes + out_bytes) / ms / 1e6:.0f} GB/s incl. host-side camera arithmetic)")
