"""GPU box: f-4, the render_testing pass of forward_test at cfg2 size (render_ray.py:452-517): every ray of T target views rendered in chunks of
N_rand, with PSNR / SSIM / depth-error map (simple_test(evaluate_nerf=True)) -- ms per scene on top of the detection pass."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerfdet_amd.synth import batch_to, train_scene
dev = torch.device("cuda")
w = bench.WORKLOADS["cfg2"]
det = bench.build_model(w).to(dev)
for t_views in (1, 4):
    data = batch_to(train_scene(50, (240, 320), t_views=t_views, n_boxes=4, seed=1), dev)
    rb = det._ray_batch(data)
    res = {}
    for flag in (False, True):
        det.render_testing = flag
        with torch.no_grad():
            for _ in range(3):
                det.simple_test(data["img"], data["img_metas"], ray_batch=rb, evaluate_nerf=flag)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(10):
                det.simple_test(data["img"], data["img_metas"], ray_batch=rb, evaluate_nerf=flag)
            torch.cuda.synchronize()
        res[flag] = (time.perf_counter() - t) / 10 * 1e3
    rays = t_views * 220 * 300
    print(f"{t_views} target view(s), {rays} rays x {det.N_samples} samples: detection {res[False]:.2f} ms, with rendering + metrics {res[True]:.2f} ms "
          f"-> {(res[True] - res[False]):.2f} ms = {rays / (res[True] - res[False]) / 1e3:.2f} M rays/s; psnr {float(det.render_metrics[0]):.2f} ssim {float(det.render_metrics[1]):.4f}")
