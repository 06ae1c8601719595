"""Host-side cost of a training step: cProfile over three cfg3 steps (tools/bench_train.py's set-up), functions by own time and by cumulative time.
The autograd engine runs the backward of custom Functions on its own thread: their Python frames are profiled there separately
(threading.setprofile) and merged.

    python tools/diag/train_cprofile.py > gpurun_out/train_cprofile.txt
"""
import cProfile
import io
import os
import pstats
import sys
import threading

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from nerfdet_amd.presets import build_nerfdet
    from nerfdet_amd.synth import batch_to, train_scene
    from nerfdet_amd.train import build_optimizer, train_one_step
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    det = build_nerfdet(50, depth_supervise=True)
    with torch.no_grad():
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
    det.to(dev).train()
    data = batch_to(train_scene(40, (240, 320), t_views=10, n_boxes=8, seed=0), dev)
    opt = build_optimizer(det)
    for _ in range(3):
        train_one_step(det, data, opt)
    torch.cuda.synchronize()
    profs = {}

    def thread_prof(frame, event, arg):      # first Python frame on another thread (the autograd engine's): give it its own profiler
        p = cProfile.Profile()
        profs[threading.get_ident()] = p
        p.enable()
    threading.setprofile(thread_prof)
    main_prof = cProfile.Profile()
    main_prof.enable()
    steps = 3
    import time
    t0 = time.perf_counter()
    prev = None
    for _ in range(steps):
        cur = train_one_step(det, data, opt, lazy_log=True)
        if prev is not None:
            prev["log"].get()
        prev = cur
    t1 = time.perf_counter()
    main_prof.disable()
    threading.setprofile(None)
    for p in profs.values():
        p.disable()
    torch.cuda.synchronize()
    print(f"# {steps} steps queued in {1e3 * (t1 - t0):.1f} ms of host time ({1e3 * (t1 - t0) / steps:.1f} ms per step, profiler overhead included)")
    stats = pstats.Stats(main_prof)
    for p in profs.values():
        stats.add(p)
    for key in ("tottime", "cumtime"):
        buf = io.StringIO()
        stats.stream = buf
        stats.sort_stats(key).print_stats(45)
        print(f"# by {key}")
        print("\n".join(l[:200] for l in buf.getvalue().splitlines() if l.strip())[:12000])


if __name__ == "__main__":
    main()
