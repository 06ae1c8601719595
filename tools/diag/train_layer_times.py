"""Per-shape table of the training step's split-family convolution launches (forward, data gradient and the staged weight-gradient GEMMs all go
through conv3d._conv_split): GEMM rows, output channels, K steps, the tile conv3d.choose_tiling_split picked, launches per step, event-timed
average.  Tells which training shapes sit on a poor tile (the tuned tables were swept on the inference shapes).

    python tools/diag/train_layer_times.py > gpurun_out/train_layer_times.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from nerfdet_amd import trace
    from nerfdet_amd.presets import build_nerfdet
    from nerfdet_amd.synth import batch_to, train_scene
    from nerfdet_amd.train import build_optimizer, train_one_step
    import nerfdet_amd.conv3d as C3
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
    calls = []
    orig = C3.choose_tiling_split

    def logged(m, cout, k_iters, tile=0, splits=0, transposed=False, halo_ok=False):
        r = orig(m, cout, k_iters, tile, splits, transposed, halo_ok)
        calls.append((m, cout, k_iters, int(halo_ok), r))
        return r
    C3.choose_tiling_split = logged
    rec = trace.Recorder(sample=lambda name: name.startswith("k_conv"))
    trace.recorder = rec
    steps = 3
    for _ in range(steps):
        train_one_step(det, data, opt)
    torch.cuda.synchronize()
    trace.recorder = None
    C3.choose_tiling_split = orig
    spans = [(n, e0.elapsed_time(e1), info) for n, e0, e1, info in rec.spans]
    assert len(spans) == len(calls), (len(spans), len(calls))
    table = {}
    for (m, cout, k, halo, r), (name, ms, info) in zip(calls, spans):
        t = table.setdefault((m, cout, k, halo, name), [0, 0.0, info["flops"]])
        t[0] += 1
        t[1] += ms
    print(f"{'rows':>8} {'cout':>5} {'ksteps':>6} halo {'kernel':<38} {'n/step':>6} {'avg us':>8} {'GF':>7} {'TF/s':>6} {'ms/step':>8}")
    tot = 0.0
    for (m, cout, k, halo, name), (n, ms, fl) in sorted(table.items(), key=lambda kv: -kv[1][1]):
        tot += ms / steps
        print(f"{m:8d} {cout:5d} {k:6d} {halo:4d} {name:<38} {n / steps:6.1f} {ms / n * 1e3:8.1f} {fl / 1e9:7.2f} {fl / (ms / n) / 1e9:6.1f} {ms / steps:8.3f}")
    print(f"total {tot:.2f} ms/step over {len(spans) // steps} launches")


if __name__ == "__main__":
    main()
