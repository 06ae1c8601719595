"""Where the training step's time goes outside the hand-written kernels: three cfg3 steps under torch.profiler, device time per ATen
operator with the python call site that issued it (tools/bench_train.py's set-up).

    python tools/diag/train_profile.py [--arith bf16x3] > gpurun_out/train_profile.txt
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arith", default=None)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    from nerfdet_amd.presets import build_nerfdet
    from nerfdet_amd.synth import batch_to, train_scene
    from nerfdet_amd.train import build_optimizer, train_one_step
    import nerfdet_amd.conv3d as C3
    if args.arith:
        C3.set_arithmetic(args.arith)
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
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(args.steps):
            train_one_step(det, data, opt)
        torch.cuda.synchronize()
    n = args.steps
    print(f"# per-operator device time over {n} steps (divide by {n})")
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=70))
    print("# host side: per-operator self CPU time (synchronising calls show up here: aten::item / _local_scalar_dense / nonzero / hipStreamSynchronize)")
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=70))
    print("# grouped by input shape")
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=50, max_shapes_column_width=90))
    print("# the elementwise / copy / reduction operators by input shape (device time over the profiled steps)")
    by_shape = prof.key_averages(group_by_input_shape=True)
    for op in ("aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::fill_", "aten::cat", "aten::div", "aten::threshold_backward", "aten::where", "aten::index"):
        rows = sorted((e for e in by_shape if e.key == op), key=lambda e: -e.self_device_time_total)[:10]
        for e in rows:
            print(f"{op:26s} {e.self_device_time_total / 1e3:8.3f} ms {e.count:5d} calls  {str(e.input_shapes)[:150]}")
    print("# grouped by call site")
    print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=50, max_name_column_width=40, max_src_column_width=110))


if __name__ == "__main__":
    main()
