"""GPU box: training step with torch's fused AdamW vs the foreach form (nerfdet_amd.train.FUSED_ADAMW), two models side by side."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import train as T
from nerfdet_amd.presets import build_nerfdet
from nerfdet_amd.synth import batch_to, train_scene
dev = torch.device("cuda")
data = batch_to(train_scene(40, (240, 320), t_views=10, n_boxes=8, seed=0), dev)
res = {}
for flag in (True, False, True, False):
    T.FUSED_ADAMW = flag
    torch.manual_seed(0)
    model = build_nerfdet(50, depth_supervise=True)
    with torch.no_grad():
        model.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        model.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
    model.to(dev).train()
    opt = T.build_optimizer(model)
    for _ in range(4):
        out = T.train_one_step(model, data, opt)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(15):
        out = T.train_one_step(model, data, opt)
    torch.cuda.synchronize()
    print(f"fused={flag}: {(time.perf_counter() - t) / 15 * 1e3:.2f} ms/step, loss {out['log_vars']['loss']:.4f}", flush=True)
    del model, opt
    torch.cuda.empty_cache()
