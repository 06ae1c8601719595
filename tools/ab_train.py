"""GPU box: the training step with a module-level switch on and off, alternating in one process:  ab_train.py <module> <attribute>."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd.presets import build_nerfdet
from nerfdet_amd.synth import batch_to, train_scene
from nerfdet_amd.train import build_optimizer, train_one_step

mod = importlib.import_module(sys.argv[1])
attr = sys.argv[2]
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_nerfdet(50, depth_supervise=True)
with torch.no_grad():
    model.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
    model.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
model.to(dev).train()
opt = build_optimizer(model)
data = batch_to(train_scene(40, (240, 320), t_views=10, n_boxes=8, seed=0), dev)


def run(n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        train_one_step(model, data, opt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for flag in (True, False):
    setattr(mod, attr, flag)
    run(3)
for rep in range(3):
    for flag in (True, False):
        setattr(mod, attr, flag)
        torch.cuda.reset_peak_memory_stats()
        ms = run(10)
        print(f"rep {rep} {attr}={flag}: {ms:.2f} ms/step, peak {torch.cuda.max_memory_allocated() / 1e9:.2f} GB", flush=True)
