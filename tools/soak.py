"""GPU box: stability soak -- 400 sequential scenes, 400 with two in flight, 180 training steps; reports drift of step time and of
allocated memory, and that losses stay finite and go down."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda")
w = bench.WORKLOADS["cfg2"]
det = bench.build_model(w).to(dev)
batch = bench.to_device(bench.synth_batch(w, 0), dev)
with torch.no_grad():
    for _ in range(5):
        det(return_loss=False, **batch)
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    ts = []
    for blk in range(4):
        t = time.perf_counter()
        for _ in range(100):
            r = det(return_loss=False, **batch)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 10)
    print("sequential ms/scene per 100:", [round(x, 3) for x in ts], "alloc delta MB:", (torch.cuda.memory_allocated() - m0) / 1e6, "detections", len(r[0]["scores_3d"]))
print("two in flight:", [round(bench.serve_in_flight(det, batch, 100), 2) for _ in range(4)], "alloc delta MB:", (torch.cuda.memory_allocated() - m0) / 1e6)
del det
torch.cuda.empty_cache()
from nerfdet_amd.presets import build_nerfdet
from nerfdet_amd.synth import batch_to, train_scene
from nerfdet_amd.train import build_optimizer, train_one_step
torch.manual_seed(0)
model = build_nerfdet(50, depth_supervise=True)
with torch.no_grad():
    model.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
    model.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
model.to(dev).train()
opt = build_optimizer(model)
data = batch_to(train_scene(40, (240, 320), t_views=10, n_boxes=8, seed=0), dev)
losses, ts, mem = [], [], []
for blk in range(6):
    t = time.perf_counter()
    for _ in range(30):
        out = train_one_step(model, data, opt)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) / 30 * 1e3)
    losses.append(round(out["log_vars"]["loss"], 4))
    mem.append(round(torch.cuda.memory_allocated() / 1e6, 1))
    assert all(v == v and abs(v) < 1e6 for v in out["log_vars"].values()), out["log_vars"]
print("train ms/step per 30 (host read every step):", [round(x, 2) for x in ts], "loss after each block:", losses, "allocated MB after each block:", mem,
      "peak GB", torch.cuda.max_memory_allocated() / 1e9)
assert max(mem[2:]) - min(mem[2:]) < 64.0, mem           # no drift once the step's buffers exist (amax slot pools, weight planes, StepLog)
assert losses[-1] < losses[0]
