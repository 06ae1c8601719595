"""Training-step timing at BASELINE configs[2] shapes on one GPU (fp32): 50 sampled views -> 10 NeRF target views removed ->
40 source views 240x320, 40x40x16 voxels, 2048 rays x 64 samples, all five losses, backward, AdamW step.
Not the headline metric (bench.py is); reports ms/step and the time of the hand-written forward/backward kernels."""
import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd.boxes import DepthInstance3DBoxes
from nerfdet_amd.presets import build_nerfdet
from nerfdet_amd.synth import ring_scene_meta
from nerfdet_amd import rays


def main():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    det = build_nerfdet(50, depth_supervise=True)
    with torch.no_grad():
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
    det.to(dev).train()
    n_v, hw, t_views = 40, (240, 320), 10
    meta = ring_scene_meta(n_v, hw)
    g = torch.Generator().manual_seed(0)
    nray = (hw[0] - 20) * (hw[1] - 20)
    ang = torch.rand(1, t_views, 1, generator=g) * 2 * np.pi
    cam = torch.cat([2.5 * torch.cos(ang), 2.5 * torch.sin(ang), 1.2 + 0 * ang], -1)
    ray_o = cam.unsqueeze(2).expand(1, t_views, nray, 3).contiguous()
    ray_d = -ray_o / ray_o.norm(dim=-1, keepdim=True) + 0.35 * torch.randn(1, t_views, nray, 3, generator=g)
    batch = dict(img=torch.randn(1, n_v, 3, *hw, generator=g), img_metas=[meta], denorm_images=torch.rand(1, n_v, 3, *hw, generator=g),
                 lightpos=ray_o, raydirs=ray_d, gt_images=torch.rand(1, t_views, nray, 3, generator=g),
                 gt_depths=torch.rand(1, t_views, hw[0] - 20, hw[1] - 20, generator=g) * 5 + 0.5,
                 nerf_sizes=[torch.tensor([[hw[0] - 20, hw[1] - 20, 3]])])
    batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    ctr = torch.rand(8, 3, generator=g) * torch.tensor([5.0, 5.0, 1.5]) + torch.tensor([-2.5, -2.5, -0.5])
    size = 0.6 + torch.rand(8, 3, generator=g)
    gt_boxes = [DepthInstance3DBoxes(torch.cat([ctr, size], 1), box_dim=6, with_yaw=False, origin=(0.5, 0.5, 0.5)).to(dev)]
    gt_labels = [torch.randint(0, 18, (8,), generator=g).to(dev)]
    params = [p for p in det.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=1e-4)
    data = dict(batch, gt_bboxes_3d=gt_boxes, gt_labels_3d=gt_labels)

    def step():
        opt.zero_grad(set_to_none=True)
        out = det.train_step(data)
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(params, 35.0)
        opt.step()
        return out
    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps(dict(workload="cfg3 shapes, 1 GPU, fp32: 40 source views, 2048 rays x 64 samples, 5 losses + backward + AdamW",
                          ms_per_train_step=dt * 1e3, scenes_per_s=1 / dt, log_vars=out["log_vars"],
                          peak_mem_GB=torch.cuda.max_memory_allocated() / 1e9)))


if __name__ == "__main__":
    main()
