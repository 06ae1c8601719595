"""Deterministic synthetic ScanNet-shaped inputs for bench.py and the examples:
cameras on a ring of radius 2.5 m at height 1.2 m looking at the world origin, ``ori_shape`` = 2x the
network input, ScanNet-like focal length.  Gives a valid voxel-view fraction of about 0.30 at cfg2."""
from __future__ import annotations

import math

import numpy as np


def ring_scene_meta(n_views: int, img_hw=(240, 320), radius: float = 2.5, height: float = 1.2, origin=(0.0, 0.0, 0.5)) -> dict:
    h, w = img_hw
    oh, ow = 2 * h, 2 * w
    k = np.eye(4, dtype=np.float32)
    k[0, 0] = k[1, 1] = 577.87 * (oh / 480.0)
    k[0, 2] = ow / 2.0 - 0.5
    k[1, 2] = oh / 2.0 - 0.5
    up = np.array([0.0, 0.0, 1.0])
    extrinsic = []
    for i in range(n_views):
        a = 2.0 * math.pi * i / n_views
        c = np.array([radius * math.cos(a), radius * math.sin(a), height])
        fwd = -c / np.linalg.norm(c)
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        rot = np.stack([right, np.cross(fwd, right), fwd])
        e = np.eye(4)
        e[:3, :3] = rot
        e[:3, 3] = -rot @ c
        extrinsic.append(e.astype(np.float32))
    return dict(lidar2img=dict(intrinsic=k, extrinsic=extrinsic, origin=np.asarray(origin, dtype=np.float32)),
                ori_shape=(oh, ow, 3), img_shape=(h, w, 3))


def train_scene(n_views: int, img_hw=(240, 320), t_views: int = 10, n_boxes: int = 8, seed: int = 0, margin: int = 10) -> dict:
    """One synthetic *training* sample in the collated batch format of SURVEY.md appendix B (B = 1): ``n_views`` source views,
    ``t_views`` NeRF target views with ``(H-2*margin) x (W-2*margin)`` rays each (multi_view.py:124-132), gt colours / depths
    (config:102 range), ``n_boxes`` axis-aligned GT boxes inside the grid with labels in [0, 18)."""
    import torch
    from .boxes import DepthInstance3DBoxes
    g = torch.Generator().manual_seed(seed)
    h, w = img_hw
    rh, rw = h - 2 * margin, w - 2 * margin
    nray = rh * rw
    ang = torch.rand(1, t_views, 1, generator=g) * 2 * math.pi
    cam = torch.cat([2.5 * torch.cos(ang), 2.5 * torch.sin(ang), 1.2 + 0 * ang], -1)
    ray_o = cam.unsqueeze(2).expand(1, t_views, nray, 3).contiguous()
    ray_d = -ray_o / ray_o.norm(dim=-1, keepdim=True) + 0.35 * torch.randn(1, t_views, nray, 3, generator=g)
    ctr = torch.rand(n_boxes, 3, generator=g) * torch.tensor([5.0, 5.0, 1.5]) + torch.tensor([-2.5, -2.5, -0.5])
    size = 0.6 + torch.rand(n_boxes, 3, generator=g)
    img, denorm, gt_images = torch.randn(1, n_views, 3, h, w, generator=g), torch.rand(1, n_views, 3, h, w, generator=g), torch.rand(1, t_views, nray, 3, generator=g)
    gt_depths = torch.rand(1, t_views, rh, rw, generator=g) * 5 + 0.5
    return dict(img=img, img_metas=[ring_scene_meta(n_views, img_hw)],
                denorm_images=denorm, lightpos=ray_o, raydirs=ray_d,
                gt_images=gt_images, gt_depths=gt_depths,
                depth_rays=torch.nonzero(gt_depths.view(-1) > 0).view(1, -1),       # as the dataset pipeline leaves it (datasets.py)
                nerf_sizes=[torch.tensor([[rh, rw, 3]])],
                gt_bboxes_3d=[DepthInstance3DBoxes(torch.cat([ctr, size], 1), box_dim=6, with_yaw=False, origin=(0.5, 0.5, 0.5))],
                gt_labels_3d=[torch.randint(0, 18, (n_boxes,), generator=g)])


def batch_to(batch: dict, device) -> dict:
    """Tensors (and box containers inside lists) of a collated batch onto ``device``; everything else as is."""
    import torch
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(device)
        elif isinstance(v, list) and v and hasattr(v[0], "to") and not isinstance(v[0], dict) and k != "nerf_sizes":
            out[k] = [x.to(device) for x in v]
        else:
            out[k] = v
    return out
