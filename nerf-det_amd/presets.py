"""Model hyper-parameters of the shipped nerfdet configs, as plain dicts, for benches and tests that must run
where the reference's ``configs/`` directory is not present (the GPU box).  Values: configs/nerfdet/
nerfdet_res50_2x_low_res.py:1-47 (R50) and nerfdet_res101_2x_low_res_depth_sp.py (R101, depth_supervise)."""
from __future__ import annotations


def nerfdet_cfg(depth: int = 50, depth_supervise: bool = False, n_voxels=(40, 40, 16), voxel_size=(0.16, 0.16, 0.2)):
    model = dict(
        type="nerfdet",
        pretrained=None,  # 'torchvision://resnet50' in the shipped config: a network fetch, unavailable offline
        backbone=dict(type="ResNet", depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type="BN", requires_grad=False), norm_eval=True, style="pytorch"),
        neck=dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=4),
        neck_3d=dict(type="FastIndoorImVoxelNeck", in_channels=256, out_channels=128, n_blocks=[1, 1, 1]),
        bbox_head=dict(type="ScanNetImVoxelHeadV2", loss_bbox=dict(type="AxisAlignedIoULoss", loss_weight=1.0),
                       n_classes=18, n_channels=128, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18),
        voxel_size=tuple(voxel_size), n_voxels=tuple(n_voxels),
        aabb=([-2.7, -2.7, -0.78], [3.7, 3.7, 1.78]), near_far_range=[0.2, 8.0], N_samples=64, N_rand=2048,
        nerf_mode="image", depth_supervise=depth_supervise, use_nerf_mask=True, nerf_sample_view=20, squeeze_scale=4,
        nerf_density=True)
    return dict(model=model, train_cfg=dict(), test_cfg=dict(nms_pre=1000, iou_thr=0.25, score_thr=0.01))


def build_nerfdet(depth: int = 50, **kw):
    from .config import ConfigDict, _wrap
    from .registry import build_detector
    cfg = _wrap(nerfdet_cfg(depth, **kw))
    return build_detector(cfg["model"], train_cfg=cfg["train_cfg"], test_cfg=cfg["test_cfg"])
