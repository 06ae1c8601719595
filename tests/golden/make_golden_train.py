#!/usr/bin/env python3
"""Golden vectors for A16 (training targets and losses) from the REAL reference code.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden_train.py

Executes, from where they lie under /root/reference (nothing is copied):
  * ``ScanNetImVoxelHeadV2.get_targets`` / ``_loss_single`` / ``loss`` / ``compute_centerness``
    (mmdet3d/models/dense_heads/imvoxel_head_v2.py:65-203,457-526,558-566),
  * ``AxisAlignedIoULoss`` + ``axis_aligned_bbox_overlaps_3d`` (mmdet3d/models/losses/axis_aligned_iou_loss.py:9-78,
    mmdet3d/core/bbox/iou_calculators/iou3d_calculator.py:201-330),
  * the real ``DepthInstance3DBoxes`` (mmdet3d/core/bbox/structures/{base_box3d,depth_box3d}.py) for ``volume`` /
    ``gravity_center`` (its compiled-extension imports are replaced by empty stand-ins; none is touched by these properties).

Third-party pieces that are NOT in the reference tree (mmdet 2.10 ``weighted_loss``, ``FocalLoss``, ``CrossEntropyLoss``)
are stand-ins restated from their documented behaviour: the vectors pin the reference's *own*
arithmetic (assignment, centerness, IoU loss, masking, averaging factors) and record the stand-ins' outputs for what they
are -- parity for those two classes stays "unpinned" (DESIGN.md section 2).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (shared reference loader and npz writer)


# ---- documented mmdet 2.10 behaviour (third-party, absent from the tree) ---------------------------------------------
def _weight_reduce(loss, weight=None, reduction="mean", avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss
    if reduction == "mean":
        return loss.sum() / avg_factor
    if reduction == "none":
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def weighted_loss(fn):
    def wrapper(pred, target, weight=None, reduction="mean", avg_factor=None, **kw):
        return _weight_reduce(fn(pred, target, **kw), weight, reduction, avg_factor)
    return wrapper


class FocalStandIn(nn.Module):
    """mmdet FocalLoss(use_sigmoid=True, gamma=2, alpha=.25): labels outside [0, n_classes) are all-negative rows."""

    def forward(self, pred, target, weight=None, avg_factor=None):
        t = F.one_hot(target.clamp(min=0), pred.shape[1]).float() * (target >= 0).float()[:, None]
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        fw = (0.25 * t + 0.75 * (1 - t)) * pt.pow(2.0)
        return _weight_reduce(F.binary_cross_entropy_with_logits(pred, t, reduction="none") * fw, weight, "mean", avg_factor)


class BCEStandIn(nn.Module):
    """mmdet CrossEntropyLoss(use_sigmoid=True)."""

    def forward(self, pred, target, weight=None, avg_factor=None):
        return _weight_reduce(F.binary_cross_entropy_with_logits(pred, target.float(), reduction="none"), weight, "mean", avg_factor)


def load_training_reference():
    ref = MG.load_reference()
    sm = sys.modules
    for p in ["mmdet.models.losses", "mmdet.models.losses.utils", "mmdet.core.bbox", "mmdet.core.bbox.iou_calculators",
              "mmdet.core.bbox.iou_calculators.builder", "mmdet3d.models.losses", "mmdet3d.core.points", "mmdet3d.ops.rotated_iou",
              "mmdet3d.ops.rotated_iou.oriented_iou_loss", "mmdet3d.core.bbox.iou_calculators"]:
        MG._pkg(p)
    sm["mmdet.models.builder"].LOSSES = MG._Registry()
    sm["mmdet.models.losses.utils"].weighted_loss = weighted_loss
    sm["mmdet.core.bbox"].bbox_overlaps = None
    sm["mmdet.core.bbox.iou_calculators.builder"].IOU_CALCULATORS = MG._Registry()
    # compiled extensions / point structures that the two properties used here never touch
    sm["mmdet3d.ops.iou3d"].iou3d_cuda = None
    sm["mmdet3d.ops.rotated_iou.oriented_iou_loss"].cal_giou_3d = None
    sm["mmdet3d.core.points"].BasePoints = object
    sm["mmdet3d.ops"].points_in_boxes_batch = None
    st = "mmdet3d/core/bbox/structures/"
    MG._load("mmdet3d.core.bbox.structures.utils", st + "utils.py")
    sm["mmdet3d.core.bbox.structures"].get_box_type = sm["mmdet3d.core.bbox.structures.utils"].get_box_type
    MG._load("mmdet3d.core.bbox.structures.base_box3d", st + "base_box3d.py")
    ref.boxes = MG._load("mmdet3d.core.bbox.structures.depth_box3d", st + "depth_box3d.py")
    ref.iou_calc = MG._load("mmdet3d.core.bbox.iou_calculators.iou3d_calculator", "mmdet3d/core/bbox/iou_calculators/iou3d_calculator.py")
    sm["mmdet3d.core.bbox"].AxisAlignedBboxOverlaps3D = ref.iou_calc.AxisAlignedBboxOverlaps3D
    ref.iou_loss = MG._load("mmdet3d.models.losses.axis_aligned_iou_loss", "mmdet3d/models/losses/axis_aligned_iou_loss.py")
    return ref


def make_train_fixture(ref, name, seed, grid, voxel_size, n_boxes, batch=2):
    torch.manual_seed(seed)
    rng = np.random.RandomState(seed)

    class _Cfg(dict):
        __getattr__ = dict.__getitem__
    head = ref.head.ScanNetImVoxelHeadV2(n_classes=18, n_channels=8, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18,
                                         test_cfg=_Cfg(nms_pre=1000, iou_thr=0.25, score_thr=0.01))
    head.voxel_size = voxel_size
    head.loss_bbox = ref.iou_loss.AxisAlignedIoULoss(loss_weight=1.0)   # the reference's own
    head.loss_cls = FocalStandIn()
    head.loss_centerness = BCEStandIn()
    gx, gy, gz = grid
    ext = np.array([gx * voxel_size[0], gy * voxel_size[1], gz * voxel_size[2]], dtype=np.float32)
    origin = np.array([0.0, 0.0, 0.5], dtype=np.float32)
    arrays = dict(grid=np.array(grid), voxel_size=np.array(voxel_size, dtype=np.float32), origin=origin, batch=np.array(batch))
    metas, gts, labels_l = [], [], []
    for b in range(batch):
        # boxes from tiny (fewer than `limit` lattice points on every level) to room-sized, some overlapping, one outside
        ctr = (rng.rand(n_boxes, 3).astype(np.float32) - 0.5) * ext * 0.8 + origin
        size = (0.15 + rng.rand(n_boxes, 3).astype(np.float32) ** 2 * 0.7 * ext.min()).astype(np.float32)
        size[0] = ext * 0.9           # covers most of the grid: lands on the coarsest scale
        ctr[0] = origin
        size[1] = 0.1                  # smaller than a voxel: may contain no lattice point at all
        ctr[2] = ctr[3] + 0.05         # nested pair: min-volume tie-break
        size[2] = size[3] * 0.6
        ctr[-1] = origin + ext         # outside the grid
        t7 = np.concatenate([ctr, size, np.zeros((n_boxes, 1), np.float32)], 1)
        boxes = ref.boxes.DepthInstance3DBoxes(torch.from_numpy(t7), box_dim=7, with_yaw=False, origin=(0.5, 0.5, 0.5))
        lab = torch.from_numpy(rng.randint(0, 18, n_boxes))
        metas.append(dict(lidar2img=dict(origin=origin)))
        gts.append(boxes)
        labels_l.append(lab)
        arrays[f"gt_tensor_{b}"] = boxes.tensor          # (n,7) bottom-centred rows as the dataset hands them over
        arrays[f"gt_volume_{b}"] = boxes.volume
        arrays[f"gt_gravity_center_{b}"] = boxes.gravity_center
        arrays[f"gt_labels_{b}"] = lab
    sizes = [(gx // 2 ** i, gy // 2 ** i, gz // 2 ** i) for i in range(3)]
    ctrs = [torch.randn(batch, 1, *s) for s in sizes]
    regs = [torch.exp(0.3 * torch.randn(batch, 6, *s)) * (0.3 * 2 ** i) for i, s in enumerate(sizes)]
    clss = [torch.randn(batch, 18, *s) - 2.0 for s in sizes]
    valid = (torch.rand(batch, 1, *grid) < 0.75).float() * torch.randint(1, 6, (batch, 1, *grid)).float()
    for i in range(3):
        arrays[f"ctr_{i}"], arrays[f"reg_{i}"], arrays[f"cls_{i}"] = ctrs[i], regs[i], clss[i]
    arrays["valid"] = valid
    with torch.no_grad():
        for b in range(batch):
            pts = head.get_points([s for s in sizes], origin, torch.device("cpu"))
            ct, bt, lb = head.get_targets(pts, gts[b], labels_l[b])
            arrays[f"tgt_centerness_{b}"], arrays[f"tgt_bbox_{b}"], arrays[f"tgt_labels_{b}"] = ct, bt, lb
            assert (lb >= 0).sum() > 20, "fixture must have positives"
        valids = [nn.Upsample(size=s, mode="trilinear")(valid).round().bool() for s in sizes]
        for b in range(batch):
            lc, lbx, lcl = head._loss_single([x[b] for x in ctrs], [x[b] for x in regs], [x[b] for x in clss],
                                             [x[b] for x in valids], metas[b], gts[b], labels_l[b])
            arrays[f"loss_single_{b}"] = torch.stack([lc, lbx, lcl])
        tot = head.loss(ctrs, regs, clss, valid, metas, gts, labels_l)
        arrays["loss_total"] = torch.stack([tot["loss_centerness"], tot["loss_bbox"], tot["loss_cls"]])
        # degenerate scene: no location is valid -> the three "sum of nothing" branches (imvoxel_head_v2.py:182,200-202)
        lz = head._loss_single([x[0] for x in ctrs], [x[0] for x in regs], [x[0] for x in clss],
                               [torch.zeros_like(v[0]) for v in valids], metas[0], gts[0], labels_l[0])
        arrays["loss_single_novalid"] = torch.stack(lz)
        # compute_centerness and the IoU loss on free-standing inputs
        d = torch.rand(257, 6) * 2 + 0.01
        arrays["cc_in"], arrays["cc_out"] = d, ref.head.compute_centerness(d)
        a = torch.rand(300, 3) * 4
        pa = torch.cat([a, a + 0.05 + torch.rand(300, 3) * 2], 1)
        sh = torch.randn(300, 3) * 0.7
        pb = torch.cat([a + sh, a + sh + 0.05 + torch.rand(300, 3) * 2], 1)
        pb[:10] = pa[:10] + 50.0      # disjoint pairs: IoU 0
        pb[10:14, 3:] = pb[10:14, :3]  # zero-volume targets
        wgt = torch.rand(300)
        il = ref.iou_loss.AxisAlignedIoULoss()
        arrays.update(iou_a=pa, iou_b=pb, iou_w=wgt,
                      iou_pair=ref.iou_calc.axis_aligned_bbox_overlaps_3d(pa, pb, is_aligned=True),
                      iou_matrix=ref.iou_calc.axis_aligned_bbox_overlaps_3d(pa[:40], pb[:50]),
                      iou_loss_mean=il(pa, pb), iou_loss_weighted=il(pa, pb, weight=wgt, avg_factor=wgt.sum()),
                      iou_loss_none=il(pa, pb, reduction_override="none"), iou_loss_sum=il(pa, pb, weight=wgt, reduction_override="sum"),
                      iou_loss_zero_weight=il(pa, pb, weight=torch.zeros(300, 1).expand(300, 6)))
    MG.npz(name, **arrays)


def main():
    ref = load_training_reference()
    make_train_fixture(ref, "train_targets_s0", 0, grid=(16, 16, 8), voxel_size=(0.4, 0.4, 0.4), n_boxes=9)
    make_train_fixture(ref, "train_targets_s1", 1, grid=(24, 16, 8), voxel_size=(0.25, 0.3, 0.35), n_boxes=12, batch=1)


if __name__ == "__main__":
    main()
