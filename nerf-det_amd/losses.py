"""Losses named by the nerfdet configs (A16).  FocalLoss / CrossEntropyLoss / weighted_loss are mmdet 2.10
third-party code that is not in the reference tree: restated from their documented behaviour
(parity unpinned); AxisAlignedIoULoss follows
mmdet3d/models/losses/axis_aligned_iou_loss.py:9-78 + core/bbox/iou_calculators/iou3d_calculator.py:201-330."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .registry import LOSSES


def _reduce(loss, weight, reduction, avg_factor):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss
    if reduction == "mean":
        return loss.sum() / avg_factor
    if reduction == "none":
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def aligned_iou_3d(a, b, eps: float = 1e-6):
    """IoU of paired (x1,y1,z1,x2,y2,z2) boxes, union clamped at eps (iou3d_calculator.py:264-323)."""
    va = (a[..., 3] - a[..., 0]) * (a[..., 4] - a[..., 1]) * (a[..., 5] - a[..., 2])
    vb = (b[..., 3] - b[..., 0]) * (b[..., 4] - b[..., 1]) * (b[..., 5] - b[..., 2])
    ext = (torch.min(a[..., 3:], b[..., 3:]) - torch.max(a[..., :3], b[..., :3])).clamp(min=0)
    inter = ext[..., 0] * ext[..., 1] * ext[..., 2]
    union = (va + vb - inter).clamp(min=eps)      # = max(., eps) without shipping a constant to the device
    return inter / union


@LOSSES.register_module()
class AxisAlignedIoULoss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        assert reduction in ("none", "sum", "mean")
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kw):
        reduction = reduction_override or self.reduction
        if weight is not None and not torch.any(weight > 0) and reduction != "none":
            return (pred * weight).sum()
        return _reduce(1 - aligned_iou_3d(pred, target), weight, reduction, avg_factor) * self.loss_weight


@LOSSES.register_module()
class FocalLoss(nn.Module):
    """Sigmoid focal loss; labels outside [0, n_classes) (the head uses -1 for background,
    imvoxel_head_v2.py:521-522) are all-negative rows, as the mmcv op treats them."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0):
        super().__init__()
        assert use_sigmoid
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        n_cls = pred.shape[1]
        t = torch.zeros_like(pred)
        fg = (target >= 0) & (target < n_cls)
        t.scatter_(1, target.clamp(0, n_cls - 1).view(-1, 1), fg.to(pred.dtype).view(-1, 1))   # one-hot rows, nothing read back by the host
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        fw = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction="none") * fw
        if weight is not None:
            weight = weight.view(-1, 1) if weight.dim() == 1 else weight
        return _reduce(loss, weight, reduction_override or self.reduction, avg_factor) * self.loss_weight


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, reduction="mean", loss_weight=1.0, **kw):
        super().__init__()
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kw):
        if self.use_sigmoid:
            loss = F.binary_cross_entropy_with_logits(pred, target.float(), reduction="none")
        else:
            loss = F.cross_entropy(pred, target, reduction="none")
        return _reduce(loss, weight, reduction_override or self.reduction, avg_factor) * self.loss_weight
