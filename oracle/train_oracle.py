"""TEST INFRASTRUCTURE -- CPU restatement of row A16 (training targets and losses) of SURVEY.md section 8a.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
path (``nerfdet_amd.head`` / ``nerfdet_amd.losses``) never does.

Pinned: ``tests/test_oracle_golden.py`` / ``tests/test_train_targets.py`` hold every function here to the vectors under
``tests/golden/train_targets_s*.npz``, which ``tests/golden/make_golden_train.py`` produced by executing the reference's own
``get_targets`` / ``_loss_single`` / ``compute_centerness`` / ``AxisAlignedIoULoss``.  FocalLoss / CrossEntropyLoss are
mmdet 2.10 third-party code absent from the reference tree: restated from documented behaviour, parity unpinned.

Written box-by-box with explicit loops (the reference -- and the product code -- use dense (points x boxes) tensor
algebra), so that agreement between the two is not an artefact of shared structure.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import nerfdet_oracle as O

Tensor = torch.Tensor
FLOAT_MAX = 1e8  # imvoxel_head_v2.py:458


def level_points(grid: Sequence[int], voxel_size: Sequence[float], origin, n_scales: int = 3) -> List[Tensor]:
    """imvoxel_head_v2.py:205-214: per-level lattice (n_i,3), voxel size doubled per level."""
    out = []
    for i in range(n_scales):
        size = [int(g) // 2 ** i for g in grid]
        vs = (torch.tensor(voxel_size, dtype=torch.float32) * (2 ** i)).tolist()
        out.append(O.get_points(size, vs, origin).reshape(3, -1).transpose(0, 1).contiguous())
    return out


def compute_centerness(t: Tensor) -> Tensor:
    """imvoxel_head_v2.py:558-566."""
    lo = torch.stack([torch.minimum(t[..., 0], t[..., 1]), torch.minimum(t[..., 2], t[..., 3]), torch.minimum(t[..., 4], t[..., 5])], -1)
    hi = torch.stack([torch.maximum(t[..., 0], t[..., 1]), torch.maximum(t[..., 2], t[..., 3]), torch.maximum(t[..., 4], t[..., 5])], -1)
    return torch.sqrt(lo[..., 0] / hi[..., 0] * lo[..., 1] / hi[..., 1] * lo[..., 2] / hi[..., 2])


def face_distances(points: Tensor, centre: Tensor, size: Tensor) -> Tensor:
    """(n,6) distances of every lattice point to the six faces of one box, in the reference's op order
    ``p - c + s/2`` / ``c + s/2 - p`` (imvoxel_head_v2.py:477-483)."""
    cols = []
    for a in range(3):
        cols.append(points[:, a] - centre[a] + size[a] / 2)
        cols.append(centre[a] + size[a] / 2 - points[:, a])
    return torch.stack(cols, dim=-1)


def get_targets(points: Sequence[Tensor], gt_gravity_center: Tensor, gt_size: Tensor, gt_labels: Tensor,
                limit: int = 27, centerness_topk: int = 18) -> Tuple[Tensor, Tensor, Tensor]:
    """imvoxel_head_v2.py:457-526 -> (centerness_targets (n), bbox_targets (n,6) corners, labels (n), -1 = background).

    Per box: (1) the lattice points strictly inside it; (2) its scale = the first level holding fewer than ``limit`` inside
    points, minus one (clamped at 0; the last level when every level holds at least ``limit``); (3) on that level, the points
    whose centerness is strictly above the (topk+1)-th largest.  A point claimed by several boxes goes to the smallest
    volume (first box on ties); ``bbox_targets`` of background points follow box 0, as ``min`` over an all-``1e8`` row does."""
    n_scales = len(points)
    level = torch.cat([torch.full((len(p),), i, dtype=torch.long) for i, p in enumerate(points)])
    pts = torch.cat(list(points), dim=0)
    n = len(pts)
    n_box = len(gt_labels)
    volume = gt_size[:, 0] * gt_size[:, 1] * gt_size[:, 2]
    best_vol = torch.full((n,), FLOAT_MAX, dtype=torch.float32)
    owner = torch.zeros((n,), dtype=torch.long)
    dist_all = []
    for j in range(n_box):
        d = face_distances(pts, gt_gravity_center[j], gt_size[j])
        dist_all.append(d)
        inside = d.min(dim=-1)[0] > 0
        per_level = [int(inside[level == i].sum()) for i in range(n_scales)]
        scale = n_scales - 1
        for i in range(n_scales):
            if per_level[i] < limit:
                scale = max(i - 1, 0)
                break
        cness = compute_centerness(d)
        cness = torch.where(inside & (level == scale), cness, torch.full_like(cness, -1.0))
        kth = torch.sort(cness, descending=True)[0][centerness_topk]
        take = inside & (level == scale) & (cness > kth)
        vol_j = torch.where(take, volume[j].expand(n), torch.full((n,), FLOAT_MAX))
        better = vol_j < best_vol          # strict: earlier boxes win ties, like min(dim=1) on the CPU
        best_vol = torch.where(better, vol_j, best_vol)
        owner = torch.where(better, torch.full_like(owner, j), owner)
    labels = torch.where(best_vol == FLOAT_MAX, torch.full((n,), -1, dtype=gt_labels.dtype), gt_labels[owner])
    d = torch.stack(dist_all, dim=1)[torch.arange(n), owner]
    boxes = torch.stack([pts[:, 0] - d[:, 0], pts[:, 1] - d[:, 2], pts[:, 2] - d[:, 4],
                         pts[:, 0] + d[:, 1], pts[:, 1] + d[:, 3], pts[:, 2] + d[:, 5]], -1)   # imvoxel_head_v2.py:547-555
    return compute_centerness(d), boxes, labels


def aligned_iou_pairs(a: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """iou3d_calculator.py:264-323, ``is_aligned=True``, mode 'iou'."""
    va = (a[:, 3] - a[:, 0]) * (a[:, 4] - a[:, 1]) * (a[:, 5] - a[:, 2])
    vb = (b[:, 3] - b[:, 0]) * (b[:, 4] - b[:, 1]) * (b[:, 5] - b[:, 2])
    out = torch.empty(len(a), dtype=a.dtype)
    for i in range(len(a)):
        w = [torch.clamp(torch.minimum(a[i, 3 + k], b[i, 3 + k]) - torch.maximum(a[i, k], b[i, k]), min=0) for k in range(3)]
        inter = w[0] * w[1] * w[2]
        union = torch.maximum(va[i] + vb[i] - inter, torch.tensor(eps, dtype=a.dtype))
        out[i] = inter / union
    return out


def iou_loss(pred: Tensor, target: Tensor, weight=None, avg_factor=None, reduction: str = "mean") -> Tensor:
    """AxisAlignedIoULoss.forward (axis_aligned_iou_loss.py:44-78) over mmdet's ``weighted_loss`` reduction."""
    if weight is not None and not bool(torch.any(weight > 0)) and reduction != "none":
        return (pred * weight).sum()
    loss = 1 - aligned_iou_pairs(pred, target)
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss
    assert reduction in ("mean", "none")
    return loss.sum() / avg_factor if reduction == "mean" else loss


def focal_loss(pred: Tensor, labels: Tensor, avg_factor, gamma: float = 2.0, alpha: float = 0.25) -> Tensor:
    """mmdet 2.10 sigmoid focal loss, label -1 = background (third-party; documented behaviour, unpinned)."""
    total = pred.new_zeros(())
    p = pred.sigmoid()
    for c in range(pred.shape[1]):
        t = (labels == c).to(pred.dtype)
        pt = (1 - p[:, c]) * t + p[:, c] * (1 - t)
        w = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
        total = total + (F.binary_cross_entropy_with_logits(pred[:, c], t, reduction="none") * w).sum()
    return total / avg_factor


def loss_single(ctr: Sequence[Tensor], reg: Sequence[Tensor], cls: Sequence[Tensor], valids: Sequence[Tensor], points: Sequence[Tensor],
                gt_gravity_center: Tensor, gt_size: Tensor, gt_labels: Tensor, limit: int = 27, centerness_topk: int = 18,
                world_n_pos: float = None) -> Tuple[Tensor, Tensor, Tensor]:
    """imvoxel_head_v2.py:116-203 for one scene -> (loss_centerness, loss_bbox, loss_cls)."""
    n_cls = cls[0].shape[0]
    ct, bt, lab = get_targets(points, gt_gravity_center, gt_size, gt_labels, limit, centerness_topk)
    f_ctr = torch.cat([c.permute(1, 2, 3, 0).reshape(-1) for c in ctr])
    f_reg = torch.cat([r.permute(1, 2, 3, 0).reshape(-1, r.shape[0]) for r in reg])
    f_cls = torch.cat([c.permute(1, 2, 3, 0).reshape(-1, n_cls) for c in cls])
    f_val = torch.cat([v.permute(1, 2, 3, 0).reshape(-1) for v in valids]).bool()
    pts = torch.cat(list(points))
    pos = [i for i in range(len(lab)) if lab[i] >= 0 and f_val[i]]
    n_pos = max(float(len(pos)) if world_n_pos is None else world_n_pos, 1.0)
    loss_cls = focal_loss(f_cls[f_val], lab[f_val], n_pos) if bool(f_val.any()) else f_cls[f_val].sum()
    if not pos:
        return f_ctr[pos].sum(), f_reg[pos].sum(), loss_cls
    pos = torch.tensor(pos)
    loss_ctr = F.binary_cross_entropy_with_logits(f_ctr[pos], ct[pos], reduction="sum") / n_pos
    d, p = f_reg[pos], pts[pos]
    pred = torch.stack([p[:, 0] - d[:, 0], p[:, 1] - d[:, 2], p[:, 2] - d[:, 4], p[:, 0] + d[:, 1], p[:, 1] + d[:, 3], p[:, 2] + d[:, 5]], -1)
    loss_box = iou_loss(pred, bt[pos], weight=ct[pos], avg_factor=ct[pos].sum())
    return loss_ctr, loss_box, loss_cls


def level_valids(valid: Tensor, sizes: Sequence[Sequence[int]]) -> List[Tensor]:
    """imvoxel_head_v2.py:92-94: trilinear resample of the view count, round, bool."""
    return [F.interpolate(valid, size=tuple(s), mode="trilinear").round().bool() for s in sizes]
