"""The fused inference pipeline of the hot path: FPN level-0 features -> gated voxel volume.

Replaces steps 2-11 of ``nerfdet.extract_feat`` (mmdet3d/models/detectors/nerfdet.py:152-261, image
mode, nerf_density=True).  The reference materialises a (n_views, C, N) volume and passes over it
five times; here nothing larger than the inputs and the outputs ever exists:

    mapped  = Linear(C->cm)(features)            library GEMM, (n_v,h,w,cm) channels-last
    glob    = K2 density_features(mapped, rgb)   (N, 2*(3+cm))
    alpha   = 1-exp(-relu(sigma_MLP([posenc|glob])))
    volume, count = K1 backproject_aggregate(features, alpha)   gating fused into the aggregation

The order differs from the reference (density first, aggregation last) so that the alpha gating
costs no extra pass over the (C, N) volume; the arithmetic per element is the reference's.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import ops

Tensor = torch.Tensor


def map_features_2d(features: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    """``self.mapping`` on every feature pixel (nerfdet.py:194-197).  Logical (n_v,cm,h,w), channels-last
    memory -- with channels-last ``features`` this is a single GEMM on a view, no permute copies."""
    f = ops.to_channels_last(features)
    rows = f.permute(0, 2, 3, 1)  # (n_v,h,w,C) view
    if not rows.is_contiguous():  # an [:h,:w] crop of a padded map
        rows = rows.contiguous()
    from .autograd import LINEAR_ROWS_MIN, LinearRows
    if torch.is_grad_enabled() and rows.is_cuda and rows.numel() // rows.shape[-1] >= LINEAR_ROWS_MIN:
        return LinearRows.apply(rows, weight, bias, False).permute(0, 3, 1, 2)     # weight gradient over 192 000 rows: split (autograd.LinearRows)
    return F.linear(rows, weight, bias).permute(0, 3, 1, 2)


def map_features_2d_hip(features: Tensor, lin: torch.nn.Linear) -> Tensor:
    """Inference form of :func:`map_features_2d`: the Linear as a 1x1 launch of the MFMA kernel (bias in the epilogue)."""
    from .conv3d import conv2d_nhwc, packed_linear
    f = ops.to_channels_last(features)
    rows = f.permute(0, 2, 3, 1)
    if not rows.is_contiguous():
        rows = rows.contiguous()
    return conv2d_nhwc(rows, packed_linear(lin), amax=False).permute(0, 3, 1, 2)


def scene_geometry(img_meta: dict, n_voxels, voxel_size, stride: int, device) -> Dict[str, Tensor]:
    """Per-scene constants of the path (A1 + A2): stride-4 / stride-1 projections and the voxel lattice, on the GPU."""
    return dict(proj=ops.compute_projection(img_meta, stride, device), rgb_proj=ops.compute_projection(img_meta, 1, device),
                points=ops.get_points(n_voxels, voxel_size, img_meta["lidar2img"]["origin"], device))


def density_alpha(features: Tensor, denorm_images: Tensor, img_meta: dict, n_voxels, voxel_size, mapping: torch.nn.Module, nerf_mlp,
                  stride: int = 4, feature_2d: Optional[Tensor] = None, geometry: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    """First half of the inference path: mapping GEMM -> K2 -> sigma-MLP -> per-voxel alpha (nerfdet.py:190-197,232-257).
    Returns everything the aggregation kernel needs (``feat``, ``points``, ``projection``, ``alpha``)."""
    dev = features.device
    h = img_meta["img_shape"][0] // stride
    w = img_meta["img_shape"][1] // stride
    feat = ops.to_channels_last(features)[:, :, :h, :w]
    if geometry is None:  # hipGraph replay passes static device buffers instead (graphed.py)
        geometry = scene_geometry(img_meta, n_voxels, voxel_size, stride, dev)
    proj, rgb_proj, pts = geometry["proj"], geometry["rgb_proj"], geometry["points"]
    lin = mapping[0] if isinstance(mapping, torch.nn.Sequential) else mapping
    training = torch.is_grad_enabled() and (feat.requires_grad or lin.weight.requires_grad)
    if feature_2d is None:
        if not training and lin.in_features % 32 == 0:
            feature_2d = map_features_2d_hip(feat, lin)
        else:
            feature_2d = map_features_2d(feat, lin.weight, lin.bias)
    rgb = denorm_images[:, :, :img_meta["img_shape"][0], :img_meta["img_shape"][1]]
    out = dict(feat=feat, feature_2d=feature_2d, points=pts, projection=proj, rgb_projection=rgb_proj, rgb=rgb, lin=lin)
    if training:
        return out  # the caller continues under autograd
    glob = ops.density_features(feature_2d, lin.bias, rgb, pts, proj, rgb_proj)
    if hasattr(nerf_mlp, "hip_trunk_ok") and nerf_mlp.hip_trunk_ok():
        out.update(global_feat=glob, raw_sigma=None, alpha=nerf_mlp.alpha_from_points(pts, glob))
    else:  # other MLP shapes: library GEMMs
        raw_sigma = nerf_mlp.raw_sigma_from_rows(ops.posenc_concat(pts, glob))
        out.update(global_feat=glob, raw_sigma=raw_sigma, alpha=ops.sigma_to_alpha(raw_sigma))
    return out


def extract_volume(features: Tensor, denorm_images: Tensor, img_meta: dict, n_voxels, voxel_size,
                   mapping: torch.nn.Module, nerf_mlp, stride: int = 4, channels_last_out: bool = True,
                   feature_2d: Optional[Tensor] = None, geometry: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    """One scene.  ``features`` (n_v,C,Hf,Wf) FPN level 0 (channels-last preferred), ``denorm_images``
    (n_v,3,H,W).  Returns ``volume`` (C,X,Y,Z) = alpha * mean (zero where unseen), ``valid`` (1,X,Y,Z) int64
    view count, plus ``feature_2d`` (the mapped map, reused by the ray branch) and ``density``."""
    d = density_alpha(features, denorm_images, img_meta, n_voxels, voxel_size, mapping, nerf_mlp, stride, feature_2d, geometry)
    if "alpha" not in d:
        return _extract_volume_train(d["feat"], d["rgb"], d["points"], d["projection"], d["rgb_projection"], d["lin"], nerf_mlp,
                                     d["feature_2d"], channels_last_out)
    volume, count = ops.backproject_aggregate(d["feat"], d["points"], d["projection"], alpha=d["alpha"], channels_last_out=channels_last_out)
    return dict(volume=volume, valid=count, feature_2d=d["feature_2d"], global_feat=d["global_feat"], raw_sigma=d["raw_sigma"],
                alpha=d["alpha"], points=d["points"], projection=d["projection"], rgb_projection=d["rgb_projection"])


def _extract_volume_train(feat, rgb, pts, proj, rgb_proj, lin, nerf_mlp, feature_2d, channels_last_out):
    """Training form: the same kernels under autograd (nerf_det_amd.autograd); gating stays a differentiable
    tensor product so that d(alpha) and d(mean) come out separately (nerfdet.py:257-261)."""
    from .autograd import BackprojectMean, DensityFeatures
    glob = DensityFeatures.apply(feature_2d, lin.bias, rgb, pts, proj, rgb_proj)
    rows = torch.cat([ops.posenc_concat(pts, None), glob], dim=1)
    raw_sigma = nerf_mlp.raw_sigma_from_rows(rows)
    alpha = 1 - torch.exp(-F.relu(raw_sigma))
    mean, count = BackprojectMean.apply(feat, pts, proj, channels_last_out)
    volume = alpha.view(1, *mean.shape[1:]) * mean
    volume = torch.where((count == 0), torch.zeros_like(volume), volume)
    return dict(volume=volume, valid=count, feature_2d=feature_2d, global_feat=glob, raw_sigma=raw_sigma, alpha=alpha.reshape(-1),
                points=pts, projection=proj, rgb_projection=rgb_proj)
