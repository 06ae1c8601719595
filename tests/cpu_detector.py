"""Test helper: run the ``nerfdet`` module tree on the CPU by standing the ORACLE in for the HIP ops.

The product path has no CPU fallback.  The multi-process tests that must run without a GPU
(DistributedDataParallel over gloo: parameter set, dead parameters, frozen stages, loss reductions) only need *a*
differentiable stand-in for the three places where the detector calls into ``libnerfdet_hip.so``; inside ``tests/`` the
oracle may play that part.  Nothing here is importable from the package."""
from __future__ import annotations

import contextlib

import numpy as np
import torch

from oracle import nerfdet_oracle as O


def _sd(module):
    d = dict(module.named_parameters())
    d.update(dict(module.named_buffers()))
    return d


def _extract_volume(features, denorm_images, img_meta, n_voxels, voxel_size, mapping, nerf_mlp, stride=4, channels_last_out=True,
                    feature_2d=None, geometry=None):
    lin = mapping[0]
    out = O.extract_volume(features.contiguous(), denorm_images, img_meta, list(n_voxels), list(voxel_size), lin.weight, lin.bias,
                           _sd(nerf_mlp), stride=stride)
    h, w = img_meta["img_shape"][0] // stride, img_meta["img_shape"][1] // stride
    return dict(volume=out["volume"], valid=out["valid"], feature_2d=O.map_features_2d(features[:, :, :h, :w].contiguous(), lin.weight, lin.bias))


def _make_render_rays(rng_holder):
    def render_rays(ray_batch, mean_volume, cov_volume, features_2D, img, aabb, near_far_range, N_samples, N_rand=4096, nerf_mlp=None,
                    img_meta=None, projector=None, mode="volume", nerf_sample_view=3, inv_uniform=False, N_importance=0, det=False,
                    is_train=True, white_bkgd=False, render_testing=False, selection=None):
        if not is_train:
            return None
        ro, rd, rgb, dep = O.select_training_rays(ray_batch, N_rand, rng_holder["rng"])
        ret = O.render_rays_func(ro, rd, features_2D, img, near_far_range, N_samples, _sd(nerf_mlp), img_meta, det=True)
        ret.update(gt_rgb=rgb, gt_depth=dep)
        return ret
    return render_rays


@contextlib.contextmanager
def oracle_backed_cpu_ops(seed: int = 234):
    """Inside the block ``nerfdet(...).forward_train`` runs on CPU tensors (deterministic ray sampling)."""
    import nerfdet_amd.detector as D
    import nerfdet_amd.head as H
    import nerfdet_amd.rays as R
    holder = {"rng": np.random.RandomState(seed)}
    saved = (D.extract_volume, R.render_rays, H.ops.get_points)
    saved_sel = (R.begin_selection, R.finish_selection)      # the oracle draws the rays itself (same RandomState protocol)
    R.begin_selection, R.finish_selection = (lambda ray_batch: None), (lambda begun, n_rand: None)

    def get_points(n_voxels, voxel_size, origin, device=None):
        vs = voxel_size.tolist() if isinstance(voxel_size, torch.Tensor) else list(voxel_size)
        return O.get_points([int(v) for v in n_voxels], vs, origin)
    D.extract_volume, R.render_rays, H.ops.get_points = _extract_volume, _make_render_rays(holder), get_points
    try:
        yield holder
    finally:
        D.extract_volume, R.render_rays, H.ops.get_points = saved
        R.begin_selection, R.finish_selection = saved_sel
