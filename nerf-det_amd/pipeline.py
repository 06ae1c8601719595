"""Host side of the input contract: the batch ``nerfdet.forward_*`` expects, assembled on
the GPU from decoded frames -- the counterpart of ``ScanNetMultiViewDataset.get_data_info``
(mmdet3d/datasets/scannet_monocular_dataset.py:16-76), ``MultiViewPipeline`` (datasets/pipelines/multi_view.py:13-196),
``get_dtu_raydir`` (data_augment_utils.py:410-424) and ``DefaultFormatBundle3D`` (formating.py:33-117, 216-290).

JPEG decoding, ``Resize(keep_ratio)`` and ``Pad`` stay upstream (third-party mmdet / cv2 transforms): this module takes the
scene's frames already at network resolution as one uint8 BGR tensor on the device, so per step only the selected views
are normalised and only the target views get rays -- two kernel launches instead of 50-100 per-frame numpy passes on the
single data-loader worker the reference configures (config:134).
"""
from __future__ import annotations

from ctypes import c_void_p
from typing import Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check

IMG_MEAN = (123.675, 116.28, 103.53)   # configs/nerfdet/*.py img_norm_cfg (RGB)
IMG_STD = (58.395, 57.12, 57.375)


def _ptr(t):
    return c_void_p(t.data_ptr())


def get_dtu_raydir(pixelcoords, intrinsic, rot, dir_norm=None):
    """Same signature and arithmetic as the reference helper (numpy): un-normalised camera rays rotated to the world."""
    x = (pixelcoords[..., 0] + 0.5 - intrinsic[0, 2]) / intrinsic[0, 0]
    y = (pixelcoords[..., 1] + 0.5 - intrinsic[1, 2]) / intrinsic[1, 1]
    dirs = np.stack([x, y, np.ones_like(x)], axis=-1) @ rot[:, :].T
    if dir_norm:
        dirs = dirs / (np.linalg.norm(dirs, axis=-1, keepdims=True) + 1e-5)
    return dirs


def scene_cameras(info: dict) -> dict:
    """The camera part of ``get_data_info``: ``extrinsic = inv(axis_align @ pose)`` (world -> camera), ``c2w``, its rotation
    and translation (``camrotc2w``, ``lightpos``), the fp32 intrinsics and the fixed voxel origin (0, 0, 0.5)."""
    axis_align = np.asarray(info["annos"]["axis_align_matrix"]).astype(np.float32)
    c2w = [(axis_align @ np.asarray(p)).astype(np.float32) for p in info["extrinsics"]]
    return dict(extrinsic=[np.linalg.inv(axis_align @ np.asarray(p)).astype(np.float32) for p in info["extrinsics"]],
                intrinsic=np.asarray(info["intrinsics"]).astype(np.float32), origin=np.array([.0, .0, .5], dtype=np.float32),
                c2w=c2w, camrotc2w=[m[0:3, 0:3] for m in c2w], lightpos=[m[0:3, 3] for m in c2w])


def select_views(n_frames: int, n_images: int, nerf_target_views: int, loading: str = "random", sample_freq: int = 3):
    """View sampling of ``MultiViewPipeline`` (multi_view.py:60-83) on numpy's global RNG stream, like the reference:
    'random' draws ``n_images`` (with replacement only when the scene is shorter), takes ``nerf_target_views`` of them as
    NeRF targets and removes those from the sources with ``setdiff1d`` (which also sorts and de-duplicates); any other
    mode takes every ``sample_freq``-th frame and uses all of them as targets too."""
    if loading == "random":
        ids = np.random.choice(np.arange(n_frames), n_images, replace=n_images > n_frames)
        if nerf_target_views == 0:
            return ids.tolist(), []
        target = np.random.choice(ids, nerf_target_views, replace=False)
        return np.setdiff1d(ids, target).tolist(), target.tolist()
    ids = np.arange(0, n_images * sample_freq, sample_freq).tolist()
    return ids, (list(ids) if nerf_target_views != 0 else [])


class MultiViewPipeline:
    """``MultiViewPipeline`` + ``DefaultFormatBundle3D`` + batch-1 collate for frames resident on the GPU.

    ``__call__(frames, cams, ori_shape)`` -> the keyword arguments of ``nerfdet.forward_train / forward_test``:
    ``img (1,n_v,3,H,W)``, ``denorm_images (1,n_v,3,H,W)``, ``raydirs / lightpos / gt_images (1,T,R,3)``, ``nerf_sizes``,
    ``depth_range`` and ``img_metas`` with ``lidar2img{intrinsic, extrinsic[], origin}``, ``ori_shape``, ``img_shape``."""

    def __init__(self, n_images: int, mean: Sequence[float] = IMG_MEAN, std: Sequence[float] = IMG_STD, margin: int = 10,
                 depth_range=(0.5, 5.5), loading: str = "random", nerf_target_views: int = 0, sample_freq: int = 3):
        self.n_images, self.margin, self.depth_range = n_images, margin, list(depth_range)
        self.loading, self.nerf_target_views, self.sample_freq = loading, nerf_target_views, sample_freq
        self.mean = np.asarray(mean, dtype=np.float64)
        self.std = np.asarray(std, dtype=np.float64)

    def __call__(self, frames: torch.Tensor, cams: dict, ori_shape, ids=None, target_ids=None) -> dict:
        if not frames.is_cuda:
            raise RuntimeError("nerfdet_amd.pipeline: frames must live on the GPU (no CPU fallback)")
        assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3 and frames.is_contiguous()
        n_frames, h, w, _ = frames.shape
        if ids is None:
            ids, target_ids = select_views(n_frames, self.n_images, self.nerf_target_views, self.loading, self.sample_freq)
        target_ids = list(target_ids or [])
        dev = frames.device
        lib = _lib.load()
        st = c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        mean, std = self.mean, self.std
        mp, sp = mean.ctypes.data_as(c_void_p), std.ctypes.data_as(c_void_p)
        ids_d = torch.tensor(ids, dtype=torch.int32, device=dev)
        img = torch.empty((len(ids), 3, h, w), dtype=torch.float32, device=dev)
        denorm = torch.empty_like(img)
        check(lib.ndet_normalize_views(_ptr(frames), _ptr(ids_d), len(ids), h, w, mp, sp, _ptr(img), _ptr(denorm), st), "normalize_views")
        meta = dict(lidar2img=dict(intrinsic=cams["intrinsic"], extrinsic=[cams["extrinsic"][i] for i in ids], origin=cams["origin"]),
                    ori_shape=tuple(ori_shape), img_shape=(h, w, 3), pad_shape=(h, w, 3))
        batch = dict(img=img.unsqueeze(0), img_metas=[meta])
        if target_ids:
            ratio = ori_shape[0] / h
            k = cams["intrinsic"].copy()
            k[:2] = k[:2] / ratio                                     # multi_view.py:113-114
            krows = torch.from_numpy(np.ascontiguousarray(k[:2, :3])).to(dev)
            rot = torch.from_numpy(np.stack(cams["camrotc2w"])).to(dev).contiguous()
            lpos = torch.from_numpy(np.stack(cams["lightpos"])).to(dev).contiguous()
            t_d = torch.tensor(target_ids, dtype=torch.int32, device=dev)
            rays_per = (h - 2 * self.margin) * (w - 2 * self.margin)
            raydirs = torch.empty((len(target_ids), rays_per, 3), dtype=torch.float32, device=dev)
            lightpos, gt = torch.empty_like(raydirs), torch.empty_like(raydirs)
            check(lib.ndet_target_rays(_ptr(frames), _ptr(t_d), len(target_ids), h, w, self.margin, _ptr(krows), _ptr(rot), _ptr(lpos), mp, sp,
                                       _ptr(raydirs), _ptr(lightpos), _ptr(gt), st), "target_rays")
            size = torch.tensor([[h - 2 * self.margin, w - 2 * self.margin, 3]])
            batch.update(denorm_images=denorm.unsqueeze(0), raydirs=raydirs.unsqueeze(0), lightpos=lightpos.unsqueeze(0),
                         gt_images=gt.unsqueeze(0), gt_depths=[], nerf_sizes=[size.clone() for _ in target_ids],
                         depth_range=torch.tensor([[self.depth_range]], dtype=torch.float32, device=dev),
                         c2w=[cams["c2w"][i] for i in target_ids])
        return batch
