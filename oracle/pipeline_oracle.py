"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the input contract of the hot path.

Follows, statement for statement:
  * ``ScanNetMultiViewDataset.get_data_info``  mmdet3d/datasets/scannet_monocular_dataset.py:16-76
  * ``MultiViewPipeline.__call__``             mmdet3d/datasets/pipelines/multi_view.py:46-196
  * ``get_dtu_raydir``                         mmdet3d/datasets/pipelines/data_augment_utils.py:410-424
  * ``DefaultFormatBundle.__call__``           mmdet3d/datasets/pipelines/formating.py:33-117
Pinned by ``tests/golden/pipeline_small.npz`` (made by running those four callables of the real reference, see
``tests/golden/make_golden_pipeline.py``).  mmcv's imnormalize / imdenormalize are third-party and absent: restated from
their documented formulas, parity unpinned for the last float bit of the uint8 round trip.
Only ``tests/`` may import this module.
"""
from __future__ import annotations

import numpy as np


def scene_cameras(info: dict) -> dict:
    """get_data_info (use_ray=True, no depth / lidar): scannet_monocular_dataset.py:33-63."""
    axis_align = info["annos"]["axis_align_matrix"].astype(np.float32)
    extrinsic, c2ws, rots, lightpos = [], [], [], []
    for pose in info["extrinsics"]:
        extrinsic.append(np.linalg.inv(axis_align @ pose).astype(np.float32))
        c2w = (axis_align @ pose).astype(np.float32)
        c2ws.append(c2w)
        rots.append(c2w[0:3, 0:3])
        lightpos.append(c2w[0:3, 3])
    return dict(extrinsic=extrinsic, intrinsic=info["intrinsics"].astype(np.float32), origin=np.array([.0, .0, .5], np.float32),
                c2w=c2ws, camrotc2w=rots, lightpos=lightpos)


def select_views(n_frames: int, n_images: int, nerf_target_views: int, loading: str, sample_freq: int = 3):
    """multi_view.py:60-83, drawing from numpy's global RNG exactly as the reference does."""
    if loading == "random":
        ids = np.arange(n_frames)
        ids = np.random.choice(ids, n_images, replace=n_images > len(ids))
        target = []
        if nerf_target_views != 0:
            target = np.random.choice(ids, nerf_target_views, replace=False)
            ids = np.setdiff1d(ids, target).tolist()
            target = target.tolist()
        return list(ids), list(target)
    ids = np.arange(0, n_images * sample_freq, sample_freq)
    return list(ids), (list(ids) if nerf_target_views != 0 else [])


def get_dtu_raydir(pixelcoords, intrinsic, rot, dir_norm=None):
    """data_augment_utils.py:410-424."""
    x = (pixelcoords[..., 0] + 0.5 - intrinsic[0, 2]) / intrinsic[0, 0]
    y = (pixelcoords[..., 1] + 0.5 - intrinsic[1, 2]) / intrinsic[1, 1]
    dirs = np.stack([x, y, np.ones_like(x)], axis=-1) @ rot.T
    if dir_norm:
        dirs = dirs / (np.linalg.norm(dirs, axis=-1, keepdims=True) + 1e-5)
    return dirs


def imnormalize(img_u8_bgr, mean, std):
    img = img_u8_bgr[..., ::-1].astype(np.float32)
    return (img - mean.astype(np.float32)) * (1.0 / std).astype(np.float32)


def imdenormalize_u8(img, mean, std):
    """imdenormalize(to_bgr=True).astype(np.uint8): multi_view.py:107-109."""
    return (img * std.astype(np.float32) + mean.astype(np.float32))[..., ::-1].astype(np.uint8)


def multi_view_batch(frames_u8_bgr, cams: dict, ids, target_ids, ori_h: int, mean, std, margin: int = 10):
    """MultiViewPipeline.__call__ + DefaultFormatBundle on already resized / padded frames: the tensors of the batch."""
    mean, std = np.asarray(mean, np.float64), np.asarray(std, np.float64)
    h, w = frames_u8_bgr.shape[1:3]
    norm = {int(i): imnormalize(frames_u8_bgr[i], mean, std) for i in set(list(ids) + list(target_ids))}
    img = np.stack([norm[int(i)].transpose(2, 0, 1) for i in ids])
    denorm = np.stack([(imdenormalize_u8(norm[int(i)], mean, std) / 255.0).transpose(2, 0, 1) for i in ids]).astype(np.float32)
    out = dict(img=img, denorm_images=denorm, extrinsic=np.stack([cams["extrinsic"][int(i)] for i in ids]))
    if len(target_ids):
        ratio = ori_h / h
        k = cams["intrinsic"].copy()
        k[:2] = k[:2] / ratio
        px, py = np.meshgrid(np.arange(margin, w - margin).astype(np.float32), np.arange(margin, h - margin).astype(np.float32))
        pix = np.stack((px, py), axis=-1).astype(np.float32)
        rays, lights, gts, sizes = [], [], [], []
        for t in target_ids:
            rays.append(np.reshape(get_dtu_raydir(pix, k, cams["camrotc2w"][int(t)]).astype(np.float32), (-1, 3)))
            g = imdenormalize_u8(norm[int(t)], mean, std)[py.astype(np.int32), px.astype(np.int32), :]
            sizes.append(np.array(g.shape))
            gts.append(np.reshape(g, (-1, 3)) / 255.0)
            lights.append(cams["lightpos"][int(t)])
        rays = np.stack(rays)
        out.update(raydirs=rays, gt_images=np.stack(gts), nerf_sizes=np.stack(sizes),
                   lightpos=np.repeat(np.stack(lights)[:, None, :], rays.shape[1], axis=1), c2w=np.stack([cams["c2w"][int(t)] for t in target_ids]))
    return out
