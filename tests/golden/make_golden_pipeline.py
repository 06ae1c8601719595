#!/usr/bin/env python3
"""Golden vectors for the input contract of the hot path, from the REAL reference code.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden_pipeline.py

The reference's dataset / pipeline modules cannot be imported whole here (mmcv, mmdet, cv2, skimage, matplotlib are
absent), so the four callables of the contract are compiled *from where they lie* -- their function definitions are cut
out of the reference files with ``ast`` at run time (nothing is copied into this repo) and executed unchanged:

  * ``ScanNetMultiViewDataset.get_data_info``   (mmdet3d/datasets/scannet_monocular_dataset.py:16-76)
  * ``MultiViewPipeline.__call__``              (mmdet3d/datasets/pipelines/multi_view.py:46-196)
  * ``get_dtu_raydir``                          (mmdet3d/datasets/pipelines/data_augment_utils.py:410-424)
  * ``DefaultFormatBundle3D.__call__``          (mmdet3d/datasets/pipelines/formating.py:33-117 and its subclass)

Stand-ins, because the third-party pieces are not here: the per-frame image transform chain (``Compose`` of mmdet's
LoadImageFromFile / Resize / Normalize / Pad) is replaced by a table of already transformed frames; ``mmcv.imdenormalize``
by its documented formula ``img * std + mean`` then RGB->BGR in float32; ``DataContainer`` by a plain holder;
``to_tensor`` by ``torch.from_numpy``.  Values that pass through ``imdenormalize(...).astype(uint8)`` (``denorm_images``,
``gt_images``) are therefore pinned up to that stand-in's float rounding.

Outputs are data only: seeded inputs and the reference's outputs for them.
"""
from __future__ import annotations

import ast
import os
import types
from collections import defaultdict
from os import path as osp

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def cut(rel, name, cls=None, glb=None):
    """Compile one function (or method of ``cls``) of a reference file in the namespace ``glb``."""
    src = open(os.path.join(REF, rel)).read()
    tree = ast.parse(src)
    body = tree.body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    fn = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.decorator_list = []
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = dict(glb or {})
    exec(compile(mod, os.path.join(REF, rel), "exec"), ns)
    return ns[name]


class DC:  # mmcv.parallel.DataContainer stand-in
    def __init__(self, data, stack=False, cpu_only=False):
        self.data, self.stack, self.cpu_only = data, stack, cpu_only


def imnormalize(img_u8_bgr, mean, std):
    """mmcv.imnormalize(to_rgb=True): float32, BGR->RGB, (x - mean) * (1 / std)."""
    img = img_u8_bgr[..., ::-1].astype(np.float32)
    return (img - mean.astype(np.float32)) * (1.0 / std).astype(np.float32)


def imdenormalize(img, mean, std, to_bgr=True):
    """mmcv.imdenormalize: img * std + mean, then RGB->BGR (float32)."""
    assert img.dtype != np.uint8
    out = img * std.astype(np.float32) + mean.astype(np.float32)
    return np.ascontiguousarray(out[..., ::-1]) if to_bgr else out


def main():
    rng = np.random.RandomState(7)
    n_frames, hw, ori_hw, margin = 14, (24, 32), (48, 64), 3
    mean = np.array([123.675, 116.28, 103.53])
    std = np.array([58.395, 57.12, 57.375])

    # ---- a synthetic scene in the layout of the ScanNet info pickle --------------------------------------------
    frames = rng.randint(0, 256, size=(n_frames, hw[0], hw[1], 3)).astype(np.uint8)   # already resized + padded, BGR
    poses = []
    for i in range(n_frames):
        a = 2 * np.pi * i / n_frames
        c = np.array([2.5 * np.cos(a), 2.5 * np.sin(a), 1.2])
        fwd = -c / np.linalg.norm(c)
        right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
        r_c2w = np.stack([right, np.cross(fwd, right), fwd], axis=1)
        p = np.eye(4); p[:3, :3] = r_c2w; p[:3, 3] = c
        poses.append(p)
    th = 0.3
    axis_align = np.array([[np.cos(th), -np.sin(th), 0, 0.1], [np.sin(th), np.cos(th), 0, -0.2], [0, 0, 1, 0.05], [0, 0, 0, 1]])
    intrinsic = np.array([[57.8, 0, 31.5, 0], [0, 57.8, 23.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    info = dict(img_paths=[f"posed_images/scene/{i:05d}.jpg" for i in range(n_frames)], extrinsics=poses, intrinsics=intrinsic,
                annos=dict(axis_align_matrix=axis_align, gt_num=0))

    # ---- ScanNetMultiViewDataset.get_data_info ----------------------------------------------------------------
    get_data_info = cut("mmdet3d/datasets/scannet_monocular_dataset.py", "get_data_info", cls="ScanNetMultiViewDataset",
                        glb=dict(np=np, osp=osp, defaultdict=defaultdict))
    ds = types.SimpleNamespace(data_infos=[info], data_root="data/scannet", test_mode=True, filter_empty_gt=False,
                               modality=dict(use_depth=False, use_neuralrecon_depth=False, use_lidar=False, use_ray=True),
                               get_ann_info=lambda index: dict(gt_bboxes_3d=np.zeros((0, 6), np.float32), gt_labels_3d=np.zeros((0,), np.int64)))
    data_info = get_data_info(ds, 0)

    # ---- MultiViewPipeline.__call__ ----------------------------------------------------------------------------
    get_dtu_raydir = cut("mmdet3d/datasets/pipelines/data_augment_utils.py", "get_dtu_raydir", glb=dict(np=np))
    mmcv = types.SimpleNamespace(imdenormalize=imdenormalize)
    call = cut("mmdet3d/datasets/pipelines/multi_view.py", "__call__", cls="MultiViewPipeline",
               glb=dict(np=np, mmcv=mmcv, get_dtu_raydir=get_dtu_raydir))
    table = {p: imnormalize(frames[i], mean, std) for i, p in enumerate(osp.join("data/scannet", q) for q in info["img_paths"])}

    def transforms(res):
        img = table[res["img_info"]["filename"]]
        return dict(img=img, ori_shape=(ori_hw[0], ori_hw[1], 3), img_shape=(hw[0], hw[1], 3), pad_shape=(hw[0], hw[1], 3),
                    img_prefix=res["img_prefix"], img_info=res["img_info"])

    out = {}
    for tag, loading, n_images, n_target, seed in (("random", "random", 9, 3, 11), ("seq", "sequence", 4, 1, 0)):
        pipe = types.SimpleNamespace(transforms=transforms, n_images=n_images, mean=mean, std=std, margin=margin, depth_range=[0.5, 5.5],
                                     loading=loading, sample_freq=3, nerf_target_views=n_target)
        import copy
        res = copy.deepcopy(data_info)
        np.random.seed(seed)
        res = call(pipe, res)
        fmt = cut("mmdet3d/datasets/pipelines/formating.py", "__call__", cls="DefaultFormatBundle",
                  glb=dict(np=np, DC=DC, to_tensor=lambda a: torch.from_numpy(a) if isinstance(a, np.ndarray) else torch.as_tensor(a),
                           BaseInstance3DBoxes=type("B", (), {}), BasePoints=type("P", (), {})))
        bundle = fmt(types.SimpleNamespace(), dict(res))
        out.update({
            f"{tag}__seed": np.int64(seed), f"{tag}__n_images": np.int64(n_images), f"{tag}__n_target": np.int64(n_target),
            f"{tag}__loading": np.array(loading),
            f"{tag}__extrinsic": np.stack(res["lidar2img"]["extrinsic"]),
            f"{tag}__img": bundle["img"].data.numpy(),
            f"{tag}__denorm_images": bundle["denorm_images"].data.numpy(),
            f"{tag}__raydirs": bundle["raydirs"].data.numpy(),
            f"{tag}__lightpos": bundle["lightpos"].data.numpy(),
            f"{tag}__gt_images": bundle["gt_images"].data.numpy(),
            f"{tag}__nerf_sizes": np.stack(res["nerf_sizes"]),
            f"{tag}__depth_range": np.asarray(res["depth_range"]),
            f"{tag}__c2w": np.stack(res["c2w"]),
        })
    # a direct get_dtu_raydir vector (with and without normalisation)
    px, py = np.meshgrid(np.arange(2, 11).astype(np.float32), np.arange(1, 8).astype(np.float32))
    pix = np.stack((px, py), axis=-1).astype(np.float32)
    rot = data_info["ray_info"]["camrotc2w"][5]
    k = intrinsic.astype(np.float32)
    out.update(raydir_pixels=pix, raydir_intrinsic=k, raydir_rot=rot, raydir_plain=get_dtu_raydir(pix, k, rot),
               raydir_normed=get_dtu_raydir(pix, k, rot, dir_norm=True))
    out.update(frames=frames, poses=np.stack(poses), axis_align=axis_align, intrinsic=intrinsic, mean=mean, std=std,
               margin=np.int64(margin), ori_hw=np.asarray(ori_hw), hw=np.asarray(hw),
               info_extrinsic=np.stack(data_info["lidar2img"]["extrinsic"]), info_intrinsic=data_info["lidar2img"]["intrinsic"],
               info_origin=data_info["lidar2img"]["origin"], info_c2w=np.stack(data_info["ray_info"]["c2w"]),
               info_camrotc2w=np.stack(data_info["ray_info"]["camrotc2w"]), info_lightpos=np.stack(data_info["ray_info"]["lightpos"]))
    np.savez_compressed(os.path.join(OUT, "pipeline_small.npz"), **out)
    for k_, v in sorted(out.items()):
        print(f"{k_:28s} {getattr(v, 'shape', ())} {getattr(v, 'dtype', type(v))}")


if __name__ == "__main__":
    main()
