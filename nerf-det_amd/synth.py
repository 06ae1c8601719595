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
