#!/usr/bin/env python3
"""BASELINE-size golden vectors from the REAL reference: ``nerfdet.extract_feat`` (mmdet3d/models/detectors/nerfdet.py:133-267)
at cfg2 (50 views, 256x60x80 features, 40x40x16 voxels) and cfg1 (10 views), run in the build container only:

    python tests/golden/make_golden_fullsize.py

The inputs are not stored: ``fullsize_inputs.py`` rebuilds them bit-identically from a seed (integer draws scaled by powers
of two) and the fixture keeps their SHA-256.  Stored per configuration:

  cnt            int8  (X,Y,Z)   view count of EVERY voxel (the reference's ``valids``)
  projection / rgb_projection    the reference's ``_compute_projection`` at stride 4 / 1 (what its ``torch.bmm`` multiplied)
  near_half      bool  (N,)      voxels with a view whose stride-4 or stride-1 pixel coordinate -- as the reference computed it --
                                 lies within 1e-3 px of a .5 rounding boundary (the only voxels where a last-ulp difference
                                 could pick a neighbouring pixel)
  sel / volume_sel / global_sel / alpha_sel       gated volume (all 256 channels), the 70 conditioning values the reference
                                 handed to ``query_density`` and its alpha at 4 096 (2 048) seeded voxels
  near_idx / volume_near / global_near / alpha_near    the same for every ``near_half`` voxel outside the sample, every 4th channel
  camera rig (intrinsic, extrinsic, origin, shapes)

``tests/test_fullsize_reference_gpu.py`` holds the HIP path to these with NO exclusion band: counts bit-exact on all voxels,
values <= 1e-4.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import fullsize_inputs as FI  # noqa: E402
from make_golden import load_reference, meta_arrays, npz  # noqa: E402
from oracle import nerfdet_oracle as O  # only for the shared synthetic camera rig  # noqa: E402


def float_param_shapes(module, prefix):
    return {prefix + k: tuple(v.shape) for k, v in module.state_dict().items() if v.is_floating_point()}


def make_fullsize_fixture(ref, name, cfg):
    t0 = time.time()
    n_v, c = cfg["n_views"], cfg["channels"]
    h, w = cfg["img_hw"]
    meta = O.ring_scene_meta(n_v, cfg["img_hw"])
    meta["box_type_3d"] = None
    det = ref.nerfdet.nerfdet(
        backbone=dict(type="backbone"), neck=dict(type="fpn", out_channels=c), neck_3d=dict(type="id"),
        bbox_head=dict(), n_voxels=list(cfg["n_voxels"]), voxel_size=list(cfg["voxel_size"]), aabb=None, near_far_range=[0.2, 8.0],
        N_samples=64, N_rand=2048, nerf_mode="image", squeeze_scale=4, nerf_density=True)
    shapes = {**float_param_shapes(det.mapping, "mapping."), **float_param_shapes(det.nerf_mlp, "nerf_mlp.")}
    wts = FI.weights(cfg, shapes)
    with torch.no_grad():
        for prefix, mod in (("mapping.", det.mapping), ("nerf_mlp.", det.nerf_mlp)):
            sd = mod.state_dict()
            for k in sd:
                if sd[k].is_floating_point():
                    sd[k].copy_(torch.from_numpy(wts[prefix + k]))
    det.eval()
    feats = torch.from_numpy(FI.features(cfg))
    denorm = torch.from_numpy(FI.denorm_images(cfg))
    det.backbone.payload = feats
    img = torch.zeros(1, n_v, 3, h, w)         # only its shape is read (the stand-in backbone hands back ``feats``)
    ray_batch = dict(ray_o=torch.zeros(1, 1, 4, 3), ray_d=torch.ones(1, 1, 4, 3), gt_rgb=torch.zeros(1, 1, 4, 3),
                     gt_depth=[], nerf_sizes=[torch.tensor([[2, 2, 3]])], denorm_images=denorm.unsqueeze(0))
    seen = {}
    real_query = det.nerf_mlp.query_density

    def spy(points, glob):
        seen["points"], seen["glob"] = points.clone(), glob.clone()
        seen["density"] = real_query(points, glob)
        return seen["density"]

    det.nerf_mlp.query_density = spy
    with torch.no_grad():
        x, valids, _, rgb_preds, _ = det.extract_feat(img, [meta], "test", None, ray_batch)
        proj = det._compute_projection(meta, 4, None)
        rgb_proj = det._compute_projection(meta, 1, None)
    assert rgb_preds == [None]
    volume = x[0].reshape(c, -1)                                    # (C, N), N = X*Y*Z with Z fastest
    cnt = valids[0][0]
    n = volume.shape[1]
    glob = seen["glob"]                                             # (N, 70)
    alpha = (1 - torch.exp(-seen["density"])).reshape(-1)

    # voxels near a rounding boundary, from the reference's own arithmetic (nerfdet.py:398-404)
    pts = seen["points"].t().contiguous()                          # (3, N)
    hom = torch.cat([pts, torch.ones(1, n)], 0).unsqueeze(0).expand(n_v, 4, n)
    near = torch.zeros(n, dtype=torch.bool)
    for p, ww, hh in ((proj, w // 4, h // 4), (rgb_proj, w, h)):
        p3 = torch.bmm(p, hom)
        u, v, d = p3[:, 0] / p3[:, 2], p3[:, 1] / p3[:, 2], p3[:, 2]
        close = ((u - torch.floor(u) - 0.5).abs() < FI.NEAR_TOL) | ((v - torch.floor(v) - 0.5).abs() < FI.NEAR_TOL) | (d.abs() < 1e-6)
        inside = (u > -1) & (u < ww) & (v > -1) & (v < hh)
        near |= (close & inside).any(0)
    sel = FI.sample_voxels(cfg)
    near_idx = np.setdiff1d(np.nonzero(near.numpy())[0], sel).astype(np.int32)
    step = FI.NEAR_CHANNEL_STEP
    assert int(cnt.max()) < 128
    npz(name,
        inputs_sha256=np.array(FI.checksum(feats.numpy(), denorm.numpy(), *[wts[k] for k in sorted(wts)])),
        weight_keys=np.array(sorted(wts)), n_voxels=np.array(cfg["n_voxels"]), voxel_size=np.array(cfg["voxel_size"], dtype=np.float32),
        cnt=cnt.to(torch.int8), projection=proj, rgb_projection=rgb_proj, near_half=near,
        sel=sel, volume_sel=volume[:, sel].t().contiguous(), global_sel=glob[sel], alpha_sel=alpha[sel],
        near_idx=near_idx, volume_near=volume[::step, near_idx].t().contiguous(), global_near=glob[near_idx], alpha_near=alpha[near_idx],
        **meta_arrays(meta))
    print(f"  {name}: {n} voxels, seen {float((cnt > 0).float().mean()):.3f}, max count {int(cnt.max())}, near-boundary {int(near.sum())}, "
          f"alpha mean {float(alpha.mean()):.3f} [{float(alpha.min()):.3f}, {float(alpha.max()):.3f}], |volume| max {float(volume.abs().max()):.3f}, "
          f"{time.time() - t0:.1f} s")


def main():
    ref = load_reference()
    for name, cfg in FI.CONFIGS.items():
        make_fullsize_fixture(ref, f"fullsize_{name}", cfg)


if __name__ == "__main__":
    main()
