#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference code.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

It imports the reference's hot-path Python *from where it lies* (nothing is copied
into this repo) by file path, with ``sys.modules`` stand-ins for the third-party
packages the reference needs at import time but that are absent here (mmcv, mmdet,
numba, the compiled iou3d extension) -- the recipe recorded in SURVEY.md section 8c.
The stand-ins carry no arithmetic of the path except the documented mmcv helpers
(``Scale`` = learnable scalar multiply, ``multi_apply`` = map+zip).

Outputs are data only: seeded inputs and the reference's outputs for them.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(OUT, "..", ".."))
from oracle import nerfdet_oracle as O  # only for the shared synthetic camera rig  # noqa: E402


# --------------------------------------------------------------------------- #
# reference loader
# --------------------------------------------------------------------------- #
def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


class _Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


class _BaseDetector(nn.Module):
    def init_weights(self, pretrained=None):
        pass


class _Feed(nn.Module):
    """stand-in for ResNet / FPN / neck_3d: hands back what it is told to."""

    def __init__(self, mode):
        super().__init__()
        self.mode = mode
        self.payload = None

    def init_weights(self, *a, **k):
        pass

    def forward(self, x):
        if self.mode == "backbone":
            return (self.payload,)
        if self.mode == "fpn":
            return [x[0]]
        return x  # neck_3d: identity


class _DummyHead(nn.Module):
    def init_weights(self):
        pass


def load_reference():
    for p in ["mmdet", "mmdet.models", "mmdet.models.detectors", "mmdet.models.builder", "mmdet.core",
              "mmcv", "mmcv.cnn", "mmcv.runner", "numba",
              "mmdet3d", "mmdet3d.core", "mmdet3d.core.bbox", "mmdet3d.core.bbox.structures",
              "mmdet3d.core.post_processing", "mmdet3d.ops", "mmdet3d.ops.iou3d", "mmdet3d.ops.iou3d.iou3d_utils",
              "mmdet3d.models", "mmdet3d.models.detectors", "mmdet3d.models.model_utils",
              "mmdet3d.models.model_utils.save_rendered_img"]:
        _pkg(p)
    sm = sys.modules
    sm["mmdet.models"].DETECTORS = _Registry()
    sm["mmdet.models"].NECKS = _Registry()
    sm["mmdet.models.builder"].HEADS = _Registry()
    sm["mmdet.models.builder"].build_loss = lambda cfg: None
    sm["mmdet.models"].build_backbone = lambda cfg: _Feed("backbone")
    sm["mmdet.models"].build_neck = lambda cfg: _Feed(cfg["type"])
    sm["mmdet.models"].build_head = lambda cfg: _DummyHead()
    sm["mmdet.models.detectors"].BaseDetector = _BaseDetector
    sm["mmdet.core"].multi_apply = lambda f, *a: tuple(map(list, zip(*map(f, *a))))
    sm["mmdet.core"].reduce_mean = lambda t: t
    sm["mmcv.cnn"].Scale = _Scale
    sm["mmcv.cnn"].normal_init = lambda *a, **k: None
    sm["mmcv.cnn"].bias_init_with_prob = lambda p: float(-np.log((1 - p) / p))
    sm["mmcv.runner"].auto_fp16 = lambda *a, **k: (lambda f: f)
    sm["numba"].jit = lambda *a, **k: (lambda f: f)
    sm["mmdet3d.ops.iou3d.iou3d_utils"].nms_gpu = None
    sm["mmdet3d.ops.iou3d.iou3d_utils"].nms_normal_gpu = None
    sm["mmdet3d.core"].bbox3d2result = lambda b, s, l: dict(boxes_3d=b, scores_3d=s, labels_3d=l)
    sm["mmdet3d.core.bbox.structures"].rotation_3d_in_axis = None
    sm["mmdet3d.models.model_utils.save_rendered_img"].save_rendered_img = None

    ref = types.SimpleNamespace()
    mu = "mmdet3d/models/model_utils/"
    ref.projection = _load("mmdet3d.models.model_utils.projection", mu + "projection.py")
    ref.nerf_mlp = _load("mmdet3d.models.model_utils.nerf_mlp", mu + "nerf_mlp.py")
    ref.render_ray = _load("mmdet3d.models.model_utils.render_ray", mu + "render_ray.py")
    ref.nerfdet = _load("mmdet3d.models.detectors.nerfdet", "mmdet3d/models/detectors/nerfdet.py")
    ref.nms = _load("mmdet3d.core.post_processing.box3d_nms", "mmdet3d/core/post_processing/box3d_nms.py")
    sm["mmdet3d.core.post_processing"].aligned_3d_nms = ref.nms.aligned_3d_nms
    sm["mmdet3d.core.post_processing"].box3d_multiclass_nms = ref.nms.box3d_multiclass_nms
    ref.neck = _load("ref_imvoxel_neck", "mmdet3d/models/necks/imvoxelnet.py")
    ref.head = _load("ref_imvoxel_head_v2", "mmdet3d/models/dense_heads/imvoxel_head_v2.py")
    return ref


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def sd_arrays(prefix, module):
    return {prefix + k: v for k, v in module.state_dict().items()}


# --------------------------------------------------------------------------- #
# fixtures
# --------------------------------------------------------------------------- #
def meta_arrays(meta):
    return dict(intrinsic=meta["lidar2img"]["intrinsic"], extrinsic=np.stack(meta["lidar2img"]["extrinsic"]),
                origin=meta["lidar2img"]["origin"], ori_shape=np.array(meta["ori_shape"]),
                img_shape=np.array(meta["img_shape"]))


def make_volume_fixture(ref, name, seed, n_v, c, img_hw, n_voxels, voxel_size, mlp_width=256, crop_shape=None):
    """A1-A6: real ``nerfdet.extract_feat`` (test mode) + its module-level functions."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    h, w = img_hw
    meta = O.ring_scene_meta(n_v, img_hw)
    if crop_shape is not None:  # real ScanNet frames resize to 239x320 before padding
        meta["img_shape"] = crop_shape
    meta["box_type_3d"] = None
    det = ref.nerfdet.nerfdet(
        backbone=dict(type="backbone"), neck=dict(type="fpn", out_channels=c), neck_3d=dict(type="id"),
        bbox_head=dict(), n_voxels=n_voxels, voxel_size=voxel_size, aabb=None, near_far_range=[0.2, 8.0],
        N_samples=8, N_rand=16, nerf_mode="image", squeeze_scale=4, nerf_density=True)
    if mlp_width != 256:
        det.nerf_mlp = ref.nerf_mlp.VanillaNeRFRadianceField(
            net_depth=4, net_width=mlp_width, skip_layer=3, feature_dim=c // 4 + 6,
            net_depth_condition=1, net_width_condition=mlp_width // 2)
    # non-zero biases so the "bias at unseen views" convention (nerfdet.py:233) is exercised
    with torch.no_grad():
        det.mapping[0].bias.normal_(0, 0.5)
        for p in det.nerf_mlp.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    det.eval()
    feats = torch.randn(n_v, c, h // 4, w // 4)
    img = torch.randn(1, n_v, 3, h, w)
    denorm = torch.rand(1, n_v, 3, h, w)
    det.backbone.payload = feats
    ray_batch = dict(ray_o=torch.zeros(1, 1, 4, 3), ray_d=torch.ones(1, 1, 4, 3), gt_rgb=torch.zeros(1, 1, 4, 3),
                     gt_depth=[], nerf_sizes=[torch.tensor([[2, 2, 3]])], denorm_images=denorm)
    with torch.no_grad():
        x, valids, _, rgb_preds, _ = det.extract_feat(img, [meta], "test", None, ray_batch)
        proj = det._compute_projection(meta, 4, None)
        rgb_proj = det._compute_projection(meta, 1, None)
        pts = ref.nerfdet.get_points(torch.tensor(n_voxels), torch.tensor(voxel_size), torch.tensor(meta["lidar2img"]["origin"]))
        hh, ww = meta["img_shape"][0] // 4, meta["img_shape"][1] // 4
        vol, valid = ref.nerfdet.backproject(feats[:, :, :hh, :ww], pts, proj, None, voxel_size)
        rgb_vol, rgb_valid = ref.nerfdet.backproject(denorm[0][:, :, :meta["img_shape"][0], :meta["img_shape"][1]], pts, rgb_proj, None, voxel_size)
    assert rgb_preds == [None]
    arrays = dict(features=feats, denorm_images=denorm[0], n_voxels=np.array(n_voxels), voxel_size=np.array(voxel_size, dtype=np.float32),
                  out_volume=x[0], out_valid=valids[0], projection=proj, rgb_projection=rgb_proj, points=pts,
                  bp_valid=valid, rgb_bp_valid=rgb_valid,
                  # per-view volumes are big; keep the view-sum (order-exact in the oracle test) and two full views
                  bp_volume_v0=vol[0], bp_volume_vlast=vol[-1], bp_volume_sum=vol.sum(0),
                  rgb_bp_volume_sum=rgb_vol.sum(0),
                  **meta_arrays(meta))
    arrays.update(sd_arrays("mapping.", det.mapping))
    arrays.update(sd_arrays("nerf_mlp.", det.nerf_mlp))
    npz(name, **arrays)


def make_ray_fixture(ref, name, seed, n_v, d, img_hw, n_rays, n_samples, width):
    """A7-A12: Projector.compute, compute_mask_points, sampling, MLP, compositing, render_rays(_func)."""
    torch.manual_seed(seed)
    h, w = img_hw
    meta = O.ring_scene_meta(n_v, img_hw)
    mlp = ref.nerf_mlp.VanillaNeRFRadianceField(net_depth=4, net_width=width, skip_layer=3, feature_dim=2 * (d + 3),
                                                net_depth_condition=1, net_width_condition=width // 2)
    with torch.no_grad():
        for p in mlp.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    mlp.eval()
    feat2d = torch.randn(n_v, d, h // 4, w // 4)
    img = torch.rand(n_v, 3, h, w)
    # rays from a virtual camera on the ring radius, pointing roughly inwards with spread
    ang = torch.rand(n_rays) * 2 * np.pi
    ray_o = torch.stack([2.0 * torch.cos(ang), 2.0 * torch.sin(ang), 1.0 + 0.3 * torch.rand(n_rays)], -1)
    ray_d = -ray_o / ray_o.norm(dim=-1, keepdim=True) + 0.35 * torch.randn(n_rays, 3)
    proj = ref.projection.Projector()
    with torch.no_grad():
        # A9 deterministic + jittered (replay the rand_like draw)
        pts_det, z_det = ref.render_ray.sample_along_camera_ray(ray_o, ray_d, [0.2, 8.0], n_samples, det=True)
        torch.manual_seed(seed + 100)
        t_rand = torch.rand_like(z_det)
        torch.manual_seed(seed + 100)
        pts_rnd, z_rnd = ref.render_ray.sample_along_camera_ray(ray_o, ray_d, [0.2, 8.0], n_samples, det=False)
        # A1 twin, A7, A8
        cams = ref.render_ray._compute_projection(meta)
        imgs_nhwc = img.permute(0, 2, 3, 1).unsqueeze(0)
        rgb_feat, mask = proj.compute(pts_rnd, imgs_nhwc, cams, feat2d, grid_sample=True)
        mean, var = ref.render_ray.compute_mask_points(rgb_feat, mask)
        glob = torch.cat([mean, var], dim=-1).squeeze(2)
        # A10
        rgb_pts, sigma_pts = mlp(pts_rnd, ray_d, glob)
        dens = mlp.query_density(pts_rnd.reshape(-1, 3), glob.reshape(-1, glob.shape[-1]))
        # A11
        raw = torch.cat([rgb_pts, sigma_pts], -1)
        pixel_mask = mask[..., 0].sum(dim=2) > 1
        comp = ref.render_ray.raw2outputs(raw, z_rnd, pixel_mask)
        raw_rand = torch.cat([torch.rand(n_rays, n_samples, 3), 3 * torch.rand(n_rays, n_samples, 1) ** 3], -1)
        comp2 = ref.render_ray.raw2outputs(raw_rand, z_rnd, pixel_mask, white_bkgd=True)
        # A12 end-to-end, deterministic sampling
        ret = ref.render_ray.render_rays_func(ray_o, ray_d, None, None, feat2d, img, None, [0.2, 8.0], n_samples,
                                              n_rays, mlp, meta, proj, "image", det=True)
    arrays = dict(features_2d=feat2d, img=img, ray_o=ray_o, ray_d=ray_d, n_samples=np.array(n_samples),
                  pts_det=pts_det, z_det=z_det, t_rand=t_rand, pts_rnd=pts_rnd, z_rnd=z_rnd,
                  cameras=cams, rgb_feat=rgb_feat, mask=mask, mean=mean, var=var,
                  rgb_pts=rgb_pts, sigma_pts=sigma_pts, density_q=dens,
                  comp_rgb=comp["rgb"], comp_depth=comp["depth"], comp_weights=comp["weights"],
                  comp_mask=comp["mask"], comp_alpha=comp["alpha"], comp_T=comp["transparency"],
                  raw_rand=raw_rand, comp2_rgb=comp2["rgb"], comp2_depth=comp2["depth"],
                  func_rgb=ret["outputs_coarse"]["rgb"], func_depth=ret["outputs_coarse"]["depth"],
                  func_mask=ret["outputs_coarse"]["mask"], func_weights=ret["outputs_coarse"]["weights"],
                  func_sigma=ret["sigma"], **meta_arrays(meta))
    arrays.update(sd_arrays("nerf_mlp.", mlp))
    npz(name, **arrays)


def make_ray_select_fixture(ref, name):
    """A12 ray selection: the module-global RandomState(234) of render_ray.py:20 on its FIRST draw,
    plus the train-mode losses of nerfdet.py:296-321 through the real ``render_rays``."""
    torch.manual_seed(7)
    n_v, d, h, w, width = 5, 8, 48, 64, 32
    meta = O.ring_scene_meta(n_v, (h, w))
    mlp = ref.nerf_mlp.VanillaNeRFRadianceField(net_depth=4, net_width=width, skip_layer=3, feature_dim=2 * (d + 3),
                                                net_depth_condition=1, net_width_condition=width // 2)
    mlp.eval()
    feat2d = torch.randn(n_v, d, h // 4, w // 4)
    img = torch.rand(n_v, 3, h, w)
    t_views, hw = 2, 120
    ang = torch.rand(1, t_views, hw) * 2 * np.pi
    ray_o = torch.stack([2.0 * torch.cos(ang), 2.0 * torch.sin(ang), 1.0 + 0 * ang], -1)
    ray_d = -ray_o / ray_o.norm(dim=-1, keepdim=True) + 0.3 * torch.randn(1, t_views, hw, 3)
    gt_rgb = torch.rand(1, t_views, hw, 3)
    gt_depth = torch.rand(1, t_views, 10, 12) * 5 + 0.5
    gt_depth[0, 0, 0, :5] = 0.0  # some rays without depth are dropped
    rb = dict(ray_o=ray_o, ray_d=ray_d, gt_rgb=gt_rgb, gt_depth=gt_depth, nerf_sizes=[torch.tensor([[10, 12, 3]])])
    ref.render_ray.rng = np.random.RandomState(234)  # same state as a fresh import (render_ray.py:20)
    torch.manual_seed(11)
    n_samples, n_rand = 12, 48
    t_rand = torch.rand(n_rand, n_samples)
    torch.manual_seed(11)
    with torch.no_grad():
        ret = ref.render_ray.render_rays(rb, None, None, feat2d, img, None, [0.2, 8.0], n_samples, n_rand, mlp, meta,
                                         ref.projection.Projector(), "image", is_train=True)
        det = types.SimpleNamespace(use_nerf_mask=True)
        l_nvs = ref.nerfdet.nerfdet.nvs_loss_func(det, [ret])["loss_nvs"]
        l_depth = ref.nerfdet.nerfdet.depth_loss_func(det, [ret])["loss_depth"]
    arrays = dict(features_2d=feat2d, img=img, ray_o=ray_o, ray_d=ray_d, gt_rgb=gt_rgb, gt_depth=gt_depth,
                  n_samples=np.array(n_samples), n_rand=np.array(n_rand), t_rand=t_rand,
                  sel_gt_rgb=ret["gt_rgb"], sel_gt_depth=ret["gt_depth"],
                  rgb=ret["outputs_coarse"]["rgb"], depth=ret["outputs_coarse"]["depth"],
                  mask=ret["outputs_coarse"]["mask"], loss_nvs=l_nvs, loss_depth=l_depth, **meta_arrays(meta))
    arrays.update(sd_arrays("nerf_mlp.", mlp))
    npz(name, **arrays)


def make_render_testing_fixture(ref, name):
    """f-4: the novel-view path, render_rays(render_testing=True) (render_ray.py:452-517): every ray of the target views in chunks
    of N_rand, deterministic sampling; plus compute_psnr of save_rendered_img.py:10-19 restated inline (that module needs cv2 /
    skimage / imageio at import time; PSNR is -10 log10(mse))."""
    torch.manual_seed(21)
    n_v, d, h, w, width = 6, 8, 48, 64, 32
    meta = O.ring_scene_meta(n_v, (h, w))
    mlp = ref.nerf_mlp.VanillaNeRFRadianceField(net_depth=4, net_width=width, skip_layer=3, feature_dim=2 * (d + 3),
                                                net_depth_condition=1, net_width_condition=width // 2)
    with torch.no_grad():
        for p in mlp.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    mlp.eval()
    feat2d = torch.randn(n_v, d, h // 4, w // 4)
    img = torch.rand(n_v, 3, h, w)
    t_views, rh, rw = 2, 7, 9                       # 63 rays per view, chunks of 16: ragged last chunk, chunks straddle views
    ang = torch.rand(1, t_views, 1) * 2 * np.pi
    cam = torch.cat([2.0 * torch.cos(ang), 2.0 * torch.sin(ang), 1.0 + 0 * ang], -1)
    ray_o = cam.unsqueeze(2).expand(1, t_views, rh * rw, 3).contiguous()
    ray_d = -ray_o / ray_o.norm(dim=-1, keepdim=True) + 0.3 * torch.randn(1, t_views, rh * rw, 3)
    gt_rgb = torch.rand(1, t_views, rh * rw, 3)
    gt_depth = torch.rand(1, t_views, rh, rw) * 5 + 0.5
    rb = dict(ray_o=ray_o, ray_d=ray_d, gt_rgb=gt_rgb, gt_depth=gt_depth, nerf_sizes=[torch.tensor([[rh, rw, 3]])])
    import contextlib
    import io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):          # the reference prints gt_rgb.shape
        ret = ref.render_ray.render_rays(rb, None, None, feat2d, img, None, [0.2, 8.0], 12, 16, mlp, meta, ref.projection.Projector(), "image",
                                         is_train=False, render_testing=True)
        ret_nodepth = ref.render_ray.render_rays(dict(rb, gt_depth=[]), None, None, feat2d, img, None, [0.2, 8.0], 12, 16, mlp, meta,
                                                 ref.projection.Projector(), "image", is_train=False, render_testing=True)
    rgb, gt = ret["outputs_coarse"]["rgb"], ret["gt_rgb"]
    psnr = torch.stack([-10.0 * torch.log(((rgb[v] - gt[v]) ** 2).mean()) / np.log(10.0) for v in range(t_views)])
    assert ret_nodepth["gt_depth"] is None and torch.equal(ret_nodepth["outputs_coarse"]["rgb"], rgb)
    arrays = dict(features_2d=feat2d, img=img, ray_o=ray_o, ray_d=ray_d, gt_rgb=gt_rgb, gt_depth=gt_depth, n_samples=np.array(12), n_rand=np.array(16),
                  nerf_size=np.array([rh, rw, 3]), out_rgb=rgb, out_depth=ret["outputs_coarse"]["depth"], out_gt_rgb=gt, out_gt_depth=ret["gt_depth"],
                  psnr=psnr, **meta_arrays(meta))
    arrays.update(sd_arrays("nerf_mlp.", mlp))
    npz(name, **arrays)


def make_head_fixture(ref, name, seed, c_in, c_mid, grid):
    """A13-A15: FastIndoorImVoxelNeck, ScanNetImVoxelHeadV2 forward + get_bboxes (+NMS)."""
    torch.manual_seed(seed)
    neck = ref.neck.FastIndoorImVoxelNeck(in_channels=c_in, n_blocks=[1, 1, 1], out_channels=c_mid)
    with torch.no_grad():
        for m in neck.modules():
            if isinstance(m, nn.BatchNorm3d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    x = torch.randn(1, c_in, *grid)
    neck.eval()
    with torch.no_grad():
        outs_eval = neck(x)
    neck.train()
    with torch.no_grad():
        sd_before = {k: v.clone() for k, v in neck.state_dict().items()}
        outs_train = neck(x)
    neck.load_state_dict(sd_before)

    class _Cfg(dict):
        __getattr__ = dict.__getitem__
    head = ref.head.ScanNetImVoxelHeadV2(n_classes=18, n_channels=c_mid, n_reg_outs=6, n_scales=3, limit=27,
                                         centerness_topk=18, test_cfg=_Cfg(nms_pre=120, iou_thr=0.25, score_thr=0.01))
    head.voxel_size = (0.16, 0.16, 0.2)
    with torch.no_grad():  # a "trained-like" head: default init would score < score_thr everywhere
        head.cls_conv.weight.normal_(0, 0.08)
        head.cls_conv.bias.fill_(-1.0)
        head.centerness_conv.weight.normal_(0, 0.05)
        head.reg_conv.weight.normal_(0, 0.03)
        for i, s in enumerate(head.scales):
            s.scale.fill_(1.0 + 0.25 * i)
    valid = (torch.rand(1, 1, *grid) < 0.7).float() * torch.randint(1, 6, (1, 1, *grid)).float()

    class _Boxes:  # 3-line stand-in for DepthInstance3DBoxes (result wrapper only)
        def __init__(self, t, **k):
            self.tensor = t
    meta = dict(lidar2img=dict(origin=np.array([0.0, 0.0, 0.5], dtype=np.float32)), box_type_3d=_Boxes)
    with torch.no_grad():
        ctr, reg, cls = head(outs_eval)
        (boxes, scores, labels), = head.get_bboxes(ctr, reg, cls, valid, [meta])
    arrays = dict(x=x, valid=valid, voxel_size=np.array(head.voxel_size, dtype=np.float32), origin=meta["lidar2img"]["origin"],
                  nms_pre=np.array(120), iou_thr=np.array(0.25), score_thr=np.array(0.01),
                  det_boxes=boxes.tensor, det_scores=scores, det_labels=labels)
    for i in range(3):
        arrays[f"neck_eval_{i}"] = outs_eval[i]
        arrays[f"neck_train_{i}"] = outs_train[i]
        arrays[f"ctr_{i}"], arrays[f"reg_{i}"], arrays[f"cls_{i}"] = ctr[i], reg[i], cls[i]
    arrays.update(sd_arrays("neck_3d.", neck))
    arrays.update(sd_arrays("bbox_head.", head))
    npz(name, **arrays)


def make_nms_fixture(ref, name):
    """A15 on random clustered boxes (no score ties: argsort on ties is not stable)."""
    g = torch.Generator().manual_seed(3)
    n = 400
    ctr = torch.rand(n, 3, generator=g) * torch.tensor([6.0, 6.0, 2.5])
    ctr[n // 2:] = ctr[: n // 2] + 0.08 * torch.randn(n // 2, 3, generator=g)  # near-duplicates get suppressed
    size = 0.3 + 1.2 * torch.rand(n, 3, generator=g)
    boxes = torch.cat([ctr - size / 2, ctr + size / 2], 1)
    scores = torch.rand(n, generator=g)
    assert scores.unique().numel() == n
    classes = torch.randint(0, 4, (n,), generator=g)
    out = {}
    for thr in (0.25, 0.5):
        out[f"pick_{int(thr * 100)}"] = ref.nms.aligned_3d_nms(boxes, scores, classes, thr)
    # degenerate boxes: zero volume pairs give 0/0 = NaN -> suppressed (box3d_nms.py:131-135)
    deg = boxes[:40].clone()
    deg[5:15, 3:] = deg[5:15, :3]
    out["deg_boxes"] = deg
    out["deg_pick"] = ref.nms.aligned_3d_nms(deg, scores[:40], torch.zeros(40, dtype=torch.long), 0.25)
    npz(name, boxes=boxes, scores=scores, classes=classes, **out)


def main():
    ref = load_reference()
    make_volume_fixture(ref, "volume_small_s0", 0, n_v=6, c=16, img_hw=(60, 80), n_voxels=(8, 8, 4),
                        voxel_size=(0.8, 0.8, 0.8), mlp_width=32)
    make_volume_fixture(ref, "volume_small_s1", 1, n_v=7, c=32, img_hw=(60, 80), n_voxels=(10, 6, 5),
                        voxel_size=(0.6, 0.9, 0.6), mlp_width=32, crop_shape=(59, 80, 3))
    make_volume_fixture(ref, "volume_medium_s2", 2, n_v=10, c=32, img_hw=(120, 160), n_voxels=(20, 20, 8),
                        voxel_size=(0.32, 0.32, 0.4), mlp_width=64)
    make_ray_fixture(ref, "rays_small_s0", 0, n_v=6, d=8, img_hw=(60, 80), n_rays=32, n_samples=16, width=32)
    make_ray_fixture(ref, "rays_small_s1", 1, n_v=9, d=32, img_hw=(60, 80), n_rays=24, n_samples=64, width=64)
    make_ray_select_fixture(ref, "rays_select")
    make_render_testing_fixture(ref, "render_testing")
    make_head_fixture(ref, "head_small_s0", 0, c_in=8, c_mid=8, grid=(8, 8, 4))
    make_nms_fixture(ref, "nms_random")


if __name__ == "__main__":
    main()
