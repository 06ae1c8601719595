"""CPU oracle for the NeRF-Det volumetric hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch-CPU restatement of the
reference's algorithm for the hot path.  It
exists to *check* the HIP kernels; nothing under ``nerf-det_amd/`` may import
it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` use it.

Pinning: every function below is compared against outputs of the reference's
own Python (imported from /root/reference in the build container by
``tests/golden/make_golden.py``) through the fixtures committed under
``tests/golden/*.npz`` -- see ``tests/test_oracle_golden.py``.  The only piece
pinned by one of the reference's own tests is ``aligned_3d_nms``
(reference ``tests/test_nms.py:5-58``), which is replayed as well.

Each function cites the reference file:line it follows.  The code keeps the
reference's *order of floating point operations* (materialised per-view volume,
un-fused mul/add, the unmasked variance convention) because the GPU kernels are
judged against exactly that arithmetic.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# The reference leaves three small products to the host's BLAS (``intrinsic @ extrinsic[:3]``, nerfdet.py:377; ``torch.bmm(projection,
# points)``, nerfdet.py:398), whose last bit depends on the kernel that BLAS picks for the CPU it runs on -- and ``.round()`` turns a last
# bit into a neighbouring pixel (3 of 25 600 voxels on the GPU box's CPU).  With PINNED_ARITHMETIC the oracle evaluates them
# in the order MKL uses in the build container -- a k-ordered chain of fp32 FMAs -- emulated in integer-exact numpy, so that it returns the
# REFERENCE-AS-RUN-IN-THE-BUILD-CONTAINER on any host: that is what the golden fixtures hold (tests/golden/fullsize_cfg*.npz were written
# by the real reference there) and tests/test_oracle_golden.py checks the emulation against ``torch.bmm`` itself where it runs.
# bench.py's cpu_baseline leg switches it off (the baseline times the reference's own library calls).
PINNED_ARITHMETIC = True


def fma_chain_matmul(a, b) -> np.ndarray:
    """(...,m,K) @ (...,K,n) in fp32 as ``fma(a_K-1, b_K-1, ... fma(a_1, b_1, a_0 * b_0))``, every step rounded ONCE.

    A product of two fp32 numbers is exact in fp64; the sum with the fp32 accumulator is made exact by TwoSum and rounded to odd
    in fp64, after which the single rounding to fp32 equals the hardware FMA's (53 >= 24 + 2 bits)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    kdim = a.shape[-1]
    assert b.shape[-2] == kdim
    acc = (a[..., :, 0, None].astype(np.float64) * b[..., None, 0, :].astype(np.float64)).astype(np.float32)
    for k in range(1, kdim):
        p = a[..., :, k, None].astype(np.float64) * b[..., None, k, :].astype(np.float64)
        c = acc.astype(np.float64)
        s = p + c
        t = s - p
        err = (p - (s - t)) + (c - t)                       # p + c == s + err exactly
        bits = s.view(np.int64)
        fix = (err != 0) & ((bits & 1) == 0)
        away = (err > 0) == (s > 0)
        bits = np.where(fix, np.where(away, bits + 1, bits - 1), bits)
        acc = bits.view(np.float64).astype(np.float32)
    return acc


# --------------------------------------------------------------------------- #
# A1  camera matrices
# --------------------------------------------------------------------------- #
def compute_projection(img_meta: dict, stride: int) -> Tensor:
    """Per-view 3x4 pixel projection ``K' @ E[:3]``.

    Follows mmdet3d/models/detectors/nerfdet.py:363-378 (angles=None branch):
    the 3x3 intrinsic has its first two rows divided by
    ``ori_h / (img_h / stride)`` (the *height* ratio for both axes).
    Returns (n_views, 3, 4) float32.
    """
    k = torch.tensor(np.asarray(img_meta["lidar2img"]["intrinsic"])[:3, :3])
    ratio = img_meta["ori_shape"][0] / (img_meta["img_shape"][0] / stride)
    k[:2] /= ratio
    if PINNED_ARITHMETIC:
        ext = np.stack([np.asarray(e, dtype=np.float32)[:3] for e in img_meta["lidar2img"]["extrinsic"]])
        return torch.from_numpy(fma_chain_matmul(k.numpy()[None], ext))
    mats = [k @ torch.tensor(np.asarray(e))[:3] for e in img_meta["lidar2img"]["extrinsic"]]
    return torch.stack(mats)


def compute_ray_cameras(img_meta: dict) -> Tensor:
    """Packed per-view camera rows ``[h, w | K(4x4) | E(4x4)]`` = 34 floats.

    Follows mmdet3d/models/model_utils/render_ray.py:48-69: stride-1 variant,
    ratio = ``ori_h / img_h``.  Returns (1, n_views, 34) float32.
    """
    ext = img_meta["lidar2img"]["extrinsic"]
    n = len(ext)
    k = torch.tensor(np.asarray(img_meta["lidar2img"]["intrinsic"])[:4, :4])
    k[:2] /= img_meta["ori_shape"][0] / img_meta["img_shape"][0]
    size = torch.tensor([float(img_meta["img_shape"][0]), float(img_meta["img_shape"][1])])
    rows = []
    for v in range(n):
        e = torch.tensor(np.asarray(ext[v]), dtype=torch.float32)
        rows.append(torch.cat([size, k.reshape(16), e.reshape(16)]))
    return torch.stack(rows).unsqueeze(0)


# --------------------------------------------------------------------------- #
# A2  voxel lattice
# --------------------------------------------------------------------------- #
def get_points(n_voxels: Sequence[int], voxel_size: Sequence[float], origin: Sequence[float]) -> Tensor:
    """Voxel lower-corner lattice, (3, X, Y, Z) float32, Z fastest.

    Follows nerfdet.py:380-390: ``idx * voxel_size + (origin - n/2 * voxel_size)``
    as two separate fp32 roundings (no fused multiply-add).
    """
    n = torch.as_tensor(n_voxels)
    vs = torch.as_tensor(voxel_size, dtype=torch.float32)
    org = torch.as_tensor(origin, dtype=torch.float32)
    axes = [torch.arange(int(n[i])) for i in range(3)]
    idx = torch.stack(torch.meshgrid(axes, indexing="ij"))
    shifted = org - n / 2.0 * vs
    return idx * vs.view(3, 1, 1, 1) + shifted.view(3, 1, 1, 1)


# --------------------------------------------------------------------------- #
# A3  nearest-neighbour back-projection into a per-view volume
# --------------------------------------------------------------------------- #
def project_voxels(points: Tensor, projection: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """Homogeneous projection of every voxel into every view.

    nerfdet.py:396-402.  Returns float pixel coordinates before rounding
    ``(u/w, v/w)`` and ``w`` -- split out so tests can flag voxel-views whose
    coordinate sits on a rounding boundary.
    """
    n_v = projection.shape[0]
    p = points.reshape(1, 3, -1).expand(n_v, 3, -1)
    p = torch.cat((p, torch.ones_like(p[:, :1])), dim=1)
    if PINNED_ARITHMETIC and projection.dtype == torch.float32 and not projection.requires_grad:
        hom = torch.cat((points.reshape(3, -1), torch.ones_like(points.reshape(3, -1)[:1])), dim=0)
        uvw = torch.from_numpy(fma_chain_matmul(projection.numpy(), hom.numpy()[None]))
    else:
        uvw = torch.bmm(projection, p)
    return uvw[:, 0] / uvw[:, 2], uvw[:, 1] / uvw[:, 2], uvw[:, 2]


def backproject(features: Tensor, points: Tensor, projection: Tensor,
                depth: Optional[Tensor] = None, voxel_size=None) -> Tuple[Tensor, Tensor]:
    """features (n_v,C,h,w) -> volume (n_v,C,X,Y,Z), valid (n_v,1,X,Y,Z) bool.

    Follows nerfdet.py:393-420: round-half-even pixel index, validity =
    in-image and in front of the camera, masked gather per view into a
    zero-initialised materialised volume.  ``depth`` gating (nerfdet.py:405-411)
    is dead under every shipped config and is not restated; passing it raises.
    """
    if depth is not None:
        raise NotImplementedError("depth-gated backproject is dead code in the reference configs")
    n_v, c, h, w = features.shape
    gx, gy, gz = points.shape[-3:]
    fu, fv, fw = project_voxels(points, projection)
    x = fu.round().long()
    y = fv.round().long()
    valid = (x >= 0) & (y >= 0) & (x < w) & (y < h) & (fw > 0)
    volume = torch.zeros((n_v, c, gx * gy * gz), dtype=features.dtype)
    for i in range(n_v):
        m = valid[i]
        volume[i, :, m] = features[i, :, y[i, m], x[i, m]]
    return volume.view(n_v, c, gx, gy, gz), valid.view(n_v, 1, gx, gy, gz)


# --------------------------------------------------------------------------- #
# A4  view aggregation
# --------------------------------------------------------------------------- #
def aggregate_views(volume: Tensor, valid: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """(n_v,C,X,Y,Z),(n_v,1,X,Y,Z) -> mean (C,X,Y,Z), count (1,X,Y,Z) int64, cov (C,X,Y,Z).

    nerfdet.py:171-181.  Note the variance sums over *all* views, including the
    zero rows of views that do not see the voxel.
    """
    total = volume.sum(dim=0)
    cnt = valid.sum(dim=0)
    mean = total / (cnt + 1e-8)
    mean[:, cnt[0] == 0] = 0.0
    cov = torch.sum((volume - mean.unsqueeze(0)) ** 2, dim=0) / (cnt + 1e-8)
    cov[:, cnt[0] == 0] = 1e6
    cov = torch.exp(-cov)
    return mean, cnt, cov


# --------------------------------------------------------------------------- #
# NeRF MLP (A6 / A10), functional on a reference-named state dict
# --------------------------------------------------------------------------- #
def sinusoidal_encode(x: Tensor, n_deg: int) -> Tensor:
    """``[x | sin(x*2^k) k<n | sin(x*2^k + pi/2) k<n]``, degree-major.

    nerf_mlp.py:181-197 (min_deg=0, use_identity=True).
    """
    scales = torch.tensor([2 ** i for i in range(n_deg)])
    xb = (x[..., None, :] * scales[:, None]).reshape(*x.shape[:-1], n_deg * x.shape[-1])
    return torch.cat([x, torch.sin(torch.cat([xb, xb + 0.5 * math.pi], dim=-1))], dim=-1)


def _mlp_trunk(sd: Dict[str, Tensor], x: Tensor, depth: int = 4, skip: int = 3) -> Tensor:
    """nerf_mlp.py:80-90 with output disabled: Linear+ReLU, input re-joined after layer `skip`."""
    inp = x
    for i in range(depth):
        x = F.relu(F.linear(x, sd[f"mlp.base.hidden_layers.{i}.weight"], sd[f"mlp.base.hidden_layers.{i}.bias"]))
        if i % skip == 0 and i > 0:
            x = torch.cat([x, inp], dim=-1)
    return x


def nerf_query_density(sd: Dict[str, Tensor], xyz: Tensor, features: Tensor) -> Tensor:
    """relu(sigma) for points (N,3) and global features (N,70).  nerf_mlp.py:224-227,138-144."""
    h = _mlp_trunk(sd, torch.cat([sinusoidal_encode(xyz, 10), features], dim=-1))
    s = F.linear(h, sd["mlp.sigma_layer.output_layer.weight"], sd["mlp.sigma_layer.output_layer.bias"])
    return F.relu(s)


def nerf_forward(sd: Dict[str, Tensor], xyz: Tensor, ray_d: Tensor, features: Tensor) -> Tuple[Tensor, Tensor]:
    """(rgb, sigma) for samples (R,S,3); ray_d (R,3) broadcast over S.  nerf_mlp.py:229-234,146-161."""
    h = _mlp_trunk(sd, torch.cat([sinusoidal_encode(xyz, 10), features], dim=-1))
    sigma = F.linear(h, sd["mlp.sigma_layer.output_layer.weight"], sd["mlp.sigma_layer.output_layer.bias"])
    cond = sinusoidal_encode(ray_d, 4)
    if cond.shape[:-1] != h.shape[:-1]:
        cond = cond.view([cond.shape[0]] + [1] * (h.dim() - cond.dim()) + [cond.shape[-1]]).expand(*h.shape[:-1], cond.shape[-1])
    b = F.linear(h, sd["mlp.bottleneck_layer.output_layer.weight"], sd["mlp.bottleneck_layer.output_layer.bias"])
    z = torch.cat([b, cond], dim=-1)
    z = F.relu(F.linear(z, sd["mlp.rgb_layer.hidden_layers.0.weight"], sd["mlp.rgb_layer.hidden_layers.0.bias"]))
    rgb = F.linear(z, sd["mlp.rgb_layer.output_layer.weight"], sd["mlp.rgb_layer.output_layer.bias"])
    return torch.sigmoid(rgb), F.relu(sigma)


# --------------------------------------------------------------------------- #
# A5 + A6  density branch on the voxel grid and alpha gating
# --------------------------------------------------------------------------- #
def density_features(volume: Tensor, rgb_volume: Tensor, cnt: Tensor,
                     map_w: Tensor, map_b: Tensor) -> Tensor:
    """Per-voxel 70-ch NeRF conditioning vector, (N, 70), channels interleaved mean/cov (see below).

    nerfdet.py:234-253: Linear(256->32) on every voxel-view (so a view that
    does not see the voxel contributes the *bias*), concat RGB in front,
    mean over views / (cnt+1e-8) (NOT zeroed at cnt==0), unmasked variance,
    exp(-var) with 1e6 at cnt==0.
    """
    n_v, c = volume.shape[:2]
    grid = volume.shape[2:]
    flat = volume.reshape(n_v, c, -1).permute(0, 2, 1).contiguous()
    mapped = F.linear(flat, map_w, map_b).permute(0, 2, 1).contiguous().view(n_v, -1, *grid)
    both = torch.cat([rgb_volume, mapped], dim=1)
    mean = both.sum(dim=0) / (cnt + 1e-8)
    var = torch.sum((both - mean.unsqueeze(0)) ** 2, dim=0) / (cnt + 1e-8)
    var[:, cnt[0] == 0] = 1e6
    var = torch.exp(-var)
    # nerfdet.py:251-253 concatenates the two (35,X,Y,Z) tensors along dim=1 -- the X axis, there is
    # no batch dim here -- and then views the (35,2X,Y,Z) result as (70, N).  Net effect: the 70
    # channels are INTERLEAVED [mean_0, cov_0, mean_1, cov_1, ...], unlike the ray branch's
    # [mean(35) | cov(35)] (render_ray.py:303).  This is what the shipped weights were trained with.
    g = torch.stack([mean, var], dim=1).reshape(2 * mean.shape[0], -1)
    return g.permute(1, 0).contiguous()


def gate_volume(mean: Tensor, cnt: Tensor, density: Tensor) -> Tensor:
    """``(1 - exp(-density)) * mean``, zero where no view sees the voxel.  nerfdet.py:257-261."""
    alpha = 1 - torch.exp(-density)
    out = alpha.view(1, *mean.shape[1:]) * mean
    out[:, cnt[0] == 0] = 0.0
    return out


def extract_volume(features: Tensor, denorm_images: Tensor, img_meta: dict,
                   n_voxels, voxel_size, map_w: Tensor, map_b: Tensor,
                   nerf_sd: Dict[str, Tensor], stride: int = 4) -> Dict[str, Tensor]:
    """Steps 2-11 of nerfdet.extract_feat (nerfdet.py:152-261) for one scene, image mode,
    nerf_density=True: FPN level-0 features (n_v,C,Hf,Wf) + de-normalised images
    (n_v,3,H,W) -> gated voxel volume (C,X,Y,Z) and the view count (1,X,Y,Z).
    Every intermediate the reference materialises is materialised here too.
    """
    proj = compute_projection(img_meta, stride)
    pts = get_points(n_voxels, voxel_size, img_meta["lidar2img"]["origin"])
    h = img_meta["img_shape"][0] // stride
    w = img_meta["img_shape"][1] // stride
    volume, valid = backproject(features[:, :, :h, :w], pts, proj)
    mean, cnt, cov = aggregate_views(volume, valid)
    rgb_proj = compute_projection(img_meta, 1)
    rgb_volume, _ = backproject(denorm_images[:, :, :img_meta["img_shape"][0], :img_meta["img_shape"][1]], pts, rgb_proj)
    glob = density_features(volume, rgb_volume, cnt, map_w, map_b)
    xyz = pts.view(3, -1).permute(1, 0).contiguous()
    density = nerf_query_density(nerf_sd, xyz, glob)
    out = gate_volume(mean, cnt, density)
    return dict(volume=out, valid=cnt, mean=mean, cov=cov, global_feat=glob, density=density,
                points=pts, projection=proj, rgb_projection=rgb_proj)


def map_features_2d(features: Tensor, map_w: Tensor, map_b: Tensor) -> Tensor:
    """Linear(256->32) on every feature pixel, (n_v,C,h,w)->(n_v,32,h,w).  nerfdet.py:194-197."""
    n_v, c, h, w = features.shape
    f = features.reshape(n_v, c, -1).permute(0, 2, 1).contiguous()
    return F.linear(f, map_w, map_b).permute(0, 2, 1).contiguous().view(n_v, -1, h, w)


# --------------------------------------------------------------------------- #
# A7  ray samples -> every source view, bilinear
# --------------------------------------------------------------------------- #
def project_samples(xyz: Tensor, cameras: Tensor) -> Tuple[Tensor, Tensor]:
    """projection.py:42-64.  xyz (R,S,3), cameras (n_v,34) -> pixel (n_v,R,S,2), in_front (n_v,R,S)."""
    shape = xyz.shape[:2]
    p = xyz.reshape(-1, 3)
    n_v = len(cameras)
    k = cameras[:, 2:18].reshape(-1, 4, 4)
    e = cameras[:, -16:].reshape(-1, 4, 4)
    ph = torch.cat([p, torch.ones_like(p[..., :1])], dim=-1)
    q = k.bmm(e).bmm(ph.t()[None, ...].repeat(n_v, 1, 1)).permute(0, 2, 1)
    pix = q[..., :2] / torch.clamp(q[..., 2:3], min=1e-8)
    pix = torch.clamp(pix, min=-1e6, max=1e6)
    return pix.reshape((n_v,) + shape + (2,)), (q[..., 2] > 0).reshape((n_v,) + shape)


def projector_compute(xyz: Tensor, train_imgs: Tensor, train_cameras: Tensor,
                      featmaps: Optional[Tensor]) -> Tuple[Optional[Tensor], Tensor]:
    """projection.py:91-151 (grid_sample=True path).

    xyz (R,S,3); train_imgs (1,n_v,H,W,3); train_cameras (1,n_v,34); featmaps (n_v,d,h,w)
    -> rgb_feat (R,S,n_v,3+d), mask (R,S,n_v,1) float.
    """
    assert train_imgs.shape[0] == 1 and train_cameras.shape[0] == 1
    imgs = train_imgs.squeeze(0).permute(0, 3, 1, 2)
    cams = train_cameras.squeeze(0)
    h, w = cams[0][:2]
    pix, in_front = project_samples(xyz, cams)
    scale = torch.tensor([w - 1.0, h - 1.0])[None, None, :]
    norm = 2 * pix / scale - 1.0
    rgb = F.grid_sample(imgs, norm, align_corners=True).permute(2, 3, 0, 1)
    out = None
    if featmaps is not None:
        ft = F.grid_sample(featmaps, norm, align_corners=True).permute(2, 3, 0, 1)
        out = torch.cat([rgb, ft], dim=-1)
    inb = (pix[..., 0] <= w - 1.0) & (pix[..., 0] >= 0) & (pix[..., 1] <= h - 1.0) & (pix[..., 1] >= 0)
    mask = (inb * in_front).float().permute(1, 2, 0)[..., None]
    return out, mask


# --------------------------------------------------------------------------- #
# A8  masked multi-view statistics
# --------------------------------------------------------------------------- #
def compute_mask_points(feature: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    """render_ray.py:71-93: masked mean, unmasked-sum variance / sum(mask), exp(-var)."""
    denom = torch.sum(mask, dim=2, keepdim=True) + 1e-8
    mean = torch.sum(feature * (mask / denom), dim=2, keepdim=True)
    var = torch.sum((feature - mean) ** 2, dim=2, keepdim=True) / denom
    return mean, torch.exp(-var)


# --------------------------------------------------------------------------- #
# A9  samples along rays
# --------------------------------------------------------------------------- #
def sample_along_camera_ray(ray_o: Tensor, ray_d: Tensor, depth_range, n_samples: int,
                            det: bool = False, t_rand: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """render_ray.py:145-189 (inv_uniform=False).  ``t_rand`` injects the jitter
    the reference draws with ``torch.rand_like`` so a test can replay it."""
    near, far = depth_range
    assert near > 0 and far > 0 and far > near
    near_t = near * torch.ones_like(ray_d[..., 0])
    far_t = far * torch.ones_like(ray_d[..., 0])
    step = (far_t - near_t) / (n_samples - 1)
    z = torch.stack([near_t + i * step for i in range(n_samples)], dim=1)
    if not det:
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat([mids, z[:, -1:]], dim=-1)
        lower = torch.cat([z[:, 0:1], mids], dim=-1)
        if t_rand is None:
            t_rand = torch.rand_like(z)
        z = lower + (upper - lower) * t_rand
    d = ray_d.unsqueeze(1).repeat(1, n_samples, 1)
    o = ray_o.unsqueeze(1).repeat(1, n_samples, 1)
    return z.unsqueeze(2) * d + o, z


# --------------------------------------------------------------------------- #
# A11  compositing
# --------------------------------------------------------------------------- #
def raw2outputs(raw: Tensor, z_vals: Tensor, mask: Optional[Tensor], white_bkgd: bool = False) -> "OrderedDict[str, Tensor]":
    """render_ray.py:196-247: alpha = 1-exp(-sigma) (no interval), exclusive cumprod
    transmittance with +1e-10, weight-normalised depth clamped to the global z range,
    ray mask = more than 8 samples seen by >1 view."""
    rgb = raw[:, :, :3]
    sigma = raw[:, :, 3]
    alpha = 1.0 - torch.exp(-sigma)
    t = torch.cumprod(1.0 - alpha + 1e-10, dim=-1)[:, :-1]
    t = torch.cat((torch.ones_like(t[:, 0:1]), t), dim=-1)
    wgt = alpha * t
    rgb_map = torch.sum(wgt.unsqueeze(2) * rgb, dim=1)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - torch.sum(wgt, dim=-1, keepdim=True))
    if mask is not None:
        mask = mask.float().sum(dim=1) > 8
    depth = torch.sum(wgt * z_vals, dim=-1) / (torch.sum(wgt, dim=-1) + 1e-8)
    depth = torch.clamp(depth, z_vals.min(), z_vals.max())
    return OrderedDict([("rgb", rgb_map), ("depth", depth), ("weights", wgt), ("mask", mask),
                        ("alpha", alpha), ("z_vals", z_vals), ("transparency", t)])


# --------------------------------------------------------------------------- #
# A12  ray rendering, image mode
# --------------------------------------------------------------------------- #
def render_rays_func(ray_o: Tensor, ray_d: Tensor, features_2d: Tensor, img: Tensor,
                     near_far_range, n_samples: int, nerf_sd: Dict[str, Tensor], img_meta: dict,
                     det: bool = False, t_rand: Optional[Tensor] = None, white_bkgd: bool = False) -> dict:
    """render_ray.py:250-327, mode="image", N_importance=0.

    img (n_v,3,H,W) de-normalised images; features_2d (n_v,32,h,w) mapped features.
    """
    pts, z = sample_along_camera_ray(ray_o, ray_d, near_far_range, n_samples, det=det, t_rand=t_rand)
    imgs = img.permute(0, 2, 3, 1).unsqueeze(0)
    cams = compute_ray_cameras(img_meta)
    rgb_feat, mask = projector_compute(pts, imgs, cams, features_2d)
    pixel_mask = mask[..., 0].sum(dim=2) > 1
    mean, var = compute_mask_points(rgb_feat, mask)
    glob = torch.cat([mean, var], dim=-1).squeeze(2)
    rgb_pts, sigma_pts = nerf_forward(nerf_sd, pts, ray_d, glob)
    out = raw2outputs(torch.cat([rgb_pts, sigma_pts], dim=-1), z, pixel_mask, white_bkgd=white_bkgd)
    return dict(outputs_coarse=out, sigma=sigma_pts, globalfeat=glob, pts=pts)


def select_training_rays(ray_batch: dict, n_rand: int, rng: np.random.RandomState):
    """Ray selection of render_ray.py:408-427: flatten target views, drop rays whose
    gt depth is 0, draw ``n_rand`` without replacement from the reference's RandomState."""
    ray_o = ray_batch["ray_o"].view(-1, 3)
    ray_d = ray_batch["ray_d"].view(-1, 3)
    gt_rgb = ray_batch["gt_rgb"].view(-1, 3)
    gt_depth = ray_batch["gt_depth"]
    if len(gt_depth) != 0:
        gt_depth = gt_depth.view(-1, 1)
        keep = (gt_depth > 0).squeeze(-1)
        ray_o, ray_d, gt_rgb, gt_depth = ray_o[keep], ray_d[keep], gt_rgb[keep], gt_depth[keep]
    else:
        gt_depth = None
    sel = rng.choice(ray_d.shape[0], size=(n_rand,), replace=False)
    return ray_o[sel], ray_d[sel], gt_rgb[sel], (gt_depth[sel] if gt_depth is not None else None)


def nvs_loss(rgb: Tensor, gt: Tensor, mask: Tensor) -> Tensor:
    """nerfdet.py:296-307 (use_nerf_mask=True)."""
    return torch.sum(mask.unsqueeze(-1) * (rgb - gt) ** 2) / (mask.sum() + 1e-6)


def depth_loss(depth: Tensor, gt: Tensor, mask: Tensor) -> Tensor:
    """nerfdet.py:309-321 (use_nerf_mask=True)."""
    return torch.sum(mask * torch.abs(depth - gt.squeeze(-1))) / (mask.sum() + 1e-6)


# --------------------------------------------------------------------------- #
# A13  3D neck, functional on a reference-named state dict
# --------------------------------------------------------------------------- #
def _bn3d(sd, prefix: str, x: Tensor, training: bool) -> Tensor:
    return F.batch_norm(x, sd[prefix + ".running_mean"].clone(), sd[prefix + ".running_var"].clone(),
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training, 0.1, 1e-5)


def _basic_block3d(sd, prefix: str, x: Tensor, stride: int, training: bool) -> Tensor:
    """necks/imvoxelnet.py:233-260."""
    out = F.conv3d(x, sd[prefix + ".conv1.weight"], None, stride, 1)
    out = F.relu(_bn3d(sd, prefix + ".norm1", out, training))
    out = F.conv3d(out, sd[prefix + ".conv2.weight"], None, 1, 1)
    out = _bn3d(sd, prefix + ".norm2", out, training)
    idt = x
    if stride != 1:
        idt = _bn3d(sd, prefix + ".downsample.1", F.conv3d(x, sd[prefix + ".downsample.0.weight"], None, stride), training)
    return F.relu(out + idt)


def neck3d_forward(sd: Dict[str, Tensor], x: Tensor, n_scales: int = 3, training: bool = False) -> List[Tensor]:
    """FastIndoorImVoxelNeck.forward, necks/imvoxelnet.py:22-34 with n_blocks=[1,1,1]."""
    downs = []
    for i in range(n_scales):
        x = _basic_block3d(sd, f"down_layer_{i}.0", x, 1 if i == 0 else 2, training)
        downs.append(x)
    outs = []
    for i in range(n_scales - 1, -1, -1):
        if i < n_scales - 1:
            p = f"up_block_{i + 1}"
            x = F.conv_transpose3d(x, sd[p + ".0.weight"], None, 2)
            x = F.relu(_bn3d(sd, p + ".1", x, training))
            x = F.conv3d(x, sd[p + ".3.weight"], None, 1, 1)
            x = F.relu(_bn3d(sd, p + ".4", x, training))
            x = downs[i] + x
        p = f"out_block_{i}"
        o = F.relu(_bn3d(sd, p + ".1", F.conv3d(x, sd[p + ".0.weight"], None, 1, 1), training))
        outs.append(o)
    return outs[::-1]


# --------------------------------------------------------------------------- #
# A14  detection head forward + box decoding
# --------------------------------------------------------------------------- #
def head_forward(sd: Dict[str, Tensor], feats: Sequence[Tensor]):
    """ScanNetImVoxelHeadV2.forward_single per level, imvoxel_head_v2.py:442-449."""
    ctr, reg, cls = [], [], []
    for i, x in enumerate(feats):
        ctr.append(F.conv3d(x, sd["centerness_conv.weight"], None, 1, 1))
        reg.append(torch.exp(F.conv3d(x, sd["reg_conv.weight"], None, 1, 1) * sd[f"scales.{i}.scale"]))
        cls.append(F.conv3d(x, sd["cls_conv.weight"], sd["cls_conv.bias"], 1, 1))
    return ctr, reg, cls


def _decode_boxes(points: Tensor, d: Tensor) -> Tensor:
    """imvoxel_head_v2.py:547-555: distances (x-,x+,y-,y+,z-,z+) -> (x1,y1,z1,x2,y2,z2)."""
    return torch.stack([points[:, 0] - d[:, 0], points[:, 1] - d[:, 2], points[:, 2] - d[:, 4],
                        points[:, 0] + d[:, 1], points[:, 1] + d[:, 3], points[:, 2] + d[:, 5]], -1)


def head_get_bboxes(ctr, reg, cls, valid: Tensor, origin, voxel_size, nms_pre: int,
                    score_thr: float, iou_thr: float, n_classes: int = 18):
    """get_bboxes/_get_bboxes_single/_nms for batch element 0, imvoxel_head_v2.py:216-285,528-545.

    ``valid`` is the float view-count (1,1,X,Y,Z).  Returns the pre-NMS candidates and the
    picked indices so the GPU path can be compared on *indices*, plus the final
    (centre,size) boxes, scores, labels.
    """
    boxes_l, scores_l = [], []
    for i in range(len(ctr)):
        size = ctr[i].shape[-3:]
        v = F.interpolate(valid, size=size, mode="trilinear").round().bool()[0]
        pts = get_points(list(size), torch.tensor(voxel_size) * (2 ** i), origin).reshape(3, -1).transpose(0, 1)
        c = ctr[i][0].permute(1, 2, 3, 0).reshape(-1).sigmoid()
        b = reg[i][0].permute(1, 2, 3, 0).reshape(-1, reg[i].shape[1])
        s = cls[i][0].permute(1, 2, 3, 0).reshape(-1, n_classes).sigmoid()
        vv = v.permute(1, 2, 3, 0).reshape(-1)
        s = s * c[:, None] * vv[:, None]
        mx, _ = s.max(dim=1)
        if len(s) > nms_pre > 0:
            _, ids = mx.topk(nms_pre)
            b, s, pts = b[ids], s[ids], pts[ids]
        boxes_l.append(_decode_boxes(pts, b))
        scores_l.append(s)
    boxes = torch.cat(boxes_l)
    scores = torch.cat(scores_l)
    best, labels = scores.max(dim=1)
    keep = best > score_thr
    cb, cs, cl = boxes[keep], best[keep], labels[keep]
    pick = aligned_3d_nms(cb, cs, cl, iou_thr)
    pb = cb[pick]
    out = torch.stack(((pb[:, 0] + pb[:, 3]) / 2.0, (pb[:, 1] + pb[:, 4]) / 2.0, (pb[:, 2] + pb[:, 5]) / 2.0,
                       pb[:, 3] - pb[:, 0], pb[:, 4] - pb[:, 1], pb[:, 5] - pb[:, 2]), dim=1)
    return dict(cand_boxes=cb, cand_scores=cs, cand_labels=cl, pick=pick,
                boxes=out, scores=cs[pick], labels=cl[pick],
                all_boxes=boxes, all_scores=scores)


# --------------------------------------------------------------------------- #
# A15  greedy class-aware axis-aligned 3D NMS
# --------------------------------------------------------------------------- #
def aligned_3d_nms(boxes: Tensor, scores: Tensor, classes: Tensor, thresh: float) -> Tensor:
    """core/post_processing/box3d_nms.py:91-138.

    Ascending argsort, take the last, keep the others whose class-masked IoU <= thresh.
    IoU has no epsilon: 0/0 -> NaN -> ``NaN <= thresh`` is False -> dropped.
    """
    lo, hi = boxes[:, :3], boxes[:, 3:6]
    vol = (hi - lo).prod(dim=1)
    order = torch.argsort(scores)
    picked: List[int] = []
    while order.numel() > 0:
        i = order[-1]
        picked.append(int(i))
        rest = order[:-1]
        a = torch.max(lo[i], lo[rest])
        b = torch.min(hi[i], hi[rest])
        ext = torch.clamp(b - a, min=0.0)
        inter = ext[:, 0] * ext[:, 1] * ext[:, 2]
        iou = inter / (vol[i] + vol[rest] - inter)
        iou = iou * (classes[i] == classes[rest]).float()
        order = rest[torch.nonzero(iou <= thresh, as_tuple=False).flatten()]
    return torch.tensor(picked, dtype=torch.long)


# --------------------------------------------------------------------------- #
# synthetic scene generator shared by tests and bench
# --------------------------------------------------------------------------- #
def ring_scene_meta(n_views: int, img_hw=(240, 320), radius: float = 2.5, height: float = 1.2,
                    origin=(0.0, 0.0, 0.5)) -> dict:
    """Seed-free camera rig of SURVEY.md 8(d): cameras on a ring looking at the world origin.

    ori_shape = 2x img_shape; fx = fy = 577.87 * (2H/480); principal point at the image centre.
    """
    h, w = img_hw
    oh, ow = 2 * h, 2 * w
    f = 577.87 * (oh / 480.0)
    k = np.eye(4, dtype=np.float32)
    k[0, 0] = k[1, 1] = f
    k[0, 2] = ow / 2.0 - 0.5
    k[1, 2] = oh / 2.0 - 0.5
    ext = []
    for i in range(n_views):
        th = 2.0 * math.pi * i / n_views
        cam = np.array([radius * math.cos(th), radius * math.sin(th), height], dtype=np.float64)
        fwd = -cam / np.linalg.norm(cam)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        r = np.stack([right, down, fwd])
        e = np.eye(4, dtype=np.float64)
        e[:3, :3] = r
        e[:3, 3] = -r @ cam
        ext.append(e.astype(np.float32))
    return dict(lidar2img=dict(intrinsic=k, extrinsic=ext, origin=np.asarray(origin, dtype=np.float32)),
                ori_shape=(oh, ow, 3), img_shape=(h, w, 3))
