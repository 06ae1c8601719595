"""Host mirror of the NeRF ray branch (A7-A12): ``Projector`` (mmdet3d/models/model_utils/projection.py:20-151),
``compute_mask_points`` / ``sample_along_camera_ray`` / ``raw2outputs`` / ``render_rays_func`` / ``render_rays``
(model_utils/render_ray.py:48-93,145-327,371-520), same signatures, bodies on the HIP kernels of
csrc/ray_kernels.hip.  The fused path (:func:`ray_view_stats`) never builds the (R,S,n_views,35) tensor."""
from __future__ import annotations

from collections import OrderedDict
from ctypes import c_void_p
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib, ops, trace
from ._lib import check
from .hostmath import matmul_mul_add

Tensor = torch.Tensor
rng = np.random.RandomState(234)  # render_ray.py:20 -- module-global stream used for ray selection


def _ptr(t):
    return c_void_p(0 if t is None else t.data_ptr())


def _stream(t):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


# ------------------------------------------------------------------------------------------------------
# A1 (ray-branch twin)
# ------------------------------------------------------------------------------------------------------
def _compute_projection(img_meta: dict) -> Tensor:
    """(1,n_views,34) rows ``[h, w | K(4x4) | E(4x4)]``, intrinsics rows 0-1 / (ori_h/img_h).  render_ray.py:48-69."""
    ext = img_meta["lidar2img"]["extrinsic"]
    n = len(ext)
    k = torch.tensor(np.asarray(img_meta["lidar2img"]["intrinsic"], dtype=np.float32)[:4, :4])
    k[:2] /= img_meta["ori_shape"][0] / img_meta["img_shape"][0]
    size = torch.tensor([float(img_meta["img_shape"][0]), float(img_meta["img_shape"][1])])
    e = torch.from_numpy(np.stack([np.asarray(x, dtype=np.float32) for x in ext])).reshape(n, 16)
    return torch.cat([size.expand(n, 2), k.reshape(1, 16).expand(n, 16), e], dim=-1).unsqueeze(0)


def _camera_matrices(train_cameras: Tensor) -> Tuple[Tensor, float, float]:
    """cameras (n_views,34) -> KE (n_views,3,4) = rows 0..2 of K @ E (projection.py:52-58), h, w."""
    cams = train_cameras.detach().to("cpu", torch.float32)
    k = cams[:, 2:18].reshape(-1, 4, 4)
    e = cams[:, -16:].reshape(-1, 4, 4)
    ke = torch.from_numpy(matmul_mul_add(k.numpy(), e.numpy())[:, :3, :].copy())     # = k.bmm(e), in a host-independent op order
    return ke, float(cams[0, 0]), float(cams[0, 1])


def _to_device(host: Tensor, device) -> Tensor:
    """Small per-scene constants go up through the pinned staging ring of nerfdet_amd.ops: a pageable copy waits for everything queued."""
    if torch.device(device).type != "cuda" or host.is_cuda:
        return host.to(device)
    from .ops import _upload_async
    return _upload_async(host, device)


def _prep_sources(train_imgs: Tensor, featmaps: Tensor):
    """train_imgs (1,n_v,H,W,3) as the reference passes it (a permuted view of the (n_v,3,H,W) images) or the
    (n_v,3,H,W) tensor itself; featmaps logical (n_v,d,h,w)."""
    if train_imgs.dim() == 5:
        assert train_imgs.shape[0] == 1, "only support batch_size=1 for now"  # projection.py:100-101
        rgb = train_imgs.squeeze(0).permute(0, 3, 1, 2)
    else:
        rgb = train_imgs
    if rgb.dtype != torch.float32:
        rgb = rgb.float()
    if rgb.stride(3) != 1:
        rgb = rgb.contiguous()
    f = ops.to_channels_last(featmaps)
    return rgb, f


_PACKED_RGB = {}  # one entry: the (n_v,H,W,4) copy of the scene's source images, reused by every chunk / backward of that scene


def packed_rgb(rgb: Tensor) -> Tensor:
    """(n_v,3,H,W) -> dense (n_v,H,W,4) fp32, 4th component 0 (one image pixel = one 16-byte load in the packed sampler).  Cached
    for the tensor it was made from (``render_testing`` walks one scene in hundreds of chunks)."""
    key = (rgb.data_ptr(), rgb._version, tuple(rgb.shape), tuple(rgb.stride()), str(rgb.device))
    hit = _PACKED_RGB.get("last")
    if hit is not None and hit[0] == key:
        return hit[1]
    n_v, _, h, w = rgb.shape
    out = torch.empty((n_v, h, w, 4), dtype=torch.float32, device=rgb.device)
    check(_lib.load().ndet_pack_rgb_nhwc4(_ptr(rgb), n_v, h, w, rgb.stride(0), rgb.stride(1), rgb.stride(2), _ptr(out), _stream(rgb)),
          "pack_rgb_nhwc4")
    _PACKED_RGB["last"] = (key, out, rgb)   # keeps the source alive so that its data_ptr stays unique
    return out


def packed_ok(n_views: int, d: int, backward: bool = False) -> bool:
    """Shapes the packed sampler (csrc/ray_stats_kernels.hip) takes; everything else runs on the generic kernel."""
    if d % 4 or d > 128 or n_views > 128:
        return False
    lanes = d // 4 + (0 if backward else 1)
    g, nvp = 64 // lanes, (n_views + 63) // 64 * 64
    if backward:
        return d <= 64 and 4 * g * nvp * 8 + 4 * g * d * 16 + 4 * g * 32 <= 64 * 1024
    return 4 * g * nvp * 8 <= 64 * 1024


def ray_view_stats(xyz: Tensor, train_imgs: Tensor, train_cameras: Tensor, featmaps: Tensor):
    """Fused A7+A8: sample points (R,S,3) -> ``globalfeat`` (R,S,2*(3+d)), ``pixel_mask`` (R,S) bool,
    ``view_count`` (R,S) int32.  Equals Projector.compute + compute_mask_points + cat + ``mask.sum(2) > 1``
    (render_ray.py:299-303)."""
    if not xyz.is_cuda:
        raise RuntimeError("nerfdet_amd.rays: tensors must live on the GPU (no CPU fallback)")
    cams = train_cameras.squeeze(0) if train_cameras.dim() == 3 else train_cameras
    ke, h, w = _camera_matrices(cams)
    ke = _to_device(ke, xyz.device)
    rgb, f = _prep_sources(train_imgs, featmaps)
    n_v, d, hf, wf = f.shape
    assert rgb.shape[0] == n_v and ke.shape[0] == n_v
    shape = xyz.shape[:-1]
    pts = xyz.to(torch.float32).reshape(-1, 3).contiguous()
    n = pts.shape[0]
    glob = torch.empty((n, 2 * (3 + d)), dtype=torch.float32, device=xyz.device)
    pm = torch.empty((n,), dtype=torch.bool, device=xyz.device)
    vc = torch.empty((n,), dtype=torch.int32, device=xyz.device)
    nbytes = 4 * (n_v * 3 * rgb.shape[2] * rgb.shape[3] + n_v * d * hf * wf + 2 * (3 + d) * n)
    if packed_ok(n_v, d) and f.stride(0) % 4 == 0 and f.stride(2) % 4 == 0 and f.data_ptr() % 16 == 0:
        rgb4 = packed_rgb(rgb)
        trace.span("k_ray_stats_packed", lambda: check(
            _lib.load().ndet_ray_view_stats_packed(_ptr(pts), n, _ptr(ke), n_v, h, w, _ptr(rgb4), rgb.shape[2], rgb.shape[3], _ptr(f), d, hf, wf,
                                                   f.stride(0), f.stride(2), _ptr(glob), _ptr(pm), _ptr(vc), _stream(xyz)),
            "ray_view_stats_packed"), bytes=nbytes, kind="hbm")
        return glob.view(*shape, -1), pm.view(*shape), vc.view(*shape)
    trace.span("k_ray_view_stats", lambda: check(
        _lib.load().ndet_ray_view_stats(_ptr(pts), n, _ptr(ke), n_v, h, w, _ptr(rgb), rgb.shape[2], rgb.shape[3], rgb.stride(0), rgb.stride(1),
                                        rgb.stride(2), _ptr(f), d, hf, wf, f.stride(0), f.stride(2), _ptr(glob), _ptr(pm), _ptr(vc),
                                        _stream(xyz)), "ray_view_stats"),
        bytes=nbytes, kind="hbm")
    return glob.view(*shape, -1), pm.view(*shape), vc.view(*shape)


class Projector:
    """projection.py:20-151.  ``compute`` keeps the reference's signature and materialised outputs."""

    def __init__(self, device="cuda"):
        self.device = device

    def compute(self, xyz, train_imgs, train_cameras, featmaps=None, grid_sample=True):
        assert (train_imgs.shape[0] == 1) and (train_cameras.shape[0] == 1)  # projection.py:100-101
        assert grid_sample, "grid_sample=False indexes (y, y) in the reference (projection.py:142) and is dead code"
        assert featmaps is not None, "featmaps=None is only reached in nerf_mode='volume' (dead in the configs)"
        if not xyz.is_cuda:
            raise RuntimeError("nerfdet_amd.rays: tensors must live on the GPU (no CPU fallback)")
        ke, h, w = _camera_matrices(train_cameras.squeeze(0))
        ke = _to_device(ke, xyz.device)
        rgb, f = _prep_sources(train_imgs, featmaps)
        n_v, d, hf, wf = f.shape
        r, s = xyz.shape[:2]
        pts = xyz.to(torch.float32).reshape(-1, 3).contiguous()
        out = torch.empty((r, s, n_v, 3 + d), dtype=torch.float32, device=xyz.device)
        mask = torch.empty((r, s, n_v, 1), dtype=torch.float32, device=xyz.device)
        check(_lib.load().ndet_project_sample(_ptr(pts), r * s, _ptr(ke), n_v, h, w, _ptr(rgb), rgb.shape[2], rgb.shape[3],
                                              rgb.stride(0), rgb.stride(1), rgb.stride(2), _ptr(f), d, hf, wf, f.stride(0),
                                              f.stride(2), _ptr(out), _ptr(mask), _stream(xyz)), "Projector.compute")
        return out, mask


def compute_mask_points(feature: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    """render_ray.py:71-93 on already-materialised samples (API parity; the fused path is ray_view_stats).
    Plain tensor algebra on the GPU."""
    denom = torch.sum(mask, dim=2, keepdim=True) + 1e-8
    mean = torch.sum(feature * (mask / denom), dim=2, keepdim=True)
    var = torch.sum((feature - mean) ** 2, dim=2, keepdim=True) / denom
    return mean, torch.exp(-var)


# ------------------------------------------------------------------------------------------------------
# A9
# ------------------------------------------------------------------------------------------------------
def sample_along_camera_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform=False, det=False, t_rand: Optional[Tensor] = None):
    """render_ray.py:145-189.  ``t_rand`` (R,S) replaces the reference's ``torch.rand_like`` draw when given."""
    assert not inv_uniform, "inv_uniform sampling is never enabled by the nerfdet configs"
    near, far = float(depth_range[0]), float(depth_range[1])
    assert near > 0 and far > 0 and far > near  # render_ray.py:161
    if not ray_d.is_cuda:
        raise RuntimeError("nerfdet_amd.rays: tensors must live on the GPU (no CPU fallback)")
    o = ray_o.to(torch.float32).contiguous()
    d = ray_d.to(torch.float32).contiguous()
    r = d.shape[0]
    if not det and t_rand is None:
        t_rand = torch.rand((r, N_samples), dtype=torch.float32, device=d.device)
    if det:
        t_rand = None
    elif t_rand is not None:
        t_rand = t_rand.to(torch.float32).contiguous()
    pts = torch.empty((r, N_samples, 3), dtype=torch.float32, device=d.device)
    z = torch.empty((r, N_samples), dtype=torch.float32, device=d.device)
    check(_lib.load().ndet_sample_along_rays(_ptr(o), _ptr(d), r, N_samples, near, far, _ptr(t_rand), _ptr(pts), _ptr(z), _stream(d)),
          "sample_along_camera_ray")
    return pts, z


# ------------------------------------------------------------------------------------------------------
# A11
# ------------------------------------------------------------------------------------------------------
def raw2outputs(raw, z_vals, mask, white_bkgd=False):
    """render_ray.py:196-247 -> OrderedDict(rgb, depth, weights, mask, alpha, z_vals, transparency).  Differentiable in
    ``raw`` (through rgb and depth) when it requires grad."""
    if torch.is_grad_enabled() and raw.requires_grad:
        from .autograd import Composite
        rgb, depth, wts, alpha, trans, rmask = Composite.apply(raw, z_vals, mask, white_bkgd)
        return OrderedDict([("rgb", rgb), ("depth", depth), ("weights", wts), ("mask", None if mask is None else rmask),
                            ("alpha", alpha), ("z_vals", z_vals), ("transparency", trans)])
    return _raw2outputs_impl(raw, z_vals, mask, white_bkgd)


def _raw2outputs_impl(raw, z_vals, mask, white_bkgd=False):
    if not raw.is_cuda:
        raise RuntimeError("nerfdet_amd.rays: tensors must live on the GPU (no CPU fallback)")
    r, s = raw.shape[:2]
    raw_c = raw.to(torch.float32).contiguous()
    z = z_vals.to(torch.float32).contiguous()
    dev = raw.device
    zmm = torch.stack([z.min(), z.max()])
    pm = None if mask is None else mask.to(torch.bool).contiguous()
    rgb = torch.empty((r, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((r,), dtype=torch.float32, device=dev)
    wts = torch.empty((r, s), dtype=torch.float32, device=dev)
    alpha = torch.empty((r, s), dtype=torch.float32, device=dev)
    trans = torch.empty((r, s), dtype=torch.float32, device=dev)
    rmask = None if mask is None else torch.empty((r,), dtype=torch.bool, device=dev)
    check(_lib.load().ndet_composite(_ptr(raw_c), _ptr(z), _ptr(pm), r, s, int(bool(white_bkgd)), _ptr(zmm), _ptr(rgb), _ptr(depth),
                                     _ptr(wts), _ptr(rmask), _ptr(alpha), _ptr(trans), _stream(raw)), "raw2outputs")
    return OrderedDict([("rgb", rgb), ("depth", depth), ("weights", wts), ("mask", rmask), ("alpha", alpha),
                        ("z_vals", z_vals), ("transparency", trans)])


# ------------------------------------------------------------------------------------------------------
# A12
# ------------------------------------------------------------------------------------------------------
def render_rays_func(ray_o, ray_d, mean_volume, cov_volume, features_2D, img, aabb, near_far_range, N_samples, N_rand=4096,
                     nerf_mlp=None, img_meta=None, projector=None, mode="volume", nerf_sample_view=3, inv_uniform=False,
                     N_importance=0, det=False, is_train=True, white_bkgd=False, gt_rgb=None, gt_depth=None, t_rand=None):
    """render_ray.py:250-369, ``mode='image'``, ``N_importance=0`` (the only reachable branch)."""
    assert mode == "image", "nerf_mode='volume' is not used by any nerfdet config"
    assert N_importance == 0, "the N_importance>0 branch references undefined names in the reference (render_ray.py:329-367)"
    ret = {"outputs_coarse": None, "outputs_fine": None, "gt_rgb": gt_rgb, "gt_depth": gt_depth}
    pts, z_vals = sample_along_camera_ray(ray_o, ray_d, near_far_range, N_samples, inv_uniform=inv_uniform, det=det, t_rand=t_rand)
    cams = _compute_projection(img_meta)
    if torch.is_grad_enabled() and features_2D.requires_grad:
        from .autograd import RayViewStats
        globalfeat, pixel_mask, _ = RayViewStats.apply(features_2D, pts, img, cams)
    else:
        globalfeat, pixel_mask, _ = ray_view_stats(pts, img, cams, features_2D)
    rgb_pts, density_pts = nerf_mlp(pts, ray_d, globalfeat)
    ret["sigma"] = density_pts
    ret["outputs_coarse"] = raw2outputs(torch.cat([rgb_pts, density_pts], dim=-1), z_vals, pixel_mask, white_bkgd=white_bkgd)
    return ret


RENDER_TESTING_RAYS = 8192     # rays per pass of the render_testing walk on the GPU (a multiple of N_rand is used)


def begin_selection(ray_batch):
    """First half of the training-time ray draw of render_ray.py:386-404: the rays WITH depth (``gt_depth > 0``; all rays when the scene has
    no depth maps), as indices into the flattened ray list, and their number.  Depends on the inputs only, so the detector takes it
    before it queues the backbone: the one host sync it needs then waits for nothing."""
    gt_depth = ray_batch["gt_depth"]
    rays = ray_batch["ray_d"].view(-1, 3)
    if len(gt_depth) == 0:
        return None, rays.shape[0], rays.device
    given = ray_batch.get("depth_rays")
    if given is not None and (given.dim() == 1 or given.shape[0] == 1):
        # found by the loader on the host, before the upload (datasets.py: DefaultFormatBundle3D; batches of one scene, config:133): the count is a
        # shape.  On the device the same ``nonzero`` reads its count back through a stream synchronisation -- with the step's log read lazily
        # (train.StepLog) the host would wait there for the whole previous step's backward.
        kept = given.reshape(-1)
        return kept, int(kept.numel()), rays.device
    kept = torch.nonzero(gt_depth.view(-1) > 0).view(-1)
    return kept, int(kept.numel()), rays.device


def _draw(n, n_rand):
    return rng.choice(n, size=(n_rand,), replace=False)


_DRAW_WORKER = None
THREADED_DRAW = False     # True: the permutation runs on a worker thread beside the backbone's launches.  Measured (tools/bench_train.py --threaded-draw,
                          # same box, alternating): no difference while the host runs ahead of the device (38.2 vs 38.3 ms per step), and 2 - 4 ms
                          # WORSE when it does not (host read every step: 47.0 - 49.8 vs 45.0 - 45.7 ms) -- the hand-over between the threads costs more
                          # than the 5 ms it hides behind launches that are themselves host-bound there


def submit_draw(begun, n_rand):
    """Start the host half of the ray draw on a worker thread: ``RandomState.choice(n, N_rand, replace=False)`` is a full Fisher-Yates shuffle of
    all n rays with depth (660 000 at the configs' sizes: ~5 ms) whatever N_rand is, and the reference's RNG stream (render_ray.py:20,398) leaves no
    way around it.  With THREADED_DRAW numpy shuffles on a worker (GIL released) while the backbone's launches go out -- ONE worker, submissions in
    call order: the module-global RandomState is consumed in the same order as by direct calls; by default the draw runs right here (see
    THREADED_DRAW for the measurement).  Returns a handle for :func:`collect_draw`."""
    global _DRAW_WORKER
    kept, n, dev = begun
    if not THREADED_DRAW:
        from concurrent.futures import Future
        done = Future()
        done.set_result(_draw(n, n_rand))
        return kept, dev, done
    if _DRAW_WORKER is None:
        from concurrent.futures import ThreadPoolExecutor
        _DRAW_WORKER = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ndet-ray-draw")
    return kept, dev, _DRAW_WORKER.submit(_draw, n, n_rand)


def collect_draw(handle):
    """Second half of :func:`submit_draw`: wait for the draw, upload it (pinned staging), map it through the rays-with-depth indices."""
    kept, dev, fut = handle
    draw = torch.from_numpy(fut.result())
    if torch.device(dev).type == "cuda":
        from .ops import _upload_async
        draw = _upload_async(draw, dev)           # pinned staging: a pageable copy here would wait for the queued backbone to drain
    return draw if kept is None else kept[draw]


def finish_selection(begun, n_rand):
    """Second half: ``N_rand`` of those rays drawn without replacement from the module-global RandomState (the reference's stream and call,
    render_ray.py:20,398 -- a permutation of all kept rays on the host, milliseconds).  Returns indices into the flattened ray list (device
    tensor).  The detector uses :func:`submit_draw` / :func:`collect_draw` instead, which run the permutation beside the backbone's launches."""
    return collect_draw(submit_draw(begun, n_rand))


def render_rays(ray_batch, mean_volume, cov_volume, features_2D, img, aabb, near_far_range, N_samples, N_rand=4096, nerf_mlp=None,
                img_meta=None, projector=None, mode="volume", nerf_sample_view=3, inv_uniform=False, N_importance=0, det=False,
                is_train=True, white_bkgd=False, render_testing=False, selection=None):
    """render_ray.py:371-520: training = drop rays without depth, draw ``N_rand`` rays from the module-global
    RandomState, one ``render_rays_func``; ``render_testing`` = every ray of the target views in chunks of
    ``N_rand``, deterministic sampling; otherwise ``None``."""
    ray_o, ray_d, gt_rgb, gt_depth = ray_batch["ray_o"], ray_batch["ray_d"], ray_batch["gt_rgb"], ray_batch["gt_depth"]
    if is_train:
        ray_o, ray_d, gt_rgb = ray_o.view(-1, 3), ray_d.view(-1, 3), gt_rgb.view(-1, 3)
        gt_depth = gt_depth.view(-1, 1) if len(gt_depth) != 0 else None
        sel = selection if selection is not None else finish_selection(begin_selection(ray_batch), N_rand)
        ray_o, ray_d, gt_rgb = ray_o[sel], ray_d[sel], gt_rgb[sel]
        if gt_depth is not None:
            gt_depth = gt_depth[sel]
        return render_rays_func(ray_o, ray_d, mean_volume, cov_volume, features_2D, img, aabb, near_far_range, N_samples, N_rand,
                                nerf_mlp, img_meta, projector, mode, nerf_sample_view, inv_uniform, N_importance, det, is_train,
                                white_bkgd, gt_rgb, gt_depth)
    if render_testing:
        nerf_size = ray_batch["nerf_sizes"][0]
        view_num = ray_o.shape[1]
        hh, ww = int(nerf_size[0][0]), int(nerf_size[0][1])
        ray_o, ray_d, gt_rgb = ray_o.view(-1, 3), ray_d.view(-1, 3), gt_rgb.view(-1, 3)
        gt_depth = gt_depth.view(-1, 1) if len(gt_depth) != 0 else None
        assert view_num * hh * ww == ray_o.shape[0]  # render_ray.py:468
        rgbs, depths = [], []
        # the reference walks the rays N_rand at a time (render_ray.py:470); a ray's result does not depend on its chunk mates (deterministic
        # sampling, row-wise MLP), so on the GPU several of its chunks go through one pass: fewer, larger GEMMs
        step = N_rand * max(1, RENDER_TESTING_RAYS // N_rand) if ray_o.is_cuda else N_rand
        for i in range(0, ray_o.shape[0], step):
            ret = render_rays_func(ray_o[i:i + step], ray_d[i:i + step], mean_volume, cov_volume, features_2D, img, aabb,
                                   near_far_range, N_samples, N_rand, nerf_mlp, img_meta, projector, mode, nerf_sample_view,
                                   inv_uniform, N_importance, True, is_train, white_bkgd, gt_rgb, gt_depth)
            rgbs.append(ret["outputs_coarse"]["rgb"])
            depths.append(ret["outputs_coarse"]["depth"])
        return {"outputs_coarse": {"rgb": torch.cat(rgbs, dim=0).view(view_num, hh, ww, 3),
                                   "depth": torch.cat(depths, dim=0).view(view_num, hh, ww, 1)},
                "gt_rgb": gt_rgb.view(view_num, hh, ww, 3),
                "gt_depth": gt_depth.view(view_num, hh, ww, 1) if gt_depth is not None else None}
    return None


def compute_psnr(pred: Tensor, target: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """PSNR of a rendered view against its ground truth, maximum pixel value 1: ``-10 log10(mean((pred - target)^2))``
    (model_utils/save_rendered_img.py:10-19)."""
    if mask is not None:
        pred, target = pred[mask], target[mask]
    return -10.0 * torch.log(((pred - target) ** 2).mean()) / float(np.log(10.0))


def compute_ssim(pred: Tensor, target: Tensor) -> Tensor:
    """SSIM of a rendered (H,W,3) view as the reference's ``compute_ssim`` obtains it (model_utils/save_rendered_img.py:21-36 ->
    ``skimage.metrics.structural_similarity(..., multichannel=True)`` of the pinned scikit-image 0.18.1): per channel, 7 x 7 uniform
    window, sample covariance, K1 = 0.01, K2 = 0.03, data range 2 (float images), float64, mean over the window-valid interior, then over
    the channels.  The 49-pixel box means are one average pooling each, on the device."""
    assert pred.shape == target.shape and pred.dim() == 3 and pred.shape[-1] == 3 and min(pred.shape[:2]) >= 7
    x, y = pred.permute(2, 0, 1).unsqueeze(0).double(), target.permute(2, 0, 1).unsqueeze(0).double()
    box = lambda t: torch.nn.functional.avg_pool2d(t, 7, stride=1)
    ux, uy, uxx, uyy, uxy = box(x), box(y), box(x * x), box(y * y), box(x * y)
    norm = 49.0 / 48.0
    vx, vy, vxy = norm * (uxx - ux * ux), norm * (uyy - uy * uy), norm * (uxy - ux * uy)
    c1, c2 = (0.01 * 2.0) ** 2, (0.03 * 2.0) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    return s.mean(dim=(2, 3)).mean()


def rendering_metrics(rendered: dict):
    """What ``save_rendered_img`` returns (model_utils/save_rendered_img.py:38-78) for one ``render_rays(render_testing=True)`` result,
    without its PNG dump: mean PSNR and mean SSIM over the target views (device scalars) and the mean squared depth error MAP
    (H,W,1) the reference calls rmse (None without depth maps)."""
    rgb, gt = rendered["outputs_coarse"]["rgb"], rendered["gt_rgb"]
    depth, gt_depth = rendered["outputs_coarse"]["depth"], rendered["gt_depth"]
    n = gt.shape[0]
    psnr = torch.stack([compute_psnr(rgb[v], gt[v]) for v in range(n)]).mean()
    ssim = torch.stack([compute_ssim(rgb[v], gt[v]) for v in range(n)]).mean()
    err = None if gt_depth is None else ((depth - gt_depth) ** 2).mean(dim=0)
    return psnr, ssim, err

