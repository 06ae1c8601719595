"""torch.autograd glue for the training path: forward and backward of each op are HIP kernels
(csrc/volume_kernels.hip, ray_kernels.hip, backward_kernels.hip); the reference gets the same derivatives from
autograd over its materialised tensors ("autograd must flow through A3-A11 in training")."""
from __future__ import annotations

from ctypes import c_void_p

import torch

from . import _lib, ops
from ._lib import NDET_LAYOUT_CN, NDET_LAYOUT_NC, check


def _ptr(t):
    return c_void_p(0 if t is None else t.data_ptr())


def _stream(t):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dense_nhwc(t):
    """Logical (n,C,h,w) channels-last tensor whose (n,h,w,C) memory is DENSE.  The backward kernels walk the saved input and
    the gradient buffer with one index, so both must share their pitches; an [:h,:w] crop of a padded map (real ScanNet frames:
    59 of 60 feature rows) is copied once here."""
    f = ops.to_channels_last(t)
    rows = f.permute(0, 2, 3, 1)
    return f if rows.is_contiguous() else rows.contiguous().permute(0, 3, 1, 2)


# ---- deterministic gradient scatter (tests) ----
# The backward kernels scatter with float atomics: the order of the adds, and with it the last bits of every gradient, changes from run to run
# (Adam's sign-like first steps amplify that into visibly different training trajectories: DESIGN.md, training).  ``set_deterministic(True)`` switches
# them to 64-bit fixed-point integer atomics (csrc/ndet_common.hpp::ndet_scatter_add: contributions rounded to multiples of 2^-40, integer sums are
# order-independent): bitwise reproducible gradients, ~2x the scatter time and twice the buffer.  The default stays the fast float path.
DETERMINISTIC = False
_TORCH_PREV = None
_FIX = 2.0 ** -40


def set_deterministic(on: bool) -> bool:
    """Switch the gradient scatter of K1 / K2 / K4 backward to the order-independent fixed-point form; returns the previous setting."""
    global DETERMINISTIC, _TORCH_PREV
    prev, DETERMINISTIC = DETERMINISTIC, bool(on)
    check(_lib.load().ndet_measurement_knob(b"deterministic_scatter", int(DETERMINISTIC)), "measurement_knob")
    # the vendor library's part of the step (the data gradient of the stride-2 convolutions, ATen's index / scatter ops): its own deterministic
    # algorithms -- measured: without this 33 of 122 parameter gradients still differed between runs, all upstream of the library's stride-2 dgrad
    if DETERMINISTIC and _TORCH_PREV is None:
        _TORCH_PREV = (torch.backends.cudnn.deterministic, torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled())
        torch.backends.cudnn.deterministic = True
        torch.use_deterministic_algorithms(True, warn_only=True)
    elif not DETERMINISTIC and _TORCH_PREV is not None:
        torch.backends.cudnn.deterministic = _TORCH_PREV[0]
        torch.use_deterministic_algorithms(_TORCH_PREV[1], warn_only=_TORCH_PREV[2])
        _TORCH_PREV = None
    return prev


def _grad_buffer(shape, device):
    """Zeroed accumulation buffer of the scatter: fp32, or int64 fixed point in the deterministic mode (same element indexing)."""
    return torch.zeros(shape, dtype=torch.int64 if DETERMINISTIC else torch.float32, device=device)


def _grad_result(buf):
    return (buf.double() * _FIX).float() if buf.dtype == torch.int64 else buf


class BackprojectMean(torch.autograd.Function):
    """features (n_v,C,h,w) -> (mean (C,X,Y,Z), count (1,X,Y,Z)); nerfdet.py:164-176."""

    @staticmethod
    def forward(ctx, features, points, projection, channels_last_out):
        f = ops.to_channels_last(features.detach())
        out, cnt = ops.backproject_aggregate(f, points, projection, None, channels_last_out)
        ctx.save_for_backward(points, projection)
        ctx.fshape, ctx.fstrides = tuple(f.shape), (f.stride(0), f.stride(2))
        ctx.mark_non_differentiable(cnt)
        return out, cnt

    @staticmethod
    def backward(ctx, g_out, _g_cnt):
        points, projection = ctx.saved_tensors
        n_v, c, h, w = ctx.fshape
        n = g_out[0].numel()
        g_nc = g_out.permute(1, 2, 3, 0)
        if g_nc.is_contiguous():
            g, layout = g_nc, NDET_LAYOUT_NC
        else:
            g, layout = g_out.contiguous(), NDET_LAYOUT_CN
        g = g.float()
        dfeat = _grad_buffer((n_v, h, w, c), g.device)
        pts = points.float().contiguous()
        pj = projection.float().contiguous()
        check(_lib.load().ndet_backproject_aggregate_bwd(_ptr(g), layout, n_v, c, h, w, dfeat.stride(0), dfeat.stride(1), _ptr(pts), n,
                                                         _ptr(pj), _ptr(dfeat), _stream(g)), "backproject_aggregate_bwd")
        return _grad_result(dfeat).permute(0, 3, 1, 2), None, None, None


class DensityFeatures(torch.autograd.Function):
    """(mapped (n_v,cm,h,w), bias (cm)) -> global_feat (N, 2*(3+cm)); nerfdet.py:234-253."""

    @staticmethod
    def forward(ctx, mapped, bias, denorm_images, points, projection, rgb_projection):
        m = _dense_nhwc(mapped.detach())
        out = ops.density_features(m, bias.detach(), denorm_images, points, projection, rgb_projection)
        ctx.save_for_backward(m, bias.detach(), points, projection)
        return out

    @staticmethod
    def backward(ctx, g):
        m, bias, points, projection = ctx.saved_tensors
        n_v, cm, h, w = m.shape
        n = points[0].numel()
        g = g.float().contiguous()
        dm = _grad_buffer((n_v, h, w, cm), g.device)
        assert (m.stride(0), m.stride(2)) == (dm.stride(0), dm.stride(1)), "saved input and gradient buffer must share their pitches"
        db = _grad_buffer((cm,), g.device)
        check(_lib.load().ndet_density_features_bwd(_ptr(g), _ptr(m), n_v, cm, h, w, m.stride(0), m.stride(2), _ptr(bias.float().contiguous()),
                                                    _ptr(points.float().contiguous()), n, _ptr(projection.float().contiguous()), _ptr(dm),
                                                    _ptr(db), _stream(g)), "density_features_bwd")
        return _grad_result(dm).permute(0, 3, 1, 2), _grad_result(db), None, None, None, None


class RayViewStats(torch.autograd.Function):
    """featmaps (n_v,d,h,w) -> globalfeat (R,S,2*(3+d)) (+ masks); projection.py:91-151 + render_ray.py:71-93."""

    @staticmethod
    def forward(ctx, featmaps, xyz, train_imgs, train_cameras):
        from . import rays
        f = _dense_nhwc(featmaps.detach())
        glob, pm, vc = rays.ray_view_stats(xyz, train_imgs, train_cameras, f)
        cams = train_cameras.squeeze(0) if train_cameras.dim() == 3 else train_cameras
        ke, h, w = rays._camera_matrices(cams)
        ctx.save_for_backward(f, xyz.detach().float().reshape(-1, 3).contiguous(), rays._to_device(ke, xyz.device))
        ctx.hw = (h, w)
        ctx.mark_non_differentiable(pm, vc)
        return glob, pm, vc

    @staticmethod
    def backward(ctx, g, _gpm, _gvc):
        f, pts, ke = ctx.saved_tensors
        n_v, d, hf, wf = f.shape
        g = g.float().reshape(pts.shape[0], -1).contiguous()
        df = _grad_buffer((n_v, hf, wf, d), g.device)
        assert (f.stride(0), f.stride(2)) == (df.stride(0), df.stride(1)), "saved input and gradient buffer must share their pitches"
        from . import rays
        fn = _lib.load().ndet_ray_view_stats_packed_bwd if rays.packed_ok(n_v, d, backward=True) else _lib.load().ndet_ray_view_stats_bwd
        from . import trace
        # one float atomic per (sample, view, bilinear tap, channel) whose view sees the sample: the upper bound n * n_v * 4 * d is what the
        # kernel would issue with every view valid (the ring scenes of bench/tests: ~30 % are)
        trace.span("k_ray_stats_packed_bwd" if rays.packed_ok(n_v, d, backward=True) else "k_ray_view_stats_bwd",
                   lambda: check(fn(_ptr(g), _ptr(pts), pts.shape[0], _ptr(ke), n_v, ctx.hw[0], ctx.hw[1], _ptr(f), d, hf, wf, f.stride(0), f.stride(2),
                                    _ptr(df), _stream(g)), "ray_view_stats_bwd"),
                   bytes=4 * (g.numel() + n_v * d * hf * wf), atomics_max=pts.shape[0] * n_v * 4 * d, kind="atomics")
        return _grad_result(df).permute(0, 3, 1, 2), None, None, None


class Composite(torch.autograd.Function):
    """raw (R,S,4) -> (rgb, depth, weights, alpha, transparency, ray_mask); render_ray.py:196-247.
    Gradients flow through rgb and depth (what the losses of nerfdet.py:296-321 use)."""

    @staticmethod
    def forward(ctx, raw, z_vals, pixel_mask, white_bkgd):
        from . import rays
        out = rays._raw2outputs_impl(raw.detach(), z_vals, pixel_mask, white_bkgd)
        z = z_vals.detach().float().contiguous()
        zmm = torch.stack([z.min(), z.max()])
        ctx.save_for_backward(raw.detach().float().contiguous(), z, out["transparency"], zmm)
        ctx.white = int(bool(white_bkgd))
        ctx.mark_non_differentiable(out["weights"], out["alpha"], out["transparency"])
        mask = out["mask"]
        if mask is None:
            mask = torch.empty(0, dtype=torch.bool, device=raw.device)
        ctx.mark_non_differentiable(mask)
        return out["rgb"], out["depth"], out["weights"], out["alpha"], out["transparency"], mask

    @staticmethod
    def backward(ctx, g_rgb, g_depth, *_):
        raw, z, trans, zmm = ctx.saved_tensors
        r, s = raw.shape[:2]
        g_rgb = g_rgb.float().contiguous()
        g_depth = None if g_depth is None else g_depth.float().contiguous()
        d_raw = torch.empty_like(raw)
        check(_lib.load().ndet_composite_bwd(_ptr(raw), _ptr(z), _ptr(trans), r, s, ctx.white, _ptr(zmm), _ptr(g_rgb), _ptr(g_depth), _ptr(d_raw),
                                             _stream(raw)), "composite_bwd")
        return d_raw, None, None, None
