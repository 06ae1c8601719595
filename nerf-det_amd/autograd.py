"""torch.autograd glue for the training path: forward and backward of each op are HIP kernels
(csrc/volume_kernels.hip, ray_kernels.hip, backward_kernels.hip); the reference gets the same derivatives from
autograd over its materialised tensors ("autograd must flow through A3-A11 in training")."""
from __future__ import annotations

from ctypes import c_void_p

import torch

from . import _lib, ops
from ._lib import NDET_LAYOUT_CN, NDET_LAYOUT_NC, check
from ._lib import raw_stream


def _ptr(t):
    return c_void_p(0 if t is None else t.data_ptr())


def _stream(t):
    return c_void_p(raw_stream(t.device))


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


# ---- dense layers over many rows (the radiance MLP on rays x samples rows, the 2D feature mapping) ----
LINEAR_ROWS_MIN = 16384     # rows from which the weight gradient is worth splitting (below: ATen's own Linear)
LINEAR_SPLIT_ROWS = 2048    # contraction chunk of the weight-gradient GEMM


def _split_rows(n: int) -> int:
    """Largest chunk count S <= n / LINEAR_SPLIT_ROWS with n % S == 0 (1: no split)."""
    s = max(1, n // LINEAR_SPLIT_ROWS)
    while s > 1 and n % s:
        s -= 1
    return s


class LinearRows(torch.autograd.Function):
    """y = act(x W^T + b) over N rows, N >> the layer's widths (mmdet3d/models/model_utils/nerf_mlp.py:80-90: 131 072 sample rows through
    256-wide layers; nerfdet.py:194-197: 192 000 feature pixels through the 256 -> 32 mapping).  Forward and data gradient are the library
    GEMMs ATen would run.  The weight gradient dW = g^T x contracts over the N rows into a (Cout, Cin) result: handed to the GEMM library whole,
    that is one 32x64-tile workgroup per output tile (32 workgroups on 256 CUs, 389 us for 256x256x131072, 1.14 ms with Cin = 389: measured,
    profiles/r04_c_train_profile.txt); here the rows are cut into chunks of LINEAR_SPLIT_ROWS, one batched GEMM forms the per-chunk products and
    one reduction adds them (fixed order: deterministic).  The bias gradient (a column sum over N rows, 307 us through ATen's single-pass
    reduction) goes the same way in two stages.  The ReLU mask is applied to g in the same pass that feeds all three."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        lead = x.shape[:-1]
        x2 = x.detach().reshape(-1, x.shape[-1])
        y = torch.addmm(bias.detach(), x2, weight.detach().t()) if bias is not None else x2 @ weight.detach().t()
        if relu:
            torch.relu_(y)
        ctx.relu, ctx.has_bias = bool(relu), bias is not None
        ctx.save_for_backward(x2, weight.detach(), y if relu else None)
        return y.view(*lead, weight.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2, w, y = ctx.saved_tensors
        lead = g.shape[:-1]
        g2 = g.reshape(-1, g.shape[-1])
        if ctx.relu:
            g2 = torch.ops.aten.threshold_backward(g2, y, 0.0)
        n, s = g2.shape[0], _split_rows(g2.shape[0])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = (g2 @ w).view(*lead, w.shape[1])
        if ctx.needs_input_grad[1]:
            dw = g2.t() @ x2 if s == 1 else torch.bmm(g2.view(s, n // s, -1).transpose(1, 2), x2.view(s, n // s, -1)).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g2.sum(0) if s == 1 else g2.view(s, n // s, -1).sum(1).sum(0)
        return dx, dw, db, None


def linear_rows_train(x: torch.Tensor, lin: torch.nn.Linear, relu: bool = False) -> torch.Tensor:
    """``act(lin(x))``; with gradients enabled on the GPU and many rows, through :class:`LinearRows`."""
    rows = x.numel() // max(1, x.shape[-1])
    if torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and rows >= LINEAR_ROWS_MIN:
        return LinearRows.apply(x, lin.weight, lin.bias, relu)
    y = torch.nn.functional.linear(x, lin.weight, lin.bias)
    return torch.relu(y) if relu else y
