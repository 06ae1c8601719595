"""Training-time convolutions on the hand-written MFMA kernels (SURVEY.md 8a rows A13/A14 in ``forward_train``).

Forward, data gradient and weight gradient of the same-padded convolutions -- the convolutions of ``FastIndoorImVoxelNeck``
(mmdet3d/models/necks/imvoxelnet.py:22-67,233-260) and of the trainable ResNet stages / FPN -- run on ``ndet_conv_ndhwc_split``
(csrc/conv_split_kernels.hip):

  forward   y = conv(x, W)                                         the inference kernel, no epilogue (any stride)
  dgrad     dx = conv(dy, W'),  W'[ci, co, t] = W[co, ci, flip(t)]  stride 1: the same kernel on the transposed, tap-flipped weight
                                                                   (stride 2: the vendor library's data gradient)
  wgrad     dW[t][co][ci] = sum_o dy[o][co] x[s o + t - p][ci]      ONE launch of the same kernel as a plain GEMM: both tensors are
            staged channel-major over the flattened OUTPUT grid (csrc/pipeline_kernels.hip::k_wgrad_rows): the T copies of x, each
            sampled at its tap's positions, are stacked as GEMM rows (T*Cin, L), dy (Cout, L) plays the "weight" operand (split into
            its bf16 planes once), and the contraction runs over the L output voxels -- for any stride.

Transposed convolutions stay on the vendor library.  BatchNorm in training mode (batch statistics, BasicBlock3dV2) is left to ATen on
the same channels-last memory; nothing is copied between layouts."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch
from torch import nn

from . import conv3d as C
from ._lib import raw_stream

Tensor = torch.Tensor


def _raw_pack(w_taps_co_ci: Tensor, kernel: Sequence[int], stride: int = 1, pads=None) -> dict:
    """Pack dict of nerfdet_amd.conv3d for a bare weight already in (taps, Cout, Cin) order: no BatchNorm, no bias."""
    k = tuple(int(v) for v in kernel)
    ndim = len(k)
    return dict(w=w_taps_co_ci.contiguous().float(), scale=None, shift=None, cout=int(w_taps_co_ci.shape[1]), cin=int(w_taps_co_ci.shape[2]),
                ksize=k[0], stride=stride, transposed=False, kernel=k, strides=(stride,) * ndim,
                pads=tuple(v // 2 for v in k) if pads is None else tuple(pads), ndim=ndim, arith="bf16x3")


_STEP_SLOTS = {}      # (weight address, version) -> amax slot, filled by prepare_step for ONE step (cleared by the next call)
import weakref

_STEP_PARAMS = weakref.WeakKeyDictionary()     # model -> its convolution modules (the entry goes with the model: no model is kept alive by it)


def prepare_step(module: nn.Module) -> int:
    """Before a training step's forward (train.train_one_step): max |w| of every trainable convolution weight into an amax slot with two launches
    for the whole model (``torch._foreach_norm`` + one strided copy) instead of one ndet_amax_f32 per tensor -- 65 launches per step in the shipped
    model.  Weights it did not see fall back to their own pass (:func:`_split_both`).  Returns the number of slots prepared."""
    _STEP_SLOTS.clear()
    if C.train_arithmetic() != "f16x2":
        return 0
    mods = _STEP_PARAMS.get(module)
    if mods is None:      # the convolution MODULES are cached (a stable list); their weights are read off them every step (.to() / load_state_dict may replace them)
        mods = [m for m in module.modules() if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose3d))]
        if not any(m is module for m in mods):         # (a value that holds its own key would keep the entry alive for ever)
            _STEP_PARAMS[module] = mods
    ws = [m.weight for m in mods if m.weight.requires_grad and m.weight.is_cuda and m.weight.dtype == torch.float32]
    if not ws:
        return 0
    with torch.no_grad():
        norms = torch._foreach_norm([w.detach() for w in ws], float("inf"))
        block = C.AMAX.take_many(ws[0].device, len(ws))
        block[:, 0] = torch.stack(norms)
    for i, w in enumerate(ws):
        # with a weak reference to the parameter itself: an entry outlives its step (a caller may run forward / backward without coming through
        # train_one_step), and a NEW model's weight can land on a freed one's address with the same version count -- found by a test that built a
        # second model after a first had trained: stale maxima, wrong scales, NaN gradients.  A freed parameter's reference is dead; a live one
        # cannot share its address with another tensor.
        _STEP_SLOTS[(w.data_ptr(), w._version)] = (weakref.ref(w), block[i])
    return len(ws)


def _step_slot(w: Tensor):
    """The amax slot prepare_step left for the parameter behind ``w`` (a detached alias of it), or None."""
    hit = _STEP_SLOTS.get((w.data_ptr(), w._version))
    if hit is None:
        return None
    owner = hit[0]()
    if owner is None or owner.data_ptr() != w.data_ptr() or owner._version != w._version or owner.shape != w.shape:
        return None
    return hit[1]


def _split_both(w: Tensor, taps: int, arith: str, want_adjoint: bool):
    """(planes, adjoint planes or None, amax slot or None) of the torch-layout weight ``w``: ONE launch writes both packs
    (ndet_split_weights_train); the forward hands the adjoint planes to its backward through the autograd context, so a step splits each weight
    once.  fp16-pair: the scale comes from the weight's amax slot (one ndet_amax_f32 launch); nothing is read back to the host."""
    from ctypes import c_void_p
    from . import _lib
    cout, cin = int(w.shape[0]), int(w.shape[1])
    npl = 2 if arith == "f16x2" else 3
    planes = torch.empty((taps, cin // 32, npl, cout, 32), dtype=torch.int16, device=w.device)
    adj = torch.empty((taps, (cout + 31) // 32, npl, cin, 32), dtype=torch.int16, device=w.device) if want_adjoint else None
    wc = w.contiguous()
    st = c_void_p(raw_stream(w.device))
    slot = None
    if arith == "f16x2":
        slot = _step_slot(w)
        if slot is None:
            slot = C.AMAX.take(w.device)
            _lib.check(_lib.load().ndet_amax_f32(c_void_p(wc.data_ptr()), wc.numel(), c_void_p(slot.data_ptr()), st), "amax_f32")
    _lib.check(_lib.load().ndet_split_weights_train(c_void_p(wc.data_ptr()), taps, cout, cin, 1 if arith == "f16x2" else 0, c_void_p(0 if slot is None else slot.data_ptr()),
                                                    c_void_p(planes.data_ptr()), c_void_p(0 if adj is None else adj.data_ptr()), st), "split_weights_train")
    return planes, adj, slot


def _plane_pack(w: Tensor, planes: Tensor, slot, arith: str, kernel, adjoint: bool, stride: int = 1, pads=None) -> dict:
    cout, cin = int(w.shape[0]), int(w.shape[1])
    no, ki = (cin, (cout + 31) // 32 * 32) if adjoint else (cout, cin)
    k = tuple(int(v) for v in kernel)
    pk = dict(w=w, scale=None, shift=None, cout=no, cin=ki, ksize=k[0], stride=stride, transposed=False, kernel=k,
              strides=(stride,) * len(k), pads=tuple(v // 2 for v in k) if pads is None else tuple(pads), ndim=len(k), arith="bf16x3")
    if arith == "f16x2":
        pk.update(arith="f16x2", w_f16=(planes, 1.0), w_amax=slot, keep_amax=True)
    else:
        pk["w_split"] = planes
    return pk


def _train_pack(w: Tensor, kernel, adjoint: bool, stride: int = 1, pads=None, want_adjoint: bool = False) -> dict:
    """Pack dict for the layer (or, ``adjoint``, for its data gradient) straight from the torch-layout weight.  Split-family kernels: one launch
    writes the layer's planes and -- ``want_adjoint`` -- its data gradient's, returned under ``"_adjoint"`` for :func:`_adjoint_pack`; the
    fp32-MFMA family goes through the generic packer."""
    cout, cin = int(w.shape[0]), int(w.shape[1])
    taps = 1
    for v in kernel:
        taps *= int(v)
    arith = C.train_arithmetic()
    if arith not in ("bf16x3", "bf16", "f16x2"):
        if not adjoint:
            return _raw_pack(C.pack_weight(w), kernel, stride, pads)
        flip = w.flip(tuple(range(2, w.dim()))).transpose(0, 1)
        if cout % 32:
            flip = torch.nn.functional.pad(flip, (0, 0) * (w.dim() - 2) + (0, 32 - cout % 32))
        return _raw_pack(C.pack_weight(flip), kernel)
    if taps <= 27:
        planes, adj, slot = _split_both(w, taps, arith, adjoint or want_adjoint)
        pk = _plane_pack(w, adj if adjoint else planes, slot, arith, kernel, adjoint, stride, pads)
        if want_adjoint and not adjoint:
            pk["_adjoint"] = (adj, slot, arith)
        return pk
    # more taps than one workgroup's LDS block takes (none among the shipped models' trainable layers): one pack per launch, six-product arithmetic
    from ctypes import c_void_p
    from . import _lib
    no, ki = (cin, (cout + 31) // 32 * 32) if adjoint else (cout, cin)
    planes = torch.empty((taps, ki // 32, 3, no, 32), dtype=torch.int16, device=w.device)
    wc = w.contiguous()
    _lib.check(_lib.load().ndet_split_weights_bf16x3_torch(c_void_p(wc.data_ptr()), taps, cout, cin, int(adjoint), c_void_p(planes.data_ptr()),
                                                           c_void_p(raw_stream(w.device))), "split_weights_torch")
    return _plane_pack(w, planes, None, "bf16x3", kernel, adjoint, stride, pads)


def _adjoint_pack(handed, w: Tensor, kernel) -> dict:
    """The data gradient's pack: the planes the forward wrote alongside its own (``handed`` = its ``"_adjoint"`` entry), or a fresh split."""
    if handed is not None and handed[2] == C.train_arithmetic():
        return _plane_pack(w, handed[0], handed[1], handed[2], kernel, True)
    return _train_pack(w, kernel, True)


def _conv(x: Tensor, pk: dict) -> Tensor:
    return C.conv3d_ndhwc(x, pk) if pk["ndim"] == 3 else C.conv2d_nhwc(x, pk)


def eligible(conv: nn.Module, x: Tensor) -> bool:
    """Odd, same-padded (pad = k // 2), un-dilated, un-grouped convolution of uniform stride 1 or 2 whose input channel count the MFMA
    kernel steps through (a multiple of 32), on a float32 GPU tensor."""
    if not isinstance(conv, (nn.Conv3d, nn.Conv2d)) or not x.is_cuda or x.dtype != torch.float32:
        return False
    k = conv.kernel_size
    if isinstance(conv, nn.Conv3d) and len(set(k)) != 1:
        return False
    return (len(set(conv.stride)) == 1 and conv.stride[0] in (1, 2) and all(d == 1 for d in conv.dilation) and conv.groups == 1
            and all(v % 2 == 1 and p == v // 2 for v, p in zip(k, conv.padding)) and conv.padding_mode == "zeros"
            and conv.in_channels % 32 == 0)


FUSED_DY_PLANES = True    # dy goes to the weight-gradient GEMM's bf16 planes in one pass (k_wgrad_dy_planes) instead of transpose + split
IMPLICIT_WGRAD = True     # multi-tap weight gradients on large grids read x in place (k_wgrad_split); False: always the staged form
                          # (tap copies by k_wgrad_rows + one GEMM)
IMPLICIT_MIN_TAPS = 9     # fewer taps: the staged form (its GEMM runs on the faster tiles and the tap copies are 1 - 8 x the input)


def _rows(x: Tensor, k3, stride3, pads, t0: int, n_taps: int, lrow: int) -> Tensor:
    """csrc/pipeline_kernels.hip::k_wgrad_rows: (D,H,W,C) -> (n_taps, C, lrow) channel-major rows over the convolution's output grid."""
    import ctypes
    from ctypes import c_void_p
    from . import _lib
    d, h, w, c = x.shape
    out = torch.empty((n_taps, c, lrow), dtype=torch.float32, device=x.device)
    i3 = lambda v: (ctypes.c_int * 3)(*v)
    _lib.check(_lib.load().ndet_wgrad_rows(c_void_p(x.data_ptr()), d, h, w, c, i3(k3), i3(stride3), i3(pads), t0, n_taps, lrow,
                                           c_void_p(out.data_ptr()), c_void_p(raw_stream(x.device))), "wgrad_rows")
    return out


def _to_torch_layout(dw_rows: Tensor, taps: int, cin: int, cout: int, kernel, splits: int = 1) -> Tensor:
    """(taps * Cin, Cout) GEMM rows -> (Cout, Cin, *kernel): one coalesced pass (csrc: k_wgrad_to_torch) instead of ATen's strided copy.  ``splits`` > 1:
    ``dw_rows`` is a split-K workspace, its partial sums are added on the way (fixed order)."""
    if taps > 27 or cin % 32:
        assert splits == 1
        return dw_rows.view(taps, cin, cout).permute(2, 1, 0).reshape(cout, cin, *kernel)
    from ctypes import c_void_p
    from . import _lib
    out = torch.empty((cout, cin) + tuple(kernel), dtype=torch.float32, device=dw_rows.device)
    _lib.check(_lib.load().ndet_wgrad_to_torch(c_void_p(dw_rows.data_ptr()), splits, taps, cout, cin, c_void_p(out.data_ptr()),
                                               c_void_p(raw_stream(dw_rows.device))), "wgrad_to_torch")
    return out


def weight_grad(x: Tensor, g: Tensor, kernel: Sequence[int], stride: int = 1, pads=None, implicit=None) -> Tensor:
    """dW of a convolution of uniform stride (same-padded unless ``pads`` says otherwise).  x (D,H,W,Cin), g (OD,OH,OW,Cout) contiguous fp32
    channels-last (2D: D = batch, kernel (kh,kw)) -> (Cout, Cin, *kernel) in torch's layout."""
    two_d = len(kernel) == 2
    k3 = (1,) + tuple(kernel) if two_d else tuple(kernel)
    s3 = (1, stride, stride) if two_d else (stride,) * 3
    d, h, w, cin = x.shape
    cout = g.shape[3]
    pads = tuple(v // 2 for v in k3) if pads is None else ((0,) + tuple(pads) if two_d else tuple(pads))
    assert tuple(g.shape[:3]) == tuple((n + 2 * p - k) // s + 1 for n, p, k, s in zip((d, h, w), pads, k3, s3)), (tuple(x.shape), tuple(g.shape))
    lo = g.shape[0] * g.shape[1] * g.shape[2]
    lrow = ((lo + 31) // 32) * 32                           # the kernel steps the contraction by 32; the tail is staged as zeros
    taps = k3[0] * k3[1] * k3[2]
    # dy as the GEMM's "weight" operand (Cout rows over the output grid), split into its planes once
    arith = C.train_arithmetic()
    f16 = arith == "f16x2"
    x_slot = dy_slot = None
    if f16:
        dy_slot, x_slot = C.amax_of(g), C.amax_of(x)          # device slots; the scales and their inverses never leave the device
    if FUSED_DY_PLANES and arith in ("bf16x3", "bf16", "f16x2"):
        from ctypes import c_void_p
        from . import _lib
        st = c_void_p(raw_stream(x.device))
        planes = torch.empty((1, lrow // 32, 2 if f16 else 3, cout, 32), dtype=torch.int16, device=x.device)     # one pass: transpose + split
        if f16:
            _lib.check(_lib.load().ndet_wgrad_dy_planes_f16x2(c_void_p(g.data_ptr()), lo, cout, lrow, c_void_p(dy_slot.data_ptr()), c_void_p(planes.data_ptr()), st),
                       "wgrad_dy_planes_f16x2")
            # (no range guard: its bound, 2^-39 max|x| L max|dy|, is an absolute floor on a SUM over L output voxels -- for a weight gradient
            # the error that matters is relative to that sum's own size, and the bound trips on every early layer)
            pk = dict(w=planes, w_f16=(planes, 1.0), w_amax=dy_slot, guard=False, scale=None, shift=None, cout=cout, cin=lrow, ksize=1, stride=1, transposed=False,
                      kernel=(1, 1), strides=(1, 1), pads=(0, 0), ndim=2, arith="f16x2")
        else:
            _lib.check(_lib.load().ndet_wgrad_dy_planes(c_void_p(g.data_ptr()), lo, cout, lrow, c_void_p(planes.data_ptr()), st), "wgrad_dy_planes")
            pk = dict(w=planes, w_split=planes, scale=None, shift=None, cout=cout, cin=lrow, ksize=1, stride=1, transposed=False, kernel=(1, 1),
                      strides=(1, 1), pads=(0, 0), ndim=2, arith="bf16x3")
    else:
        grows = _rows(g, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0, 1, lrow)
        pk = dict(w=grows, scale=None, shift=None, cout=cout, cin=lrow, ksize=1, stride=1, transposed=False, kernel=(1, 1), strides=(1, 1),
                  pads=(0, 0), ndim=2, arith="bf16x3")
    if implicit is None:
        implicit = IMPLICIT_WGRAD and taps >= IMPLICIT_MIN_TAPS and lo >= 16384
    if implicit and arith in ("bf16x3", "bf16", "f16x2") and cin % 64 == 0:
        # multi-tap layers on large grids: x read in place (csrc/conv_split_kernels.hip::k_wgrad_split), no tap copies -- there the staged
        # form writes and re-reads taps x the input (707 MB for a 3x3x3 layer at 40x40x16x256); on small grids and 1x1 layers the staged
        # GEMM runs on the faster tiles and wins (tools/bench_wgrad.py)
        import ctypes
        from ctypes import c_void_p
        from . import _lib
        planes = pk["w_f16"][0] if f16 else C.split_planes(pk)
        bm = 128 if cin % 128 == 0 else 64
        bn = 256 if (f16 and bm == 128 and cout % 256 == 0 and taps * (cin // 128) >= 32) else (128 if cout > 64 else 64)      # (the library's own choice: ndet_wgrad_split*)
        tiles = (taps * cin // bm) * ((cout + bn - 1) // bn)
        ksteps = lrow // 32
        # ~1 000 workgroups of the 128 x 128 tile (768 left the FPN's 36-tile layer at 24 splits: 1 505 us against 1 260 at 32), ~800 of the 128 x 256
        # tile (the neck's 54-tile layers: 16 splits 439 us, 24 splits 477 us) -- tools/diag/wgrad_ab.py
        splits = max(1, min(32, ksteps // 8, -(-(768 if bn == 256 else 1024) // tiles)))
        if splits >= 6:
            splits = min(32, (splits + 7) // 8 * 8)           # a multiple of 8: one K split per XCD at a time (k_wgrad_split: the tiles of a split share its slice of x and dy in that L2)
        m = taps * cin
        ws = torch.empty((splits * m * cout,), dtype=torch.float32, device=x.device) if splits > 1 else None
        keep = f16 and splits > 1 and taps <= 27       # the partials go straight to ndet_wgrad_to_torch: no reduction pass, no (m, cout) intermediate
        dw = None if keep else torch.empty((m, cout), dtype=torch.float32, device=x.device)
        i3 = lambda v: (ctypes.c_int * 3)(*v)
        st = c_void_p(raw_stream(x.device))
        if f16:
            _lib.check(_lib.load().ndet_wgrad_split_f16x2(c_void_p(x.data_ptr()), d, h, w, cin, i3(k3), i3(s3), i3(pads), c_void_p(planes.data_ptr()), cout, lrow,
                                                          splits, c_void_p(x_slot.data_ptr()), c_void_p(dy_slot.data_ptr()), c_void_p(0 if ws is None else ws.data_ptr()),
                                                          c_void_p(ws.data_ptr() if keep else dw.data_ptr()), int(keep), st), "wgrad_split_f16x2")
            if keep:
                return _to_torch_layout(ws, taps, cin, cout, kernel, splits)
        else:
            _lib.check(_lib.load().ndet_wgrad_split(c_void_p(x.data_ptr()), d, h, w, cin, i3(k3), i3(s3), i3(pads), c_void_p(planes.data_ptr()), cout, lrow,
                                                    splits, 0 if arith == "bf16" else 2, c_void_p(0 if ws is None else ws.data_ptr()),
                                                    c_void_p(dw.data_ptr()), st), "wgrad_split")
        return _to_torch_layout(dw, taps, cin, cout, kernel)
    per = max(1, min(taps, (1 << 30) // (cin * lrow * 4)))  # the kernel addresses its operand with 32-bit byte offsets: <= 1 GiB per launch
    if f16 and per >= taps and taps <= 27 and cin % 32 == 0:
        pk["keep_partials"] = True                          # one GEMM: its split-K partials are added by ndet_wgrad_to_torch (no reduction pass)
    parts = []
    for t0 in range(0, taps, per):
        a_all = _rows(x, k3, s3, pads, t0, min(per, taps - t0), lrow)          # rows (t, ci): x sampled at tap t of every output voxel
        if f16:
            C._tag_amax(a_all, x_slot)                                         # copies of x's elements and zeros: max |rows| <= max |x|
        parts.append(C.linear_rows(C.carry_amax(a_all, a_all.view(-1, lrow)), pk))     # (taps*Cin, Cout): the sum over the output voxels
    kept = pk.pop("_partials", None)
    if kept is not None:
        return _to_torch_layout(kept[0], taps, cin, cout, kernel, kept[1])
    dw = parts[0] if len(parts) == 1 else torch.cat(parts)
    return _to_torch_layout(dw, taps, cin, cout, kernel)


def _dgrad_library(g: Tensor, x: Tensor, w: Tensor, stride: int) -> Tensor:
    """Data gradient of a strided convolution through ATen (channels-last memory in and out): g (OD,OH,OW,Cout) / (N,OH,OW,Cout)."""
    nd = w.dim() - 2
    if nd == 3:
        gl, xl = g.permute(3, 0, 1, 2).unsqueeze(0), x.permute(3, 0, 1, 2).unsqueeze(0)
    else:
        gl, xl = g.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2)
    pad = [int(v) // 2 for v in w.shape[2:]]
    dx = torch.ops.aten.convolution_backward(gl, xl, w, None, [stride] * nd, pad, [1] * nd, False, [0] * nd, 1, [True, False, False])[0]
    dx = dx[0].permute(1, 2, 3, 0) if nd == 3 else dx.permute(0, 2, 3, 1)
    return dx.contiguous()


def _dgrad_strided(g: Tensor, x: Tensor, w: Tensor, kernel, stride: int, handed=None) -> Tensor:
    """Data gradient of a same-padded convolution of stride 2.  Default: the vendor library (ATen -> MIOpen).  Its backward-data kernels sum
    with atomics: measured, they are what kept 33 of 122 parameter gradients from reproducing bit for bit with everything else deterministic
    (exactly the parameters upstream of layer4.0.conv2's data gradient).  In the deterministic mode (autograd.set_deterministic) the gradient
    is evaluated on the MFMA kernels instead: dy is written into a zero tensor of the input's extent at the positions stride * o, and the
    stride-1 adjoint convolution of that tensor is the gradient (dx[i] = sum_t w[t] dy_up[i - t + p]) -- 4x (2D) / 8x (3D) the necessary
    multiplications on three small layers, fixed summation order."""
    from . import autograd as A
    if not A.DETERMINISTIC:
        return _dgrad_library(g, x, w, stride)
    cout = g.shape[-1]
    up = g.new_zeros(tuple(x.shape[:-1]) + (cout + (-cout) % 32,))
    if len(kernel) == 3:
        up[::stride, ::stride, ::stride, :cout][:g.shape[0], :g.shape[1], :g.shape[2]] = g
    else:
        up[:, ::stride, ::stride, :cout][:, :g.shape[1], :g.shape[2]] = g
    return _conv(up, _adjoint_pack(handed, w, kernel))


class ConvS1(torch.autograd.Function):
    """y = conv(x, weight) for channels-last x (D,H,W,Cin) / (N,H,W,Cin) and a torch-layout weight; see the module docstring."""

    @staticmethod
    def forward(ctx, x, weight, stride=1):
        kernel = tuple(weight.shape[2:])
        ctx.kernel, ctx.stride = kernel, int(stride)
        w = weight.detach()
        xc = C.carry_amax(x, x.detach().contiguous())           # the backward kernels index it densely
        ctx.save_for_backward(xc, w)
        pk = _train_pack(w, kernel, False, int(stride), want_adjoint=ctx.needs_input_grad[0])
        ctx.adjoint = pk.pop("_adjoint", None)
        return _conv(xc, pk)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous().float()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if ctx.stride == 1:
                # the adjoint convolution: W'[ci, co, t] = W[co, ci, flip(t)]; the kernel steps its input channels by 32, so dy (and W')
                # are zero-padded when Cout is not a multiple
                gd = g if g.shape[-1] % 32 == 0 else torch.nn.functional.pad(g, (0, 32 - g.shape[-1] % 32))
                dx = _conv(gd, _adjoint_pack(ctx.adjoint, w, ctx.kernel))
            else:
                dx = _dgrad_strided(g, x, w, ctx.kernel, ctx.stride, ctx.adjoint)
        if ctx.needs_input_grad[1]:
            dw = weight_grad(x, g, ctx.kernel, ctx.stride)
        return dx, dw, None


class ConvAffineAct(torch.autograd.Function):
    """y = act(conv(x, W) * scale + shift (+ residual)) with a CONSTANT per-channel affine -- a convolution followed by an eval-mode
    BatchNorm whose parameters are frozen (the ResNet of the configs: ``norm_eval=True``, ``norm_cfg.requires_grad=False``), ReLU and the
    bottleneck's identity add -- as ONE launch of the inference kernel with its fused epilogue.  Backward: the ReLU mask and the affine
    are one elementwise pass over dy (g' = dy [y > 0] * scale, the identity's gradient is dy [y > 0]); g' then goes through the data and
    weight gradients of ConvS1.  Everything stays in channels-last memory: the library's BatchNorm backward hands its gradient back
    in NCHW order, which cost a transposing copy per layer."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, residual, relu, stride):
        kernel = tuple(weight.shape[2:])
        w = weight.detach()
        pk = _train_pack(w, kernel, False, int(stride), want_adjoint=ctx.needs_input_grad[0])
        ctx.adjoint = pk.pop("_adjoint", None)
        pk["scale"], pk["shift"] = scale, shift
        res = None if residual is None else residual.detach().contiguous()
        xc = C.carry_amax(x, x.detach().contiguous())           # the backward kernels index it densely
        y = (C.conv3d_ndhwc if pk["ndim"] == 3 else C.conv2d_nhwc)(xc, pk, residual=res, relu=1 if relu else 0)
        ctx.kernel, ctx.stride, ctx.relu, ctx.has_res = kernel, int(stride), bool(relu), residual is not None
        ctx.save_for_backward(xc, w, scale, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, scale, y = ctx.saved_tensors
        g = g.contiguous().float()
        want_res = ctx.has_res and ctx.needs_input_grad[4]
        from ctypes import c_void_p
        from . import _lib
        d_res = torch.empty_like(g) if want_res else None
        gs = torch.empty_like(g)
        if C.train_arithmetic() == "f16x2":      # the same pass leaves max |gs| for the fp16-pair data / weight gradients below
            slot = C.AMAX.take(g.device)
            _lib.check(_lib.load().ndet_relu_affine_bwd_amax(c_void_p(g.data_ptr()), c_void_p(y.data_ptr() if ctx.relu else 0), c_void_p(scale.data_ptr()),
                                                             g.numel() // g.shape[-1], g.shape[-1], int(ctx.relu), c_void_p(d_res.data_ptr() if want_res else 0),
                                                             c_void_p(gs.data_ptr()), c_void_p(slot.data_ptr()),
                                                             c_void_p(raw_stream(g.device))), "relu_affine_bwd_amax")
            C._tag_amax(gs, slot)
        else:
            _lib.check(_lib.load().ndet_relu_affine_bwd(c_void_p(g.data_ptr()), c_void_p(y.data_ptr() if ctx.relu else 0), c_void_p(scale.data_ptr()),
                                                        g.numel() // g.shape[-1], g.shape[-1], int(ctx.relu), c_void_p(d_res.data_ptr() if want_res else 0),
                                                        c_void_p(gs.data_ptr()), c_void_p(raw_stream(g.device))), "relu_affine_bwd")
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if ctx.stride == 1:
                gd = gs if gs.shape[-1] % 32 == 0 else torch.nn.functional.pad(gs, (0, 32 - gs.shape[-1] % 32))
                dx = _conv(gd, _adjoint_pack(ctx.adjoint, w, ctx.kernel))
            else:
                dx = _dgrad_strided(gs, x, w, ctx.kernel, ctx.stride, ctx.adjoint)
        if ctx.needs_input_grad[1]:
            dw = weight_grad(x, gs, ctx.kernel, ctx.stride)
        return dx, dw, None, None, d_res, None, None


def frozen_eval_bn(bn: nn.Module) -> bool:
    return (isinstance(bn, (nn.BatchNorm2d, nn.BatchNorm3d)) and not bn.training and bn.track_running_stats and bn.affine
            and not bn.weight.requires_grad and not bn.bias.requires_grad)


def conv_bn_act(conv: nn.Module, bn: nn.Module, x: Tensor, relu: bool = True, residual: Tensor = None) -> Tensor:
    """``act(bn(conv(x)) + residual)`` for a logical (B,C,H,W) tensor.  With an eligible convolution and a frozen eval-mode BatchNorm the
    whole expression is one launch (ConvAffineAct); anything else is evaluated op by op."""
    # Cout % 4: the backward's elementwise pass (ndet_relu_affine_bwd) works on channel quads
    if (torch.is_grad_enabled() and isinstance(conv, nn.Conv2d) and conv.bias is None and conv.out_channels % 4 == 0 and eligible(conv, x)
            and frozen_eval_bn(bn)):
        scale, shift = C.bn_affine(bn)
        xb = C.carry_amax(x, x.permute(0, 2, 3, 1))            # (the max |x| slot a previous launch left on x travels with its views)
        rb = None if residual is None else residual.permute(0, 2, 3, 1)
        y = ConvAffineAct.apply(xb if xb.is_contiguous() else C.carry_amax(xb, xb.contiguous()), conv.weight, scale, shift, rb, relu, conv.stride[0])
        return C.carry_amax(y, y.permute(0, 3, 1, 2))
    y = bn(conv_forward(conv, x))
    if residual is not None:
        y = y + residual
    return torch.nn.functional.relu(y, inplace=True) if relu else y


class ConvT2(torch.autograd.Function):
    """y = ConvTranspose3d(k = 2, s = 2)(x) for channels-last x (D,H,W,Cin) and the torch-layout weight (Cin, Cout, 2,2,2) -- the up-blocks
    of mmdet3d/models/necks/imvoxelnet.py:233-260.  Forward: the transposed form of the inference kernel (each input voxel writes its
    2x2x2 block).  The data gradient is the stride-2, unpadded 2x2x2 CONVOLUTION of dy with the very same tensor read as a Conv3d weight
    (out = Cin, in = Cout, no flip); the weight gradient is that convolution's weight gradient with the roles of x and dy swapped."""

    @staticmethod
    def forward(ctx, x, weight):
        w = weight.detach()
        xc = x.detach().contiguous()                            # the backward kernels index it densely
        ctx.save_for_backward(xc, w)
        pk = dict(w=C.pack_weight(w, True), scale=None, shift=None, cout=int(w.shape[1]), cin=int(w.shape[0]), ksize=2, stride=2, transposed=True,
                  kernel=(2, 2, 2), strides=(2, 2, 2), pads=(0, 0, 0), ndim=3, arith="bf16x3")
        return C.conv3d_ndhwc(xc, pk)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous().float()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _conv(g, _train_pack(w, (2, 2, 2), False, 2, (0, 0, 0)))
        if ctx.needs_input_grad[1]:
            dw = weight_grad(g, x, (2, 2, 2), 2, (0, 0, 0))
        return dx, dw


def eligible_transposed(conv: nn.Module, x: Tensor) -> bool:
    return (isinstance(conv, nn.ConvTranspose3d) and C.train_arithmetic() in ("bf16x3", "bf16", "f16x2") and x.is_cuda and x.dtype == torch.float32 and tuple(conv.kernel_size) == (2, 2, 2)
            and tuple(conv.stride) == (2, 2, 2) and tuple(conv.padding) == (0, 0, 0) and tuple(conv.output_padding) == (0, 0, 0)
            and tuple(conv.dilation) == (1, 1, 1) and conv.groups == 1 and conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0)


def _scene(x: Tensor, b: int) -> Tensor:
    """``x[b]``; for a batch of one as a VIEW op (squeeze): the backward of a select allocates a zero tensor of the batch's shape and copies the
    gradient into its slice -- two passes over every 3D convolution's input gradient (measured: 9 fills + 9 copies of 26 MB per training step)."""
    return x.squeeze(0) if x.shape[0] == 1 else x[b]


def conv_forward(conv: nn.Module, x: Tensor) -> Tensor:
    """``conv(x)`` for a logical (B,C,...) tensor: eligible layers run on the MFMA kernels under autograd (channels-last memory in,
    channels-last memory out, logical shape unchanged); everything else goes to the module itself."""
    if torch.is_grad_enabled() and eligible_transposed(conv, x):
        outs = []
        for b in range(x.shape[0]):
            xb = C.carry_amax(x, _scene(x, b).permute(1, 2, 3, 0))      # (one scene of the batch: max |x[b]| <= max |x|, an upper bound is all the scale needs)
            outs.append(ConvT2.apply(xb if xb.is_contiguous() else xb.contiguous(), conv.weight).permute(3, 0, 1, 2))
        y = outs[0].unsqueeze(0) if len(outs) == 1 else torch.stack(outs)
        return y if conv.bias is None else y + conv.bias.view(1, -1, 1, 1, 1)
    if not (torch.is_grad_enabled() and eligible(conv, x)):
        return conv(x)
    three_d = isinstance(conv, nn.Conv3d)
    outs = []
    if three_d:
        for b in range(x.shape[0]):                                           # one scene at a time: the kernel's depth axis is X
            xb = C.carry_amax(x, _scene(x, b).permute(1, 2, 3, 0))      # (one scene of the batch: max |x[b]| <= max |x|, an upper bound is all the scale needs)
            y = ConvS1.apply(xb if xb.is_contiguous() else xb.contiguous(), conv.weight, conv.stride[0])
            outs.append(y.permute(3, 0, 1, 2))
        y = outs[0].unsqueeze(0) if len(outs) == 1 else torch.stack(outs)
    else:
        xb = x.permute(0, 2, 3, 1)
        y = ConvS1.apply(xb if xb.is_contiguous() else xb.contiguous(), conv.weight, conv.stride[0]).permute(0, 3, 1, 2)
    if conv.bias is not None:
        y = y + conv.bias.view(1, -1, *([1] * (y.dim() - 2)))
    return y


def conv_forward_shared(convs: Sequence[nn.Module], x: Tensor):
    """Several convolutions reading the same input (the head's centerness / regression / class layers,
    mmdet3d/models/dense_heads/imvoxel_head_v2.py:444-449) as ONE convolution over their concatenated output channels -- one forward,
    one data gradient and one weight gradient launch instead of three each.  Returns the per-layer outputs (biases added)."""
    if not (torch.is_grad_enabled() and all(eligible(c, x) and c.stride[0] == 1 for c in convs) and len({tuple(c.kernel_size) for c in convs}) == 1):
        return [c(x) for c in convs]
    w = torch.cat([c.weight for c in convs], dim=0)
    outs = []
    for b in range(x.shape[0]):
        xb = C.carry_amax(x, _scene(x, b).permute(1, 2, 3, 0))      # (one scene of the batch: max |x[b]| <= max |x|, an upper bound is all the scale needs)
        outs.append(ConvS1.apply(xb if xb.is_contiguous() else xb.contiguous(), w).permute(3, 0, 1, 2))
    y = outs[0].unsqueeze(0) if len(outs) == 1 else torch.stack(outs)
    res, o = [], 0
    for c in convs:
        part = y[:, o:o + c.out_channels]
        if c.bias is not None:
            part = part + c.bias.view(1, -1, 1, 1, 1)
        res.append(part)
        o += c.out_channels
    return res


BN_KERNELS = True     # BatchNorm on batch statistics (+ ReLU, + residual) of the 3D neck on csrc/bn_kernels.hip; False: F.batch_norm + separate passes


def bn_rows_ok(bn: nn.Module, rows: Tensor) -> bool:
    """Training-mode BatchNorm with affine parameters over contiguous fp32 GPU rows whose width the kernels' lane layout takes."""
    c = rows.shape[-1]
    return (BN_KERNELS and bn.training and bn.affine and rows.is_cuda and rows.dtype == torch.float32 and rows.is_contiguous() and torch.is_grad_enabled()
            and c % 4 == 0 and c // 4 <= 1024 and 1024 % (c // 4) == 0 and rows.numel() // c >= 2)


def _bump_version(*tensors) -> None:
    """The library wrote these tensors in place through raw pointers: advance their version counters as an in-place torch op would have."""
    bump = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
    if bump is not None:
        bump(list(tensors), [t._version + 1 for t in tensors])
    else:
        for t in tensors:
            t.add_(0)


class BatchNormRows(torch.autograd.Function):
    """y = relu?(batch_norm(x) (+ residual)) over (N, C) rows on batch statistics: csrc/bn_kernels.hip (three launches forward, three backward; the
    library path is a statistics kernel, a normalise pass, a ReLU pass, an add pass, and their four backward counterparts).  In the fp16-pair mode the
    apply passes leave max |y| / max |dx| behind for the convolutions that read them."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, residual):
        from ctypes import c_void_p
        from . import _lib
        lib = _lib.load()
        xc = x.detach()
        n, c = xc.shape
        res = None if residual is None else residual.detach().contiguous()
        y = torch.empty_like(xc)
        mean, invstd = torch.empty(c, dtype=torch.float32, device=x.device), torch.empty(c, dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib.ndet_bn_workspace_floats(n, c)), dtype=torch.float32, device=x.device)
        slot = C.AMAX.take(x.device) if C.train_arithmetic() == "f16x2" else None
        p = lambda t: c_void_p(0 if t is None else t.data_ptr())
        _lib.check(lib.ndet_bn_train_forward(p(xc), n, c, p(weight.detach()), p(bias.detach()), p(running_mean), p(running_var), float(momentum), float(eps), p(res),
                                             int(bool(relu)), p(y), p(mean), p(invstd), p(slot), p(ws), c_void_p(raw_stream(x.device))),
                   "bn_train_forward")
        if slot is not None:
            C._tag_amax(y, slot)
        if running_mean is not None:
            _bump_version(running_mean, running_var)       # written through raw pointers: the eval-mode packs key on the buffers' versions (conv3d.bn_affine)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.save_for_backward(xc, y if relu else None, weight.detach(), mean, invstd)
        return y

    @staticmethod
    def backward(ctx, g):
        from ctypes import c_void_p
        from . import _lib
        lib = _lib.load()
        x, y, w, mean, invstd = ctx.saved_tensors
        g = g.contiguous()
        n, c = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[8]) else None
        dgamma, dbeta = torch.empty(c, dtype=torch.float32, device=x.device), torch.empty(c, dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib.ndet_bn_workspace_floats(n, c)), dtype=torch.float32, device=x.device)
        slot = C.AMAX.take(x.device) if C.train_arithmetic() == "f16x2" else None
        p = lambda t: c_void_p(0 if t is None else t.data_ptr())
        _lib.check(lib.ndet_bn_train_backward(p(g), p(x), p(y), n, c, p(w), p(mean), p(invstd), int(ctx.relu), p(dx), p(dres), p(dgamma), p(dbeta), p(slot), p(ws),
                                              c_void_p(raw_stream(x.device))), "bn_train_backward")
        if slot is not None:
            C._tag_amax(dx, slot)
        return dx, dgamma, dbeta, None, None, None, None, None, dres
