"""2D backbone + FPN in plain PyTorch-ROCm (vendor-library convolutions: MIOpen).

mmdet 2.10.0's ``ResNet`` and ``FPN`` are third-party and absent from the reference tree; these are
restatements of their documented behaviour with the same state-dict key names
(torchvision-layout ``conv1/bn1/layer{1-4}.{i}.{conv,bn}{1-3}/downsample.{0,1}``;
``lateral_convs.{i}.conv`` / ``fpn_convs.{i}.conv``) so torchvision / released checkpoints load.
Not part of the hand-written hot path; they feed it channels-last features."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .conv3d import (bn_relu_maxpool_nhwc, bottleneck_ok, carry_amax, chain_ok, conv2d_bottleneck_nhwc, conv2d_chain_nhwc, conv2d_nhwc, packed,
                     stem_conv_bn_relu_maxpool, stem_ok)
from .conv_train import conv_bn_act, conv_forward
from .registry import BACKBONES, NECKS


def _nhwc(x):
    """logical (N,C,H,W) -> contiguous (N,H,W,C) (free for channels-last memory)."""
    y = x.permute(0, 2, 3, 1)
    return carry_amax(x, y if y.is_contiguous() else y.contiguous())


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)  # style='pytorch': stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        # training on the GPU: the stride-1 convolutions run on the MFMA kernels, forward and backward (nerfdet_amd/conv_train.py)
        idt = x if self.downsample is None else conv_bn_act(self.downsample[0], self.downsample[1], x, relu=False)
        out = conv_bn_act(self.conv1, self.bn1, x)
        out = conv_bn_act(self.conv2, self.bn2, out)
        return conv_bn_act(self.conv3, self.bn3, out, relu=True, residual=idt)

    def forward_nhwc(self, x):
        """Inference form on (N,H,W,C): every conv carries its frozen BatchNorm, ReLU and (last one) the residual
        add in the MFMA kernel's epilogue -- 4 kernels instead of 4 convs + 10 elementwise passes."""
        pk1, pk2, pk3 = packed([self.conv1], self.bn1), packed([self.conv2], self.bn2), packed([self.conv3], self.bn3)
        pkd = None if self.downsample is None else packed([self.downsample[0]], self.downsample[1])
        if bottleneck_ok(x, pk1, pk2, pk3, pkd):          # stage 1: the whole block in one launch, the 64-channel intermediates never leave the CU
            return conv2d_bottleneck_nhwc(x, pk1, pk2, pk3, pkd)
        y = conv2d_nhwc(x, pk1, relu=1)
        idt = x if pkd is None else conv2d_nhwc(x, pkd, amax=False)
        if chain_ok(pk2, pk3):          # stages 1 / 2: the 64- / 128-channel intermediate never leaves the CU
            return conv2d_chain_nhwc(y, pk2, pk3, residual=idt, relu=1)
        y = conv2d_nhwc(y, pk2, relu=1)
        return conv2d_nhwc(y, pk3, residual=idt, relu=1)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1, norm_cfg=None, norm_eval=True,
                 style="pytorch", **kw):
        super().__init__()
        assert style == "pytorch" and depth in self.arch
        self.out_indices, self.frozen_stages, self.norm_eval = tuple(out_indices), frozen_stages, norm_eval
        self.bn_requires_grad = True if norm_cfg is None else norm_cfg.get("requires_grad", True)
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for i, n in enumerate(self.arch[depth][:num_stages]):
            planes, stride = 64 * 2 ** i, 1 if i == 0 else 2
            blocks = []
            for b in range(n):
                ds = None
                if b == 0 and (stride != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
                blocks.append(Bottleneck(inplanes, planes, stride if b == 0 else 1, ds))
                inplanes = planes * 4
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.num_stages = num_stages
        if not self.bn_requires_grad:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    for p in m.parameters():
                        p.requires_grad = False
        self._freeze()

    def _freeze(self):
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f"layer{i}")
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        """``torchvision://resnet50`` is a network fetch (unavailable offline); a local path loads."""
        if isinstance(pretrained, str) and not pretrained.startswith(("torchvision://", "http")):
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, map_location="cpu", strict=False)
            return
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def train(self, mode=True):
        super().train(mode)
        self._freeze()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    def _bns(self, prefix_only: bool = False):
        """The BatchNorm modules (of the frozen prefix), listed once: walking ``modules()`` on every call cost 0.2 ms of host time in
        front of the step's first launch."""
        key = "_bn_prefix" if prefix_only else "_bn_all"
        hit = self.__dict__.get(key)
        if hit is None:
            if prefix_only:
                hit = [self.bn1] + [m for i in range(1, self.frozen_stages + 1) for m in getattr(self, f"layer{i}").modules()
                                    if isinstance(m, nn.BatchNorm2d)]
            else:
                hit = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
            self.__dict__[key] = hit
        return hit

    def forward(self, x):
        bn_frozen = not any(m.training for m in self._bns())
        if x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and bn_frozen and self.use_hip:
            return self.forward_hip(x)
        if (x.is_cuda and x.dtype == torch.float32 and self.use_hip and self.frozen_stages >= 1 and not x.requires_grad
                and self._frozen_prefix_in_eval()):
            return self.forward_frozen_prefix(x)
        return self.forward_library(x)

    def _frozen_prefix_in_eval(self) -> bool:
        return not any(m.training for m in self._bns(prefix_only=True))

    def forward_frozen_prefix(self, x):
        """Training: the frozen stem and stages (``frozen_stages``, eval-mode BatchNorm, no parameter gradients, no gradient
        into the images) run on the fused inference kernels without autograd bookkeeping; the trainable stages follow on the
        library modules."""
        outs = []
        with torch.no_grad():
            y = self._stem(x, training=True)
            for i in range(self.frozen_stages):
                for blk in getattr(self, f"layer{i + 1}"):
                    y = blk.forward_nhwc(y)
                if i in self.out_indices:
                    outs.append(y.permute(0, 3, 1, 2))
            y = y.permute(0, 3, 1, 2)   # logical NCHW on channels-last memory: what the library convolutions like best
        for i in range(self.frozen_stages, self.num_stages):
            y = getattr(self, f"layer{i + 1}")(y)
            if i in self.out_indices:
                outs.append(y)
        return tuple(outs)

    def _stem(self, x, training: bool = False):
        """conv1 + bn1 (eval) + ReLU + max-pool -> (N,H/4,W/4,64) channels-last: one launch (csrc/stem_kernels.hip); with the exact
        fp32-MFMA arithmetic selected, the vendor library's convolution followed by the fused BN + ReLU + pool pass.  (The bf16 training
        step keeps the library stem it was validated with: tests/test_train_gpu.py's bf16 gradient bands were measured on it.)"""
        from . import conv3d as _c
        if stem_ok(self.conv1, self.bn1, x) and not (training and _c.ARITHMETIC == "bf16"):
            return stem_conv_bn_relu_maxpool(x, self.conv1, self.bn1)
        return bn_relu_maxpool_nhwc(_nhwc(self.conv1(x.contiguous(memory_format=torch.channels_last))), self.bn1)

    use_hip = True  # inference on the GPU: bottlenecks through the fused MFMA convolution (csrc/conv3d_kernels.hip)

    def forward_hip(self, x):
        x = self._stem(x)
        outs = []
        for i in range(self.num_stages):
            for blk in getattr(self, f"layer{i + 1}"):
                x = blk.forward_nhwc(x)
            if i in self.out_indices:
                outs.append(carry_amax(x, x.permute(0, 3, 1, 2)))  # logical NCHW view of channels-last memory
        return tuple(outs)

    def forward_library(self, x):
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)      # the layout the library's convolutions are fastest in
        x = F.relu(self.bn1(self.conv1(x)), inplace=True)
        x = F.max_pool2d(x, 3, 2, 1)
        outs = []
        for i in range(self.num_stages):
            x = getattr(self, f"layer{i + 1}")(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class _Conv(nn.Module):
    """mmcv ``ConvModule`` without norm/activation: keeps the ``.conv`` key level."""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding)

    def forward(self, x):
        return conv_forward(self.conv, x)


def _chain_pack(pk: dict, lin: nn.Linear):
    """(map_w (Cout, 32), map_b (32)) of conv3d.conv2d_nhwc's chained projection for ``lin`` behind the convolution packed as ``pk``: the
    convolution's own per-channel affine (bias / folded BatchNorm) is folded in, out = (acc * scale + shift) . Wm^T + bm.  Cached on the Linear."""
    store = lin.__dict__.setdefault("_ndet_chain", {})
    stamp = (lin.weight.data_ptr(), lin.weight._version, lin.bias.data_ptr(), lin.bias._version, id(pk),
             None if pk["scale"] is None else (pk["scale"].data_ptr(), pk["scale"]._version), None if pk["shift"] is None else (pk["shift"].data_ptr(), pk["shift"]._version))
    hit = store.get("pack")
    if hit is None or hit[0] != stamp:
        with torch.no_grad():
            wm, bm = lin.weight.detach().float(), lin.bias.detach().float()
            scale = pk["scale"] if pk["scale"] is not None else torch.ones(pk["cout"], device=wm.device)
            shift = pk["shift"] if pk["shift"] is not None else torch.zeros(pk["cout"], device=wm.device)
            map_w = (wm * scale.view(1, -1)).t().contiguous()
            map_b = (wm @ shift + bm).contiguous()
        hit = (stamp, (map_w, map_b), pk)          # (pk kept alive: its id is part of the stamp)
        store["pack"] = hit
    return hit[1]


@NECKS.register_module()
class FPN(nn.Module):
    """1x1 laterals, top-down nearest upsample-add, 3x3 output convs; no extra levels when
    num_outs == len(in_channels).  ``active_outs`` lets the detector skip output convs whose result it
    drops (nerfdet.py:142 keeps level 0 only) -- same numerics for the kept level."""

    def __init__(self, in_channels, out_channels, num_outs, **kw):
        super().__init__()
        assert num_outs == len(in_channels), "extra FPN levels are not used by the nerfdet configs"
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.lateral_convs = nn.ModuleList(_Conv(c, out_channels, 1) for c in in_channels)
        self.fpn_convs = nn.ModuleList(_Conv(out_channels, out_channels, 3, 1) for _ in in_channels)
        self.active_outs = None  # None = all
        self.__dict__["chain_linear"] = None   # set by the detector (not a submodule: the Linear stays the detector's): see forward_hip

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

    use_hip = True

    def forward(self, inputs):
        hip = self.use_hip and inputs[0].is_cuda and inputs[0].dtype == torch.float32 and not torch.is_grad_enabled()
        if hip and all(tuple(inputs[i].shape[2:]) == tuple((s + 1) // 2 for s in inputs[i - 1].shape[2:]) for i in range(1, len(inputs))):
            return self.forward_hip(inputs)
        lat = [l(x) for l, x in zip(self.lateral_convs, inputs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        act = range(len(lat)) if self.active_outs is None else self.active_outs
        return tuple(self.fpn_convs[i](lat[i]) if i in act else None for i in range(len(lat)))

    def forward_hip(self, inputs):
        """Coarse to fine: each lateral 1x1 conv adds the (nearest x2 upsampled) coarser lateral in its epilogue, so the
        top-down pathway costs no extra pass; then the 3x3 output convs of the active levels."""
        n = len(inputs)
        lat = [None] * n
        for i in range(n - 1, -1, -1):
            lat[i] = conv2d_nhwc(_nhwc(inputs[i]), packed([self.lateral_convs[i].conv]), residual=lat[i + 1] if i + 1 < n else None,
                                 residual_up2=i + 1 < n)
        act = range(n) if self.active_outs is None else self.active_outs
        outs = []
        for i in range(n):
            if i not in act:
                outs.append(None)
                continue
            pk = packed([self.fpn_convs[i].conv])
            lin = self.chain_linear if i == 0 else None
            if lin is not None and lin.in_features == pk["cout"] == 256 and lin.out_features == 32 and lin.weight.is_cuda:
                # the detector's 256 -> 32 feature mapping (nerfdet.py:194-197) rides in the level-0 output convolution's epilogue: the mapped map
                # travels on the output tensor as ``_ndet_feature_2d`` (logical (N, 32, H, W), channels-last memory)
                o, mapped = conv2d_nhwc(lat[i], pk, amax=False, chain=_chain_pack(pk, lin))
                o = o.permute(0, 3, 1, 2)
                if mapped is not None:
                    o._ndet_feature_2d = mapped.view(o.shape[0], o.shape[2], o.shape[3], 32).permute(0, 3, 1, 2)
                outs.append(o)
            else:
                outs.append(conv2d_nhwc(lat[i], pk, amax=False).permute(0, 3, 1, 2))
        return tuple(outs)
