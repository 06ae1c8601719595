"""A13: 3-level 3D ResNet-FPN over the voxel volume -- mirror of ``FastIndoorImVoxelNeck`` and
``BasicBlock3dV2`` (mmdet3d/models/necks/imvoxelnet.py:8-67, 233-260), same state-dict keys.

Inference (``eval()`` on the GPU) runs the hand-written fp32-MFMA implicit-GEMM convolution of csrc/conv3d_kernels.hip
with BatchNorm / ReLU / residual adds fused into its epilogue, channels-last end to end.  Training mode (batch
statistics, autograd) goes through the vendor library modules below."""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F

from .conv3d import carry_amax, conv3d_ndhwc, packed, to_ndhwc
from .conv_train import conv_forward
from .registry import NECKS


class BasicBlock3dV2(nn.Module):
    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv3d(in_channels, out_channels, 3, stride, 1, bias=False)
        self.norm1 = nn.BatchNorm3d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv3d(out_channels, out_channels, 3, 1, 1, bias=False)
        self.norm2 = nn.BatchNorm3d(out_channels)
        if stride != 1:
            self.downsample = nn.Sequential(nn.Conv3d(in_channels, out_channels, 1, stride, bias=False),
                                            nn.BatchNorm3d(out_channels))

    def forward(self, x):
        """Library form (training: BatchNorm on batch statistics, autograd).  On the GPU the stride-1 convolutions still run on the
        MFMA kernels, forward and backward (nerfdet_amd/conv_train.py)."""
        out = _bn(self.norm1, conv_forward(self.conv1, x), relu=True)
        idt = _run(self.downsample, x) if self.stride != 1 else x
        return _bn(self.norm2, conv_forward(self.conv2, out), relu=True, residual=idt)       # relu(norm2(.) + identity), imvoxelnet.py:60-66

    def forward_ndhwc(self, x):
        """x (D,H,W,C) -> (D',H',W',C'); eval-mode BN folded, ``relu(bn2(conv2(.)) + identity)`` in one epilogue."""
        y = conv3d_ndhwc(x, packed([self.conv1], self.norm1), relu=1)
        idt = x if self.stride == 1 else conv3d_ndhwc(x, packed([self.downsample[0]], self.downsample[1]), amax=False)
        return conv3d_ndhwc(y, packed([self.conv2], self.norm2), residual=idt, relu=1)


def _bn(bn: nn.BatchNorm3d, x, relu: bool = False, residual=None):
    """``relu?(bn(x) (+ residual))`` for a logical (B,C,D,H,W) tensor held in channels-last memory, evaluated on its (voxels, C) row view: the 2D
    form reduces over the same elements per channel (same statistics, same running-average update), and neither it nor its backward leaves
    channels-last memory -- the 5D library path returns NCDHW memory, forward and backward, and every convolution after it then starts with a
    transposing copy.  In training on the GPU the whole expression is csrc/bn_kernels.hip (conv_train.BatchNormRows)."""
    def tail(y):
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    rows = x.permute(0, 2, 3, 4, 1) if (x.is_cuda and x.dim() == 5) else None
    if rows is None or not rows.is_contiguous():
        return tail(bn(x))
    factor = 0.0 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            factor = 1.0 / float(bn.num_batches_tracked)
    training = bn.training or (bn.running_mean is None and bn.running_var is None)
    from .conv_train import BatchNormRows, bn_rows_ok
    flat = rows.reshape(-1, x.shape[1])
    res_rows = None
    if residual is not None:
        rr = residual.permute(0, 2, 3, 4, 1)
        res_rows = rr.reshape(-1, x.shape[1]) if rr.is_contiguous() else None
    if bn_rows_ok(bn, flat) and (residual is None or res_rows is not None):
        track = bn.training and bn.track_running_stats
        y = BatchNormRows.apply(flat, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None, factor, bn.eps, relu, res_rows)
        return carry_amax(y, y.view(rows.shape).permute(0, 4, 1, 2, 3))
    y = F.batch_norm(flat, bn.running_mean if (not bn.training or bn.track_running_stats) else None,
                     bn.running_var if (not bn.training or bn.track_running_stats) else None, bn.weight, bn.bias, training, factor, bn.eps)
    return tail(y.view(rows.shape).permute(0, 4, 1, 2, 3))


def _conv_bn_relu(cin, cout):
    return nn.Sequential(nn.Conv3d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm3d(cout), nn.ReLU(inplace=True))


def _run(seq: nn.Sequential, x):
    """``seq(x)`` with its convolutions routed through :func:`conv_forward` and every BatchNorm + ReLU pair through one :func:`_bn`."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
            x = conv_forward(m, x)
        elif isinstance(m, nn.BatchNorm3d):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = _bn(m, x, relu=fuse)
            i += int(fuse)
        else:
            x = m(x)
        i += 1
    return x


@NECKS.register_module()
class FastIndoorImVoxelNeck(nn.Module):
    def __init__(self, in_channels, n_blocks, out_channels):
        super().__init__()
        self.n_scales = len(n_blocks)
        c = in_channels
        for i, nb in enumerate(n_blocks):
            stride = 1 if i == 0 else 2
            blocks = []
            for b in range(nb):
                if b == 0 and stride != 1:
                    blocks.append(BasicBlock3dV2(c, c * 2, stride))
                    c *= 2
                else:
                    blocks.append(BasicBlock3dV2(c, c))
            setattr(self, f"down_layer_{i}", nn.Sequential(*blocks))
            if i > 0:
                setattr(self, f"up_block_{i}", nn.Sequential(
                    nn.ConvTranspose3d(c, c // 2, 2, 2, bias=False), nn.BatchNorm3d(c // 2), nn.ReLU(inplace=True),
                    nn.Conv3d(c // 2, c // 2, 3, 1, 1, bias=False), nn.BatchNorm3d(c // 2), nn.ReLU(inplace=True)))
            setattr(self, f"out_block_{i}", _conv_bn_relu(c, out_channels))

    def init_weights(self):
        pass

    def forward(self, x):
        # the MFMA kernel steps K by 32 input channels; every shipped config (256/512/1024) qualifies
        if x.is_cuda and not self.training and not torch.is_grad_enabled() and x.shape[1] % 32 == 0:
            return self.forward_hip(x)
        return self.forward_library(x)

    def forward_hip(self, x):
        """(B,C,X,Y,Z) -> 3 x (B,128,X_i,Y_i,Z_i), logical NCDHW views of channels-last results."""
        per_level = [[] for _ in range(self.n_scales)]
        for b in range(x.shape[0]):
            t = to_ndhwc(x[b].float())
            downs = []
            for i in range(self.n_scales):
                for blk in getattr(self, f"down_layer_{i}"):
                    t = blk.forward_ndhwc(t)
                downs.append(t)
            for i in range(self.n_scales - 1, -1, -1):
                if i < self.n_scales - 1:
                    up = getattr(self, f"up_block_{i + 1}")
                    t = conv3d_ndhwc(t, packed([up[0]], up[1]), relu=1)
                    # x = down_outs[i] + up(x): ReLU first, then the skip add (imvoxelnet.py:30-31)
                    t = conv3d_ndhwc(t, packed([up[3]], up[4]), residual=downs[i], relu=2)
                ob = getattr(self, f"out_block_{i}")
                o = conv3d_ndhwc(t, packed([ob[0]], ob[1]), relu=1)
                per_level[i].append(carry_amax(o, o.permute(3, 0, 1, 2)))
        # (one scene: the views keep the maximum the convolution left behind, so the head's first launch needs no pass of its own)
        return [carry_amax(lv[0], lv[0].unsqueeze(0)) if len(lv) == 1 else torch.stack(lv) for lv in per_level]

    def forward_library(self, x):
        downs = []
        for i in range(self.n_scales):
            x = getattr(self, f"down_layer_{i}")(x)
            downs.append(x)
        outs = []
        for i in range(self.n_scales - 1, -1, -1):
            if i < self.n_scales - 1:
                x = downs[i] + _run(getattr(self, f"up_block_{i + 1}"), x)
            outs.append(_run(getattr(self, f"out_block_{i}"), x))
        return outs[::-1]
