"""A13: 3-level 3D ResNet-FPN over the voxel volume -- mirror of ``FastIndoorImVoxelNeck`` and
``BasicBlock3dV2`` (mmdet3d/models/necks/imvoxelnet.py:8-67, 233-260), same state-dict keys.
Dense contraction: library (MIOpen) convolutions for now; measured rates in DESIGN.md."""
from __future__ import annotations

from torch import nn

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
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        idt = self.downsample(x) if self.stride != 1 else x
        return self.relu(out + idt)


def _conv_bn_relu(cin, cout):
    return nn.Sequential(nn.Conv3d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm3d(cout), nn.ReLU(inplace=True))


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
        downs = []
        for i in range(self.n_scales):
            x = getattr(self, f"down_layer_{i}")(x)
            downs.append(x)
        outs = []
        for i in range(self.n_scales - 1, -1, -1):
            if i < self.n_scales - 1:
                x = downs[i] + getattr(self, f"up_block_{i + 1}")(x)
            outs.append(getattr(self, f"out_block_{i}")(x))
        return outs[::-1]
