"""NeRF radiance-field MLP with the reference's parameter names.

Mirror of mmdet3d/models/model_utils/nerf_mlp.py (MLP :11-90, NerfMLP :103-161, SinusoidalEncoder
:164-197, VanillaNeRFRadianceField :200-234): same constructor arguments, same state-dict keys
(``posi_encoder.scales``, ``mlp.base.hidden_layers.N.*``, ``mlp.sigma_layer.output_layer.*``,
``mlp.bottleneck_layer.output_layer.*``, ``mlp.rgb_layer.*``) so released checkpoints load, same
math.  On the GPU the encoder + concat run in one HIP kernel (ops.posenc_concat); the dense
layers are library GEMMs (hipBLASLt through torch).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn


class LayerStack(nn.Module):
    """``depth`` x (Linear + ReLU) of width ``width``; after hidden layer i with i % skip == 0 (i > 0) the
    stack input is concatenated back; optional final Linear to ``out_dim``.  Xavier-uniform weights,
    zero biases (nerf_mlp.py:60-78)."""

    def __init__(self, in_dim: int, out_dim: Optional[int], depth: int, width: int, skip: Optional[int]):
        super().__init__()
        self.in_dim, self.skip, self.depth = in_dim, skip, depth
        self.hidden_layers = nn.ModuleList()
        feed = in_dim
        for i in range(depth):
            self.hidden_layers.append(nn.Linear(feed, width))
            feed = width + in_dim if self._rejoin(i) else width
        self.out_features = feed if out_dim is None else out_dim
        if out_dim is not None:
            self.output_layer = nn.Linear(feed, out_dim)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

    def _rejoin(self, i: int) -> bool:
        return self.skip is not None and i > 0 and i % self.skip == 0

    def forward(self, x):
        from .autograd import linear_rows_train      # training on many rows: split weight / bias gradients (autograd.LinearRows)
        x0 = x
        for i, layer in enumerate(self.hidden_layers):
            x = linear_rows_train(x, layer, relu=True)
            if self._rejoin(i):
                x = torch.cat([x, x0], dim=-1)
        if hasattr(self, "output_layer"):
            x = linear_rows_train(x, self.output_layer)
        return x


class NerfMLP(nn.Module):
    """Trunk + sigma head + (bottleneck, view-conditioned rgb head).  nerf_mlp.py:103-161."""

    def __init__(self, input_dim: int, condition_dim: int, feature_dim: int = 0, net_depth: int = 8, net_width: int = 256,
                 skip_layer: int = 4, net_depth_condition: int = 1, net_width_condition: int = 128):
        super().__init__()
        self.base = LayerStack(input_dim + feature_dim, None, net_depth, net_width, skip_layer)
        hid = self.base.out_features
        self.sigma_layer = LayerStack(hid, 1, 0, net_width, None)
        if condition_dim > 0:
            self.bottleneck_layer = LayerStack(hid, net_width, 0, net_width, None)
            self.rgb_layer = LayerStack(net_width + condition_dim, 3, net_depth_condition, net_width_condition, None)
        else:
            self.rgb_layer = LayerStack(hid, 3, 0, net_width, None)

    def _trunk(self, x, features):
        return self.base(x if features is None else torch.cat([x, features], dim=-1))

    def query_density(self, x, features=None):
        return self.sigma_layer(self._trunk(x, features))

    def forward(self, x, condition=None, features=None):
        h = self._trunk(x, features)
        raw_sigma = self.sigma_layer(h)
        if condition is None:
            return self.rgb_layer(h), raw_sigma
        if condition.shape[:-1] != h.shape[:-1]:  # one view direction per ray, broadcast over its samples
            r, d = condition.shape
            condition = condition.view([r] + [1] * (h.dim() - 2) + [d]).expand(*h.shape[:-1], d)
        return self.rgb_layer(torch.cat([self.bottleneck_layer(h), condition], dim=-1)), raw_sigma


class SinusoidalEncoder(nn.Module):
    """``[x | sin(2^k x) | sin(2^k x + pi/2)]`` for k in [min_deg, max_deg).  nerf_mlp.py:164-197."""

    def __init__(self, x_dim: int, min_deg: int, max_deg: int, use_identity: bool = True):
        super().__init__()
        self.x_dim, self.min_deg, self.max_deg, self.use_identity = x_dim, min_deg, max_deg, use_identity
        self.register_buffer("scales", torch.tensor([2 ** i for i in range(min_deg, max_deg)]))

    @property
    def latent_dim(self) -> int:
        return (int(self.use_identity) + 2 * (self.max_deg - self.min_deg)) * self.x_dim

    def forward(self, x):
        if self.max_deg == self.min_deg:
            return x
        xb = (x[..., None, :] * self.scales[:, None]).reshape(*x.shape[:-1], -1)
        enc = torch.sin(torch.cat([xb, xb + 0.5 * math.pi], dim=-1))
        return torch.cat([x, enc], dim=-1) if self.use_identity else enc


class VanillaNeRFRadianceField(nn.Module):
    """nerf_mlp.py:200-234."""

    def __init__(self, net_depth: int = 8, net_width: int = 256, skip_layer: int = 4, feature_dim: int = 0,
                 net_depth_condition: int = 1, net_width_condition: int = 128):
        super().__init__()
        self.posi_encoder = SinusoidalEncoder(3, 0, 10, True)
        self.view_encoder = SinusoidalEncoder(3, 0, 4, True)
        self.mlp = NerfMLP(self.posi_encoder.latent_dim, self.view_encoder.latent_dim, feature_dim, net_depth, net_width,
                           skip_layer, net_depth_condition, net_width_condition)

    def query_density(self, x, features=None):
        return F.relu(self.mlp.query_density(self.posi_encoder(x), features))

    def raw_sigma_from_rows(self, rows):
        """sigma-MLP on pre-assembled ``[posenc | features]`` rows (ops.posenc_concat); no relu."""
        return self.mlp.sigma_layer(self.mlp.base(rows))

    def hip_trunk_ok(self) -> bool:
        """The shipped architecture (4 hidden layers, input re-joined after the last one, widths % 32 == 0): its trunk runs
        as 4 launches of the MFMA kernel + one fused sigma/alpha kernel."""
        b = self.mlp.base
        return (b.depth == 4 and b.skip == 3 and all(l.out_features % 32 == 0 for l in b.hidden_layers)
                and not hasattr(b, "output_layer") and len(self.mlp.sigma_layer.hidden_layers) == 0)

    def alpha_from_points(self, points, global_feat):
        """Inference: voxel points (3,N)/(3,X,Y,Z) + conditioning rows (N,F) -> alpha (N) = 1-exp(-relu(sigma)), all in
        hand-written kernels: posenc+concat (zero-padded to a multiple of 32), 4 x [Linear+ReLU] on the fp32 matrix cores,
        fused sigma head.  nerf_mlp.py:224-227 + nerfdet.py:255-257."""
        from . import ops
        from .conv3d import linear_rows, packed_linear
        b = self.mlp.base
        n_in = 63 + global_feat.shape[1]
        width = (n_in + 31) // 32 * 32
        out = self.mlp.sigma_layer.output_layer
        if self.fused_ok(width):      # the shipped architecture: the whole chain in one launch, the 256-wide rows never leave the CU
            return ops.point_mlp_alpha(points, global_feat, self._fused_layers(width), out.weight, out.bias)
        rows = ops.posenc_concat(points, global_feat, pad_to=width)
        h = rows
        for i, lin in enumerate(b.hidden_layers):
            h = linear_rows(h, packed_linear(lin, pad_in_to=width if i == 0 else 0), relu=1)
        return ops.sigma_head(h, rows, n_in, out.weight, out.bias)

    FUSED_MLP = True     # csrc/point_mlp_kernels.hip (False: the layer-by-layer launches, kept for other widths and as the comparison in the tests)

    def fused_ok(self, width: int) -> bool:
        from . import conv3d
        b = self.mlp.base
        return (self.FUSED_MLP and conv3d.ARITHMETIC in ("f16x2", "bf16x3") and width <= 256 and b.hidden_layers[0].in_features <= width
                and all(l.out_features == 256 for l in b.hidden_layers))

    def _fused_layers(self, width: int):
        """(fp16-pair planes, 1 / scale, bias) of the four hidden layers, cached with the packs (rebuilt when a parameter changes)."""
        from .conv3d import packed_linear, split_planes_f16
        layers = []
        for i, lin in enumerate(self.mlp.base.hidden_layers):
            pk = packed_linear(lin, pad_in_to=width if i == 0 else 0)
            planes, winv = split_planes_f16(pk)
            layers.append((planes, winv, pk["shift"]))
        return layers

    def forward_rows_hip(self, x, condition, features):
        """Inference form of :meth:`forward` for the ray branch (A10, nerf_mlp.py:146-161,229-234): the dense layers of the trunk,
        the bottleneck and the view-conditioned colour layer run on the hand-written MFMA convolution kernel as 1x1
        convolutions over the (R*S) sample rows; the 389->1 and 128->3 output layers are per-row dot products."""
        from . import ops
        from .conv3d import linear_rows, packed_linear
        lead = x.shape[:-1]
        n = int(torch.tensor(lead).prod())
        b = self.mlp.base
        n_in = 63 + features.shape[-1]
        width = (n_in + 31) // 32 * 32
        pts = x.reshape(n, 3).t().contiguous()
        rows = ops.posenc_concat(pts, features.reshape(n, -1), pad_to=width)       # (the bottleneck layer below re-joins them)
        out = self.mlp.sigma_layer.output_layer
        if self.fused_ok(width):      # trunk + sigma layer in one launch, the 256-wide rows written once (for the bottleneck layer)
            _, raw_sigma, h = ops.point_mlp_alpha(pts, features.reshape(n, -1), self._fused_layers(width), out.weight, out.bias, want_raw=True, want_h=True)
        else:
            h = rows
            for i, lin in enumerate(b.hidden_layers):
                h = linear_rows(h, packed_linear(lin, pad_in_to=width if i == 0 else 0), relu=1)
            _, raw_sigma = ops.sigma_head(h, rows, n_in, out.weight, out.bias, want_raw=True)
        wide = (h.shape[1] + n_in + 31) // 32 * 32
        hc = torch.zeros((n, wide), dtype=torch.float32, device=h.device)
        hc[:, :h.shape[1]] = h
        hc[:, h.shape[1]:h.shape[1] + n_in] = rows[:, :n_in]
        bott = linear_rows(hc, packed_linear(self.mlp.bottleneck_layer.output_layer, pad_in_to=wide), relu=0)
        cond = self.view_encoder(condition)
        if cond.shape[:-1] != lead:   # one view direction per ray, broadcast over its samples
            cond = cond.view([cond.shape[0]] + [1] * (len(lead) - 1) + [cond.shape[-1]]).expand(*lead, cond.shape[-1])
        cw = (bott.shape[1] + cond.shape[-1] + 31) // 32 * 32
        rc = torch.zeros((n, cw), dtype=torch.float32, device=h.device)
        rc[:, :bott.shape[1]] = bott
        rc[:, bott.shape[1]:bott.shape[1] + cond.shape[-1]] = cond.reshape(n, -1)
        rl = self.mlp.rgb_layer
        hid = linear_rows(rc, packed_linear(rl.hidden_layers[0], pad_in_to=cw), relu=1)
        rgb = F.linear(hid, rl.output_layer.weight, rl.output_layer.bias)
        return torch.sigmoid(rgb).view(*lead, 3), F.relu(raw_sigma).view(*lead, 1)

    def forward(self, x, condition=None, features=None):
        if (not torch.is_grad_enabled() and x.is_cuda and condition is not None and features is not None and self.hip_trunk_ok()
                and hasattr(self.mlp, "bottleneck_layer") and len(self.mlp.rgb_layer.hidden_layers) == 1
                and len(self.mlp.bottleneck_layer.hidden_layers) == 0 and self.mlp.rgb_layer.hidden_layers[0].out_features % 32 == 0):
            return self.forward_rows_hip(x, condition, features)
        x = self.posi_encoder(x)
        if condition is not None:
            condition = self.view_encoder(condition)
        rgb, sigma = self.mlp(x, condition=condition, features=features)
        return torch.sigmoid(rgb), F.relu(sigma)
