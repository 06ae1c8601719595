"""Minimal depth-coordinate 3D box container -- only what the nerfdet path touches
(result wrapper of imvoxel_head_v2.py:544, ``gravity_center`` / ``tensor`` / ``volume`` used by
get_targets :457-470).  Semantics of mmdet3d/core/bbox/structures/base_box3d.py:37-64: stored as
(x, y, z_bottom, dx, dy, dz, yaw); a (N,6) input gets a zero yaw; ``origin`` re-bases the centre."""
from __future__ import annotations

import torch


class DepthInstance3DBoxes:
    def __init__(self, tensor, box_dim: int = 7, with_yaw: bool = True, origin=(0.5, 0.5, 0)):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device("cpu")
        t = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if t.numel() == 0:
            t = t.reshape((0, box_dim))
        assert t.dim() == 2 and t.size(-1) == box_dim, t.size()
        if t.shape[-1] == 6:
            t = torch.cat((t, t.new_zeros(t.shape[0], 1)), dim=-1)
            self.box_dim, self.with_yaw = box_dim + 1, False
        else:
            self.box_dim, self.with_yaw = box_dim, with_yaw
        self.tensor = t.clone()
        if tuple(origin) != (0.5, 0.5, 0):
            self.tensor[:, :3] += self.tensor[:, 3:6] * (self.tensor.new_tensor((0.5, 0.5, 0)) - self.tensor.new_tensor(origin))

    @property
    def device(self):
        return self.tensor.device

    @property
    def volume(self):
        return self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5]

    @property
    def gravity_center(self):
        c = self.tensor[:, :3].clone()
        c[:, 2] = self.tensor[:, 2] + self.tensor[:, 5] * 0.5
        return c

    def to(self, device):
        out = object.__new__(DepthInstance3DBoxes)
        out.tensor, out.box_dim, out.with_yaw = self.tensor.to(device), self.box_dim, self.with_yaw
        return out

    def __len__(self):
        return self.tensor.shape[0]


def bbox3d2result(bboxes, scores, labels):
    """mmdet3d/core/bbox/transforms.py:49-67: results live on the CPU.  From the GPU they travel as ONE device-to-host copy (three
    separate copies are three stream synchronisations at the end of every scene); class ids are exact in fp32."""
    t = bboxes.tensor
    if t.is_cuda and scores.is_cuda and labels.is_cuda and t.dtype == torch.float32 and scores.dtype == torch.float32 and t.dim() == 2:
        k, c = t.shape
        packed = torch.cat([t, scores.reshape(k, 1), labels.reshape(k, 1).to(torch.float32)], dim=1).cpu()
        out = object.__new__(type(bboxes))
        out.tensor, out.box_dim, out.with_yaw = packed[:, :c].contiguous(), bboxes.box_dim, bboxes.with_yaw
        return dict(boxes_3d=out, scores_3d=packed[:, c].contiguous(), labels_3d=packed[:, c + 1].to(torch.int64))
    return dict(boxes_3d=bboxes.to("cpu"), scores_3d=scores.cpu(), labels_3d=labels.cpu())
