"""A15 host mirror: ``aligned_3d_nms(boxes, scores, classes, thresh)`` with the reference's signature
(mmdet3d/core/post_processing/box3d_nms.py:91-138), executed by the HIP kernels in csrc/nms_kernels.hip."""
from __future__ import annotations

from ctypes import c_void_p

import torch

from . import _lib
from ._lib import check


def aligned_3d_nms(boxes: torch.Tensor, scores: torch.Tensor, classes: torch.Tensor, thresh: float) -> torch.Tensor:
    """boxes (n,6) x1y1z1x2y2z2, scores (n), classes (n) -> picked indices (k) int64, highest score first."""
    if not boxes.is_cuda:
        raise RuntimeError("nerfdet_amd.nms.aligned_3d_nms: tensors must live on the GPU (no CPU fallback)")
    n = boxes.shape[0]
    assert boxes.dim() == 2 and boxes.shape[1] == 6 and scores.shape[0] == n and classes.shape[0] == n
    dev = boxes.device
    lib = _lib.load()
    b = boxes.to(torch.float32).contiguous()
    s = scores.to(torch.float32).contiguous()
    c = classes.to(torch.int64).contiguous()
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    n_keep = torch.empty((1,), dtype=torch.int64, device=dev)
    ws = torch.empty((max(int(lib.ndet_nms_workspace_bytes(n)), 8),), dtype=torch.uint8, device=dev)
    st = c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    check(lib.ndet_aligned_3d_nms(c_void_p(b.data_ptr()), c_void_p(s.data_ptr()), c_void_p(c.data_ptr()), n, float(thresh),
                                  c_void_p(keep.data_ptr()), c_void_p(n_keep.data_ptr()), c_void_p(ws.data_ptr()), st),
          "aligned_3d_nms")
    return keep[: int(n_keep.item())]  # the one host sync of post-processing (the reference syncs once per pick)
