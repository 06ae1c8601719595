"""A15 host mirror: ``aligned_3d_nms(boxes, scores, classes, thresh)`` with the reference's signature
(mmdet3d/core/post_processing/box3d_nms.py:91-138), executed by the HIP kernels in csrc/nms_kernels.hip."""
from __future__ import annotations

from ctypes import c_void_p

import torch

from . import _lib
from ._lib import check


NMS_MAX = 4096   # candidates one launch of the kernels takes (one 64-bit word of the removed-set per lane, csrc/nms_kernels.hip)


def _nms_in_windows(boxes, scores, classes, thresh):
    """More candidates than one launch takes (the reference has no cap: ``nms_pre <= 0`` hands it every voxel): greedy NMS is
    sequential in score order, so it can be run window by window -- the boxes kept so far (all of higher score, mutually
    non-suppressing, hence kept again) go back in together with the next-best candidates that fit."""
    n = boxes.shape[0]
    # descending by (score, index), the order of k_nms_sort: stable sort of the index-reversed array
    rev = torch.arange(n - 1, -1, -1, device=boxes.device)
    order = rev[torch.sort(scores.float()[rev], descending=True, stable=True)[1]]
    kept = order[:0]
    pos = 0
    while pos < n:
        room = NMS_MAX - kept.numel()
        if room <= 0:
            raise ValueError(f"aligned_3d_nms: more than {NMS_MAX} boxes survive; raise score_thr or lower nms_pre")
        cand = torch.cat([kept, order[pos:pos + room]])
        pos += room
        kept = cand[aligned_3d_nms(boxes[cand], scores[cand], classes[cand], thresh)]
    return kept


def aligned_3d_nms(boxes: torch.Tensor, scores: torch.Tensor, classes: torch.Tensor, thresh: float) -> torch.Tensor:
    """boxes (n,6) x1y1z1x2y2z2, scores (n), classes (n) -> picked indices (k) int64, highest score first."""
    if not boxes.is_cuda:
        raise RuntimeError("nerfdet_amd.nms.aligned_3d_nms: tensors must live on the GPU (no CPU fallback)")
    n = boxes.shape[0]
    assert boxes.dim() == 2 and boxes.shape[1] == 6 and scores.shape[0] == n and classes.shape[0] == n
    if n > NMS_MAX:
        return _nms_in_windows(boxes, scores, classes, thresh)
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
