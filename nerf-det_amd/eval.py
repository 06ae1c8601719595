"""mAP@0.25 / mAP@0.5 for indoor detection results -- restatement of mmdet3d/core/evaluation/indoor_eval.py:7-310 for
the axis-aligned depth boxes NeRF-Det produces.  VOC-style: per class, detections of all scenes
sorted by confidence, greedy matching to the not-yet-matched ground truth with the highest IoU, AP = area under the
monotone precision envelope.  For yaw-free boxes the reference's rotated-IoU ``overlaps`` equals the axis-aligned 3D IoU
used here.  Host-side metric code (numpy): it is not on the device hot path."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch


def average_precision(recalls: np.ndarray, precisions: np.ndarray, mode: str = "area") -> np.ndarray:
    """indoor_eval.py:7-52.  'area': sum of recall steps times the precision envelope; '11points' keeps the
    reference's in-loop ``ap /= 11`` (so it is not the textbook 11-point AP; reference test pins 0.0661)."""
    if recalls.ndim == 1:
        recalls, precisions = recalls[np.newaxis, :], precisions[np.newaxis, :]
    assert recalls.shape == precisions.shape and recalls.ndim == 2
    n = recalls.shape[0]
    ap = np.zeros(n, dtype=np.float32)
    if mode == "area":
        mrec = np.hstack((np.zeros((n, 1), recalls.dtype), recalls, np.ones((n, 1), recalls.dtype)))
        mpre = np.hstack((np.zeros((n, 1), recalls.dtype), precisions, np.zeros((n, 1), recalls.dtype)))
        mpre = np.maximum.accumulate(mpre[:, ::-1], axis=1)[:, ::-1]
        for i in range(n):
            step = np.where(mrec[i, 1:] != mrec[i, :-1])[0]
            ap[i] = np.sum((mrec[i, step + 1] - mrec[i, step]) * mpre[i, step + 1])
    elif mode == "11points":
        for i in range(n):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                sel = precisions[i, recalls[i, :] >= thr]
                ap[i] += sel.max() if sel.size > 0 else 0
            ap /= 11
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap


def _corners(boxes: torch.Tensor) -> torch.Tensor:
    """(n,7) depth boxes (x, y, z_bottom, dx, dy, dz, yaw=0) -> (n,6) min/max corners."""
    lo = torch.stack((boxes[:, 0] - boxes[:, 3] / 2, boxes[:, 1] - boxes[:, 4] / 2, boxes[:, 2]), 1)
    hi = torch.stack((boxes[:, 0] + boxes[:, 3] / 2, boxes[:, 1] + boxes[:, 4] / 2, boxes[:, 2] + boxes[:, 5]), 1)
    return torch.cat((lo, hi), 1)


def box_overlaps(a: torch.Tensor, b: torch.Tensor) -> np.ndarray:
    """pairwise 3D IoU of yaw-free depth boxes, (n,7) x (m,7) -> (n,m)."""
    ca, cb = _corners(a.float()), _corners(b.float())
    ext = (torch.min(ca[:, None, 3:], cb[None, :, 3:]) - torch.max(ca[:, None, :3], cb[None, :, :3])).clamp(min=0)
    inter = ext.prod(-1)
    va, vb = (ca[:, 3:] - ca[:, :3]).prod(-1), (cb[:, 3:] - cb[:, :3]).prod(-1)
    return (inter / torch.clamp(va[:, None] + vb[None, :] - inter, min=1e-8)).numpy()


def eval_det_cls(pred: Dict[int, list], gt: Dict[int, torch.Tensor], iou_thr: Sequence[float], overlaps=box_overlaps):
    """One class.  ``pred[scene] = [(box (7,), score), ...]``, ``gt[scene] = (k,7)`` tensor.  Returns a list of
    (recall, precision, ap) per threshold.  indoor_eval.py:55-162."""
    used = {s: [np.zeros(len(g), dtype=bool) for _ in iou_thr] for s, g in gt.items()}
    npos = sum(len(g) for g in gt.values())
    scene_of, conf, iou_rows = [], [], []
    for s, dets in pred.items():
        if len(dets) == 0:
            continue
        boxes = torch.stack([torch.as_tensor(b, dtype=torch.float32) for b, _ in dets])
        g = gt.get(s, torch.zeros((0, 7)))
        ious = overlaps(boxes, g) if len(g) > 0 else np.zeros((len(dets), 1))
        for i, (_, score) in enumerate(dets):
            scene_of.append(s)
            conf.append(score)
            iou_rows.append(ious[i])
    order = np.argsort(-np.array(conf))
    nd = len(order)
    tp = [np.zeros(nd) for _ in iou_thr]
    fp = [np.zeros(nd) for _ in iou_thr]
    for d, k in enumerate(order):
        s, row = scene_of[k], iou_rows[k]
        n_gt = len(gt.get(s, ()))
        jmax, best = -1, -np.inf
        for j in range(n_gt):  # first maximum wins, as the reference's strict '>'
            if row[j] > best:
                best, jmax = row[j], j
        for t, thr in enumerate(iou_thr):
            if best > thr and not used[s][t][jmax]:
                tp[t][d] = 1.0
                used[s][t][jmax] = True
            else:
                fp[t][d] = 1.0
    out = []
    for t in range(len(iou_thr)):
        ctp, cfp = np.cumsum(tp[t]), np.cumsum(fp[t])
        recall = ctp / float(npos)
        precision = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        out.append((recall, precision, average_precision(recall, precision)))
    return out


def indoor_eval(gt_annos: List[dict], dt_annos: List[dict], metric: Sequence[float], label2cat: Dict[int, str]) -> Dict[str, float]:
    """``gt_annos[i] = dict(gt_num, gt_boxes_upright_depth (k,6) centre+size, class (k,))``;
    ``dt_annos[i] = dict(boxes_3d (DepthInstance3DBoxes or (n,7) tensor), scores_3d, labels_3d)`` (the detector's output).
    Returns ``{<cat>_AP_<thr>, mAP_<thr>, <cat>_rec_<thr>, mAR_<thr>}`` like indoor_eval.py:203-310."""
    assert len(gt_annos) == len(dt_annos)
    pred: Dict[int, Dict[int, list]] = {}
    gt: Dict[int, Dict[int, list]] = {}
    for s, det in enumerate(dt_annos):
        boxes = det["boxes_3d"].tensor if hasattr(det["boxes_3d"], "tensor") else det["boxes_3d"]
        for b, sc, lb in zip(boxes.cpu(), det["scores_3d"].cpu().numpy(), det["labels_3d"].cpu().numpy()):
            lb = int(lb)
            pred.setdefault(lb, {}).setdefault(s, []).append((b, float(sc)))
            gt.setdefault(lb, {}).setdefault(s, [])
        ann = gt_annos[s]
        if ann["gt_num"] != 0:
            g = torch.as_tensor(np.asarray(ann["gt_boxes_upright_depth"]), dtype=torch.float32)
            g7 = torch.cat((g[:, :3], g[:, 3:6], g.new_zeros(len(g), 1)), 1)
            g7[:, 2] -= g7[:, 5] * 0.5  # origin (0.5,0.5,0.5) -> bottom centre (base_box3d.py:61-64)
            for b, lb in zip(g7, np.asarray(ann["class"])):
                gt.setdefault(int(lb), {}).setdefault(s, []).append(b)
    gt_t = {c: {s: (torch.stack(v) if len(v) else torch.zeros((0, 7))) for s, v in d.items()} for c, d in gt.items()}
    res = {c: eval_det_cls(pred[c], gt_t[c], metric) for c in gt_t if c in pred}
    out: Dict[str, float] = {}
    for t, thr in enumerate(metric):
        aps, recs = [], []
        for c in gt_t:
            if c in res:
                rec, _, ap = res[c][t]
                ap_v, rec_v = float(ap[0]), float(rec[-1]) if len(rec) else 0.0
            else:
                ap_v, rec_v = 0.0, 0.0
            out[f"{label2cat[c]}_AP_{thr:.2f}"] = ap_v
            out[f"{label2cat[c]}_rec_{thr:.2f}"] = rec_v
            aps.append(ap_v)
            recs.append(rec_v)
        out[f"mAP_{thr:.2f}"] = float(np.mean(aps))
        out[f"mAR_{thr:.2f}"] = float(np.mean(recs))
    return out
