#!/usr/bin/env python3
"""Golden vectors for the metric code from the reference's own eval_map_recall
(mmdet3d/core/evaluation/indoor_eval.py:55-200), imported from /root/reference with stand-ins for mmcv / terminaltables
and a 10-line axis-aligned box class in place of DepthInstance3DBoxes (whose ``overlaps`` needs a compiled CUDA op; for
yaw-free boxes it equals the axis-aligned IoU).  Run in the build container only."""
import importlib.util, os, sys, types
import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
for name in ("mmcv", "mmcv.utils", "terminaltables"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["mmcv.utils"].print_log = lambda *a, **k: None
sys.modules["terminaltables"].AsciiTable = object
spec = importlib.util.spec_from_file_location("ref_indoor_eval", "/root/reference/mmdet3d/core/evaluation/indoor_eval.py")
ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)


class Box:
    def __init__(self, t): self.tensor = torch.as_tensor(t, dtype=torch.float32).reshape(-1, 7)
    def __len__(self): return self.tensor.shape[0]
    def __getitem__(self, i): return Box(self.tensor[i])
    def new_box(self, t): return Box(t)
    @staticmethod
    def overlaps(a, b):
        def c(x):
            t = x.tensor
            return torch.stack((t[:, 0]-t[:, 3]/2, t[:, 1]-t[:, 4]/2, t[:, 2], t[:, 0]+t[:, 3]/2, t[:, 1]+t[:, 4]/2, t[:, 2]+t[:, 5]), 1)
        ca, cb = c(a), c(b)
        ext = (torch.min(ca[:, None, 3:], cb[None, :, 3:]) - torch.max(ca[:, None, :3], cb[None, :, :3])).clamp(min=0)
        inter = ext.prod(-1)
        va, vb = (ca[:, 3:]-ca[:, :3]).prod(-1), (cb[:, 3:]-cb[:, :3]).prod(-1)
        return (inter / torch.clamp(va[:, None] + vb[None, :] - inter, min=1e-8)).numpy()


rng = np.random.RandomState(0)
n_scenes, n_cls = 6, 4
gt_boxes, gt_cls, dt_boxes, dt_scores, dt_labels = [], [], [], [], []
pred, gt = {}, {}
for s in range(n_scenes):
    k = rng.randint(0, 6)
    ctr = rng.rand(k, 3) * [6, 6, 2]; size = 0.4 + rng.rand(k, 3)
    g = np.concatenate([ctr, size, np.zeros((k, 1))], 1).astype(np.float32); g[:, 2] -= g[:, 5] / 2
    cls = rng.randint(0, n_cls, k)
    m = rng.randint(3, 12)
    # detections: jittered copies of GT + random false positives
    src = rng.randint(0, max(k, 1), m)
    d = (g[src] if k else np.zeros((m, 7), np.float32)).copy()
    d[:, :6] += rng.randn(m, 6).astype(np.float32) * 0.12
    fp = rng.rand(m) < 0.3
    d[fp, :3] = (rng.rand(int(fp.sum()), 3) * [6, 6, 2]).astype(np.float32)
    d[:, 3:6] = np.abs(d[:, 3:6]) + 0.05
    dl = np.where(fp | (k == 0), rng.randint(0, n_cls, m), cls[src] if k else 0)
    ds = rng.rand(m).astype(np.float32)
    gt_boxes.append(g); gt_cls.append(cls); dt_boxes.append(d.astype(np.float32)); dt_scores.append(ds); dt_labels.append(dl)
    for i in range(m):
        lb = int(dl[i])
        pred.setdefault(lb, {}).setdefault(s, []).append((Box(d[i]), float(ds[i])))
        gt.setdefault(lb, {}).setdefault(s, [])
    for i in range(k):
        gt.setdefault(int(cls[i]), {}).setdefault(s, []).append(Box(g[i]))
for c in gt:  # the reference passes an (empty) box structure for scenes without GT of that class
    for s in gt[c]:
        if len(gt[c][s]) == 0:
            gt[c][s] = Box(np.zeros((0, 7), np.float32))
rec, prec, ap = ref.eval_map_recall(pred, gt, [0.25, 0.5])
out = dict(n_scenes=np.array(n_scenes))
for s in range(n_scenes):
    out[f"gt_boxes_{s}"], out[f"gt_cls_{s}"] = gt_boxes[s], gt_cls[s]
    out[f"dt_boxes_{s}"], out[f"dt_scores_{s}"], out[f"dt_labels_{s}"] = dt_boxes[s], dt_scores[s], dt_labels[s]
for t, thr in enumerate((25, 50)):
    for c in ap[t]:
        out[f"ap_{thr}_{c}"] = np.asarray(ap[t][c], dtype=np.float32).reshape(-1)[:1]
        out[f"rec_{thr}_{c}"] = np.asarray(rec[t][c], dtype=np.float64).reshape(-1)[-1:]
out["ap11"] = ref.average_precision(np.array([[0.25, 0.5, 0.75], [0.25, 0.5, 0.75]]), np.array([[1., 1., 1.], [1., 1., 1.]]), "11points")
out["ap_area"] = ref.average_precision(np.array([0.1, 0.1, 0.4, 0.7, 0.7, 1.0]), np.array([1.0, 0.5, 0.66, 0.75, 0.6, 0.5]))
np.savez_compressed(os.path.join(OUT, "indoor_eval.npz"), **out)
print("indoor_eval.npz", {k: v for k, v in out.items() if k.startswith("ap_")})
