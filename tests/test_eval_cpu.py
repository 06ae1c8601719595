"""Metric code: nerfdet_amd.eval against golden values produced by the reference's own
eval_map_recall / average_precision (tests/golden/make_golden_eval.py) and the reference's known-answer AP test
(tests/test_indoor_eval.py:183-187)."""
import numpy as np
import torch

from conftest import load_golden


def test_average_precision_reference_known_answers():
    from nerfdet_amd.eval import average_precision
    g = load_golden("indoor_eval")
    ap = average_precision(np.array([[0.25, 0.5, 0.75], [0.25, 0.5, 0.75]]), np.array([[1., 1., 1.], [1., 1., 1.]]), "11points")
    assert abs(ap[0] - 0.06611571) < 0.001                       # the reference's own assertion
    assert np.allclose(ap, g["ap11"].numpy())
    ap = average_precision(np.array([0.1, 0.1, 0.4, 0.7, 0.7, 1.0]), np.array([1.0, 0.5, 0.66, 0.75, 0.6, 0.5]))
    assert np.allclose(ap, g["ap_area"].numpy())


def test_indoor_eval_matches_reference_matching_and_ap():
    from nerfdet_amd.eval import indoor_eval
    g = load_golden("indoor_eval")
    n = int(g["n_scenes"])
    gt_annos, dt_annos = [], []
    for s in range(n):
        gb = g[f"gt_boxes_{s}"].clone()
        if len(gb):
            gb[:, 2] += gb[:, 5] / 2  # annotation files store gravity centres (indoor_eval.py:258-262)
        gt_annos.append(dict(gt_num=len(gb), gt_boxes_upright_depth=gb[:, :6].numpy(), **{"class": g[f"gt_cls_{s}"].numpy()}))
        dt_annos.append(dict(boxes_3d=g[f"dt_boxes_{s}"], scores_3d=g[f"dt_scores_{s}"], labels_3d=g[f"dt_labels_{s}"]))
    cats = {i: f"c{i}" for i in range(4)}
    res = indoor_eval(gt_annos, dt_annos, (0.25, 0.5), cats)
    for thr, tag in ((0.25, 25), (0.5, 50)):
        aps = []
        for c in range(4):
            assert abs(res[f"c{c}_AP_{thr:.2f}"] - float(g[f"ap_{tag}_{c}"][0])) < 1e-6, (thr, c)
            assert abs(res[f"c{c}_rec_{thr:.2f}"] - float(g[f"rec_{tag}_{c}"][0])) < 1e-9, (thr, c)
            aps.append(float(g[f"ap_{tag}_{c}"][0]))
        assert abs(res[f"mAP_{thr:.2f}"] - np.mean(aps)) < 1e-6
    assert res["mAP_0.25"] > res["mAP_0.50"] > 0


def test_indoor_eval_accepts_detector_output_structure():
    from nerfdet_amd.boxes import DepthInstance3DBoxes
    from nerfdet_amd.eval import indoor_eval
    gt = np.array([[1.0, 1.0, 0.5, 1.0, 1.0, 1.0], [3.0, 3.0, 0.5, 1.0, 2.0, 1.0]], dtype=np.float32)
    det = DepthInstance3DBoxes(torch.tensor(gt), box_dim=6, with_yaw=False, origin=(0.5, 0.5, 0.5))  # perfect detections
    res = indoor_eval([dict(gt_num=2, gt_boxes_upright_depth=gt, **{"class": np.array([0, 1])})],
                      [dict(boxes_3d=det, scores_3d=torch.tensor([0.9, 0.8]), labels_3d=torch.tensor([0, 1]))], (0.25, 0.5),
                      {0: "a", 1: "b"})
    assert res["mAP_0.25"] == 1.0 and res["mAP_0.50"] == 1.0 and res["mAR_0.50"] == 1.0
