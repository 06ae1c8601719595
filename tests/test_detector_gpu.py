"""GPU parity for A13-A15 and the assembled detector: neck/head vs golden, NMS vs the reference's known answer,
``nerfdet.forward_test`` vs the oracle pipeline on identical weights (identical box indices)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub_state
from oracle import nerfdet_oracle as O

pytestmark = pytest.mark.gpu


def test_nms_reference_known_answer(device):
    """tests/test_nms.py:5-58 of the reference, through the HIP kernels."""
    from nerfdet_amd.nms import aligned_3d_nms
    from test_oracle_golden import test_nms_reference_known_answer as cpu_case  # noqa: F401  (same vectors)
    import inspect
    src = inspect.getsource(cpu_case)
    ns = {"torch": torch}
    body = src.split('"""', 2)[2].split("assert torch.equal")[0]
    exec("\n".join(l[4:] for l in body.splitlines()), ns)
    pick = aligned_3d_nms(ns["boxes"].to(device), ns["scores"].to(device), ns["cls"].to(device), 0.25)
    assert pick.dtype == torch.int64 and torch.equal(pick.cpu(), ns["expected"])


def test_nms_random_golden_and_edges(device):
    from nerfdet_amd.nms import aligned_3d_nms
    g = load_golden("nms_random")
    b, s, c = g["boxes"].to(device), g["scores"].to(device), g["classes"].to(device)
    for thr in (0.25, 0.5):
        assert torch.equal(aligned_3d_nms(b, s, c, thr).cpu(), g[f"pick_{int(thr * 100)}"])
    # zero-volume boxes: 0/0 = NaN IoU suppresses (box3d_nms.py:131-135)
    pick = aligned_3d_nms(g["deg_boxes"].to(device), s[:40], torch.zeros(40, dtype=torch.long, device=device), 0.25)
    assert torch.equal(pick.cpu(), g["deg_pick"])
    # empty and single inputs
    e = aligned_3d_nms(torch.zeros(0, 6, device=device), torch.zeros(0, device=device), torch.zeros(0, dtype=torch.long, device=device), 0.25)
    assert e.numel() == 0 and e.dtype == torch.int64
    one = aligned_3d_nms(b[:1], s[:1], c[:1], 0.25)
    assert one.tolist() == [0]
    # maximum size the head can produce: 3 levels x nms_pre=1000
    gen = torch.Generator().manual_seed(5)
    n = 3000
    ctr = torch.rand(n, 3, generator=gen) * torch.tensor([6.4, 6.4, 3.2])
    size = 0.2 + torch.rand(n, 3, generator=gen)
    bb = torch.cat([ctr - size / 2, ctr + size / 2], 1)
    ss = torch.rand(n, generator=gen)
    cc = torch.randint(0, 18, (n,), generator=gen)
    ref = O.aligned_3d_nms(bb, ss, cc, 0.25)
    got = aligned_3d_nms(bb.to(device), ss.to(device), cc.to(device), 0.25)
    assert torch.equal(got.cpu(), ref) and 500 < len(ref) < 3000
    # idempotence: NMS of the survivors keeps all of them, in order
    again = aligned_3d_nms(bb[ref].to(device), ss[ref].to(device), cc[ref].to(device), 0.25)
    assert torch.equal(again.cpu(), torch.arange(len(ref)))
    # more candidates than one launch takes (nms_pre <= 0 hands the NMS every voxel): windowed, same picks as the sequential loop
    g2 = torch.Generator().manual_seed(7)
    n = 9000
    ctr = torch.rand(n, 3, generator=g2) * torch.tensor([6.4, 6.4, 3.2])
    size = 0.6 + torch.rand(n, 3, generator=g2)
    bb = torch.cat([ctr - size / 2, ctr + size / 2], 1)
    ss = torch.rand(n, generator=g2)
    cc = torch.randint(0, 3, (n,), generator=g2)
    ref = O.aligned_3d_nms(bb, ss, cc, 0.25)
    got = aligned_3d_nms(bb.to(device), ss.to(device), cc.to(device), 0.25)
    assert torch.equal(got.cpu(), ref) and 1000 < len(ref) < 4096
    with pytest.raises(ValueError):      # more survivors than one launch can carry: refused loudly, never truncated
        aligned_3d_nms(torch.cat([ctr - 0.01, ctr + 0.01], 1).to(device), ss.to(device), cc.to(device), 0.25)


def test_neck_and_head_match_reference_golden(device):
    from nerfdet_amd.boxes import DepthInstance3DBoxes
    from nerfdet_amd.config import ConfigDict
    from nerfdet_amd.head import ScanNetImVoxelHeadV2
    from nerfdet_amd.neck3d import FastIndoorImVoxelNeck
    g = load_golden("head_small_s0")
    neck = FastIndoorImVoxelNeck(8, [1, 1, 1], 8)
    neck.load_state_dict(sub_state(g, "neck_3d."))
    neck.to(device).eval()
    head = ScanNetImVoxelHeadV2(n_classes=18, n_channels=8, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18,
                                test_cfg=ConfigDict(nms_pre=120, iou_thr=0.25, score_thr=0.01))
    head.load_state_dict(sub_state(g, "bbox_head."))
    head.voxel_size = tuple(g["voxel_size"].tolist())
    head.to(device).eval()
    with torch.no_grad():
        outs = neck(g["x"].to(device))
        for i in range(3):
            torch.testing.assert_close(outs[i].cpu(), g[f"neck_eval_{i}"], rtol=1e-4, atol=1e-5)
        # feed the golden neck outputs so the comparison below is on identical head inputs
        ctr, reg, cls = head([g[f"neck_eval_{i}"].to(device) for i in range(3)])
        for i in range(3):
            torch.testing.assert_close(ctr[i].cpu(), g[f"ctr_{i}"], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(reg[i].cpu(), g[f"reg_{i}"], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(cls[i].cpu(), g[f"cls_{i}"], rtol=1e-4, atol=1e-5)
        meta = dict(lidar2img=dict(origin=g["origin"].numpy()), box_type_3d=DepthInstance3DBoxes)
        (boxes, scores, labels), = head.get_bboxes([g[f"ctr_{i}"].to(device) for i in range(3)],
                                                   [g[f"reg_{i}"].to(device) for i in range(3)],
                                                   [g[f"cls_{i}"].to(device) for i in range(3)], g["valid"].to(device), [meta])
    assert torch.equal(labels.cpu(), g["det_labels"])            # identical box indices -> identical labels, order
    torch.testing.assert_close(scores.cpu(), g["det_scores"], rtol=1e-5, atol=1e-6)
    # the golden wrapper kept raw (centre,size) rows; ours re-bases z to the box bottom like DepthInstance3DBoxes
    raw = boxes.tensor[:, :6].clone()
    raw[:, 2] += raw[:, 5] * 0.5
    torch.testing.assert_close(raw.cpu(), g["det_boxes"], rtol=1e-5, atol=1e-5)


def test_detector_forward_test_vs_oracle_pipeline(device):
    """Whole ``nerfdet.forward_test`` (ResNet+FPN, HIP hot path, 3D neck, head, HIP NMS) against the oracle fed with
    the same weights and the same FPN features: identical detections."""
    from nerfdet_amd.config import _wrap
    from nerfdet_amd.presets import nerfdet_cfg
    from nerfdet_amd.registry import build_detector
    torch.manual_seed(0)
    cfg = _wrap(nerfdet_cfg(50, n_voxels=(16, 16, 8), voxel_size=(0.4, 0.4, 0.4)))
    cfg["test_cfg"]["nms_pre"] = 200
    det = build_detector(cfg["model"], train_cfg=cfg["train_cfg"], test_cfg=cfg["test_cfg"])
    with torch.no_grad():
        # a random-init ResNet has no calibrated BN statistics: rescale so features are O(1), make the density
        # branch and the head fire (default inits give density 0 and scores ~0.005 < score_thr everywhere)
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(2.0)
        det.bbox_head.cls_conv.weight.normal_(0, 0.3)
        det.bbox_head.cls_conv.bias.fill_(-2.0)
        det.bbox_head.centerness_conv.weight.normal_(0, 0.1)
        det.bbox_head.reg_conv.weight.normal_(0, 0.05)
        det.mapping[0].bias.normal_(0, 0.3)
    det.eval()
    n_v, hw = 6, (64, 96)
    meta = O.ring_scene_meta(n_v, hw)
    img = torch.randn(1, n_v, 3, *hw)
    denorm = torch.rand(1, n_v, 3, *hw)
    rays = dict(lightpos=torch.zeros(1, 1, 4, 3), raydirs=torch.ones(1, 1, 4, 3), gt_images=torch.zeros(1, 1, 4, 3),
                gt_depths=[], nerf_sizes=[torch.tensor([[2, 2, 3]])])
    # oracle side, CPU
    with torch.no_grad():
        feats = det.neck(det.backbone(img[0]))[0]
        ov = O.extract_volume(feats, denorm[0], meta, (16, 16, 8), (0.4, 0.4, 0.4), det.mapping[0].weight, det.mapping[0].bias,
                              det.nerf_mlp.state_dict())
        n3 = O.neck3d_forward({k: v for k, v in det.neck_3d.state_dict().items()}, ov["volume"].unsqueeze(0))
        ctr, reg, cls = O.head_forward(det.bbox_head.state_dict(), n3)
        ref = O.head_get_bboxes(ctr, reg, cls, ov["valid"].unsqueeze(0).float(), meta["lidar2img"]["origin"], (0.4, 0.4, 0.4), 200, 0.01, 0.25)
    det.to(device)
    with torch.no_grad():
        res = det(img.to(device), [dict(meta)], return_loss=False, denorm_images=denorm.to(device),
                  **{k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in rays.items()})
    assert isinstance(res, list) and set(res[0]) == {"boxes_3d", "scores_3d", "labels_3d"}
    assert len(ref["labels"]) > 50 and ref["labels"].unique().numel() > 5, "test must exercise NMS"
    assert ref["cand_scores"].unique().numel() == len(ref["cand_scores"]), "score ties make the pick order undefined"
    assert torch.equal(res[0]["labels_3d"], ref["labels"])
    torch.testing.assert_close(res[0]["scores_3d"], ref["scores"], rtol=1e-3, atol=1e-5)
    got = res[0]["boxes_3d"].tensor[:, :6].clone()
    got[:, 2] += got[:, 5] * 0.5
    torch.testing.assert_close(got, ref["boxes"], rtol=1e-3, atol=1e-4)


def test_graphed_forward_test_equals_eager(device):
    """hipGraph replay of the static part of forward_test gives the eager result, also for a second scene with
    different cameras and images fed through the same captured graph."""
    from nerfdet_amd.config import _wrap
    from nerfdet_amd.graphed import GraphedForwardTest
    from nerfdet_amd.presets import nerfdet_cfg
    from nerfdet_amd.registry import build_detector
    torch.manual_seed(0)
    cfg = _wrap(nerfdet_cfg(50, n_voxels=(16, 16, 8), voxel_size=(0.4, 0.4, 0.4)))
    cfg["test_cfg"]["nms_pre"] = 200
    det = build_detector(cfg["model"], train_cfg=cfg["train_cfg"], test_cfg=cfg["test_cfg"])
    with torch.no_grad():
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(2.0)
        det.bbox_head.cls_conv.weight.normal_(0, 0.3)
        det.bbox_head.cls_conv.bias.fill_(-2.0)
        det.bbox_head.centerness_conv.weight.normal_(0, 0.1)
        det.bbox_head.reg_conv.weight.normal_(0, 0.05)
    det.to(device).eval()
    graphed = GraphedForwardTest(det)
    n_v, hw = 6, (64, 96)
    rays = dict(lightpos=torch.zeros(1, 1, 4, 3, device=device), raydirs=torch.ones(1, 1, 4, 3, device=device),
                gt_images=torch.zeros(1, 1, 4, 3, device=device), gt_depths=[], nerf_sizes=[torch.tensor([[2, 2, 3]])])
    for seed, radius in ((0, 2.5), (1, 2.2), (2, 2.8)):
        g = torch.Generator().manual_seed(seed)
        meta = O.ring_scene_meta(n_v, hw, radius=radius)
        img = torch.randn(1, n_v, 3, *hw, generator=g).to(device)
        dn = torch.rand(1, n_v, 3, *hw, generator=g).to(device)
        with torch.no_grad():
            eager = det(img, [dict(meta)], return_loss=False, denorm_images=dn, **rays)
        got = graphed(img, [dict(meta)], return_loss=False, denorm_images=dn, **rays)
        assert len(eager[0]["scores_3d"]) > 5
        assert torch.equal(got[0]["labels_3d"], eager[0]["labels_3d"])
        assert torch.equal(got[0]["scores_3d"], eager[0]["scores_3d"])
        assert torch.equal(got[0]["boxes_3d"].tensor, eager[0]["boxes_3d"].tensor)


def test_level_valid_equals_trilinear_round_bool(device):
    """ndet_level_valid == F.interpolate(valid, size, 'trilinear').round().bool() for the FPN levels' down-scales."""
    from ctypes import c_void_p
    from nerfdet_amd import _lib
    torch.manual_seed(5)
    valid = ((torch.rand(1, 1, 24, 16, 8, device=device) < 0.3).float() * torch.randint(1, 4, (1, 1, 24, 16, 8), device=device).float())
    lib = _lib.load()
    st = c_void_p(torch.cuda.current_stream(device).cuda_stream)
    for f in (1, 2, 4):
        size = (24 // f, 16 // f, 8 // f)
        ref = torch.nn.functional.interpolate(valid, size=size, mode="trilinear").round().bool().reshape(-1)
        out = torch.empty((size[0] * size[1] * size[2],), dtype=torch.uint8, device=device)
        _lib.check(lib.ndet_level_valid(c_void_p(valid.data_ptr()), 24, 16, 8, f, c_void_p(out.data_ptr()), st), "level_valid")
        assert torch.equal(out.bool(), ref), f
        assert 0.02 < ref.float().mean() < 0.98
    with pytest.raises(ValueError):
        _lib.check(lib.ndet_level_valid(c_void_p(valid.data_ptr()), 24, 16, 8, 3, c_void_p(out.data_ptr()), st), "level_valid")


@pytest.mark.parametrize("nms_pre", [300, 5000])   # 300: a level has more survivors than nms_pre -> generic path; 5000: fused path
def test_fused_head_decode_equals_standard_get_bboxes(device, nms_pre):
    """simple_test_fused (one decode kernel per level) == forward + get_bboxes (the reference's op chain) on the GPU."""
    from nerfdet_amd.boxes import DepthInstance3DBoxes
    from nerfdet_amd.config import ConfigDict
    from nerfdet_amd.head import ScanNetImVoxelHeadV2
    torch.manual_seed(3)
    head = ScanNetImVoxelHeadV2(n_classes=18, n_channels=128, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18,
                                test_cfg=ConfigDict(nms_pre=nms_pre, iou_thr=0.25, score_thr=0.01))
    head.voxel_size = (0.16, 0.16, 0.2)
    with torch.no_grad():
        head.cls_conv.weight.normal_(0, 0.05); head.cls_conv.bias.normal_(-2, 0.3)
        head.reg_conv.weight.normal_(0, 0.02); head.centerness_conv.weight.normal_(0, 0.05)
        for i, sc in enumerate(head.scales):
            sc.scale.fill_(1.0 + 0.3 * i)
    head.to(device).eval()
    feats = [torch.randn(1, 128, 24 // 2 ** i, 24 // 2 ** i, 8 // 2 ** i, device=device) for i in range(3)]
    valid = ((torch.rand(1, 1, 24, 24, 8, device=device) < 0.7).float() * torch.randint(1, 6, (1, 1, 24, 24, 8), device=device).float())
    meta = dict(lidar2img=dict(origin=np.array([0.0, 0.0, 0.5], dtype=np.float32)), box_type_3d=DepthInstance3DBoxes)
    with torch.no_grad():
        (b0, s0, l0), = head.get_bboxes(*head(feats), valid, [meta])
        (b1, s1, l1), = head.simple_test_fused(feats, valid, [meta])
    assert len(s0) > 30
    assert torch.equal(l0.cpu(), l1.cpu())        # the fused tail hands its picks over on the host (one packed copy)
    torch.testing.assert_close(s1.cpu(), s0.cpu(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(b1.tensor.cpu(), b0.tensor.cpu(), rtol=1e-5, atol=1e-5)


def _small_detector(device, seed=0):
    from nerfdet_amd.config import _wrap
    from nerfdet_amd.presets import nerfdet_cfg
    from nerfdet_amd.registry import build_detector
    torch.manual_seed(seed)
    cfg = _wrap(nerfdet_cfg(50, n_voxels=(16, 16, 8), voxel_size=(0.4, 0.4, 0.4)))
    cfg["test_cfg"]["nms_pre"] = 200
    det = build_detector(cfg["model"], train_cfg=cfg["train_cfg"], test_cfg=cfg["test_cfg"])
    with torch.no_grad():
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(2.0)
        det.bbox_head.cls_conv.weight.normal_(0, 0.3)
        det.bbox_head.cls_conv.bias.fill_(-2.0)
        det.bbox_head.centerness_conv.weight.normal_(0, 0.1)
        det.bbox_head.reg_conv.weight.normal_(0, 0.05)
    return det.to(device).eval()


def _scene(device, seed, n_v=6, hw=(64, 96)):
    g = torch.Generator().manual_seed(seed)
    rays = dict(lightpos=torch.zeros(1, 1, 4, 3, device=device), raydirs=torch.ones(1, 1, 4, 3, device=device),
                gt_images=torch.zeros(1, 1, 4, 3, device=device), gt_depths=[], nerf_sizes=[torch.tensor([[2, 2, 3]])])
    return (torch.randn(1, n_v, 3, *hw, generator=g).to(device), torch.rand(1, n_v, 3, *hw, generator=g).to(device),
            O.ring_scene_meta(n_v, hw), rays)


def test_forward_test_agrees_between_the_two_convolution_arithmetics(device):
    """Exact fp32-MFMA kernels vs the 3-term bf16 split on the bf16 matrix cores: same detections, boxes to 1e-4."""
    from nerfdet_amd import conv3d
    det = _small_detector(device)
    img, dn, meta, rays = _scene(device, 4)
    out = {}
    for mode in ("f32", "bf16x3"):
        prev = conv3d.set_arithmetic(mode)
        try:
            with torch.no_grad():
                out[mode] = det(img, [dict(meta)], return_loss=False, denorm_images=dn, **rays)[0]
        finally:
            conv3d.set_arithmetic(prev)
    a, b = out["f32"], out["bf16x3"]
    assert len(a["scores_3d"]) > 5
    assert torch.equal(a["labels_3d"], b["labels_3d"])
    torch.testing.assert_close(a["scores_3d"], b["scores_3d"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(a["boxes_3d"].tensor, b["boxes_3d"].tensor, rtol=1e-4, atol=1e-4)


def test_forward_test_with_no_detection_and_with_unseen_scene(device):
    """Edge cases of the fused post-processing: nothing clears the score threshold; no voxel is seen by any camera."""
    det = _small_detector(device)
    img, dn, meta, rays = _scene(device, 5)
    det.bbox_head.test_cfg["score_thr"] = 0.9999
    with torch.no_grad():
        res = det(img, [dict(meta)], return_loss=False, denorm_images=dn, **rays)[0]
    assert len(res["scores_3d"]) == 0 and res["boxes_3d"].tensor.shape[0] == 0 and res["labels_3d"].dtype == torch.int64
    det.bbox_head.test_cfg["score_thr"] = 0.01
    far = O.ring_scene_meta(6, (64, 96))
    for e in far["lidar2img"]["extrinsic"]:
        e[2, 3] -= 1000.0          # every voxel ends up behind every camera
    with torch.no_grad():
        res = det(img, [dict(far)], return_loss=False, denorm_images=dn, **rays)[0]
    assert len(res["scores_3d"]) == 0   # the head masks voxels no view sees


def test_full_size_cfg2_voxel_features_agree_between_arithmetics(device):
    """BASELINE's full size (50 views 240x320, 40x40x16 voxels, ResNet-50): the voxel features the hot path produces from the
    bf16x3 backbone agree with those from the exact fp32-MFMA backbone within the north-star tolerance (1e-4, relative to the
    feature scale), the view counts exactly, and the detections are the same set."""
    import importlib.util
    import os
    from nerfdet_amd import conv3d
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import nerfdet_amd.volume as V
    w = bench.WORKLOADS["cfg2"]
    det = bench.build_model(w).to(device)
    batch = bench.to_device(bench.synth_batch(w, 0), device)
    out, res = {}, {}
    for mode in ("f32", "bf16x3"):
        prev = conv3d.set_arithmetic(mode)
        try:
            with torch.no_grad():
                x, b, stride = det.extract_2d(batch["img"])
                out[mode] = V.extract_volume(x, batch["denorm_images"][0], batch["img_metas"][0], det.n_voxels, det.voxel_size,
                                             det.mapping, det.nerf_mlp, stride=stride, channels_last_out=True)
                res[mode] = det(return_loss=False, **{k: (list(v) if isinstance(v, list) else v) for k, v in batch.items()})[0]
        finally:
            conv3d.set_arithmetic(prev)
    a, c = out["f32"], out["bf16x3"]
    assert torch.equal(a["valid"], c["valid"]) and float((a["valid"] > 0).float().mean()) > 0.2
    scale = max(1.0, float(a["volume"].abs().max()))
    assert float((a["volume"] - c["volume"]).abs().max()) <= 1e-4 * scale
    assert float((a["global_feat"] - c["global_feat"]).abs().max()) <= 1e-4 * max(1.0, float(a["global_feat"].abs().max()))
    assert len(res["f32"]["scores_3d"]) > 50
    assert torch.equal(res["f32"]["labels_3d"], res["bf16x3"]["labels_3d"])
    torch.testing.assert_close(res["f32"]["scores_3d"], res["bf16x3"]["scores_3d"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("nms_pre", [50, 400, 5000])
def test_device_side_topk_compaction_keeps_the_reference_set(device, nms_pre):
    """ndet_select_candidates_topk: per level, the candidates left by ``topk(nms_pre)`` followed by ``score > thr``
    (imvoxel_head_v2.py:272-276,533-536) -- as a set, since the reference orders them by score inside a level -- incl. exact ties
    at the cut (the first in voxel order are kept)."""
    import ctypes
    from ctypes import c_void_p
    from nerfdet_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(nms_pre)
    sizes = [6000, 1500, 377]
    thr = 0.05
    bests = [torch.rand(n, generator=g) for n in sizes]
    bests[0][100:140] = 0.625              # exact ties, inside the cut region for nms_pre = 400 / straddling it for others
    bests[1][:] = torch.rand(1500, generator=g) * 0.04   # a level with nothing above the threshold
    labels = [torch.randint(0, 18, (n,), generator=g) for n in sizes]
    boxes = [torch.rand(n, 6, generator=g) for n in sizes]
    db, dl, dx = [t.to(device) for t in bests], [t.to(device) for t in labels], [t.to(device) for t in boxes]
    tot = sum(sizes)
    o_b = torch.empty(tot, device=device)
    o_l = torch.empty(tot, dtype=torch.int64, device=device)
    o_x = torch.empty(tot, 6, device=device)
    counts = torch.empty(5, dtype=torch.int32, device=device)
    pa = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    _lib.check(lib.ndet_select_candidates_topk(3, pa(db), pa(dl), pa(dx), (ctypes.c_int * 3)(*sizes), thr, nms_pre, c_void_p(o_b.data_ptr()),
                                               c_void_p(o_l.data_ptr()), c_void_p(o_x.data_ptr()), c_void_p(counts.data_ptr()),
                                               c_void_p(torch.cuda.current_stream(device).cuda_stream)), "select_candidates_topk")
    cnt = counts.cpu().tolist()
    off = 0
    for l in range(3):
        keep = bests[l] > thr
        idx = keep.nonzero().flatten()
        if len(idx) > nms_pre:
            order = torch.sort(bests[l][idx], descending=True, stable=True)[1][:nms_pre]   # stable: ties by voxel order, as the kernel
            idx = idx[order].sort()[0]
        assert cnt[l] == len(idx), (l, cnt[l], len(idx))
        got = o_b[off:off + cnt[l]].cpu()
        assert torch.equal(got, bests[l][idx])                                         # same scores in voxel order
        assert torch.equal(o_l[off:off + cnt[l]].cpu(), labels[l][idx]) and torch.equal(o_x[off:off + cnt[l]].cpu(), boxes[l][idx])
        off += cnt[l]
    assert cnt[3] == off and cnt[4] == sum(int((b > thr).sum()) for b in bests)


def test_simple_test_with_evaluate_nerf_keeps_the_rendering_metrics(device):
    """nerfdet.py:338-343 with ``render_testing=True`` and ``evaluate_nerf=True``: every ray of the target views is rendered in the test
    pass, and (psnr, ssim, depth-error map) of save_rendered_img.py:38-78 are computed from that rendering -- here kept on the detector;
    the detections are those of the plain pass."""
    from nerfdet_amd import rays
    from nerfdet_amd.synth import batch_to, train_scene
    det = _small_detector(device)
    det.render_testing = True
    data = batch_to(train_scene(6, (64, 96), t_views=2, n_boxes=2, seed=4), device)
    rb = det._ray_batch(data)
    with torch.no_grad():
        res = det.simple_test(data["img"], data["img_metas"], ray_batch=rb, evaluate_nerf=True)
        psnr, ssim, err = det.render_metrics
        _, _, _, rgb_preds, _ = det.extract_feat(data["img"], data["img_metas"], "test", ray_batch=rb)
        want = rays.rendering_metrics(rgb_preds[0])
        det.render_testing = False
        plain = det.simple_test(data["img"], data["img_metas"], ray_batch=rb)
    rh, rw = 64 - 20, 96 - 20
    assert rgb_preds[0]["outputs_coarse"]["rgb"].shape == (2, rh, rw, 3) and err.shape == (rh, rw, 1)
    assert float(psnr) == float(want[0]) and float(ssim) == float(want[1]) and torch.equal(err, want[2])
    assert torch.isfinite(psnr) and 0.0 < float(ssim) < 1.0
    assert torch.equal(res[0]["labels_3d"], plain[0]["labels_3d"]) and torch.equal(res[0]["scores_3d"], plain[0]["scores_3d"])


def test_forward_test_async_two_scenes_in_flight_equals_sequential(device):
    """nerfdet.forward_test_async: two different scenes queued on two streams, collected one behind, repeatedly -- detections identical to
    the synchronous forward_test of each scene (same boxes, scores, labels), including a scene without detections."""
    det = _small_detector(device)
    scenes = [_scene(device, s) for s in (1, 2, 3)]
    with torch.no_grad():
        want = [det.forward_test(img, [meta], denorm_images=dn, **rays)[0] for img, dn, meta, rays in scenes]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        pend, got = [], []
        for i in range(9):
            img, dn, meta, rays = scenes[i % 3]
            with torch.cuda.stream(streams[i % 2]):
                pend.append(det.forward_test_async(img, [meta], denorm_images=dn, **rays))
            if len(pend) == 2:
                got.append(pend.pop(0)()[0])
        while pend:
            got.append(pend.pop(0)()[0])
    assert len(got) == 9
    for i, g in enumerate(got):
        w_ = want[i % 3]
        assert torch.equal(g["labels_3d"], w_["labels_3d"]) and torch.equal(g["scores_3d"], w_["scores_3d"])
        assert torch.equal(g["boxes_3d"].tensor, w_["boxes_3d"].tensor)


def test_async_tail_falls_back_like_the_synchronous_one_when_the_device_side_tail_overflows(device):
    """More than 1 024 picks (no suppression, score threshold 0): the device-side tail flags the overflow, and both forms -- forward_test
    and forward_test_async's finish() on a side stream -- redo the tail on the host-driven path with the same detections."""
    det = _small_detector(device)
    det.bbox_head.test_cfg["iou_thr"] = 0.999
    det.bbox_head.test_cfg["score_thr"] = 0.0
    det.bbox_head.test_cfg["nms_pre"] = 5000
    img, dn, meta, rays = _scene(device, 5)
    with torch.no_grad():
        want = det.forward_test(img, [meta], denorm_images=dn, **rays)[0]
        s = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            fin = det.forward_test_async(img, [meta], denorm_images=dn, **rays)
        got = fin()[0]
    assert len(want["scores_3d"]) > 1024
    assert torch.equal(got["labels_3d"], want["labels_3d"]) and torch.equal(got["scores_3d"], want["scores_3d"])
    assert torch.equal(got["boxes_3d"].tensor, want["boxes_3d"].tensor)


def test_async_handles_on_one_stream_keep_their_own_detections(device):
    """Three different scenes queued back to back on ONE stream, collected afterwards in reverse order: every handle owns its pinned
    landing buffer, so a later scene cannot overwrite an uncollected earlier one; a handle may be collected twice."""
    det = _small_detector(device)
    scenes = [_scene(device, s) for s in (1, 2, 3)]
    with torch.no_grad():
        want = [det.forward_test(img, [meta], denorm_images=dn, **rays)[0] for img, dn, meta, rays in scenes]
        assert not torch.equal(want[0]["scores_3d"], want[1]["scores_3d"]) and len(want[0]["scores_3d"]) > 5
        s = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            handles = [det.forward_test_async(img, [meta], denorm_images=dn, **rays) for img, dn, meta, rays in scenes]
        got = {i: handles[i]()[0] for i in (2, 0, 1)}
        again = handles[0]()[0]
    for i in range(3):
        assert torch.equal(got[i]["labels_3d"], want[i]["labels_3d"]) and torch.equal(got[i]["scores_3d"], want[i]["scores_3d"]), i
        assert torch.equal(got[i]["boxes_3d"].tensor, want[i]["boxes_3d"].tensor)
    assert torch.equal(again["scores_3d"], want[0]["scores_3d"])


def test_checkpoint_loaded_after_a_warm_forward_takes_effect_and_round_trips(device, tmp_path):
    """tools/test.py:113-117 builds the model, then loads the checkpoint; a server may do so after it has served scenes.  Every cached
    weight pack (conv3d.packed / bn_affine / packed_linear / the stem and bottleneck-chain packs / the head's fused pack) must notice
    the in-place overwrite: detections after ``load_checkpoint`` equal those of a detector BUILT with the new weights, bit for bit, and
    save -> load reproduces them."""
    from nerfdet_amd.checkpoint import load_checkpoint, save_checkpoint
    img, dn, meta, rays = _scene(device, 4)

    def run(d):
        with torch.no_grad():
            return d(img, [dict(meta)], return_loss=False, denorm_images=dn, **rays)[0]
    det_a, det_b = _small_detector(device, seed=0), _small_detector(device, seed=1)
    path_b = str(tmp_path / "b.pth")
    save_checkpoint(det_b, path_b, meta=dict(CLASSES=("x",)))
    res_b = run(det_b)
    res_a = run(det_a)                          # warm: every pack of det_a is cached now
    run(det_a)
    assert len(res_a["scores_3d"]) > 5 and len(res_b["scores_3d"]) > 5
    assert not (len(res_a["scores_3d"]) == len(res_b["scores_3d"]) and torch.equal(res_a["scores_3d"], res_b["scores_3d"]))
    ck = load_checkpoint(det_a, path_b, map_location="cpu", strict=True)
    assert ck["meta"]["CLASSES"] == ("x",)
    res_ab = run(det_a)
    for k in ("labels_3d", "scores_3d"):
        assert torch.equal(res_ab[k], res_b[k]), f"{k}: stale weight pack after load_checkpoint"
    assert torch.equal(res_ab["boxes_3d"].tensor, res_b["boxes_3d"].tensor)
    # the same through the hipGraph replay path's static copies and the two-in-flight path
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.no_grad(), torch.cuda.stream(s):
        fin = det_a.forward_test_async(img, [dict(meta)], denorm_images=dn, **rays)
    assert torch.equal(fin()[0]["scores_3d"], res_b["scores_3d"])
    # save -> load into a third detector reproduces the detections bit for bit
    path_a = str(tmp_path / "a.pth")
    save_checkpoint(det_a, path_a)
    det_c = _small_detector(device, seed=2)
    run(det_c)
    load_checkpoint(det_c, path_a, map_location="cpu", strict=True)
    res_c = run(det_c)
    assert torch.equal(res_c["scores_3d"], res_b["scores_3d"]) and torch.equal(res_c["boxes_3d"].tensor, res_b["boxes_3d"].tensor)


def test_all_levels_decoded_in_one_launch_equal_the_per_level_launches(device):
    """ndet_head_decode_levels against ndet_level_valid + ndet_head_decode level by level (imvoxel_head_v2.py:262-271,442-449,547-555): the same
    scores, labels and boxes bit for bit, on three levels of a 16 x 12 x 8 grid with a ragged validity volume."""
    import ctypes
    from ctypes import c_void_p
    import numpy as np
    from nerfdet_amd import _lib
    from nerfdet_amd._lib import check, float3
    lib = _lib.load()
    torch.manual_seed(3)
    X, Y, Z, ncls = 16, 12, 8, 18
    valid = (torch.rand(X, Y, Z, device=device) > 0.35).float().contiguous()
    st = c_void_p(torch.cuda.current_stream(device).cuda_stream)
    vs0, origin = np.float32([0.16, 0.16, 0.2]), np.float32([0.3, -0.2, 1.1])
    raws, scales, grids, facs = [], [], [], []
    for l in range(3):
        f = 2 ** l
        g = (X // f, Y // f, Z // f)
        raws.append(torch.randn(*g, 7 + ncls, device=device))
        scales.append(torch.tensor([0.7 + 0.2 * l], device=device))
        grids.append(g)
        facs.append(f)
    ref = []
    for l, (raw, sc, g, f) in enumerate(zip(raws, scales, grids, facs)):
        n = g[0] * g[1] * g[2]
        v = torch.empty(n, dtype=torch.uint8, device=device)
        check(lib.ndet_level_valid(c_void_p(valid.data_ptr()), X, Y, Z, f, c_void_p(v.data_ptr()), st), "level_valid")
        best, lab, box = torch.empty(n, device=device), torch.empty(n, dtype=torch.int64, device=device), torch.empty(n, 6, device=device)
        check(lib.ndet_head_decode(c_void_p(raw.data_ptr()), ncls, c_void_p(v.data_ptr()), c_void_p(sc.data_ptr()), *g, float3(np.float32(vs0 * f)), float3(origin),
                                   c_void_p(best.data_ptr()), c_void_p(lab.data_ptr()), c_void_p(box.data_ptr()), st), "head_decode")
        ref.append((best, lab, box))
    outs = [(torch.empty(g[0] * g[1] * g[2], device=device), torch.empty(g[0] * g[1] * g[2], dtype=torch.int64, device=device),
             torch.empty(g[0] * g[1] * g[2], 6, device=device)) for g in grids]
    vp = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    vsz = np.concatenate([np.float32(vs0 * f) for f in facs]).astype(np.float32)
    check(lib.ndet_head_decode_levels(3, vp(raws), vp(scales), (ctypes.c_int * 9)(*[v for g in grids for v in g]), (ctypes.c_int * 3)(*facs),
                                      vsz.ctypes.data_as(c_void_p), origin.ctypes.data_as(c_void_p), ncls, c_void_p(valid.data_ptr()), X, Y, Z,
                                      vp([o[0] for o in outs]), vp([o[1] for o in outs]), vp([o[2] for o in outs]), st), "head_decode_levels")
    for (b0, l0, x0), (b1, l1, x1) in zip(ref, outs):
        assert torch.equal(b0, b1) and torch.equal(l0, l1) and torch.equal(x0, x1)
    assert float(ref[0][0].max()) > 0 and int((ref[1][0] == 0).sum()) > 0       # some voxels valid, some masked
