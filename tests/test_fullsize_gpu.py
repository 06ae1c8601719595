"""Parity at BASELINE.json's stated sizes: whole ``nerfdet.forward_test`` against the oracle pipeline on the same
weights -- view counts exact, voxel features <= 1e-4 (north_star), identical detection labels / order.

cfg1  nerfdet_res50, 10 views 240x320, 40x40x16 voxels
cfg2  nerfdet_res50, 50 views 240x320, 40x40x16 voxels          (the headline workload of bench.py)
cfg5  nerfdet_res101, 101 views 320x480, 80x80x32 voxels        (the reference would materialise 21 GB: the oracle runs on a
                                                                 random sample of voxels, which are independent of each other)

Where exact equality is defined it is demanded: counts, masks, and the post-processing (decode, top-k, NMS) on identical head
outputs.  The backbone (third-party ResNet/FPN, parity unpinned) and the dense layers run in different arithmetic on the two
sides (MFMA bf16x3 vs PyTorch-CPU fp32), so the end-to-end detections are additionally compared through a fully independent CPU
pipeline with an explicit near-tie allowance."""
import importlib.util
import os

import pytest
import torch

from oracle import nerfdet_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _iou_matrix(a, b):
    lo = torch.max(a[:, None, :3], b[None, :, :3])
    hi = torch.min(a[:, None, 3:6], b[None, :, 3:6])
    inter = (hi - lo).clamp(min=0).prod(-1)
    va, vb = (a[:, 3:6] - a[:, :3]).prod(-1), (b[:, 3:6] - b[:, :3]).prod(-1)
    return inter / (va[:, None] + vb[None, :] - inter).clamp(min=1e-9)


def _corners(centre_size):
    c, s = centre_size[:, :3], centre_size[:, 3:6]
    return torch.cat([c - s / 2, c + s / 2], 1)


def _detections_agree(res, ref, what):
    """Exact label sequence when nothing is near a tie; otherwise every detection must have a counterpart (same label,
    IoU > 0.98, score within 1e-3) in the other list for at least 99 % of either list."""
    got = res["boxes_3d"].tensor[:, :6].clone()
    got[:, 2] += got[:, 5] * 0.5
    if len(res["labels_3d"]) == len(ref["labels"]) and torch.equal(res["labels_3d"], ref["labels"]):
        torch.testing.assert_close(res["scores_3d"], ref["scores"], rtol=1e-3, atol=2e-5)
        torch.testing.assert_close(got, ref["boxes"], rtol=1e-3, atol=1e-3)
        return "identical"
    iou = _iou_matrix(_corners(got), _corners(ref["boxes"]))
    same = (res["labels_3d"][:, None] == ref["labels"][None, :]) & (iou > 0.98) & ((res["scores_3d"][:, None] - ref["scores"][None, :]).abs() < 1e-3)
    f_got, f_ref = same.any(1).float().mean().item(), same.any(0).float().mean().item()
    assert f_got >= 0.99 and f_ref >= 0.99, f"{what}: only {f_got:.3f} / {f_ref:.3f} of the detections have a counterpart"
    return f"near-tie differences: {len(res['labels_3d'])} vs {len(ref['labels'])} detections, matched {f_got:.3f}/{f_ref:.3f}"


def _near_boundary(meta, n_voxels, voxel_size, hw, tol=1e-4):
    """Voxels with a view whose stride-4 or stride-1 pixel coordinate lies within ``tol`` of a .5 rounding boundary: the only
    places where a 1-ulp difference of the projection could pick the neighbouring pixel ("index parity at
    rounding boundaries").  Reported, no longer excluded: they must agree like every other voxel."""
    from test_volume_gpu import near_boundary_voxels
    pts = O.get_points(n_voxels, voxel_size, meta["lidar2img"]["origin"])
    h, w = hw
    return (near_boundary_voxels(pts, O.compute_projection(meta, 4), w // 4, h // 4, tol)
            | near_boundary_voxels(pts, O.compute_projection(meta, 1), w, h, tol)).reshape(-1)


def _gpu_side(det, batch, device):
    import nerfdet_amd.volume as V
    with torch.no_grad():
        x, b, stride = det.extract_2d(batch["img"])
        out = V.extract_volume(x, batch["denorm_images"][0], batch["img_metas"][0], det.n_voxels, det.voxel_size, det.mapping, det.nerf_mlp,
                               stride=stride, channels_last_out=True)
        x3 = det.neck_3d(out["volume"].unsqueeze(0))
        ctr, reg, cls = det.bbox_head(x3)
        res = det(return_loss=False, **{k: (list(v) if isinstance(v, list) else v) for k, v in batch.items()})[0]
    return x, out, x3, (ctr, reg, cls), res


@pytest.mark.parametrize("workload", ["cfg1", "cfg2"])
def test_forward_test_at_baseline_size_vs_oracle(device, workload):
    bench = _bench()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    w = bench.WORKLOADS[workload]
    det_cpu = bench.build_model(w)
    batch_cpu = bench.synth_batch(w, 0)
    meta = batch_cpu["img_metas"][0]
    tc = det_cpu.bbox_head.test_cfg
    import copy
    det = copy.deepcopy(det_cpu).to(device)
    feats_gpu, out, x3, (ctr, reg, cls), res = _gpu_side(det, bench.to_device(batch_cpu, device), device)

    # --- the hot path against the oracle on the SAME feature maps: counts exact, voxel features <= 1e-4
    f_host = feats_gpu.float().cpu().contiguous()
    with torch.no_grad():
        ov = O.extract_volume(f_host, batch_cpu["denorm_images"][0], meta, w["n_voxels"], w["voxel_size"], det_cpu.mapping[0].weight,
                              det_cpu.mapping[0].bias, det_cpu.nerf_mlp.state_dict())
    # no exclusion band: the oracle evaluates the projection in the build container's operation order on any host (oracle.PINNED_ARITHMETIC,
    # pinned to the real reference at this size by tests/golden/fullsize_cfg*.npz), so rounding-boundary voxels must agree too
    excl = _near_boundary(meta, w["n_voxels"], w["voxel_size"], w["img_hw"])
    cnt_bad = (out["valid"].cpu() != ov["valid"]).reshape(-1)
    assert not cnt_bad.any(), f"view counts differ from the oracle in {int(cnt_bad.sum())} voxels ({int((cnt_bad & excl).sum())} of them near a rounding boundary)"
    assert float((ov["valid"] > 0).float().mean()) > 0.2
    scale = max(1.0, float(ov["volume"].abs().max()))
    verr = (out["volume"].cpu() - ov["volume"]).abs().reshape(256, -1).max(0)[0]
    err = float(verr.max())
    assert err <= 1e-4 * scale, f"gated voxel features differ from the oracle by {err} (scale {scale})"
    n_flip = int((verr > 1e-4 * scale).sum())
    seen = ov["valid"].reshape(-1) > 0     # unseen voxels: the reference's n_v*b/1e-8 "mean", zeroed by the gating
    gerr = float((out["global_feat"].cpu() - ov["global_feat"])[seen].abs().max())
    assert gerr <= 1e-4 * max(1.0, float(ov["global_feat"][seen].abs().max())), gerr

    # --- 3D neck + head against the oracle on the SAME volume (the GPU's; 40x40x16 at width 256 -> 128)
    with torch.no_grad():
        n3 = O.neck3d_forward(dict(det_cpu.neck_3d.state_dict()), out["volume"].cpu().contiguous().unsqueeze(0))
        octr, oreg, ocls = O.head_forward(det_cpu.bbox_head.state_dict(), n3)
    for lvl in range(3):
        s = max(1.0, float(n3[lvl].abs().max()))
        assert float((x3[lvl].cpu() - n3[lvl]).abs().max()) <= 1e-4 * s, f"neck level {lvl}"
        assert float((cls[lvl].cpu() - ocls[lvl]).abs().max()) <= 2e-4 * max(1.0, float(ocls[lvl].abs().max())), f"cls logits level {lvl}"
        assert float((ctr[lvl].cpu() - octr[lvl]).abs().max()) <= 2e-4 * max(1.0, float(octr[lvl].abs().max())), f"centerness level {lvl}"
        torch.testing.assert_close(reg[lvl].cpu(), oreg[lvl], rtol=2e-4, atol=1e-5)

    # --- post-processing on IDENTICAL head outputs: identical box indices (labels, order), scores, boxes
    valid_f = out["valid"].cpu().unsqueeze(0).float()
    same_in = O.head_get_bboxes([t.cpu() for t in ctr], [t.cpu() for t in reg], [t.cpu() for t in cls], valid_f, meta["lidar2img"]["origin"],
                                w["voxel_size"], tc.nms_pre, tc.score_thr, tc.iou_thr)
    assert len(same_in["labels"]) > 50 and same_in["labels"].unique().numel() > 5, "the workload must exercise NMS"
    assert torch.equal(res["labels_3d"], same_in["labels"]), "detections differ from sequential NMS on the same head outputs"
    torch.testing.assert_close(res["scores_3d"], same_in["scores"], rtol=1e-5, atol=1e-6)
    got = res["boxes_3d"].tensor[:, :6].clone()
    got[:, 2] += got[:, 5] * 0.5
    torch.testing.assert_close(got, same_in["boxes"], rtol=1e-5, atol=1e-5)

    # --- a fully independent CPU pipeline (PyTorch-CPU ResNet/FPN -> oracle -> sequential NMS)
    with torch.no_grad():
        feats_cpu = det_cpu.neck(det_cpu.backbone(batch_cpu["img"][0]))[0]
        fs = max(1.0, float(feats_cpu.abs().max()))
        assert float((f_host - feats_cpu).abs().max()) <= 1e-4 * fs, "FPN level 0 differs from PyTorch-CPU"
        ov2 = O.extract_volume(feats_cpu, batch_cpu["denorm_images"][0], meta, w["n_voxels"], w["voxel_size"], det_cpu.mapping[0].weight,
                               det_cpu.mapping[0].bias, det_cpu.nerf_mlp.state_dict())
        assert float((out["volume"].cpu() - ov2["volume"]).abs().max()) <= 1e-4 * scale
        n3b = O.neck3d_forward(dict(det_cpu.neck_3d.state_dict()), ov2["volume"].unsqueeze(0))
        ref = O.head_get_bboxes(*O.head_forward(det_cpu.bbox_head.state_dict(), n3b), ov2["valid"].unsqueeze(0).float(), meta["lidar2img"]["origin"], w["voxel_size"],
                                tc.nms_pre, tc.score_thr, tc.iou_thr)
    verdict = _detections_agree(res, ref, workload)
    print(f"{workload}: volume err {err:.2e} (scale {scale:.2f}), {int(excl.sum())} boundary voxels of which {n_flip} picked the neighbouring "
          f"pixel, {len(res['labels_3d'])} detections, end-to-end vs CPU pipeline: {verdict}")


def test_cfg5_res101_101_views_320x480_80x80x32(device):
    """BASELINE configs[4] as stated.  Hot path: oracle on 4 096 sampled voxels (counts exact, gated features and conditioning rows
    <= 1e-4).  Dense part: the oracle's neck/head on the GPU's own volume; post-processing exact on identical head outputs."""
    import copy
    import nerfdet_amd.volume as V  # noqa: F401
    bench = _bench()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    w = bench.WORKLOADS["cfg5"]
    assert w["n_views"] == 101 and tuple(w["img_hw"]) == (320, 480) and tuple(w["n_voxels"]) == (80, 80, 32) and w["depth"] == 101
    det_cpu = bench.build_model(w)
    assert len(det_cpu.backbone.layer3) == 23, "ResNet-101"
    batch_cpu = bench.synth_batch(w, 0)
    meta = batch_cpu["img_metas"][0]
    tc = det_cpu.bbox_head.test_cfg
    det = copy.deepcopy(det_cpu).to(device)
    feats_gpu, out, x3, (ctr, reg, cls), res = _gpu_side(det, bench.to_device(batch_cpu, device), device)
    assert feats_gpu.shape == (101, 256, 80, 120)
    f_host = feats_gpu.float().cpu().contiguous()
    n = out["valid"].numel()
    g = torch.Generator().manual_seed(9)
    sel = torch.randperm(n, generator=g)[:4096]
    pts = O.get_points(w["n_voxels"], w["voxel_size"], meta["lidar2img"]["origin"])
    assert torch.equal(out["points"].cpu(), pts)
    sub = pts.reshape(3, -1)[:, sel].reshape(3, -1, 1, 1).contiguous()
    with torch.no_grad():
        proj, rgb_proj = O.compute_projection(meta, 4), O.compute_projection(meta, 1)
        vol, valid = O.backproject(f_host, sub, proj)
        mean, cnt, _ = O.aggregate_views(vol, valid)
        rgb_vol, _ = O.backproject(batch_cpu["denorm_images"][0], sub, rgb_proj)
        glob = O.density_features(vol, rgb_vol, cnt, det_cpu.mapping[0].weight, det_cpu.mapping[0].bias)
        dens = O.nerf_query_density(det_cpu.nerf_mlp.state_dict(), sub.view(3, -1).permute(1, 0).contiguous(), glob)
        exp = O.gate_volume(mean, cnt, dens)
    cnt_bad = out["valid"].reshape(-1)[sel.to(device)].cpu() != cnt.reshape(-1)
    assert not cnt_bad.any(), f"view counts differ from the oracle in {int(cnt_bad.sum())} of the sampled voxels"
    assert int(cnt.max()) > 20 and float((cnt > 0).float().mean()) > 0.2
    gv = out["volume"].reshape(256, -1)[:, sel.to(device)].cpu()
    scale = max(1.0, float(exp.abs().max()))
    err = float((gv - exp.reshape(256, -1)).abs().max())
    assert err <= 1e-4 * scale, f"gated voxel features differ from the oracle by {err}"
    seen = cnt.reshape(-1) > 0
    gerr = float((out["global_feat"][sel.to(device)].cpu() - glob)[seen].abs().max())
    assert gerr <= 1e-4 * max(1.0, float(glob[seen].abs().max())), gerr
    # dense part on the GPU's own volume
    with torch.no_grad():
        n3 = O.neck3d_forward(dict(det_cpu.neck_3d.state_dict()), out["volume"].cpu().contiguous().unsqueeze(0))
        octr, oreg, ocls = O.head_forward(det_cpu.bbox_head.state_dict(), n3)
    for lvl in range(3):
        s = max(1.0, float(n3[lvl].abs().max()))
        assert float((x3[lvl].cpu() - n3[lvl]).abs().max()) <= 1e-4 * s, f"neck level {lvl}"
        assert float((cls[lvl].cpu() - ocls[lvl]).abs().max()) <= 2e-4 * max(1.0, float(ocls[lvl].abs().max())), f"cls logits level {lvl}"
    same_in = O.head_get_bboxes([t.cpu() for t in ctr], [t.cpu() for t in reg], [t.cpu() for t in cls], out["valid"].cpu().unsqueeze(0).float(),
                                meta["lidar2img"]["origin"], w["voxel_size"], tc.nms_pre, tc.score_thr, tc.iou_thr)
    assert len(same_in["labels"]) > 50
    assert torch.equal(res["labels_3d"], same_in["labels"])
    torch.testing.assert_close(res["scores_3d"], same_in["scores"], rtol=1e-5, atol=1e-6)
    print(f"cfg5: volume err {err:.2e} (scale {scale:.2f}), {len(res['labels_3d'])} detections from {len(same_in['cand_scores'])} candidates")


def test_cfg4_depth_supervised_train_step_losses_vs_oracle(device):
    """BASELINE configs[3] (nerfdet_res50_2x_low_res_depth_sp, 50 sampled views -> 10 NeRF targets + 40 sources 240x320, 2 048 rays x 64
    samples, depth supervision): the five losses of one ``forward_train`` on the GPU (HIP autograd Functions, MFMA training
    convolutions) against the same module evaluated on the CPU with the ORACLE standing in for the HIP ops (tests/cpu_detector.py),
    same weights, same rays, deterministic sampling; then the backward runs and every trainable group receives a finite gradient.
    cfg3 is the same step per rank (DDP: tests/test_ddp.py) and, with ``set_arithmetic("bf16")``, in bf16: losses within 2 %."""
    import copy
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_detector import oracle_backed_cpu_ops
    from nerfdet_amd import conv3d, rays
    from nerfdet_amd.presets import build_nerfdet
    from nerfdet_amd.synth import batch_to, train_scene
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    torch.manual_seed(0)
    det_cpu = build_nerfdet(50, depth_supervise=True)
    with torch.no_grad():
        det_cpu.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det_cpu.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
    det_cpu.train()
    scene = train_scene(40, (240, 320), t_views=10, n_boxes=8, seed=4)
    det = copy.deepcopy(det_cpu).to(device).train()
    with oracle_backed_cpu_ops() as holder, torch.no_grad():
        holder["rng"] = np.random.RandomState(234)
        ref = det_cpu.train_step(scene)["log_vars"]
    orig = rays.sample_along_camera_ray
    rays.sample_along_camera_ray = lambda *a, **k: orig(*a, **{**k, "det": True})     # the oracle stand-in samples deterministically
    got = {}
    try:
        for mode in ("bf16", "bf16x3"):          # fp32-class last: its gradients are the ones inspected below
            prev = conv3d.set_arithmetic(mode)
            try:
                rays.rng = np.random.RandomState(234)
                det.zero_grad(set_to_none=True)
                out = det.train_step(batch_to(scene, device))
                got[mode] = out["log_vars"]
                if mode == "bf16x3":
                    out["loss"].backward()
            finally:
                conv3d.set_arithmetic(prev)
    finally:
        rays.sample_along_camera_ray = orig
    keys = ("loss_centerness", "loss_bbox", "loss_cls", "loss_nvs", "loss_depth", "loss")
    assert set(keys) <= set(ref) and all(np.isfinite(ref[k]) for k in keys)
    for k in keys:
        assert abs(got["bf16x3"][k] - ref[k]) <= 2e-3 * max(1.0, abs(ref[k])), (k, got["bf16x3"][k], ref[k])
        assert abs(got["bf16"][k] - ref[k]) <= 2e-2 * max(1.0, abs(ref[k])), (k, got["bf16"][k], ref[k])
    named = dict(det.named_parameters())
    for name in ("backbone.layer2.0.conv1.weight", "backbone.layer4.2.conv3.weight", "neck.fpn_convs.0.conv.weight", "mapping.0.weight",
                 "nerf_mlp.mlp.rgb_layer.output_layer.weight", "neck_3d.down_layer_0.0.conv1.weight", "neck_3d.out_block_2.0.weight",
                 "bbox_head.cls_conv.weight", "bbox_head.reg_conv.weight"):
        g = named[name].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0, name
    print("cfg4 losses (oracle-backed CPU | GPU fp32-class | GPU bf16):", {k: (round(ref[k], 5), round(got["bf16x3"][k], 5), round(got["bf16"][k], 5)) for k in keys})
