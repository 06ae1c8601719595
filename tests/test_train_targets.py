"""A16: FCOS-3D target assignment and the three head losses against vectors produced by the reference's own
``get_targets`` / ``_loss_single`` / ``loss`` / ``compute_centerness`` / ``AxisAlignedIoULoss``
(tests/golden/make_golden_train.py; imvoxel_head_v2.py:65-203,457-526,558-566, axis_aligned_iou_loss.py:9-78,
iou3d_calculator.py:201-330).  Bars: labels / assignment bit-exact, targets <= 1e-6, losses <= 1e-5."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import nerfdet_oracle as O
from oracle import train_oracle as TO

FIXTURES = ["train_targets_s0", "train_targets_s1"]


def _gt(g, b, device="cpu"):
    from nerfdet_amd.boxes import DepthInstance3DBoxes
    boxes = DepthInstance3DBoxes(g[f"gt_tensor_{b}"].to(device), box_dim=7, with_yaw=False)
    return boxes, g[f"gt_labels_{b}"].to(device)


def _head(g, device="cpu"):
    from nerfdet_amd.config import ConfigDict
    from nerfdet_amd.head import ScanNetImVoxelHeadV2
    head = ScanNetImVoxelHeadV2(n_classes=18, n_channels=8, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18,
                                test_cfg=ConfigDict(nms_pre=1000, iou_thr=0.25, score_thr=0.01))
    head.voxel_size = tuple(float(v) for v in g["voxel_size"])
    return head.to(device)


# ---------------------------------------------------------------------------------------------------------------
# CPU: the oracle and the product's torch code against the reference's vectors
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", FIXTURES)
def test_box_container_matches_reference_structure(name):
    g = load_golden(name)
    for b in range(int(g["batch"])):
        boxes, _ = _gt(g, b)
        assert torch.equal(boxes.volume, g[f"gt_volume_{b}"])
        assert torch.equal(boxes.gravity_center, g[f"gt_gravity_center_{b}"])


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_targets_and_losses_match_reference(name):
    g = load_golden(name)
    grid, vs, origin = g["grid"].tolist(), g["voxel_size"].tolist(), g["origin"].numpy()
    pts = TO.level_points(grid, vs, origin)
    sizes = [[s // 2 ** i for s in grid] for i in range(3)]
    valids = TO.level_valids(g["valid"], sizes)
    singles = []
    for b in range(int(g["batch"])):
        gc, size, lab = g[f"gt_gravity_center_{b}"], g[f"gt_tensor_{b}"][:, 3:6], g[f"gt_labels_{b}"]
        ct, bt, lb = TO.get_targets(pts, gc, size, lab)
        assert torch.equal(lb, g[f"tgt_labels_{b}"])
        assert torch.equal(bt, g[f"tgt_bbox_{b}"]), "the owning box of every location (incl. background -> box 0) must agree"
        torch.testing.assert_close(ct, g[f"tgt_centerness_{b}"], rtol=0, atol=1e-6, equal_nan=True)
        ls = TO.loss_single([g[f"ctr_{i}"][b] for i in range(3)], [g[f"reg_{i}"][b] for i in range(3)], [g[f"cls_{i}"][b] for i in range(3)],
                            [v[b] for v in valids], pts, gc, size, lab)
        torch.testing.assert_close(torch.stack(ls), g[f"loss_single_{b}"], rtol=1e-5, atol=1e-5)
        singles.append(torch.stack(ls))
    torch.testing.assert_close(torch.stack(singles).mean(0), g["loss_total"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(TO.compute_centerness(g["cc_in"]), g["cc_out"], rtol=0, atol=1e-6)
    torch.testing.assert_close(TO.aligned_iou_pairs(g["iou_a"], g["iou_b"]), g["iou_pair"], rtol=0, atol=1e-6)
    torch.testing.assert_close(TO.iou_loss(g["iou_a"], g["iou_b"]), torch.as_tensor(g["iou_loss_mean"]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(TO.iou_loss(g["iou_a"], g["iou_b"], g["iou_w"], g["iou_w"].sum()), torch.as_tensor(g["iou_loss_weighted"]), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", FIXTURES)
def test_product_targets_and_iou_loss_match_reference_on_cpu(name):
    """The product's assignment / IoU-loss code is plain tensor algebra: held to the reference's vectors on the CPU too
    (lattice points from the oracle here; the HIP ``get_points`` feeds it in the GPU test below)."""
    from nerfdet_amd.head import compute_centerness
    from nerfdet_amd.losses import AxisAlignedIoULoss, aligned_iou_3d
    g = load_golden(name)
    head = _head(g)
    pts = TO.level_points(g["grid"].tolist(), g["voxel_size"].tolist(), g["origin"].numpy())
    for b in range(int(g["batch"])):
        boxes, lab = _gt(g, b)
        ct, bt, lb = head.get_targets(pts, boxes, lab)
        assert torch.equal(lb, g[f"tgt_labels_{b}"])
        assert torch.equal(bt, g[f"tgt_bbox_{b}"])
        torch.testing.assert_close(ct, g[f"tgt_centerness_{b}"], rtol=0, atol=1e-6, equal_nan=True)
    torch.testing.assert_close(compute_centerness(g["cc_in"]), g["cc_out"], rtol=0, atol=1e-6)
    torch.testing.assert_close(aligned_iou_3d(g["iou_a"], g["iou_b"]), g["iou_pair"], rtol=0, atol=1e-6)
    il = AxisAlignedIoULoss()
    w = g["iou_w"]
    torch.testing.assert_close(il(g["iou_a"], g["iou_b"]), torch.as_tensor(g["iou_loss_mean"]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(il(g["iou_a"], g["iou_b"], weight=w, avg_factor=w.sum()), torch.as_tensor(g["iou_loss_weighted"]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(il(g["iou_a"], g["iou_b"], reduction_override="none"), g["iou_loss_none"], rtol=0, atol=1e-6)
    torch.testing.assert_close(il(g["iou_a"], g["iou_b"], weight=w, reduction_override="sum"), torch.as_tensor(g["iou_loss_sum"]), rtol=1e-6, atol=1e-5)
    zw = torch.zeros(300, 1).expand(300, 6)
    torch.testing.assert_close(il(g["iou_a"], g["iou_b"], weight=zw), torch.as_tensor(g["iou_loss_zero_weight"]))


# ---------------------------------------------------------------------------------------------------------------
# GPU: the head as the training step runs it (HIP lattice kernel, device tensors)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_head_targets_and_losses_on_gpu_match_reference(device, name):
    g = load_golden(name)
    head = _head(g, device)
    grid, origin = g["grid"].tolist(), g["origin"].numpy()
    sizes = [tuple(s // 2 ** i for s in grid) for i in range(3)]
    pts = head.get_points(sizes, origin, device)
    ref_pts = TO.level_points(grid, g["voxel_size"].tolist(), origin)
    for p, r in zip(pts, ref_pts):
        assert torch.equal(p.cpu(), r)
    metas = [dict(lidar2img=dict(origin=origin)) for _ in range(int(g["batch"]))]
    gts, labs = zip(*[_gt(g, b, device) for b in range(int(g["batch"]))])
    for b in range(len(metas)):
        ct, bt, lb = head.get_targets(pts, gts[b], labs[b])
        assert torch.equal(lb.cpu(), g[f"tgt_labels_{b}"])
        pos = g[f"tgt_labels_{b}"] >= 0     # background rows follow an arbitrary box on the GPU (min over an all-1e8 row)
        assert torch.equal(bt.cpu()[pos], g[f"tgt_bbox_{b}"][pos])
        torch.testing.assert_close(ct.cpu()[pos], g[f"tgt_centerness_{b}"][pos], rtol=0, atol=1e-6)
    ctr, reg, cls = ([g[f"{k}_{i}"].to(device) for i in range(3)] for k in ("ctr", "reg", "cls"))
    valid = g["valid"].to(device)
    valids = head._level_valids(ctr, valid)
    for b in range(len(metas)):
        ls = head._loss_single([x[b] for x in ctr], [x[b] for x in reg], [x[b] for x in cls], [v[b] for v in valids], metas[b], gts[b], labs[b])
        torch.testing.assert_close(torch.stack(ls).cpu(), g[f"loss_single_{b}"], rtol=1e-5, atol=1e-5)
    tot = head.loss(ctr, reg, cls, valid, metas, list(gts), list(labs))
    got = torch.stack([tot["loss_centerness"], tot["loss_bbox"], tot["loss_cls"]]).cpu()
    torch.testing.assert_close(got, g["loss_total"], rtol=1e-5, atol=1e-5)
    lz = head._loss_single([x[0] for x in ctr], [x[0] for x in reg], [x[0] for x in cls], [torch.zeros_like(v[0]) for v in valids],
                           metas[0], gts[0], labs[0])
    torch.testing.assert_close(torch.stack(lz).cpu(), g["loss_single_novalid"])


@pytest.mark.gpu
def test_head_loss_at_cfg2_size_matches_oracle(device):
    """BASELINE's grid (40x40x16 -> 29 200 locations over 3 levels), 12 GT boxes: assignment identical to the (golden-pinned)
    oracle, losses to 1e-5, and the gradient of the loss w.r.t. the predictions agrees with autograd through the oracle."""
    from nerfdet_amd.boxes import DepthInstance3DBoxes
    from nerfdet_amd.config import ConfigDict
    from nerfdet_amd.head import ScanNetImVoxelHeadV2
    torch.manual_seed(3)
    rng = np.random.RandomState(3)
    grid, vs, origin = (40, 40, 16), (0.16, 0.16, 0.2), np.array([0.0, 0.0, 0.5], dtype=np.float32)
    head = ScanNetImVoxelHeadV2(n_classes=18, n_channels=128, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18,
                                test_cfg=ConfigDict(nms_pre=1000, iou_thr=0.25, score_thr=0.01))
    head.voxel_size = vs
    head.to(device)
    n_box = 12
    ext = np.array([6.4, 6.4, 3.2], dtype=np.float32)
    ctr_b = (rng.rand(n_box, 3).astype(np.float32) - 0.5) * ext * 0.8 + origin
    size = (0.2 + rng.rand(n_box, 3).astype(np.float32) * 1.8).astype(np.float32)
    t7 = torch.from_numpy(np.concatenate([ctr_b, size, np.zeros((n_box, 1), np.float32)], 1))
    boxes = DepthInstance3DBoxes(t7, box_dim=7, with_yaw=False, origin=(0.5, 0.5, 0.5))
    labels = torch.from_numpy(rng.randint(0, 18, n_box))
    sizes = [tuple(s // 2 ** i for s in grid) for i in range(3)]
    ctr = [torch.randn(1, 1, *s, requires_grad=True) for s in sizes]
    reg = [(torch.exp(0.3 * torch.randn(1, 6, *s)) * 0.3 * 2 ** i).requires_grad_() for i, s in enumerate(sizes)]
    cls = [(torch.randn(1, 18, *s) - 2).requires_grad_() for s in sizes]
    valid = (torch.rand(1, 1, *grid) < 0.6).float() * torch.randint(1, 9, (1, 1, *grid)).float()
    pts = TO.level_points(grid, vs, origin)
    ref = TO.loss_single([x[0] for x in ctr], [x[0] for x in reg], [x[0] for x in cls], [v[0] for v in TO.level_valids(valid, sizes)], pts,
                         boxes.gravity_center, boxes.tensor[:, 3:6], labels)
    ref_tgt = TO.get_targets(pts, boxes.gravity_center, boxes.tensor[:, 3:6], labels)
    assert int((ref_tgt[2] >= 0).sum()) > 100
    torch.stack(ref).sum().backward()
    ref_grads = [x.grad.clone() for x in ctr + reg + cls]
    d = lambda ts: [t.detach().to(device).requires_grad_() for t in ts]
    ctr_g, reg_g, cls_g = d(ctr), d(reg), d(cls)
    got_tgt = head.get_targets(head.get_points(sizes, origin, device), boxes.to(device), labels.to(device))
    assert torch.equal(got_tgt[2].cpu(), ref_tgt[2])
    pos = ref_tgt[2] >= 0
    assert torch.equal(got_tgt[1].cpu()[pos], ref_tgt[1][pos])
    tot = head.loss(ctr_g, reg_g, cls_g, valid.to(device), [dict(lidar2img=dict(origin=origin))], [boxes.to(device)], [labels.to(device)])
    got = torch.stack([tot["loss_centerness"], tot["loss_bbox"], tot["loss_cls"]])
    torch.testing.assert_close(got.detach().cpu(), torch.stack(ref).detach(), rtol=1e-5, atol=1e-5)
    got.sum().backward()
    for a, r in zip(ctr_g + reg_g + cls_g, ref_grads):
        torch.testing.assert_close(a.grad.cpu(), r, rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
def test_head_loss_without_positives_is_zero_with_finite_gradients(device):
    """No location inside any box (imvoxel_head_v2.py:196-199 returns ``pred[pos].sum()`` = 0 for the centerness and box terms): the
    masked GPU form gives exact zeros there, the classification term equals the CPU (gathered, reference-structured) form, and no
    gradient is NaN -- the rows outside the mask are evaluated on benign operands, not multiplied out after the fact."""
    from nerfdet_amd.boxes import DepthInstance3DBoxes
    from nerfdet_amd.config import ConfigDict
    from nerfdet_amd.head import ScanNetImVoxelHeadV2
    torch.manual_seed(5)
    grid, vs, origin = (16, 16, 8), (0.16, 0.16, 0.2), np.array([0.0, 0.0, 0.5], dtype=np.float32)
    head = ScanNetImVoxelHeadV2(n_classes=18, n_channels=8, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18,
                                test_cfg=ConfigDict(nms_pre=1000, iou_thr=0.25, score_thr=0.01))
    head.voxel_size = vs
    far = torch.tensor([[50.0, 50.0, 50.0, 0.5, 0.5, 0.5, 0.0], [-40.0, 3.0, 1.0, 0.3, 0.3, 0.3, 0.0]])
    labels = torch.tensor([3, 7])
    sizes = [tuple(s // 2 ** i for s in grid) for i in range(3)]
    ctr = [torch.randn(1, 1, *s) for s in sizes]
    reg = [torch.randn(1, 6, *s) * 3 for s in sizes]          # negative distances included: degenerate boxes on unmasked rows would be NaN
    cls = [torch.randn(1, 18, *s) for s in sizes]
    valid = (torch.rand(1, 1, *grid) < 0.5).float()
    meta = [dict(lidar2img=dict(origin=origin))]

    def run(dev):
        h = head.to(dev)
        leaves = [t.clone().to(dev).requires_grad_() for t in ctr + reg + cls]
        c, r, k = leaves[:3], leaves[3:6], leaves[6:]
        tot = h.loss(c, r, k, valid.to(dev), meta, [DepthInstance3DBoxes(far.to(dev), box_dim=7, with_yaw=False, origin=(0.5, 0.5, 0.5))], [labels.to(dev)])
        sum(tot.values()).backward()
        return {n: float(v) for n, v in tot.items()}, [t.grad.cpu() for t in leaves]
    from cpu_detector import oracle_backed_cpu_ops
    got, g_gpu = run(device)
    with oracle_backed_cpu_ops():                      # the package has no CPU lattice kernel: the oracle's stands in
        ref, g_cpu = run("cpu")
    assert got["loss_centerness"] == 0.0 and got["loss_bbox"] == 0.0 and ref["loss_centerness"] == 0.0 and ref["loss_bbox"] == 0.0
    assert abs(got["loss_cls"] - ref["loss_cls"]) <= 1e-5 * max(1.0, abs(ref["loss_cls"]))
    for a, b in zip(g_gpu, g_cpu):
        assert torch.isfinite(a).all()
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-7)
