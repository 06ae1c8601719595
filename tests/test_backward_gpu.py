"""Training path: gradients of the HIP forward/backward kernel pairs against PyTorch autograd through the oracle
(the reference's own arithmetic on the CPU).  Tolerance 2e-5 relative to the gradient scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden_meta, load_golden, sub_state
from oracle import nerfdet_oracle as O

pytestmark = pytest.mark.gpu


def _close(a, b, tol=2e-5, what=""):
    s = max(1e-6, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * s, f"{what}: max err {err:.3e} vs scale {s:.3e}"


@pytest.mark.parametrize("name", ["volume_small_s0", "volume_small_s1", "volume_medium_s2"])
@pytest.mark.parametrize("cl_out", [True, False])
def test_backproject_mean_backward(device, name, cl_out):
    from nerfdet_amd.autograd import BackprojectMean
    g = load_golden(name)
    meta = golden_meta(g)
    h, w = meta["img_shape"][0] // 4, meta["img_shape"][1] // 4
    feats = g["features"][:, :, :h, :w].clone().requires_grad_(True)
    vol, valid = O.backproject(feats, g["points"], g["projection"])
    mean, cnt, _ = O.aggregate_views(vol, valid)
    wt = torch.randn_like(mean)
    (mean * wt).sum().backward()
    fd = g["features"].to(device).contiguous(memory_format=torch.channels_last)[:, :, :h, :w].detach().requires_grad_(True)
    out, c = BackprojectMean.apply(fd, g["points"].to(device), g["projection"].to(device), cl_out)
    assert torch.equal(c.cpu(), cnt) and not c.requires_grad
    (out * wt.to(device)).sum().backward()
    _close(fd.grad.cpu(), feats.grad, what="d features")


@pytest.mark.parametrize("name", ["volume_small_s0", "volume_small_s1", "volume_medium_s2"])
def test_extract_volume_backward_matches_oracle_autograd(device, name):
    """d(volume)/d(features, mapping, sigma-MLP) through K2/K1 backward == autograd through the materialised reference path."""
    from nerfdet_amd.radiance_field import VanillaNeRFRadianceField
    from nerfdet_amd.volume import extract_volume
    g = load_golden(name)
    meta = golden_meta(g)
    sd = sub_state(g, "nerf_mlp.")
    width, fdim = sd["mlp.base.hidden_layers.0.weight"].shape[0], sd["mlp.base.hidden_layers.0.weight"].shape[1] - 63
    # oracle side
    feats = g["features"].clone().requires_grad_(True)
    mw, mb = g["mapping.0.weight"].clone().requires_grad_(True), g["mapping.0.bias"].clone().requires_grad_(True)
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    ref = O.extract_volume(feats, g["denorm_images"], meta, g["n_voxels"].tolist(), g["voxel_size"].tolist(), mw, mb, osd)
    wt = torch.randn_like(ref["volume"])
    (ref["volume"] * wt).sum().backward()
    # GPU side
    mlp = VanillaNeRFRadianceField(4, width, 3, fdim, 1, width // 2)
    mlp.load_state_dict(sd)
    mlp.to(device)
    mapping = torch.nn.Sequential(torch.nn.Linear(mw.shape[1], mw.shape[0]))
    mapping.load_state_dict(sub_state(g, "mapping."))
    mapping.to(device)
    fd = g["features"].to(device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = extract_volume(fd, g["denorm_images"].to(device), meta, g["n_voxels"].tolist(), g["voxel_size"].tolist(), mapping, mlp,
                         channels_last_out=False)
    torch.testing.assert_close(out["volume"].detach().cpu(), ref["volume"].detach(), rtol=0, atol=2e-5)
    (out["volume"] * wt.to(device)).sum().backward()
    _close(fd.grad.cpu(), feats.grad, what="d features")
    _close(mapping[0].weight.grad.cpu(), mw.grad, tol=1e-3, what="d mapping.weight")
    _close(mapping[0].bias.grad.cpu(), mb.grad, tol=1e-3, what="d mapping.bias")
    for k in ("mlp.base.hidden_layers.0.weight", "mlp.base.hidden_layers.3.weight", "mlp.sigma_layer.output_layer.weight",
              "mlp.sigma_layer.output_layer.bias"):
        # parameter gradients are long sums with cancellation, evaluated by library GEMMs in another order
        _close(dict(mlp.named_parameters())[k].grad.cpu(), osd[k].grad, tol=2e-3, what=k)


@pytest.mark.parametrize("name", ["rays_small_s0", "rays_small_s1"])
def test_ray_branch_backward_matches_oracle_autograd(device, name):
    """NVS + depth loss gradients w.r.t. the mapped feature map and the NeRF-MLP: K4 / compositing backward vs autograd
    through grid_sample / cumprod on the CPU."""
    from nerfdet_amd import rays
    from nerfdet_amd.radiance_field import VanillaNeRFRadianceField
    g = load_golden(name)
    meta = golden_meta(g)
    sd = sub_state(g, "nerf_mlp.")
    width, fdim = sd["mlp.base.hidden_layers.0.weight"].shape[0], sd["mlp.base.hidden_layers.0.weight"].shape[1] - 63
    s = int(g["n_samples"])
    gt_rgb = torch.rand(g["ray_o"].shape[0], 3)
    gt_depth = torch.rand(g["ray_o"].shape[0], 1) * 4 + 0.5
    # oracle
    f2d = g["features_2d"].clone().requires_grad_(True)
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    ret = O.render_rays_func(g["ray_o"], g["ray_d"], f2d, g["img"], [0.2, 8.0], s, osd, meta, det=False, t_rand=g["t_rand"])
    oc = ret["outputs_coarse"]
    loss_ref = O.nvs_loss(oc["rgb"], gt_rgb, oc["mask"]) + O.depth_loss(oc["depth"], gt_depth, oc["mask"])
    loss_ref.backward()
    assert float(oc["mask"].float().sum()) > 0
    # GPU
    mlp = VanillaNeRFRadianceField(4, width, 3, fdim, 1, width // 2)
    mlp.load_state_dict(sd)
    mlp.to(device)
    fd = g["features_2d"].to(device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = rays.render_rays_func(g["ray_o"].to(device), g["ray_d"].to(device), None, None, fd, g["img"].to(device), None, [0.2, 8.0], s, 4096,
                                mlp, meta, None, "image", det=False, t_rand=g["t_rand"].to(device))
    gc = out["outputs_coarse"]
    m = gc["mask"]
    loss = torch.sum(m.unsqueeze(-1) * (gc["rgb"] - gt_rgb.to(device)) ** 2) / (m.sum() + 1e-6) \
        + torch.sum(m * torch.abs(gc["depth"] - gt_depth.to(device).squeeze(-1))) / (m.sum() + 1e-6)
    torch.testing.assert_close(loss.detach().cpu(), loss_ref.detach(), rtol=1e-5, atol=1e-6)
    loss.backward()
    _close(fd.grad.cpu(), f2d.grad, tol=1e-4, what="d features_2d")
    for k in ("mlp.base.hidden_layers.0.weight", "mlp.rgb_layer.output_layer.weight", "mlp.sigma_layer.output_layer.weight"):
        _close(dict(mlp.named_parameters())[k].grad.cpu(), osd[k].grad, tol=2e-3, what=k)


def test_composite_backward_random(device):
    from nerfdet_amd import rays
    gen = torch.Generator().manual_seed(0)
    r, s = 37, 24
    raw = torch.cat([torch.rand(r, s, 3, generator=gen), 3 * torch.rand(r, s, 1, generator=gen) ** 3], -1)
    raw[3, 5:8, 3] = 40.0  # opaque samples: transmittance collapses to ~1e-10 per sample
    z = torch.sort(torch.rand(r, s, generator=gen) * 7 + 0.2, dim=1)[0]
    w_rgb, w_dep = torch.randn(r, 3, generator=gen), torch.randn(r, generator=gen)
    for white in (False, True):
        a = raw.clone().requires_grad_(True)
        o = O.raw2outputs(a, z, None, white_bkgd=white)
        ((o["rgb"] * w_rgb).sum() + (o["depth"] * w_dep).sum()).backward()
        b = raw.to(device).requires_grad_(True)
        p = rays.raw2outputs(b, z.to(device), None, white_bkgd=white)
        ((p["rgb"] * w_rgb.to(device)).sum() + (p["depth"] * w_dep.to(device)).sum()).backward()
        _close(b.grad.cpu(), a.grad, tol=1e-4, what=f"d raw white={white}")


def test_detector_train_step(device):
    """forward_train + backward on the GPU: every loss finite, gradients reach backbone, FPN, mapping, NeRF-MLP, 3D neck
    and head; two SGD steps reduce the loss."""
    from nerfdet_amd.boxes import DepthInstance3DBoxes
    from nerfdet_amd.config import _wrap
    from nerfdet_amd.presets import nerfdet_cfg
    from nerfdet_amd.registry import build_detector
    from nerfdet_amd import rays
    torch.manual_seed(0)
    cfg = _wrap(nerfdet_cfg(50, n_voxels=(16, 16, 8), voxel_size=(0.4, 0.4, 0.4), depth_supervise=True))
    cfg["model"]["N_rand"], cfg["model"]["N_samples"] = 128, 16
    det = build_detector(cfg["model"], train_cfg=cfg["train_cfg"], test_cfg=cfg["test_cfg"])
    with torch.no_grad():
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
    det.to(device).train()
    n_v, hw, t_views = 6, (64, 96), 2
    meta = O.ring_scene_meta(n_v, hw)
    gen = torch.Generator().manual_seed(0)
    nray = (hw[0] - 20) * (hw[1] - 20)
    ang = torch.rand(1, t_views, nray, generator=gen) * 2 * np.pi
    ray_o = torch.stack([2.0 * torch.cos(ang), 2.0 * torch.sin(ang), 1.0 + 0 * ang], -1)
    ray_d = -ray_o / ray_o.norm(dim=-1, keepdim=True) + 0.3 * torch.randn(1, t_views, nray, 3, generator=gen)
    batch = dict(img=torch.randn(1, n_v, 3, *hw, generator=gen), img_metas=[meta],
                 denorm_images=torch.rand(1, n_v, 3, *hw, generator=gen), lightpos=ray_o, raydirs=ray_d,
                 gt_images=torch.rand(1, t_views, nray, 3, generator=gen), gt_depths=torch.rand(1, t_views, hw[0] - 20, hw[1] - 20, generator=gen) * 5 + 0.5,
                 nerf_sizes=[torch.tensor([[hw[0] - 20, hw[1] - 20, 3]])])
    batch = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    boxes = torch.tensor([[0.0, 0.0, 0.5, 1.6, 1.6, 1.2], [1.2, -1.0, 0.4, 1.2, 2.0, 1.0], [-1.5, 1.0, 0.8, 2.0, 1.2, 1.6]])
    gt_boxes = [DepthInstance3DBoxes(boxes, box_dim=6, with_yaw=False, origin=(0.5, 0.5, 0.5)).to(device)]
    gt_labels = [torch.tensor([2, 7, 11], device=device)]
    params = [p for p in det.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=2e-5)
    losses = []
    for it in range(4):
        rays.rng = np.random.RandomState(234)  # same rays every step so the loss is comparable
        torch.manual_seed(1)
        opt.zero_grad()
        out = det.train_step(dict(batch, gt_bboxes_3d=gt_boxes, gt_labels_3d=gt_labels))
        assert set(out["log_vars"]) >= {"loss_centerness", "loss_bbox", "loss_cls", "loss_nvs", "loss_depth", "loss"}
        assert all(np.isfinite(v) for v in out["log_vars"].values()), out["log_vars"]
        out["loss"].backward()
        if it == 0:
            for name in ("backbone.layer2.0.conv1.weight", "backbone.layer4.2.conv3.weight", "neck.lateral_convs.0.conv.weight",
                         "neck.fpn_convs.0.conv.weight", "mapping.0.weight", "mapping.0.bias", "nerf_mlp.mlp.base.hidden_layers.0.weight",
                         "nerf_mlp.mlp.rgb_layer.output_layer.weight", "neck_3d.down_layer_0.0.conv1.weight", "bbox_head.cls_conv.weight"):
                gr = dict(det.named_parameters())[name].grad
                assert gr is not None and torch.isfinite(gr).all() and float(gr.abs().max()) > 0, name
            assert det.backbone.conv1.weight.grad is None  # frozen stem (config:9)
        torch.nn.utils.clip_grad_norm_(params, 35.0)     # config:173
        opt.step()
        losses.append(out["log_vars"]["loss"])
    assert losses[-1] < losses[0], losses


def test_backward_through_cropped_feature_maps(device):
    """The mapped map handed to K2 / K4 may be an [:h,:w] crop of a padded channels-last map (real ScanNet frames use 59 of 60
    feature rows): gradients must land on the cropped pixels -- same values as for a dense copy of the crop,
    nothing outside it."""
    from nerfdet_amd.autograd import DensityFeatures, RayViewStats
    from nerfdet_amd import ops, rays
    g = load_golden("volume_small_s1")        # generated with img_shape (59, 80): a real crop
    meta = golden_meta(g)
    n_v = g["features"].shape[0]
    cm, h, w = 8, meta["img_shape"][0] // 4, meta["img_shape"][1] // 4
    torch.manual_seed(0)
    padded = torch.randn(n_v, h + 1, w + 2, cm, device=device)            # (n,H,W,C) memory with spare rows and columns
    bias = torch.randn(cm, device=device)
    pts, proj, rgb_proj = g["points"].to(device), g["projection"].to(device), g["rgb_projection"].to(device)
    rgb = g["denorm_images"][:, :, :meta["img_shape"][0], :meta["img_shape"][1]].to(device)
    outs = {}
    for kind in ("crop", "dense"):
        base = padded.clone().requires_grad_(True)
        view = base.permute(0, 3, 1, 2)[:, :, :h, :w]
        m = view if kind == "crop" else view.contiguous(memory_format=torch.channels_last)
        b = bias.clone().requires_grad_(True)
        glob = DensityFeatures.apply(m, b, rgb, pts, proj, rgb_proj)
        wt = torch.randn(glob.shape, generator=torch.Generator().manual_seed(1)).to(device)
        (glob * wt).sum().backward()
        xyz = (torch.rand(40, 6, 3, generator=torch.Generator().manual_seed(2)) * 4 - 2).to(device)
        cams = rays._compute_projection(meta)
        base2 = padded.clone().requires_grad_(True)
        view2 = base2.permute(0, 3, 1, 2)[:, :, :h, :w]
        f = view2 if kind == "crop" else view2.contiguous(memory_format=torch.channels_last)
        gf, _, _ = RayViewStats.apply(f, xyz, rgb, cams)
        wt2 = torch.randn(gf.shape, generator=torch.Generator().manual_seed(3)).to(device)
        (gf * wt2).sum().backward()
        outs[kind] = (glob.detach(), base.grad, b.grad, gf.detach(), base2.grad)
    for a, c in zip(outs["crop"], outs["dense"]):
        _close(a.cpu(), c.cpu(), tol=1e-5, what="crop vs dense")
    for gr in (outs["crop"][1], outs["crop"][4]):
        assert float(gr[:, h:].abs().max()) == 0 and float(gr[:, :, w:].abs().max()) == 0     # nothing outside the crop
        assert float(gr[:, :h, :w].abs().max()) > 0
