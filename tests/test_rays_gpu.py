"""GPU parity for the NeRF ray branch (A7-A12) through the C ABI, against the golden vectors the real
reference produced (tests/golden/rays_*.npz) and against the oracle at training size.
Tolerances: masks / counts bit-exact; sampled features, statistics, colours, depths <= 2e-5 absolute."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden, sub_state
from oracle import nerfdet_oracle as O

pytestmark = pytest.mark.gpu
ATOL = 2e-5


def _mlp(g, device):
    from nerfdet_amd.radiance_field import VanillaNeRFRadianceField
    sd = sub_state(g, "nerf_mlp.")
    width = sd["mlp.base.hidden_layers.0.weight"].shape[0]
    fdim = sd["mlp.base.hidden_layers.0.weight"].shape[1] - 63
    m = VanillaNeRFRadianceField(4, width, 3, fdim, 1, width // 2)
    m.load_state_dict(sd)
    return m.to(device).eval()


@pytest.mark.parametrize("name", ["rays_small_s0", "rays_small_s1"])
def test_ray_branch_pieces_match_reference_golden(device, name):
    from nerfdet_amd import rays
    g = load_golden(name)
    meta = golden_meta(g)
    s = int(g["n_samples"])
    o, d = g["ray_o"].to(device), g["ray_d"].to(device)
    # A9: deterministic and replayed-jitter sampling
    pts_det, z_det = rays.sample_along_camera_ray(o, d, [0.2, 8.0], s, det=True)
    torch.testing.assert_close(z_det.cpu(), g["z_det"], rtol=0, atol=1e-6)
    torch.testing.assert_close(pts_det.cpu(), g["pts_det"], rtol=0, atol=2e-6)
    pts, z = rays.sample_along_camera_ray(o, d, [0.2, 8.0], s, det=False, t_rand=g["t_rand"].to(device))
    torch.testing.assert_close(z.cpu(), g["z_rnd"], rtol=0, atol=1e-6)
    torch.testing.assert_close(pts.cpu(), g["pts_rnd"], rtol=0, atol=2e-6)
    with pytest.raises(AssertionError):
        rays.sample_along_camera_ray(o, d, [0.0, 8.0], s, det=True)  # render_ray.py:161
    # A1 twin
    cams = rays._compute_projection(meta)
    assert torch.equal(cams, g["cameras"])
    # A7 exact API on the golden sample points
    gp = g["pts_rnd"].to(device)
    imgs = g["img"].to(device).permute(0, 2, 3, 1).unsqueeze(0)
    feat = g["features_2d"].to(device)
    rgb_feat, mask = rays.Projector().compute(gp, imgs, cams.to(device), feat, grid_sample=True)
    assert torch.equal(mask.cpu(), g["mask"])
    torch.testing.assert_close(rgb_feat.cpu(), g["rgb_feat"], rtol=0, atol=ATOL)
    # A8 on materialised samples, and the fused A7+A8 kernel
    mean, var = rays.compute_mask_points(rgb_feat, mask)
    torch.testing.assert_close(mean.cpu(), g["mean"], rtol=0, atol=ATOL)
    torch.testing.assert_close(var.cpu(), g["var"], rtol=0, atol=ATOL)
    glob, pm, vc = rays.ray_view_stats(gp, g["img"].to(device), cams, feat.contiguous(memory_format=torch.channels_last))
    ref_glob = torch.cat([g["mean"], g["var"]], dim=-1).squeeze(2)
    torch.testing.assert_close(glob.cpu(), ref_glob, rtol=0, atol=ATOL)
    assert torch.equal(vc.cpu().long(), g["mask"][..., 0].sum(dim=2).long())
    assert torch.equal(pm.cpu(), g["mask"][..., 0].sum(dim=2) > 1)
    # A10 (library GEMMs) + A11
    mlp = _mlp(g, device)
    with torch.no_grad():
        rgb_pts, sigma_pts = mlp(gp, d, ref_glob.to(device))
    torch.testing.assert_close(rgb_pts.cpu(), g["rgb_pts"], rtol=1e-5, atol=ATOL)
    torch.testing.assert_close(sigma_pts.cpu(), g["sigma_pts"], rtol=1e-5, atol=ATOL)
    raw = torch.cat([g["rgb_pts"], g["sigma_pts"]], -1).to(device)
    comp = rays.raw2outputs(raw, g["z_rnd"].to(device), (g["mask"][..., 0].sum(dim=2) > 1).to(device))
    for k, gk in [("rgb", "comp_rgb"), ("depth", "comp_depth"), ("weights", "comp_weights"), ("alpha", "comp_alpha"),
                  ("transparency", "comp_T")]:
        torch.testing.assert_close(comp[k].cpu(), g[gk], rtol=1e-5, atol=2e-6)
    assert torch.equal(comp["mask"].cpu(), g["comp_mask"])
    comp2 = rays.raw2outputs(g["raw_rand"].to(device), g["z_rnd"].to(device), None, white_bkgd=True)
    torch.testing.assert_close(comp2["rgb"].cpu(), g["comp2_rgb"], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(comp2["depth"].cpu(), g["comp2_depth"], rtol=1e-5, atol=2e-6)
    assert comp2["mask"] is None


@pytest.mark.parametrize("name", ["rays_small_s0", "rays_small_s1"])
def test_render_rays_func_matches_reference_golden(device, name):
    from nerfdet_amd import rays
    g = load_golden(name)
    mlp = _mlp(g, device)
    with torch.no_grad():
        ret = rays.render_rays_func(g["ray_o"].to(device), g["ray_d"].to(device), None, None, g["features_2d"].to(device),
                                    g["img"].to(device), None, [0.2, 8.0], int(g["n_samples"]), 4096, mlp, golden_meta(g), None,
                                    "image", det=True)
    oc = ret["outputs_coarse"]
    torch.testing.assert_close(oc["rgb"].cpu(), g["func_rgb"], rtol=1e-5, atol=ATOL)
    torch.testing.assert_close(oc["depth"].cpu(), g["func_depth"], rtol=1e-5, atol=ATOL)
    torch.testing.assert_close(oc["weights"].cpu(), g["func_weights"], rtol=1e-5, atol=ATOL)
    torch.testing.assert_close(ret["sigma"].cpu(), g["func_sigma"], rtol=1e-5, atol=ATOL)
    assert torch.equal(oc["mask"].cpu(), g["func_mask"])


def test_render_rays_training_selection_and_losses(device):
    """render_rays(is_train=True): same rays as the reference's RandomState(234) first draw, same losses."""
    from nerfdet_amd import rays
    from nerfdet_amd.presets import build_nerfdet
    g = load_golden("rays_select")
    mlp = _mlp(g, device)
    rb = dict(ray_o=g["ray_o"].to(device), ray_d=g["ray_d"].to(device), gt_rgb=g["gt_rgb"].to(device),
              gt_depth=g["gt_depth"].to(device), nerf_sizes=[torch.tensor([[10, 12, 3]])])
    rays.rng = np.random.RandomState(234)
    n_s, n_r = int(g["n_samples"]), int(g["n_rand"])
    orig = rays.sample_along_camera_ray
    rays.sample_along_camera_ray = lambda *a, **k: orig(*a, **{**k, "t_rand": g["t_rand"].to(device)})  # replay the jitter
    try:
        with torch.no_grad():
            ret = rays.render_rays(rb, None, None, g["features_2d"].to(device), g["img"].to(device), None, [0.2, 8.0], n_s, n_r, mlp,
                                   golden_meta(g), None, "image", is_train=True)
    finally:
        rays.sample_along_camera_ray = orig
    assert torch.equal(ret["gt_rgb"].cpu(), g["sel_gt_rgb"]) and torch.equal(ret["gt_depth"].cpu(), g["sel_gt_depth"])
    oc = ret["outputs_coarse"]
    torch.testing.assert_close(oc["rgb"].cpu(), g["rgb"], rtol=1e-5, atol=ATOL)
    torch.testing.assert_close(oc["depth"].cpu(), g["depth"], rtol=1e-5, atol=ATOL)
    assert torch.equal(oc["mask"].cpu(), g["mask"])
    import types
    from nerfdet_amd.detector import nerfdet
    det = types.SimpleNamespace(use_nerf_mask=True)
    torch.testing.assert_close(nerfdet.nvs_loss_func(det, [ret])["loss_nvs"].cpu(), torch.as_tensor(g["loss_nvs"]), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(nerfdet.depth_loss_func(det, [ret])["loss_depth"].cpu(), torch.as_tensor(g["loss_depth"]), rtol=1e-4, atol=1e-6)
    # test mode without render_testing returns None (render_ray.py:518-519)
    assert rays.render_rays(rb, None, None, None, None, None, [0.2, 8.0], n_s, n_r, mlp, golden_meta(g), None, "image", is_train=False) is None


def test_ray_stats_training_size_vs_oracle_and_properties(device):
    """cfg3 shapes: 2048 rays x 64 samples, 40 source views, 32 mapped channels (oracle on a 256-ray slice)."""
    from nerfdet_amd import rays
    gen = torch.Generator().manual_seed(1)
    n_v, d, hw, R, S = 40, 32, (240, 320), 2048, 64
    meta = O.ring_scene_meta(n_v, hw)
    feat = torch.randn(n_v, d, hw[0] // 4, hw[1] // 4, generator=gen)
    img = torch.rand(n_v, 3, *hw, generator=gen)
    ang = torch.rand(R, generator=gen) * 2 * np.pi
    ray_o = torch.stack([2.0 * torch.cos(ang), 2.0 * torch.sin(ang), 1.0 + 0.3 * torch.rand(R, generator=gen)], -1)
    ray_d = -ray_o / ray_o.norm(dim=-1, keepdim=True) + 0.35 * torch.randn(R, 3, generator=gen)
    pts, z = rays.sample_along_camera_ray(ray_o.to(device), ray_d.to(device), [0.2, 8.0], S, det=True)
    cams = rays._compute_projection(meta)
    fd = feat.to(device).contiguous(memory_format=torch.channels_last)
    glob, pm, vc = rays.ray_view_stats(pts, img.to(device), cams, fd)
    assert glob.shape == (R, S, 70) and pm.shape == (R, S)
    sl = slice(0, 256)
    opts = pts[sl].cpu()
    rf, mk = O.projector_compute(opts, img.permute(0, 2, 3, 1).unsqueeze(0), cams, feat)
    mean, var = O.compute_mask_points(rf, mk)
    ref = torch.cat([mean, var], dim=-1).squeeze(2)
    cnt = mk[..., 0].sum(dim=2)
    # a sample whose pixel sits exactly on the image border may flip its in-image test by 1 ulp: exclude |cnt diff|
    same = vc[sl].cpu().long() == cnt.long()
    assert same.float().mean() > 0.999
    torch.testing.assert_close(glob[sl].cpu()[same], ref[same], rtol=0, atol=ATOL)
    assert 0.05 < pm.float().mean() < 0.95
    # properties at full size: means are convex combinations of source values; exp(-var) in (0, 1]; unseen -> mean 0
    g = glob.view(-1, 70)
    assert float(g[:, :3].min()) >= 0 and float(g[:, :3].max()) <= 1 + 1e-6          # RGB means of U[0,1) images
    assert float(g[:, 35:].min()) >= 0 and float(g[:, 35:].max()) <= 1
    unseen = vc.view(-1) == 0
    assert unseen.any() and float(g[unseen][:, :35].abs().max()) == 0
    # permuting the source views leaves count and (up to summation order) the statistics unchanged
    perm = torch.randperm(n_v, generator=gen)
    meta_p = dict(meta)
    meta_p["lidar2img"] = dict(meta["lidar2img"], extrinsic=[meta["lidar2img"]["extrinsic"][i] for i in perm.tolist()])
    glob_p, pm_p, vc_p = rays.ray_view_stats(pts, img[perm].to(device), rays._compute_projection(meta_p),
                                             feat[perm].to(device).contiguous(memory_format=torch.channels_last))
    assert torch.equal(vc_p, vc) and torch.equal(pm_p, pm)
    torch.testing.assert_close(glob_p, glob, rtol=0, atol=ATOL)


def test_render_testing_matches_reference_golden(device):
    """f-4: ``render_rays(render_testing=True)`` (render_ray.py:452-517) against the reference's own output -- every ray of two target
    views in chunks of N_rand = 16 (ragged last chunk, chunks straddling the view boundary), colours, depths, shapes, the
    no-depth variant, and the PSNR of save_rendered_img.py:10-19."""
    from nerfdet_amd import rays
    g = load_golden("render_testing")
    mlp = _mlp(g, device)
    rh, rw = int(g["nerf_size"][0]), int(g["nerf_size"][1])
    rb = dict(ray_o=g["ray_o"].to(device), ray_d=g["ray_d"].to(device), gt_rgb=g["gt_rgb"].to(device), gt_depth=g["gt_depth"].to(device),
              nerf_sizes=[torch.tensor([[rh, rw, 3]])])
    args = (g["features_2d"].to(device), g["img"].to(device), None, [0.2, 8.0], int(g["n_samples"]), int(g["n_rand"]), mlp, golden_meta(g), None, "image")
    with torch.no_grad():
        ret = rays.render_rays(rb, None, None, *args, is_train=False, render_testing=True)
        ret2 = rays.render_rays(dict(rb, gt_depth=[]), None, None, *args, is_train=False, render_testing=True)
    assert ret["outputs_coarse"]["rgb"].shape == (2, rh, rw, 3) and ret["outputs_coarse"]["depth"].shape == (2, rh, rw, 1)
    torch.testing.assert_close(ret["outputs_coarse"]["rgb"].cpu(), g["out_rgb"], rtol=1e-5, atol=ATOL)
    torch.testing.assert_close(ret["outputs_coarse"]["depth"].cpu(), g["out_depth"], rtol=1e-5, atol=ATOL)
    assert torch.equal(ret["gt_rgb"].cpu(), g["out_gt_rgb"]) and torch.equal(ret["gt_depth"].cpu(), g["out_gt_depth"])
    assert ret2["gt_depth"] is None and torch.equal(ret2["outputs_coarse"]["rgb"], ret["outputs_coarse"]["rgb"])
    psnr = torch.stack([rays.compute_psnr(ret["outputs_coarse"]["rgb"][v], ret["gt_rgb"][v]) for v in range(2)])
    torch.testing.assert_close(psnr.cpu(), g["psnr"], rtol=1e-4, atol=1e-4)


def test_render_testing_chunks_equal_one_pass(device):
    """render_rays(render_testing=True) (render_ray.py:452-517): every ray of the target views, chunks of N_rand,
    deterministic sampling -- equals one render_rays_func over all rays, reshaped to (views, H, W, .)."""
    from nerfdet_amd import rays
    g = load_golden("rays_small_s0")
    mlp = _mlp(g, device)
    meta = golden_meta(g)
    t_views, hh, ww = 2, 6, 7
    gen = torch.Generator().manual_seed(4)
    ang = torch.rand(1, t_views, hh * ww, generator=gen) * 2 * np.pi
    ray_o = torch.stack([2.0 * torch.cos(ang), 2.0 * torch.sin(ang), 1.0 + 0 * ang], -1).to(device)
    ray_d = (-ray_o.cpu() / ray_o.cpu().norm(dim=-1, keepdim=True) + 0.3 * torch.randn(1, t_views, hh * ww, 3, generator=gen)).to(device)
    rb = dict(ray_o=ray_o, ray_d=ray_d, gt_rgb=torch.rand(1, t_views, hh * ww, 3, generator=gen).to(device),
              gt_depth=torch.rand(1, t_views, hh, ww, generator=gen).to(device), nerf_sizes=[torch.tensor([[hh, ww, 3]])])
    f2d, img, s = g["features_2d"].to(device), g["img"].to(device), int(g["n_samples"])
    with torch.no_grad():
        ret = rays.render_rays(rb, None, None, f2d, img, None, [0.2, 8.0], s, 16, mlp, meta, None, "image", is_train=False,
                               render_testing=True)
        one = rays.render_rays_func(ray_o.view(-1, 3), ray_d.view(-1, 3), None, None, f2d, img, None, [0.2, 8.0], s, 16, mlp, meta,
                                    None, "image", det=True)
    assert ret["outputs_coarse"]["rgb"].shape == (t_views, hh, ww, 3) and ret["outputs_coarse"]["depth"].shape == (t_views, hh, ww, 1)
    torch.testing.assert_close(ret["outputs_coarse"]["rgb"].view(-1, 3), one["outputs_coarse"]["rgb"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ret["outputs_coarse"]["depth"].view(-1), one["outputs_coarse"]["depth"], rtol=1e-5, atol=1e-6)
    assert ret["gt_rgb"].shape == (t_views, hh, ww, 3) and ret["gt_depth"].shape == (t_views, hh, ww, 1)


def test_ray_branch_mlp_on_matrix_cores_matches_library_forward(device):
    """A10 at inference: VanillaNeRFRadianceField.forward takes the hand-written MFMA path (1x1 convolutions over the sample
    rows) and matches the library Linear layers (nerf_mlp.py:146-161,229-234)."""
    from nerfdet_amd.radiance_field import VanillaNeRFRadianceField
    torch.manual_seed(2)
    mlp = VanillaNeRFRadianceField(net_depth=4, net_width=256, skip_layer=3, feature_dim=70, net_depth_condition=1,
                                   net_width_condition=128).to(device).eval()
    pts = torch.randn(96, 40, 3, device=device) * 2.0
    dirs = torch.randn(96, 3, device=device)
    feat = torch.randn(96, 40, 70, device=device)
    with torch.no_grad():
        rgb1, sig1 = mlp(pts, dirs, feat)                      # no grad, GPU -> forward_rows_hip
        rgb1b, sig1b = mlp.forward_rows_hip(pts, dirs, feat)
    with torch.enable_grad():
        rgb0, sig0 = mlp(pts, dirs, feat)                      # library path
    assert torch.equal(rgb1, rgb1b) and torch.equal(sig1, sig1b)
    assert rgb1.shape == rgb0.shape and sig1.shape == sig0.shape
    torch.testing.assert_close(rgb1, rgb0.detach(), rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(sig1, sig0.detach(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("n_v,d,hw", [(40, 32, (240, 320)), (9, 8, (60, 80)), (100, 32, (120, 160)), (5, 48, (48, 64))])
def test_packed_sampler_equals_generic_kernel(device, n_v, d, hw):
    """csrc/ray_stats_kernels.hip (one projection per pair, near-view walk, shifted one-pass variance, image as NHWC4) against the
    generic kernel (two-pass, every view) on the same inputs, forward and backward: masks and counts bit-exact, statistics to
    2e-5, gradients to 1e-4 of their scale.  Covers 1 and 2 view rounds, 2 / 7 / 21 samples per wave, ragged tails."""
    from ctypes import c_void_p
    from nerfdet_amd import _lib, rays
    from nerfdet_amd._lib import check
    gen = torch.Generator().manual_seed(n_v * 100 + d)
    R, S = 257, 19                                   # 4883 samples: not a multiple of any samples-per-block
    meta = O.ring_scene_meta(n_v, hw)
    feat = torch.randn(n_v, d, hw[0] // 4, hw[1] // 4, generator=gen).to(device).contiguous(memory_format=torch.channels_last)
    img = torch.rand(n_v, 3, *hw, generator=gen).to(device)
    ang = torch.rand(R, generator=gen) * 2 * np.pi
    ray_o = torch.stack([2.0 * torch.cos(ang), 2.0 * torch.sin(ang), 1.0 + 0.3 * torch.rand(R, generator=gen)], -1)
    ray_d = -ray_o / ray_o.norm(dim=-1, keepdim=True) + 0.35 * torch.randn(R, 3, generator=gen)
    pts, _ = rays.sample_along_camera_ray(ray_o.to(device), ray_d.to(device), [0.2, 8.0], S, det=True)
    cams = rays._compute_projection(meta)
    assert rays.packed_ok(n_v, d)
    glob, pm, vc = rays.ray_view_stats(pts, img, cams, feat)
    saved = rays.packed_ok
    rays.packed_ok = lambda *a, **k: False           # force the generic kernels
    try:
        glob_g, pm_g, vc_g = rays.ray_view_stats(pts, img, cams, feat)
    finally:
        rays.packed_ok = saved
    assert torch.equal(pm, pm_g) and torch.equal(vc, vc_g)
    assert 0.02 < pm.float().mean() < 0.98
    torch.testing.assert_close(glob, glob_g, rtol=0, atol=ATOL)
    # backward through both
    if not rays.packed_ok(n_v, d, backward=True):
        assert d == 8          # 32 samples per wave x 64 view slots: the projection records alone fill the 64 KB of LDS
        return
    lib = _lib.load()
    ke, h, w = rays._camera_matrices(cams.squeeze(0))
    ke = ke.to(device)
    g = torch.randn(R * S, 2 * (3 + d), generator=gen).to(device)
    p = pts.reshape(-1, 3).contiguous()
    st = c_void_p(torch.cuda.current_stream(device).cuda_stream)
    outs = []
    for fn in (lib.ndet_ray_view_stats_packed_bwd, lib.ndet_ray_view_stats_bwd):
        df = torch.zeros(n_v, hw[0] // 4, hw[1] // 4, d, device=device)
        check(fn(c_void_p(g.data_ptr()), c_void_p(p.data_ptr()), R * S, c_void_p(ke.data_ptr()), n_v, h, w, c_void_p(feat.data_ptr()), d,
                 hw[0] // 4, hw[1] // 4, feat.stride(0), feat.stride(2), c_void_p(df.data_ptr()), st), "bwd")
        outs.append(df)
    scale = float(outs[1].abs().max())
    assert scale > 0
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-4 * scale


def test_rendering_metrics_match_the_cpu_restatement(device):
    """f-4: PSNR / SSIM / depth-error map of ``simple_test(evaluate_nerf=True)`` (nerfdet.py:342-343, save_rendered_img.py:38-78) on the
    device against oracle/render_eval_oracle.py (scikit-image's SSIM restated: unpinned, see its header)."""
    import numpy as np
    from nerfdet_amd import rays
    from oracle import render_eval_oracle as R
    torch.manual_seed(0)
    gt = torch.rand(3, 20, 27, 3)
    rgb = (gt + 0.05 * torch.randn_like(gt)).clamp(0, 1)
    gt_depth = torch.rand(3, 20, 27, 1) * 4
    depth = gt_depth + 0.1 * torch.randn_like(gt_depth)
    ret = dict(outputs_coarse=dict(rgb=rgb.to(device), depth=depth.to(device)), gt_rgb=gt.to(device), gt_depth=gt_depth.to(device))
    psnr, ssim, err = rays.rendering_metrics(ret)
    p, s, e = R.rendering_metrics(rgb.numpy(), gt.numpy(), depth.numpy(), gt_depth.numpy())
    assert abs(float(psnr) - p) <= 1e-4 and abs(float(ssim) - s) <= 1e-9
    assert np.allclose(err.cpu().numpy(), e, rtol=1e-5, atol=1e-7) and err.shape == (20, 27, 1)
    assert rays.rendering_metrics(dict(ret, gt_depth=None))[2] is None
    assert abs(float(rays.compute_ssim(gt[0].to(device), gt[0].to(device))) - 1.0) < 1e-12
