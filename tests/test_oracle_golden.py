"""Pin the CPU oracle against the golden vectors produced by the real reference code
(tests/golden/make_golden.py) and against the reference's own known-answer NMS test."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden, sub_state
from oracle import nerfdet_oracle as O

VOLUME_FIXTURES = ["volume_small_s0", "volume_small_s1", "volume_medium_s2"]
RAY_FIXTURES = ["rays_small_s0", "rays_small_s1"]


@pytest.mark.parametrize("name", VOLUME_FIXTURES)
def test_projection_points_backproject(name):
    g = load_golden(name)
    meta = golden_meta(g)
    proj = O.compute_projection(meta, 4)
    rgb_proj = O.compute_projection(meta, 1)
    assert torch.equal(proj, g["projection"])
    assert torch.equal(rgb_proj, g["rgb_projection"])
    pts = O.get_points(g["n_voxels"].tolist(), g["voxel_size"].tolist(), meta["lidar2img"]["origin"])
    assert torch.equal(pts, g["points"])
    h, w = meta["img_shape"][0] // 4, meta["img_shape"][1] // 4
    vol, valid = O.backproject(g["features"][:, :, :h, :w], pts, proj)
    assert torch.equal(valid, g["bp_valid"])
    assert torch.equal(vol[0], g["bp_volume_v0"]) and torch.equal(vol[-1], g["bp_volume_vlast"])
    assert torch.equal(vol.sum(0), g["bp_volume_sum"])
    rvol, rvalid = O.backproject(g["denorm_images"][:, :, :meta["img_shape"][0], :meta["img_shape"][1]], pts, rgb_proj)
    assert torch.equal(rvalid, g["rgb_bp_valid"])
    assert torch.equal(rvol.sum(0), g["rgb_bp_volume_sum"])
    frac = valid.float().mean().item()
    assert 0.05 < frac < 0.95, f"degenerate fixture, valid fraction {frac}"


@pytest.mark.parametrize("name", VOLUME_FIXTURES)
def test_extract_volume_matches_reference_extract_feat(name):
    g = load_golden(name)
    meta = golden_meta(g)
    out = O.extract_volume(g["features"], g["denorm_images"], meta, g["n_voxels"].tolist(), g["voxel_size"].tolist(),
                           g["mapping.0.weight"], g["mapping.0.bias"], sub_state(g, "nerf_mlp."))
    assert torch.equal(out["valid"], g["out_valid"])
    # same ops in the same order on the same CPU -> expect (near) bit equality
    torch.testing.assert_close(out["volume"], g["out_volume"], rtol=0, atol=1e-6)
    assert (g["out_volume"] != 0).float().mean() > 0.2


@pytest.mark.parametrize("name", RAY_FIXTURES)
def test_ray_branch_pieces(name):
    g = load_golden(name)
    meta = golden_meta(g)
    sd = sub_state(g, "nerf_mlp.")
    s = int(g["n_samples"])
    pts_det, z_det = O.sample_along_camera_ray(g["ray_o"], g["ray_d"], [0.2, 8.0], s, det=True)
    assert torch.equal(pts_det, g["pts_det"]) and torch.equal(z_det, g["z_det"])
    pts, z = O.sample_along_camera_ray(g["ray_o"], g["ray_d"], [0.2, 8.0], s, det=False, t_rand=g["t_rand"])
    assert torch.equal(pts, g["pts_rnd"]) and torch.equal(z, g["z_rnd"])
    cams = O.compute_ray_cameras(meta)
    assert torch.equal(cams, g["cameras"])
    imgs = g["img"].permute(0, 2, 3, 1).unsqueeze(0)
    rgb_feat, mask = O.projector_compute(pts, imgs, cams, g["features_2d"])
    assert torch.equal(mask, g["mask"])
    torch.testing.assert_close(rgb_feat, g["rgb_feat"], rtol=0, atol=1e-6)
    assert 0.05 < mask.mean() < 0.95
    mean, var = O.compute_mask_points(rgb_feat, mask)
    torch.testing.assert_close(mean, g["mean"], rtol=0, atol=1e-6)
    torch.testing.assert_close(var, g["var"], rtol=0, atol=1e-6)
    glob = torch.cat([mean, var], dim=-1).squeeze(2)
    rgb_pts, sigma_pts = O.nerf_forward(sd, pts, g["ray_d"], glob)
    torch.testing.assert_close(rgb_pts, g["rgb_pts"], rtol=0, atol=1e-6)
    torch.testing.assert_close(sigma_pts, g["sigma_pts"], rtol=1e-6, atol=1e-6)
    dens = O.nerf_query_density(sd, pts.reshape(-1, 3), glob.reshape(-1, glob.shape[-1]))
    torch.testing.assert_close(dens, g["density_q"], rtol=1e-6, atol=1e-6)
    comp = O.raw2outputs(torch.cat([rgb_pts, sigma_pts], -1), z, mask[..., 0].sum(dim=2) > 1)
    for k, gk in [("rgb", "comp_rgb"), ("depth", "comp_depth"), ("weights", "comp_weights"),
                  ("alpha", "comp_alpha"), ("transparency", "comp_T")]:
        torch.testing.assert_close(comp[k], g[gk], rtol=1e-6, atol=1e-6)
    assert torch.equal(comp["mask"], g["comp_mask"])
    comp2 = O.raw2outputs(g["raw_rand"], z, mask[..., 0].sum(dim=2) > 1, white_bkgd=True)
    torch.testing.assert_close(comp2["rgb"], g["comp2_rgb"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(comp2["depth"], g["comp2_depth"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", RAY_FIXTURES)
def test_render_rays_func(name):
    g = load_golden(name)
    ret = O.render_rays_func(g["ray_o"], g["ray_d"], g["features_2d"], g["img"], [0.2, 8.0], int(g["n_samples"]),
                             sub_state(g, "nerf_mlp."), golden_meta(g), det=True)
    oc = ret["outputs_coarse"]
    torch.testing.assert_close(oc["rgb"], g["func_rgb"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(oc["depth"], g["func_depth"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(oc["weights"], g["func_weights"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ret["sigma"], g["func_sigma"], rtol=1e-6, atol=1e-6)
    assert torch.equal(oc["mask"], g["func_mask"])


def test_training_ray_selection_and_losses():
    g = load_golden("rays_select")
    rb = dict(ray_o=g["ray_o"], ray_d=g["ray_d"], gt_rgb=g["gt_rgb"], gt_depth=g["gt_depth"])
    rng = np.random.RandomState(234)  # render_ray.py:20
    ray_o, ray_d, gt_rgb, gt_depth = O.select_training_rays(rb, int(g["n_rand"]), rng)
    assert torch.equal(gt_rgb, g["sel_gt_rgb"]) and torch.equal(gt_depth, g["sel_gt_depth"])
    ret = O.render_rays_func(ray_o, ray_d, g["features_2d"], g["img"], [0.2, 8.0], int(g["n_samples"]),
                             sub_state(g, "nerf_mlp."), golden_meta(g), det=False, t_rand=g["t_rand"])
    oc = ret["outputs_coarse"]
    torch.testing.assert_close(oc["rgb"], g["rgb"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(oc["depth"], g["depth"], rtol=1e-6, atol=1e-6)
    assert torch.equal(oc["mask"], g["mask"])
    torch.testing.assert_close(O.nvs_loss(oc["rgb"], gt_rgb, oc["mask"]), torch.as_tensor(g["loss_nvs"]), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(O.depth_loss(oc["depth"], gt_depth, oc["mask"]), torch.as_tensor(g["loss_depth"]), rtol=1e-6, atol=1e-7)


def test_neck_head_nms_small():
    g = load_golden("head_small_s0")
    nsd = sub_state(g, "neck_3d.")
    outs = O.neck3d_forward(nsd, g["x"], training=False)
    outs_t = O.neck3d_forward(nsd, g["x"], training=True)
    for i in range(3):
        torch.testing.assert_close(outs[i], g[f"neck_eval_{i}"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(outs_t[i], g[f"neck_train_{i}"], rtol=1e-5, atol=1e-5)
    hsd = sub_state(g, "bbox_head.")
    ctr, reg, cls = O.head_forward(hsd, [g[f"neck_eval_{i}"] for i in range(3)])
    for i in range(3):
        torch.testing.assert_close(ctr[i], g[f"ctr_{i}"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(reg[i], g[f"reg_{i}"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(cls[i], g[f"cls_{i}"], rtol=1e-6, atol=1e-6)
    res = O.head_get_bboxes([g[f"ctr_{i}"] for i in range(3)], [g[f"reg_{i}"] for i in range(3)],
                            [g[f"cls_{i}"] for i in range(3)], g["valid"], g["origin"].numpy(), g["voxel_size"].tolist(),
                            int(g["nms_pre"]), float(g["score_thr"]), float(g["iou_thr"]))
    assert len(g["det_scores"]) > 10, "fixture should exercise NMS"
    assert torch.equal(res["labels"], g["det_labels"])
    assert torch.equal(res["scores"], g["det_scores"])
    torch.testing.assert_close(res["boxes"], g["det_boxes"], rtol=0, atol=0)


def test_nms_reference_known_answer():
    """The reference's own pinned vector: tests/test_nms.py:5-58 (values are data, re-typed here)."""
    boxes = torch.tensor([
        [1.2261, 0.6679, -1.2678, 2.6547, 1.0428, 0.1000], [5.0919, 0.6512, 0.7238, 5.4821, 1.2451, 2.1095],
        [6.8392, -1.2205, 0.8570, 7.6920, 0.3220, 3.2223], [3.6900, -0.4235, -1.0380, 4.4415, 0.2671, -0.1442],
        [4.8071, -1.4311, 0.7004, 5.5788, -0.6837, 1.2487], [2.1807, -1.5811, -1.1289, 3.0151, -0.1346, -0.5351],
        [4.4631, -4.2588, -1.1403, 5.3012, -3.4463, -0.3212], [4.7607, -3.3311, 0.5993, 5.2976, -2.7874, 1.2273],
        [3.1265, 0.7113, -0.0296, 3.8944, 1.3532, 0.9785], [5.5828, -3.5350, 1.0105, 8.2841, -0.0405, 3.3614],
        [3.0003, -2.1099, -1.0608, 5.3423, 0.0328, 0.6252], [2.7148, 0.6082, -1.1738, 3.6995, 1.2375, -0.0209],
        [4.9263, -0.2152, 0.2889, 5.6963, 0.3416, 1.3471], [5.0713, 1.3459, -0.2598, 5.6278, 1.9300, 1.2835],
        [4.5985, -2.3996, -0.3393, 5.2705, -1.7306, 0.5698], [4.1386, 0.5658, 0.0422, 4.8937, 1.1983, 0.9911],
        [2.7694, -1.9822, -1.0637, 4.0691, 0.3575, -0.1393], [4.6464, -3.0123, -1.0694, 5.1421, -2.4450, -0.3758],
        [3.4754, 0.4443, -1.1282, 4.6727, 1.3786, 0.2550], [2.5905, -0.3504, -1.1202, 3.1599, 0.1153, -0.3036],
        [4.1336, -3.4813, 1.1477, 6.2091, -0.8776, 2.6757], [3.9966, 0.2069, -1.1148, 5.0841, 1.0525, -0.0648],
        [4.3216, -1.8647, 0.4733, 6.2069, 0.6671, 3.3363], [4.7683, 0.4286, -0.0500, 5.5642, 1.2906, 0.8902],
        [1.7337, 0.7625, -1.0058, 3.0675, 1.3617, 0.3849], [4.7193, -3.3687, -0.9635, 5.1633, -2.7656, 1.1001],
        [4.4704, -2.7744, -1.1127, 5.0971, -2.0228, -0.3150], [2.7027, 0.6122, -0.9169, 3.3083, 1.2117, 0.6129],
        [4.8789, -2.0025, 0.8385, 5.5214, -1.3668, 1.3552], [3.7856, -1.7582, -0.1738, 5.3373, -0.6300, 0.5558]])
    scores = torch.tensor([
        3.6414e-03, 2.2901e-02, 2.7576e-04, 1.2238e-02, 5.9310e-04, 1.2659e-01, 2.4104e-02, 5.0742e-03, 2.3581e-03,
        2.0946e-07, 8.8039e-01, 1.9127e-01, 5.0469e-05, 9.3638e-03, 3.0663e-03, 9.4350e-03, 5.3380e-02, 1.7895e-01,
        2.0048e-01, 1.1294e-03, 3.0304e-08, 2.0237e-01, 1.0894e-08, 6.7972e-02, 6.7156e-01, 9.3986e-04, 7.9470e-01,
        3.9736e-01, 1.8000e-04, 7.9151e-04])
    cls = torch.tensor([8, 8, 8, 3, 3, 1, 3, 3, 7, 8, 0, 6, 7, 8, 3, 7, 2, 7, 6, 3, 8, 6, 6, 7, 6, 8, 7, 6, 3, 1])
    expected = torch.tensor([10, 26, 24, 27, 21, 18, 17, 5, 23, 16, 6, 1, 3, 15, 13, 7, 0, 14, 8, 19, 25, 29, 4, 2,
                             28, 12, 9, 20, 22])
    assert torch.equal(O.aligned_3d_nms(boxes, scores, cls, 0.25), expected)


def test_nms_random_golden():
    g = load_golden("nms_random")
    for thr in (0.25, 0.5):
        pick = O.aligned_3d_nms(g["boxes"], g["scores"], g["classes"], thr)
        assert torch.equal(pick, g[f"pick_{int(thr * 100)}"])
        assert 50 < len(pick) < 400
    pick = O.aligned_3d_nms(g["deg_boxes"], g["scores"][:40], torch.zeros(40, dtype=torch.long), 0.25)
    assert torch.equal(pick, g["deg_pick"])


def test_render_eval_oracle_ssim_is_the_windowed_definition():
    """oracle/render_eval_oracle.py: the filter form of SSIM equals the window-by-window definition with sample statistics (scikit-image
    is absent: parity unpinned, see the module header), identical images score 1, and PSNR follows save_rendered_img.py:10-19."""
    from oracle import render_eval_oracle as R
    rng = np.random.RandomState(0)
    x = rng.rand(12, 15).astype(np.float32)
    y = np.clip(x + 0.1 * rng.randn(12, 15), 0, 1).astype(np.float32)
    assert abs(R.ssim_channel(x, y) - R.ssim_by_definition(x, y)) < 1e-12
    assert abs(R.ssim_channel(x, x) - 1.0) < 1e-12 and R.ssim_channel(x, y) < 0.99
    img = rng.rand(9, 11, 3).astype(np.float32)
    assert abs(R.ssim(img, img) - 1.0) < 1e-12
    assert abs(R.psnr(np.zeros((4, 4, 3), np.float32), np.full((4, 4, 3), 0.1, np.float32)) - 20.0) < 1e-4
