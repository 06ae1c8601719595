"""Training-path insurance on the GPU (nerfdet.py:271-321, imvoxel_head_v2.py:116-203): a fixed scene is fitted for 30 optimizer steps and
every one of the five losses must fall; bf16 and fp32-class gradients of one cfg3-shaped step must point the same way, parameter group
by parameter group.  (After MIOpen's channels-last-3d backward turned out 43 % wrong -- DESIGN.md 9.2 -- aggregate checks like these are
cheap insurance on top of the per-kernel gradient tests of tests/test_backward_gpu.py and tests/test_conv_train_gpu.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOSSES = ("loss_centerness", "loss_bbox", "loss_cls", "loss_nvs", "loss_depth")
GROUPS = ("backbone.layer2", "backbone.layer3", "backbone.layer4", "neck.lateral_convs", "neck.fpn_convs.0", "mapping", "nerf_mlp",
          "neck_3d.down_layer_0", "neck_3d.down_layer_1", "neck_3d.down_layer_2", "neck_3d.up_block", "neck_3d.out_block", "bbox_head")


@pytest.mark.timeout(900)
def test_thirty_steps_on_a_fixed_scene_lower_all_five_losses(device):
    """Thirty optimizer steps (config:167-173: AdamW 2e-4, backbone x0.1, clip 35) on one scene, every step with its own ray draw and
    sampling noise as in training; the five losses are read on a FIXED probe -- the same 512 rays, the same noise -- before and after."""
    from test_ddp import _build, _scene
    import nerfdet_amd.rays as R
    from nerfdet_amd.train import build_optimizer, train_one_step
    det = _build(device)
    det.N_rand = 512
    opt = build_optimizer(det)
    scene = _scene(0, device)

    def probe():
        R.rng = np.random.RandomState(99)
        torch.manual_seed(7)
        with torch.no_grad():
            return {k: float(v) for k, v in det.train_step(scene)["log_vars"].items()}
    first = probe()
    R.rng = np.random.RandomState(234)
    torch.manual_seed(3)
    hist = [train_one_step(det, scene, opt)["log_vars"] for _ in range(30)]
    assert all(np.isfinite(h["loss"]) for h in hist)
    last = probe()
    print("fixed scene, probe before -> after 30 steps:", {k: (round(first[k], 4), round(last[k], 4)) for k in first})
    for k in LOSSES + ("loss",):
        assert last[k] < first[k], f"{k} did not fall: {first[k]:.4f} -> {last[k]:.4f}"
    assert last["loss"] < 0.9 * first["loss"]


@pytest.mark.timeout(900)
def test_bf16_gradients_point_along_the_fp32_class_gradients_at_cfg3_shapes(device):
    """BASELINE configs[2] trains in bf16: one step at its per-rank shapes (40 source + 10 target views 240x320, 40x40x16 voxels, 2 048 rays x 64
    samples) in three arithmetics, same rays and sampling noise.

    * exact fp32-MFMA kernels vs the fp32-class bf16x3 kernels: the two independent kernel families must agree per parameter group to
      cosine >= 0.9999 and 1 % in norm -- this is the check that no backward path is wrong (it would have caught the library's
      channels-last-3d backward);
    * bf16 vs fp32-class: every single layer is within 2.4e-3 of fp64 in output, data and weight gradient (operand rounding, measured
      layer by layer), but a bf16 forward flips the ReLU mask of the ~0.5 % of units whose pre-activation lies within its error, and
      every flipped unit passes or blocks its whole gradient: ~7 % relative gradient noise per ReLU layer, 12 - 26 % after the 3D neck
      and the bottlenecks (measured: cosine 0.965 - 0.9999 per group).  The bar is what that mechanism explains: cosine >= 0.95, norms
      within 12 %; the five losses agree to 2e-2 (tests/test_fullsize_gpu.py)."""
    from nerfdet_amd import conv3d, rays
    from nerfdet_amd.presets import build_nerfdet
    from nerfdet_amd.synth import batch_to, train_scene
    torch.manual_seed(0)
    det = build_nerfdet(50, depth_supervise=True)
    with torch.no_grad():
        det.neck.fpn_convs[0].conv.weight.mul_(1 / 30.0)
        det.nerf_mlp.mlp.sigma_layer.output_layer.bias.fill_(1.0)
    det.to(device).train()
    scene = batch_to(train_scene(40, (240, 320), t_views=10, n_boxes=8, seed=4), device)
    grads = {}
    for mode in ("bf16x3", "f32", "bf16"):
        prev = conv3d.set_arithmetic(mode)
        try:
            rays.rng = np.random.RandomState(234)
            torch.manual_seed(5)
            det.zero_grad(set_to_none=True)
            det.train_step(scene)["loss"].backward()
            named = dict(det.named_parameters())
            grads[mode] = {g: torch.cat([p.grad.detach().float().reshape(-1) for n, p in named.items() if n.startswith(g) and p.grad is not None])
                           for g in GROUPS}
        finally:
            conv3d.set_arithmetic(prev)
    report = {}
    for mode in ("f32", "bf16"):
        for g in GROUPS:
            a, b = grads["bf16x3"][g].double(), grads[mode][g].double()
            assert a.numel() > 0 and float(a.norm()) > 0, g
            report[(mode, g)] = (float(torch.dot(a, b) / (a.norm() * b.norm())), float(b.norm() / a.norm()))
    for mode in ("f32", "bf16"):
        print(f"{mode} vs fp32-class gradients (cosine, norm ratio):", {g: (round(report[(mode, g)][0], 5), round(report[(mode, g)][1], 4)) for g in GROUPS})
    for g in GROUPS:
        c, r = report[("f32", g)]
        assert c >= 0.9999 and abs(r - 1) <= 0.01, f"{g}: the two fp32-class kernel families disagree: cosine {c:.6f}, norm ratio {r:.4f}"
        c, r = report[("bf16", g)]
        assert c >= 0.95, f"{g}: bf16 cosine {c:.5f}"
        assert abs(r - 1) <= 0.12, f"{g}: bf16 norm ratio {r:.4f}"
