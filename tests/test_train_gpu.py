"""Training-path insurance on the GPU (nerfdet.py:271-321, imvoxel_head_v2.py:116-203): five optimizer steps must follow the trajectory the
oracle-backed CPU module takes (tests/golden/train_traj.npz), with the gradient scatter in its deterministic mode (bitwise reproducible, checked);
a fixed scene is fitted for 30 optimizer steps and the total loss must fall (deterministic mode; the float-atomic path keeps the looser
per-loss bounds of round 3); bf16 and fp32-class gradients of one cfg3-shaped step must point the same way, parameter group by parameter group.  (After MIOpen's channels-last-3d backward turned out 43 % wrong -- DESIGN.md 9.2 -- aggregate checks like these are
cheap insurance on top of the per-kernel gradient tests of tests/test_backward_gpu.py and tests/test_conv_train_gpu.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOSSES = ("loss_centerness", "loss_bbox", "loss_cls", "loss_nvs", "loss_depth")
GROUPS = ("backbone.layer2", "backbone.layer3", "backbone.layer4", "neck.lateral_convs", "neck.fpn_convs.0", "mapping", "nerf_mlp",
          "neck_3d.down_layer_0", "neck_3d.down_layer_1", "neck_3d.down_layer_2", "neck_3d.up_block", "neck_3d.out_block", "bbox_head")


def _learnable_scene(device):
    """tests/test_ddp.py's scene with NeRF targets that can be fitted: one colour and one depth for every target ray (the generator's
    uniform-random colours and depths leave the two NeRF losses at their noise floor)."""
    from test_ddp import _scene
    scene = _scene(0, device)
    scene["gt_depths"] = torch.full_like(scene["gt_depths"], 2.0)
    scene["gt_images"] = torch.ones_like(scene["gt_images"]) * torch.tensor([0.3, 0.5, 0.7], device=device)
    return scene


@pytest.mark.timeout(900)
def test_each_of_the_five_losses_descends_along_its_own_gradient(device):
    """First-order check of the whole backward, one loss at a time: with g = d loss_k / d theta from ``backward()``, the step
    theta - eps g / |g| must lower loss_k, and by eps |g| once eps is small enough for the curvature not to matter, on a fixed probe (same
    2 048 rays, same sampling noise: the forward has no atomics and is deterministic).  A wrong term anywhere on a loss's path -- scatter
    kernels, convolution data / weight gradients, BatchNorm, the fused epilogue's backward -- leaves the observed / predicted ratio away
    from 1 at every step size.  Steps predicted to lower the loss by 1e-2, 1e-3, 1e-4 and 1e-5 of its value are tried (fp32 resolves
    the last one to ~1 %): four of the losses meet the prediction within 2 - 12 % at the first; the L1 depth loss is sharply curved
    along the feature directions (0.20, 0.25, 0.50, 0.92 of the prediction as the step shrinks; 1.00 along the MLP's own parameters)."""
    from test_ddp import _build
    import nerfdet_amd.rays as R
    det = _build(device)
    det.N_rand = 2048
    scene = _learnable_scene(device)
    params = [p for p in det.parameters() if p.requires_grad]

    def losses():
        R.rng = np.random.RandomState(99)
        torch.manual_seed(7)
        return det(return_loss=True, **scene)

    def value(out, k):
        return out[k] if isinstance(out[k], torch.Tensor) else sum(out[k])
    report = {}
    for k in LOSSES:
        det.zero_grad(set_to_none=True)
        lk = value(losses(), k)
        l0 = float(lk.detach())
        lk.backward()
        grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
        gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads if g is not None)))
        assert np.isfinite(gnorm) and gnorm > 0, k
        ratios = []
        for frac in (1e-2, 1e-3, 1e-4, 1e-5):
            eps = frac * max(l0, 1e-3) / gnorm                 # predicted decrease: frac of the loss
            with torch.no_grad():
                for p, g in zip(params, grads):
                    if g is not None:
                        p.sub_(g * (eps / gnorm))
                l1 = float(value(losses(), k))
                for p, g in zip(params, grads):
                    if g is not None:
                        p.add_(g * (eps / gnorm))
            ratios.append((l0 - l1) / (eps * gnorm))
        report[k] = (l0, ratios)
    print("descent along the own gradient, observed / predicted decrease at 1e-2 .. 1e-5 of the loss:", {k: (round(a, 5), [round(r, 3) for r in rs]) for k, (a, rs) in report.items()})
    for k, (l0, ratios) in report.items():
        assert all(r > 0 for r in ratios[:3]), f"{k} does not fall along its own negative gradient: {ratios}"
        assert any(0.8 <= r <= 1.2 for r in ratios), f"{k}: the first-order prediction is met at no step size: {ratios}"


@pytest.mark.timeout(900)
@pytest.mark.parametrize("deterministic", [True, False])
def test_thirty_optimizer_steps_on_a_fixed_scene_lower_the_loss(device, deterministic):
    """Thirty steps of the reference's optimizer set-up (config:167-173: AdamW 2e-4, backbone x0.1, clip 35) on one scene, every step with
    its own ray draw and sampling noise; read on a fixed probe before and after.  The three losses that move by tens of per cent in 30 steps,
    and the sum of everything but the classification term, must fall.  (Adam's sign-like first steps make the trajectory chaotic under the float atomics' run-to-run noise:
    the classification loss alone ended between 0.24 and 1.35 from 1.11 in repeated runs, the depth loss within +-1 % of its start -- the
    per-loss statement is the first-order test above.)"""
    from test_ddp import _build
    import nerfdet_amd.rays as R
    from nerfdet_amd.train import build_optimizer, train_one_step
    from nerfdet_amd import autograd as A
    det = _build(device)
    det.N_rand = 512
    opt = build_optimizer(det)
    scene = _learnable_scene(device)
    prev_det = A.set_deterministic(deterministic)

    def probe():
        det.N_rand = 2048
        R.rng = np.random.RandomState(99)
        torch.manual_seed(7)
        with torch.no_grad():
            out = {k: float(v) for k, v in det.train_step(scene)["log_vars"].items()}
        det.N_rand = 512
        return out
    first = probe()
    R.rng = np.random.RandomState(234)
    torch.manual_seed(3)
    try:
        hist = [train_one_step(det, scene, opt)["log_vars"] for _ in range(30)]
    finally:
        A.set_deterministic(prev_det)
    assert all(np.isfinite(h["loss"]) for h in hist)
    last = probe()
    print("fixed scene, probe before -> after 30 steps:", {k: (round(first[k], 4), round(last[k], 4)) for k in first})
    # measured over 16 fresh processes (gpurun_out of round 3): centerness 0.749 -> 0.50 - 0.55, bbox 0.804 -> 0.58 - 0.67, nvs 0.115 -> 0.090 - 0.095 every
    # time; classification 1.11 -> 0.57 - 1.80 and with it the total (3 of 16 runs above its start), depth 1.063 -> 0.97 - 1.12: chaotic, not asserted
    for k in ("loss_centerness", "loss_bbox", "loss_nvs"):
        assert last[k] < 0.92 * first[k], f"{k} did not fall: {first[k]:.4f} -> {last[k]:.4f}"
    rest = lambda d: d["loss"] - d["loss_cls"]
    assert rest(last) < rest(first), f"the losses other than the classification term did not fall: {rest(first):.4f} -> {rest(last):.4f}"
    assert last["loss_depth"] < 1.15 * first["loss_depth"] and last["loss_cls"] < 3.0 * first["loss_cls"]
    if deterministic:      # the run is reproducible bit for bit: the statement round 3 had to drop for the float-atomic path holds here, every time
        assert last["loss"] < first["loss"], f"the total loss did not fall over 30 steps: {first['loss']:.4f} -> {last['loss']:.4f}"


@pytest.mark.timeout(900)
def test_bf16_gradients_point_along_the_fp32_class_gradients_at_cfg3_shapes(device):
    """BASELINE configs[2] trains in bf16: one step at its per-rank shapes (40 source + 10 target views 240x320, 40x40x16 voxels, 2 048 rays x 64
    samples) in three arithmetics, same rays and sampling noise.

    * exact fp32-MFMA kernels and the default fp16-pair training arithmetic vs the fp32-class bf16x3 kernels: the independent kernel families
      must agree per parameter group to cosine >= 0.9999 and 1 % in norm -- this is the check that no backward path is wrong (it would have caught the library's
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
    for mode in ("bf16x3", "f32", "bf16", "f16x2"):
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
    for mode in ("f32", "bf16", "f16x2"):
        for g in GROUPS:
            a, b = grads["bf16x3"][g].double(), grads[mode][g].double()
            assert a.numel() > 0 and float(a.norm()) > 0, g
            report[(mode, g)] = (float(torch.dot(a, b) / (a.norm() * b.norm())), float(b.norm() / a.norm()))
    for mode in ("f32", "bf16", "f16x2"):
        print(f"{mode} vs fp32-class gradients (cosine, norm ratio):", {g: (round(report[(mode, g)][0], 5), round(report[(mode, g)][1], 4)) for g in GROUPS})
    for g in GROUPS:
        c, r = report[("f32", g)]
        assert c >= 0.9999 and abs(r - 1) <= 0.01, f"{g}: the two fp32-class kernel families disagree: cosine {c:.6f}, norm ratio {r:.4f}"
        # the DEFAULT training arithmetic (fp16 pairs with device-side scales, conv3d.TRAIN_F16X2) is held to the same bar as the exact kernels
        c, r = report[("f16x2", g)]
        assert c >= 0.9999 and abs(r - 1) <= 0.01, f"{g}: the fp16-pair training arithmetic leaves the fp32-class gradients: cosine {c:.6f}, norm ratio {r:.4f}"
        c, r = report[("bf16", g)]
        assert c >= 0.95, f"{g}: bf16 cosine {c:.5f}"
        assert abs(r - 1) <= 0.12, f"{g}: bf16 norm ratio {r:.4f}"


def _five_steps(device, deterministic):
    """The protocol of tests/golden/make_golden_traj.py on the GPU: 5 x train_one_step, the ray draw of step k from RandomState(1000 + k),
    deterministic sampling along the rays (the oracle stand-in of the CPU run samples that way)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden_traj import KEYS, STEPS, learnable_scene
    from test_ddp import _build
    import nerfdet_amd.rays as R
    from nerfdet_amd import autograd as A
    from nerfdet_amd.train import build_optimizer, train_one_step
    det = _build(device)
    det.N_rand = 256
    opt = build_optimizer(det)
    scene = learnable_scene(device)
    orig = R.sample_along_camera_ray
    R.sample_along_camera_ray = lambda *a, **k: orig(*a, **{**k, "det": True})
    prev = A.set_deterministic(deterministic)
    rows, norms = [], []
    try:
        for k in range(STEPS):
            R.rng = np.random.RandomState(1000 + k)
            out = train_one_step(det, scene, opt)
            rows.append([out["log_vars"][n] for n in KEYS])
            norms.append(out["grad_norm"])
    finally:
        A.set_deterministic(prev)
        R.sample_along_camera_ray = orig
    w = torch.cat([p.detach().reshape(-1) for p in det.parameters() if p.requires_grad]).cpu()
    return np.array(rows), np.array(norms), w


@pytest.mark.timeout(900)
def test_five_optimizer_steps_follow_the_cpu_trajectory(device):
    """"chaotic" and "wrong" separated by evidence.  (a) With the deterministic scatter two runs of the same five steps on
    freshly built models end within 1e-6 of each other (the backward itself: bit for bit, next test) -- the run-to-run spread of round 3 was the
    float atomics' ordering and nothing else.  (b) That reproducible
    run follows the trajectory of the oracle-backed CPU module (tests/golden/train_traj.npz) loss by loss, step by step.  (c) The float-atomic
    default follows it as well over these five steps (its noise is 1e-6 of the gradients; what Adam makes of it shows later)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_traj.npz"))
    ref, ref_norm = g["losses"], g["grad_norm"]
    a, na, wa = _five_steps(device, True)
    b, nb, wb = _five_steps(device, True)
    # two runs on freshly built models (other addresses: the vendor library's reductions and GEMMs pick alignment-dependent paths) agree to the
    # last ulp or two; the backward itself is reproducible bit for bit (test_deterministic_scatter_makes_one_step_bitwise_reproducible)
    assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max() and float((wa - wb).abs().max()) <= 1e-6, "the deterministic mode did not reproduce its own run"
    c, nc, wc = _five_steps(device, False)
    rel = lambda x: np.abs(x - ref) / np.maximum(np.abs(ref), 1e-3)
    # The yardstick: how far the CPU trajectory moves when every weight is perturbed by 1e-6 relative (tests/golden/make_golden_traj.py, maximum of
    # three draws): 0.4 % at step 1, 2 - 5 % at step 3, 55 % of the classification loss at step 4 -- Adam's sign-like first steps and the batch
    # statistics of a one-scene batch amplify rounding-level differences by four orders of magnitude in four steps.  A GPU run (other summation
    # orders in every kernel) cannot stay closer to the trajectory than the trajectory stays to itself: the bar is 3 x that spread, with a floor of
    # 1e-3; the first step (same weights on both sides, nothing amplified yet) must agree to 1e-5.
    spread = g["spread_1e6"] / np.maximum(np.abs(ref), 1e-3)
    bar = np.maximum(3.0 * spread, 1e-3)
    print("deterministic GPU run vs CPU trajectory, relative per loss per step:\n", np.array2string(rel(a), precision=5, suppress_small=True))
    print("float-atomic GPU run vs CPU trajectory:\n", np.array2string(rel(c), precision=5, suppress_small=True))
    print("the CPU trajectory's own spread under 1e-6 weight perturbations:\n", np.array2string(spread, precision=5, suppress_small=True))
    print("gradient norms (CPU | deterministic | atomics):", np.round(ref_norm, 3), np.round(na, 3), np.round(nc, 3))
    assert rel(a)[0].max() <= 1e-5, "the first step's losses (same weights on both sides) differ from the CPU module's"
    assert (rel(a) <= bar).all(), f"the deterministic GPU run leaves the CPU trajectory by more than 3 x the trajectory's own 1e-6 sensitivity:\n{rel(a) / bar}"
    assert (rel(c) <= bar).all(), f"the float-atomic GPU run leaves the CPU trajectory:\n{rel(c) / bar}"
    assert abs(na[0] - ref_norm[0]) <= 1e-3 * ref_norm[0]

@pytest.mark.timeout(600)
def test_deterministic_scatter_makes_one_step_bitwise_reproducible(device):
    """One training step (forward, five losses, backward) three times on the same model: with ``autograd.set_deterministic(True)`` every one of
    the parameter gradients is the same bit pattern in all three runs (K1 / K2 / K4 backward scatter with 64-bit fixed-point integer atomics:
    integer sums do not depend on the order of the adds; every other kernel of the step -- split-K and weight-gradient reductions included --
    sums in a fixed order already).  With the default float atomics some gradients differ from run to run: that, and nothing else, was the
    "chaotic" part of round 3's 30-step test.  The two modes agree to 1e-5 of each gradient's scale."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden_traj import learnable_scene
    from test_ddp import _build
    import nerfdet_amd.rays as R
    from nerfdet_amd import autograd as A
    det = _build(device)
    det.N_rand = 256
    scene = learnable_scene(device)
    orig = R.sample_along_camera_ray
    R.sample_along_camera_ray = lambda *a, **k: orig(*a, **{**k, "det": True})

    def grads():
        R.rng = np.random.RandomState(1000)
        det.zero_grad(set_to_none=True)
        det.train_step(scene)["loss"].backward()
        return {n: p.grad.detach().clone() for n, p in det.named_parameters() if p.grad is not None}
    try:
        out = {}
        for mode in (True, False):
            prev = A.set_deterministic(mode)
            try:
                out[mode] = [grads() for _ in range(3)]
            finally:
                A.set_deterministic(prev)
    finally:
        R.sample_along_camera_ray = orig
    det_runs, flt_runs = out[True], out[False]
    assert len(det_runs[0]) > 100
    bad = [n for n in det_runs[0] if not all(torch.equal(det_runs[0][n], r[n]) for r in det_runs[1:])]
    assert not bad, f"{len(bad)} gradients are not reproducible in the deterministic mode, e.g. {bad[:5]}"
    moved = [n for n in flt_runs[0] if not all(torch.equal(flt_runs[0][n], r[n]) for r in flt_runs[1:])]
    print(f"float atomics: {len(moved)} of {len(flt_runs[0])} parameter gradients differ between three runs of the same step; deterministic mode: 0")
    assert moved, "the float-atomic scatter reproduced itself bit for bit three times: the deterministic mode would have nothing to fix"
    for n, g in det_runs[0].items():
        scale = float(g.abs().max())
        assert float((g - flt_runs[0][n]).abs().max()) <= 1e-5 * max(scale, 1e-12), n
