"""Host-side logic that needs no GPU: the convolution tiling tables / heuristics and the bench's bookkeeping."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tiling_tables_and_heuristics_are_well_formed():
    from nerfdet_amd import conv3d
    from nerfdet_amd.conv_tuning import TUNED, TUNED_SPLIT
    f32_tiles, split_tiles = {64, 128}, {64, 128, 12864, 128256, 129256, 129257, 129064, 3128, 3256, 3257}
    for table, tiles in ((TUNED, f32_tiles), (TUNED_SPLIT, split_tiles)):
        assert len(table) >= 20
        for (m, cout, k_iters, tr), (tile, splits) in table.items():
            assert m > 0 and cout > 0 and k_iters > 0 and tr in (0, 1)
            assert tile in tiles and 1 <= splits <= 32 and splits <= k_iters
            assert not (tr and splits != 1), "transposed convolutions never split K"
            assert not (tr and tile in (3128, 3256, 3257)), "halo tiles are for stride-1 same-padded layers only"
    # every (arithmetic, tile) the tables can select has a kernel name for the bench's per-kernel roofline
    for t in split_tiles:
        assert ("bf16x3", t) in conv3d.KERNEL_NAMES
    for t in f32_tiles:
        assert ("f32", t) in conv3d.KERNEL_NAMES
    # shapes outside the tables: the heuristic returns something launchable
    for m in (400, 3200, 25600, 48000, 240000, 2_000_000):
        for cout in (25, 64, 128, 256, 1024):
            for k_iters in (2, 8, 72, 216, 864):
                for halo_ok in (False, True):
                    tile, splits = conv3d.choose_tiling_split(m + 1, cout, k_iters, halo_ok=halo_ok)   # +1: never a table key
                    assert tile in split_tiles and 1 <= splits <= min(32, k_iters)
                    assert halo_ok or tile not in (3128, 3256, 3257)
                tile, splits = conv3d.choose_tiling(m + 1, cout, k_iters)
                assert tile in f32_tiles and 1 <= splits <= min(8, k_iters)


def test_bench_bookkeeping():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    w = bench.WORKLOADS["cfg2"]
    assert (w["n_views"], w["img_hw"], w["n_voxels"]) == (50, (240, 320), (40, 40, 16))
    # every FPN feature row once + (C fp32 + int64 count) per voxel
    assert bench.k1_algorithmic_bytes(w) == 50 * 256 * 60 * 80 * 4 + (256 * 4 + 8) * 25600 == 272179200
    # 46.08 MB of images + 30.72 MB of mapped features + 70 floats per voxel
    assert bench.k2_algorithmic_bytes(w) == 4 * (50 * 3 * 240 * 320 + 50 * 32 * 60 * 80 + 70 * 25600) == 83968000
    traffic, src = bench.measured_traffic("cfg2")
    assert traffic["k_backproject_aggregate"] > bench.k1_algorithmic_bytes(w) and src.startswith("profiles/")
    # the bench names kernels without the trailing template arguments the profile carries: the dominant convolution must be found
    # span names -> the profile's instantiation names (the arithmetic scheme is the split-family kernels' last-but-one template argument)
    assert bench.profile_kernel_name("k_conv_split_halo<4,4>/f16x2", "f16x2") == "k_conv_split_halo<4,4,1,4>"
    assert bench.profile_kernel_name("k_conv_split_halo<4,4,p8>/f16x2", "f16x2") == "k_conv_split_halo<4,4,1,8>"
    assert bench.profile_kernel_name("k_conv_split_ws", "bf16x3") == "k_conv_split_ws<0>" and bench.profile_kernel_name("k_conv_split_wsp<64>", "bf16") == "k_conv_split_wsp<2,64,4>"
    assert bench.profile_kernel_name("k_conv_split_chain<64>/f16x2", "f16x2") == "k_conv_split_chain<64,1>"
    assert bench.traffic_of(traffic, "k_conv_split_halo<4,4>/f16x2", "f16x2") > 0 and bench.traffic_of(traffic, "k_density_features_packed") > 0
    assert bench.traffic_of({"k_conv_split_halo<4,4,0>": 7, "k_conv_split_halo<4,2,0>": 9}, "k_conv_split_halo<4,2>") == 9
    assert bench.traffic_of({"k_conv_split<64,64,2,2,0>": 5, "k_conv_split<64,64,2,2,1>": 6}, "k_conv_split<64,64,2,2>/f16x2", "f16x2") == 6
    assert bench.traffic_of({}, "k_x") is None
    # BASELINE.json configs[4] as stated: ResNet-101, 101 views 320x480, 80x80x32 voxels
    w5 = bench.WORKLOADS["cfg5"]
    assert (w5["n_views"], w5["img_hw"], w5["n_voxels"], w5["depth"]) == (101, (320, 480), (80, 80, 32), 101)
    assert bench.k1_algorithmic_bytes(w5) == 101 * 256 * 80 * 120 * 4 + (256 * 4 + 8) * 204800
    batch = bench.synth_batch(bench.WORKLOADS["tiny"], 3)
    assert batch["img"].shape == (1, 6, 3, 64, 96) and len(batch["img_metas"][0]["lidar2img"]["extrinsic"]) == 6
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.MFMA_BF16_PEAK_TFLOPS == 2500.0 and bench.MFMA_F32_PEAK_TFLOPS == 157.3


def test_hand_placed_vmcnt_waits_cover_their_dma_groups():
    """tools/audit_vmcnt.py: every hand-placed ``s_waitcnt vmcnt(N)`` of the convolution kernels has, on every control-flow path of
    the gfx950 disassembly, at least N vector-memory instructions between the LDS-DMA group it protects and itself."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_vmcnt.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "0 short windows" in out.stdout and "k_conv_split_ws" in out.stdout and "k_conv_split_halo" in out.stdout


def test_training_and_fusion_eligibility_rules():
    """Host-side decisions of round 2: which convolutions train on the MFMA kernels, when the bottleneck tail is chained, when the stem
    kernel and the direct epilogue apply -- pure functions of module hyper-parameters and tensor metadata (no GPU needed: CPU tensors
    are never eligible, so the rules are probed through the attributes they read)."""
    import torch
    from torch import nn
    from nerfdet_amd import conv3d, conv_train

    class FakeCuda:
        """Stands in for a float32 GPU tensor: only the attributes the rules read."""
        is_cuda, dtype = True, torch.float32
        def __init__(self, shape):
            self.shape = shape
        def dim(self):
            return len(self.shape)
    x5, x4 = FakeCuda((1, 256, 40, 40, 16)), FakeCuda((50, 64, 60, 80))
    assert conv_train.eligible(nn.Conv3d(256, 512, 3, 2, 1), x5) and conv_train.eligible(nn.Conv3d(256, 512, 1, 2, 0), x5)      # stride 2: round 2
    assert conv_train.eligible(nn.Conv2d(64, 64, 3, 1, 1), x4) and not conv_train.eligible(nn.Conv2d(3, 64, 7, 2, 3), x4)         # Cin % 32
    assert not conv_train.eligible(nn.Conv3d(256, 256, 3, 1, 0), x5) and not conv_train.eligible(nn.Conv3d(256, 256, 3, 3, 1), x5)
    assert not conv_train.eligible(nn.Conv2d(64, 64, 3, 1, 1, groups=2), x4) and not conv_train.eligible(nn.Conv2d(64, 64, 3, 1, 2, dilation=2), x4)
    assert conv_train.eligible_transposed(nn.ConvTranspose3d(1024, 512, 2, 2), x5) and not conv_train.eligible_transposed(nn.ConvTranspose3d(1024, 512, 3, 2), x5)
    bn = nn.BatchNorm2d(8)
    assert not conv_train.frozen_eval_bn(bn)                                       # training mode, trainable affine
    bn.eval(); bn.weight.requires_grad_(False); bn.bias.requires_grad_(False)
    assert conv_train.frozen_eval_bn(bn)
    pk = lambda cin, cout, k, s: dict(ndim=2, cin=cin, cout=cout, kernel=(k, k), strides=(s, s), pads=(k // 2, k // 2), transposed=False, scale=1, shift=1)
    assert conv3d.chain_ok(pk(64, 64, 3, 1), pk(64, 256, 1, 1)) and conv3d.chain_ok(pk(128, 128, 3, 2), pk(128, 512, 1, 1))
    assert not conv3d.chain_ok(pk(256, 256, 3, 1), pk(256, 1024, 1, 1))           # the 256-channel intermediate does not fit the tile
    assert not conv3d.chain_ok(pk(64, 64, 3, 1), pk(64, 256, 1, 2)) and not conv3d.chain_ok(pk(64, 64, 3, 1), pk(64, 100, 1, 1))
    prev = conv3d.set_arithmetic("f32")
    try:
        assert not conv3d.chain_ok(pk(64, 64, 3, 1), pk(64, 256, 1, 1))           # the exact fp32-MFMA family has no chained kernel
    finally:
        conv3d.set_arithmetic(prev)
    assert conv3d.choose_tiling_split(240000, 256, 2, 100064, 0) == (100064, 1)   # direct-epilogue tiles never split K
    assert conv3d.choose_tiling_split(2304, 256, 375)[1] > 4 and conv3d.choose_tiling_split(2304, 256, 375)[0] in (128256, 129256)   # ... the 128 x 256 tile (code 128256, or its persistent form since the training sweep) does


def test_ssim_closed_form_cases():
    """scikit-image is absent, so ``structural_similarity`` cannot pin rays.compute_ssim (save_rendered_img.py:21-36); it is held to the cases
    whose SSIM follows from the definition in closed form (data range 2, K1 = 0.01, K2 = 0.03 -> C1 = 4e-4, C2 = 3.6e-3), for both the
    product's torch form and the oracle's scipy form."""
    import numpy as np
    import torch
    from nerfdet_amd import rays
    from oracle import render_eval_oracle as R
    c1, c2 = (0.01 * 2.0) ** 2, (0.03 * 2.0) ** 2
    rs = np.random.RandomState(0)
    x = rs.rand(19, 23, 3)

    def both(a, b):
        got = float(rays.compute_ssim(torch.from_numpy(a), torch.from_numpy(b)))
        assert abs(got - R.ssim(a, b)) < 1e-12
        return got
    # identical images: every window has SSIM exactly 1
    assert abs(both(x, x) - 1.0) < 1e-14
    # two constant images a and a + d: variances and covariance vanish -> (2 a (a+d) + C1) / (a^2 + (a+d)^2 + C1)
    a, d = 0.3, 0.25
    want = (2 * a * (a + d) + c1) / (a * a + (a + d) ** 2 + c1)
    assert abs(both(np.full((9, 11, 3), a), np.full((9, 11, 3), a + d)) - want) < 1e-12
    # a constant offset of a textured image leaves the contrast / structure term at exactly 1: SSIM = mean over windows of the luminance term
    y = x + 0.1
    lum = []
    for c in range(3):
        for i in range(19 - 6):
            for j in range(23 - 6):
                m = x[i:i + 7, j:j + 7, c].mean()
                lum.append((2 * m * (m + 0.1) + c1) / (m * m + (m + 0.1) ** 2 + c1))
    assert abs(both(x, y) - float(np.mean(lum))) < 1e-10
    # a +-0.5 stripe image against its negative: covariance = -variance, means of opposite sign:
    # SSIM window = (-2 m^2 + C1)(-2 v + C2) / ((2 m^2 + C1)(2 v + C2))
    s = np.where((np.arange(23) % 2) == 0, 0.5, -0.5)[None, :, None] * np.ones((19, 1, 3))
    vals = []
    for j in range(23 - 6):
        w = np.tile(s[0, j:j + 7, 0], (7, 1))
        m, v = w.mean(), w.var(ddof=1)
        vals.append(((-2 * m * m + c1) * (-2 * v + c2)) / ((2 * m * m + c1) * (2 * v + c2)))
    assert abs(both(s, -s) - float(np.mean(vals))) < 1e-10 and both(s, -s) < 1.0
    # symmetric, and bounded by 1
    z = rs.rand(19, 23, 3)
    assert abs(both(x, z) - both(z, x)) < 1e-14 and both(x, z) < 1.0


def test_depth_rays_from_the_loader_replace_the_device_nonzero():
    """The training-time ray draw keeps the rays WITH depth (render_ray.py:386-404).  The loader finds them on the host (batch key
    ``depth_rays``) so that the step needs no stream synchronisation for their count; rays.begin_selection must hand back exactly what
    ``nonzero(gt_depth > 0)`` gives, and fall back to that when the key is missing."""
    import torch
    from nerfdet_amd import rays
    from nerfdet_amd.datasets import Collect3D
    from nerfdet_amd.synth import train_scene
    scene = train_scene(3, (48, 64), t_views=2, n_boxes=2, seed=3)
    scene["gt_depths"][0, 0, 5:9, 7:30] = 0.0                      # holes, as real depth maps have
    scene["depth_rays"] = torch.nonzero(scene["gt_depths"].view(-1) > 0).view(1, -1)
    rb = dict(ray_d=scene["raydirs"], gt_depth=scene["gt_depths"], depth_rays=scene["depth_rays"])
    kept, n, _ = rays.begin_selection(rb)
    ref, n_ref, _ = rays.begin_selection(dict(ray_d=scene["raydirs"], gt_depth=scene["gt_depths"]))
    assert n == n_ref < scene["gt_depths"].numel() and torch.equal(kept, ref)
    # the pipeline's collector passes the key on whenever it passes the depth maps
    out = Collect3D(keys=["gt_depths"])(dict(gt_depths=scene["gt_depths"][0], depth_rays=scene["depth_rays"][0]))
    assert "depth_rays" in out and "depth_rays" not in Collect3D(keys=["img"])(dict(img=0, depth_rays=1))


def test_linear_rows_splits_the_weight_gradient_without_changing_it():
    """autograd.LinearRows (the radiance MLP's Linear layers over 131 072 rows, the 192 000-row feature mapping): the weight gradient formed chunk by
    chunk with one batched GEMM, the bias gradient in two stages, the ReLU mask applied once -- against autograd's own Linear (+ ReLU) in fp64, for
    row counts that do and do not divide into chunks, with and without bias / ReLU, and a leading batch axis."""
    import torch
    import nerfdet_amd.autograd as A
    torch.manual_seed(0)
    keep = A.LINEAR_SPLIT_ROWS
    A.LINEAR_SPLIT_ROWS = 1024
    try:
        for n, ci, co, relu, bias in [(4096 * 3, 37, 16, True, True), (8192, 64, 3, False, True), (6000, 32, 8, True, False), (100, 5, 4, True, True)]:
            x = torch.randn(2, n // 2, ci, dtype=torch.float64, requires_grad=True)
            w = torch.randn(co, ci, dtype=torch.float64, requires_grad=True)
            b = torch.randn(co, dtype=torch.float64, requires_grad=True) if bias else None
            y = A.LinearRows.apply(x, w, b, relu)
            gy = torch.randn_like(y)
            y.backward(gy)
            got = [x.grad.clone(), w.grad.clone(), None if b is None else b.grad.clone()]
            x.grad = w.grad = None
            if b is not None:
                b.grad = None
            ref = torch.nn.functional.linear(x, w, b)
            ref = torch.relu(ref) if relu else ref
            ref.backward(gy)
            assert A._split_rows(n) >= 1 and n % A._split_rows(n) == 0
            assert torch.equal(y, ref) and torch.equal(got[0], x.grad)
            assert float((got[1] - w.grad).abs().max()) <= 1e-11 * max(1.0, float(w.grad.abs().max()))
            if b is not None:
                assert float((got[2] - b.grad).abs().max()) <= 1e-11 * max(1.0, float(b.grad.abs().max()))
    finally:
        A.LINEAR_SPLIT_ROWS = keep
