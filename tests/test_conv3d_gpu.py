"""GPU parity of the hand-written fp32-MFMA Conv3d (csrc/conv3d_kernels.hip) against plain PyTorch fp32 references:
F.conv3d / F.conv_transpose3d / F.batch_norm evaluated on the CPU (the oracle's ops), and the library modules."""
import copy
import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu


def _ref(x_ndhwc, conv, bn=None, residual=None, relu=0):
    x = x_ndhwc.permute(3, 0, 1, 2).unsqueeze(0)
    y = conv(x)
    if bn is not None:
        y = bn(y)
    if relu == 2:
        y = F.relu(y)
    if residual is not None:
        y = y + residual.permute(3, 0, 1, 2).unsqueeze(0)
    if relu == 1:
        y = F.relu(y)
    return y[0].permute(1, 2, 3, 0).contiguous()


CASES = [
    # cin, cout, grid, k, stride, transposed, bn, relu, residual, splits, tile
    (32, 64, (6, 5, 4), 3, 1, False, True, 1, True, 1, 64),
    (64, 40, (7, 6, 5), 3, 1, False, False, 0, False, 1, 64),     # Cout not a tile multiple, conv bias
    (64, 128, (9, 8, 6), 3, 2, False, True, 1, False, 1, 64),     # stride 2, odd sizes
    (32, 64, (8, 8, 4), 1, 2, False, True, 0, False, 1, 64),      # 1x1x1 stride-2 downsample
    (64, 32, (5, 4, 3), 2, 2, True, True, 1, False, 1, 64),       # ConvTranspose3d k2 s2
    (64, 128, (6, 6, 4), 3, 1, False, True, 2, True, 3, 64),      # split-K + relu-before-residual
    (128, 256, (10, 10, 8), 3, 1, False, True, 1, True, 1, 128),  # big tile
    (256, 128, (12, 12, 8), 3, 1, False, True, 1, False, 2, 128), # big tile + split-K
    (128, 25, (10, 10, 4), 3, 1, False, False, 0, False, 0, 0),   # head-like, auto
]


@pytest.fixture(params=["f32", "bf16x3"])
def arith(request):
    """Run a test on both kernel families: fp32-input MFMA and the 3-term bf16 split on the bf16 matrix cores."""
    from nerfdet_amd import conv3d
    prev = conv3d.set_arithmetic(request.param)
    yield request.param
    conv3d.set_arithmetic(prev)


@pytest.mark.parametrize("cin,cout,grid,k,stride,tr,use_bn,relu,use_res,splits,tile", CASES)
def test_conv3d_matches_torch_fp32(device, arith, cin, cout, grid, k, stride, tr, use_bn, relu, use_res, splits, tile):
    from nerfdet_amd.conv3d import conv3d_ndhwc, packed
    torch.manual_seed(cin * 7 + cout)
    if tr:
        conv = nn.ConvTranspose3d(cin, cout, 2, 2, bias=False)
    else:
        conv = nn.Conv3d(cin, cout, k, stride, k // 2, bias=not use_bn)
    bn = None
    if use_bn:
        bn = nn.BatchNorm3d(cout).eval()
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    x = torch.randn(*grid, cin)
    with torch.no_grad():
        probe = _ref(x, conv, bn)
        res = torch.randn_like(probe) if use_res else None
        ref = _ref(x, conv, bn, res, relu)
        conv_d, bn_d = conv.to(device), (bn.to(device) if bn is not None else None)
        got = conv3d_ndhwc(x.to(device), packed([conv_d], bn_d), residual=None if res is None else res.to(device), relu=relu,
                           splits=splits, tile=tile)
    assert got.shape == ref.shape
    scale = max(1.0, float(ref.abs().max()))
    err = float((got.cpu() - ref).abs().max())
    assert err <= 2e-5 * scale, (err, scale)
    # split-K is a fixed-order reduction: bitwise reproducible
    got2 = conv3d_ndhwc(x.to(device), packed([conv_d], bn_d), residual=None if res is None else res.to(device), relu=relu,
                        splits=splits, tile=tile)
    assert torch.equal(got, got2)


def test_conv3d_rejects_unsupported(device):
    from nerfdet_amd.conv3d import conv3d_ndhwc, packed
    conv = nn.Conv3d(24, 32, 3, 1, 1).to(device)
    with pytest.raises(ValueError):
        conv3d_ndhwc(torch.randn(4, 4, 4, 24, device=device), packed([conv]))
    with pytest.raises(RuntimeError):
        conv3d_ndhwc(torch.randn(4, 4, 4, 32), packed([nn.Conv3d(32, 32, 3, 1, 1)]))


def test_neck_hip_matches_library_and_oracle_at_real_width(device):
    """FastIndoorImVoxelNeck(256 -> 128) on a 16x16x8 volume: MFMA path vs the vendor-library path on the GPU and vs
    the oracle's functional restatement on the CPU."""
    from nerfdet_amd.neck3d import FastIndoorImVoxelNeck
    from oracle import nerfdet_oracle as O
    torch.manual_seed(0)
    neck = FastIndoorImVoxelNeck(256, [1, 1, 1], 128)
    with torch.no_grad():
        for m in neck.modules():
            if isinstance(m, nn.BatchNorm3d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.8, 1.2); m.weight.uniform_(0.8, 1.2); m.bias.normal_(0, 0.1)
    neck.eval()
    x = torch.randn(1, 256, 16, 16, 8) * 0.5
    with torch.no_grad():
        ref = O.neck3d_forward(dict(neck.state_dict()), x)
        neck.to(device)
        xd = x.to(device).contiguous(memory_format=torch.channels_last_3d)
        hip = neck(xd)
        lib = neck.forward_library(x.to(device))
    for i in range(3):
        assert hip[i].shape == ref[i].shape
        s = float(ref[i].abs().max())
        assert float((hip[i].cpu() - ref[i]).abs().max()) <= 1e-4 * max(1.0, s), i
        assert float((hip[i] - lib[i]).abs().max()) <= 1e-4 * max(1.0, s), i
    # training mode keeps batch statistics -> library path, and still differentiable
    neck.train()
    y = neck(x.to(device).requires_grad_(True))
    assert y[0].requires_grad


def test_head_hip_matches_library(device):
    from nerfdet_amd.config import ConfigDict
    from nerfdet_amd.head import ScanNetImVoxelHeadV2
    torch.manual_seed(1)
    head = ScanNetImVoxelHeadV2(n_classes=18, n_channels=128, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18,
                                test_cfg=ConfigDict(nms_pre=1000, iou_thr=0.25, score_thr=0.01))
    with torch.no_grad():
        head.cls_conv.weight.normal_(0, 0.05); head.cls_conv.bias.normal_(-2, 0.3)
        head.reg_conv.weight.normal_(0, 0.02); head.centerness_conv.weight.normal_(0, 0.05)
        for i, sc in enumerate(head.scales):
            sc.scale.fill_(1.0 + 0.3 * i)
    head.to(device).eval()
    feats = [torch.randn(1, 128, 16 // 2 ** i, 16 // 2 ** i, 8 // 2 ** i, device=device) for i in range(3)]
    with torch.no_grad():
        ctr, reg, cls = head(feats)
        for i in range(3):
            c0, r0, k0 = head.centerness_conv(feats[i]), torch.exp(head.scales[i](head.reg_conv(feats[i]))), head.cls_conv(feats[i])
            torch.testing.assert_close(ctr[i], c0, rtol=1e-4, atol=2e-5)
            torch.testing.assert_close(reg[i], r0, rtol=1e-4, atol=2e-5)
            torch.testing.assert_close(cls[i], k0, rtol=1e-4, atol=2e-5)


CASES_2D = [
    # cin, cout, (n,h,w), k, stride, bn, relu, residual
    (64, 64, (3, 12, 16), 1, 1, True, 1, False),
    (64, 256, (3, 12, 16), 1, 1, True, 1, True),
    (128, 128, (2, 13, 17), 3, 2, True, 1, False),     # 3x3 stride 2, odd sizes
    (256, 512, (2, 9, 11), 1, 2, True, 0, False),      # 1x1 stride-2 downsample
    (256, 256, (2, 15, 20), 3, 1, False, 0, False),    # FPN output conv (bias, no norm)
    (2048, 256, (2, 8, 10), 1, 1, False, 0, False),    # FPN lateral
]


@pytest.mark.parametrize("cin,cout,nhw,k,stride,use_bn,relu,use_res", CASES_2D)
def test_conv2d_nhwc_matches_torch_fp32(device, arith, cin, cout, nhw, k, stride, use_bn, relu, use_res):
    from nerfdet_amd.conv3d import conv2d_nhwc, packed
    torch.manual_seed(cin + cout + k)
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=not use_bn)
    bn = None
    if use_bn:
        bn = nn.BatchNorm2d(cout).eval()
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    x = torch.randn(*nhw, cin)
    with torch.no_grad():
        y = conv(x.permute(0, 3, 1, 2))
        if bn is not None:
            y = bn(y)
        res = torch.randn_like(y) if use_res else None
        if res is not None:
            y = y + res
        if relu:
            y = F.relu(y)
        ref = y.permute(0, 2, 3, 1).contiguous()
        got = conv2d_nhwc(x.to(device), packed([conv.to(device)], None if bn is None else bn.to(device)),
                          residual=None if res is None else res.permute(0, 2, 3, 1).contiguous().to(device), relu=relu)
    assert got.shape == ref.shape
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("tile", [64, 128, 12864, 128256])
def test_split_conv_is_fp32_accurate_against_fp64(device, tile):
    """The 3-term bf16 split keeps fp32 accuracy: against an fp64 convolution its error is at the level of the
    fp32-MFMA kernel's (and of the CPU fp32 convolution), orders below a plain bf16 product (~4e-3)."""
    from nerfdet_amd import conv3d
    torch.manual_seed(5)
    conv = nn.Conv3d(256, 128, 3, 1, 1, bias=False)
    x = torch.randn(12, 12, 8, 256) * torch.logspace(-3, 3, 256)  # channels spanning six decades
    with torch.no_grad():
        ref64 = F.conv3d(x.double().permute(3, 0, 1, 2).unsqueeze(0), conv.weight.double(), padding=1)[0].permute(1, 2, 3, 0)
        ref32 = F.conv3d(x.permute(3, 0, 1, 2).unsqueeze(0), conv.weight, padding=1)[0].permute(1, 2, 3, 0)
        pk = conv3d.packed([conv.to(device)])
        prev = conv3d.set_arithmetic("bf16x3")
        try:
            got_split = conv3d.conv3d_ndhwc(x.to(device), pk, tile=tile, splits=1).cpu()
        finally:
            conv3d.set_arithmetic(prev)
            conv3d.DIRECT_EPILOGUE = True
        got_f32 = conv3d.conv3d_ndhwc(x.to(device), pk, splits=1).cpu()
    scale = float(ref64.abs().mean())
    e_split = float((got_split.double() - ref64).abs().max()) / scale
    e_f32 = float((got_f32.double() - ref64).abs().max()) / scale
    e_cpu = float((ref32.double() - ref64).abs().max()) / scale
    assert e_split <= 2.0 * max(e_f32, e_cpu) and e_split < 1e-4, (e_split, e_f32, e_cpu)


WS_CASES = [
    # cin, cout, grid, k, stride, transposed, relu, residual, splits   (wave-specialised 128 x 256 tile of the bf16x3 kernel)
    (64, 256, (9, 8, 6), 3, 1, False, 1, True, 1),      # M = 432: ragged last M tile, padding taps everywhere
    (96, 300, (7, 6, 5), 3, 2, False, 0, False, 1),     # Cout past one N tile and not a tile multiple, stride 2
    (128, 256, (6, 6, 4), 3, 1, False, 2, True, 3),     # split-K + ReLU before the residual
    (64, 512, (5, 4, 3), 2, 2, True, 1, False, 1),      # transposed k2 s2
    (256, 25, (10, 10, 4), 3, 1, False, 0, False, 2),   # Cout < 4-aligned (scalar column epilogue)
    (32, 256, (3, 12, 16), 1, 1, False, 1, False, 1),   # 1x1x1, a single K step
]


@pytest.mark.parametrize("cin,cout,grid,k,stride,tr,relu,use_res,splits", WS_CASES)
def test_split_conv_wave_specialised_tile(device, cin, cout, grid, k, stride, tr, relu, use_res, splits):
    from nerfdet_amd import conv3d
    torch.manual_seed(cin + cout + k)
    conv = nn.ConvTranspose3d(cin, cout, 2, 2, bias=False) if tr else nn.Conv3d(cin, cout, k, stride, k // 2, bias=False)
    bn = nn.BatchNorm3d(cout).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    x = torch.randn(*grid, cin)
    with torch.no_grad():
        probe = _ref(x, conv, bn)
        res = torch.randn_like(probe) if use_res else None
        ref = _ref(x, conv, bn, res, relu)
        pk = conv3d.packed([conv.to(device)], bn.to(device))
        prev = conv3d.set_arithmetic("bf16x3")
        try:
            got = conv3d.conv3d_ndhwc(x.to(device), pk, residual=None if res is None else res.to(device), relu=relu, splits=splits, tile=128256)
        finally:
            conv3d.set_arithmetic(prev)
            conv3d.DIRECT_EPILOGUE = True
    assert got.shape == ref.shape
    scale = max(1.0, float(ref.abs().max()))
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * scale


WSP_CASES = [
    # cin, cout, grid, k, stride, relu, residual (0 none / 1 plain / 2 nearest-x2 upsampled), splits, tile   (persistent wave-specialised tiles)
    (64, 256, (9, 8, 6), 3, 1, 1, 1, 1, 129256),        # M = 432: ragged last M tile, padding taps everywhere
    (96, 304, (7, 6, 5), 3, 2, 0, 0, 1, 129256),        # Cout past one N tile (304 = 256 + 48), stride 2
    (128, 256, (6, 6, 4), 3, 1, 2, 1, 3, 129256),       # split-K + ReLU before the residual
    (32, 256, (3, 12, 16), 1, 1, 1, 0, 1, 129256),      # 1x1x1, a single K step
    (256, 512, (1, 40, 52), 1, 1, 1, 1, 1, 129256),     # 2 080 rows x 512: 17 x 2 tiles over the persistent grid, 8 K steps
    (64, 256, (1, 21, 30), 1, 1, 0, 2, 1, 129256),      # FPN lateral: nearest-x2 upsampled residual, odd map height
    (128, 256, (1, 150, 200), 1, 1, 1, 1, 1, 129256),   # 30 000 rows: more tiles than CUs, every workgroup walks several
    (64, 256, (9, 8, 6), 3, 1, 1, 1, 1, 129064),        # the same set on 64-row tiles
    (96, 304, (7, 6, 5), 3, 2, 0, 0, 1, 129064),
    (128, 256, (6, 6, 4), 3, 1, 2, 1, 3, 129064),
    (32, 256, (3, 12, 16), 1, 1, 1, 0, 1, 129064),
    (256, 512, (1, 40, 52), 1, 1, 1, 1, 1, 129064),
    (64, 256, (1, 21, 30), 1, 1, 0, 2, 1, 129064),
    (1024, 256, (1, 150, 100), 1, 1, 1, 0, 2, 129064),  # stage-3 conv1 shape (15 000 rows, 32 K steps), split-K 2
    (64, 256, (9, 8, 6), 3, 1, 1, 1, 1, 129257),        # the same set with eight consumer waves
    (96, 304, (7, 6, 5), 3, 2, 0, 0, 1, 129257),
    (128, 256, (6, 6, 4), 3, 1, 2, 1, 3, 129257),
    (32, 256, (3, 12, 16), 1, 1, 1, 0, 1, 129257),
    (256, 512, (1, 40, 52), 1, 1, 1, 1, 1, 129257),
    (64, 256, (1, 21, 30), 1, 1, 0, 2, 1, 129257),
    (128, 256, (1, 150, 200), 1, 1, 1, 1, 1, 129257),
    (64, 96, (2, 21, 30), 1, 1, 1, 2, 1, 129256),       # upsampled residual with Cout * 4 NOT a power of two (row offset + column offset, not OR)
    (64, 304, (1, 21, 30), 1, 1, 0, 2, 1, 129064),
    (96, 96, (2, 13, 18), 1, 1, 1, 2, 1, 129257),
]


@pytest.mark.parametrize("cin,cout,grid,k,stride,relu,res_mode,splits,tile", WSP_CASES)
def test_split_conv_persistent_wave_specialised_tile(device, cin, cout, grid, k, stride, relu, res_mode, splits, tile):
    """k_conv_split_wsp (tile codes 129256 / 129064): one workgroup per CU walks a list of tiles, the consumers store from the MFMA's C layout
    while the producers stage the next tile.  Against the fp32 reference, and BIT-IDENTICAL to the one-shot tile 128256 (same K walk, same
    epilogue arithmetic)."""
    from nerfdet_amd import conv3d
    torch.manual_seed(cin + cout + k)
    two_d = res_mode == 2
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False) if two_d else nn.Conv3d(cin, cout, k, stride, k // 2, bias=False)
    bn = (nn.BatchNorm2d if two_d else nn.BatchNorm3d)(cout).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    x = torch.randn(*grid, cin)
    with torch.no_grad():
        if two_d:                                # FPN lateral: (N,H,W,C) maps, residual (N, ceil(OH/2), ceil(OW/2), Cout) read at (h >> 1, w >> 1)
            y = bn(conv(x.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
            n, oh, ow = y.shape[:3]
            up = torch.randn(n, (oh + 1) // 2, (ow + 1) // 2, cout)
            y = y + up[:, torch.arange(oh) // 2][:, :, torch.arange(ow) // 2]
            ref = F.relu(y) if relu == 1 else y
        else:
            probe = _ref(x, conv, bn)
            res = torch.randn_like(probe) if res_mode == 1 else None
            ref = _ref(x, conv, bn, res, relu)
        pk = conv3d.packed([conv.to(device)], bn.to(device))
        prev = conv3d.set_arithmetic("bf16x3")
        try:
            if two_d:
                kw = dict(residual=up.to(device), residual_up2=True, relu=relu)
                got = conv3d.conv2d_nhwc(x.to(device), pk, tile=tile, **kw)
                one_shot = conv3d.conv2d_nhwc(x.to(device), pk, tile=128256, **kw)
            else:
                kw = dict(residual=None if res is None else res.to(device), relu=relu, splits=splits)
                got = conv3d.conv3d_ndhwc(x.to(device), pk, tile=tile, **kw)
                one_shot = conv3d.conv3d_ndhwc(x.to(device), pk, tile=128256, **kw)
        finally:
            conv3d.set_arithmetic(prev)
            conv3d.DIRECT_EPILOGUE = True
    assert got.shape == ref.shape
    scale = max(1.0, float(ref.abs().max()))
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * scale
    assert torch.equal(got, one_shot), "the persistent tile must reproduce the one-shot tile bit for bit"


HALO_CASES = [
    # cin, cout, grid (D,H,W), kernel (kd,kh,kw), relu, residual, splits, tile   (halo-stationary tiles of the bf16x3 kernel)
    (64, 128, (8, 8, 8), (3, 3, 3), 1, True, 1, 3128),        # exact patches
    (64, 160, (5, 7, 9), (3, 3, 3), 0, False, 1, 3128),       # ragged on every axis, Cout past one N tile
    (96, 128, (6, 6, 4), (3, 3, 3), 2, True, 3, 3128),        # split-K over the 3 chunks + ReLU before the residual
    (64, 256, (3, 12, 16), (1, 3, 3), 1, False, 1, 3256),     # 2D 3x3 over a batch of 3 maps, 256 channels
    (128, 300, (2, 15, 20), (1, 3, 3), 1, True, 2, 3256),     # odd map size, Cout ragged, split-K
    (64, 25, (10, 10, 4), (3, 3, 3), 0, False, 1, 3128),      # scalar-column epilogue
    (32, 128, (4, 9, 5), (3, 1, 3), 1, False, 1, 3128),       # mixed kernel extents, a single chunk
    (64, 256, (6, 8, 16), (3, 3, 3), 1, True, 1, 3256),       # 3x3x3 on the 256-channel tile: depth taps looped outside the halo
    (96, 288, (5, 7, 9), (3, 3, 3), 0, False, 3, 3256),       # the same, ragged everywhere, split-K
    (64, 256, (3, 12, 16), (1, 3, 3), 1, True, 1, 3257),      # 256 channels on eight consumer waves (two per SIMD)
    (96, 300, (5, 7, 9), (3, 3, 3), 2, True, 3, 3257),        # the same: 3x3x3, ragged, split-K, ReLU before the residual
    (256, 256, (2, 24, 32), (1, 3, 3), 1, False, 1, 3257),    # 8 chunks x 9 taps
]


@pytest.mark.parametrize("cin,cout,grid,kern,relu,use_res,splits,tile", HALO_CASES)
def test_split_conv_halo_tile(device, cin, cout, grid, kern, relu, use_res, splits, tile):
    from nerfdet_amd import conv3d
    torch.manual_seed(cin + cout + sum(kern))
    conv = nn.Conv3d(cin, cout, kern, 1, tuple(k // 2 for k in kern), bias=False)
    bn = nn.BatchNorm3d(cout).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    x = torch.randn(*grid, cin)
    with torch.no_grad():
        probe = _ref(x, conv, bn)
        res = torch.randn_like(probe) if use_res else None
        ref = _ref(x, conv, bn, res, relu)
        pk = conv3d.packed([conv.to(device)], bn.to(device))
        out = torch.empty(ref.shape, device=device)
        got = conv3d._conv_split(x.to(device), pk, out, grid, kern, (1, 1, 1), tuple(k // 2 for k in kern), False,
                                 None if res is None else res.to(device), False, relu, splits, tile, ref.shape[0] * ref.shape[1] * ref.shape[2],
                                 kern[0] * kern[1] * kern[2] * (cin // 32), 0)
    scale = max(1.0, float(ref.abs().max()))
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * scale


def test_split_planes_sum_exactly(device):
    from nerfdet_amd import conv3d
    torch.manual_seed(6)
    w = (torch.randn(3, 37, 64) * torch.logspace(-20, 20, 64)).to(device)
    planes = conv3d.split_planes(dict(w=w)).view(torch.bfloat16).double()   # (taps, Cin/32, 3, Cout, 32)
    back = planes.sum(2).permute(0, 2, 1, 3).reshape(3, 37, 64)
    assert torch.equal(back, w.double())


def test_resnet_fpn_hip_matches_library(device):
    """ResNet-50 + FPN inference: fused MFMA bottlenecks vs the vendor-library modules on the same weights."""
    from nerfdet_amd.backbone import FPN, ResNet
    torch.manual_seed(0)
    net = ResNet(50, frozen_stages=1, norm_cfg=dict(type="BN", requires_grad=False), norm_eval=True)
    net.init_weights()
    with torch.no_grad():  # calibrated-looking BN statistics so activations stay O(1) through 50 layers
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_var.fill_(2.0); m.running_mean.normal_(0, 0.05); m.weight.uniform_(0.8, 1.2); m.bias.normal_(0, 0.05)
    fpn = FPN([256, 512, 1024, 2048], 256, 4)
    fpn.init_weights()
    net.to(device).eval().to(memory_format=torch.channels_last)
    fpn.to(device).eval().to(memory_format=torch.channels_last)
    x = torch.randn(4, 3, 96, 128, device=device).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        c_hip = net(x)
        p_hip = fpn(c_hip)
        c_lib = net.forward_library(x)
        fpn.use_hip = False
        p_lib = fpn(c_lib)
        fpn.use_hip = True
    for a, b in zip(c_hip + p_hip, c_lib + p_lib):
        assert a.shape == b.shape
        s = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, s), (tuple(a.shape), float((a - b).abs().max()), s)
    # training keeps autograd -> library path
    net.train()
    y = net(x.requires_grad_(True))
    assert y[-1].requires_grad


@pytest.mark.parametrize("cin,cout,grid,k,relu,use_res,tile", [(64, 96, (10, 9, 7), 3, 1, True, 0), (256, 256, (16, 16, 8), 3, 0, False, 3257),
                                                               (128, 64, (1, 24, 20), 1, 1, False, 0), (256, 128, (12, 12, 8), 3, 1, False, 128256)])
def test_bf16_arithmetic_is_a_bf16_rounded_convolution(device, cin, cout, grid, k, relu, use_res, tile):
    """``set_arithmetic("bf16")``: exactly the convolution of the bf16-ROUNDED operands accumulated in fp32 (what bf16 autocast
    computes) -- checked against PyTorch-CPU fp32 on operands rounded the same way; and measurably different from the fp32 result,
    so the mode really drops the residual products."""
    from nerfdet_amd import conv3d
    torch.manual_seed(cin + cout)
    conv = torch.nn.Conv3d(cin, cout, k, 1, k // 2, bias=False)
    x = torch.randn(*grid, cin)
    res = torch.randn(*grid, cout) if use_res else None
    xr, wr = x.bfloat16().float(), conv.weight.detach().bfloat16().float()
    ref = torch.nn.functional.conv3d(xr.permute(3, 0, 1, 2).unsqueeze(0), wr, padding=k // 2)[0].permute(1, 2, 3, 0)
    full = torch.nn.functional.conv3d(x.permute(3, 0, 1, 2).unsqueeze(0), conv.weight.detach(), padding=k // 2)[0].permute(1, 2, 3, 0)
    if res is not None:
        ref, full = ref + res, full + res
    if relu:
        ref, full = ref.relu(), full.relu()
    conv.to(device)
    prev = conv3d.set_arithmetic("bf16")
    try:
        got = conv3d.conv3d_ndhwc(x.to(device), conv3d.packed([conv]), residual=None if res is None else res.to(device), relu=relu, tile=tile).cpu()
    finally:
        conv3d.set_arithmetic(prev)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * scale
    assert float((got - full).abs().max()) >= 1e-4 * scale


@pytest.mark.parametrize("cin,mid,cout,nhw,k,stride,use_res,relu3", [
    (64, 64, 256, (3, 13, 17), 3, 1, True, 1),      # stage-1 bottleneck tail, ragged last tile
    (64, 64, 256, (2, 12, 16), 3, 1, False, 0),     # no residual, no final ReLU
    (128, 128, 512, (2, 13, 18), 3, 2, True, 1),    # stage-2 first block: stride-2 3x3, 128-channel intermediate
    (128, 128, 512, (2, 10, 12), 3, 1, True, 2),    # ReLU before the residual add
    (256, 64, 128, (2, 9, 11), 1, 1, True, 1),      # 1x1 -> 1x1
])
def test_conv_chain_matches_torch_fp32_and_the_two_launches(device, cin, mid, cout, nhw, k, stride, use_res, relu3):
    """k_conv_split_chain (conv -> BN -> ReLU -> 1x1 conv -> BN -> (+residual) -> ReLU in one launch) against PyTorch-CPU fp32 and against the
    two separate launches it replaces: same arithmetic, the intermediate is the same fp32 value -> equal to rounding of the fp32 sums."""
    from nerfdet_amd import conv3d
    from nerfdet_amd.conv3d import conv2d_chain_nhwc, conv2d_nhwc, packed, chain_ok
    torch.manual_seed(cin + mid + cout + k)
    c2, c3 = nn.Conv2d(cin, mid, k, stride, k // 2, bias=False), nn.Conv2d(mid, cout, 1, bias=False)
    b2, b3 = nn.BatchNorm2d(mid).eval(), nn.BatchNorm2d(cout).eval()
    with torch.no_grad():
        for bn in (b2, b3):
            bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    x = torch.randn(*nhw, cin)
    with torch.no_grad():
        y = b3(c3(F.relu(b2(c2(x.permute(0, 3, 1, 2))))))
        res = torch.randn_like(y) if use_res else None
        if relu3 == 2:
            y = F.relu(y)
        if res is not None:
            y = y + res
        if relu3 == 1:
            y = F.relu(y)
        ref = y.permute(0, 2, 3, 1).contiguous()
        for m in (c2, c3, b2, b3):
            m.to(device)
        pk2, pk3 = packed([c2], b2), packed([c3], b3)
        assert chain_ok(pk2, pk3)
        rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(device)
        got = conv2d_chain_nhwc(x.to(device), pk2, pk3, residual=rd, relu=relu3)
        two = conv2d_nhwc(conv2d_nhwc(x.to(device), pk2, relu=1), pk3, residual=rd, relu=relu3)
        prev = conv3d.set_arithmetic("bf16")
        try:
            got16 = conv2d_chain_nhwc(x.to(device), pk2, pk3, residual=rd, relu=relu3)
            two16 = conv2d_nhwc(conv2d_nhwc(x.to(device), pk2, relu=1), pk3, residual=rd, relu=relu3)
        finally:
            conv3d.set_arithmetic(prev)
            conv3d.DIRECT_EPILOGUE = True
    scale = max(1.0, float(ref.abs().max()))
    assert got.shape == ref.shape
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * scale
    assert float((got - two).abs().max()) <= 4e-6 * scale
    assert float((got16 - two16).abs().max()) <= 4e-6 * scale          # one-product arithmetic: same bf16-rounded operands either way
    assert float((got16.cpu() - ref).abs().max()) > 1e-4 * scale          # ... and really one product


@pytest.mark.parametrize("nhw,layout", [((3, 64, 96), "nchw"), ((2, 61, 83), "nchw"), ((2, 50, 70), "nhwc"), ((1, 240, 320), "nchw"), ((2, 17, 23), "view")])
def test_fused_stem_matches_torch_fp32(device, nhw, layout):
    """k_stem_conv_pool (7x7 stride-2 conv + eval BatchNorm + ReLU + 3x3 stride-2 max-pool in one launch) against PyTorch-CPU fp32 of the
    same four modules: odd sizes (ragged tiles, pool windows hanging over the border), both image layouts and a strided view."""
    from nerfdet_amd.conv3d import stem_conv_bn_relu_maxpool, stem_ok
    torch.manual_seed(sum(nhw))
    n, h, w = nhw
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    bn = nn.BatchNorm2d(64).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    x = torch.randn(n, 3, h, w)
    with torch.no_grad():
        ref = F.max_pool2d(F.relu(bn(conv(x))), 3, 2, 1).permute(0, 2, 3, 1).contiguous()
    xd = x.to(device)
    if layout == "nhwc":
        xd = xd.contiguous(memory_format=torch.channels_last)
    elif layout == "view":
        big = torch.zeros(n, 3, h + 5, w + 9, device=device)
        big[:, :, 2:2 + h, 4:4 + w] = xd
        xd = big[:, :, 2:2 + h, 4:4 + w]
    conv.to(device); bn.to(device)
    assert stem_ok(conv, bn, xd)
    with torch.no_grad():
        got = stem_conv_bn_relu_maxpool(xd, conv, bn)
    assert got.shape == ref.shape
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    from nerfdet_amd import conv3d as C
    if C.ARITHMETIC == "f16x2":      # the stem leaves max |out| behind for the first bottleneck's fp16-pair scale (no separate pass over its output)
        assert C.amax_value(got._ndet_amax) == float(got.abs().max())


@pytest.mark.parametrize("cin,cout,nhw,k,stride,relu,use_res,tile", [(64, 256, (3, 13, 17), 1, 1, 1, True, 100064), (256, 64, (2, 12, 16), 1, 1, 1, False, 112864),
                                                                     (128, 128, (2, 13, 17), 3, 2, 1, False, 100128), (128, 96, (2, 9, 11), 3, 1, 2, True, 100064),
                                                                     (256, 512, (2, 9, 11), 1, 2, 0, False, 100128)])
def test_direct_epilogue_equals_the_staged_epilogue(device, cin, cout, nhw, k, stride, relu, use_res, tile):
    """The unified tiles' direct epilogue (stores from the MFMA's C layout through buffer operations, tile codes 100064 / 100128 / 112864)
    against the LDS-staged one: the same values, bit for bit (ragged last tile, residual, both ReLU positions, stride 2)."""
    from nerfdet_amd import conv3d
    from nerfdet_amd.conv3d import conv2d_nhwc, packed
    torch.manual_seed(cin + cout + k)
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False).to(device)
    bn = nn.BatchNorm2d(cout).eval().to(device)
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
        x = torch.randn(*nhw, cin, device=device)
        pk = packed([conv], bn)
        oh, ow = (nhw[1] + 2 * (k // 2) - k) // stride + 1, (nhw[2] + 2 * (k // 2) - k) // stride + 1
        res = torch.randn(nhw[0], oh, ow, cout, device=device) if use_res else None
        conv3d.DIRECT_EPILOGUE = False           # (otherwise the plain tile codes are promoted to the direct form as well)
        try:
            staged = conv2d_nhwc(x, pk, residual=res, relu=relu, tile=tile - 100000, splits=1)
        finally:
            conv3d.DIRECT_EPILOGUE = True
        direct = conv2d_nhwc(x, pk, residual=res, relu=relu, tile=tile)
    assert torch.equal(direct, staged)


@pytest.mark.parametrize("arith", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("grid", [(5, 9, 7), (3, 21, 30), (2, 8, 3)])
def test_upsampled_residual_rows_on_every_tile(device, arith, grid):
    """FPN lateral (1x1 conv + bias + the coarser lateral read nearest-x2-upsampled, neck fpn.py behind nerfdet.py:140): the staged epilogue
    steps the residual's (w, h, n) along a thread's rows instead of dividing per element -- maps narrower than the step (OW = 7, 3), several
    maps per tile and tiles that end mid-row, on every tile family (staged and direct epilogues), against fp64."""
    from nerfdet_amd import conv3d
    torch.manual_seed(sum(grid))
    n, h, w = grid
    conv = nn.Conv2d(64, 256, 1)
    x = torch.randn(n, h, w, 64)
    up = torch.randn(n, (h + 1) // 2, (w + 1) // 2, 256)
    with torch.no_grad():
        ref = conv.double()(x.permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1) + up.double()[:, torch.arange(h) // 2][:, :, torch.arange(w) // 2]
        conv.float()
        pk = conv3d.packed([conv.to(device)])
        prev = conv3d.set_arithmetic(arith)
        try:
            for tile in (0, 64, 128, 12864, 128256, 129256, 129064, 100064, 100128, 112864):
                conv3d.DIRECT_EPILOGUE = tile in (100064, 100128, 112864)      # the plain codes of the unified tiles: the staged epilogue
                got = conv3d.conv2d_nhwc(x.to(device), pk, residual=up.to(device), residual_up2=True, tile=tile, splits=1 if tile else 0)
                assert float((got.cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), tile
                if tile in (100064, 100128, 112864):
                    conv3d.DIRECT_EPILOGUE = False
                    staged = conv3d.conv2d_nhwc(x.to(device), pk, residual=up.to(device), residual_up2=True, tile={100064: 64, 100128: 128, 112864: 12864}[tile], splits=1)
                    assert torch.equal(got, staged), tile
        finally:
            conv3d.set_arithmetic(prev)
            conv3d.DIRECT_EPILOGUE = True


@pytest.mark.parametrize("inplanes,nhw", [(256, (3, 13, 37)), (64, (3, 13, 37)), (256, (2, 60, 80)), (64, (2, 60, 80)), (256, (1, 4, 16)), (64, (5, 3, 5))])
def test_whole_bottleneck_in_one_launch(device, inplanes, nhw):
    """csrc/bottleneck_kernels.hip: conv1 -> bn1 -> ReLU -> conv2 -> bn2 -> ReLU -> conv3 -> bn3 -> (+ identity | + bnD(convD x)) -> ReLU of a stage-1
    ResNet bottleneck (mmdet Bottleneck.forward, style 'pytorch', behind nerfdet.py:140) in ONE launch, on 4 x 16 pixel patches with conv1 evaluated
    on the patch's halo.  Against the module in fp64 on the CPU: rms error <= 1.15 x the two-launch path's (conv1, then conv2 -> conv3 chained) and
    < 1e-6 of the output's rms; elementwise <= 2e-5 x the output scale.  NOT bit-identical to the two launches, by construction: there conv1's
    output is scaled by the whole tensor's maximum, here by the patch's own (a finer scale -- the whole tensor's maximum does not exist before
    every patch has been evaluated).  Map sizes that are no multiples of the patch (ragged right / bottom edges, maps smaller than one patch)."""
    from nerfdet_amd import conv3d as C
    from nerfdet_amd.backbone import Bottleneck
    torch.manual_seed(inplanes + nhw[1])
    ds = None
    if inplanes != 256:
        ds = nn.Sequential(nn.Conv2d(inplanes, 256, 1, 1, bias=False), nn.BatchNorm2d(256))
    blk = Bottleneck(inplanes, 64, 1, ds).eval()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
        x = torch.relu(torch.randn(*nhw, inplanes)) * torch.exp(0.5 * torch.randn(*nhw, 1))
        ref = copy.deepcopy(blk).double()(x.permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1)
        blk.to(device)
        prev = C.set_arithmetic("f16x2")
        try:
            xd = x.to(device)
            C.guard_begin(device)
            assert C.bottleneck_ok(xd, C.packed([blk.conv1], blk.bn1), C.packed([blk.conv2], blk.bn2), C.packed([blk.conv3], blk.bn3),
                                   None if ds is None else C.packed([ds[0]], ds[1]))
            fused = blk.forward_nhwc(xd)
            assert C.amax_value(fused._ndet_amax) == float(fused.abs().max()), "the epilogue's max |out| is not the tensor's"
            C.FUSE_BOTTLENECKS = False
            try:
                two = blk.forward_nhwc(x.to(device))
            finally:
                C.FUSE_BOTTLENECKS = True
            assert not C.guard_tripped(device)
        finally:
            C.set_arithmetic(prev)
    rms = lambda a: ((a.cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    e_fused, e_two = rms(fused), rms(two)
    scale = max(1.0, float(ref.abs().max()))
    assert fused.shape == ref.shape
    assert float((fused.cpu().double() - ref).abs().max()) <= 2e-5 * scale
    assert e_fused < 1e-6 and e_fused <= 1.15 * e_two, (e_fused, e_two)


@pytest.mark.parametrize("tile", [3256, 3257, 3258])
@pytest.mark.parametrize("nhw", [(2, 24, 40), (3, 13, 21)])
def test_chained_projection_in_the_halo_epilogue(device, tile, nhw):
    """conv2d_nhwc(..., chain=...): the detector's 256 -> 32 feature mapping (mmdet3d/models/detectors/nerfdet.py:194-197) computed in the FPN output
    convolution's epilogue (csrc/conv_common.hpp::conv_map_rows) against the two separate modules in fp64, on every 256-column halo tile, with
    patches that hang over the map's edges; the convolution's own output must be what the plain launch writes, bit for bit."""
    from torch import nn
    import nerfdet_amd.conv3d as C
    from nerfdet_amd.backbone import _chain_pack
    if not C.projection_ok():
        pytest.skip("the chained projection belongs to the fp16-pair mode")
    torch.manual_seed(tile + nhw[1])
    conv = nn.Conv2d(256, 256, 3, 1, 1).to(device)
    lin = nn.Linear(256, 32).to(device)
    with torch.no_grad():
        conv.bias.normal_(0, 0.5)
        lin.bias.normal_(0, 0.3)
    x = torch.randn(*nhw, 256, device=device)
    pk = C.packed([conv])
    plain = C.conv2d_nhwc(x, pk, amax=False, tile=tile, splits=1)
    out, mapped = C.conv2d_nhwc(x, pk, amax=False, tile=tile, splits=1, chain=_chain_pack(pk, lin))
    assert mapped is not None and mapped.shape == (nhw[0] * nhw[1] * nhw[2], 32)
    assert torch.equal(out, plain)
    ref_o = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), conv.weight.double(), conv.bias.double(), padding=1).permute(0, 2, 3, 1)
    ref_m = torch.nn.functional.linear(ref_o, lin.weight.double(), lin.bias.double()).reshape(-1, 32)
    err = float((mapped.double() - ref_m).abs().max()) / float(ref_m.abs().max())
    assert err <= 2e-5, err
    # a tile that does not own whole rows hands the projection back to the caller
    o2, m2 = C.conv2d_nhwc(x, pk, amax=False, tile=128, chain=_chain_pack(pk, lin))
    assert m2 is None and torch.equal(o2, C.conv2d_nhwc(x, pk, amax=False, tile=128))
