"""Training-time convolutions on the MFMA kernels (nerfdet_amd/conv_train.py): forward, data gradient (the same kernel on the
tap-flipped transposed weight) and weight gradient (one GEMM over the flattened output grid, any stride) against PyTorch-CPU fp32
autograd of ``F.conv3d`` / ``F.conv2d`` -- the arithmetic mmdet3d/models/necks/imvoxelnet.py:22-67,233-260 runs in training."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f16x2", "bf16x3"], autouse=True)
def train_arith(request):
    """Every test of this file on both training arithmetics: the three-product fp16 pairs with device-side scales (the default) and the
    six-product bf16x3 kernels (conv3d.TRAIN_F16X2 = False)."""
    import nerfdet_amd.conv3d as C
    prev, C.TRAIN_F16X2 = C.TRAIN_F16X2, request.param == "f16x2"
    yield request.param
    C.TRAIN_F16X2 = prev


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


@pytest.mark.parametrize("dims,cin,cout,k,stride", [((9, 7, 5), 64, 96, 3, 1), ((6, 6, 4), 32, 32, 1, 1), ((12, 10, 6), 128, 64, 3, 1), ((8, 8, 4), 128, 25, 3, 1),
                                                   ((3, 11, 13), 64, 32, (3, 3), 1), ((2, 8, 6), 96, 64, (1, 1), 1), ((5, 9, 9), 32, 64, (3, 3), 1),
                                                   ((9, 8, 6), 64, 128, 3, 2), ((8, 6, 4), 64, 96, 1, 2),            # the neck's stride-2 3x3x3 / 1x1x1 downsample layers
                                                   ((3, 11, 14), 64, 64, (3, 3), 2), ((2, 9, 12), 128, 256, (1, 1), 2)])
def test_conv_s1_forward_dgrad_wgrad_vs_torch_cpu(device, dims, cin, cout, k, stride):
    from nerfdet_amd.conv_train import ConvS1
    torch.manual_seed(sum(dims) + cin)
    two_d = isinstance(k, tuple)
    ks = k if two_d else (k,) * 3
    x = torch.randn(*dims, cin)                                    # channels-last: (D,H,W,C) or (N,H,W,C)
    w = torch.randn(cout, cin, *ks) / (cin * max(1, ks[0] * ks[-1])) ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    if two_d:
        ref = F.conv2d(xr.permute(0, 3, 1, 2), wr, stride=stride, padding=tuple(v // 2 for v in ks)).permute(0, 2, 3, 1)
    else:
        ref = F.conv3d(xr.permute(3, 0, 1, 2).unsqueeze(0), wr, stride=stride, padding=k // 2)[0].permute(1, 2, 3, 0)
    gy = torch.randn(*ref.shape)
    (ref * gy).sum().backward()
    xg, wg = x.to(device).requires_grad_(True), w.to(device).requires_grad_(True)
    out = ConvS1.apply(xg, wg, stride)
    (out * gy.to(device)).sum().backward()
    assert _rel(out.detach().cpu(), ref.detach()) <= 2e-5
    assert _rel(xg.grad.cpu(), xr.grad) <= 2e-5
    assert _rel(wg.grad.cpu(), wr.grad) <= 5e-5      # a sum over every voxel, in another order


def test_neck_training_step_matches_library_path(device):
    """``FastIndoorImVoxelNeck`` in training mode (BatchNorm on batch statistics): the routed path (stride-1 convolutions on the MFMA
    kernels, forward and backward; the rest on the library) against an fp64 CPU evaluation of the same module -- outputs,
    input gradient, parameter gradients and running statistics at least as close as the all-library GPU path gets (BatchNorm over
    the 24 voxels of the coarsest level amplifies rounding, so the bar is relative to what the library achieves)."""
    import copy
    from nerfdet_amd import conv_train
    from nerfdet_amd.neck3d import FastIndoorImVoxelNeck
    torch.manual_seed(0)
    neck = FastIndoorImVoxelNeck(64, [1, 1, 1], 32).train()
    x = torch.randn(1, 64, 16, 12, 8)
    wts = [torch.randn(1, 32, 16 // 2 ** i, 12 // 2 ** i, 8 // 2 ** i) for i in range(3)]

    def run(mod, inp, w):
        inp = inp.clone().requires_grad_(True)
        outs = mod(inp)
        sum((o * ww).sum() for o, ww in zip(outs, w)).backward()
        return [o.detach().double().cpu() for o in outs], inp.grad.double().cpu(), {n: p.grad.double().cpu() for n, p in mod.named_parameters()}, \
            {n: b.double().cpu() for n, b in mod.named_buffers() if b.dtype.is_floating_point}
    exact = run(copy.deepcopy(neck).double(), x.double(), [w.double() for w in wts])
    xg = x.to(device).contiguous(memory_format=torch.channels_last_3d)
    ours = run(copy.deepcopy(neck).to(device), xg, [w.to(device) for w in wts])
    saved = conv_train.eligible
    conv_train.eligible = lambda *a, **k: False       # library everywhere
    try:
        lib = run(copy.deepcopy(neck).to(device), xg, [w.to(device) for w in wts])
    finally:
        conv_train.eligible = saved

    def worst(got):
        e = [max(_rel(a, b) for a, b in zip(got[0], exact[0])), _rel(got[1], exact[1])]
        e.append(max(_rel(got[2][n], exact[2][n]) for n in exact[2]))
        e.append(max(_rel(got[3][n], exact[3][n]) for n in exact[3]))
        return e
    e_ours, e_lib = worst(ours), worst(lib)
    print("routed path vs fp64:", e_ours, " library path vs fp64:", e_lib)
    for a, b, what in zip(e_ours, e_lib, ("outputs", "input gradient", "parameter gradients", "running statistics")):
        assert a <= max(2.0 * b, 1e-4), f"{what}: {a:.2e} (library path: {b:.2e})"


def test_head_training_forward_shares_one_convolution(device):
    """The head's three layers as one convolution over 1 + 6 + 18 output channels (training): outputs and gradients against an fp64
    CPU evaluation of the three separate layers (imvoxel_head_v2.py:444-449)."""
    import copy
    from nerfdet_amd.config import ConfigDict
    from nerfdet_amd.head import ScanNetImVoxelHeadV2
    torch.manual_seed(1)
    head = ScanNetImVoxelHeadV2(n_classes=18, n_channels=128, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18,
                                test_cfg=ConfigDict(nms_pre=1000, iou_thr=0.25, score_thr=0.01)).train()
    with torch.no_grad():
        for m in (head.centerness_conv, head.reg_conv, head.cls_conv):
            m.weight.normal_(0, 0.02)
    x = torch.randn(1, 128, 10, 8, 6)
    wts = [torch.randn(1, c, 10, 8, 6) for c in (1, 6, 18)]

    def run(h, inp, w):
        inp = inp.clone().requires_grad_(True)
        outs = h.forward_single(inp, h.scales[1])
        sum((o * ww).sum() for o, ww in zip(outs, w)).backward()
        return [o.detach().double().cpu() for o in outs], inp.grad.double().cpu(), {n: p.grad.double().cpu() for n, p in h.named_parameters() if p.grad is not None}
    exact = run(copy.deepcopy(head).double(), x.double(), [w.double() for w in wts])
    ours = run(copy.deepcopy(head).to(device), x.to(device).contiguous(memory_format=torch.channels_last_3d), [w.to(device) for w in wts])
    for a, b in zip(ours[0], exact[0]):
        assert _rel(a, b) <= 2e-5
    assert _rel(ours[1], exact[1]) <= 2e-5
    assert set(ours[2]) == set(exact[2]) and "cls_conv.bias" in ours[2] and "scales.1.scale" in ours[2]
    for n in exact[2]:
        assert _rel(ours[2][n], exact[2][n]) <= 5e-5, n


@pytest.mark.parametrize("dims,cin,cout", [((5, 4, 3), 64, 32), ((6, 5, 2), 128, 64)])
def test_conv_transposed_k2s2_forward_dgrad_wgrad_vs_torch_cpu(device, dims, cin, cout):
    """ConvT2 (the neck's up-blocks, necks/imvoxelnet.py:233-260) against PyTorch-CPU fp32 autograd of F.conv_transpose3d."""
    from nerfdet_amd.conv_train import ConvT2
    torch.manual_seed(sum(dims) + cin)
    x = torch.randn(*dims, cin)
    w = torch.randn(cin, cout, 2, 2, 2) / cin ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv_transpose3d(xr.permute(3, 0, 1, 2).unsqueeze(0), wr, stride=2)[0].permute(1, 2, 3, 0)
    gy = torch.randn(*ref.shape)
    (ref * gy).sum().backward()
    xg, wg = x.to(device).requires_grad_(True), w.to(device).requires_grad_(True)
    out = ConvT2.apply(xg, wg)
    (out * gy.to(device)).sum().backward()
    assert out.shape == ref.shape
    assert _rel(out.detach().cpu(), ref.detach()) <= 2e-5
    assert _rel(xg.grad.cpu(), xr.grad) <= 2e-5
    assert _rel(wg.grad.cpu(), wr.grad) <= 5e-5


def test_bottleneck_training_forward_backward_vs_fp64(device):
    """A ResNet bottleneck with frozen eval-mode BatchNorm in training (conv -> affine -> ReLU (+ identity) as single launches,
    nerfdet_amd/conv_train.py::ConvAffineAct; stride-2 3x3 and the 1x1 stride-2 downsample included) against an fp64 CPU evaluation
    of the same module: output, input gradient, weight gradients."""
    import copy
    from nerfdet_amd.backbone import Bottleneck
    from torch import nn
    torch.manual_seed(3)
    ds = nn.Sequential(nn.Conv2d(128, 256, 1, 2, bias=False), nn.BatchNorm2d(256))
    blk = Bottleneck(128, 64, stride=2, downsample=ds)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 2.0); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
                m.weight.requires_grad_(False); m.bias.requires_grad_(False)
    blk.eval()                                   # norm_eval=True: BatchNorm on running statistics while the weights train
    x = torch.randn(3, 128, 14, 18)
    gy = torch.randn(3, 256, 7, 9)

    def run(mod, inp, g):
        inp = inp.clone().requires_grad_(True)
        out = mod(inp)
        (out * g).sum().backward()
        return out.detach().double().cpu(), inp.grad.double().cpu(), {n: p.grad.double().cpu() for n, p in mod.named_parameters() if p.grad is not None}
    exact = run(copy.deepcopy(blk).double(), x.double(), gy.double())
    ours = run(copy.deepcopy(blk).to(device), x.to(device).contiguous(memory_format=torch.channels_last), gy.to(device))
    assert set(ours[2]) == set(exact[2]) and len(exact[2]) == 4
    assert _rel(ours[0], exact[0]) <= 2e-5 and _rel(ours[1], exact[1]) <= 2e-5
    for n in exact[2]:
        assert _rel(ours[2][n], exact[2][n]) <= 5e-5, n


@pytest.mark.parametrize("dims,cin,cout,kernel,stride,pads", [((12, 10, 6), 128, 64, (3, 3, 3), 1, None), ((9, 8, 6), 64, 128, (3, 3, 3), 2, None),
                                                              ((8, 8, 4), 128, 25, (3, 3, 3), 1, None), ((10, 6, 4), 256, 96, (2, 2, 2), 2, (0, 0, 0)),
                                                              ((3, 11, 14), 64, 64, (3, 3), 2, None), ((2, 9, 12), 192, 256, (1, 1), 1, None),
                                                              ((7, 6, 5), 128, 256, (3, 3, 3), 1, None), ((3, 9, 10), 256, 512, (3, 3), 1, None)])      # (the 128 x 256 tile of the fp16-pair form)
def test_implicit_weight_gradient_equals_the_staged_form(device, dims, cin, cout, kernel, stride, pads):
    """k_wgrad_split (x read in place, transposed into LDS) against the staged form (tap copies + one GEMM): the same products summed in
    another order, and both against an fp64 evaluation of the sum."""
    from nerfdet_amd import conv_train
    torch.manual_seed(cin + cout)
    two_d = len(kernel) == 2
    k3 = ((1,) + tuple(kernel)) if two_d else tuple(kernel)
    s3 = (1, stride, stride) if two_d else (stride,) * 3
    p3 = tuple(v // 2 for v in k3) if pads is None else tuple(pads)
    out_dims = tuple((n + 2 * p - k) // s + 1 for n, p, k, s in zip(dims, p3, k3, s3))
    x = torch.randn(*dims, cin, device=device)
    g = torch.randn(*out_dims, cout, device=device)
    staged = conv_train.weight_grad(x, g, kernel, stride, pads, implicit=False)
    implicit = conv_train.weight_grad(x, g, kernel, stride, pads, implicit=True)
    xd, gd = x.double().cpu(), g.double().cpu()
    if two_d:
        xd, gd = xd.unsqueeze(0).permute(0, 4, 1, 2, 3), gd.unsqueeze(0).permute(0, 4, 1, 2, 3)      # (1, C, N, H, W): the batch as a depth axis
    else:
        xd, gd = xd.permute(3, 0, 1, 2).unsqueeze(0), gd.permute(3, 0, 1, 2).unsqueeze(0)
    exact = torch.nn.grad.conv3d_weight(xd, (cout, cin) + k3, gd, stride=s3, padding=p3).reshape(staged.shape)
    assert implicit.shape == staged.shape
    assert _rel(implicit.cpu().double(), exact) <= 2e-5 and _rel(staged.cpu().double(), exact) <= 2e-5


@pytest.mark.parametrize("cout,cin,kernel", [(96, 64, (3, 3, 3)), (25, 128, (3, 3, 3)), (64, 32, (1, 1)), (256, 64, (3, 3)), (33, 32, (2, 2, 2))])
def test_both_weight_packs_in_one_launch(device, train_arith, cout, cin, kernel):
    """ndet_split_weights_train against the one-pack-per-launch kernel (bf16x3: the same planes bit for bit, both packs) and against the
    definition of the fp16 pair (hi = fp16(w s), lo = fp16(w s - hi), s the power of two that puts max |w| in [2^14, 2^15))."""
    import math
    from ctypes import c_void_p
    from nerfdet_amd import _lib, conv_train
    torch.manual_seed(cout + cin)
    w = (torch.randn(cout, cin, *kernel) * 0.07).to(device)
    taps = math.prod(kernel)
    planes, adj, slot = conv_train._split_both(w, taps, train_arith, True)
    kp = (cout + 31) // 32 * 32
    wt = w.reshape(cout, cin, taps)
    fwd_ref = wt.permute(2, 0, 1)                                          # (taps, Cout, Cin)
    adj_ref = torch.zeros(taps, cin, kp, device=device)
    adj_ref[:, :, :cout] = wt.flip(2).permute(2, 1, 0)                     # W'[t][ci][co] = W[co][ci][taps-1-t]
    if train_arith == "bf16x3":
        st = c_void_p(torch.cuda.current_stream().cuda_stream)
        for adjoint, got in ((0, planes), (1, adj)):
            ref = torch.empty_like(got)
            _lib.check(_lib.load().ndet_split_weights_bf16x3_torch(c_void_p(w.data_ptr()), taps, cout, cin, adjoint, c_void_p(ref.data_ptr()), st), "split")
            assert torch.equal(got, ref), adjoint
        return
    scale = 2.0 ** (15 - math.frexp(float(w.abs().max()))[1])
    for got, ref, no, ki in ((planes, fwd_ref, cout, cin), (adj, adj_ref, cin, kp)):
        hi = (ref * scale).half()
        lo = (ref * scale - hi.float()).half()
        want = torch.stack([hi, lo], 0).view(2, taps, no, ki // 32, 32).permute(1, 3, 0, 2, 4).contiguous()      # (taps, K/32, 2, No, 32)
        assert torch.equal(got.view(torch.float16), want)
    from nerfdet_amd.conv3d import amax_value
    assert amax_value(slot) == float(w.abs().max())


@pytest.mark.parametrize("n,c,relu,with_res", [(25600, 256, True, False), (3200, 512, True, True), (400, 1024, False, False), (777, 128, True, True), (50, 64, False, True),
                                               (100000, 32, True, False)])
def test_batch_norm_rows_forward_backward_vs_fp64(device, n, c, relu, with_res):
    """csrc/bn_kernels.hip (BatchNorm on batch statistics + ReLU + residual over channels-last rows, mmdet3d/models/necks/imvoxelnet.py:22-67,233-260)
    against an fp64 evaluation of relu(F.batch_norm(x) + residual): output, input / residual / affine gradients, running statistics; and not
    further from it than the library's fp32 path on the same tensors."""
    from nerfdet_amd.conv_train import BatchNormRows
    torch.manual_seed(n + c)
    x = torch.randn(n, c) * 1.7 + torch.linspace(-3, 3, c)          # per-channel means away from zero
    res = torch.randn(n, c) if with_res else None
    w, b = torch.rand(c) + 0.5, torch.randn(c) * 0.1
    gy = torch.randn(n, c)
    mom, eps = 0.1, 1e-5

    def run(dtype, dev, ours):
        xs, ws, bs = (t.to(dev, dtype).requires_grad_(True) for t in (x, w, b))
        rs = None if res is None else res.to(dev, dtype).requires_grad_(True)
        rm, rv = torch.zeros(c, device=dev, dtype=dtype), torch.ones(c, device=dev, dtype=dtype)
        if ours:
            y = BatchNormRows.apply(xs, ws, bs, rm, rv, mom, eps, relu, rs)
        else:
            y = F.batch_norm(xs, rm, rv, ws, bs, True, mom, eps)
            y = y if rs is None else y + rs
            y = torch.relu(y) if relu else y
        (y * gy.to(dev, dtype)).sum().backward()
        out = [y.detach(), xs.grad, ws.grad, bs.grad, rm, rv] + ([rs.grad] if rs is not None else [])
        return [t.double().cpu() for t in out]
    exact = run(torch.float64, "cpu", False)
    lib = run(torch.float32, device, False)
    ours = run(torch.float32, device, True)
    names = ["y", "dx", "dgamma", "dbeta", "running_mean", "running_var", "dres"]
    for name, e, l, o in zip(names, exact, lib, ours):
        err_o, err_l = _rel(o, e), _rel(l, e)
        assert err_o <= max(3e-6, 2.0 * err_l), (name, err_o, err_l)


def test_prepare_step_hands_every_convolution_weight_its_maximum(device, train_arith):
    """conv_train.prepare_step: max |w| of all trainable convolution weights in two launches; _split_both must find the slot (no per-tensor pass)
    and the slot must hold the tensor's maximum.  A weight it has not seen (another module, a concatenated head weight) falls back to its own pass."""
    from torch import nn
    from nerfdet_amd import conv_train
    from nerfdet_amd.conv3d import amax_value
    torch.manual_seed(5)
    net = nn.Sequential(nn.Conv3d(32, 64, 3, 1, 1), nn.Conv2d(64, 32, 1), nn.ConvTranspose3d(64, 32, 2, 2), nn.Linear(8, 8)).to(device)
    net[1].weight.requires_grad_(False)                       # frozen layers are not prepared
    n = conv_train.prepare_step(net)
    if train_arith != "f16x2":
        assert n == 0 and not conv_train._STEP_SLOTS
        return
    assert n == 2
    for m in (net[0], net[2]):
        slot = conv_train._step_slot(m.weight.detach())
        assert slot is not None and amax_value(slot) == float(m.weight.abs().max())
    _, _, slot = conv_train._split_both(net[0].weight.detach(), 27, "f16x2", False)
    assert slot.data_ptr() == conv_train._step_slot(net[0].weight.detach()).data_ptr()
    with torch.no_grad():
        net[0].weight.mul_(2.0)                               # the optimizer moved it: the old slot no longer applies
    _, _, slot2 = conv_train._split_both(net[0].weight.detach(), 27, "f16x2", False)
    assert slot2.data_ptr() != slot.data_ptr() and amax_value(slot2) == float(net[0].weight.abs().max())
    conv_train.prepare_step(net)
    # a model that is gone must not lend its maxima to a new one whose weights land on the freed addresses (same shapes, same version counts)
    shapes = [(m.weight.data_ptr(), m.weight._version) for m in (net[0], net[2])]
    del net, m, slot, slot2
    import gc
    gc.collect()
    torch.manual_seed(6)
    net2 = nn.Sequential(nn.Conv3d(32, 64, 3, 1, 1), nn.Conv2d(64, 32, 1), nn.ConvTranspose3d(64, 32, 2, 2), nn.Linear(8, 8)).to(device)
    with torch.no_grad():
        net2[0].weight.mul_(100.0)
    reused = (net2[0].weight.data_ptr(), net2[0].weight._version) in shapes
    _, _, slot3 = conv_train._split_both(net2[0].weight.detach(), 27, "f16x2", False)
    assert amax_value(slot3) == float(net2[0].weight.abs().max()), ("stale slot", reused)
