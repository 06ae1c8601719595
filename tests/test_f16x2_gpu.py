"""The fp16-pair arithmetic of the convolution kernels (conv3d.set_arithmetic("f16x2"), csrc/conv_split_kernels.hip SCH 1; the default):
fp32 operands as hi + lo fp16 pairs of the power-of-two pre-scaled tensors, three MFMA products per multiply.  Replaces the same reference
modules as the bf16x3 kernels (mmdet3d/models/necks/imvoxelnet.py:22-67,233-260, dense_heads/imvoxel_head_v2.py:45-49, the ResNet/FPN
behind detectors/nerfdet.py:140).  The claims tested here, each against an fp64 convolution on the CPU:

* on every tile family its error is at or below the six-product bf16x3 form's (three accumulator roundings per K step instead of six) and
  within 2x of the exact fp32-MFMA kernel's;
* the per-tensor maximum every epilogue leaves behind (the next layer's activation scale) is exact, on all eight XCD sub-slots' maximum;
* the scale follows the tensor: inputs of magnitude 3e4 or 1e-6 lose nothing;
* the documented limit: an element 2^-k below the tensor's maximum carries an absolute error of ~2^-40 of that maximum."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _rel_rms(a, ref):
    a, ref = a.double().cpu(), ref.double()
    return ((a - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def _layer(cin, cout, k, stride, seed):
    torch.manual_seed(seed)
    conv = nn.Conv3d(cin, cout, k, stride, k // 2, bias=False)
    bn = nn.BatchNorm3d(cout).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    return conv, bn


def _reference(conv, bn, x, res, relu):
    ref = bn.double()(conv.double()(x.permute(3, 0, 1, 2).unsqueeze(0).double()))
    if res is not None:
        ref = ref + res.permute(3, 0, 1, 2).unsqueeze(0).double()
    if relu:
        ref = torch.relu(ref)
    conv.float(); bn.float()
    return ref[0].permute(1, 2, 3, 0)


def _run(C, arith, x, pk, res, relu, device, **kw):
    prev = C.set_arithmetic(arith)
    try:
        y = C.conv3d_ndhwc(x.to(device), pk, residual=None if res is None else res.to(device), relu=relu, **kw)
        torch.cuda.synchronize()
        return y
    finally:
        C.set_arithmetic(prev)


CASES = [
    # cin, cout, grid, k, stride, tile, splits, residual
    (64, 128, (6, 10, 12), 3, 1, 64, 1, False),
    (64, 128, (6, 10, 12), 3, 1, 128, 3, True),         # split-K: the reduce pass applies the epilogue and commits the maximum
    (128, 64, (6, 10, 12), 1, 1, 12864, 1, True),
    (256, 256, (6, 10, 12), 3, 1, 128256, 1, True),     # wave-specialised one-shot tile
    (256, 512, (6, 10, 12), 1, 1, 129256, 2, False),    # persistent tiles
    (256, 256, (6, 10, 12), 3, 1, 129257, 1, True),
    (256, 512, (6, 10, 12), 1, 1, 129064, 1, False),
    (256, 256, (8, 12, 12), 3, 1, 3128, 1, True),       # halo-stationary tiles
    (128, 256, (8, 12, 12), 3, 1, 3256, 2, False),
    (256, 256, (8, 12, 12), 3, 1, 3257, 1, False),
    (256, 512, (8, 12, 12), 3, 2, 128256, 2, False),    # stride 2
    (256, 256, (8, 12, 12), 3, 1, 0, 0, False),         # the tables' own choice
]


@pytest.mark.parametrize("cin,cout,grid,k,stride,tile,splits,use_res", CASES)
def test_f16x2_error_not_above_bf16x3(device, cin, cout, grid, k, stride, tile, splits, use_res):
    from nerfdet_amd import conv3d as C
    conv, bn = _layer(cin, cout, k, stride, 11)
    x = torch.relu(torch.randn(*grid, cin)) * torch.exp(torch.randn(*grid, 1))      # post-ReLU-like, per-voxel magnitudes spread over e^+-3
    with torch.no_grad():
        od = [(g + 2 * (k // 2) - k) // stride + 1 for g in grid]
        res = torch.randn(*od, cout) if use_res else None
        ref = _reference(conv, bn, x, res, 1)
        pk = C.packed([conv.to(device)], bn.to(device))
        err = {}
        for arith in ("f32", "bf16x3", "f16x2"):
            kw = dict(tile=tile, splits=splits) if arith != "f32" else {}
            y = _run(C, arith, x, pk, res, 1, device, **kw)
            err[arith] = _rel_rms(y, ref)
            if arith == "f16x2":
                assert C.amax_value(y._ndet_amax) == float(y.abs().max()), "the epilogue's max |out| is not the tensor's"
    assert err["f16x2"] <= 1.15 * err["bf16x3"], err      # measured 0.80 - 1.04
    assert err["f16x2"] <= 2.0 * err["f32"] and err["f16x2"] < 2e-6, err


@pytest.mark.parametrize("mag", [3.0e4, 1.0, 1.0e-6, 1.0e-30])
def test_f16x2_scale_follows_the_tensor(device, mag):
    """Activations far from 1 (up to the edge of fp16's range and far below it) cost nothing: the scale is a power of two derived from max |x|."""
    from nerfdet_amd import conv3d as C
    torch.manual_seed(3)
    conv = nn.Conv3d(256, 256, 3, 1, 1, bias=False)
    x = torch.randn(8, 12, 12, 256) * mag
    with torch.no_grad():
        ref = conv.double()(x.permute(3, 0, 1, 2).unsqueeze(0).double())[0].permute(1, 2, 3, 0)
        conv.float()
        pk = C.packed([conv.to(device)])
        e16 = _rel_rms(_run(C, "f16x2", x, pk, None, 0, device, tile=3257, splits=1), ref)
        e3 = _rel_rms(_run(C, "bf16x3", x, pk, None, 0, device, tile=3257, splits=1), ref)
    assert e16 <= 1.15 * e3 and e16 < 2e-6, (mag, e16, e3)


def test_f16x2_dynamic_range_limit_is_what_the_header_says(device):
    """One scale per tensor: an element far below the tensor's maximum keeps an ABSOLUTE error of ~2^-40 of that maximum (fp16's subnormal
    spacing under the scale), not a relative one.  Rows 2^-12 below the maximum are still fp32-class; rows 2^-30 below keep ~10 bits -- the reason
    the point MLPs, whose inputs hold the reference's 1e9 garbage rows (nerfdet.py:236-243), stay on bf16x3 (conv3d.packed_linear)."""
    from nerfdet_amd import conv3d as C
    torch.manual_seed(4)
    lin = nn.Conv3d(128, 128, 1, bias=False)
    x = torch.randn(4, 8, 8, 128)
    x[0] *= 2.0 ** 30                      # one slab of "garbage" rows sets the tensor's maximum
    x[1] *= 2.0 ** 18
    with torch.no_grad():
        ref = lin.double()(x.permute(3, 0, 1, 2).unsqueeze(0).double())[0].permute(1, 2, 3, 0)
        lin.float()
        pk = C.packed([lin.to(device)])
        y = _run(C, "f16x2", x, pk, None, 0, device).cpu().double()
    rel = lambda d: ((y[d] - ref[d]).pow(2).mean().sqrt() / ref[d].pow(2).mean().sqrt()).item()
    assert rel(0) < 1e-6 and rel(1) < 1e-6, (rel(0), rel(1))              # within 2^-12 of the maximum: full precision
    amax = float(x.abs().max())
    abs_err = float((y[2:] - ref[2:]).abs().max())
    assert abs_err <= 128 * 2.0 ** -38 * amax * float(lin.weight.detach().abs().max()), (abs_err, amax)    # K x |w| x 2^-40 amax, with margin
    assert 1e-5 < rel(2) < 1e-2, rel(2)                                    # ... which is ~10 bits for rows 2^-30 below it


def test_f16x2_input_without_a_slot_takes_its_own_pass(device):
    """A tensor no convolution kernel wrote (the voxel volume, a user's tensor) has no `_ndet_amax`: ndet_amax_f32 computes it, same result."""
    from nerfdet_amd import conv3d as C
    conv, bn = _layer(64, 128, 3, 1, 5)
    x = torch.randn(6, 10, 12, 64)
    with torch.no_grad():
        pk = C.packed([conv.to(device)], bn.to(device))
        before = C.amax_fallbacks
        xd = x.to(device)
        y1 = _run(C, "f16x2", xd, pk, None, 1, device)
        assert C.amax_fallbacks == before + 1 and C.amax_value(xd._ndet_amax) == float(x.abs().max())
        y2 = _run(C, "f16x2", xd, pk, None, 1, device)           # the slot is remembered on the tensor
        assert C.amax_fallbacks == before + 1
        assert torch.equal(y1, y2)
        # an in-place write makes the remembered maximum stale (here it would overflow fp16: x grows 2^20-fold): a fresh pass is taken
        xd.mul_(2.0 ** 20)
        y3 = _run(C, "f16x2", xd, pk, None, 1, device)
        assert C.amax_fallbacks == before + 2 and C.amax_value(xd._ndet_amax) == float(xd.abs().max())
        assert torch.isfinite(y3).all()
        # ... also through a view the package made of it
        y4 = _run(C, "f16x2", y3, pk2 := C.packed([torch.nn.Conv3d(128, 64, 1, bias=False).to(device)]), None, 0, device)
        v = C.carry_amax(y3, y3.view(-1, 128).view(y3.shape))
        y3.add_(1.0e3)
        fb = C.amax_fallbacks
        y5 = _run(C, "f16x2", v, pk2, None, 0, device)
        assert C.amax_fallbacks == fb + 1 and torch.isfinite(y5).all() and not torch.equal(y4, y5)


def test_f16x2_under_inference_mode(device):
    """torch.inference_mode(): inference tensors keep no version counter (``t._version`` raises).  The library's own outputs still carry their
    slot from layer to layer; a caller's tensor is never trusted there (torch could write it in place unseen): it takes a pass per call."""
    from nerfdet_amd import conv3d as C
    conv, bn = _layer(64, 128, 3, 1, 5)
    conv2 = torch.nn.Conv3d(128, 64, 1, bias=False)
    x = torch.randn(6, 10, 12, 64)
    with torch.no_grad():
        pk, pk2 = C.packed([conv.to(device)], bn.to(device)), C.packed([conv2.to(device)])
        want = _run(C, "f16x2", _run(C, "f16x2", x.to(device), pk, None, 1, device), pk2, None, 0, device)
    with torch.inference_mode():
        xd = x.to(device)
        assert xd.is_inference()
        before = C.amax_fallbacks
        y = _run(C, "f16x2", xd, pk, None, 1, device)
        assert C.amax_fallbacks == before + 1 and not hasattr(xd, "_ndet_amax")       # the caller's tensor: a pass, no tag
        z = _run(C, "f16x2", y, pk2, None, 0, device)
        assert C.amax_fallbacks == before + 1                                          # the library's own output carried its slot
        xd.mul_(2.0 ** 20)                                                             # unseen by any counter: the next call must still be right
        y2 = _run(C, "f16x2", xd, pk, None, 1, device)
        assert C.amax_fallbacks == before + 2 and torch.isfinite(y2).all()
    assert torch.equal(z, want)


def test_f16x2_raw_pointer_write_invalidates_the_slot(device):
    """A kernel of the library refilling a caller-owned buffer (``out=``: the static volume of graphed.py) is invisible to torch's version
    counter; the write is counted per storage, so a slot tagged before it -- on that tensor or on any view over the storage -- is stale."""
    from nerfdet_amd import conv3d as C, ops
    torch.manual_seed(3)
    n_v, c, h, w, grid = 4, 64, 12, 16, (6, 6, 4)
    from oracle import nerfdet_oracle as O
    meta = O.ring_scene_meta(n_v, (4 * h, 4 * w))
    proj = ops.compute_projection(meta, 4, device)
    pts = ops.get_points(grid, (0.5, 0.5, 0.5), meta["lidar2img"]["origin"], device)
    feats = torch.randn(n_v, c, h, w, device=device).contiguous(memory_format=torch.channels_last)
    vol = torch.empty((*grid, c), device=device).permute(3, 0, 1, 2)
    cnt = torch.empty((1, *grid), dtype=torch.int64, device=device)
    conv = torch.nn.Conv3d(c, 64, 3, 1, 1, bias=False).to(device)
    with torch.no_grad():
        pk = C.packed([conv])
        ops.backproject_aggregate(feats, pts, proj, out=(vol, cnt))
        view = vol.permute(1, 2, 3, 0)
        before = C.amax_fallbacks
        y1 = _run(C, "f16x2", view, pk, None, 0, device)
        y1b = _run(C, "f16x2", view, pk, None, 0, device)
        assert C.amax_fallbacks == before + 1 and torch.equal(y1, y1b)                # tagged once, trusted while nothing wrote the storage
        ops.backproject_aggregate(feats * 2.0 ** 20, pts, proj, out=(vol, cnt))       # same Python objects, new contents 2^20 times larger
        y2 = _run(C, "f16x2", view, pk, None, 0, device)
        assert C.amax_fallbacks == before + 2, "the refill through the raw pointer left a stale max |x| slot in use"
        assert torch.isfinite(y2).all()
        torch.testing.assert_close(y2, y1 * 2.0 ** 20, rtol=1e-5, atol=0)


@pytest.mark.parametrize("cin,mid,cout,use_res", [(64, 64, 256, True), (128, 128, 512, True), (64, 64, 256, False)])
def test_f16x2_chained_bottleneck(device, cin, mid, cout, use_res):
    """conv2 -> conv3 of a ResNet bottleneck in one launch: the intermediate's scale is the workgroup's own maximum."""
    from nerfdet_amd import conv3d as C
    torch.manual_seed(8)
    c2 = nn.Conv2d(cin, mid, 3, 1, 1, bias=False); b2 = nn.BatchNorm2d(mid).eval()
    c3 = nn.Conv2d(mid, cout, 1, bias=False); b3 = nn.BatchNorm2d(cout).eval()
    with torch.no_grad():
        for b in (b2, b3):
            b.weight.uniform_(0.5, 1.5); b.bias.normal_(0, 0.2); b.running_mean.normal_(0, 0.2); b.running_var.uniform_(0.5, 1.5)
        x = torch.relu(torch.randn(2, 24, 32, cin)) * torch.exp(torch.randn(2, 24, 32, 1))
        r = torch.randn(2, 24, 32, cout) if use_res else None
        ref = b3.double()(c3.double()(torch.relu(b2.double()(c2.double()(x.permute(0, 3, 1, 2).double())))))
        if use_res:
            ref = ref + r.permute(0, 3, 1, 2).double()
        ref = torch.relu(ref).permute(0, 2, 3, 1)
        for m in (c2, b2, c3, b3):
            m.float().to(device)
        pk2, pk3 = C.packed([c2], b2), C.packed([c3], b3)
        err = {}
        for arith in ("bf16x3", "f16x2"):
            prev = C.set_arithmetic(arith)
            try:
                y = C.conv2d_chain_nhwc(x.to(device), pk2, pk3, residual=None if r is None else r.to(device), relu=1)
                torch.cuda.synchronize()
            finally:
                C.set_arithmetic(prev)
            err[arith] = _rel_rms(y, ref)
            if arith == "f16x2":
                assert C.amax_value(y._ndet_amax) == float(y.abs().max())
    assert err["f16x2"] <= 1.15 * err["bf16x3"] and err["f16x2"] < 1e-6, err


def test_f16x2_transposed_and_upsampled_residual(device):
    """The two epilogue forms the 3D neck's up path and the FPN's top-down path use (imvoxelnet.py:30-31; mmdet FPN behind nerfdet.py:141)."""
    from nerfdet_amd import conv3d as C
    torch.manual_seed(9)
    up = nn.ConvTranspose3d(256, 128, 2, 2, bias=False)
    x = torch.relu(torch.randn(5, 6, 4, 256))
    lat = nn.Conv2d(256, 256, 1)
    xf = torch.randn(3, 10, 12, 256)
    coarse = torch.randn(3, 5, 6, 256)
    with torch.no_grad():
        ref_up = up.double()(x.permute(3, 0, 1, 2).unsqueeze(0).double())[0].permute(1, 2, 3, 0)
        ref_lat = lat.double()(xf.permute(0, 3, 1, 2).double()) + torch.nn.functional.interpolate(coarse.permute(0, 3, 1, 2).double(), scale_factor=2, mode="nearest")
        ref_lat = ref_lat.permute(0, 2, 3, 1)
        up.float(); lat.float()
        pku, pkl = C.packed([up.to(device)]), C.packed([lat.to(device)])
        prev = C.set_arithmetic("f16x2")
        try:
            yu = C.conv3d_ndhwc(x.to(device), pku)
            yl = C.conv2d_nhwc(xf.to(device), pkl, residual=coarse.to(device), residual_up2=True)
            torch.cuda.synchronize()
        finally:
            C.set_arithmetic(prev)
    assert _rel_rms(yu, ref_up) < 1e-6 and _rel_rms(yl, ref_lat) < 1e-6
    assert C.amax_value(yu._ndet_amax) == float(yu.abs().max()) and C.amax_value(yl._ndet_amax) == float(yl.abs().max())


def test_detections_agree_between_the_two_fp32_class_arithmetics(device):
    """forward_test of a small detector in the fp16-pair and in the bf16x3 arithmetic: the same boxes (labels, order), scores to 1e-4."""
    import copy
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from nerfdet_amd import conv3d as C
    w = bench.WORKLOADS["tiny"]
    det = copy.deepcopy(bench.build_model(w)).to(device)
    batch = bench.to_device(bench.synth_batch(w, 0), device)
    out = {}
    for arith in ("bf16x3", "f16x2"):
        prev = C.set_arithmetic(arith)
        try:
            with torch.no_grad():
                out[arith] = det(return_loss=False, **batch)[0]
        finally:
            C.set_arithmetic(prev)
    a, b = out["bf16x3"], out["f16x2"]
    assert len(a["labels_3d"]) > 0
    assert torch.equal(a["labels_3d"], b["labels_3d"])
    torch.testing.assert_close(a["scores_3d"], b["scores_3d"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a["boxes_3d"].tensor, b["boxes_3d"].tensor, rtol=1e-4, atol=1e-4)


def test_range_guard_trips_exactly_when_the_floor_exceeds_the_tolerance(device):
    """csrc/conv_common.hpp::conv_guard_check: a fp16-pair launch compares 2^-39 max|in| guard_l1 with conv3d.GUARD_TOL on the device.  Ordinary
    activations leave the word clear; the same layer on an input whose maximum is 1e9 (the sigma-MLP's unseen-voxel rows, nerfdet.py:236-243)
    raises it -- and there the result really is outside the elementwise bar the guard protects, while bf16x3 is inside it."""
    from nerfdet_amd import conv3d as C
    torch.manual_seed(11)
    conv, bn = _layer(64, 128, 3, 1, 5)
    x = torch.relu(torch.randn(6, 10, 12, 64))
    xb = x.clone()
    xb[0, 0, :3] *= 1.0e9         # a few rows 1e9 times larger: every other row then lies 2^-30 below the tensor's maximum
    with torch.no_grad():
        ref = _reference(conv, bn, x, None, 1)
        refb = _reference(conv, bn, xb, None, 1)
    with torch.no_grad():
        pk = C.packed([conv.to(device)], bn.to(device))
        l1 = C.guard_l1(pk)
        w = pk["w"].abs().cpu() * pk["scale"].abs().cpu()[None, :, None]
        assert abs(l1 - float(w.sum(dim=(0, 2)).max())) <= 1e-4 * l1          # no weight of this layer lies 2^-16 below the weight maximum
        C.guard_begin(device)
        y = _run(C, "f16x2", x, pk, None, 1, device)
        assert not C.guard_tripped(device)
        assert _rel_rms(y, ref) < 1e-6
        C.guard_begin(device)
        yb = _run(C, "f16x2", xb, pk, None, 1, device)
        assert C.guard_tripped(device), "max|in| 1e9 x ||w||_1 x 2^-39 is far above the tolerance: the word must be set"
        far = torch.ones(6, 10, 12, dtype=torch.bool)
        far[:2, :2, :5] = False                                                       # outputs the huge rows do not reach (3x3x3 taps)
        err_f16 = float((yb.cpu().double() - refb)[far].abs().max())
        C.guard_begin(device)
        y3 = _run(C, "bf16x3", xb, pk, None, 1, device)
        assert not C.guard_tripped(device)                                            # the exact-operand arithmetic has no floor to report
        err_b3 = float((y3.cpu().double() - refb)[far].abs().max())
        assert err_b3 <= 1e-4 and err_f16 > 10 * err_b3, (err_f16, err_b3)            # what the guard is there to catch
        # the two conditions one at a time.  (1) the floor against the tolerance, on a tensor that does have a part far below its maximum (the second half of
        # the grid 1e-7 times the first): scaled so that the floor sits just below / just above the tolerance
        xd = x.clone()
        xd[3:] *= 1.0e-7
        amax = float(xd.abs().max())
        for factor, want in ((0.5, False), (2.0, True)):
            s = factor * C.GUARD_TOL * 2.0 ** 39 / (l1 * amax)
            C.guard_begin(device)
            _run(C, "f16x2", xd * s, pk, None, 1, device)
            assert C.guard_tripped(device) == want, (factor, s)
        # (2) a UNIFORMLY large tensor (a deep un-normalised network: activations of 1e7 everywhere) is inside the fp16-pair window everywhere: its
        # floor is far above the absolute tolerance and far below its own fp32-class error -- no reason to leave the fast arithmetic
        C.guard_begin(device)
        yu = _run(C, "f16x2", x * 1.0e7, pk, None, 1, device)
        assert not C.guard_tripped(device)
        assert _rel_rms(yu, _reference(conv.cpu(), bn.cpu(), x * 1.0e7, None, 1)) < 1e-6


def test_range_guard_of_the_chained_bottleneck(device):
    """The chained kernel checks its FIRST convolution against the input tensor (floor and tile minimum, as every launch); the chained 1x1 product
    scales its operand by the workgroup's own maximum -- already the granularity the guard looks at -- and needs no check of its own."""
    from nerfdet_amd import conv3d as C
    torch.manual_seed(12)
    c2 = nn.Conv2d(64, 64, 3, 1, 1, bias=False); b2 = nn.BatchNorm2d(64).eval()
    c3 = nn.Conv2d(64, 256, 1, bias=False); b3 = nn.BatchNorm2d(256).eval()
    x = torch.relu(torch.randn(2, 24, 32, 64))
    xb = x.clone()
    xb[0] *= 3.0e7                                      # one of the two maps 3e7 times brighter than the other
    with torch.no_grad():
        for m in (c2, b2, c3, b3):
            m.to(device)
        pk2, pk3 = C.packed([c2], b2), C.packed([c3], b3)
        prev = C.set_arithmetic("f16x2")
        try:
            C.guard_begin(device)
            C.conv2d_chain_nhwc(x.to(device), pk2, pk3, relu=1)
            assert not C.guard_tripped(device)
            C.guard_begin(device)
            C.conv2d_chain_nhwc((x * 3.0e7).to(device), pk2, pk3, relu=1)      # uniformly large: inside the window everywhere
            assert not C.guard_tripped(device)
            C.guard_begin(device)
            C.conv2d_chain_nhwc(xb.to(device), pk2, pk3, relu=1)
            assert C.guard_tripped(device)
        finally:
            C.set_arithmetic(prev)
