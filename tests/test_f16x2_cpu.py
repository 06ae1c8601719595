"""Host side and arithmetic of the fp16-pair convolution mode (nerfdet_amd/conv3d.py, csrc/conv_common.hpp) without a GPU: the scale choices,
the per-layer arithmetic rule, and a numpy emulation of the scheme itself (operand representation and the three-product sum) against fp64 --
the same modules as tests/test_f16x2_gpu.py (necks/imvoxelnet.py:22-67, imvoxel_head_v2.py:45-49, the backbone behind nerfdet.py:140)."""
import math
import struct

import numpy as np


def _bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def _from_bits(u):
    return struct.unpack("<f", struct.pack("<I", u & 0xffffffff))[0]


def xscale_of(amax):
    """conv_common.hpp::conv_xscale_of / conv_xinv_of, bit for bit."""
    eb = max((_bits(amax) >> 23) & 0xff, 30)
    return _from_bits((268 - eb) << 23), _from_bits((eb - 14) << 23)


def test_device_scale_puts_the_maximum_below_2_to_15():
    for amax in (1.0, 1.999999, 2.0, 3.0e4, 65504.0, 1e-6, 7.3e-20, 1e30, 3.4e38, 2.0 ** -96, 2.0 ** -97):
        s, inv = xscale_of(amax)
        assert s * inv == 1.0 and math.log2(s) == int(math.log2(s)), (amax, s)
        assert 2.0 ** 14 <= amax * s < 2.0 ** 15, (amax, s, amax * s)
    for amax in (0.0, 1e-45, 1e-40, 2.0 ** -100):            # zero, subnormal and tiny tensors: the scale stops at 2^111, nothing overflows
        s, inv = xscale_of(amax)
        assert s == 2.0 ** 111 and s * inv == 1.0 and amax * s < 2.0 ** 15


def test_weight_scale_and_layer_rule():
    from nerfdet_amd import conv3d as C
    for wmax in (0.37, 1.0, 0.024, 5e-7, 123.0, 2.0 ** -120):
        s = C.f16_weight_scale(wmax)
        assert math.log2(s) == int(math.log2(s))
        if wmax > 2.0 ** -97:
            assert 2.0 ** 14 <= wmax * s < 2.0 ** 15
    assert C.f16_weight_scale(0.0) == 1.0 and C.f16_weight_scale(float("inf")) == 1.0 and C.f16_weight_scale(float("nan")) == 1.0
    prev = C.set_arithmetic("f16x2")
    try:
        assert C.layer_arithmetic(C.F16_MIN_KSTEPS) == "f16x2" and C.layer_arithmetic(C.F16_MIN_KSTEPS - 1) == "bf16x3"
        assert C.train_arithmetic() == ("f16x2" if C.TRAIN_F16X2 else "bf16x3")      # training follows the inference arithmetic unless switched off
        keep, C.TRAIN_F16X2 = C.TRAIN_F16X2, False
        assert C.train_arithmetic() == "bf16x3"
        C.TRAIN_F16X2 = keep
        C.set_arithmetic("bf16")
        assert C.layer_arithmetic(1) == "bf16" and C.train_arithmetic() == "bf16"
    finally:
        C.set_arithmetic(prev)
    # the point MLPs are pinned (their inputs carry the reference's 1e9 rows, nerfdet.py:236-243)
    import torch
    pk = C.packed_linear(torch.nn.Linear(64, 32))
    assert pk["arith"] == "bf16x3"


def _split_f16(x, scale):
    xs = (x.astype(np.float64) * scale).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def _split_bf16x3(x):
    def rn(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
    a0 = rn(x); r = (x - a0).astype(np.float32); a1 = rn(r); a2 = rn((r - a1).astype(np.float32))
    return a0, a1, a2


def test_pair_represents_an_operand_to_2_to_minus_22():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.standard_normal(200000) * 3)).astype(np.float32)
    s, _ = xscale_of(float(np.abs(x).max()))
    hi, lo = _split_f16(x, s)
    err = np.abs((hi.astype(np.float64) + lo) / s - x)
    big = np.abs(x) >= np.abs(x).max() * 2.0 ** -16
    assert (err[big] <= 2.0 ** -22 * np.abs(x[big])).all()                      # two 11-bit halves (+ the sign of the low one)
    assert (err <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -25 / s)).all()      # below that: half of fp16's subnormal spacing under the scale (<= 2^-39 max |x|)


def _accumulate(planes, k):
    """fp32 accumulator rounded once per 16-wide MFMA step and product plane (the products inside a step summed exactly)."""
    m, n = planes[0][0].shape[0], planes[0][1].shape[1]
    acc = np.zeros((m, n), np.float32)
    for k0 in range(0, k, 16):
        for a, b in planes:
            acc = (acc.astype(np.float64) + a[:, k0:k0 + 16].astype(np.float64) @ b[k0:k0 + 16].astype(np.float64)).astype(np.float32)
    return acc


def test_three_product_sum_is_not_worse_than_the_six_product_one():
    """The emulated kernels against fp64 (what tests/test_f16x2_gpu.py measures on the matrix cores): the fp16-pair sum rounds its accumulator
    three times per step, the bf16x3 sum six times -- its error is the smaller one."""
    rng = np.random.default_rng(1)
    for k in (128, 576, 2304):
        x = (np.maximum(rng.standard_normal((96, k)), 0) * np.exp(rng.standard_normal((96, 1)))).astype(np.float32)
        w = (rng.standard_normal((k, 48)) / np.sqrt(k)).astype(np.float32)
        ref = x.astype(np.float64) @ w.astype(np.float64)
        nrm = np.sqrt((ref ** 2).mean())
        sx, ix = xscale_of(float(np.abs(x).max()))
        sw, iw = xscale_of(float(np.abs(w).max()))
        xh, xl = _split_f16(x, sx)
        wh, wl = _split_f16(w, sw)
        y16 = _accumulate([(xl, wh), (xh, wl), (xh, wh)], k) * np.float32(ix * iw)
        a, b = _split_bf16x3(x), _split_bf16x3(w)
        y3 = _accumulate([(a[i], b[j]) for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0))], k)
        e16 = np.sqrt(((y16 - ref) ** 2).mean()) / nrm
        e3 = np.sqrt(((y3 - ref) ** 2).mean()) / nrm
        assert e16 <= 1.05 * e3 and e16 < 1e-6, (k, e16, e3)
