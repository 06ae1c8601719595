"""Host side of the MFMA 3D convolution (csrc/conv3d_kernels.hip): weight / BatchNorm packing and the launcher.

Layout: activations channels-last ``(D,H,W,C)`` fp32; weights packed ``(taps, Cout, Cin)``.  Eval-mode BatchNorm3d is
folded into the epilogue as ``alpha = gamma/sqrt(var+eps)``, ``beta = bias - mean*alpha`` (the form ATen's CPU batch-norm
uses), followed by optional residual add and ReLU -- replacing nn.Conv3d/ConvTranspose3d + BatchNorm3d + ReLU of
mmdet3d/models/necks/imvoxelnet.py:36-67,233-260."""
from __future__ import annotations

import ctypes
import math
from ctypes import c_void_p
from typing import Optional, Sequence

import torch
from torch import nn

from . import _lib, trace
from ._lib import check
from .conv_tuning import TUNED, TUNED_BF16, TUNED_F16, TUNED_SPLIT
from ._lib import raw_stream

# Arithmetic of the convolution kernels: "bf16x3" = fp32 operands split exactly into three bf16 terms, six bf16-MFMA
# products accumulated in fp32 (csrc/conv_split_kernels.hip; fp32-level error, 16x the MFMA rate per product);
# "f32" = the fp32-input MFMA kernel (csrc/conv3d_kernels.hip; bit-exact FMA chains);
# "bf16" = the bf16x3 kernels issuing only the leading product: operands rounded to bf16, fp32 accumulate, fp32 activations in HBM
# (what bf16 autocast computes; BASELINE.json configs 3 and 5);
# "f16x2" = fp32 operands as fp16 PAIRS (hi + lo of the power-of-two pre-scaled tensor, 2 x 11 significand bits + sign), three fp16-MFMA products
# accumulated in fp32: half the matrix-core work of bf16x3 at the same or a smaller measured error against fp64 (three accumulator
# roundings per K step instead of six; tests/test_conv3d_gpu.py::test_f16x2_error_not_above_bf16x3).  The activation scale is derived on the
# device from the input's max |x|, which every convolution epilogue leaves behind for the next layer (the `_ndet_amax` attribute of its output
# tensor); layers with fewer than F16_MIN_KSTEPS K steps stay on bf16x3 (HBM-bound: nothing to gain); training follows it with
# every scale taken on the device (TRAIN_F16X2 below, conv_train.py).
ARITHMETIC = "f16x2"
F16_MIN_KSTEPS = 4
SPLIT_FAMILY = ("bf16x3", "bf16", "f16x2")    # the arithmetics of csrc/conv_split_kernels.hip


def set_arithmetic(mode: str) -> str:
    """Select the kernel family for every following convolution launch; returns the previous mode."""
    global ARITHMETIC
    if mode not in ("f32", "bf16x3", "bf16", "f16x2"):
        raise ValueError(f"unknown conv arithmetic {mode!r}")
    prev, ARITHMETIC = ARITHMETIC, mode
    return prev


TRAIN_F16X2 = True     # the training step's convolutions (forward, data and weight gradients: conv_train.py) on the fp16-pair arithmetic as well, every
                       # scale taken on the device from an amax slot (weights: one ndet_amax_f32 per tensor and step; dy: per gradient tensor).  False: the
                       # six-product bf16x3 kernels (exact operands; what the deterministic-trajectory tests of rounds 3-4 were pinned on)


def train_arithmetic() -> str:
    """What the training kernels (forward under autograd, data / weight gradients: nerfdet_amd/conv_train.py) compute in under the current mode."""
    if ARITHMETIC == "f16x2":
        return "f16x2" if TRAIN_F16X2 else "bf16x3"
    return ARITHMETIC


class _AmaxSlots:
    """Device slots (1 KiB: one sub-slot per XCD) for the tensors' max |x| (fp16-pair arithmetic).  A slot is handed out once and never reused: it
    has to be zero before the producing launch, and the zeros come from one fill per 4096 slots, stream-ordered before every launch that follows
    on that stream."""

    WIDTH = 256     # floats per slot = NDET_AMAX_SUB x NDET_AMAX_STRIDE of csrc/conv_common.hpp (8 sub-slots, one per XCD, 128 bytes apart); checked
                    # against the library at load time (_lib.load: ndet_amax_slot_floats)

    def __init__(self):
        self.pools = {}

    def take(self, device) -> torch.Tensor:
        key = (device, raw_stream(device))
        pool = self.pools.get(key)
        if pool is None or pool[1] + self.WIDTH > pool[0].numel():
            pool = self.pools[key] = [torch.zeros(4096 * self.WIDTH, dtype=torch.float32, device=device), 0]
        slot = pool[0][pool[1]:pool[1] + self.WIDTH]
        pool[1] += self.WIDTH
        return slot

    def take_many(self, device, n: int) -> torch.Tensor:
        """``n`` consecutive zeroed slots as one (n, WIDTH) tensor (row i is a slot)."""
        key = (device, raw_stream(device))
        pool = self.pools.get(key)
        if pool is None or pool[1] + n * self.WIDTH > pool[0].numel():
            pool = self.pools[key] = [torch.zeros(max(4096, n) * self.WIDTH, dtype=torch.float32, device=device), 0]
        block = pool[0][pool[1]:pool[1] + n * self.WIDTH].view(n, self.WIDTH)
        pool[1] += n * self.WIDTH
        return block

    def fresh(self, device):
        """Drop the current stream's pool: the next slot comes from a new zero fill (graph capture: the fill must be part of the graph)."""
        self.pools.pop((device, raw_stream(device)), None)


AMAX = _AmaxSlots()
# measurement only (set by tools/diag/amax_cost.sh through conv3d.measurement_mode, never by the environment of a production run): no
# epilogue commits a maximum, every fp16-pair launch takes a separate ndet_amax_f32 pass over its input instead
NO_AMAX_COMMIT = False
amax_fallbacks = 0      # how many inputs needed their own ndet_amax_f32 pass (diagnostic: the hot path should carry the attribute)


def measurement_mode(no_amax_commit: bool = False) -> None:
    """Explicit switch for the measurement builds of DESIGN.md 11.2 (was an environment variable: a stray NDET_NO_AMAX_COMMIT in a production shell
    changed the production path's launches)."""
    global NO_AMAX_COMMIT
    NO_AMAX_COMMIT = bool(no_amax_commit)


# ---- when is a tensor's slot still the maximum of its contents? ----
# Two kinds of writes can make it stale: torch's own in-place operations (seen through the tensor's version counter -- which INFERENCE tensors do
# not have: ``t._version`` raises under torch.inference_mode()) and this library's kernels writing through raw pointers into caller-owned
# buffers (``out=`` arguments: invisible to torch, counted here per storage).  A stale maximum that is too small overflows fp16 silently, so
# anything that cannot be proven fresh takes a new ndet_amax_f32 pass instead.
_RAW_WRITES = {}    # storage address -> library writes through raw pointers into caller-owned tensors


def _stamp(t: torch.Tensor):
    try:
        v = t._version
    except RuntimeError:        # inference tensor: no version counter
        v = None
    return v, _RAW_WRITES.get(t.untyped_storage().data_ptr(), 0)


def note_raw_write(t: torch.Tensor) -> None:
    """A kernel of this library wrote ``t`` through its raw pointer (an ``out=`` buffer the caller owns): every slot tagged on a tensor over the
    same storage is stale from here on."""
    key = t.untyped_storage().data_ptr()
    _RAW_WRITES[key] = _RAW_WRITES.get(key, 0) + 1
    t.__dict__.pop("_ndet_amax", None)


def _tag_amax(t: torch.Tensor, slot: torch.Tensor, produced: bool = True) -> None:
    """Attach the slot together with the tensor's write stamp.  ``produced``: the library itself just wrote ``t`` (or ``t`` is a view it made of
    such a tensor).  An inference tensor that the library did not produce is never tagged: torch could write it in place unseen."""
    stamp = _stamp(t)
    if stamp[0] is None and not produced:
        return
    t._ndet_amax = slot
    t._ndet_amax_stamp = stamp


def _amax_tag(t: torch.Tensor):
    """The tensor's slot if it is provably fresh, else None."""
    slot = getattr(t, "_ndet_amax", None)
    if slot is not None and getattr(t, "_ndet_amax_stamp", None) == _stamp(t):
        return slot
    return None


def carry_amax(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """``dst`` is a view / permutation / contiguous copy of ``src`` (same elements) made by this package: it inherits the max |x| slot."""
    slot = _amax_tag(src)
    if slot is not None:
        _tag_amax(dst, slot)
    return dst


def amax_value(slot: torch.Tensor) -> float:
    """Host read-back of a slot (tests / diagnostics; synchronises)."""
    return float(slot.view(-1, 32)[:, 0].max())


_slot_width_checked = False


def amax_of(x: torch.Tensor) -> torch.Tensor:
    """The device slot holding max |x| of ``x``: left by the kernel that wrote it, or computed here in one pass."""
    global amax_fallbacks, _slot_width_checked
    if not _slot_width_checked:
        assert _lib.load().ndet_amax_slot_floats() == _AmaxSlots.WIDTH, "amax slot width of the library differs from conv3d._AmaxSlots.WIDTH"
        _slot_width_checked = True
    slot = _amax_tag(x)
    if slot is None:
        amax_fallbacks += 1
        slot = AMAX.take(x.device)
        st = c_void_p(raw_stream(x.device))
        check(_lib.load().ndet_amax_f32(_ptr(x), x.numel(), _ptr(slot), st), "amax_f32")
        _tag_amax(x, slot, produced=False)
    return slot


# ---- range guard of the fp16-pair arithmetic (csrc/conv_common.hpp::conv_guard_check) ----
# Every fp16-pair launch compares its absolute error floor, 2^-39 max|in| * guard_l1, with GUARD_TOL on the device and raises the scene's guard
# word when it is exceeded AND the smallest workgroup-tile maximum recorded for its input lies below 2^-16 of max|in| (some part of the tensor really
# is outside the fp16-pair window; a uniformly large tensor -- ResNet-101 without calibrated statistics: activations of 1e6 everywhere -- is not).  The detector zeroes the word at the start of a scene (guard_begin), the word reaches the host with the scene's
# detections (head.simple_test_fused: header word 3 of the packed picks, no extra copy or sync), and a scene whose word is set is repeated on the
# six-product bf16x3 arithmetic, whose operands are exact (detector.simple_test).  Other callers read it with guard_tripped() (synchronises).
GUARD_ENABLED = True
GUARD_TOL = 2.0 ** -15        # absolute; north_star's bar is 1e-4 on O(1) voxel features: a third of it, for the floors of a few layers in a row
guard_trips = 0               # scenes repeated on bf16x3 because their guard word was set (diagnostic)
_GUARD_WORDS = {}


def guard_word(device) -> Optional[torch.Tensor]:
    """The guard word (int32, on the device) of the current stream, or None when the guard is off."""
    if not GUARD_ENABLED:
        return None
    key = (device, raw_stream(device))
    w = _GUARD_WORDS.get(key)
    if w is None:
        w = _GUARD_WORDS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def guard_begin(device) -> None:
    """Start of a scene: clear the current stream's guard word (stream-ordered before the scene's launches)."""
    w = guard_word(device)
    if w is not None:
        w.zero_()


def guard_tripped(device) -> bool:
    """Has a fp16-pair launch on the current stream exceeded the tolerance since guard_begin?  (synchronises; the detector does not use this)"""
    w = guard_word(device)
    return bool(w is not None and int(w.item()) & 1)


def guard_l1(pk: dict) -> float:
    """max over output channels j of |scale_j| (sum_k |w_jk| + max|w| #{k: 0 < |w_jk| < 2^-16 max|w|}): what the launch multiplies 2^-39 max|in|
    by to bound its absolute error floor (the second term: weights so far below the weight maximum that THEIR error is absolute).  Once per pack."""
    hit = pk.get("guard_l1")
    if hit is None:
        w = pk["w"].abs()
        wmax = w.max()
        tiny = ((w > 0) & (w < wmax * 2.0 ** -16)).sum(dim=(0, 2)).float()
        l1 = w.sum(dim=(0, 2)) + wmax * tiny
        if pk.get("scale") is not None:
            l1 = l1 * pk["scale"].abs()
        v = float(l1.max())
        hit = pk["guard_l1"] = v if math.isfinite(v) else float("inf")
    return hit


launch_hook = None  # bench.py: callable(flops, thunk, kernel_name) wrapping every MFMA-conv launch (event timing); None = direct

KERNEL_NAMES = {("bf16x3", 64): "k_conv_split<64,64,2,2>", ("bf16x3", 128): "k_conv_split<128,128,2,2>", ("bf16x3", 12864): "k_conv_split<128,64,2,2>",
                ("bf16x3", 128256): "k_conv_split_ws", ("bf16x3", 129256): "k_conv_split_wsp<128>", ("bf16x3", 129257): "k_conv_split_wsp<128,8>", ("bf16x3", 129064): "k_conv_split_wsp<64>", ("bf16x3", 3128): "k_conv_split_halo<4,2>", ("bf16x3", 3256): "k_conv_split_halo<8,2>", ("bf16x3", 3257): "k_conv_split_halo<4,4>", ("bf16x3", 3258): "k_conv_split_halo<4,4,p8>",
                ("f32", 64): "k_conv3d_igemm<64,64,2,2>", ("f32", 128): "k_conv3d_igemm<128,128,4,2>"}
KERNEL_NAMES.update({("bf16x3", 100000 + t if t != 12864 else 112864): KERNEL_NAMES[("bf16x3", t)] for t in (64, 128, 12864)})   # direct-epilogue forms
KERNEL_NAMES.update({("bf16", t): n for (a, t), n in list(KERNEL_NAMES.items()) if a == "bf16x3"})
KERNEL_NAMES.update({("f16x2", t): n + "/f16x2" for (a, t), n in list(KERNEL_NAMES.items()) if a == "bf16x3"})


def _launch(flops, thunk, arith="f32", tile=0, nbytes=0):
    """Every MFMA-convolution launch goes through here: ``nbytes`` = algorithmic traffic (input + weights + output + residual, each
    touched once), ``flops`` = algorithmic multiply-adds x 2."""
    name = KERNEL_NAMES.get((arith, tile), f"{arith}:{tile}")
    if launch_hook is not None:
        return launch_hook(flops, thunk, name)
    return trace.span(name, thunk, flops=flops, bytes=nbytes, kind="conv")


def choose_tiling(m: int, cout: int, k_iters: int, tile: int = 0, splits: int = 0, transposed: bool = False):
    """Tile edge (64 / 128) and split-K factor, from sweeps on MI355X (tools/tune_conv3d.py, tools/tune_conv2d.py):
    128x128 tiles once there are >= 100 of them and the K walk is long enough to amortise the larger epilogue;
    split K only while every split keeps >= 64 K-steps and the grid stays <= ~1200 (128) / ~2400 (64) workgroups --
    below that the partial-sum round trip costs more than the extra parallelism buys."""
    if tile == 0 and splits == 0:
        hit = TUNED.get((m, cout, k_iters, int(transposed)))
        if hit is not None:
            return hit
    if tile == 0:
        big = ((m + 127) // 128) * ((cout + 127) // 128)
        tile = 128 if (big >= 120 and cout >= 128 and k_iters > 4) else 64
    if splits == 0:
        if transposed:
            return tile, 1
        tiles = ((m + tile - 1) // tile) * ((cout + tile - 1) // tile)
        cap = 1200 if tile == 128 else 2400
        splits = 1
        while splits < 8 and k_iters // (splits + 1) >= 64 and tiles * (splits + 1) <= cap:
            splits += 1
    return tile, splits


def choose_tiling_split(m: int, cout: int, k_iters: int, tile: int = 0, splits: int = 0, transposed: bool = False, halo_ok: bool = False):
    """Tile (64 = 64x64, 128 = 128x128, 12864 = 128 rows x 64 channels, 128256 = wave-specialised 128 rows x 256 channels, 129256 / 129064 = its
    persistent form with 128- / 64-row tiles, 3128 / 3256 / 3257 / 3258 = halo-stationary 128-voxel patch x 128 / 256 channels) and split-K factor for the
    bf16x3 kernel."""
    if tile == 0 and splits in (0, 1):     # splits == 1: the caller cannot split K (transposed, upsampled residual): the table's tile, unsplit
        key = (m, cout, k_iters, int(transposed))
        hit = (TUNED_BF16.get(key) if ARITHMETIC == "bf16" else TUNED_F16.get(key) if ARITHMETIC == "f16x2" else None) or TUNED_SPLIT.get(key)
        if hit is not None:
            return hit if splits == 0 else (hit[0], 1)
    if tile == 0:   # shapes outside the measured table: the pattern the sweeps showed
        mt = (m + 127) // 128
        big = mt * ((cout + 127) // 128)
        wide = mt * ((cout + 255) // 256)                        # 128 x 256 tiles
        if k_iters <= 4 and not transposed:                      # 64- / 128-channel 1x1 layers are HBM-bound: the small tile keeps the most loads in flight
            tile = 64                                            # (same answer in the cfg1, cfg2 and cfg5 sweeps)
        elif cout > 128 and wide * max(1, min(32, k_iters // 24)) >= 192:
            tile = 3256 if (halo_ok and mt >= 64) else 128256    # halo-stationary when the layer has taps to share
        elif big >= 120 and cout > 64:
            tile = 128
        elif cout <= 64 and mt >= 256:
            tile = 12864
        else:
            tile = 64
    if splits == 0:
        if transposed or tile in (100064, 100128, 112864):     # (the direct-epilogue forms of the unified tiles write final values: no split-K)
            return tile, 1
        tm, tn = {12864: (128, 64), 128256: (128, 256), 129256: (128, 256), 129257: (128, 256), 129064: (64, 256), 3128: (128, 128), 3256: (128, 256), 3257: (128, 256), 3258: (128, 256)}.get(tile, (tile, tile))
        tiles = ((m + tm - 1) // tm) * ((cout + tn - 1) // tn)
        splits = 1
        while splits < 32 and k_iters // (splits + 1) >= 24 and tiles * (splits + 1) <= 768:
            splits += 1
    return tile, splits


DIRECT_EPILOGUE = True     # unified bf16x3 tiles store straight from the accumulators' layout whenever they write final values


def split_planes(pk: dict) -> torch.Tensor:
    """The packed weight (taps, Cout, Cin) as three bf16 planes tiled per 32-channel K step, (taps, Cin/32, 3, Cout, 32), with
    w = p0 + p1 + p2 exactly (built once per pack)."""
    planes = pk.get("w_split")
    if planes is None:
        w = pk["w"]
        taps, cout, cin = w.shape
        if cin % 32:
            raise ValueError(f"conv_ndhwc_split: Cin={cin} must be a multiple of 32")
        planes = torch.empty((taps, cin // 32, 3, cout, 32), dtype=torch.int16, device=w.device)
        st = c_void_p(raw_stream(w.device))
        check(_lib.load().ndet_split_weights_bf16x3(_ptr(w), taps, cout, cin, _ptr(planes), st), "split_weights_bf16x3")
        pk["w_split"] = planes
    return planes


def f16_weight_scale(wmax: float) -> float:
    """The power of two that puts a weight tensor's largest magnitude in [2^14, 2^15) (below fp16's 65504, the low halves of everything within
    2^-16 of it still normal); 1 for an all-zero / non-finite tensor, capped for denormal-sized weights."""
    if not (wmax > 0.0 and math.isfinite(wmax)):
        return 1.0
    e = max(math.frexp(wmax)[1], -96)          # wmax = m 2^e, m in [1/2, 1)
    return math.ldexp(1.0, 15 - e)


def split_planes_f16(pk: dict):
    """(planes, 1 / scale): the packed weight times a power of two that puts max |w| in [2^14, 2^15), as two fp16 planes tiled per 32-channel K
    step, (taps, Cin/32, 2, Cout, 32) (built once per pack; reading max |w| back is the one host synchronisation, at pack time)."""
    hit = pk.get("w_f16")
    if hit is None:
        w = pk["w"]
        taps, cout, cin = w.shape
        if cin % 32:
            raise ValueError(f"conv_ndhwc_arith: Cin={cin} must be a multiple of 32")
        scale = f16_weight_scale(float(w.abs().max()))
        planes = torch.empty((taps, cin // 32, 2, cout, 32), dtype=torch.int16, device=w.device)
        st = c_void_p(raw_stream(w.device))
        check(_lib.load().ndet_split_weights_f16x2(_ptr(w), taps, cout, cin, scale, _ptr(planes), st), "split_weights_f16x2")
        hit = pk["w_f16"] = (planes, 1.0 / scale)
    return hit


def layer_arithmetic(k_iters: int) -> str:
    """The arithmetic one split-family launch runs in under the current mode."""
    if ARITHMETIC == "f16x2" and k_iters < F16_MIN_KSTEPS:
        return "bf16x3"
    return ARITHMETIC


def _conv_split(x, pk, out, dims, kernel, stride, pad, transposed, residual, residual_up2, relu, splits, tile, m, k_iters, flops, want_amax=True, chain=None):
    halo_ok = (not transposed and all(s == 1 for s in stride) and all(k % 2 == 1 and q == k // 2 for k, q in zip(kernel, pad))
               and kernel[0] * kernel[1] * kernel[2] > 1)
    tile, splits = choose_tiling_split(m, pk["cout"], k_iters, tile, 1 if (transposed or residual_up2) else splits, transposed, halo_ok)
    if tile in (3128, 3256, 3257, 3258):   # halo-stationary tiles: stride-1 same-padded multi-tap convolutions only, K split over the channel chunks
        if not halo_ok:
            tile = 128256 if pk["cout"] > 128 else 128
        else:
            splits = min(splits, pk["cin"] // 32)
    if tile in (129256, 129257, 129064) and (transposed or pk["cout"] % 16 or kernel[0] * kernel[1] * kernel[2] > 32):
        tile = 128256                                        # the persistent form takes plain convolutions with Cout % 16 == 0
    # unified tiles that write final values: the epilogue straight from the MFMA's C layout (no LDS staging, no barriers)
    base = {100064: 64, 100128: 128, 112864: 12864}.get(tile, tile)
    direct_ok = (splits == 1 and not transposed and pk["cout"] % 32 == 0 and (m + 128) * pk["cout"] * 4 < (1 << 32)
                 and (not residual_up2 or _lib.load().ndet_version() >= 105))     # (the upsampled residual in the direct epilogue: ABI 105)
    tile = {64: 100064, 128: 100128, 12864: 112864}[base] if (base in (64, 128, 12864) and direct_ok and DIRECT_EPILOGUE) else base
    ws = torch.empty((m * pk["cout"] * splits * 4,), dtype=torch.uint8, device=x.device) if splits > 1 else None
    i3 = lambda v: (ctypes.c_int * 3)(*v)
    st = c_void_p(raw_stream(x.device))
    lib = _lib.load()
    d, h, w = dims
    nbytes = 4 * (x.numel() + pk["w"].numel() + out.numel() + (0 if residual is None else residual.numel()))
    arith = layer_arithmetic(k_iters)
    if ARITHMETIC == "f16x2" and pk.get("arith"):
        arith = pk["arith"]            # pinned packs: training (conv_train.py: the weights change every step) and the point MLPs (packed_linear)
        want_amax = want_amax and bool(pk.get("keep_amax"))    # (otherwise a reader in the fp16-pair arithmetic takes its own pass, amax_of)
    if NO_AMAX_COMMIT:
        want_amax = False
    if ARITHMETIC != "f16x2":
        planes = split_planes(pk)
        fn = lib.ndet_conv_ndhwc_bf16 if arith == "bf16" else lib.ndet_conv_ndhwc_split
        _launch(flops, lambda: check(fn(_ptr(x), _ptr(planes), _ptr(out), d, h, w, pk["cin"], pk["cout"], i3(kernel), i3(stride), i3(pad), int(transposed),
                                        _ptr(pk["scale"]), _ptr(pk["shift"]), _ptr(residual), int(residual_up2), relu, splits, tile, _ptr(ws), st),
                                     "conv_ndhwc_split"), arith, tile, nbytes)
        return out
    # fp16-pair mode: every launch leaves max |out| behind; the fp16-pair launches read their input's
    if arith == "f16x2" and pk.get("w_amax") is not None:
        # training packs (conv_train.py): planes scaled on the device by the slot pk["w_amax"]; plain convolutions only
        assert not transposed and not residual_up2
        planes, in_amax = pk["w_f16"][0], amax_of(x)
        keep = bool(pk.get("keep_partials"))       # weight-gradient GEMMs: a split-K launch leaves its partials for ndet_wgrad_to_torch (no reduction pass)
        want_amax = want_amax and not keep
        out_amax = AMAX.take(x.device) if want_amax else None
        gw = guard_word(x.device) if pk.get("guard", True) else None
        _launch(flops, lambda: check(lib.ndet_conv_ndhwc_train(_ptr(x), _ptr(planes), _ptr(out), d, h, w, pk["cin"], pk["cout"], i3(kernel), i3(stride), i3(pad),
                                                               _ptr(pk["scale"]), _ptr(pk["shift"]), _ptr(residual), relu, splits, tile, _ptr(in_amax),
                                                               _ptr(pk["w_amax"]), _ptr(out_amax), _ptr(ws), float(kernel[0] * kernel[1] * kernel[2] * pk["cin"]),
                                                               GUARD_TOL, _ptr(gw), int(keep), st),
                                     "conv_ndhwc_train"), arith, tile, nbytes)
        if keep:
            pk["_partials"] = (ws.view(torch.float32), splits) if splits > 1 else None     # (splits == 1: the launch wrote `out` as usual)
        if want_amax:
            _tag_amax(out, out_amax)
        return out
    if arith == "f16x2":
        planes, winv = split_planes_f16(pk)
        in_amax = amax_of(x)
    else:
        planes, winv, in_amax = split_planes(pk), 1.0, None
    out_amax = AMAX.take(x.device) if want_amax else None
    gw = guard_word(x.device) if arith == "f16x2" else None
    gl1 = guard_l1(pk) if gw is not None else 0.0
    if chain is not None:
        # the chained 32-channel projection of the output rows in the same launch (csrc: conv_map_rows): the 256-column halo tiles only
        mapped = None
        if tile in (3256, 3257, 3258) and splits == 1 and residual is None and relu == 0 and not transposed and pk["cout"] == 256 and arith in ("f16x2", "bf16x3"):
            mapped = torch.empty((m, 32), dtype=torch.float32, device=x.device)
            _launch(flops, lambda: check(lib.ndet_conv_ndhwc_mapped(_ptr(x), _ptr(planes), _ptr(out), d, h, w, pk["cin"], pk["cout"], i3(kernel), i3(stride), i3(pad),
                                                                    _ptr(pk["scale"]), _ptr(pk["shift"]), tile, 1 if arith == "f16x2" else 0, _ptr(in_amax), winv,
                                                                    _ptr(out_amax), gl1, GUARD_TOL, _ptr(gw), _ptr(chain[0]), _ptr(chain[1]), _ptr(mapped), st),
                                         "conv_ndhwc_mapped"), arith, tile, nbytes + 4 * mapped.numel())
            if want_amax:
                _tag_amax(out, out_amax)
            return out, mapped
        chain = None      # another tile / arithmetic took the layer: the caller projects in a launch of its own
    _launch(flops, lambda: check(lib.ndet_conv_ndhwc_guarded(_ptr(x), _ptr(planes), _ptr(out), d, h, w, pk["cin"], pk["cout"], i3(kernel), i3(stride), i3(pad),
                                                             int(transposed), _ptr(pk["scale"]), _ptr(pk["shift"]), _ptr(residual), int(residual_up2), relu,
                                                             splits, tile, 1 if arith == "f16x2" else 0, _ptr(in_amax), winv, _ptr(out_amax), _ptr(ws),
                                                             gl1, GUARD_TOL, _ptr(gw), st),
                                 "conv_ndhwc_guarded"), arith, tile, nbytes)
    if want_amax:
        _tag_amax(out, out_amax)
    return out


def projection_ok() -> bool:
    """Does the current arithmetic have the chained-projection epilogue (conv2d_nhwc(..., chain=...))?"""
    return ARITHMETIC == "f16x2"


def _ptr(t):
    return c_void_p(0 if t is None else t.data_ptr())


def pack_weight(w: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """Conv3d weight (Cout,Cin,k,k,k) -> (k^3, Cout, Cin); Conv2d weight (Cout,Cin,kh,kw) -> (kh*kw, Cout, Cin);
    ConvTranspose3d weight (Cin,Cout,2,2,2) -> (8, Cout, Cin)."""
    if w.dim() == 4:
        cout, cin = w.shape[:2]
        return w.permute(2, 3, 0, 1).reshape(-1, cout, cin).contiguous().float()
    if transposed:
        cin, cout = w.shape[:2]
        return w.permute(2, 3, 4, 1, 0).reshape(-1, cout, cin).contiguous().float()
    cout, cin = w.shape[:2]
    return w.permute(2, 3, 4, 0, 1).reshape(-1, cout, cin).contiguous().float()


def fold_bn(bn, bias: Optional[torch.Tensor], cout: int, device):
    """(scale, shift) of the fused epilogue, or (None, None) when there is nothing to apply."""
    if bn is None and bias is None:
        return None, None
    if bn is None:
        return torch.ones(cout, device=device), bias.detach().float().contiguous()
    alpha = bn.weight.detach().float() * (1.0 / torch.sqrt(bn.running_var.float() + bn.eps))
    beta = bn.bias.detach().float() - bn.running_mean.float() * alpha
    if bias is not None:
        beta = beta + bias.detach().float() * alpha
    return alpha.contiguous(), beta.contiguous()


def bn_affine(bn: nn.Module):
    """(scale, shift) of an eval-mode BatchNorm, cached on the module until one of its tensors changes."""
    store = bn.__dict__.setdefault("_ndet_packed", {})
    stamp = tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
    hit = store.get("affine")
    if hit is None or hit[0] != stamp:
        hit = (stamp, fold_bn(bn, None, bn.num_features, bn.weight.device))
        store["affine"] = hit
    return hit[1]


def packed(convs: Sequence[nn.Module], bn: Optional[nn.BatchNorm3d] = None):
    """Pack (and cache) one conv, or several convs sharing an input concatenated along Cout.  The cache entry is
    rebuilt when any parameter was updated in place (optimizer step, load_state_dict)."""
    tensors = [t for c in convs for t in (c.weight, c.bias) if t is not None]
    if bn is not None:
        tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    # the cache lives ON the first module (dies with it: ids and data pointers of freed modules get reused)
    store = convs[0].__dict__.setdefault("_ndet_packed", {})
    key = tuple(id(c) for c in convs[1:]) + (id(bn),)
    stamp = tuple((t.data_ptr(), t._version) for t in tensors)
    hit = store.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    tr = isinstance(convs[0], nn.ConvTranspose3d)
    w = torch.cat([pack_weight(c.weight.detach(), tr) for c in convs], dim=1).contiguous()
    cout = w.shape[1]
    bias = None
    if any(c.bias is not None for c in convs):
        bias = torch.cat([c.bias.detach() if c.bias is not None else torch.zeros(c.weight.shape[1 if tr else 0], device=w.device)
                          for c in convs])
    scale, shift = fold_bn(bn, bias, cout, w.device)
    c0 = convs[0]
    val = dict(w=w, scale=scale, shift=shift, cout=cout, cin=w.shape[2], ksize=c0.kernel_size[0], stride=c0.stride[0], transposed=tr,
               kernel=tuple(c0.kernel_size), strides=tuple(c0.stride), pads=tuple(c0.padding), ndim=len(c0.kernel_size))
    store[key] = (stamp, val, convs[1:], bn)  # keep the partner modules alive so their ids stay unique
    return val


def packed_linear(lin: nn.Linear, pad_in_to: int = 0):
    """nn.Linear as a 1x1 convolution for the MFMA kernel: weight (Cout, Cin) -> (1, Cout, Cin_padded), bias in the epilogue shift."""
    store = lin.__dict__.setdefault("_ndet_packed", {})
    stamp = (lin.weight.data_ptr(), lin.weight._version, None if lin.bias is None else (lin.bias.data_ptr(), lin.bias._version), pad_in_to)
    hit = store.get("linear")
    if hit is not None and hit[0] == stamp:
        return hit[1]
    w = lin.weight.detach().float()
    cout, cin = w.shape
    width = max(cin, pad_in_to)
    assert width % 32 == 0, f"input width {width} must be a multiple of 32"
    if width > cin:
        w = torch.cat([w, w.new_zeros(cout, width - cin)], dim=1)
    scale = torch.ones(cout, device=w.device) if lin.bias is not None else None
    shift = lin.bias.detach().float().contiguous() if lin.bias is not None else None
    # arith: the point MLPs stay on bf16x3 under the fp16-pair mode.  Their inputs hold the reference's own garbage rows -- nerfdet.py:236-243 divides
    # by (count + 1e-8), so a voxel no view sees carries ~1e9 where the seen ones carry O(1) -- and the fp16-pair scheme is exact only to
    # 2^-40 of the TENSOR's maximum per element (its scale is per tensor): rows 2^-30 below the maximum would keep 10 bits.
    val = dict(w=w.unsqueeze(0).contiguous(), scale=scale, shift=shift, cout=cout, cin=width, ksize=1, stride=1, transposed=False,
               kernel=(1, 1), strides=(1, 1), pads=(0, 0), ndim=2, arith="bf16x3")
    store["linear"] = (stamp, val)
    return val


def linear_rows(x: torch.Tensor, pk: dict, relu: int = 0) -> torch.Tensor:
    """(M, Cin) rows -> (M, Cout): a Linear (+ ReLU) as one launch of the MFMA kernel (bias and ReLU in the epilogue)."""
    assert x.dim() == 2 and x.is_contiguous()
    y = conv2d_nhwc(carry_amax(x, x.view(1, 1, x.shape[0], x.shape[1])), pk, relu=relu)
    return carry_amax(y, y.view(x.shape[0], -1))


def conv3d_ndhwc(x: torch.Tensor, pk: dict, residual: Optional[torch.Tensor] = None, relu: int = 0, splits: int = 0, tile: int = 0,
                 amax: bool = True):
    """x (D,H,W,Cin) contiguous fp32 on the GPU -> (OD,OH,OW,Cout).  relu: 0 none, 1 after the residual add, 2 before it.  amax (fp16-pair
    mode): leave max |out| behind for a following convolution -- False for outputs no convolution reads (identity branches, final heads)."""
    if not x.is_cuda:
        raise RuntimeError("nerfdet_amd.conv3d: tensors must live on the GPU (no CPU fallback)")
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    d, h, w, cin = x.shape
    assert cin == pk["cin"], (cin, pk["cin"])
    k, s, tr, cout = pk["ksize"], pk["stride"], pk["transposed"], pk["cout"]
    if tr:
        od, oh, ow = 2 * d, 2 * h, 2 * w
    else:
        pad = int(pk["pads"][0]) if "pads" in pk else k // 2
        assert pad == k // 2 or ARITHMETIC in SPLIT_FAMILY, "the fp32-MFMA family pads by k // 2"
        od, oh, ow = ((v + 2 * pad - k) // s + 1 for v in (d, h, w))
    out = torch.empty((od, oh, ow, cout), dtype=torch.float32, device=x.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous()
    lib = _lib.load()
    m = d * h * w if tr else od * oh * ow
    flops = 2 * m * cout * cin * (1 if tr else k ** 3) * (8 if tr else 1)
    if ARITHMETIC in SPLIT_FAMILY:
        kk, ss, pp = ((2, 2, 2), (2, 2, 2), (0, 0, 0)) if tr else ((k,) * 3, (s,) * 3, (pad,) * 3)
        return _conv_split(x, pk, out, (d, h, w), kk, ss, pp, tr, residual, False, relu, splits, tile, m,
                           (cin // 32) * (1 if tr else k ** 3), flops, amax)
    if tr:
        tile, splits = (tile or choose_tiling(m, cout, cin // 32, 0, 0, True)[0]), 1
    else:
        tile, splits = choose_tiling(m, cout, k ** 3 * (cin // 32), tile, splits)
    ws = None
    if splits > 1:
        ws = torch.empty((int(lib.ndet_conv3d_workspace_bytes(d, h, w, cin, cout, k, s, splits)),), dtype=torch.uint8, device=x.device)
    st = c_void_p(raw_stream(x.device))
    _launch(flops, lambda: check(lib.ndet_conv3d_ndhwc(_ptr(x), _ptr(pk["w"]), _ptr(out), d, h, w, cin, cout, k, s, int(tr), _ptr(pk["scale"]),
                                                       _ptr(pk["shift"]), _ptr(residual), relu, splits, tile, _ptr(ws), st), "conv3d_ndhwc"), "f32", tile,
            4 * (x.numel() + pk["w"].numel() + out.numel() + (0 if residual is None else residual.numel())))
    return out


def conv2d_nhwc(x: torch.Tensor, pk: dict, residual: Optional[torch.Tensor] = None, relu: int = 0, splits: int = 0, tile: int = 0,
                residual_up2: bool = False, amax: bool = True, chain=None):
    """Batch of 2D maps, x (N,H,W,Cin) contiguous fp32 -> (N,OH,OW,Cout): Conv2d (+ eval BatchNorm2d / bias) + ReLU +
    residual in one pass of the MFMA kernel (the batch is the kernel's depth axis with extent-1 taps).

    ``chain`` = (map_w (Cout, 32), map_b (32)) (fp16-pair mode, :func:`projection_ok`): returns ``(out, mapped)`` with ``mapped`` (N*OH*OW, 32) the
    projection of every output row computed in the same launch -- or ``(out, None)`` when the layer's tile cannot take it."""
    if not x.is_cuda:
        raise RuntimeError("nerfdet_amd.conv3d: tensors must live on the GPU (no CPU fallback)")
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32 and pk["ndim"] == 2
    n, h, w, cin = x.shape
    assert cin == pk["cin"], (cin, pk["cin"])
    (kh, kw), (sh, sw), (ph, pw), cout = pk["kernel"], pk["strides"], pk["pads"], pk["cout"]
    oh, ow = (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
    out = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    if residual is not None:
        want = (n, (oh + 1) // 2, (ow + 1) // 2, cout) if residual_up2 else tuple(out.shape)
        assert tuple(residual.shape) == want and residual.is_contiguous(), (tuple(residual.shape), want)
        if residual_up2:
            splits = 1
    m = n * oh * ow
    pair = chain is not None                       # the caller unpacks (out, mapped) whatever path the layer takes
    if ARITHMETIC in SPLIT_FAMILY:
        if chain is not None and not projection_ok():
            chain = None
        r = _conv_split(x, pk, out, (n, h, w), (1, kh, kw), (1, sh, sw), (0, ph, pw), False, residual, residual_up2, relu, splits, tile, m,
                        kh * kw * (cin // 32), 2 * m * cout * cin * kh * kw, amax, chain)
        return r if (not pair or isinstance(r, tuple)) else (r, None)
    tile, splits = choose_tiling(m, cout, kh * kw * (cin // 32), tile, splits)
    ws = torch.empty((m * cout * splits * 4,), dtype=torch.uint8, device=x.device) if splits > 1 else None
    i3 = lambda a, b, c: (ctypes.c_int * 3)(a, b, c)
    st = c_void_p(raw_stream(x.device))
    lib = _lib.load()
    _launch(2 * m * cout * cin * kh * kw,
            lambda: check(lib.ndet_conv_ndhwc(_ptr(x), _ptr(pk["w"]), _ptr(out), n, h, w, cin, cout, i3(1, kh, kw), i3(1, sh, sw), i3(0, ph, pw),
                                              _ptr(pk["scale"]), _ptr(pk["shift"]), _ptr(residual), int(residual_up2), relu, splits, tile, _ptr(ws), st),
                          "conv2d_nhwc"), "f32", tile, 4 * (x.numel() + pk["w"].numel() + out.numel() + (0 if residual is None else residual.numel())))
    return (out, None) if pair else out


CHAIN_BOTTLENECKS = True    # conv2 -> conv3 of the 64- / 128-channel bottlenecks in one launch (k_conv_split_chain)


def chain_ok(pk: dict, pk3: dict) -> bool:
    """The chained kernel holds ALL output channels of the first convolution in one 128 x 64 / 128 x 128 tile and multiplies them by a
    1x1 layer: a 2D convolution to 64 / 128 channels followed by a stride-1 1x1 layer to a multiple of 64, in the bf16 family."""
    return (CHAIN_BOTTLENECKS and ARITHMETIC in SPLIT_FAMILY and pk["ndim"] == 2 and pk3["ndim"] == 2 and pk["cout"] in (64, 128)
            and not pk["transposed"] and tuple(pk3["kernel"]) == (1, 1) and tuple(pk3["strides"]) == (1, 1) and pk3["cin"] == pk["cout"]
            and pk3["cout"] % 64 == 0 and pk["cin"] % 32 == 0 and pk["scale"] is not None and pk3["scale"] is not None)


def conv2d_chain_nhwc(x: torch.Tensor, pk: dict, pk3: dict, residual: Optional[torch.Tensor] = None, relu: int = 1) -> torch.Tensor:
    """relu_mode(bn3(conv1x1(relu(bn(conv(x))))) + residual) in ONE launch (csrc/conv_split_kernels.hip::k_conv_split_chain): conv2 -> conv3 of a
    ResNet bottleneck without the intermediate's round trip through HBM.  x (N,H,W,Cin) contiguous fp32 -> (N,OH,OW,pk3 cout)."""
    if not x.is_cuda:
        raise RuntimeError("nerfdet_amd.conv3d: tensors must live on the GPU (no CPU fallback)")
    assert chain_ok(pk, pk3) and x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    n, h, w, cin = x.shape
    assert cin == pk["cin"], (cin, pk["cin"])
    (kh, kw), (sh, sw), (ph, pw), mid, cout = pk["kernel"], pk["strides"], pk["pads"], pk["cout"], pk3["cout"]
    oh, ow = (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
    out = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    if residual is not None:
        assert tuple(residual.shape) == tuple(out.shape) and residual.is_contiguous(), (tuple(residual.shape), tuple(out.shape))
    m = n * oh * ow
    i3 = lambda a, b, c: (ctypes.c_int * 3)(a, b, c)
    st = c_void_p(raw_stream(x.device))
    lib = _lib.load()
    flops = 2 * m * mid * (cin * kh * kw + cout)
    nbytes = 4 * (x.numel() + pk["w"].numel() + pk3["w"].numel() + out.numel() + (0 if residual is None else residual.numel()))
    name = f"k_conv_split_chain<{mid}>"
    if ARITHMETIC == "f16x2":
        (p1, w1inv), (p3, w3inv) = split_planes_f16(pk), split_planes_f16(pk3)
        in_amax, out_amax = amax_of(x), (None if NO_AMAX_COMMIT else AMAX.take(x.device))
        name += "/f16x2"
        gw = guard_word(x.device)
        gl1, gl3 = (guard_l1(pk), guard_l1(pk3)) if gw is not None else (0.0, 0.0)
        thunk = lambda: check(lib.ndet_conv_chain_guarded(_ptr(x), _ptr(p1), n, h, w, cin, mid, i3(1, kh, kw), i3(1, sh, sw), i3(0, ph, pw), _ptr(pk["scale"]),
                                                          _ptr(pk["shift"]), _ptr(p3), cout, _ptr(pk3["scale"]), _ptr(pk3["shift"]), _ptr(residual), relu,
                                                          _ptr(out), 1, _ptr(in_amax), w1inv, w3inv, _ptr(out_amax), gl1, gl3, GUARD_TOL, _ptr(gw), st),
                              "conv_chain_guarded")
        if out_amax is not None:
            _tag_amax(out, out_amax)
    else:
        p1, p3 = split_planes(pk), split_planes(pk3)
        thunk = lambda: check(lib.ndet_conv_chain_split(_ptr(x), _ptr(p1), n, h, w, cin, mid, i3(1, kh, kw), i3(1, sh, sw), i3(0, ph, pw), _ptr(pk["scale"]),
                                                        _ptr(pk["shift"]), _ptr(p3), cout, _ptr(pk3["scale"]), _ptr(pk3["shift"]), _ptr(residual), relu,
                                                        _ptr(out), 0 if ARITHMETIC == "bf16" else 2, st), "conv_chain_split")
    if launch_hook is not None:
        launch_hook(flops, thunk, name)
    else:
        trace.span(name, thunk, flops=flops, bytes=nbytes, kind="conv")
    return out


FUSE_BOTTLENECKS = True     # the whole stage-1 bottleneck (conv1 -> conv2 -> conv3 + identity / downsample) in one launch (k_bottleneck_f16x2)


def bottleneck_ok(x: torch.Tensor, pk1: dict, pk2: dict, pk3: dict, pkd: Optional[dict]) -> bool:
    """Shapes csrc/bottleneck_kernels.hip takes: fp16-pair arithmetic, a 64-channel 3x3 stride-1 middle convolution between two 1x1 layers, frozen
    BatchNorm on all of them, identity residual (Cin == Cout) or a stride-1 1x1 downsample of a 64-channel input."""
    def one(pk):
        return pk["ndim"] == 2 and tuple(pk["kernel"]) == (1, 1) and tuple(pk["strides"]) == (1, 1) and not pk["transposed"] and pk["scale"] is not None
    if not (FUSE_BOTTLENECKS and ARITHMETIC == "f16x2" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        return False
    if not (one(pk1) and one(pk3) and pk2["ndim"] == 2 and tuple(pk2["kernel"]) == (3, 3) and tuple(pk2["strides"]) == (1, 1) and tuple(pk2["pads"]) == (1, 1)
            and pk2["scale"] is not None and pk1["cout"] == 64 and pk2["cin"] == 64 and pk2["cout"] == 64 and pk3["cin"] == 64 and pk3["cout"] % 32 == 0
            and pk1["cin"] % 32 == 0 and pk1["cin"] == x.shape[3]):
        return False
    if pkd is not None:
        if not (one(pkd) and pkd["cin"] == 64 and pk1["cin"] == 64 and pkd["cout"] == pk3["cout"]):
            return False
    elif pk1["cin"] != pk3["cout"]:
        return False
    return x.numel() // x.shape[3] * max(pk1["cin"], pk3["cout"]) * 4 < 0xfffffff0


def conv2d_bottleneck_nhwc(x: torch.Tensor, pk1: dict, pk2: dict, pk3: dict, pkd: Optional[dict] = None) -> torch.Tensor:
    """relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + identity) of a stage-1 ResNet bottleneck in ONE launch (csrc/bottleneck_kernels.hip);
    identity = x, or bnD(convD(x)) with ``pkd``.  x (N,H,W,Cin) contiguous fp32 -> (N,H,W,pk3 cout)."""
    assert bottleneck_ok(x, pk1, pk2, pk3, pkd)
    n, h, w, cin = x.shape
    cout = pk3["cout"]
    out = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    st = c_void_p(raw_stream(x.device))
    (p1, i1), (p2, i2), (p3, i3) = split_planes_f16(pk1), split_planes_f16(pk2), split_planes_f16(pk3)
    pd, idd = split_planes_f16(pkd) if pkd is not None else (None, 1.0)
    in_amax = amax_of(x)
    out_amax = None if NO_AMAX_COMMIT else AMAX.take(x.device)
    gw = guard_word(x.device)
    gl = (ctypes.c_float * 4)(*([guard_l1(pk1), guard_l1(pk2), guard_l1(pk3), guard_l1(pkd) if pkd is not None else 0.0] if gw is not None else [0.0] * 4))
    m = n * h * w
    flops = 2 * m * 64 * (cin + 9 * 64 + cout) + (2 * m * cin * cout if pkd is not None else 0)
    nbytes = 4 * (x.numel() + out.numel() + sum(pk["w"].numel() for pk in (pk1, pk2, pk3) + ((pkd,) if pkd is not None else ())))
    name = "k_bottleneck/f16x2"
    thunk = lambda: check(lib.ndet_bottleneck_f16x2(_ptr(x), n, h, w, cin, cout, _ptr(p1), i1, _ptr(pk1["scale"]), _ptr(pk1["shift"]), _ptr(p2), i2,
                                                    _ptr(pk2["scale"]), _ptr(pk2["shift"]), _ptr(p3), i3, _ptr(pk3["scale"]), _ptr(pk3["shift"]), _ptr(pd), idd,
                                                    _ptr(None if pkd is None else pkd["scale"]), _ptr(None if pkd is None else pkd["shift"]), _ptr(in_amax),
                                                    _ptr(out_amax), _ptr(out), gl, GUARD_TOL, _ptr(gw), st), "bottleneck_f16x2")
    if launch_hook is not None:
        launch_hook(flops, thunk, name)
    else:
        trace.span(name, thunk, flops=flops, bytes=nbytes, kind="conv")
    if out_amax is not None:
        _tag_amax(out, out_amax)
    return out


def bn_relu_maxpool_nhwc(x: torch.Tensor, bn: nn.BatchNorm2d) -> torch.Tensor:
    """x (N,H,W,C) contiguous -> MaxPool2d(3,2,1)(relu(bn_eval(x))) in one pass (ResNet stem tail)."""
    assert x.is_cuda and x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    n, h, w, c = x.shape
    scale, shift = bn_affine(bn)
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=torch.float32, device=x.device)
    st = c_void_p(raw_stream(x.device))
    check(_lib.load().ndet_bn_relu_maxpool_nhwc(_ptr(x), _ptr(scale), _ptr(shift), n, h, w, c, _ptr(out), st), "bn_relu_maxpool_nhwc")
    return out


def stem_ok(conv: nn.Module, bn: nn.Module, x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and isinstance(conv, nn.Conv2d) and conv.in_channels == 3 and conv.out_channels == 64
            and tuple(conv.kernel_size) == (7, 7) and tuple(conv.stride) == (2, 2) and tuple(conv.padding) == (3, 3) and conv.bias is None
            and not bn.training and ARITHMETIC in SPLIT_FAMILY and min(x.shape[2:]) >= 7)      # (bf16 mode too: the fused stem computes in bf16x3,
                                                                                                   # finer than asked; the library path costs 2.5x its time)


def stem_conv_bn_relu_maxpool(x: torch.Tensor, conv: nn.Conv2d, bn: nn.BatchNorm2d) -> torch.Tensor:
    """maxpool(relu(bn_eval(conv7x7s2(x)))) of the ResNet stem in one launch (csrc/stem_kernels.hip): logical (N,3,H,W) images in any
    strided layout -> (N, PH, PW, 64) channels-last."""
    assert stem_ok(conv, bn, x)
    n, _, h, w = x.shape
    store = conv.__dict__.setdefault("_ndet_packed", {})
    f16 = ARITHMETIC == "f16x2"          # fp16-pair form: three products per multiply, the patch scaled by its own maximum inside the kernel
    stamp = (conv.weight.data_ptr(), conv.weight._version, f16)
    hit = store.get("stem")
    lib = _lib.load()
    st = c_void_p(raw_stream(x.device))
    if hit is None or hit[0] != stamp:
        wf = conv.weight.detach().float().contiguous()
        if f16:
            wscale = f16_weight_scale(float(wf.abs().max()))
            planes = torch.empty((2, 64, 176), dtype=torch.int16, device=x.device)
            check(lib.ndet_stem_pack_weights_f16x2(_ptr(wf), wscale, _ptr(planes), st), "stem_pack_weights_f16x2")
            hit = store["stem"] = (stamp, planes, 1.0 / wscale)
        else:
            planes = torch.empty((3, 64, 176), dtype=torch.int16, device=x.device)
            check(lib.ndet_stem_pack_weights(_ptr(wf), _ptr(planes), st), "stem_pack_weights")
            hit = store["stem"] = (stamp, planes, 0.0)
    scale, shift = bn_affine(bn)
    ch, cw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty((n, (ch - 1) // 2 + 1, (cw - 1) // 2 + 1, 64), dtype=torch.float32, device=x.device)
    sn, sc, sy, sx = x.stride()
    out_amax = AMAX.take(x.device) if (ARITHMETIC == "f16x2" and not NO_AMAX_COMMIT) else None       # the first bottleneck's activation scale
    trace.span("k_stem_conv_pool", lambda: check(lib.ndet_stem_conv_bn_relu_maxpool(_ptr(x), n, h, w, sn, sc, sy, sx, _ptr(hit[1]), hit[2], _ptr(scale), _ptr(shift),
                                                                                  _ptr(out), _ptr(out_amax), st), "stem_conv_bn_relu_maxpool"),
               flops=2 * n * ch * cw * 64 * 147, bytes=4 * (x.numel() + out.numel()), kind="stem")
    if out_amax is not None:
        _tag_amax(out, out_amax)
    return out


def to_ndhwc(x: torch.Tensor) -> torch.Tensor:
    """logical (C,X,Y,Z) -> contiguous (X,Y,Z,C) (free when the memory already is channels-last)."""
    y = x.permute(1, 2, 3, 0)
    return carry_amax(x, y if y.is_contiguous() else y.contiguous())
