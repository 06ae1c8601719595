"""Scratch (GPU box): the fp16-pair / three-product arithmetic on the halo tiles against the bf16x3 kernels: time and accuracy (vs fp64)."""
import os, sys, math, ctypes
from ctypes import c_void_p
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import _lib, conv3d as C
from nerfdet_amd._lib import check

dev = torch.device("cuda")


def pow2_scale(t):
    m = float(t.abs().max())
    return 2.0 ** (14 - math.floor(math.log2(m))) if m > 0 else 1.0


def conv_f16x2(x, pk, kernel, tile, relu=0, residual=None, splits=1):
    lib = _lib.load()
    st = c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    w = pk["w"]
    taps, cout, cin = w.shape
    key = "w_f16"
    if key not in pk:
        sw = pow2_scale(w)
        planes = torch.empty((taps, cin // 32, 2, cout, 32), dtype=torch.int16, device=x.device)
        check(lib.ndet_split_weights_f16x2(c_void_p(w.data_ptr()), taps, cout, cin, sw, c_void_p(planes.data_ptr()), st), "split f16")
        pk[key] = (planes, sw)
    planes, sw = pk[key]
    sx = pow2_scale(x)
    base = pk["scale"] if pk["scale"] is not None else torch.ones(cout, device=x.device)
    shift = pk["shift"] if pk["shift"] is not None else torch.zeros(cout, device=x.device)
    scale = (base / (sw * sx)).contiguous()
    d, h, wd, _ = x.shape
    out = torch.empty((d, h, wd, cout), dtype=torch.float32, device=x.device)
    i3 = lambda v: (ctypes.c_int * 3)(*v)
    ws = torch.empty((d * h * wd * cout * splits * 4,), dtype=torch.uint8, device=x.device) if splits > 1 else None
    pad = tuple(k // 2 for k in kernel)
    check(lib.ndet_conv_ndhwc_f16x2(c_void_p(x.data_ptr()), c_void_p(planes.data_ptr()), c_void_p(out.data_ptr()), d, h, wd, cin, cout, i3(kernel), i3((1, 1, 1)),
                                    i3(pad), c_void_p(scale.data_ptr()), c_void_p(shift.data_ptr()), c_void_p(0 if residual is None else residual.data_ptr()),
                                    relu, splits, tile, sx, c_void_p(0 if ws is None else ws.data_ptr()), st), "conv f16x2")
    return out


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2]


def main():
    torch.manual_seed(0)
    # accuracy: channels spanning six decades, against fp64
    for (grid, cin, cout, k3) in (((6, 10, 12), 256, 256, (3, 3, 3)), ((4, 12, 16), 128, 128, (1, 3, 3))):
        conv = nn.Conv3d(cin, cout, k3, 1, tuple(k // 2 for k in k3), bias=False)
        with torch.no_grad():
            conv.weight.mul_(torch.logspace(-2, 1, cin).view(1, cin, 1, 1, 1))
        x = torch.randn(*grid, cin) * torch.logspace(1, -2, cin)
        ref = torch.nn.functional.conv3d(x.double().permute(3, 0, 1, 2).unsqueeze(0), conv.weight.double(), padding=tuple(k // 2 for k in k3))[0].permute(1, 2, 3, 0)
        ref32 = torch.nn.functional.conv3d(x.permute(3, 0, 1, 2).unsqueeze(0), conv.weight, padding=tuple(k // 2 for k in k3))[0].permute(1, 2, 3, 0)
        conv.to(dev)
        pk = dict(C.packed([conv]))
        xg = x.to(dev)
        scale = float(ref.abs().mean())
        for tile_b, tile_f in ((3257 if cout == 256 else 3128, 4257 if cout == 256 else 4128),):
            if k3[0] == 1:
                got_b = C.conv2d_nhwc(xg, dict(pk, kernel=k3[1:], strides=(1, 1), pads=(1, 1), ndim=2), tile=tile_b)
            else:
                got_b = C.conv3d_ndhwc(xg, pk, tile=tile_b)
            got_f = conv_f16x2(xg, pk, k3, tile_f)
            eb = float((got_b.double().cpu() - ref).abs().max()) / scale
            ef = float((got_f.double().cpu() - ref).abs().max()) / scale
            e32 = float((ref32.double() - ref).abs().max()) / scale
            print(f"accuracy grid={grid} cin={cin} cout={cout} k={k3}: max err / mean|ref|: bf16x3 {eb:.2e}  f16x2 {ef:.2e}  cpu-fp32 {e32:.2e}")
    # speed: the FPN output conv (50 x 60 x 80, 256 -> 256, 3x3) and the neck's 256-channel 3x3x3 layer (40 x 40 x 16)
    for name, grid, cin, cout, k3 in (("fpn.out0 3x3 256->256 @50x60x80", (50, 60, 80), 256, 256, (1, 3, 3)), ("neck 3x3x3 256->256 @40x40x16", (40, 40, 16), 256, 256, (3, 3, 3)),
                                      ("neck 3x3x3 512->512 @20x20x8", (20, 20, 8), 512, 512, (3, 3, 3)), ("l3.conv2 3x3 256->256 @50x15x20", (50, 15, 20), 256, 256, (1, 3, 3))):
        conv = nn.Conv3d(cin, cout, k3, 1, tuple(k // 2 for k in k3), bias=False).to(dev)
        pk = dict(C.packed([conv]))
        x = torch.randn(*grid, cin, device=dev)
        flops = 2 * grid[0] * grid[1] * grid[2] * cin * cout * k3[0] * k3[1] * k3[2]
        pk2 = dict(pk, kernel=k3[1:], strides=(1, 1), pads=(1, 1), ndim=2)
        for tb, tf in ((3257, 4257), (3256, 4256)):
            for splits in (1, 2, 4):
                if splits > cin // 32:
                    continue
                try:
                    fb = (lambda: C.conv2d_nhwc(x, pk2, tile=tb, splits=splits)) if k3[0] == 1 else (lambda: C.conv3d_ndhwc(x, pk, tile=tb, splits=splits))
                    t_b = timeit(fb)
                    t_f = timeit(lambda: conv_f16x2(x, pk, k3, tf, splits=splits))
                    print(f"{name:36s} tile {tb}/{tf} splits {splits}: bf16x3 {t_b * 1e3:7.1f} us ({flops / t_b / 1e9:6.1f} TF)   f16x2 {t_f * 1e3:7.1f} us ({flops / t_f / 1e9:6.1f} TF)   x{t_b / t_f:.2f}", flush=True)
                except Exception as e:
                    print(name, tb, splits, "failed", repr(e)[:120])


if __name__ == "__main__":
    main()
