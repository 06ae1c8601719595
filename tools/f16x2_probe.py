"""GPU box: the fp16-pair arithmetic against fp64 on every tile family, beside bf16x3 and the fp32-MFMA kernel (error ratios), the amax slots,
and the chained bottleneck kernel.  argv: [quick]"""
import sys
import torch
from torch import nn
import torch.nn.functional as F
import nerfdet_amd  # noqa: F401
from nerfdet_amd import conv3d as C

dev = torch.device("cuda:0")
torch.manual_seed(0)


def rel(a, ref):
    a, ref = a.double().cpu(), ref.double()
    return ((a - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), ((a - ref).abs().max() / ref.abs().max()).item()


def run3d(cin, cout, dims, k, tile, splits, stride=1, res=False, relu=1, xmag=1.0):
    conv = nn.Conv3d(cin, cout, k, stride, k // 2, bias=False)
    bn = nn.BatchNorm3d(cout).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(*dims, cin) * xmag
    x = torch.relu(x) * torch.exp(torch.randn(*dims, 1))          # post-ReLU-like, per-voxel magnitude spread
    xin = x.permute(3, 0, 1, 2).unsqueeze(0).double()
    ref = bn.double()(conv.double()(xin))
    od = ref.shape[2:]
    r = torch.randn(*od, cout) if res else None
    if res:
        ref = ref + r.permute(3, 0, 1, 2).unsqueeze(0).double()
    if relu:
        ref = torch.relu(ref)
    ref = ref[0].permute(1, 2, 3, 0)
    conv.float(); bn.float()
    pk = C.packed([conv.to(dev)], bn.to(dev))
    out = {}
    for arith in ("f32", "bf16x3", "f16x2"):
        prev = C.set_arithmetic(arith)
        try:
            kw = dict(splits=splits, tile=tile) if arith != "f32" else {}
            xd = x.to(dev)
            y = C.conv3d_ndhwc(xd, pk, residual=None if r is None else r.to(dev), relu=relu, **kw)
            torch.cuda.synchronize()
            out[arith] = rel(y, ref)
            if arith == "f16x2":
                slot = getattr(y, "_ndet_amax", None)
                assert slot is not None
                assert C.amax_value(slot) == y.abs().max().item(), (C.amax_value(slot), y.abs().max().item())
        finally:
            C.set_arithmetic(prev)
    print(f"3d cin{cin} cout{cout} {dims} k{k} s{stride} tile {tile} splits {splits} res {int(res)}: "
          + "  ".join(f"{a} rms {v[0]:.2e} max {v[1]:.2e}" for a, v in out.items())
          + f"  | f16x2/bf16x3 rms {out['f16x2'][0] / out['bf16x3'][0]:.2f}  f16x2/f32 {out['f16x2'][0] / out['f32'][0]:.2f}", flush=True)
    return out


def run_chain(n, h, w, cin, mid, cout, res=True):
    c2 = nn.Conv2d(cin, mid, 3, 1, 1, bias=False); b2 = nn.BatchNorm2d(mid).eval()
    c3 = nn.Conv2d(mid, cout, 1, bias=False); b3 = nn.BatchNorm2d(cout).eval()
    with torch.no_grad():
        for b in (b2, b3):
            b.weight.uniform_(0.5, 1.5); b.bias.normal_(0, 0.2); b.running_mean.normal_(0, 0.2); b.running_var.uniform_(0.5, 1.5)
    x = torch.relu(torch.randn(n, h, w, cin)) * torch.exp(torch.randn(n, h, w, 1))
    r = torch.randn(n, h, w, cout) if res else None
    xin = x.permute(0, 3, 1, 2).double()
    ref = b3.double()(c3.double()(torch.relu(b2.double()(c2.double()(xin)))))
    if res:
        ref = ref + r.permute(0, 3, 1, 2).double()
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    for m in (c2, b2, c3, b3):
        m.float().to(dev)
    pk2, pk3 = C.packed([c2], b2), C.packed([c3], b3)
    out = {}
    for arith in ("bf16x3", "f16x2"):
        prev = C.set_arithmetic(arith)
        try:
            y = C.conv2d_chain_nhwc(x.to(dev), pk2, pk3, residual=None if r is None else r.to(dev), relu=1)
            torch.cuda.synchronize()
            out[arith] = rel(y, ref)
            if arith == "f16x2":
                assert C.amax_value(y._ndet_amax) == y.abs().max().item()
        finally:
            C.set_arithmetic(prev)
    print(f"chain n{n} {h}x{w} {cin}->{mid}->{cout}: " + "  ".join(f"{a} rms {v[0]:.2e} max {v[1]:.2e}" for a, v in out.items())
          + f"  | ratio {out['f16x2'][0] / out['bf16x3'][0]:.2f}", flush=True)


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    # unified tiles (staged / direct epilogue), split-K
    for tile in (64, 128, 12864):
        run3d(64, 128, (6, 10, 12), 3, tile, 1)
        run3d(64, 128, (6, 10, 12), 3, tile, 3, res=True)
        run3d(128, 64, (6, 10, 12), 1, tile, 1, res=True)
    # wave-specialised one-shot and persistent tiles
    for tile in (128256, 129256, 129257, 129064):
        run3d(256, 256, (6, 10, 12), 3, tile, 1, res=True)
        run3d(256, 512, (6, 10, 12), 1, tile, 2)
    # halo tiles
    for tile in (3128, 3256, 3257, 3258):
        run3d(256, 256, (8, 12, 12), 3, tile, 1, res=True)
        run3d(128, 256, (8, 12, 12), 3, tile, 2)
    # stride 2, magnitudes far from 1
    run3d(256, 512, (8, 12, 12), 3, 128256, 2, stride=2)
    run3d(256, 256, (8, 12, 12), 3, 3257, 1, xmag=3.0e4)
    run3d(256, 256, (8, 12, 12), 3, 3257, 1, xmag=1.0e-6)
    run3d(256, 256, (8, 12, 12), 3, 0, 0)
    run_chain(2, 24, 32, 64, 64, 256)
    run_chain(2, 24, 32, 128, 128, 512)
    run_chain(2, 24, 32, 64, 64, 256, res=False)
    print("amax fallbacks", C.amax_fallbacks)
