"""scratch: s_memtime stamps of a diagnostic build of k_conv_split_halo (first / last producer wave and first / last consumer wave of workgroup 0, left in
free words of the output's amax slot): per K step, ticks spent issuing + waiting for the weight DMA, at the step barrier, in the chunk transition."""
import os, sys, torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerfdet_amd import conv3d as C3
dev = torch.device("cuda")
C3.set_arithmetic("f16x2")
for name, mk, shape in (("3x3x3 256->256 40x40x16", lambda: nn.Conv3d(256, 256, 3, 1, 1, bias=False), (40, 40, 16, 256)),
                        ("3x3 256->256 50x60x80 (fpn.out0)", lambda: nn.Conv3d(256, 256, (1, 3, 3), 1, (0, 1, 1), bias=False), (50, 60, 80, 256))):
    for tile in (3257, 3258, 3256):
        conv = mk().to(dev)
        pk = C3.packed([conv])
        x = torch.randn(*shape, device=dev)
        try:
            for _ in range(3):
                y = C3.conv3d_ndhwc(x, pk, relu=1, tile=tile, splits=1) if conv.kernel_size[0] == 3 else None
        except Exception as e:
            print("skip", name, tile, e); continue
        if y is None:
            from nerfdet_amd.conv3d import conv2d_nhwc
            c2 = nn.Conv2d(256, 256, 3, 1, 1, bias=False).to(dev); pk = C3.packed([c2])
            for _ in range(3):
                y = conv2d_nhwc(x, pk, relu=1, tile=tile, splits=1)
        torch.cuda.synchronize()
        sl = y._ndet_amax.cpu().tolist()
        st = max(sl[8], 1.0)
        f = lambda o, n: "  ".join(f"{v / st:.0f}" for v in sl[o + 1:o + n])
        print(f"{name} tile {tile}: steps {int(sl[8])} | producer first [dma+wait, P1, restage, P2, total]: {f(8, 6)} | producer last: {f(40, 6)} | "
              f"consumer first [reads+mfma, P1, P2, total]: {f(16, 5)} | consumer last: {f(48, 5)}", flush=True)
