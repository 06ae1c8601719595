"""Scratch tuner (GPU box) for plain GEMM shapes through the split-family convolution kernels: rows x Cout outputs contracted over 32 * ksteps input
channels -- the staged weight-gradient GEMMs of the training step (rows = taps * Cin, K = output voxels / 32) and its 1x1 layers.  Prints one
TUNED_JSON line per shape (key = conv_tuning's (rows, cout, ksteps, 0)).

    python tools/tune_gemm_shapes.py f16x2 2304,256,375 1024,256,375 ...
"""
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from nerfdet_amd import conv3d as C3
    C3.set_arithmetic(sys.argv[1])
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
    tiles = (64, 128, 12864, 128256, 129256, 129064, 100064, 100128, 112864)
    dev = torch.device("cuda")
    for m, cout, ksteps in shapes:
        cin = 32 * ksteps
        w = torch.randn(1, cout, cin, device=dev) / cin ** 0.5
        pk = dict(w=w, scale=None, shift=None, cout=cout, cin=cin, ksize=1, stride=1, transposed=False, kernel=(1, 1), strides=(1, 1), pads=(0, 0), ndim=2)
        x = torch.randn(1, 1, m, cin, device=dev)

        def run(**kw):
            ts = []
            for i in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); C3.conv2d_nhwc(x, pk, **kw); e1.record(); torch.cuda.synchronize()
                if i >= 2:
                    ts.append(e0.elapsed_time(e1))
            return sorted(ts)[1]
        best = None
        for tile, splits in itertools.product(tiles, (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)):
            if splits > ksteps:
                continue
            try:
                t = run(tile=tile, splits=splits)
            except Exception:
                continue
            if best is None or t < best[0]:
                best = (t, tile, splits)
        auto = run()
        fl = 2.0 * m * cout * cin
        print("TUNED_JSON", json.dumps(dict(key=[m, cout, ksteps, 0], tile=best[1], splits=best[2], us=best[0] * 1e3, auto_us=auto * 1e3,
                                            tflops=fl / best[0] / 1e9, auto_choice=list(C3.choose_tiling_split(m, cout, ksteps)))), flush=True)


if __name__ == "__main__":
    main()
