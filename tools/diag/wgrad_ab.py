"""A/B of the implicit weight-gradient kernel's measurement knobs on the training step's large layers (one process, alternating):
    python tools/diag/wgrad_ab.py            # wgrad_xcd 0 / 1, wgrad_wide 0 / 1
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from nerfdet_amd import _lib, conv_train
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    shapes = [("neck 27-tap 256->256 @40x40x16", (40, 40, 16), 256, 256, (3, 3, 3)), ("fpn 9-tap 256->256 @40x60x80", (40, 60, 80), 256, 256, (3, 3)),
              ("layer2 9-tap 128->128 @40x30x40", (40, 30, 40), 128, 128, (3, 3)), ("neck 27-tap 256->128 @40x40x16", (40, 40, 16), 256, 128, (3, 3, 3))]
    for name, dims, cin, cout, k in shapes:
        x = torch.randn(*dims, cin, device=dev)
        g = torch.randn(*dims, cout, device=dev)

        def run():
            ts = []
            for i in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); conv_train.weight_grad(x, g, k, 1, None, implicit=True); e1.record(); torch.cuda.synchronize()
                if i >= 2:
                    ts.append(e0.elapsed_time(e1))
            return sorted(ts)[2] * 1e3
        row = []
        for knob, val in (("wgrad_xcd", 0), ("wgrad_xcd", 1), ("wgrad_xcd", 0), ("wgrad_xcd", 1), ("wgrad_wide", 0), ("wgrad_wide", 1)):
            _lib.check(lib.ndet_measurement_knob(knob.encode(), val), "knob")
            row.append(f"{knob}={val}: {run():7.1f} us")
        print(f"{name:36s} " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
