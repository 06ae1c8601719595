"""Scratch probes run on the GPU box (not part of the product): library conv3d/conv2d rates, CPU thread scaling."""
import os, sys, time, json
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

def conv3d_probe():
    dev = torch.device("cuda")
    print("miopen/cudnn enabled", torch.backends.cudnn.enabled, "benchmark", torch.backends.cudnn.benchmark)
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        for dtype in (torch.float32, torch.bfloat16):
            for cl in (False, True):
                x = torch.randn(1, 256, 40, 40, 16, device=dev, dtype=dtype)
                w = torch.randn(256, 256, 3, 3, 3, device=dev, dtype=dtype) * 0.01
                if cl:
                    x = x.contiguous(memory_format=torch.channels_last_3d); w = w.contiguous(memory_format=torch.channels_last_3d)
                try:
                    t = timeit(lambda: F.conv3d(x, w, None, 1, 1))
                    fl = 2 * 25600 * 256 * 256 * 27
                    print(f"conv3d 256->256 @40x40x16 bench={bench} {dtype} channels_last={cl}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s", flush=True)
                except Exception as e:
                    print("conv3d failed", dtype, cl, repr(e)[:200], flush=True)

def conv2d_probe():
    dev = torch.device("cuda")
    torch.backends.cudnn.benchmark = True
    for dtype in (torch.float32, torch.bfloat16):
        for cl in (False, True):
            x = torch.randn(50, 256, 60, 80, device=dev, dtype=dtype)
            w = torch.randn(256, 256, 3, 3, device=dev, dtype=dtype) * 0.01
            if cl:
                x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
            t = timeit(lambda: F.conv2d(x, w, None, 1, 1))
            fl = 2 * 50 * 4800 * 256 * 256 * 9
            print(f"conv2d 256->256 3x3 @50x60x80 {dtype} channels_last={cl}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s", flush=True)



def backbone_probe():
    import nerfdet_amd.backbone as B
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = B.ResNet(50, frozen_stages=1, norm_cfg=dict(type="BN", requires_grad=False), norm_eval=True).to(dev).eval().to(memory_format=torch.channels_last)
    fpn = B.FPN([256, 512, 1024, 2048], 256, 4).to(dev).eval().to(memory_format=torch.channels_last)
    fpn.active_outs = (0,)
    x = torch.randn(50, 3, 240, 320, device=dev).contiguous(memory_format=torch.channels_last)
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        with torch.no_grad():
            t = timeit(lambda: fpn(net(x)), n=5, warm=3)
        print(f"backbone+fpn fp32 channels_last cudnn.benchmark={bench}: {t*1e3:.2f} ms", flush=True)
    # fused conv+bias+relu through MIOpen's fusion path
    conv_w = torch.randn(256, 256, 3, 3, device=dev).contiguous(memory_format=torch.channels_last) * 0.01
    bias = torch.randn(256, device=dev)
    y = torch.randn(50, 256, 60, 80, device=dev).contiguous(memory_format=torch.channels_last)
    try:
        f = lambda: torch.ops.aten.miopen_convolution_relu(y, conv_w, bias, [1, 1], [1, 1], [1, 1], 1)
        t1 = timeit(f)
        g = lambda: F.relu(F.conv2d(y, conv_w, bias, 1, 1))
        t2 = timeit(g)
        h = lambda: F.conv2d(y, conv_w, None, 1, 1)
        t3 = timeit(h)
        print(f"3x3 256->256: miopen_convolution_relu {t1*1e3:.3f} ms | conv+bias then relu {t2*1e3:.3f} ms | conv only {t3*1e3:.3f} ms", flush=True)
        w1 = torch.randn(256, 64, 1, 1, device=dev).contiguous(memory_format=torch.channels_last) * 0.05
        z = torch.randn(50, 64, 60, 80, device=dev).contiguous(memory_format=torch.channels_last)
        t1 = timeit(lambda: torch.ops.aten.miopen_convolution_relu(z, w1, bias, [1, 1], [0, 0], [1, 1], 1))
        t2 = timeit(lambda: F.relu(F.conv2d(z, w1, bias)))
        t3 = timeit(lambda: F.conv2d(z, w1, None))
        print(f"1x1 64->256: miopen_convolution_relu {t1*1e3:.3f} ms | conv+bias then relu {t2*1e3:.3f} ms | conv only {t3*1e3:.3f} ms", flush=True)
    except Exception as e:
        print("miopen_convolution_relu failed:", repr(e)[:300], flush=True)


if __name__ == "__main__":
    for name in sys.argv[1:]:
        globals()[name]()
