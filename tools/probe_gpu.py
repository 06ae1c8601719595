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

if __name__ == "__main__":
    for name in sys.argv[1:]:
        globals()[name]()
