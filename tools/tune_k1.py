"""Scratch tuner (GPU box): builds K1 variants with different -D knobs and times them interleaved in one process."""
import ctypes, os, subprocess, sys, itertools
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfdet_amd import _lib
from nerfdet_amd.synth import ring_scene_meta
from nerfdet_amd import ops

CS = os.path.join(ROOT, "nerf-det_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "tune")
os.makedirs(OUT, exist_ok=True)

def build(tag, defs):
    so = os.path.join(OUT, f"lib_{tag}.so")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-shared", "-o", so, os.path.join(CS, "volume_kernels.hip")] + [f"-D{d}" for d in defs]
    subprocess.run(cmd, check=True)
    lib = ctypes.CDLL(so)
    fn = lib.ndet_backproject_aggregate
    fn.argtypes, fn.restype = _lib.SIGNATURES["ndet_backproject_aggregate"]
    return fn

def main():
    dev = torch.device("cuda")
    variants = {}
    for b, mw in itertools.product((4, 8, 12, 16), (1, 4, 6, 8)):
        variants[f"B{b}_W{mw}"] = build(f"B{b}_W{mw}", [f"GATHER_BATCH={b}", f"K1_MIN_WAVES={mw}"])
    n_v, C, hw, grid = 50, 256, (240, 320), (40, 40, 16)
    meta = ring_scene_meta(n_v, hw)
    f = torch.randn(n_v, hw[0] // 4, hw[1] // 4, C, device=dev)
    proj = ops.compute_projection(meta, 4, dev).contiguous()
    pts = ops.get_points(grid, (0.16, 0.16, 0.2), meta["lidar2img"]["origin"], dev)
    N = pts[0].numel()
    alpha = torch.rand(N, device=dev)
    cnt = torch.empty(N, dtype=torch.int64, device=dev)
    outs = {0: torch.empty(C, N, device=dev), 1: torch.empty(N, C, device=dev)}
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = None
    res = {k: {0: [], 1: []} for k in variants}
    for rnd in range(12):
        for name, fn in variants.items():
            for layout in (0, 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(P(f), n_v, C, hw[0] // 4, hw[1] // 4, f.stride(0), f.stride(1), P(pts), N, P(proj), P(alpha), P(outs[layout]), layout, P(cnt), st)
                e1.record()
                assert rc == 0
                torch.cuda.synchronize()
                if rnd >= 2:
                    res[name][layout].append(e0.elapsed_time(e1) * 1e3)
                if layout == 1:
                    if ref is None:
                        ref = outs[1].clone()
                    assert torch.equal(ref, outs[1]), name
    ab = n_v * C * (hw[0] // 4) * (hw[1] // 4) * 4 + (C * 4 + 8) * N
    for name in variants:
        for layout in (0, 1):
            t = sorted(res[name][layout])
            med = t[len(t) // 2]
            print(f"{name:10s} layout={'CN' if layout == 0 else 'NC'} median {med:7.1f} us  min {t[0]:7.1f} us  {ab / med / 1e3:7.1f} GB/s  frac {ab / med / 1e3 / 8000:.3f}", flush=True)

if __name__ == "__main__":
    main()
