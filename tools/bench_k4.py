"""Scratch (GPU box): K4 at the training shapes of BASELINE configs[2] (2048 rays x 64 samples, 40 source views 240x320, 32 mapped
channels): generic vs packed kernel, forward and backward, event-timed; algorithmic bytes per SURVEY.md 8(d) = 113.5 MB."""
import os, sys, json
from ctypes import c_void_p
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfdet_amd import _lib, rays
from nerfdet_amd._lib import check
from nerfdet_amd.synth import ring_scene_meta


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def main():
    dev = torch.device("cuda")
    n_v, d, hw, R, S = int(os.environ.get("NV", 40)), 32, (240, 320), 2048, 64
    gen = torch.Generator().manual_seed(1)
    meta = ring_scene_meta(n_v, hw)
    feat = torch.randn(n_v, d, hw[0] // 4, hw[1] // 4, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    img = torch.rand(n_v, 3, *hw, generator=gen).to(dev)
    # rays of 10 target views looking inwards from the camera ring, as the training step draws them
    ang = torch.rand(R, generator=gen) * 2 * np.pi
    ray_o = torch.stack([2.5 * torch.cos(ang), 2.5 * torch.sin(ang), 1.2 + 0 * ang], -1)
    ray_d = -ray_o / ray_o.norm(dim=-1, keepdim=True) + 0.35 * torch.randn(R, 3, generator=gen)
    pts, _ = rays.sample_along_camera_ray(ray_o.to(dev), ray_d.to(dev), [0.2, 8.0], S, det=True)
    cams = rays._compute_projection(meta)
    n = R * S
    abytes = 4 * (n_v * 3 * hw[0] * hw[1] + n_v * d * (hw[0] // 4) * (hw[1] // 4) + 70 * n)
    out = {"algorithmic_MB": abytes / 1e6}
    glob, pm, vc = rays.ray_view_stats(pts, img, cams, feat)
    out["mean_views_seeing_a_sample"] = float(vc.float().mean())
    out["fwd_packed_ms"] = timeit(lambda: rays.ray_view_stats(pts, img, cams, feat))
    out["pack_rgb_ms"] = timeit(lambda: (rays._PACKED_RGB.clear(), rays.packed_rgb(img)))
    saved = rays.packed_ok
    rays.packed_ok = lambda *a, **k: False
    out["fwd_generic_ms"] = timeit(lambda: rays.ray_view_stats(pts, img, cams, feat), 5)
    rays.packed_ok = saved
    lib = _lib.load()
    ke, h, w = rays._camera_matrices(cams.squeeze(0))
    ke = ke.to(dev)
    g = torch.randn(n, 70, generator=gen).to(dev)
    p = pts.reshape(-1, 3).contiguous()
    st = c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    df = torch.zeros(n_v, hw[0] // 4, hw[1] // 4, d, device=dev)
    for name, fn, reps in (("bwd_packed_ms", lib.ndet_ray_view_stats_packed_bwd, 20), ("bwd_generic_ms", lib.ndet_ray_view_stats_bwd, 5)):
        out[name] = timeit(lambda: check(fn(c_void_p(g.data_ptr()), c_void_p(p.data_ptr()), n, c_void_p(ke.data_ptr()), n_v, h, w,
                                            c_void_p(feat.data_ptr()), d, hw[0] // 4, hw[1] // 4, feat.stride(0), feat.stride(2),
                                            c_void_p(df.data_ptr()), st), "bwd"), reps)
    out["fwd_packed_GBs"] = abytes / out["fwd_packed_ms"] / 1e6
    out["fwd_packed_frac_of_8TBs"] = out["fwd_packed_GBs"] / 8000
    print(json.dumps(out))


if __name__ == "__main__":
    main()
