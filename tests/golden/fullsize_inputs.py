"""Inputs of the BASELINE-size fixtures (``fullsize_cfg*.npz``), regenerated bit-identically on any host.

A 50-view 256x60x80 feature stack is 246 MB: it cannot be stored, so both sides -- ``make_golden_fullsize.py`` (which feeds
it to the REAL ``nerfdet.extract_feat`` in the build container) and ``tests/test_fullsize_reference_gpu.py`` (which feeds it
to the HIP path on the GPU box) -- rebuild it from a seed.  Everything is integer arithmetic on ``numpy.random.RandomState``
draws (a frozen stream, unlike ``torch.randn`` whose vectorised transforms may differ between CPUs) scaled by powers of two,
so every value is an exact fp32 number on every machine; the fixture stores a checksum of what the reference was fed.

No arithmetic of the reference lives here -- only data.
"""
from __future__ import annotations

import hashlib

import numpy as np

CONFIGS = {
    # BASELINE.json configs[1] / configs[0]: 50 / 10 views 240x320, FPN level 0 = 256 x 60 x 80, 40x40x16 voxels of 0.16x0.16x0.2
    "cfg2": dict(seed=20, n_views=50, channels=256, img_hw=(240, 320), n_voxels=(40, 40, 16), voxel_size=(0.16, 0.16, 0.2), n_sample=4096),
    "cfg1": dict(seed=21, n_views=10, channels=256, img_hw=(240, 320), n_voxels=(40, 40, 16), voxel_size=(0.16, 0.16, 0.2), n_sample=2048),
    # BASELINE.json configs[4] views and maps (101 views 320x480, FPN level 0 = 256 x 80 x 120) into an 80 x 80 x 4 SLAB of its 80x80x32 grid at its
    # voxel size: the reference materialises (n_v, C, N) and would need 21 GB per tensor for the whole grid; the slab is a grid of its own
    # (both sides are handed the same n_voxels and origin), 25 600 voxels seen by up to 101 views (the kernels' second 64-view round)
    "cfg5slab": dict(seed=22, n_views=101, channels=256, img_hw=(320, 480), n_voxels=(80, 80, 4), voxel_size=(0.08, 0.08, 0.1), n_sample=2048),
}
NEAR_TOL = 1e-3        # px: voxels with a view this close to a .5 rounding boundary are all stored
NEAR_CHANNEL_STEP = 4  # ... with every 4th channel (a neighbouring pixel changes every channel)


def _units(rs, shape, bits=16):
    """exact fp32 values k / 2^(bits-1) - 1 in [-1, 1)"""
    k = rs.randint(0, 1 << bits, size=shape, dtype=np.uint16 if bits <= 16 else np.uint32)
    return (k.astype(np.float32) - np.float32(1 << (bits - 1))) * np.float32(2.0 ** -(bits - 1))


def features(cfg):
    """(n_v, C, H/4, W/4) fp32, uniform in [-2, 2): the scale of the trained-like FPN level 0 of bench.py"""
    rs = np.random.RandomState(cfg["seed"])
    h, w = cfg["img_hw"]
    return _units(rs, (cfg["n_views"], cfg["channels"], h // 4, w // 4)) * np.float32(2.0)


def denorm_images(cfg):
    """(n_v, 3, H, W) fp32 = uint8 / 255 like ``MultiViewPipeline`` produces (multi_view.py:107-110)"""
    rs = np.random.RandomState(cfg["seed"] + 1000)
    h, w = cfg["img_hw"]
    return rs.randint(0, 256, size=(cfg["n_views"], 3, h, w), dtype=np.uint8).astype(np.float32) / np.float32(255.0)


def weights(cfg, shapes):
    """``shapes``: {state-dict key: shape} of the float parameters of ``mapping`` / ``nerf_mlp`` (integer buffers keep their
    constructor values).  Uniform in +-2^-k with 2^-k ~ sqrt(3 / fan_in) (unit gain), biases in +-1/8; the sigma head's bias
    is +1 so that alpha spreads over (0, 1)."""
    rs = np.random.RandomState(cfg["seed"] + 2000)
    out = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        if len(shape) >= 2:
            k = int(round(np.log2(np.sqrt(3.0 / shape[1]))))
            out[key] = _units(rs, shape) * np.float32(2.0 ** k)
        else:
            out[key] = _units(rs, shape) * np.float32(0.125)
            if key.endswith("sigma_layer.output_layer.bias"):
                out[key] = out[key] + np.float32(1.0)
    return out


def checksum(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def sample_voxels(cfg):
    n = int(np.prod(cfg["n_voxels"]))
    return np.sort(np.random.RandomState(cfg["seed"] + 3000).permutation(n)[: cfg["n_sample"]]).astype(np.int32)
