"""nerfdet_amd -- MI355X (gfx950) native implementation of the NeRF-Det volumetric hot path.

Layout (only what the path needs):
  csrc/      hand-written HIP kernels + the C ABI (include/nerfdet_hip.h) -> lib/libnerfdet_hip.so
  _lib.py    ctypes binding (no fallback: a missing .so raises)
  ops.py     host mirror of the reference's module-level callables (get_points, backproject, ...)
  nerf_mlp.py  VanillaNeRFRadianceField mirror (same state-dict keys as the reference)
  volume.py  the fused inference pipeline: FPN features -> gated voxel volume + view count
"""
from . import _lib  # noqa: F401
from ._lib import LIB_PATH, NdetError  # noqa: F401

__all__ = ["ops", "nerf_mlp", "volume", "LIB_PATH", "NdetError"]


def __getattr__(name):  # lazy submodules: importing the package never needs the GPU
    if name in ("ops", "nerf_mlp", "volume", "rays", "detector", "registry", "config", "backbone", "neck3d", "head", "nms"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
