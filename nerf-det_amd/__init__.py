"""nerfdet_amd -- MI355X (gfx950) native implementation of the NeRF-Det volumetric hot path.

Layout (only what the path needs):
  csrc/             hand-written HIP kernels + the C ABI (include/nerfdet_hip.h) -> lib/libnerfdet_hip.so
  _lib.py           ctypes binding (no fallback: a missing .so raises)
  ops.py            A1-A6 mirrors of the reference's module-level callables (get_points, backproject, ...)
  rays.py           A7-A12 mirrors (Projector, compute_mask_points, sample_along_camera_ray, raw2outputs, render_rays)
  nms.py            A15 aligned_3d_nms
  conv3d.py         MFMA convolution host side (weight/BN packing, tiling table) for the 3D neck/head and ResNet/FPN
  autograd.py       torch.autograd.Function pairs over the forward/backward kernels (training)
  volume.py         the fused inference pipeline: FPN features -> gated voxel volume + view count
  radiance_field.py VanillaNeRFRadianceField mirror (nerf_mlp.py is the reference-named alias)
  backbone.py, neck3d.py, head.py, losses.py, boxes.py, detector.py   the mirrored model surface
  registry.py, config.py, presets.py   mmdet-style registry / config loading so configs/nerfdet/*.py build unmodified
  graphed.py        optional hipGraph replay of the static part of forward_test
  dist.py           one-process-per-GPU scene sharding, result gather, reductions
  eval.py           indoor mAP@0.25/0.5
  pipeline.py       input contract: scene cameras, view sampling, normalised views + target rays on the GPU
  synth.py          deterministic synthetic ScanNet-shaped scenes for bench.py
  checkpoint.py     load_checkpoint / save_checkpoint in mmcv's file layout (meta + state_dict, module. prefix, key report)
  hostmath.py       the camera-matrix products in a fixed fp32 operation order (host independent)
"""
__version__ = "0.3.0"

from . import _lib  # noqa: F401
from ._lib import LIB_PATH, NdetError  # noqa: F401

_SUBMODULES = ("ops", "rays", "nms", "conv3d", "conv_tuning", "autograd", "volume", "radiance_field", "nerf_mlp", "backbone", "neck3d",
               "head", "losses", "boxes", "detector", "registry", "config", "presets", "graphed", "dist", "eval", "pipeline", "synth", "checkpoint", "hostmath", "train", "datasets")
__all__ = list(_SUBMODULES) + ["LIB_PATH", "NdetError"]


def __getattr__(name):  # lazy submodules: importing the package never needs the GPU
    if name in _SUBMODULES:
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
