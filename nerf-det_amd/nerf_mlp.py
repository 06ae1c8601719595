"""Reference module name kept for drop-in imports (``model_utils/nerf_mlp.py``); the implementation lives in
:mod:`nerfdet_amd.radiance_field`."""
from .radiance_field import LayerStack, NerfMLP, SinusoidalEncoder, VanillaNeRFRadianceField  # noqa: F401
