"""Import alias.  The package directory is ``nerf-det_amd/`` (not a valid Python identifier), so
``import nerfdet_amd`` lands here and this shim loads that directory as the package of the same
name: ``nerfdet_amd.ops`` is ``nerf-det_amd/ops.py`` and so on."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nerf-det_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
