"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports exactly what
include/nerfdet_hip.h declares, and rejects bad arguments before touching the GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = ""
    for f in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if f.endswith(".h"):
            src += open(os.path.join(ROOT, "include", f)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ndet_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from nerfdet_amd import _lib
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported by {_lib.LIB_PATH}"
    # and the ctypes table binds exactly the declared surface
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.ndet_version() >= 100


def test_every_entry_point_cites_the_reference():
    txt = open(os.path.join(ROOT, "include", "nerfdet_hip.h")).read()
    for name in header_functions():
        if name in ("ndet_version", "ndet_last_error", "ndet_nchw_to_nhwc", "ndet_nms_workspace_bytes", "ndet_conv3d_workspace_bytes"):
            continue
        # the comment block right above the declaration names a reference file:line
        idx = re.search(r"\b\w+\s+" + name + r"\(", txt).start()
        block = txt[txt.rfind("/*", 0, idx):idx]
        assert re.search(r"\w+\.py:\d+", block), f"{name}: no reference file:line in its header comment"


def test_bad_arguments_are_rejected_without_a_gpu():
    """validation happens on the host before any HIP call, so this runs on a CPU-only box."""
    from nerfdet_amd import _lib
    lib = _lib.load()
    vs = _lib.float3([0.1, 0.1, 0.1])
    assert lib.ndet_get_points(None, 4, 4, 4, vs, vs, None) == -1
    assert b"null" in lib.ndet_last_error()
    fake = ctypes.c_void_p(0x1000)
    assert lib.ndet_get_points(fake, 0, 4, 4, vs, vs, None) == -1
    # C not a multiple of 4 -> unsupported
    assert lib.ndet_backproject_aggregate(fake, 2, 6, 4, 4, 96, 24, fake, 16, fake, None, fake, 1, fake, None) == -2
    # bad layout id
    assert lib.ndet_backproject_aggregate(fake, 2, 8, 4, 4, 128, 32, fake, 16, fake, None, fake, 7, fake, None) == -1
    # pitch smaller than a row
    assert lib.ndet_backproject_aggregate(fake, 2, 8, 4, 4, 128, 16, fake, 16, fake, None, fake, 1, fake, None) == -1
    assert lib.ndet_density_features(fake, 2, 62, 4, 4, 992, 248, fake, fake, 16, 16, 768, 256, 16, fake, 16, fake, fake, fake, None) == -2
    with pytest.raises(AssertionError):
        _lib.check(-1, "x")
    with pytest.raises(ValueError):
        _lib.check(-2, "x")


def test_product_path_has_no_cpu_fallback():
    import torch
    import nerfdet_amd.ops as ops
    with pytest.raises(RuntimeError):
        ops.backproject(torch.zeros(1, 4, 2, 2), torch.zeros(3, 1, 1, 1), torch.zeros(1, 3, 4))
    # nothing in the package imports the oracle
    pkg = os.path.join(ROOT, "nerf-det_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                assert "oracle" not in open(os.path.join(dp, f)).read().replace("oracle check", ""), f


def test_synthetic_rig_matches_the_oracles():
    import numpy as np
    from nerfdet_amd.synth import ring_scene_meta
    from oracle import nerfdet_oracle as O
    a, b = ring_scene_meta(50), O.ring_scene_meta(50)
    assert np.array_equal(a["lidar2img"]["intrinsic"], b["lidar2img"]["intrinsic"])
    assert all(np.array_equal(x, y) for x, y in zip(a["lidar2img"]["extrinsic"], b["lidar2img"]["extrinsic"]))
    assert a["ori_shape"] == b["ori_shape"] and a["img_shape"] == b["img_shape"]
