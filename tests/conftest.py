import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def load_golden(name):
    """npz fixture -> dict of torch tensors (numpy scalars/ints kept as numpy)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.ndim > 0 and a.dtype.kind in "fiub" else a
    return out


def golden_meta(g):
    """rebuild the img_meta dict a golden fixture was generated with."""
    return dict(
        lidar2img=dict(intrinsic=g["intrinsic"].numpy(), extrinsic=[e for e in g["extrinsic"].numpy()],
                       origin=g["origin"].numpy()),
        ori_shape=tuple(int(v) for v in g["ori_shape"]), img_shape=tuple(int(v) for v in g["img_shape"]))


def sub_state(g, prefix):
    return {k[len(prefix):]: torch.as_tensor(v) for k, v in g.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
