"""Host-side fp32 products of the camera matrices in a FIXED order of operations.

The reference leaves ``intrinsic @ extrinsic[:3]`` (nerfdet.py:377) and ``train_intrinsics.bmm(train_poses)`` (projection.py:52-58)
to the host's BLAS, so their last bit -- and, after ``.round()``, a voxel's pixel -- depends on the CPU the process happens to run
on.  The product path pins them to what the reference computes in the build container (where the golden fixtures were written):
MKL's k-ordered FMA chain for the 3x3 @ 3x4 product, torch's plain multiply-then-add loop for the small batched 4x4 one.  50 views
are 600 scalars: numpy, ~0.2 ms, same cost as the 50 tiny library calls it replaces.
"""
from __future__ import annotations

import numpy as np


def matmul_fma_chain(a, b) -> np.ndarray:
    """(...,m,K) @ (...,K,n), fp32: ``fma(a[K-1], b[K-1], ... fma(a[1], b[1], a[0]*b[0]))`` with one rounding per step.

    The hardware FMA is emulated exactly: the product of two fp32 values is exact in fp64; adding the fp32 accumulator in fp64 is
    made exact with TwoSum, the fp64 sum is rounded to odd, and the final conversion to fp32 then rounds once (53 >= 24 + 2)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    assert a.shape[-1] == b.shape[-2]
    acc = (a[..., :, 0, None].astype(np.float64) * b[..., None, 0, :].astype(np.float64)).astype(np.float32)
    for k in range(1, a.shape[-1]):
        prod = a[..., :, k, None].astype(np.float64) * b[..., None, k, :].astype(np.float64)
        c = acc.astype(np.float64)
        s = prod + c
        t = s - prod
        err = (prod - (s - t)) + (c - t)
        bits = s.view(np.int64)
        inexact_even = (err != 0) & ((bits & 1) == 0)
        bits = np.where(inexact_even, np.where((err > 0) == (s > 0), bits + 1, bits - 1), bits)
        acc = bits.view(np.float64).astype(np.float32)
    return acc


def matmul_mul_add(a, b) -> np.ndarray:
    """(...,m,K) @ (...,K,n), fp32: products rounded, then added left to right (no fusion)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    assert a.shape[-1] == b.shape[-2]
    acc = a[..., :, 0, None] * b[..., None, 0, :]
    for k in range(1, a.shape[-1]):
        acc = acc + a[..., :, k, None] * b[..., None, k, :]
    return acc
