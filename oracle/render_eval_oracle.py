"""TEST INFRASTRUCTURE ONLY -- CPU (numpy / scipy) restatement of the novel-view metrics the reference computes when
``simple_test(evaluate_nerf=True)`` (mmdet3d/models/detectors/nerfdet.py:338-343 -> mmdet3d/models/model_utils/save_rendered_img.py:38-78):
per-view PSNR (:10-19), SSIM (:21-36) and the mean squared depth error map (:53-57).

PSNR and the depth map are the reference's own arithmetic.  SSIM is third-party: ``skimage.metrics.structural_similarity`` of
scikit-image, pinned at 0.18.1 by requirements/runtime.txt:7 and ABSENT from this image -- **parity unpinned** for it.  Restated from
the published algorithm (Wang, Bovik, Sheikh, Simoncelli 2004) with 0.18.1's documented defaults, as the reference's call reaches them:
``channel_axis`` does not exist in 0.18.1 and is swallowed by ``**kwargs``; the (H,W,3) image is then taken for a 3-D volume whose last
extent is smaller than the 7-wide window, which raises ValueError, and the ``except`` branch runs ``multichannel=True``: per-channel
SSIM, 7 x 7 uniform window, sample covariance (NP / (NP - 1)), K1 = 0.01, K2 = 0.03, ``data_range`` = 2 (the dtype range of float
images, -1 .. 1), float64 arithmetic, the mean over the window-valid interior (3-pixel border cropped), then the mean over the channels.
Only ``tests/`` may import this module.
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import uniform_filter


def psnr(pred: np.ndarray, target: np.ndarray) -> float:
    """save_rendered_img.py:10-19, maximum pixel value 1."""
    return float(-10.0 * np.log(np.mean((pred - target) ** 2)) / np.log(10.0))


def ssim_channel(x: np.ndarray, y: np.ndarray, win: int = 7, data_range: float = 2.0, k1: float = 0.01, k2: float = 0.03) -> float:
    x, y = x.astype(np.float64), y.astype(np.float64)
    n_p = win * win
    cov_norm = n_p / (n_p - 1.0)
    ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
    uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


def ssim(pred: np.ndarray, target: np.ndarray) -> float:
    """(H,W,3) float images -> mean of the three per-channel SSIMs (structural_similarity(..., multichannel=True))."""
    assert pred.shape == target.shape and pred.shape[-1] == 3
    return float(np.mean([ssim_channel(pred[..., c], target[..., c]) for c in range(3)]))


def ssim_by_definition(x: np.ndarray, y: np.ndarray, win: int = 7, data_range: float = 2.0) -> float:
    """One channel, window by window with explicit sample statistics (small images only): what the filters above compute."""
    h, w = x.shape
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    vals = []
    for i in range(h - win + 1):
        for j in range(w - win + 1):
            a, b = x[i:i + win, j:j + win].astype(np.float64).ravel(), y[i:i + win, j:j + win].astype(np.float64).ravel()
            ma, mb = a.mean(), b.mean()
            va, vb, vab = a.var(ddof=1), b.var(ddof=1), ((a - ma) * (b - mb)).sum() / (a.size - 1)
            vals.append(((2 * ma * mb + c1) * (2 * vab + c2)) / ((ma ** 2 + mb ** 2 + c1) * (va + vb + c2)))
    return float(np.mean(vals))


def rendering_metrics(rgb: np.ndarray, gt: np.ndarray, depth: np.ndarray, gt_depth: np.ndarray):
    """save_rendered_img.py:51-78 without the image dump: (mean PSNR, mean SSIM, mean squared depth error MAP) over the views."""
    n = gt.shape[0]
    p = sum(psnr(rgb[v], gt[v]) for v in range(n)) / n
    s = sum(ssim(rgb[v], gt[v]) for v in range(n)) / n
    e = sum((depth[v] - gt_depth[v]) ** 2 for v in range(n)) / n
    return p, s, e
