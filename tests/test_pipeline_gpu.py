"""Input contract on the GPU: nerfdet_amd.pipeline.MultiViewPipeline through the C ABI
against the golden vectors of the reference's MultiViewPipeline + DefaultFormatBundle and against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g():
    z = np.load(os.path.join(GOLDEN, "pipeline_small.npz"))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("tag", ["random", "seq"])
def test_pipeline_batch_matches_reference(device, g, tag):
    from nerfdet_amd import pipeline as P
    cams = P.scene_cameras(dict(extrinsics=list(g["poses"]), intrinsics=g["intrinsic"], annos=dict(axis_align_matrix=g["axis_align"])))
    pipe = P.MultiViewPipeline(int(g[f"{tag}__n_images"]), mean=g["mean"], std=g["std"], margin=int(g["margin"]),
                               loading=str(g[f"{tag}__loading"]), nerf_target_views=int(g[f"{tag}__n_target"]))
    np.random.seed(int(g[f"{tag}__seed"]))
    batch = pipe(torch.from_numpy(g["frames"]).to(device), cams, (int(g["ori_hw"][0]), int(g["ori_hw"][1]), 3))
    # exact: the normalised views, the de-normalised uint8 round trip, target colours, camera centres, view selection
    assert torch.equal(batch["img"][0].cpu(), torch.from_numpy(g[f"{tag}__img"]))
    assert torch.equal(batch["denorm_images"][0].cpu(), torch.from_numpy(g[f"{tag}__denorm_images"]))
    assert torch.equal(batch["lightpos"][0].cpu(), torch.from_numpy(g[f"{tag}__lightpos"]))
    assert np.array_equal(np.stack(batch["img_metas"][0]["lidar2img"]["extrinsic"]), g[f"{tag}__extrinsic"])
    gt = batch["gt_images"][0].cpu().double().numpy()
    assert np.abs(gt - g[f"{tag}__gt_images"]).max() <= 1e-7            # the reference keeps float64 here
    # rays: numpy's (HW,3)@(3,3) float32 product vs the kernel's FMA chain -- last-bit differences only
    rd = batch["raydirs"][0].cpu().numpy()
    assert np.abs(rd - g[f"{tag}__raydirs"]).max() <= 2e-7 * max(1.0, np.abs(g[f"{tag}__raydirs"]).max())
    assert [tuple(s[0].tolist()) for s in batch["nerf_sizes"]] == [tuple(r) for r in g[f"{tag}__nerf_sizes"]]
    assert batch["img"].shape[:2] == (1, len(g[f"{tag}__img"])) and batch["raydirs"].dim() == 4


def test_pipeline_feeds_the_detector(device, g):
    """The assembled batch has the shapes nerfdet.forward_test consumes (inference needs img + img_metas only)."""
    from nerfdet_amd import pipeline as P
    cams = P.scene_cameras(dict(extrinsics=list(g["poses"]), intrinsics=g["intrinsic"], annos=dict(axis_align_matrix=g["axis_align"])))
    pipe = P.MultiViewPipeline(6, margin=2, loading="sequence", sample_freq=2)
    batch = pipe(torch.from_numpy(g["frames"]).to(device), cams, (48, 64, 3))
    assert batch["img"].shape == (1, 6, 3, 24, 32) and "raydirs" not in batch
    assert len(batch["img_metas"][0]["lidar2img"]["extrinsic"]) == 6
    with pytest.raises(RuntimeError):
        pipe(torch.from_numpy(g["frames"]), cams, (48, 64, 3))


def test_pipeline_batch_runs_through_forward_test(device, g):
    """README usage: the batch the pipeline assembles is what nerfdet.forward_test takes."""
    from nerfdet_amd import pipeline as P
    from nerfdet_amd.config import _wrap
    from nerfdet_amd.presets import nerfdet_cfg
    from nerfdet_amd.registry import build_detector
    torch.manual_seed(0)
    cfg = _wrap(nerfdet_cfg(50, n_voxels=(16, 16, 8), voxel_size=(0.4, 0.4, 0.4)))
    det = build_detector(cfg["model"], train_cfg=cfg["train_cfg"], test_cfg=cfg["test_cfg"]).to(device).eval()
    cams = P.scene_cameras(dict(extrinsics=list(g["poses"]), intrinsics=g["intrinsic"], annos=dict(axis_align_matrix=g["axis_align"])))
    frames = torch.randint(0, 256, (14, 64, 96, 3), dtype=torch.uint8, device=device)
    batch = P.MultiViewPipeline(4, margin=3, loading="sequence", nerf_target_views=1)(frames, cams, (128, 192, 3))
    with torch.no_grad():
        res = det(return_loss=False, **batch)
    assert set(res[0]) == {"boxes_3d", "scores_3d", "labels_3d"}
