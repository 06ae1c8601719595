"""The dataset / pipeline names of the nerfdet configs on a synthetic ScanNet-format scene
directory: ``cfg.data`` builds, a sample comes out in the collated batch format of SURVEY.md appendix B, ``RandomShiftOrigin``
follows numpy's RNG stream (multi_view.py:199-207), ``results.pkl`` round-trips (custom_3d.py:212-234) and ``evaluate`` scores it."""
import os
import pickle

import numpy as np
import pytest
import torch

REF_CFG = "/root/reference/configs/nerfdet"
CLASSES = None


def _scene_dir(tmp_path, n_scenes=2, n_frames=7, hw=(48, 64)):
    from PIL import Image
    from nerfdet_amd.synth import ring_scene_meta
    rng = np.random.RandomState(0)
    infos = []
    for s in range(n_scenes):
        meta = ring_scene_meta(n_frames, hw)
        paths = []
        for i in range(n_frames):
            rel = f"posed_images/scene{s:04d}/{i:05d}.png"
            os.makedirs(os.path.join(tmp_path, os.path.dirname(rel)), exist_ok=True)
            Image.fromarray(rng.randint(0, 256, (*hw, 3), dtype=np.uint8)).save(os.path.join(tmp_path, rel))
            paths.append(rel)
        k = 5 if s == 0 else 3
        ctr = (rng.rand(k, 3) * [4, 4, 1.5] - [2, 2, 0.25]).astype(np.float32)
        size = (0.5 + rng.rand(k, 3)).astype(np.float32)
        k4 = meta["lidar2img"]["intrinsic"].copy()
        k4[:2] /= 2.0                                    # ring_scene_meta quotes the intrinsics at 2x the image size; here ori == image
        infos.append(dict(img_paths=paths, intrinsics=k4.astype(np.float64),
                          extrinsics=[np.linalg.inv(e.astype(np.float64)) for e in meta["lidar2img"]["extrinsic"]],   # camera poses (c2w)
                          annos=dict(axis_align_matrix=np.eye(4), gt_num=k, gt_boxes_upright_depth=np.concatenate([ctr, size], 1),
                                     **{"class": rng.randint(0, 18, k)})))
    ann = os.path.join(tmp_path, "infos.pkl")
    with open(ann, "wb") as f:
        pickle.dump(infos, f)
    return ann, infos


def _data_cfg(root, ann, hw=(48, 64), n_images=5, targets=2, train=True):
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    mv = dict(type="MultiViewPipeline", n_images=n_images, transforms=[dict(type="LoadImageFromFile"), dict(type="Resize", img_scale=(hw[1], hw[0]), keep_ratio=True),
                                                                      dict(type="Normalize", **norm), dict(type="Pad", size=hw)],
              mean=norm["mean"], std=norm["std"], margin=4, depth_range=[0.5, 5.5], loading="random", nerf_target_views=targets)
    keys = ["img", "lightpos", "nerf_sizes", "raydirs", "gt_images", "gt_depths", "denorm_images"]
    if train:
        pipe = [dict(type="LoadAnnotations3D"), mv, dict(type="RandomShiftOrigin", std=(.7, .7, .0)), dict(type="DefaultFormatBundle3D", class_names=None),
                dict(type="Collect3D", keys=keys + ["gt_bboxes_3d", "gt_labels_3d"])]
    else:
        pipe = [mv, dict(type="DefaultFormatBundle3D", class_names=None, with_label=False), dict(type="Collect3D", keys=keys)]
    ds = dict(type="ScanNetMultiViewDataset", data_root=str(root), ann_file=ann, pipeline=pipe, classes=None, filter_empty_gt=True, box_type_3d="Depth",
              modality=dict(use_image=True, use_depth=False, use_lidar=False, use_neuralrecon_depth=False, use_ray=True), test_mode=not train)
    return dict(type="RepeatDataset", times=3, dataset=ds) if train else ds


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference configs only exist in the build container")
def test_reference_cfg_data_builds_unmodified():
    from nerfdet_amd.config import Config
    from nerfdet_amd.datasets import MultiViewPipeline, RandomShiftOrigin, RepeatDataset, ScanNetMultiViewDataset
    from nerfdet_amd.registry import build_dataset
    for f in ("nerfdet_res50_2x_low_res.py", "nerfdet_res101_2x_low_res_depth_sp.py"):
        cfg = Config.fromfile(os.path.join(REF_CFG, f))
        train = build_dataset(cfg.data.train)
        assert isinstance(train, RepeatDataset) and train.times == 6 and isinstance(train.dataset, ScanNetMultiViewDataset)
        steps = train.dataset.pipeline.transforms
        assert [type(t).__name__ for t in steps] == ["LoadAnnotations3D", "MultiViewPipeline", "RandomShiftOrigin", "DefaultFormatBundle3D", "Collect3D"]
        mv = steps[1]
        assert isinstance(mv, MultiViewPipeline) and mv.nerf_target_views == 10 and mv.margin == 10
        assert mv.n_images == (48 if "101" in f else 50)
        assert [type(t).__name__ for t in mv.transforms.transforms] == ["LoadImageFromFile", "Resize", "Normalize", "Pad"]
        assert isinstance(steps[2], RandomShiftOrigin) and tuple(steps[2].std) == (.7, .7, .0)
        test = build_dataset(cfg.data.test)
        assert test.test_mode and test.pipeline.transforms[0].n_images == 101 and test.pipeline.transforms[0].nerf_target_views == 1
        assert len(test.CLASSES) == 18 and test.CLASSES[0] == "cabinet"


def test_sample_format_shift_origin_and_results_roundtrip(tmp_path):
    from nerfdet_amd.boxes import DepthInstance3DBoxes
    from nerfdet_amd.datasets import collate_one, load_results
    from nerfdet_amd.registry import build_dataset
    ann, infos = _scene_dir(str(tmp_path))
    hw, n_images, targets, margin = (48, 64), 5, 2, 4
    train = build_dataset(_data_cfg(tmp_path, ann, hw, n_images, targets, train=True))
    assert len(train) == 6
    np.random.seed(3)
    sample = train[0]
    # replay the RNG stream: view draw, target draw, then the origin shift
    np.random.seed(3)
    ids = np.random.choice(np.arange(7), n_images, replace=False)
    tgt = np.random.choice(ids, targets, replace=False)
    src = np.setdiff1d(ids, tgt)
    shift = np.random.normal(.0, (.7, .7, .0), 3)
    meta = sample["img_metas"]
    assert np.allclose(meta["lidar2img"]["origin"], np.array([0, 0, .5]) + shift) and meta["lidar2img"]["origin"][2] == 0.5
    assert len(meta["lidar2img"]["extrinsic"]) == len(src)
    for e, i in zip(meta["lidar2img"]["extrinsic"], src):
        assert np.array_equal(e, np.linalg.inv(infos[0]["extrinsics"][i]).astype(np.float32))
    n_src, rays = len(src), (hw[0] - 2 * margin) * (hw[1] - 2 * margin)
    assert sample["img"].shape == (n_src, 3, *hw) and sample["img"].dtype == torch.float32
    assert sample["denorm_images"].shape == (n_src, 3, *hw) and 0 <= float(sample["denorm_images"].min()) and float(sample["denorm_images"].max()) <= 1
    assert sample["raydirs"].shape == (targets, rays, 3) and sample["lightpos"].shape == (targets, rays, 3) and sample["gt_images"].shape == (targets, rays, 3)
    assert isinstance(sample["gt_bboxes_3d"], DepthInstance3DBoxes) and len(sample["gt_bboxes_3d"]) == 5 and sample["gt_labels_3d"].dtype == torch.int64
    # the de-normalised copy is the uint8 frame again (BGR), / 255
    from PIL import Image
    frame = np.asarray(Image.open(os.path.join(tmp_path, infos[0]["img_paths"][src[0]])).convert("RGB"))[:, :, ::-1]
    assert np.abs(sample["denorm_images"][0].permute(1, 2, 0).numpy() * 255 - frame).max() <= 1.0
    batch = collate_one(sample)
    assert batch["img"].shape == (1, n_src, 3, *hw) and batch["gt_images"].dtype == torch.float32 and isinstance(batch["img_metas"], list)
    assert len(batch["nerf_sizes"]) == targets and tuple(batch["nerf_sizes"][0].shape) == (1, 3)
    assert batch["nerf_sizes"][0].tolist() == [[hw[0] - 2 * margin, hw[1] - 2 * margin, 3]]
    # test split: no labels, one result dict per scene -> results.pkl -> indoor_eval
    test = build_dataset(_data_cfg(tmp_path, ann, hw, n_images, 1, train=False))
    assert len(test) == 2 and "gt_bboxes_3d" not in test[1]
    results = []
    for info in infos:                                    # "detections" = the ground truth itself: mAP must be 1
        a = info["annos"]
        b = DepthInstance3DBoxes(torch.from_numpy(a["gt_boxes_upright_depth"].astype(np.float32)), box_dim=6, with_yaw=False, origin=(0.5, 0.5, 0.5))
        results.append(dict(boxes_3d=b, scores_3d=torch.linspace(0.9, 0.5, len(b)), labels_3d=torch.from_numpy(a["class"].astype(np.int64))))
    outs, tmp = test.format_results(results, pklfile_prefix=os.path.join(tmp_path, "results"))
    back = load_results(os.path.join(tmp_path, "results.pkl"))
    assert len(back) == 2 and torch.equal(back[0]["boxes_3d"].tensor, results[0]["boxes_3d"].tensor) and torch.equal(back[1]["labels_3d"], results[1]["labels_3d"])
    ret = test.evaluate(back)
    assert abs(ret["mAP_0.25"] - 1.0) < 1e-6 and abs(ret["mAP_0.50"] - 1.0) < 1e-6


def test_imresize_linear_has_cv2_geometry_and_no_antialiasing():
    """mmcv.imresize / imrescale = cv2.resize(INTER_LINEAR) (multi_view.py:104, mmdet Resize): half-pixel centres, two taps per axis
    however large the reduction.  Closed-form cases (cv2 itself is absent from the image)."""
    import numpy as np
    from nerfdet_amd.datasets import imresize_linear
    # 4x reduction of a float ramp: dst pixel d samples the ramp at 4d + 1.5 exactly (two taps around it) -- an antialiased
    # filter would return the same on a ramp, so the discriminating case follows
    ramp = np.tile(np.arange(32, dtype=np.float64), (8, 1))
    out = imresize_linear(ramp, (8, 2))
    assert out.shape == (2, 8) and out.dtype == np.float64
    assert np.allclose(out[0], 4 * np.arange(8) + 1.5, atol=1e-6)
    # a single hot pixel at column 5 of 16 -> 4 columns: dst 1 samples at 5.5 = 0.5*src[5] + 0.5*src[6]; no other dst pixel sees it.
    # Pillow's antialiased BILINEAR spreads it over dst 0..2 with weight < 0.5.
    img = np.zeros((4, 16), dtype=np.float32)
    img[:, 5] = 1.0
    out = imresize_linear(img, (4, 1))
    assert np.allclose(out[0], [0.0, 0.5, 0.0, 0.0])
    # zero (invalid) depths do not bleed further than one destination pixel
    depth = np.full((8, 16), 2.0)
    depth[:, 9] = 0.0
    out = imresize_linear(depth, (4, 2))
    assert np.allclose(out[0], [2.0, 2.0, 1.0, 2.0])       # dst 2 samples at 9.5 -> 0.5 * 0 + 0.5 * 2
    # uint8: fixed-point weights; upscaling 2x a two-pixel row: taps at -0.25 (clamped), 0.25, 0.75, 1.25 (clamped)
    row = np.array([[0, 200]], dtype=np.uint8)
    out = imresize_linear(row, (4, 1))
    assert out.dtype == np.uint8 and out.tolist() == [[0, 50, 150, 200]]
    # three channels, keep-ratio rescale as the nerfdet configs ask (968x1296 -> 239x320): shapes and value range
    rs = np.random.RandomState(0)
    frame = rs.randint(0, 256, (97, 130, 3), dtype=np.uint8)
    out = imresize_linear(frame, (32, 24))
    assert out.shape == (24, 32, 3) and out.dtype == np.uint8
    exact = imresize_linear(frame.astype(np.float64), (32, 24))
    assert np.abs(out.astype(np.float64) - exact).max() <= 1.0          # fixed point vs exact: one grey level at most
    assert imresize_linear(frame, (130, 97)) is not frame and np.array_equal(imresize_linear(frame, (130, 97)), frame)
