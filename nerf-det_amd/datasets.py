"""The dataset / pipeline names the nerfdet configs build, host side, so that ``cfg.data``
builds and a ScanNet-format scene directory runs end to end:

  ScanNetMultiViewDataset  mmdet3d/datasets/scannet_monocular_dataset.py:13-99 + custom_3d.py:55-280 (get_data_info / get_ann_info /
                           prepare_*_data / format_results -> results.pkl / evaluate -> indoor_eval)
  RepeatDataset            mmdet's wrapper (config:136-137)
  MultiViewPipeline        datasets/pipelines/multi_view.py:12-196 (view sampling on numpy's global RNG stream, per-view transforms,
                           de-normalised copies, target-view rays / colours / depths)
  RandomShiftOrigin        multi_view.py:199-207
  LoadAnnotations3D        pipelines/loading.py:397 (the two fields the path uses)
  DefaultFormatBundle3D, Collect3D   pipelines/formating.py:33-117,186-207,237-295 (tensors instead of DataContainers: batch = 1)
  LoadImageFromFile, Resize, Normalize, Pad   mmdet 2.10 / mmcv transforms, third-party and absent from the tree: restated from their
                           documented behaviour (cv2 is not installed; PIL only decodes files; resampling is :func:`imresize_linear`, cv2.INTER_LINEAR's
                           two-tap geometry and fixed-point rounding) -- parity unpinned.

This is the reference's per-sample CPU flow (one data-loader worker per GPU, config:134).  The GPU-resident fast path for frames
already decoded to the device is nerfdet_amd.pipeline.MultiViewPipeline (two kernel launches per scene); both produce the batch
dict of SURVEY.md appendix B."""
from __future__ import annotations

import os
import pickle
import tempfile
from collections import defaultdict
from typing import List, Optional, Sequence

import numpy as np
import torch

from .boxes import DepthInstance3DBoxes
from .pipeline import get_dtu_raydir, select_views
from .registry import DATASETS, PIPELINES, build_dataset, build_pipeline  # noqa: F401  (re-exported)

SCANNET_CLASSES = ("cabinet", "bed", "chair", "sofa", "table", "door", "window", "bookshelf", "picture", "counter", "desk", "curtain",
                   "refrigerator", "showercurtrain", "toilet", "sink", "bathtub", "garbagebin")


class Compose:
    def __init__(self, transforms):
        self.transforms = [build_pipeline(t) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


# ---- third-party image transforms (mmdet 2.10 / mmcv), restated ---------------------------------------------------------------
@PIPELINES.register_module()
class LoadImageFromFile:
    """``mmcv.imread``: uint8 BGR (H,W,3); sets filename / ori_shape / img_shape."""

    def __init__(self, to_float32: bool = False, **kw):
        self.to_float32 = to_float32

    def __call__(self, results):
        from PIL import Image
        name = results["img_info"]["filename"]
        if results.get("img_prefix"):
            name = os.path.join(results["img_prefix"], name)
        img = np.asarray(Image.open(name).convert("RGB"))[:, :, ::-1].copy()
        if self.to_float32:
            img = img.astype(np.float32)
        results.update(filename=name, img=img, img_shape=img.shape, ori_shape=img.shape, img_fields=["img"])
        return results


def imresize_linear(img: np.ndarray, size_wh) -> np.ndarray:
    """``mmcv.imresize(img, (w, h))`` = ``cv2.resize(..., interpolation=cv2.INTER_LINEAR)``: two taps per axis at half-pixel centres,
    NO antialiasing however large the reduction (a 968x1296 ScanNet frame goes to 239x320 with 2x2 taps; Pillow's BILINEAR would average
    ~4x4 and smear invalid zero depths into their neighbours).  Geometry as in OpenCV's ``resize.cpp``: ``f = (d + 0.5) * (src / dst) - 0.5``
    in float, ``s = floor(f)``, taps clamped at the borders.  uint8 images use its 11-bit fixed-point weights and two-stage rounding,
    floating images are interpolated in their own precision.  cv2 / mmcv are absent from the image: restated from the published
    algorithm, parity unpinned (the fixed-point form differs from exact arithmetic by at most one grey level)."""
    nw, nh = int(size_wh[0]), int(size_wh[1])
    h, w = img.shape[:2]
    if (nw, nh) == (w, h):
        return img.copy()

    def taps(n_dst, n_src):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        f[lo], s[lo] = 0.0, 0
        hi = s >= n_src - 1
        f[hi], s[hi] = 0.0, n_src - 1
        return s, np.minimum(s + 1, n_src - 1), f
    x0, x1, fx = taps(nw, w)
    y0, y1, fy = taps(nh, h)
    src = img.reshape(h, w, -1)
    if img.dtype == np.uint8:
        ax1 = np.rint(fx * 2048.0).astype(np.int64)[None, :, None]
        ax0 = np.rint((1.0 - fx) * 2048.0).astype(np.int64)[None, :, None]
        by1 = np.rint(fy * 2048.0).astype(np.int64)[:, None, None]
        by0 = np.rint((1.0 - fy) * 2048.0).astype(np.int64)[:, None, None]
        rows = src.astype(np.int64)
        hor = rows[:, x0] * ax0 + rows[:, x1] * ax1                                # (h, nw, c), scaled by 2^11
        out = ((((by0 * (hor[y0] >> 4)) >> 16) + ((by1 * (hor[y1] >> 4)) >> 16) + 2) >> 2)
        out = np.clip(out, 0, 255).astype(np.uint8)
    else:
        ft = np.float64 if img.dtype == np.float64 else np.float32
        rows = src.astype(ft)
        ax1 = fx.astype(ft)[None, :, None]
        by1 = fy.astype(ft)[:, None, None]
        hor = rows[:, x0] * (1 - ax1) + rows[:, x1] * ax1
        out = (hor[y0] * (1 - by1) + hor[y1] * by1).astype(img.dtype if img.dtype.kind == "f" else ft)
    return out.reshape((nh, nw) + img.shape[2:])


@PIPELINES.register_module()
class Resize:
    """``Resize(img_scale=(w, h), keep_ratio=True)``: the largest rescale that fits the long edge into max(img_scale) and the short
    edge into min(img_scale) (mmcv.imrescale), bilinear in cv2's sense (:func:`imresize_linear`)."""

    def __init__(self, img_scale=None, keep_ratio: bool = True, **kw):
        self.img_scale, self.keep_ratio = tuple(img_scale), keep_ratio

    def __call__(self, results):
        img = results["img"]
        h, w = img.shape[:2]
        if self.keep_ratio:
            s = min(max(self.img_scale) / max(h, w), min(self.img_scale) / min(h, w))
            nw, nh = int(w * s + 0.5), int(h * s + 0.5)
        else:
            nw, nh = self.img_scale
        out = imresize_linear(img, (nw, nh))
        results.update(img=out, img_shape=out.shape, pad_shape=out.shape, scale_factor=np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32),
                       keep_ratio=self.keep_ratio)
        return results


@PIPELINES.register_module()
class Normalize:
    """``mmcv.imnormalize``: BGR -> RGB when ``to_rgb``, then (x - mean) / std in float32."""

    def __init__(self, mean, std, to_rgb: bool = True):
        self.mean, self.std, self.to_rgb = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32), to_rgb

    def __call__(self, results):
        img = results["img"].astype(np.float32)
        if self.to_rgb:
            img = img[:, :, ::-1]
        mean, stdinv = np.float64(self.mean.reshape(1, -1)), 1.0 / np.float64(self.std.reshape(1, -1))   # cv2.subtract / multiply in fp64 scalars
        results["img"] = ((img - mean) * stdinv).astype(np.float32)
        results["img_norm_cfg"] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results


@PIPELINES.register_module()
class Pad:
    """``Pad(size=(h, w))``: zero-pad at the bottom / right."""

    def __init__(self, size=None, size_divisor=None, pad_val=0):
        self.size, self.pad_val = size, pad_val

    def __call__(self, results):
        img = results["img"]
        h, w = self.size
        out = np.full((h, w) + img.shape[2:], self.pad_val, dtype=img.dtype)
        out[:img.shape[0], :img.shape[1]] = img
        results.update(img=out, pad_shape=out.shape, pad_fixed_size=self.size)
        return results


# ---- the reference's own pipeline steps ---------------------------------------------------------------------------------------
@PIPELINES.register_module()
class LoadAnnotations3D:
    def __init__(self, with_bbox_3d: bool = True, with_label_3d: bool = True, **kw):
        self.with_bbox_3d, self.with_label_3d = with_bbox_3d, with_label_3d

    def __call__(self, results):
        if self.with_bbox_3d:
            results["gt_bboxes_3d"] = results["ann_info"]["gt_bboxes_3d"]
            results.setdefault("bbox3d_fields", []).append("gt_bboxes_3d")
        if self.with_label_3d:
            results["gt_labels_3d"] = results["ann_info"]["gt_labels_3d"]
        return results


def _imdenormalize_bgr(img, mean, std):
    """``mmcv.imdenormalize(img, mean, std, to_bgr=True)``: img * std + mean, RGB -> BGR."""
    return (img.astype(np.float32) * np.float64(std.reshape(1, -1)) + np.float64(mean.reshape(1, -1))).astype(np.float32)[:, :, ::-1]


@PIPELINES.register_module()
class MultiViewPipeline:
    """multi_view.py:12-196 for the image modality (``pts_filename`` / point clouds belong to other detectors)."""

    def __init__(self, transforms, n_images, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), margin=10, depth_range=(0.5, 5.5),
                 loading="random", nerf_target_views=0, sample_freq=3):
        self.transforms = Compose(transforms)
        self.n_images, self.margin, self.depth_range = n_images, margin, list(depth_range)
        self.mean, self.std = np.array(mean), np.array(std)
        self.loading, self.sample_freq, self.nerf_target_views = loading, sample_freq, nerf_target_views

    def _depth(self, info, shape_hw):
        from PIL import Image
        name = info["filename"]
        if name.endswith(".npy"):
            return np.load(name)
        d = np.asarray(Image.open(name)) / 1000          # float64, as multi_view.py:102-104 hands it to mmcv.imresize
        return imresize_linear(d, (shape_hw[1], shape_hw[0]))

    def __call__(self, results):
        assert "pts_filename" not in results, "the point-cloud branch of MultiViewPipeline is not on the nerfdet path"
        ids, target_id = select_views(len(results["img_info"]), self.n_images, self.nerf_target_views, self.loading, self.sample_freq)
        imgs, depths, extrinsics, denorm = [], [], [], []
        ratio, last = 0, None
        for i in ids:
            last = self.transforms(dict(img_prefix=results["img_prefix"][i], img_info=results["img_info"][i]))
            ratio = last["ori_shape"][0] / last["img_shape"][0]
            if "depth_info" in results:
                depths.append(self._depth(results["depth_info"][i], last["img_shape"]))
            denorm.append(_imdenormalize_bgr(last["img"], self.mean, self.std).astype(np.uint8) / 255.0)
            imgs.append(last["img"])
            extrinsics.append(results["lidar2img"]["extrinsic"][i])
        height, width = imgs[0].shape[:2]
        if "ray_info" in results:
            assert self.nerf_target_views > 0
            k = results["lidar2img"]["intrinsic"].copy()
            k[:2] = k[:2] / ratio
            px, py = np.meshgrid(np.arange(self.margin, width - self.margin).astype(np.float32),
                                 np.arange(self.margin, height - self.margin).astype(np.float32))
            pix = np.stack((px, py), axis=-1).astype(np.float32)
            iy, ix = py.astype(np.int32), px.astype(np.int32)
            out = defaultdict(list)
            for i in target_id:
                out["c2w"].append(results["c2w"][i])
                out["camrotc2w"].append(results["camrotc2w"][i])
                out["lightpos"].append(results["lightpos"][i])
                out["pixels"].append(pix)
                out["raydirs"].append(np.reshape(get_dtu_raydir(pix, k, results["camrotc2w"][i]).astype(np.float32), (-1, 3)))
                t = self.transforms(dict(img_prefix=results["img_prefix"][i], img_info=results["img_info"][i]))
                bgr = _imdenormalize_bgr(t["img"], self.mean, self.std).astype(np.uint8)
                gt = bgr[iy, ix, :]
                out["nerf_sizes"].append(np.array(gt.shape))
                out["gt_images"].append(np.reshape(gt, (-1, 3)) / 255.0)
                if "depth_info" in results:
                    out["gt_depths"].append(self._depth(results["depth_info"][i], bgr.shape)[iy, ix])
            for key in ("c2w", "camrotc2w", "lightpos", "pixels", "raydirs", "gt_images", "gt_depths", "nerf_sizes"):
                results[key] = out[key]
            results["denorm_images"] = denorm
            results["depth_range"] = np.array([self.depth_range])
        for key, v in last.items():
            if key not in ("img", "img_prefix", "img_info"):
                results[key] = v
        results["img"] = imgs
        if depths:
            results["depth"] = depths
        results["lidar2img"]["extrinsic"] = extrinsics
        return results


@PIPELINES.register_module()
class RandomShiftOrigin:
    """multi_view.py:199-207: N(0, std) shift of the voxel-grid origin (numpy's global RNG stream, as the reference)."""

    def __init__(self, std):
        self.std = std

    def __call__(self, results):
        results["lidar2img"]["origin"] += np.random.normal(.0, self.std, 3)
        return results


@PIPELINES.register_module()
class DefaultFormatBundle3D:
    """formating.py:33-117,237-295: numpy -> tensors in the layouts of SURVEY.md appendix B.  ``DataContainer`` wrapping belongs
    to mmcv's collate / scatter; with batch size 1 (config:133) :func:`collate_one` does the same stacking."""

    def __init__(self, class_names, with_gt: bool = True, with_label: bool = True):
        self.class_names, self.with_gt, self.with_label = class_names, with_gt, with_label

    def __call__(self, results):
        if "img" in results:
            results["img"] = torch.from_numpy(np.ascontiguousarray(np.stack([im.transpose(2, 0, 1) for im in results["img"]], axis=0)))
        if "depth" in results:
            results["depth"] = torch.from_numpy(np.ascontiguousarray(np.stack(results["depth"], axis=0)))
        if "ray_info" in results:
            rd = np.ascontiguousarray(np.stack(results["raydirs"], axis=0))
            results["raydirs"] = torch.from_numpy(rd)
            lp = torch.from_numpy(np.ascontiguousarray(np.stack(results["lightpos"], axis=0)))
            results["lightpos"] = lp.unsqueeze(1).repeat(1, rd.shape[1], 1)
            results["gt_images"] = torch.from_numpy(np.ascontiguousarray(np.stack(results["gt_images"], axis=0)))
            if isinstance(results["gt_depths"], list) and len(results["gt_depths"]) != 0:
                gd = np.ascontiguousarray(np.stack(results["gt_depths"], axis=0))
                results["gt_depths"] = torch.from_numpy(gd)
                # the rays WITH depth, as the training-time ray draw wants them (render_ray.py:386-404), found here on the loader's side: on the
                # device the same ``nonzero`` hands its count to the host through a stream synchronisation (nerfdet_amd.rays.begin_selection)
                results["depth_rays"] = torch.from_numpy(np.flatnonzero(gd.reshape(-1) > 0))
            results["denorm_images"] = torch.from_numpy(np.ascontiguousarray(np.stack([im.transpose(2, 0, 1) for im in results["denorm_images"]],
                                                                                   axis=0))).float()
        if "gt_labels_3d" in results:
            results["gt_labels_3d"] = torch.as_tensor(np.asarray(results["gt_labels_3d"]))
        return results


@PIPELINES.register_module()
class Collect3D:
    META = ("filename", "ori_shape", "img_shape", "lidar2img", "pad_shape", "scale_factor", "flip", "pcd_horizontal_flip", "pcd_vertical_flip",
            "box_mode_3d", "box_type_3d", "img_norm_cfg", "rect", "Trv2c", "P2", "pcd_trans", "sample_idx", "pcd_scale_factor", "pcd_rotation",
            "pts_filename")

    def __init__(self, keys, meta_keys=META):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = dict(img_metas={k: results[k] for k in self.meta_keys if k in results})
        for k in self.keys:
            data[k] = results[k]
        if "gt_depths" in self.keys and "depth_rays" in results:      # travels with the depth maps it was derived from (DefaultFormatBundle3D above)
            data["depth_rays"] = results["depth_rays"]
        return data


def collate_one(sample: dict) -> dict:
    """mmcv ``collate`` + ``scatter`` for ``samples_per_gpu=1``: stacked tensors gain a batch axis, ``cpu_only`` objects become
    one-element lists, ``nerf_sizes`` a list of (1,3) tensors, float64 image-like arrays become float32 (formating.py:88)."""
    out = {}
    for k, v in sample.items():
        if k == "img_metas":
            out[k] = [v]
        elif k in ("gt_bboxes_3d", "gt_labels_3d"):
            out[k] = [v]
        elif k == "nerf_sizes":
            out[k] = [torch.as_tensor(np.asarray(s)).unsqueeze(0) for s in v]
        elif isinstance(v, torch.Tensor):
            out[k] = (v.float() if v.dtype == torch.float64 else v).unsqueeze(0)
        elif isinstance(v, np.ndarray):
            out[k] = torch.from_numpy(v).unsqueeze(0)
        else:
            out[k] = v
    return out


# ---- datasets -----------------------------------------------------------------------------------------------------------------
@DATASETS.register_module()
class ScanNetMultiViewDataset:
    CLASSES = SCANNET_CLASSES

    def __init__(self, data_root, ann_file, pipeline=None, classes=None, modality=None, box_type_3d="Depth", filter_empty_gt=True,
                 test_mode=False, data_infos: Optional[List[dict]] = None):
        assert str(box_type_3d).lower() == "depth", "ScanNet boxes live in depth coordinates (config:147)"
        self.data_root, self.ann_file, self.test_mode, self.modality = data_root, ann_file, test_mode, modality
        self.filter_empty_gt = filter_empty_gt
        self.box_type_3d = DepthInstance3DBoxes
        self.CLASSES = tuple(classes) if classes is not None else SCANNET_CLASSES
        self.cat2id = {n: i for i, n in enumerate(self.CLASSES)}
        self._infos = data_infos                     # the reference unpickles ann_file here; deferred so that a config builds without data
        self.pipeline = Compose(pipeline) if pipeline is not None else None

    @property
    def data_infos(self):
        if self._infos is None:
            with open(self.ann_file, "rb") as f:
                self._infos = pickle.load(f)
        return self._infos

    def __len__(self):
        return len(self.data_infos)

    def get_ann_info(self, index):
        """scannet_monocular_dataset.py:78-99."""
        a = self.data_infos[index]["annos"]
        if a["gt_num"] != 0:
            boxes, labels = a["gt_boxes_upright_depth"].astype(np.float32), a["class"].astype(np.int64)
        else:
            boxes, labels = np.zeros((0, 6), dtype=np.float32), np.zeros((0,), dtype=np.int64)
        return dict(gt_bboxes_3d=DepthInstance3DBoxes(boxes, box_dim=boxes.shape[-1], with_yaw=False, origin=(0.5, 0.5, 0.5)), gt_labels_3d=labels,
                    axis_align_matrix=a["axis_align_matrix"].astype(np.float32))

    def get_data_info(self, index):
        """scannet_monocular_dataset.py:16-76."""
        info = self.data_infos[index]
        m = self.modality or {}
        d = defaultdict(list)
        want_depth = m.get("use_depth") or m.get("use_neuralrecon_depth")
        if want_depth:
            d["depth_info"] = []
        align = info["annos"]["axis_align_matrix"].astype(np.float32)
        for i, rel in enumerate(info["img_paths"]):
            name = os.path.join(self.data_root, rel)
            d["img_prefix"].append(None)
            d["img_info"].append(dict(filename=name))
            if want_depth:
                d["depth_info"].append(dict(filename=name[:-4] + (".npy" if m.get("use_neuralrecon_depth") else ".png")))
            d["lidar2img"].append(np.linalg.inv(align @ info["extrinsics"][i]).astype(np.float32))
            if m.get("use_ray"):
                c2w = (align @ info["extrinsics"][i]).astype(np.float32)
                d["c2w"].append(c2w)
                d["camrotc2w"].append(c2w[0:3, 0:3])
                d["lightpos"].append(c2w[0:3, 3])
        d = dict(d)
        d["lidar2img"] = dict(extrinsic=d["lidar2img"], intrinsic=info["intrinsics"].astype(np.float32), origin=np.array([.0, .0, .5], dtype=np.float32))
        if m.get("use_ray"):
            d["ray_info"] = dict(c2w=d["c2w"], camrotc2w=d["camrotc2w"], lightpos=d["lightpos"])
        d["ann_info"] = self.get_ann_info(index)
        if not self.test_mode and self.filter_empty_gt and len(d["ann_info"]["gt_bboxes_3d"]) == 0:
            return None
        return d

    def _prepare(self, index):
        d = self.get_data_info(index)
        if d is None:
            return None
        d.update(img_fields=[], bbox3d_fields=[], box_type_3d=self.box_type_3d, box_mode_3d="depth")
        ex = self.pipeline(d)
        if not self.test_mode and self.filter_empty_gt and (ex is None or len(ex["gt_bboxes_3d"]) == 0):
            return None
        return ex

    def __getitem__(self, index):
        if self.test_mode:
            return self._prepare(index)
        while True:                                  # custom_3d.py:290-312: a scene without ground truth is replaced by another one
            ex = self._prepare(index)
            if ex is not None:
                return ex
            index = int(np.random.choice(len(self)))

    def format_results(self, outputs, pklfile_prefix=None, submission_prefix=None):
        """custom_3d.py:212-234: the detections as ``<prefix>.pkl`` (mmcv.dump of a list is pickle)."""
        tmp_dir = None
        if pklfile_prefix is None:
            tmp_dir = tempfile.TemporaryDirectory()
            pklfile_prefix = os.path.join(tmp_dir.name, "results")
        with open(f"{pklfile_prefix}.pkl", "wb") as f:
            pickle.dump(outputs, f)
        return outputs, tmp_dir

    def evaluate(self, results, metric=None, iou_thr=(0.25, 0.5), logger=None, show=False, out_dir=None):
        """custom_3d.py:236-280 -> indoor_eval (mAP@0.25 / 0.5 per class)."""
        from .eval import indoor_eval
        assert isinstance(results, list) and len(results) > 0 and len(results) == len(self.data_infos) and isinstance(results[0], dict)
        return indoor_eval([info["annos"] for info in self.data_infos], results, iou_thr, {i: n for i, n in enumerate(self.CLASSES)})


def load_results(path: str):
    """Read a ``results.pkl`` written by :meth:`ScanNetMultiViewDataset.format_results` (or by the reference's tools/test.py --out)."""
    with open(path, "rb") as f:
        return pickle.load(f)


@DATASETS.register_module()
class RepeatDataset:
    def __init__(self, dataset, times):
        self.dataset = build_dataset(dataset) if isinstance(dataset, dict) else dataset
        self.times, self.CLASSES = times, self.dataset.CLASSES

    def __len__(self):
        return self.times * len(self.dataset)

    def __getitem__(self, idx):
        return self.dataset[idx % len(self.dataset)]
