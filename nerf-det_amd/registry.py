"""Minimal registry + config-dict builder: the mmdet protocol the nerfdet configs rely on
(``dict(type='X', **kwargs)`` -> ``REGISTRY.get('X')(**kwargs)``), SURVEY.md section 8(b)-1.
mmcv/mmdet are not installed here; this keeps ``configs/nerfdet/*.py`` loadable unmodified."""
from __future__ import annotations

import copy
from typing import Any, Dict, Optional


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._modules: Dict[str, Any] = {}

    def register_module(self, name: Optional[str] = None, force: bool = False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls
        return _reg(module) if module is not None else _reg

    def get(self, key: str):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg: dict, **default_args):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"{self.name}: cfg must be a dict with a 'type' key, got {cfg!r}")
        args = copy.deepcopy(dict(cfg))
        typ = args.pop("type")
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        for k, v in default_args.items():
            args.setdefault(k, v)
        return cls(**args)


DETECTORS = Registry("detector")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
PIPELINES = Registry("pipeline")
DATASETS = Registry("dataset")


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d/models/builder.py: build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)."""
    from . import detector, backbone, neck3d, head, losses  # noqa: F401  (populate the registries)
    return DETECTORS.build(cfg, train_cfg=train_cfg, test_cfg=test_cfg)


def build_pipeline(cfg):
    from . import datasets  # noqa: F401  (populates the registry)
    return PIPELINES.build(cfg)


def build_dataset(cfg, default_args=None):
    """mmdet3d/datasets/builder.py: build_dataset(cfg.data.train / .val / .test)."""
    from . import datasets  # noqa: F401
    return DATASETS.build(cfg, **(default_args or {}))
