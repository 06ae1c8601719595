"""Tiny stand-in for ``mmcv.Config.fromfile`` on plain-Python config files (the nerfdet configs are flat:
no ``_base_``, only literals and a few computed lists -- configs/nerfdet/nerfdet_res50_2x_low_res.py:58-79)."""
from __future__ import annotations

import os
import types


class ConfigDict(dict):
    """dict with attribute access, recursively (``cfg.test_cfg.nms_pre``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


class Config(ConfigDict):
    @staticmethod
    def fromfile(path: str) -> "Config":
        path = os.path.abspath(path)
        scope: dict = {"__file__": path}
        with open(path) as f:
            exec(compile(f.read(), path, "exec"), scope)
        out = Config()
        for k, v in scope.items():
            if k.startswith("_") or isinstance(v, (types.ModuleType, types.FunctionType)):
                continue
            out[k] = _wrap(v)
        out["filename"] = path
        return out

    def merge_from_dict(self, options: dict):
        """``--options a.b=c`` overrides (tools/train.py:71-72)."""
        for key, val in options.items():
            d = self
            parts = key.split(".")
            for p in parts[:-1]:
                d = d.setdefault(p, ConfigDict())
            d[parts[-1]] = _wrap(val)
