"""Detector checkpoints in the file layout the reference's tools read and write (mmcv-full 1.2.7 ``mmcv.runner.checkpoint``; call
sites tools/test.py:117-124, tools/train.py:146-152):

    {'meta': {..., 'CLASSES': (...)}, 'state_dict': OrderedDict(name -> CPU tensor) [, 'optimizer': ...]}

``load_checkpoint`` accepts that layout or a bare state dict, strips the ``module.`` prefix a (MM)DistributedDataParallel wrapper leaves
on every key, copies matching tensors IN PLACE (so the ``_version``-stamped weight packs of conv3d.py / head.py / backbone.py notice and
repack on the next forward) and reports -- or, with ``strict=True``, raises on -- missing keys, unexpected keys and shape mismatches,
as mmcv's ``load_state_dict`` does.  State-dict names are the reference's, so the released ``.pth`` files load
key for key.  mmcv's source is absent from the tree: this restates its documented behaviour (parity unpinned), and the round trip is
tested against itself (tests/test_checkpoint_*.py).
"""
from __future__ import annotations

import logging
import os
import time
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
from torch.nn.parallel import DataParallel, DistributedDataParallel

_WRAPPERS = (DataParallel, DistributedDataParallel)


def _unwrap(model: torch.nn.Module) -> torch.nn.Module:
    return model.module if isinstance(model, _WRAPPERS) else model


def load_state_dict(module: torch.nn.Module, state_dict: Dict[str, torch.Tensor], strict: bool = False, logger=None) -> Dict[str, List[str]]:
    """mmcv's tolerant ``load_state_dict``: every key that exists with the same shape is copied in place; the rest is collected.
    Returns ``dict(missing_keys, unexpected_keys, mismatched_keys)``; ``num_batches_tracked`` counters absent from old checkpoints
    are not reported as missing (mmcv filters them too)."""
    own = module.state_dict()          # references to the live parameters and buffers
    missing, unexpected, mismatched = [], [], []
    with torch.no_grad():
        for name, value in state_dict.items():
            if name not in own:
                unexpected.append(name)
                continue
            if not isinstance(value, torch.Tensor):
                value = torch.as_tensor(value)
            if tuple(own[name].shape) != tuple(value.shape):
                mismatched.append(f"{name}: checkpoint {tuple(value.shape)} vs model {tuple(own[name].shape)}")
                continue
            own[name].copy_(value)     # in place: data_ptr stays, _version moves
    for name in own:
        if name not in state_dict and "num_batches_tracked" not in name:
            missing.append(name)
    report = dict(missing_keys=missing, unexpected_keys=unexpected, mismatched_keys=mismatched)
    msgs = []
    if unexpected:
        msgs.append("unexpected key in source state_dict: " + ", ".join(unexpected))
    if missing:
        msgs.append("missing keys in source state_dict: " + ", ".join(missing))
    if mismatched:
        msgs.append("size mismatch for " + "; ".join(mismatched))
    if msgs:
        text = "The model and loaded state dict do not match exactly\n\n" + "\n\n".join(msgs)
        if strict:
            raise RuntimeError(text)
        (logger or logging.getLogger("nerfdet_amd")).warning(text)
    return report


def load_checkpoint(model: torch.nn.Module, filename: str, map_location=None, strict: bool = False, logger=None) -> dict:
    """``mmcv.runner.load_checkpoint(model, filename, map_location='cpu')`` (tools/test.py:117).  Returns the checkpoint dict (the
    caller reads ``checkpoint['meta']['CLASSES']``, tools/test.py:122-125); the last load report is left on
    ``model._ndet_load_report``."""
    if filename.startswith(("http://", "https://", "torchvision://", "open-mmlab://")):
        raise IOError(f"{filename}: network checkpoints cannot be fetched here; pass a local file")
    if not os.path.isfile(filename):
        raise IOError(f"{filename} is not a checkpoint file")
    checkpoint = torch.load(filename, map_location=map_location, weights_only=False)
    if not isinstance(checkpoint, dict):
        raise RuntimeError(f"No state_dict found in checkpoint file {filename}")
    state_dict = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
    if state_dict and all(k.startswith("module.") for k in state_dict):
        state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
    target = _unwrap(model)
    target.__dict__["_ndet_load_report"] = load_state_dict(target, state_dict, strict, logger)
    if "meta" not in checkpoint and "state_dict" not in checkpoint:
        checkpoint = dict(meta={}, state_dict=state_dict)
    checkpoint.setdefault("meta", {})
    return checkpoint


def weights_to_cpu(state_dict: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, v.detach().cpu()) for k, v in state_dict.items())


def save_checkpoint(model: torch.nn.Module, filename: str, optimizer=None, meta: Optional[dict] = None) -> None:
    """``mmcv.runner.save_checkpoint``: unwraps (MM)DDP so that no key carries ``module.``, moves every tensor to the CPU, stamps
    ``meta`` (the runner adds the config text and ``CLASSES``, tools/train.py:146-152) and writes atomically."""
    meta = dict(meta or {})
    meta.setdefault("time", time.asctime())
    from . import __version__ as version
    meta.setdefault("nerfdet_amd_version", version)
    target = _unwrap(model)
    if "CLASSES" not in meta and getattr(target, "CLASSES", None) is not None:
        meta["CLASSES"] = target.CLASSES
    checkpoint = dict(meta=meta, state_dict=weights_to_cpu(target.state_dict()))
    if optimizer is not None:
        checkpoint["optimizer"] = ({k: o.state_dict() for k, o in optimizer.items()} if isinstance(optimizer, dict) else optimizer.state_dict())
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    tmp = f"{filename}.tmp.{os.getpid()}"
    torch.save(checkpoint, tmp)
    os.replace(tmp, filename)
