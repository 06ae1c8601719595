"""Data-parallel training plumbing for the path: one process per GPU, one scene per rank per step
(``samples_per_gpu=1``, config:133), gradients all-reduced by ``DistributedDataParallel`` -- RCCL over xGMI on the GPUs
(backend "nccl" on ROCm), gloo in the CPU tests.

Mirrors what ``mmdet.apis.train_detector`` sets up around the reference model (tools/train.py:98-155, config:167-186):
``MMDistributedDataParallel(find_unused_parameters=True)`` -- needed because ``cov.*``, ``mean_mapping``, ``cov_mapping``,
``mapping_2d`` and ``neck.fpn_convs.1-3`` never receive a gradient --, AdamW(lr 2e-4, wd 1e-4) with the
backbone at lr x0.1, gradient clipping at L2 norm 35.  The epoch runner, hooks and checkpoint cadence are out of scope."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel


class DDPDetector(DistributedDataParallel):
    """``MMDistributedDataParallel`` as the runner uses it: ``train_step(data, optimizer)`` goes through DDP's forward (so the
    gradient hooks are armed) and returns ``dict(loss, log_vars, num_samples)`` like ``BaseDetector.train_step``."""

    def train_step(self, data: Dict, optimizer=None, defer_log=False):
        losses = self(**data)
        loss, log_vars = self.module._parse_losses(losses, defer_log)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data["img_metas"]))


def wrap_ddp(model: torch.nn.Module, device: Optional[torch.device] = None, bucket_cap_mb: int = 128) -> DDPDetector:
    """config:185-186 (``find_unused_parameters=True``), tools/test.py:131-136 (``broadcast_buffers=False``).

    ``bucket_cap_mb``: the 433 MB of fp32 gradients go out in 4 buckets (:func:`ddp_bucket_plan`) instead of DDP's default 10 --
    xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring step is per-link bound, and fewer, larger messages amortise
    its 2(N-1) hops; the last bucket (the 3D neck, 77.6 M parameters, whose backward runs first) still overlaps the 2D
    backward."""
    assert dist.is_available() and dist.is_initialized(), "init the process group first (nerfdet_amd.dist.init_dist)"
    ids = None if device is None or device.type != "cuda" else [device.index if device.index is not None else torch.cuda.current_device()]
    return DDPDetector(model, device_ids=ids, find_unused_parameters=True, broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb,
                       gradient_as_bucket_view=True)


def ddp_bucket_plan(model: torch.nn.Module, bucket_cap_mb: int = 128) -> list:
    """Bytes per gradient bucket, in the order DDP reduces them (the reverse of ``model.parameters()``: the head's and the 3D neck's gradients are
    ready first), from the same assignment routine DDP's constructor calls -- a small first bucket so that the all-reduce starts early, then
    buckets that close once they have reached ``bucket_cap_mb``.  Printed by tools/bench_train.py on rank 0 and asserted in tests/test_ddp.py for the
    shipped model: 433 MB of fp32 gradients -> 4 all-reduces (7 / 184 / 136 / 106 MB) instead of the default 25 MB cap's 10."""
    params = [p for p in model.parameters() if p.requires_grad]
    first = getattr(dist, "_DEFAULT_FIRST_BUCKET_BYTES", 1024 * 1024)
    idx, _ = dist._compute_bucket_assignment_by_size(list(reversed(params)), [first, bucket_cap_mb * 1024 * 1024], [False] * len(params))
    rev = list(reversed(params))
    return [sum(rev[i].numel() * rev[i].element_size() for i in bucket) for bucket in idx]


FUSED_ADAMW = True     # torch's single-kernel-per-group AdamW when every parameter lives on the GPU (same update rule; ~100 launches fewer per step)


def build_optimizer(model: torch.nn.Module, lr: float = 2e-4, weight_decay: float = 1e-4, backbone_lr_mult: float = 0.1):
    """config:167-172: AdamW, ``paramwise_cfg=dict(custom_keys={'backbone': dict(lr_mult=0.1, decay_mult=1.0)})``."""
    module = model.module if isinstance(model, DistributedDataParallel) else model
    bb, rest = [], []
    for name, p in module.named_parameters():
        if p.requires_grad:
            (bb if name.startswith("backbone.") else rest).append(p)
    fused = FUSED_ADAMW and all(p.is_cuda for p in rest + bb)
    return torch.optim.AdamW([dict(params=rest), dict(params=bb, lr=lr * backbone_lr_mult)], lr=lr, weight_decay=weight_decay, **(dict(fused=True) if fused else {}))


class StepLog:
    """The logged scalars of one step on their way to the host: stacked on the device, copied into pinned memory without blocking, readable once
    the copy's event has passed.  Lets the caller queue step i + 1 before it reads step i's numbers -- the launch queue never drains at a step
    boundary -- where ``float(v)`` per scalar (what mmdet's ``_parse_losses`` does with ``.item()``) stalls the host until the device is idle."""

    def __init__(self, names, values):
        self.names = list(names)
        dev = torch.stack([v.detach().float().reshape(()) for v in values])
        self.host = torch.empty(dev.shape, dtype=torch.float32, pin_memory=dev.is_cuda)
        self.host.copy_(dev, non_blocking=True)
        self.event = torch.cuda.Event() if dev.is_cuda else None
        if self.event is not None:
            self.event.record(torch.cuda.current_stream(dev.device))

    def ready(self) -> bool:
        return self.event is None or self.event.query()

    def get(self) -> Dict[str, float]:
        """Blocks until THIS step's values have arrived (not until the device is idle)."""
        if self.event is not None:
            self.event.synchronize()
        return {k: float(v) for k, v in zip(self.names, self.host.tolist())}


def train_one_step(model, data: Dict, optimizer, grad_clip: float = 35.0, lazy_log: bool = False) -> Dict:
    """One iteration of the runner's loop: forward, backward (DDP all-reduces the gradients while it runs),
    clip (config:173), step.  ``lazy_log``: ``out["log"]`` is a :class:`StepLog` (``log_vars`` + ``grad_norm``, read with ``.get()``) instead of
    host floats -- the step then ends without a host synchronisation."""
    optimizer.zero_grad(set_to_none=True)
    module = model.module if isinstance(model, DistributedDataParallel) else model
    if next(module.parameters()).is_cuda:
        from . import conv_train
        conv_train.prepare_step(module)                         # fp16-pair training: every convolution weight's max |w| in two launches
    out = model.train_step(data, optimizer, defer_log=True)     # logged scalars stay on the device until the whole step is queued
    out["loss"].backward()
    # the optimizer's own parameter lists (build_optimizer: every trainable parameter of the model) instead of another walk over the module tree
    # (named_members over ~200 modules: ~0.7 ms of host time per step)
    params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None] if optimizer is not None else \
        [p for p in module.parameters() if p.requires_grad and p.grad is not None]
    norm = torch.nn.utils.clip_grad_norm_(params, grad_clip) if grad_clip and params else None
    optimizer.step()
    names = list(out["log_vars"]) + (["grad_norm"] if norm is not None else [])
    log = StepLog(names, list(out["log_vars"].values()) + ([norm] if norm is not None else []))
    if lazy_log:
        out["log"] = log
        out["log_vars"] = None
        return out
    vals = log.get()                                            # the step's only host sync after its first launches
    if norm is not None:
        out["grad_norm"] = vals.pop("grad_norm")
    out["log_vars"] = vals
    return out
