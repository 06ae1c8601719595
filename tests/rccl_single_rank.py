"""Child process of tests/test_ddp.py::test_rccl_single_rank_bringup: the RCCL code path of the job on the ONE GPU of the test box.

Started fresh (the parent hands over RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR/PORT, as torch.distributed.run would):
``init_dist('nccl')`` (tools/train.py:98-102), ``wrap_ddp`` (config:185-186), one ``train_one_step`` through
DistributedDataParallel whose bucketed gradient all-reduce, the head's ``reduce_mean`` (imvoxel_head_v2.py:175) and the logged-loss
all-reduce all run on RCCL.  With one rank the mean over ranks is the rank's own value, so every gradient must equal the plain
(non-DDP) step's up to the run-to-run noise of the float atomics, which is measured here by repeating the plain step.
Prints one JSON line."""
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    import nerfdet_amd.rays as R
    from nerfdet_amd import dist as D
    from nerfdet_amd.train import build_optimizer, train_one_step, wrap_ddp
    from test_ddp import WATCH, DEAD, _build, _grads, _scene
    rank, world, local = D.init_dist("nccl")
    assert (rank, world, local) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", local)
    assert D.max_over_ranks(1.5, dev) == 1.5                      # an all-reduce(MAX) on the device through RCCL
    t = torch.arange(8, dtype=torch.float32, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    assert t.tolist() == list(range(8))

    det = _build(dev)
    start = copy.deepcopy(det.state_dict())

    def plain():
        det.load_state_dict(start)
        det.zero_grad(set_to_none=True)
        R.rng = np.random.RandomState(234)
        torch.manual_seed(1)
        res = det.train_step(_scene(0, dev))
        res["loss"].backward()
        return _grads(det), res["log_vars"]
    g_a, logs_a = plain()
    g_b, _ = plain()
    ddp = wrap_ddp(det, dev)
    det.zero_grad(set_to_none=True)
    R.rng = np.random.RandomState(234)
    torch.manual_seed(1)
    res = ddp.train_step(_scene(0, dev))
    res["loss"].backward()
    g_d, logs_d = _grads(det), res["log_vars"]
    noise, err = {}, {}
    for k in WATCH:
        scale = max(float(g_a[k].abs().max()), 1e-12)
        noise[k] = float((g_a[k] - g_b[k]).abs().max()) / scale
        err[k] = float((g_d[k] - g_a[k]).abs().max()) / scale
    dead_ok = all(g_d[k] is None or float(g_d[k].abs().max()) == 0 for k in DEAD)
    # a whole optimizer step through the wrapper (clip + fused AdamW), twice: the reducer re-arms
    opt = build_optimizer(ddp)
    R.rng = np.random.RandomState(234)
    steps = [train_one_step(ddp, _scene(0, dev), opt) for _ in range(2)]
    finite = all(np.isfinite(s["grad_norm"]) and np.isfinite(s["log_vars"]["loss"]) for s in steps)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(dict(ok=True, backend="nccl", noise=noise, err=err, dead_ok=dead_ok, finite=finite,
                          loss_plain=logs_a["loss"], loss_ddp=logs_d["loss"], rccl=torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else None)))


if __name__ == "__main__":
    main()
