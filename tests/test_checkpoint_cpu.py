"""Checkpoint files in the layout the reference's tools exchange (tools/test.py:117-124, tools/train.py:146-152; mmcv-full 1.2.7
``save_checkpoint`` / ``load_checkpoint``): ``{'meta', 'state_dict'}``, CPU tensors, no ``module.`` prefix on save, prefix stripped on
load, missing / unexpected / mismatched keys reported (or raised with ``strict=True``)."""
import os

import pytest
import torch


def _small_detector(seed):
    from nerfdet_amd.config import _wrap
    from nerfdet_amd.presets import nerfdet_cfg
    from nerfdet_amd.registry import build_detector
    torch.manual_seed(seed)
    cfg = _wrap(nerfdet_cfg(50, n_voxels=(8, 8, 4), voxel_size=(0.8, 0.8, 0.8)))
    return build_detector(cfg["model"], train_cfg=cfg["train_cfg"], test_cfg=cfg["test_cfg"])


def test_save_load_round_trip_and_file_layout(tmp_path):
    from nerfdet_amd.checkpoint import load_checkpoint, save_checkpoint
    a, b = _small_detector(0), _small_detector(1)
    assert not torch.equal(a.neck_3d.out_block_0[0].weight, b.neck_3d.out_block_0[0].weight)
    path = str(tmp_path / "epoch_12.pth")
    opt = torch.optim.AdamW(a.parameters(), lr=2e-4)
    save_checkpoint(a, path, optimizer=opt, meta=dict(CLASSES=("cabinet", "bed"), config="model = dict(type='nerfdet')"))
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) == {"meta", "state_dict", "optimizer"}
    assert raw["meta"]["CLASSES"] == ("cabinet", "bed") and "time" in raw["meta"]
    assert list(raw["state_dict"]) == list(a.state_dict()), "keys and their order are the module's own (SURVEY.md appendix A)"
    assert all(not v.is_cuda and not v.requires_grad for v in raw["state_dict"].values())
    ptr_before = b.neck_3d.out_block_0[0].weight.data_ptr()
    ver_before = b.neck_3d.out_block_0[0].weight._version
    ck = load_checkpoint(b, path, map_location="cpu", strict=True)
    assert ck["meta"]["CLASSES"] == ("cabinet", "bed")                   # tools/test.py:122-125 reads it from the return value
    assert b.neck_3d.out_block_0[0].weight.data_ptr() == ptr_before and b.neck_3d.out_block_0[0].weight._version > ver_before, \
        "tensors are overwritten in place so that the weight-pack caches see the change"
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    assert b._ndet_load_report == dict(missing_keys=[], unexpected_keys=[], mismatched_keys=[])
    assert not any(f.startswith("epoch_12.pth.tmp") for f in os.listdir(tmp_path)), "written atomically"


def test_ddp_prefix_bare_state_dict_and_key_report(tmp_path):
    from nerfdet_amd.checkpoint import load_checkpoint
    a, b = _small_detector(0), _small_detector(1)
    sd = a.state_dict()
    # a checkpoint written from inside MMDistributedDataParallel: every key carries "module."
    p1 = str(tmp_path / "ddp.pth")
    torch.save(dict(meta=dict(), state_dict={"module." + k: v for k, v in sd.items()}), p1)
    load_checkpoint(b, p1, map_location="cpu", strict=True)
    assert torch.equal(b.bbox_head.cls_conv.weight, a.bbox_head.cls_conv.weight)
    # a bare state dict (torchvision-style file), with one key missing, one extra, one of another shape
    c = _small_detector(2)
    keep_missing = c.mapping[0].bias.clone()
    keep_mismatched = c.bbox_head.cls_conv.bias.clone()
    broken = {k: v for k, v in sd.items() if k != "mapping.0.bias"}
    broken["head_2d.fc.weight"] = torch.zeros(3)
    broken["bbox_head.cls_conv.bias"] = torch.zeros(20)
    broken.pop("neck_3d.down_layer_0.0.norm1.num_batches_tracked")        # old checkpoints lack the counters: not reported
    p2 = str(tmp_path / "bare.pth")
    torch.save(broken, p2)
    ck = load_checkpoint(c, p2, map_location="cpu")
    assert ck["meta"] == {}
    rep = c._ndet_load_report
    assert rep["missing_keys"] == ["mapping.0.bias"] and rep["unexpected_keys"] == ["head_2d.fc.weight"]
    assert len(rep["mismatched_keys"]) == 1 and rep["mismatched_keys"][0].startswith("bbox_head.cls_conv.bias")
    assert torch.equal(c.mapping[0].bias, keep_missing) and torch.equal(c.bbox_head.cls_conv.bias, keep_mismatched)
    assert torch.equal(c.mapping[0].weight, a.mapping[0].weight)
    with pytest.raises(RuntimeError, match="missing keys.*mapping.0.bias"):
        load_checkpoint(c, p2, map_location="cpu", strict=True)
    with pytest.raises(IOError):
        load_checkpoint(c, "torchvision://resnet50")
    with pytest.raises(IOError):
        load_checkpoint(c, str(tmp_path / "absent.pth"))


def test_backbone_pretrained_path_goes_through_the_same_loader(tmp_path):
    """config:3 ``pretrained='torchvision://resnet50'`` is a network fetch; a local file with torchvision names loads into the
    backbone (``fc.*`` reported as unexpected, like mmcv does for classification checkpoints)."""
    from nerfdet_amd.backbone import ResNet
    torch.manual_seed(0)
    src = ResNet(depth=50)
    sd = dict(src.state_dict())
    sd["fc.weight"] = torch.zeros(1000, 2048)
    path = str(tmp_path / "resnet50.pth")
    torch.save(sd, path)
    torch.manual_seed(1)
    dst = ResNet(depth=50)
    dst.init_weights(pretrained=path)
    assert torch.equal(dst.layer3[2].conv2.weight, src.layer3[2].conv2.weight)
    assert dst._ndet_load_report["unexpected_keys"] == ["fc.weight"] and not dst._ndet_load_report["missing_keys"]
