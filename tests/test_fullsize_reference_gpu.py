"""HIP path at BASELINE sizes against the REAL reference (tests/golden/fullsize_cfg*.npz: ``nerfdet.extract_feat`` run by
tests/golden/make_golden_fullsize.py on inputs both sides regenerate from a seed).  NO exclusion band:

  * view counts of all 25 600 voxels bit-exact (nerfdet.py:398-404 ``.round().long()`` + validity, :173 sum over views);
  * gated volume, the 70 conditioning values and alpha within 1e-4 (north_star) at the sampled voxels AND at every voxel that has a
    view within 1e-3 px of a rounding boundary;
  * A1 (``_compute_projection``, nerfdet.py:363-378) bit-exact on this host.
"""
import pytest
import torch

from test_fullsize_reference_cpu import check_against_reference, fullsize_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arithmetic", ["f16x2", "bf16x3", "f32"])      # f16x2: the default arithmetic, held to the reference-generated fixture directly
@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg5slab"])
def test_extract_volume_equals_reference_at_baseline_size(device, name, arithmetic):
    from nerfdet_amd import conv3d, ops
    from nerfdet_amd.volume import extract_volume
    g, cfg, feats, denorm, mapping, mlp, meta = fullsize_case(name)
    assert torch.equal(ops.compute_projection(meta, 4), g["projection"]), "A1 (stride 4) differs from the reference's _compute_projection"
    assert torch.equal(ops.compute_projection(meta, 1), g["rgb_projection"]), "A1 (stride 1) differs from the reference's _compute_projection"
    mapping.to(device)
    mlp.to(device)
    f = feats.to(device).contiguous(memory_format=torch.channels_last)
    prev = conv3d.set_arithmetic(arithmetic)
    try:
        with torch.no_grad():
            for cl in (True, False):
                out = extract_volume(f, denorm.to(device), meta, cfg["n_voxels"], cfg["voxel_size"], mapping, mlp, channels_last_out=cl)
                errs = check_against_reference(g, out["valid"].cpu(), out["volume"].reshape(256, -1).cpu(), out["global_feat"].cpu(),
                                               out["alpha"].reshape(-1).cpu(), tol=1e-4)
    finally:
        conv3d.set_arithmetic(prev)
    print(name, arithmetic, "HIP vs reference:", {k: f"{v:.1e}" for k, v in errs.items()}, f"{int(g['near_half'].sum())} voxels near a rounding boundary, 0 excluded")


def test_backproject_reference_api_counts_at_cfg1(device):
    """The exact-API form (``backproject``, nerfdet.py:393-420) at 10 views x 256 x 60x80 -> 40x40x16: its per-view validity summed over
    views is the reference's count in every voxel, and its view sum is the ungated mean x count."""
    from nerfdet_amd import ops
    g, cfg, feats, denorm, mapping, mlp, meta = fullsize_case("cfg1")
    pts = ops.get_points(cfg["n_voxels"], cfg["voxel_size"], meta["lidar2img"]["origin"], device)
    vol, valid = ops.backproject(feats.to(device), pts, g["projection"].to(device))
    assert vol.shape == (10, 256, 40, 40, 16) and valid.dtype == torch.bool
    cnt = valid.sum(0).reshape(-1).cpu()
    assert torch.equal(cnt, g["cnt"].reshape(-1).to(torch.int64))
    sel = g["sel"].long()
    seen = cnt[sel] > 0
    mean = (vol.sum(0).reshape(256, -1)[:, sel.to(device)].cpu() / (cnt[sel].float() + 1e-8)).t()
    got = mean * g["alpha_sel"][:, None]
    assert float((got - g["volume_sel"])[seen].abs().max()) <= 1e-5
