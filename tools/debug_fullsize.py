"""Scratch: where do the cfg2 voxel features differ from the oracle?"""
import os, sys, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import nerfdet_oracle as O
import nerfdet_amd.volume as V

torch.set_num_threads(32)
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = bench.WORKLOADS[wl]
dev = torch.device("cuda")
det_cpu = bench.build_model(w)
b = bench.synth_batch(w, 0)
meta = b["img_metas"][0]
det = copy.deepcopy(det_cpu).to(dev)
bg = bench.to_device(b, dev)
with torch.no_grad():
    x, _, stride = det.extract_2d(bg["img"])
    out = V.extract_volume(x, bg["denorm_images"][0], meta, det.n_voxels, det.voxel_size, det.mapping, det.nerf_mlp, stride=stride, channels_last_out=True)
    f_host = x.float().cpu().contiguous()
    ov = O.extract_volume(f_host, b["denorm_images"][0], meta, w["n_voxels"], w["voxel_size"], det_cpu.mapping[0].weight, det_cpu.mapping[0].bias, det_cpu.nerf_mlp.state_dict())
vol, ovol = out["volume"].cpu().reshape(256, -1), ov["volume"].reshape(256, -1)
err = (vol - ovol).abs().max(0)[0]
scale = float(ovol.abs().max())
bad = (err > 1e-4 * scale).nonzero().flatten()
print("scale", scale, "bad voxels", len(bad), "of", err.numel(), "max err", float(err.max()))
g, og = out["global_feat"].cpu(), ov["global_feat"]
gerr = (g - og).abs().max(1)[0]
print("global_feat max err", float(gerr.max()), "rows > 1e-4:", int((gerr > 1e-4).sum()))
alpha = out["alpha"].cpu().reshape(-1)
oalpha = (1 - torch.exp(-ov["density"])).reshape(-1)
aerr = (alpha - oalpha).abs()
print("alpha max err", float(aerr.max()), "n > 1e-5:", int((aerr > 1e-5).sum()))
cnt = ov["valid"].reshape(-1)
for i in bad[:10].tolist():
    ch = (g[i] - og[i]).abs()
    print(i, "cnt", int(cnt[i]), "vol err", float(err[i]), "alpha", float(alpha[i]), float(oalpha[i]), "glob err", float(gerr[i]), "worst ch", int(ch.argmax()),
          float(g[i][ch.argmax()]), float(og[i][ch.argmax()]))
# mean (ungated) comparison
mean = ov["mean"].reshape(256, -1)
print("mean scale", float(mean.abs().max()))
# density input sensitivity: raw sigma range
print("density range", float(ov["density"].min()), float(ov["density"].max()))
