"""Scratch (GPU box): per-launch event timing of one cfg2 forward_test, in launch order, averaged over a few steps."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerfdet_amd import trace

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
if len(sys.argv) > 2 and sys.argv[2] == "nochain":     # layer_times.py cfg2 nochain: conv2 / conv3 of the bottlenecks as two launches
    from nerfdet_amd import conv3d
    conv3d.CHAIN_BOTTLENECKS = False
if len(sys.argv) > 2 and sys.argv[2] in ("f16x2", "bf16x3", "bf16", "f32"):     # layer_times.py cfg2 f16x2
    from nerfdet_amd import conv3d
    conv3d.set_arithmetic(sys.argv[2])
if "no_amax_commit" in sys.argv[2:]:     # layer_times.py cfg2 f16x2 no_amax_commit: DESIGN.md 11.2's measurement (every fp16-pair launch takes its own amax pass)
    from nerfdet_amd import conv3d
    conv3d.measurement_mode(no_amax_commit=True)
for a in sys.argv[2:]:                    # layer_times.py cfg2 f16x2 nt_bytes=0 order2=0: library measurement knobs (ndet_measurement_knob)
    if "=" in a:
        from nerfdet_amd import _lib
        k, v = a.split("=")
        _lib.check(_lib.load().ndet_measurement_knob(k.encode(), int(v)), "measurement_knob")
if os.environ.get("F16_MIN_KSTEPS"):
    from nerfdet_amd import conv3d
    conv3d.F16_MIN_KSTEPS = int(os.environ["F16_MIN_KSTEPS"])
if os.environ.get("TIMING_NO_AMAX_OUT"):      # timing probe only (results are garbage): no launch commits its maximum, readers find zeroed slots
    from ctypes import c_void_p
    from nerfdet_amd import _lib
    _l = _lib.load()
    _f, _g = _l.ndet_conv_ndhwc_guarded, _l.ndet_conv_chain_guarded
    _l.ndet_conv_ndhwc_guarded = lambda *a: _f(*a[:22], c_void_p(0), *a[23:])
    _l.ndet_conv_chain_guarded = lambda *a: _g(*a[:23], c_void_p(0), *a[24:])
dev = torch.device("cuda")
det = bench.build_model(w).to(dev)
batch = bench.to_device(bench.synth_batch(w, 0), dev)
with torch.no_grad():
    for _ in range(4):
        det(return_loss=False, **batch)
    steps = 6
    recs = []
    for _ in range(steps):
        r = trace.Recorder()
        trace.recorder = r
        det(return_loss=False, **batch)
        trace.recorder = None
        torch.cuda.synchronize()
        recs.append([(n, a.elapsed_time(b), i) for n, a, b, i in r.spans])
n = len(recs[0])
assert all(len(r) == n for r in recs)
tot = 0.0
print(f"{'#':>3} {'kernel':38s} {'ms':>7s} {'GF':>7s} {'MB':>7s} {'TF/s':>6s} {'GB/s':>6s}")
for i in range(n):
    name, _, info = recs[0][i]
    ms = sorted(r[i][1] for r in recs)[steps // 2]
    tot += ms
    fl, by = info.get("flops", 0), info.get("bytes", 0)
    print(f"{i:3d} {name[:38]:38s} {ms:7.3f} {fl / 1e9:7.1f} {by / 1e6:7.1f} {fl / ms / 1e9 if fl else 0:6.1f} {by / ms / 1e6:6.0f}")
print("sum of spans", round(tot, 3), "ms")
from nerfdet_amd import conv3d as _c
print("amax fallbacks per step", _c.amax_fallbacks / (4 + steps))
