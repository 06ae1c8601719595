"""Same-box comparison of two checkouts' bench.py lines (event-timed spans, no profiler): python tools/diag/ab_bench_compare.py A.json B.json"""
import json
import sys

a, b = (json.load(open(p)) for p in sys.argv[1:3])
print("value", a["value"], b["value"], "median", a["median_ms"], b["median_ms"], "p10", a["p10_ms"], b["p10_ms"])
print("stages", a["stages_ms"], b["stages_ms"])
pa, pb = a["roofline_all_convolutions"]["per_kernel"], b["roofline_all_convolutions"]["per_kernel"]
tot = 0.0
for k in sorted(set(pa) | set(pb)):
    x, y = pa.get(k), pb.get(k)
    if x and y:
        d = (y["avg_launch_ms"] * y["launches_per_step"] - x["avg_launch_ms"] * x["launches_per_step"]) * 1e3
        tot += d
        print(f"{k:42s} {x['launches_per_step']:5.1f} x {x['avg_launch_ms'] * 1e3:7.1f} us | {y['launches_per_step']:5.1f} x {y['avg_launch_ms'] * 1e3:7.1f} us | {d:+7.1f} us/step")
    else:
        print(k, x, y)
print("sum of deltas", round(tot, 1), "us/step; conv totals", a["roofline_all_convolutions"]["total_ms_per_step"], b["roofline_all_convolutions"]["total_ms_per_step"])
for key in ("roofline_projection", "roofline_density_features"):
    print(key, a[key].get("median_launch_ms") or a[key].get("avg_launch_ms"), b[key].get("median_launch_ms") or b[key].get("avg_launch_ms"))
